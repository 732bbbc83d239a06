"""TEST INFRASTRUCTURE ONLY -- imports the reference's OWN hot-path code in this container.

The reference (OpenDrift v1.14.10, /root/reference) is pure Python but depends on
packages that are not installed here (pyproj, xarray, netCDF4, cartopy, ...).
`install()` puts MagicMock stand-ins for the non-arithmetic packages into
sys.modules and a small *functional* `pyproj` shim whose arithmetic is the C
oracle (oracle/geodesic.c, oracle/proj.c), then puts /root/reference on sys.path
so that e.g. `from opendrift.models.oceandrift import OceanDrift` works and
`OceanDrift.update()`, `Environment.get_environment`, `ReaderBlock.interpolate`
execute the reference's own NumPy/SciPy code.

This cannot travel to the GPU box (no /root/reference there): it is used only by
oracle/gen_golden.py to write tests/golden/*.npz, and by CPU-side tests that are
skipped when /root/reference is absent.
"""
import os
import re
import sys
import types
from unittest.mock import MagicMock

import numpy as np

REFERENCE = os.environ.get('ODR_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REFERENCE, 'opendrift'))


# ----------------------------------------------------------------------------
# functional pyproj shim
# ----------------------------------------------------------------------------
def _parse_proj4(s):
    d = {}
    for k, v in re.findall(r'\+([A-Za-z_0-9]+)(?:\s*=\s*(\S+))?', s):
        d[k] = v if v != '' else True
    return d


_ELLPS = {'WGS84': (6378137.0, 298.257223563), 'GRS80': (6378137.0, 298.257222101),
          'sphere': (6370997.0, 0.0)}


class _CRS:
    def __init__(self, is_geographic, srs):
        self.is_geographic = is_geographic
        self.srs = srs

    def to_proj4(self):
        return self.srs


class Proj:
    """Subset of pyproj.Proj: +proj=latlong/longlat, +proj=stere (any aspect, sphere / ellipsoid), +proj=merc, +proj=lcc,
    +proj=tmerc / utm, +proj=laea, +proj=ob_tran +o_proj=longlat.  Arithmetic = oracle/proj.c."""

    def __init__(self, projparams=None, **kwargs):
        from oracle import oracle as orc
        if isinstance(projparams, Proj):
            projparams = projparams.srs
        self.srs = str(projparams)
        p = _parse_proj4(self.srs)
        self._p = p
        name = p.get('proj', 'latlong')
        a, rf = _ELLPS[p.get('ellps', 'WGS84')] if 'ellps' in p or 'a' not in p and 'R' not in p else (None, None)
        if 'R' in p:
            a, rf = float(p['R']), 0.0
        if 'a' in p:
            a = float(p['a'])
            rf = rf if rf is not None else 0.0
            if 'rf' in p:
                rf = float(p['rf'])
            elif 'f' in p:
                rf = 0.0 if float(p['f']) == 0 else 1.0 / float(p['f'])
            elif 'b' in p:
                b = float(p['b'])
                rf = 0.0 if a == b else a / (a - b)
            elif 'e' in p:
                e = float(p['e'])
                rf = 0.0 if e == 0 else 1.0 / (1 - np.sqrt(1 - e * e))
            elif 'ellps' not in p:
                rf = 0.0
        f = 0.0 if not rf else 1.0 / rf
        es = f * (2 - f)
        if name in ('latlong', 'longlat', 'latlon', 'lonlat'):
            self.crs = _CRS(True, self.srs)
            self._orc = orc.make_proj(orc.PROJ_LATLONG)
        elif name == 'stere':
            lat0 = float(p.get('lat_0', 0))
            lon0 = float(p.get('lon_0', 0))
            lat_ts = float(p.get('lat_ts', 90.0 if abs(lat0) == 90 else 0.0))
            k0 = float(p.get('k_0', p.get('k', 1.0)))
            x0, y0 = float(p.get('x_0', 0)), float(p.get('y_0', 0))
            if abs(abs(lat0) - 90) < 1e-10:
                kind = orc.PROJ_STERE_POLAR
                if 'lat_ts' not in p:
                    lat_ts = 90.0
            elif lat0 == 0 and es == 0:
                kind = orc.PROJ_STERE_EQUIT_SPHERE
            else:                  # oblique, or equatorial on an ellipsoid (PROJ ignores +lat_ts for these aspects)
                kind = orc.PROJ_STERE_OBLIQUE
            self.crs = _CRS(False, self.srs)
            self._orc = orc.make_proj(kind, a=a, es=es, lat0=lat0, lon0=lon0, lat_ts=lat_ts, k0=k0,
                                      x0=x0, y0=y0)
        elif name in ('merc', 'lcc'):
            lat1 = float(p.get('lat_1', 0.0))
            lat2 = float(p['lat_2']) if 'lat_2' in p else lat1
            lat0 = float(p['lat_0']) if 'lat_0' in p else (lat1 if name == 'lcc' and 'lat_2' not in p else 0.0)
            self.crs = _CRS(False, self.srs)
            self._orc = orc.make_proj(orc.PROJ_MERC if name == 'merc' else orc.PROJ_LCC, a=a, es=es, lat0=lat0,
                                      lon0=float(p.get('lon_0', 0)), lat_ts=float(p.get('lat_ts', 0.0)),
                                      k0=float(p.get('k_0', p.get('k', 1.0))), x0=float(p.get('x_0', 0)),
                                      y0=float(p.get('y_0', 0)), lat1=lat1, lat2=lat2)
        elif name in ('tmerc', 'utm', 'laea'):
            lat0, lon0 = float(p.get('lat_0', 0)), float(p.get('lon_0', 0))
            k0, x0, y0 = float(p.get('k_0', p.get('k', 1.0))), float(p.get('x_0', 0)), float(p.get('y_0', 0))
            if name == 'utm':      # PROJ's utm set-up: the zone's meridian, k0 = 0.9996, false easting 500 km (northing 10 000 km south)
                zone = int(p['zone'])
                lat0, lon0, k0, x0, y0 = 0.0, 6.0 * zone - 183.0, 0.9996, 500000.0, (10000000.0 if 'south' in p else 0.0)
            self.crs = _CRS(False, self.srs)
            self._orc = orc.make_proj(orc.PROJ_LAEA if name == 'laea' else orc.PROJ_TMERC, a=a, es=es, lat0=lat0, lon0=lon0,
                                      k0=k0, x0=x0, y0=y0)
        elif name == 'ob_tran':
            # the rotated pole (+o_proj=longlat): pyproj reports such a CRS as geographic -- the reference relies on it
            # (variables.py:117-123, 800) -- and Proj.__call__ hands the rotated coordinates out / takes them in RADIANS
            # (the reference converts, :120-123, :136-138); Transformer.transform works in degrees on both sides
            if p.get('o_proj') not in ('longlat', 'latlong', 'latlon', 'lonlat') or 'to_meter' in p:
                raise NotImplementedError('pyproj shim: ' + self.srs)
            self.crs = _CRS(True, self.srs)
            self._ob_tran = True
            self._orc = orc.make_proj(orc.PROJ_OB_TRAN, lon0=float(p.get('lon_0', 0)), lat1=float(p.get('o_lat_p', 90.0)),
                                      lat2=float(p.get('o_lon_p', 0.0)))
        else:
            raise NotImplementedError('pyproj shim: +proj=%s' % name)

    def definition_string(self):
        return self.srs

    def __call__(self, x, y, inverse=False, **kw):
        from oracle import oracle as orc
        scalar = np.ndim(x) == 0
        shape = np.shape(x)
        xa = np.atleast_1d(np.asarray(x, dtype=np.float64)).ravel()
        ya = np.atleast_1d(np.asarray(y, dtype=np.float64)).ravel()
        if getattr(self, '_ob_tran', False) and not kw.get('_degrees'):
            if inverse:
                a, b = orc.proj_inv(self._orc, np.degrees(xa), np.degrees(ya))
            else:
                a, b = orc.proj_fwd(self._orc, xa, ya)
                a, b = np.radians(a), np.radians(b)
        elif inverse:
            a, b = orc.proj_inv(self._orc, xa, ya)
        else:
            a, b = orc.proj_fwd(self._orc, xa, ya)
        if scalar:
            return float(a[0]), float(b[0])
        return a.reshape(shape), b.reshape(shape)


class Transformer:
    """Transformer.from_proj between two shimmed Proj objects: inverse of the source,
    forward of the target (geographic coordinates pass through unchanged: PROJ's
    'ballpark' transformation between datum-less CRSs)."""

    def __init__(self, pf, pt):
        self.pf, self.pt = pf, pt

    @staticmethod
    def from_proj(proj_from, proj_to, **kw):
        pf = proj_from if isinstance(proj_from, Proj) else Proj(proj_from)
        pt = proj_to if isinstance(proj_to, Proj) else Proj(proj_to)
        return Transformer(pf, pt)

    def transform(self, x, y, **kw):
        lon, lat = self.pf(x, y, inverse=True, _degrees=True)
        return self.pt(lon, lat, _degrees=True)


class Geod:
    def __init__(self, ellps='WGS84', **kw):
        assert ellps == 'WGS84', 'pyproj shim: only WGS84 geodesics are restated'

    def fwd(self, lons, lats, az, dist, radians=False, **kw):
        from oracle import oracle as orc
        assert not radians
        scalar = np.ndim(lons) == 0
        lo, la, a2 = orc.geod_fwd(lons, lats, az, dist)
        back = a2 + 180.0
        back = back - 360.0 * np.round(back / 360.0)
        if scalar:
            return float(lo[0]), float(la[0]), float(back[0])
        return lo, la, back

    def inv(self, lons1, lats1, lons2, lats2, radians=False, **kw):
        from oracle import oracle as orc
        assert not radians
        scalar = np.ndim(lons1) == 0 and np.ndim(lons2) == 0
        az, s = orc.geod_inv(lons1, lats1, lons2, lats2)
        if scalar:
            return float(az[0]), float('nan'), float(s[0])
        return az, np.full_like(az, np.nan), s


def _make_pyproj():
    m = types.ModuleType('pyproj')
    m.Proj = Proj
    m.Geod = Geod
    m.Transformer = Transformer
    m.CRS = MagicMock()
    m.__version__ = '3.6.0-oracle-shim'
    m.__shim__ = True
    return m


_MOCKED = ['xarray', 'copernicusmarine', 'cartopy', 'cartopy.crs', 'cartopy.feature',
           'cartopy.io', 'cartopy.io.shapereader', 'cartopy.mpl', 'cartopy.mpl.geoaxes', 'cmocean',
           'roaring_landmask', 'geojson', 'coloredlogs', 'shapely', 'shapely.geometry', 'shapely.ops',
           'shapely.vectorized', 'netCDF4', 'dotenv', 'geopandas', 'nc_time_axis', 'trajan', 'cftime',
           'cfgrib', 'pykdtree', 'pykdtree.kdtree', 'utm', 'adios_db', 'adios_db.models', 'adios_db.models.oil',
           'adios_db.models.oil.oil', 'adios_db.computation', 'adios_db.computation.physical_properties',
           'adios_db.computation.gnome_oil', 'adios_db.computation.estimations', 'requests', 'earthaccess',
           'pyarrow', 'zarr', 'dask', 'h5netcdf']


def install():
    """Idempotent.  Returns True when the reference can be imported."""
    if not available():
        return False
    if 'pyproj' not in sys.modules or not getattr(sys.modules['pyproj'], '__shim__', False):
        sys.modules['pyproj'] = _make_pyproj()
    for name in _MOCKED:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = MagicMock()
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    import matplotlib
    matplotlib.use('Agg')
    import opendrift.readers.basereader  # noqa: F401  (must precede readers.interpolation: circular import)
    return True
