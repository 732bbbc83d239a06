"""Reader plugin surface -- host side.

Same contract as the reference's readers (opendrift/readers/basereader/): a reader sets
`proj4`, `xmin/xmax/ymin/ymax`, `variables`, `start_time/end_time/time_step` (or `times`),
optionally `z`, and implements

    get_variables(requested_variables, time=None, x=None, y=None, z=None) -> dict

returning, for a StructuredReader, one block {'x','y','z','time', var: [ny,nx] | [nz,ny,nx]}
(basereader/structured.py:125-147), or, for a ContinuousReader, arrays at the exact positions
(basereader/continuous.py:20-29).  `get_variables` stays on the host, so existing reader code
works unmodified; the interception point is the ReaderBlock: when a new time level is needed
its block is uploaded once (interpolation/structured.py:15-94 becomes odr_block_upload) and
every ReaderBlock.interpolate becomes a device gather.  Analytic readers that the device
evaluates in closed form (constant, double gyre, oscillating) are recognised by `device_kind`.
"""
from datetime import datetime, timedelta

import threading

import numpy as np

from . import projection


def _epoch(t):
    return (t - datetime(1970, 1, 1)).total_seconds()


class BaseReader:
    """Attributes of basereader/__init__.py:107-118 + variables.ReaderDomain (variables.py:20-46)."""
    name = 'reader'
    proj4 = '+proj=latlong'
    xmin = xmax = ymin = ymax = None
    zmin, zmax = -np.inf, np.inf
    start_time = end_time = time_step = times = None
    always_valid = False
    z = None
    variables = []
    device_kind = None   # 'constant' | 'double_gyre' | 'oscillating' | 'landmask' | None (gridded)

    def __init__(self):
        self.proj = projection.Proj(self.proj4)
        self.is_lazy = False
        self.number_of_fails = 0
        if self.start_time is not None and self.time_step is not None and self.times is None:
            n = int(round((self.end_time - self.start_time).total_seconds() / self.time_step.total_seconds())) + 1
            self.times = [self.start_time + k * self.time_step for k in range(n)]

    # ---- variables.py:111-143
    def lonlat2xy(self, lon, lat):
        return self.proj(lon, lat)

    def xy2lonlat(self, x, y):
        return self.proj(x, y, inverse=True)

    def covers_time(self, time):   # variables.py:392-400
        if self.always_valid or self.start_time is None:
            return True
        return self.start_time <= time <= self.end_time

    def nearest_time(self, time):  # variables.py:402-443 -> indices of the bracketing levels
        if self.times is None or len(self.times) == 1:
            return 0, 0
        import bisect
        ib = max(0, bisect.bisect_right(self.times, time) - 1)
        ia = ib if self.times[ib] == time else min(ib + 1, len(self.times) - 1)
        return ib, ia

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        raise NotImplementedError


class ContinuousReader(BaseReader):
    pass


class StructuredReader(BaseReader):
    """basereader/structured.py.  A reader that sets no projection (`proj4` None or 'fakeproj') but has 2D `lon` /
    `lat` node arrays is 'unprojected' (structured.py:44-113): x/y are pixel indices, xy2lonlat is bilinear in the
    node arrays (:421-436) and lonlat2xy is the Delaunay lookup (:438-472) -- which runs on the device
    (odr_source_grid_curvilinear); on the host it is available once the reader is bound to a device context."""
    projected = True

    def __init__(self):
        if self.proj4 is None or self.proj4 == 'fakeproj':
            self.projected = False
            self.lon = np.ascontiguousarray(self.lon, dtype=np.float64)
            self.lat = np.ascontiguousarray(self.lat, dtype=np.float64)
            self.proj4 = 'None'
            self.xmin = self.ymin = 0.
            self.delta_x = self.delta_y = 1.
            self.xmax = self.lon.shape[1] - 1
            self.ymax = self.lon.shape[0] - 1
            self.numx, self.numy = self.xmax, self.ymax
            self.x = np.arange(0, self.xmax + 1)
            self.y = np.arange(0, self.ymax + 1)
            self._device_lookup = None
            proj4, self.proj4 = self.proj4, '+proj=latlong'
            super().__init__()
            self.proj4, self.proj = proj4, None
        else:
            super().__init__()

    def xy2lonlat(self, x, y):
        if self.projected:
            return super().xy2lonlat(x, y)
        x, y = np.array(x, dtype=np.float64, ndmin=1), np.array(y, dtype=np.float64, ndmin=1)
        bad = (x < self.xmin) | (x > self.xmax) | (y < self.ymin) | ~np.isfinite(x) | ~np.isfinite(y)
        xc, yc = np.where(bad, 0.0, x), np.clip(np.where(bad, 0.0, y), 0, self.ymax)   # mode='nearest' beyond ymax
        i0 = np.minimum(xc.astype(np.int64), self.lon.shape[1] - 2)
        j0 = np.minimum(yc.astype(np.int64), self.lon.shape[0] - 2)
        tx, ty = xc - i0, yc - j0

        def bil(a):
            return ((1 - ty) * ((1 - tx) * a[j0, i0] + tx * a[j0, i0 + 1]) +
                    ty * ((1 - tx) * a[j0 + 1, i0] + tx * a[j0 + 1, i0 + 1]))
        lon, lat = bil(self.lon), bil(self.lat)
        lon[bad] = np.nan
        lat[bad] = np.nan
        return lon, lat

    def lonlat2xy(self, lon, lat):
        if self.projected:
            return super().lonlat2xy(lon, lat)
        if self._device_lookup is None:
            raise RuntimeError('reader %s has no projection: lonlat2xy is the device lookup, available once the '
                               'reader is part of a simulation (add_reader + first step)' % self.name)
        return self._device_lookup(lon, lat)


class ConstantReader(ContinuousReader):
    """reader_constant.Reader (readers/reader_constant.py): the same value everywhere."""
    device_kind = 'constant'

    def __init__(self, parameter_value_map=None, **kwargs):
        m = dict(parameter_value_map or kwargs)
        self._element_ID = None
        if 'element_ID' in m:
            # values per element (reader_constant.py:42-58,70-80): the listed IDs get their values, every other element
            # NaN -> next reader / fallback.  Evaluated on the host at every sample (no closed form for the device).
            self.device_kind = None
            self._element_ID = True              # set to the IDs of the call by the model (environment.py:621-623)
            self._ids = np.atleast_1d(np.asarray(m.pop('element_ID'))).astype(np.int64)
            self._parameter_value_map = {k: np.atleast_1d(np.asarray(v, dtype=np.float64)) for k, v in m.items()}
        else:
            self._parameter_value_map = {k: float(np.atleast_1d(v)[0]) for k, v in m.items()}
        self.variables = list(self._parameter_value_map)
        self.proj4 = '+proj=latlong'
        self.xmin, self.xmax, self.ymin, self.ymax = -180, 180, -90, 90
        self.name = 'constant_reader'
        super().__init__()

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        out = {'time': time, 'x': x, 'y': y, 'z': z}
        for v in requested_variables:
            value = self._parameter_value_map[v]
            if self._element_ID is None:
                out[v] = value * np.ones(np.shape(x))
                continue
            ids = np.atleast_1d(self._element_ID)              # the IDs of this call's elements
            a = np.full(np.shape(x), np.nan)
            pos = {int(i): k for k, i in enumerate(self._ids)}
            hit = np.array([int(i) in pos for i in ids], dtype=bool)
            if hit.any():
                a[hit] = value[0] if len(value) == 1 else value[[pos[int(i)] for i in ids[hit]]]
            out[v] = a
        return out


class LandmaskRasterReader(ContinuousReader):
    """reader_global_landmask.Reader (readers/reader_global_landmask.py:201-255) over a lon/lat raster handed in by the
    caller: land_binary_mask exactly at the element positions.  The reference's class wraps the GSHHG dataset through
    roaring_landmask, which is not part of this repository; its contains_many(x, y) is what `cells` replaces:
    cells[iy, ix] != 0 is land, cell (ix, iy) covers lon0 + [ix, ix+1) dlon x lat0 + [iy, iy+1) dlat, ocean outside.
    It is also the landmask `coastline_crossing` searches when general:coastline_approximation_precision is set."""
    device_kind = 'landmask'
    name = 'global_landmask'
    variables = ['land_binary_mask']

    def __init__(self, lon0, lat0, dlon, dlat, cells):
        self.lon0, self.lat0, self.dlon, self.dlat = float(lon0), float(lat0), float(dlon), float(dlat)
        self.cells = np.ascontiguousarray(np.asarray(cells) != 0, dtype=np.uint8)
        self.proj4 = '+proj=latlong'
        self.xmin, self.xmax, self.ymin, self.ymax = -180, 180, -90, 90
        self.always_valid = True
        super().__init__()

    def contains_many(self, x, y):
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        ix = np.floor((x - self.lon0) / self.dlon).astype(np.int64)
        iy = np.floor((y - self.lat0) / self.dlat).astype(np.int64)
        ny, nx = self.cells.shape
        ok = (ix >= 0) & (ix < nx) & (iy >= 0) & (iy < ny)
        out = np.zeros(x.shape, bool)
        out[ok] = self.cells[iy[ok], ix[ok]] != 0
        return out

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        x = np.mod(np.asarray(x, dtype=np.float64) + 180, 360) - 180      # modulate_longitude
        return {'time': time, 'x': x, 'y': y, 'z': z, 'land_binary_mask': self.contains_many(x, y)}


class FailingReader(ContinuousReader):
    """reader_failing.Reader (readers/reader_failing.py:20-43): raises in every call, for testing the quarantine of
    readers after readers:max_number_of_fails failures."""

    def __init__(self):
        self.variables = ['x_wind', 'y_wind']
        self.proj4 = '+proj=latlong'
        self.xmin, self.xmax, self.ymin, self.ymax = -180, 180, -90, 90
        self.name = 'failing_reader'
        super().__init__()

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        raise ValueError('Failing reader, for testing only.')


class DoubleGyreReader(ContinuousReader):
    """reader_double_gyre.Reader (readers/reader_double_gyre.py:24-79)."""
    device_kind = 'double_gyre'

    def __init__(self, initial_time=datetime(2000, 1, 1), epsilon=0.1, omega=0.628, A=0.25,
                 proj4='+proj=stere +lat_0=0 +lon_0=0 +lat_ts=0 +units=m +a=6.371e+06 +e=0 +no_defs'):
        self.name = 'double_gyre'
        self.proj4 = proj4
        self.xmin, self.xmax, self.ymin, self.ymax = 0., 2., 0., 1.
        self.A, self.epsilon, self.omega, self.initial_time = A, epsilon, omega, initial_time
        self.variables = ['x_sea_water_velocity', 'y_sea_water_velocity']
        super().__init__()

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        t = (time - self.initial_time).total_seconds()
        a = self.epsilon * np.sin(self.omega * t)
        b = 1 - 2 * self.epsilon * np.sin(self.omega * t)
        f = a * x * x + b * x
        dfdx = 2 * a * x + b
        return {'x_sea_water_velocity': -np.pi * self.A * np.sin(np.pi * f) * np.cos(np.pi * y),
                'y_sea_water_velocity': np.pi * self.A * np.cos(np.pi * f) * np.sin(np.pi * y) * dfdx,
                'land_binary_mask': np.zeros(np.shape(x)), 'time': time, 'x': x, 'y': y}


class OscillatingReader(ContinuousReader):
    """reader_oscillating.Reader (readers/reader_oscillating.py:23-59)."""
    device_kind = 'oscillating'

    def __init__(self, variable, amplitude, period=timedelta(hours=24), phase=0, zero_time=datetime(2017, 1, 1)):
        self.variables = [variable]
        self.amplitude, self.period_seconds, self.zero_time = amplitude, period.total_seconds(), zero_time
        self.proj4 = '+proj=latlong +datum=WGS84'
        self.xmin, self.xmax, self.ymin, self.ymax = -180, 180, -90, 90
        self.name = 'oscillating_reader'
        super().__init__()

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        phase = ((time - self.zero_time).total_seconds() / self.period_seconds) * np.pi
        return {'time': time, 'x': x, 'y': y, 'z': z,
                self.variables[0]: self.amplitude * np.sin(phase) * np.ones(np.shape(x))}


# the x / y components the reference rotates from the reader's axes to east / north (basereader/consts.py:27-36)
VECTOR_PAIRS_XY = [('x_wind', 'y_wind'), ('sea_ice_x_velocity', 'sea_ice_y_velocity'),
                   ('x_sea_water_velocity', 'y_sea_water_velocity'),
                   ('sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity')]


def wgs84_forward_azimuth(lon1, lat1, lon2, lat2):
    """Azimuth (degrees clockwise from north) at point 1 of the WGS84 geodesic to point 2 -- what pyproj.Geod(ellps='WGS84')
    .inv(...)[0] gives the reference in rotate_vectors (variables.py:94-97) -- by Vincenty's inverse iteration (short lines
    between neighbouring grid nodes: converges in a few rounds)."""
    a, f = 6378137.0, 1 / 298.257223563
    b = (1 - f) * a
    L = np.radians(np.asarray(lon2, dtype=np.float64) - np.asarray(lon1, dtype=np.float64))
    L = (L + np.pi) % (2 * np.pi) - np.pi
    U1 = np.arctan((1 - f) * np.tan(np.radians(np.asarray(lat1, dtype=np.float64))))
    U2 = np.arctan((1 - f) * np.tan(np.radians(np.asarray(lat2, dtype=np.float64))))
    sU1, cU1, sU2, cU2 = np.sin(U1), np.cos(U1), np.sin(U2), np.cos(U2)
    lam = L.copy()
    for _ in range(30):
        sl, cl = np.sin(lam), np.cos(lam)
        ss = np.hypot(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl)
        cs = sU1 * sU2 + cU1 * cU2 * cl
        sig = np.arctan2(ss, cs)
        with np.errstate(invalid='ignore', divide='ignore'):
            sa = np.where(ss > 0, cU1 * cU2 * sl / ss, 0.0)
            c2a = 1 - sa * sa
            c2m = np.where(c2a > 0, cs - 2 * sU1 * sU2 / c2a, 0.0)
        Cc = f / 16 * c2a * (4 + f * (4 - 3 * c2a))
        new = L + (1 - Cc) * f * sa * (sig + Cc * ss * (c2m + Cc * cs * (-1 + 2 * c2m * c2m)))
        done = np.max(np.abs(new - lam)) < 1e-14
        lam = new
        if done:
            break
    return np.degrees(np.arctan2(cU2 * np.sin(lam), cU1 * sU2 - sU1 * cU2 * np.cos(lam)))


def node_y_azimuth(lon2d, lat2d):
    """Azimuth of the mesh's +y direction at every node: from the node to its neighbour one row up (the last row: the
    arrival azimuth of the line from the row below), on WGS84."""
    lon2d, lat2d = np.asarray(lon2d, dtype=np.float64), np.asarray(lat2d, dtype=np.float64)
    az = np.empty(lon2d.shape)
    az[:-1] = wgs84_forward_azimuth(lon2d[:-1], lat2d[:-1], lon2d[1:], lat2d[1:])
    az[-1] = (wgs84_forward_azimuth(lon2d[-1], lat2d[-1], lon2d[-2], lat2d[-2]) + 180.0 + 180.0) % 360.0 - 180.0
    return az


class GridReader(StructuredReader):
    """In-memory StructuredReader with time levels and optional z levels (the shape of
    reader_constant_2d.py:20-49 / reader_netCDF_CF_generic / reader_ROMS_native output blocks).
    arrays: {variable: [nt, ny, nx] or [nt, nz, ny, nx], or a list of such arrays = ensemble members}.

    A projection the device has no closed form for (rotated pole `ob_tran`, utm, laea, ... -- the reference hands any proj4
    to pyproj, variables.py:111-143): give the 2-D node coordinates `lon`, `lat` that such files carry.  The reader is then
    served like one WITHOUT projection (structured.py:44-113): positions through the node-array lookup on the device
    (odr_source_grid_curvilinear), and the vector pairs of every block rotated to east / north at the nodes by the azimuth
    of the mesh's y axis (rotate_vectors, variables.py:59-108, does it after the interpolation, at the element: the
    difference is second order in the turn of the axes across one cell -- DESIGN.md 9)."""
    # A reference file reader hands out the z levels ASKED FOR (one more on either side + verticalbuffer,
    # reader_netCDF_CF_generic.py:414-423, basereader/__init__.py:52); the device block of this reader always holds every level,
    # and OceanDrift applies that cut to the diffusivity columns where it matters (drift:truncate_ocean_model_below_m).  A reader
    # standing for one that ignores the depth range asked of it sets always_delivers_all_levels = True.
    verticalbuffer = 1
    always_delivers_all_levels = False

    def __new__(cls, x, y, times, arrays, z=None, proj4='+proj=latlong', name='grid_reader', lon=None, lat=None):
        if cls is GridReader and proj4 is not None:
            try:
                projection.parse_proj4(proj4)
            except NotImplementedError:
                if lon is None or lat is None:
                    raise NotImplementedError(
                        'projection "%s" has no closed form on the device (latlong, stere, merc, lcc have): pass the 2-D node '
                        'coordinates lon=, lat= of the grid and the reader is served through the node lookup' % proj4)
                r = object.__new__(NodeLookupGridReader)     # not a GridReader instance: Python does not call __init__ on it
                r.__init__(x, y, times, arrays, z=z, proj4=proj4, name=name, lon=lon, lat=lat)
                return r
        return object.__new__(cls)

    def __init__(self, x, y, times, arrays, z=None, proj4='+proj=latlong', name='grid_reader', lon=None, lat=None):
        self.proj4, self.name = proj4, name
        self.x, self.y, self.z = np.asarray(x), np.asarray(y), (None if z is None else np.asarray(z, dtype=np.float64))
        self.xmin, self.xmax = float(self.x.min()), float(self.x.max())
        self.ymin, self.ymax = float(self.y.min()), float(self.y.max())
        self.times = list(times)
        self.start_time, self.end_time = self.times[0], self.times[-1]
        self.time_step = (self.times[1] - self.times[0]) if len(self.times) > 1 else None
        self.arrays = arrays
        # 2-D variables that are the same array at every time level (sea floor depth, land mask): declared once here, so that
        # the device reads them at one of the two bracketing levels (odr_block_set_content_ids) without comparing levels
        def same(a, b):
            return np.array_equal(a.view(np.uint32), b.view(np.uint32)) if a.dtype == np.float32 else np.array_equal(a, b)
        self.static_variables = [v for v, a in arrays.items() if isinstance(a, np.ndarray) and a.ndim == 3 and a.shape[0] > 1 and
                                 all(same(a[0], a[k]) for k in range(1, a.shape[0]))]
        self.variables = list(arrays)
        super().__init__()

    buffer = 2      # pixels around the requested positions (StructuredReader.set_buffer_size, structured.py:125-147)

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        """The block covering the requested positions x, y plus `buffer` pixels (as reader_netCDF_CF_generic /
        reader_ROMS_native cut it, reader_netCDF_CF_generic.py:404-470); without positions the whole domain."""
        it = self.times.index(time)
        jy, ix = slice(None), slice(None)
        if x is not None and y is not None and np.size(x) > 0 and self.x.ndim == 1 and len(self.x) > 1:
            def window(c, q):
                asc = c[-1] > c[0]
                ca = c if asc else c[::-1]
                lo = int(np.clip(np.searchsorted(ca, np.nanmin(q), side='right') - 1 - self.buffer, 0, len(c) - 1))
                hi = int(np.clip(np.searchsorted(ca, np.nanmax(q), side='left') + self.buffer, 0, len(c) - 1))
                return slice(lo, hi + 1) if asc else slice(len(c) - 1 - hi, len(c) - lo)
            ix, jy = window(np.asarray(self.x, dtype=np.float64), np.asarray(x, dtype=np.float64)), \
                window(np.asarray(self.y, dtype=np.float64), np.asarray(y, dtype=np.float64))
        out = {'x': self.x[ix], 'y': self.y[jy], 'time': time, 'z': self.z if self.z is not None else 0}
        for v in requested_variables:
            a = self.arrays[v]
            if isinstance(a, (list, tuple)):      # ensemble data: a list of member arrays (structured.py:125-147)
                out[v] = [m[it][..., jy, ix] for m in a]
            else:
                out[v] = a[it][..., jy, ix]
        return out


class CurvilinearGridReader(StructuredReader):
    """In-memory StructuredReader on a curvilinear mesh given by 2D lon/lat node arrays and NO projection -- the
    situation of reader_ROMS_native / reader_netCDF_CF_generic files that only carry lon(y,x), lat(y,x)
    (structured.py:44-113).  arrays: {variable: [nt, ny, nx] or [nt, nz, ny, nx]}; vector components are taken as
    east/north (the reference's rotation for such readers is by the azimuth of due north, i.e. none)."""

    def __init__(self, lon, lat, times, arrays, z=None, name='curvilinear_grid_reader'):
        self.proj4, self.name = None, name
        self.lon, self.lat = lon, lat
        self.z = None if z is None else np.asarray(z, dtype=np.float64)
        self.times = list(times)
        self.start_time, self.end_time = self.times[0], self.times[-1]
        self.time_step = (self.times[1] - self.times[0]) if len(self.times) > 1 else None
        self.arrays = arrays
        self.variables = list(arrays)
        super().__init__()

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        it = self.times.index(time)
        out = {'x': self.x, 'y': self.y, 'time': time, 'z': self.z if self.z is not None else 0}
        for v in requested_variables:
            out[v] = self.arrays[v][it]
        return out


class NodeLookupGridReader(CurvilinearGridReader):
    """What GridReader(...) returns for a proj4 string the device cannot evaluate: the same arrays behind 2-D node
    coordinates, vector pairs rotated to east / north per block (see GridReader)."""

    def __init__(self, x, y, times, arrays, z=None, proj4=None, name='grid_reader', lon=None, lat=None):
        lon, lat = np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64)
        if lon.shape != (len(y), len(x)) or lat.shape != lon.shape:
            raise ValueError('lon / lat must be [len(y), len(x)] node arrays')
        self.native_proj4 = proj4
        self.native_x, self.native_y = np.asarray(x), np.asarray(y)
        super().__init__(lon, lat, times, arrays, z=z, name=name)
        rot = -np.radians(node_y_azimuth(lon, lat))          # rot_angle_rad = -rot_angle_vectors_rad (variables.py:103)
        self._cos, self._sin = np.cos(rot), np.sin(rot)
        self._pairs = [(a, b) for a, b in VECTOR_PAIRS_XY if a in self.arrays and b in self.arrays]

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        out = super().get_variables(requested_variables, time, x, y, z)
        for a, b in self._pairs:
            if a in out and b in out:
                def rotate(u, v):
                    # masked cells (land, missing data of a netCDF block) become NaN as in the plain upload path
                    # (np.ma.filled); np.asarray() would hand the fill value 9.96921e36 on as a velocity
                    u = np.ma.filled(np.ma.asarray(u).astype(np.float32), np.nan)
                    v = np.ma.filled(np.ma.asarray(v).astype(np.float32), np.nan)
                    return ((u * self._cos - v * self._sin).astype(np.float32),      # variables.py:104-107
                            (u * self._sin + v * self._cos).astype(np.float32))
                if isinstance(out[a], (list, tuple)):      # ensemble members
                    rot = [rotate(u, v) for u, v in zip(out[a], out[b])]
                    out[a], out[b] = [r[0] for r in rot], [r[1] for r in rot]
                else:
                    out[a], out[b] = rotate(out[a], out[b])
            elif a in out or b in out:
                raise ValueError('reader %s: %s and %s are rotated together -- request both' % (self.name, a, b))
        return out


ROMS_ZLEVELS = np.array([0, -.5, -1, -3, -5, -10, -25, -50, -75, -100, -150, -200, -250, -300, -400, -500, -600, -700,
                         -800, -900, -1000, -1500, -2000, -2500, -3000, -3500, -4000, -4500, -5000, -5500, -6000, -6500,
                         -7000, -7500, -8000], dtype=np.float64)   # reader_ROMS_native.py:134-138


class SigmaGridReader(StructuredReader):
    """In-memory reader with the vertical structure of reader_ROMS_native: 3-D variables live on N terrain-following
    s-levels (arrays3d: {variable: [nt, N, ny, nx]}), 2-D ones on the grid (arrays2d: {variable: [nt, ny, nx]}); `h`
    bottom depth, `hc`, `Cs_r`, `Vtransform` as in a ROMS file.  Blocks handed to the model are on the reader's fixed
    z levels (`zlevels`, default: the reference's list down to the deepest node), regridded per time level as
    reader_ROMS_native.get_variables does (:617-684) -- on the device (`s_levels = True` tells the binding that
    get_variables returns s-level arrays plus the target levels)."""
    s_levels = True

    def __init__(self, x, y, times, arrays3d, arrays2d, h, hc, Cs_r, Vtransform=2, zlevels=None, proj4='+proj=latlong',
                 name='sigma_grid_reader'):
        self.proj4, self.name = proj4, name
        self.x, self.y = np.asarray(x), np.asarray(y)
        self.h = np.ascontiguousarray(h, dtype=np.float64)
        self.hc, self.Cs_r, self.Vtransform = float(hc), np.asarray(Cs_r, dtype=np.float64), int(Vtransform)
        if zlevels is None:
            deeper = np.nonzero(ROMS_ZLEVELS < -float(self.h.max()))[0]
            zlevels = ROMS_ZLEVELS[:deeper[0] + 1] if len(deeper) else ROMS_ZLEVELS
        self.z = np.asarray(zlevels, dtype=np.float64)
        self.xmin, self.xmax = float(self.x.min()), float(self.x.max())
        self.ymin, self.ymax = float(self.y.min()), float(self.y.max())
        self.times = list(times)
        self.start_time, self.end_time = self.times[0], self.times[-1]
        self.time_step = (self.times[1] - self.times[0]) if len(self.times) > 1 else None
        self.arrays3d, self.arrays2d = dict(arrays3d), dict(arrays2d)
        self.variables = list(self.arrays3d) + list(self.arrays2d)
        super().__init__()

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        it = self.times.index(time)
        out = {'x': self.x, 'y': self.y, 'time': time, 'z': self.z,
               's_level_variables': [v for v in requested_variables if v in self.arrays3d]}
        for v in requested_variables:
            out[v] = self.arrays3d[v][it] if v in self.arrays3d else self.arrays2d[v][it]
        return out


def _dev(a):
    """a block array as Context.upload_block_device wants it: the device pointer of a CUDA tensor, else the host array"""
    return int(a.data_ptr()) if hasattr(a, 'is_cuda') and a.is_cuda else a


class ReaderLevelsError(RuntimeError):
    """The time levels one step needs do not fit the device slots: a configuration error of the run, NOT a reader
    failure (the model's reader-failure handling must not swallow it and fall back to constants silently)."""


_READER_IO_LOCK = threading.Lock()   # one get_variables at a time, whichever thread: netCDF / HDF5 builds are rarely thread-safe


class _ShapeOnly:
    """A variable of a reader level on a rank that does not hold the arrays (they arrive by odr_block_broadcast)."""

    def __init__(self, shape):
        self.shape = tuple(int(n) for n in shape)


class ReadAhead:
    """Rank 0 of a sharded run owns the host Reader: every time level the other ranks receive passes through ITS
    get_variables.  Called inline it stops rank 0's step loop for the duration of the file read, and -- one collective per
    step -- every other rank with it.  This runs the read of the level that comes NEXT on a worker thread as soon as a
    level has been handed out; when the level is due (one period later: _prefetch_dist starts its broadcast) the block is
    there.  The reference reads inline in its single process (basereader/structured.py:121-170); the block is the same
    object either way.  A read-ahead for another level or window than the one asked for is waited for and dropped."""

    def __init__(self, reader, variables):
        self.reader, self.variables = reader, list(variables)
        self._pool, self._fut, self._key = None, None, None
        self.hits = self.misses = 0
        self.worker_s = 0.0          # time the worker spent inside get_variables (off the step loop)

    @staticmethod
    def _key_of(k, x, y):
        return (int(k), None if x is None else (np.asarray(x).tobytes(), np.asarray(y).tobytes()))

    def _get(self, k, x, y, timed=False):
        import time as _time
        r = self.reader
        t = r.times[k] if r.times is not None else None
        t0 = _time.perf_counter()
        with _READER_IO_LOCK:
            block = r.get_variables(self.variables, t, x, y, np.array([0.0]))
        if timed:
            self.worker_s += _time.perf_counter() - t0
        return block

    def start(self, k, x, y):
        """Begin reading level k on the worker (returns at once)."""
        if self._fut is not None:
            return
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='odr-reader')
        self._key = self._key_of(k, x, y)
        self._fut = self._pool.submit(self._get, k, x, y, True)

    def read(self, k, x, y):
        """The block of level k: the worker's if it was read ahead (its exception is raised here, where an inline read
        would have raised it), an inline read otherwise."""
        fut, key = self._fut, self._key
        self._fut = self._key = None
        if fut is not None:
            if key == self._key_of(k, x, y):
                self.hits += 1
                return fut.result()
            try:
                fut.result()
            except Exception:      # noqa: BLE001 -- of a level nobody asked for
                pass
        self.misses += 1
        return self._get(k, x, y)

    def close(self):
        """Waits for a read in flight (its exception, of a level nobody will ask for, is logged, not raised) and ends the worker."""
        fut, self._fut, self._key = self._fut, None, None
        if fut is not None:
            try:
                fut.result()
            except Exception as e:      # noqa: BLE001
                import logging
                logging.getLogger(__name__).debug('read-ahead of a level that is no longer wanted failed: %r', e)
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None


class DeviceReaderBinding:
    """Device image of one reader: constant/analytic source, or a grid source whose time levels
    (ReaderBlocks) are uploaded on demand.  Stands where StructuredReader keeps
    var_block_before/after (structured.py:121-123)."""
    NSLOTS = 6        # = MAXLEVELS of the device source: t0, t0 + dt/2 and t0 + dt may each sit between two different levels
    PREFETCH = True   # stage the next time level on the upload stream while the current one is in use

    def __init__(self, ctx, reader, variables=None):
        self.ctx, self.reader = ctx, reader
        self.variables = [v for v in (variables or reader.variables)]
        self.slots = {}      # time index -> slot
        self.staged = {}     # time index -> slot uploaded asynchronously, not yet committed
        self.prefetch = self.PREFETCH
        self._pinned = False
        from .distributed import env_world
        self.rank, _, self.world = env_world()
        self.stall_s = 0.0             # host time spent waiting for a level that was due (run().timing reports the sum)
        self._dist_pre = {}            # sharded run: time index -> (tensors, works) of a level whose broadcast is under way
        self._ahead, self._ahead_last = None, None   # sharded run, rank 0: the next level read on a worker thread (ReadAhead)
        kind = getattr(reader, 'device_kind', None)
        # a ContinuousReader the device has no closed form for (a user's analytic or point-wise reader,
        # basereader/continuous.py:20-46): evaluated on the host at the element positions, values uploaded
        self.host_eval = kind is None and isinstance(reader, ContinuousReader)
        if kind == 'constant':
            self.sid = ctx.add_constant({v: reader._parameter_value_map[v] for v in self.variables})
        elif kind == 'double_gyre':
            self.sid = ctx.add_double_gyre(A=reader.A, epsilon=reader.epsilon, omega=reader.omega,
                                           t0=_epoch(reader.initial_time))
            self.variables = ['x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask']
        elif kind == 'landmask':
            self.sid = ctx.add_landmask(reader.lon0, reader.lat0, reader.dlon, reader.dlat, reader.cells)
        elif kind == 'oscillating':
            self.sid = ctx.add_oscillating(reader.variables[0], reader.amplitude, reader.period_seconds,
                                           _epoch(reader.zero_time))
        else:
            self.sid = None  # created with the first block (the grid comes with it)
        if kind and reader.start_time is not None:
            ctx.set_time_coverage(self.sid, _epoch(reader.start_time), _epoch(reader.end_time), reader.always_valid)

    def _static_ids(self):
        """{variable: content id} of the 2-D variables the reader declares time-invariant (`static_variables`): the same id
        for every level of this binding (a re-cut window is a new device source: its levels are all uploaded again)."""
        sv = [v for v in getattr(self.reader, 'static_variables', ()) if v in self.variables]
        if not sv:
            return None
        base = (id(self) & 0xffffff) * 4096
        return {v: base + 1 + self.variables.index(v) for v in sv}

    def is_grid(self):
        return getattr(self.reader, 'device_kind', None) is None and not self.host_eval

    def evaluate_on_host(self, variables, time, lon, lat, z, element_ID=None):
        """ContinuousReader._get_variables_interpolated_ (continuous.py:31-46): the reader's values exactly at the element
        positions; NaN where it does not cover (position or time).  A reader that maps values to element IDs
        (`_element_ID`, environment.py:621-623) is told the IDs of the positions it receives."""
        r = self.reader
        n = len(lon)
        out = {v: np.full(n, np.nan, np.float32) for v in variables}
        if not r.covers_time(time):
            return out
        x, y = r.lonlat2xy(lon, lat)
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        ok = np.ones(n, bool)
        if r.xmin is not None:
            ok &= (x >= r.xmin) & (x <= r.xmax) & (y >= r.ymin) & (y <= r.ymax)
        ok &= (z >= r.zmin) & (z <= r.zmax)
        if ok.any():
            if getattr(r, '_element_ID', None) is not None and element_ID is not None:
                r._element_ID = np.asarray(element_ID)[ok]
            res = r.get_variables(list(variables), time, x[ok], y[ok], z[ok])
            for v in variables:
                out[v][ok] = np.asarray(np.ma.filled(res[v], np.nan), dtype=np.float32) * np.ones(int(ok.sum()), np.float32)
        return out

    def set_extent(self, lonlat_box):
        """Reader.prepare(extent, ...) (basereader/__init__.py, called from Environment.finalize :139-211): the lon / lat
        box the simulation can reach.  Blocks are then requested for that box only (+ the reader's buffer) instead of
        the reader's whole domain -- what makes a basin-scale model with many levels fit: every resident time level is
        cut to the same window.  Readers whose window would be most of their domain, readers without projection
        (whole-mesh lookup) and s-level readers keep whole-domain blocks."""
        r = self.reader
        self.extent = None
        self.extent_lonlat = None
        if not self.is_grid() or not getattr(r, 'projected', True) or getattr(r, 's_levels', False) or lonlat_box is None:
            return
        if not (hasattr(r, 'x') and np.ndim(r.x) == 1 and len(r.x) > 8 and len(r.y) > 8):
            return
        lo0, la0, lo1, la1 = [float(v) for v in lonlat_box]
        self.extent_lonlat = (lo0, la0, lo1, la1)
        x, y = self._box_outline_xy(lo0, la0, lo1, la1)
        if not (np.isfinite(x).all() and np.isfinite(y).all()):
            return
        xs, ys = np.asarray(r.x, dtype=np.float64), np.asarray(r.y, dtype=np.float64)
        fx = (min(x.max(), xs.max()) - max(x.min(), xs.min())) / (xs.max() - xs.min())
        fy = (min(y.max(), ys.max()) - max(y.min(), ys.min())) / (ys.max() - ys.min())
        if fx <= 0 or fy <= 0 or fx * fy > 0.6:
            return        # no overlap (nothing to cut) or most of the domain anyway
        self.extent = (np.array([x.min(), x.max()]), np.array([y.min(), y.max()]))

    def _box_outline_xy(self, lo0, la0, lo1, la1):
        """The outline of a lon / lat box in the reader's coordinates, 33 points per edge: on a polar-stereographic or
        Lambert grid the edges of such a box bulge in x / y, its corners alone underestimate the footprint."""
        t = np.linspace(0, 1, 33)
        lon = np.concatenate([lo0 + (lo1 - lo0) * t, lo0 + (lo1 - lo0) * t, np.full(33, lo0), np.full(33, lo1)])
        lat = np.concatenate([np.full(33, la0), np.full(33, la1), la0 + (la1 - la0) * t, la0 + (la1 - la0) * t])
        x, y = self.reader.lonlat2xy(lon, lat)
        return np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)

    def outside_window(self, lon_min, lon_max, lat_min, lat_max, guard_lon, guard_lat):
        """Do the elements (their lon / lat box) come within `guard` of the window the blocks were cut to?  (The reference's
        Reader.prepare() does not shrink a reader's coverage: its blocks follow the elements.  Here every resident level is
        cut to ONE window for the run -- set_extent() -- so elements that outrun drift:max_speed would fall off it and
        silently take the fallback values; the model checks with this and re-cuts the window, recut().)"""
        if getattr(self, 'extent', None) is None or self.extent_lonlat is None:
            return False
        lo0, la0, lo1, la1 = self.extent_lonlat
        r = self.reader
        # the window may be bounded by the reader's own domain: nothing to gain beyond it
        xs, ys = np.asarray(r.x, dtype=np.float64), np.asarray(r.y, dtype=np.float64)
        x, y = self._box_outline_xy(lon_min - guard_lon, lat_min - guard_lat, lon_max + guard_lon, lat_max + guard_lat)
        ex, ey = self.extent
        return bool((x.min() < max(ex[0], xs.min())) or (x.max() > min(ex[1], xs.max())) or
                    (y.min() < max(ey[0], ys.min())) or (y.max() > min(ey[1], ys.max())))

    def recut(self, lonlat_box):
        """A new window for the blocks: every resident level is dropped and the device source of the reader is rebuilt with
        the next upload (the model rebinds its variables: the reader's source id changes)."""
        for pool in (self.slots, self.staged):
            for k in list(pool):
                self.ctx.drop_block(self.sid, pool.pop(k))
        self._drain_prefetched()
        self.close_read_ahead()       # (a level of the old window may be on its way)
        self._dist_shapes = None
        if self.sid is not None:
            self.ctx.release_source(self.sid)      # its id serves the re-cut source: the context holds 16 sources, not 16 per run
        self.sid = None
        self.set_extent(lonlat_box)

    def close_read_ahead(self):
        """End of the run, a re-cut window, a discarded reader: the worker's read in flight is waited for (the caller may close
        the reader's file afterwards) and the pool is shut down; a later read starts a new one."""
        a, self._ahead = self._ahead, None
        if a is not None:
            self._ahead_stats = tuple(x + y for x, y in zip(getattr(self, '_ahead_stats', (0, 0, 0.0)), (a.hits, a.misses, a.worker_s)))
            a.close()

    def _drain_prefetched(self, keep=()):
        """Sharded run: levels whose broadcast was started ahead and that are not (no longer) wanted are waited for and let
        go -- their tensors and work handles must not stay pinned, and an entry left behind would block every later prefetch."""
        from . import distributed as D
        for k in [k for k in self._dist_pre if k not in keep]:
            tens, works = self._dist_pre.pop(k)
            try:
                D.finish_broadcast(works)
            except Exception:
                pass

    def ensure_levels(self, t0, t1, extent=None, broadcast=None, times=None):
        """Make the time levels the step from t0 to t1 samples resident (datetime arguments).  `times`: the instants the
        step samples THIS reader at (default: t0, the middle and t1 -- a Runge-Kutta-4 step of a reader that holds the
        current; the model passes [t0] for Euler and for readers the stages do not sample)."""
        r = self.reader
        if extent is None:
            extent = getattr(self, 'extent', None)
        if not self.is_grid():
            return
        if t1 != t0:
            self._direction = 1 if t1 > t0 else -1      # of the run (ReadAhead reads the next level in that direction)
        if r.times is None:
            need = [0]
        else:
            # The step samples the reader at t0 (main loop), t0 + dt/2 and t0 + dt (Runge-Kutta stages,
            # physics_methods.py:638-670): the two levels bracketing EACH of these times must be resident -- not the
            # whole range in between (a model time step may span many reader levels), and never a truncated range
            # (a dropped 'before' level would make the device extrapolate in time).
            if times is None:
                times = (t0, t0 + (t1 - t0) / 2, t1)
            times = [t for t in times if r.covers_time(t)]
            if not times:
                return
            need = sorted({k for t in times for k in r.nearest_time(t)})
            if len(need) > self.NSLOTS:     # cannot happen with NSLOTS = 6 (three instants, two levels each)
                raise ReaderLevelsError('reader %s: one model time step needs %d time levels resident, at most %d fit'
                                        % (r.name, len(need), self.NSLOTS))
        # sharded prefetch: an entry the run has passed (the model step skipped that level, or the direction changed) is let go
        if self._dist_pre:
            self._drain_prefetched(keep=[k for k in self._dist_pre if (k >= min(need) if t1 >= t0 else k <= max(need))])
        # make room: first the resident levels the step does not need, then prefetched levels it does not need
        missing = [k for k in need if k not in self.slots and k not in self.staged]
        for pool in (self.slots, self.staged):
            for k in list(pool):
                if len(self.slots) + len(self.staged) + len(missing) <= self.NSLOTS:
                    break
                if k not in need:
                    self.ctx.drop_block(self.sid, pool.pop(k))
        for k in need:
            if k in self.slots:
                continue
            if k in self.staged:     # prefetched on the upload stream while the previous steps ran
                self.slots[k] = self.staged.pop(k)
                self.ctx.commit_block(self.sid, self.slots[k])
                continue
            self._upload(k, extent, broadcast, asynchronous=False)
        # prefetch the time level the run will need next (readers whose arrays live in host memory)
        rccl = False
        if self.world > 1:
            from . import distributed as D
            rccl = D.backend() == 'rccl'
        if self.prefetch and self.world > 1 and not rccl and r.times is not None and broadcast is None and self.sid is not None and \
                getattr(self, '_dist_shapes', None) is not None:
            kn = (max(need) + 1) if t1 >= t0 else (min(need) - 1)
            if 0 <= kn < len(r.times) and kn not in self.slots and kn not in self._dist_pre and len(self._dist_pre) == 0:
                try:
                    self._prefetch_dist(kn, extent)
                except Exception:        # a failing read shows up -- on every rank alike -- when the level is due
                    self._dist_pre.pop(kn, None)
        elif self.prefetch and r.times is not None and broadcast is None and not getattr(r, 's_levels', False):
            kn = (max(need) + 1) if t1 >= t0 else (min(need) - 1)
            if 0 <= kn < len(r.times) and kn not in self.slots and kn not in self.staged:
                if len(self.slots) + len(self.staged) >= self.NSLOTS:     # room for the prefetch: the stalest level goes
                    stale = [k for k in self.slots if k not in need]
                    if stale:
                        k_old = min(stale) if t1 >= t0 else max(stale)
                        self.ctx.drop_block(self.sid, self.slots.pop(k_old))
                if len(self.slots) + len(self.staged) < self.NSLOTS:
                    self._upload(kn, extent, None, asynchronous=True)

    def _read_and_broadcast(self, k, x, y, async_op):
        """Sharded run: rank 0 reads time level k, every rank receives it.  A reader failure on rank 0 reaches every rank as
        RemoteReaderError (the header of broadcast_reader_block): no rank is left waiting in a collective, and all of them
        count the failure alike."""
        from . import distributed as D
        r = self.reader
        time = r.times[k] if r.times is not None else None
        block, err, cids = None, None, None
        if self.rank == 0:
            try:
                if self._ahead is None:
                    self._ahead = ReadAhead(r, self.variables)
                block = self._ahead.read(k, x, y)
                cids = self._static_ids()
                # the level after this one IN THE RUN'S DIRECTION, on the worker thread, while the steps of this period run
                # (the first read of a backward run used to assume forward time: one level read and dropped, two stalls)
                step = getattr(self, '_direction', 0) or (1 if self._ahead_last is None or k >= self._ahead_last else -1)
                kn = k + step
                self._ahead_last = k
                if self.prefetch and r.times is not None and 0 <= kn < len(r.times):   # (nothing past the last level)
                    self._ahead.start(kn, x, y)
            except Exception as e:      # noqa: BLE001 -- every reader exception is a reader failure (environment.py:640-668)
                err = e
        shapes = getattr(self, '_dist_shapes', None)
        meta, tens, works = D.broadcast_reader_block(block, self.variables, src=0, error=err, shapes=shapes, async_op=async_op,
                                                     content_ids=cids)
        if shapes is None:
            self._dist_shapes = {v: tuple(t.shape) for v, t in tens.items() if v != '__cid__'}
        return meta, tens, works

    def _read_level_rccl(self, k, x, y):
        """Sharded run over the C-ABI collectives (distributed.backend() == 'rccl'): rank 0 reads time level k; every rank
        learns its HEADER -- failure, array shapes, ensemble members, content ids and (first level) the coordinate metadata --
        in one small broadcast; the arrays themselves travel inside odr_block_broadcast.  Returns (block dict whose variables
        are host arrays on rank 0 and shape placeholders elsewhere, arrays-or-None, shapes)."""
        from . import distributed as D
        r = self.reader
        block, err, hdr, arrays = None, None, None, None
        if self.rank == 0:
            try:
                if self._ahead is None:
                    self._ahead = ReadAhead(r, self.variables)
                block = self._ahead.read(k, x, y)
                step = getattr(self, '_direction', 0) or (1 if self._ahead_last is None or k >= self._ahead_last else -1)
                kn = k + step
                self._ahead_last = k
                if self.prefetch and r.times is not None and 0 <= kn < len(r.times):
                    self._ahead.start(kn, x, y)

                def one(a):
                    return np.ascontiguousarray(np.ma.filled(a, np.nan) if isinstance(a, np.ma.MaskedArray) else a, dtype=np.float32)
                arrays, members = {}, {}
                for v in self.variables:
                    if isinstance(block[v], (list, tuple)):     # ensemble members stacked along the layer axis
                        m = np.stack([one(a) for a in block[v]])
                        arrays[v], members[v] = np.ascontiguousarray(m.reshape((-1,) + m.shape[-2:])), len(block[v])
                    else:
                        arrays[v] = block[v]      # (copied once, into page-locked staging memory: _upload)
                hdr = dict(shapes={v: tuple(np.shape(a)) for v, a in arrays.items()}, members=members, cids=self._static_ids())
                if getattr(self, '_dist_meta', None) is None:
                    hdr['meta'] = {kk: (np.asarray(block[kk]) if kk in ('x', 'y', 'z') and block.get(kk) is not None else block.get(kk))
                                   for kk in ('x', 'y', 'z', 's_level_variables') if kk in block}
            except Exception as e:      # noqa: BLE001 -- every reader exception is a reader failure (environment.py:640-668)
                err = e
                hdr = dict(error=repr(e))
        hdr = D.broadcast_object(hdr, src=0)
        if 'error' in hdr:
            if err is not None:
                raise err
            raise D.RemoteReaderError('the reader failed on rank 0: ' + hdr['error'])
        if 'meta' in hdr:
            self._dist_meta = hdr['meta']
        self._dist_members = hdr['members']
        self._level_cids = hdr['cids']
        out = dict(self._dist_meta)
        for v, shp in hdr['shapes'].items():
            out[v] = arrays[v] if arrays is not None else _ShapeOnly(shp)
        return out, arrays, hdr['shapes']

    def _prefetch_dist(self, kn, extent):
        """Start the broadcast of the level the run needs next while the current one is in use (every rank makes this call
        at the same step: the collectives stay in the same order everywhere)."""
        x = y = None
        if extent is not None:
            x, y = np.array(extent[0]), np.array(extent[1])
        meta, tens, works = self._read_and_broadcast(kn, x, y, async_op=True)
        self._dist_pre[kn] = (tens, works)

    def _page_locked(self, v, arr):
        if isinstance(arr, (list, tuple)):       # ensemble members: stacked by the context
            return arr
        a = np.asarray(arr) if not isinstance(arr, np.ma.MaskedArray) else arr
        if isinstance(a, np.ndarray) and not isinstance(a, np.ma.MaskedArray) and a.dtype == np.float32 and \
                a.flags['C_CONTIGUOUS'] and any(lo <= a.ctypes.data and a.ctypes.data + a.nbytes <= hi
                                                for lo, hi in getattr(self, '_pinned_ranges', [])):
            return a
        stage = self.__dict__.setdefault('_stage', {})
        st = stage.get((self._ring, v))
        if st is None or st.shape != a.shape:
            st = np.empty(a.shape, np.float32)
            try:
                self.ctx.pin(st)
            except Exception:
                pass
            stage[(self._ring, v)] = st
        np.copyto(st, np.ma.filled(a, np.nan) if isinstance(a, np.ma.MaskedArray) else a, casting='unsafe')
        return st

    def _upload(self, k, extent, broadcast, asynchronous):
        """One reader time level -> one device block (synchronously, or staged on the upload stream)."""
        r = self.reader
        time = r.times[k] if r.times is not None else None
        x = y = None
        rccl_arrays = rccl_shapes = None
        if extent is not None:
            x, y = np.array(extent[0]), np.array(extent[1])
        if self.world > 1:
            # sharded run (one process per GPU): the rank that owns the host Reader reads the level, every rank receives
            # it -- over RCCL / xGMI straight into device memory (opendrift_amd/distributed.py).  A level whose broadcast
            # was started one period ahead (_prefetch_dist) only has to be waited for.
            from . import distributed as D
            import time as _time
            t_wait = _time.perf_counter()
            rccl_arrays = rccl_shapes = None
            if D.backend() == 'rccl':
                if getattr(r, 's_levels', False):
                    raise NotImplementedError('a sigma-level reader in a sharded run needs ODR_DIST_BACKEND=nccl (the regridding '
                                              'reads the level on every rank)')
                block, rccl_arrays, rccl_shapes = self._read_level_rccl(k, x, y)
                block['time'] = time
                self.stall_s += _time.perf_counter() - t_wait
            else:
                if k in self._dist_pre:
                    tens, works = self._dist_pre.pop(k)
                    D.finish_broadcast(works)
                    meta = None
                else:
                    meta, tens, _ = self._read_and_broadcast(k, x, y, async_op=False)
                self.stall_s += _time.perf_counter() - t_wait
                if meta is not None:
                    self._dist_members = meta.pop('__members__', {})   # ensemble variables arrive as [members x nz, ny, nx] stacks
                    self._dist_meta = meta
                block = dict(self._dist_meta)
                block['time'] = time
                cid_t = tens.pop('__cid__', None)
                self._level_cids = None if cid_t is None else dict(zip(self.variables, [int(i) for i in cid_t.cpu().tolist()]))
                for v, t in tens.items():
                    block[v] = t if t.is_cuda else t.numpy()
                self._tensors = tens      # keep the device tensors alive until the block is built
                asynchronous = False
        else:
            block = r.get_variables(self.variables, time, x, y, np.array([0.0]))
        if broadcast is not None:
            block = broadcast(block)
        bx, by = np.asarray(block['x']), np.asarray(block['y'])
        bz = block.get('z', None)
        if self.sid is None and not getattr(r, 'projected', True):
            # reader without projection: the whole mesh is the device source (blocks must span it)
            zz = np.atleast_1d(bz) if bz is not None and np.size(bz) > 1 else None
            if (len(by), len(bx)) != r.lon.shape:
                raise NotImplementedError('a reader without projection must hand out blocks of its whole mesh')
            dom = (float(r.xmin), float(r.xmax), float(r.ymin), float(r.ymax), float(r.zmin), float(r.zmax))
            self.sid = self.ctx.add_grid_curvilinear(r.lon, r.lat, z=zz, domain=dom)
            sid, ctx = self.sid, self.ctx
            r._device_lookup = lambda lon, lat: ctx.lonlat2xy(sid, lon, lat)
            if r.start_time is not None:
                self.ctx.set_time_coverage(self.sid, _epoch(r.start_time), _epoch(r.end_time), r.always_valid)
        elif self.sid is None:
            proj = projection.parse_proj4(r.proj4)
            zz = np.atleast_1d(bz) if bz is not None and np.size(bz) > 1 else None
            # modulate_longitude (variables.py:259-280) decides on the TRUE longitudes of the domain's corners: some negative ->
            # np.mod(lon + 180, 360) - 180, else np.mod(lon, 360) (a rotated-pole reader's x is modulated the same way once more
            # for the coverage test: crs.is_geographic, :246).  In float64 the two branches agree on [0, 180); in the float32
            # arithmetic of a run's first get_environment (odr_ctx_set_position_class) they do not.
            exlons, _ = r.xy2lonlat(np.array([r.xmin, r.xmin, r.xmax, r.xmax], dtype=np.float64),
                                    np.array([r.ymin, r.ymax, r.ymax, r.ymin], dtype=np.float64))
            lon_mode = 1 if np.min(exlons) < 0 else 2
            dom = (float(r.xmin), float(r.xmax), float(r.ymin), float(r.ymax), float(r.zmin), float(r.zmax))
            if extent is not None:     # blocks cut to the simulation extent: the source covers what the blocks cover
                dom = (max(dom[0], float(np.min(bx))), min(dom[1], float(np.max(bx))), max(dom[2], float(np.min(by))),
                       min(dom[3], float(np.max(by))), dom[4], dom[5])
            self.sid = self.ctx.add_grid(bx, by, z=zz, proj=proj, lon_mode=lon_mode, domain=dom)
            if r.start_time is not None:
                self.ctx.set_time_coverage(self.sid, _epoch(r.start_time), _epoch(r.end_time), r.always_valid)
        else:
            g = self.ctx._grids[self.sid]
            if (len(by), len(bx)) != (g['ny'], g['nx']):
                raise ValueError('reader %s changed its block shape between time levels' % r.name)
        free = [s for s in range(self.NSLOTS) if s not in self.slots.values() and s not in self.staged.values()]
        slot = free[0]
        t_ep = _epoch(time) if time is not None else 0.0
        if self.world > 1 and rccl_shapes is not None:
            # the level's arrays in ONE broadcast on the upload stream, straight into the staging memory of the block
            # preparation (odr_block_broadcast); staged like an asynchronous upload
            for v, m in getattr(self, '_dist_members', {}).items():
                self.ctx.declare_members(self.sid, v, m)
            if rccl_arrays is not None:
                # rank 0: the level through page-locked staging arrays of this binding (two sets in turn, as the asynchronous
                # upload of a one-process run): its copy into the device's staging memory is a DMA transfer the host does not
                # wait behind (pageable memory would go through the library's bounce buffer, synchronously)
                self._ring = 1 - getattr(self, '_ring', 1)
                rccl_arrays = {v: self._page_locked(v, a) for v, a in rccl_arrays.items()}
            self.ctx.block_broadcast(self.sid, slot, t_ep, rccl_arrays, rccl_shapes, root=0, content_ids=getattr(self, '_level_cids', None))
            if asynchronous:
                self.staged[k] = slot
            else:
                self.ctx.commit_block(self.sid, slot)
                self.slots[k] = slot
            return
        if asynchronous:
            if not self._pinned:     # page-lock the reader's in-memory arrays once: uploads become DMA transfers
                self._pinned = True
                self._pinned_ranges = []
                for v in self.variables:
                    a = getattr(r, 'arrays', {}).get(v)
                    if isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags['C_CONTIGUOUS']:
                        try:
                            self.ctx.pin(a)
                            self._pinned_ranges.append((a.ctypes.data, a.ctypes.data + a.nbytes))
                        except Exception:
                            pass
            # what is not a contiguous float32 piece of page-locked memory (a window cut out of the reader's arrays, a
            # masked or float64 array, ensemble lists) is copied into page-locked staging arrays of this binding, two
            # sets in turn: the upload then is an asynchronous DMA transfer for these as well (pageable memory would go
            # through the library's bounce buffer, synchronously)
            self._ring = 1 - getattr(self, '_ring', 1)
            self.ctx.upload_block_async(self.sid, slot, t_ep, {v: self._page_locked(v, block[v]) for v in self.variables},
                                        content_ids=self._static_ids())
            self.staged[k] = slot
            return
        if getattr(r, 's_levels', False):
            # ROMS-type reader: the block arrives on s-levels and is regridded to the z levels on the device
            # (reader_ROMS_native.py:617-684); the float32 result never visits the host
            if getattr(self, 'sgrid', None) is None:
                from .device import SigmaGrid
                self.sgrid = SigmaGrid(self.ctx, r.h, r.hc, r.Cs_r, Vtransform=r.Vtransform)
            arrays, nzv = {}, {}
            for i, v in enumerate([v for v in self.variables if v in block['s_level_variables']]):
                arrays[v] = self.sgrid.zslice(_dev(block[v]), bz, slot=i)
                nzv[v] = len(bz)
            for v in self.variables:
                if v not in arrays:
                    arrays[v] = _dev(block[v])
                    nzv.setdefault(v, block[v].shape[0] if len(block[v].shape) == 3 else 1)
            self.ctx.upload_block_device(self.sid, slot, _epoch(time) if time is not None else 0.0, arrays, nzv,
                                         content_ids={v: i for v, i in (self._static_ids() or {}).items() if v not in block.get('s_level_variables', ())})
        elif self.world > 1:
            for v, m in getattr(self, '_dist_members', {}).items():
                self.ctx.declare_members(self.sid, v, m)
            arrays = {v: _dev(block[v]) for v in self.variables}
            nzv = {v: (block[v].shape[0] if len(block[v].shape) == 3 else 1) for v in self.variables}
            self.ctx.upload_block_device(self.sid, slot, _epoch(time) if time is not None else 0.0, arrays, nzv,
                                         content_ids=getattr(self, '_level_cids', None))
            self._tensors = None
        else:
            arrays = {v: block[v] for v in self.variables}
            self.ctx.upload_block(self.sid, slot, _epoch(time) if time is not None else 0.0, arrays, content_ids=self._static_ids())
        self.slots[k] = slot
