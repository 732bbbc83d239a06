"""Host-side (NumPy) map projections for set-up work: seeding in reader coordinates, reader
extents, proj4 parsing.  The per-step transforms run on the device (csrc/odr_field.hip.h).

Stands in for pyproj.Proj as used by readers (basereader/__init__.py:119-137,
variables.py:111-143) for the projections on the path: latlong, stereographic (Snyder, USGS PP 1395, ch. 21),
Mercator (ch. 7) and Lambert conformal conic (ch. 15), sphere or ellipsoid.
"""
import re

import numpy as np

_ELLPS = {'WGS84': (6378137.0, 298.257223563), 'GRS80': (6378137.0, 298.257222101)}


def parse_proj4(proj4):
    """proj4 string -> dict understood by device.proj_desc / Proj below."""
    p = {k: (v if v != '' else True) for k, v in re.findall(r'\+([A-Za-z_0-9]+)(?:\s*=\s*(\S+))?', proj4)}
    name = p.get('proj', 'latlong')
    if name in ('latlong', 'longlat', 'latlon', 'lonlat'):
        return dict(kind='latlong')
    if name not in ('stere', 'merc', 'lcc'):
        raise NotImplementedError('projection +proj=%s is not on the device path (latlong, stere, merc, lcc are)' % name)
    a, rf = _ELLPS[p.get('ellps', 'WGS84')]
    if 'R' in p:
        a, rf = float(p['R']), 0.0
    if 'a' in p:
        a = float(p['a'])
        if 'rf' in p:
            rf = float(p['rf'])
        elif 'f' in p:
            rf = 0.0 if float(p['f']) == 0 else 1.0 / float(p['f'])
        elif 'b' in p:
            rf = 0.0 if float(p['b']) == a else a / (a - float(p['b']))
        elif 'e' in p:
            e = float(p['e'])
            rf = 0.0 if e == 0 else 1.0 / (1 - np.sqrt(1 - e * e))
        elif 'ellps' not in p:
            rf = 0.0
    lat0 = float(p.get('lat_0', 0))
    out = dict(a=a, rf=rf, lat0=lat0, lon0=float(p.get('lon_0', 0)), k0=float(p.get('k_0', p.get('k', 1.0))),
               x0=float(p.get('x_0', 0)), y0=float(p.get('y_0', 0)))
    if name == 'merc':
        out.update(kind='merc', lat_ts=float(p.get('lat_ts', 0.0)))
    elif name == 'lcc':
        lat1 = float(p.get('lat_1', 0.0))
        lat2 = float(p['lat_2']) if 'lat_2' in p else lat1
        if 'lat_2' not in p and 'lat_0' not in p:
            out['lat0'] = lat1                       # PROJ: a tangent cone without +lat_0 has its origin on the parallel
        if abs(lat1 + lat2) < 1e-10:
            raise ValueError('lcc: lat_1 = -lat_2')
        out.update(kind='lcc', lat1=lat1, lat2=lat2, lat_ts=0.0)
    elif abs(abs(lat0) - 90) < 1e-10:
        out.update(kind='stere_polar', lat_ts=float(p.get('lat_ts', 90.0)))
    elif lat0 == 0 and not rf:
        out.update(kind='stere_equit_sphere', lat_ts=0.0)
    else:
        raise NotImplementedError('oblique / ellipsoidal equatorial stereographic is not on the device path')
    return out


def _es(rf):
    f = 0.0 if not rf else 1.0 / rf
    return f * (2 - f)


def stere_equit_sphere_forward(lon, lat, a):
    lam, phi = np.radians(lon), np.radians(lat)
    k = 2.0 / (1 + np.cos(phi) * np.cos(lam))
    return a * k * np.cos(phi) * np.sin(lam), a * k * np.sin(phi)


def stere_equit_sphere_inverse(x, y, a):
    X, Y = np.asarray(x) / a, np.asarray(y) / a
    rh = np.hypot(X, Y)
    c = 2 * np.arctan(rh / 2.0)
    with np.errstate(invalid='ignore', divide='ignore'):
        phi = np.where(rh <= 1e-10, 0.0, np.arcsin(Y * np.sin(c) / np.where(rh == 0, 1, rh)))
    lam = np.arctan2(X * np.sin(c), np.cos(c) * rh)
    return np.degrees(lam), np.degrees(phi)


def _akm1(es, lat_ts, k0):
    e, phits = np.sqrt(es), np.radians(abs(lat_ts))
    if abs(phits - np.pi / 2) < 1e-10:
        return 2 * k0 / np.sqrt((1 + e) ** (1 + e) * (1 - e) ** (1 - e)) if es else 2 * k0
    if es == 0:
        return np.cos(phits) / np.tan(0.5 * (np.pi / 2 - phits))
    t = np.sin(phits)
    ts = np.tan(0.5 * (np.pi / 2 - phits)) / ((1 - e * t) / (1 + e * t)) ** (0.5 * e)
    return np.cos(phits) / ts / np.sqrt(1 - (e * t) ** 2)


def stere_polar_forward(lon, lat, a=6378137.0, rf=298.257223563, lat0=90.0, lon0=0.0, lat_ts=90.0, k0=1.0,
                        x0=0.0, y0=0.0, **_):
    es = _es(rf)
    e = np.sqrt(es)
    south = lat0 < 0
    lam = np.radians(np.asarray(lon, dtype=np.float64)) - np.radians(lon0)
    lam = (lam + np.pi) % (2 * np.pi) - np.pi
    phi = np.radians(np.asarray(lat, dtype=np.float64))
    if south:
        phi = -phi
    sp = np.sin(phi)
    rho = _akm1(es, lat_ts, k0) * np.tan(0.5 * (np.pi / 2 - phi)) / ((1 - e * sp) / (1 + e * sp)) ** (0.5 * e)
    cl = -np.cos(lam) if south else np.cos(lam)
    return a * rho * np.sin(lam) + x0, -a * rho * cl + y0


def stere_polar_inverse(x, y, a=6378137.0, rf=298.257223563, lat0=90.0, lon0=0.0, lat_ts=90.0, k0=1.0,
                        x0=0.0, y0=0.0, **_):
    es = _es(rf)
    e = np.sqrt(es)
    south = lat0 < 0
    X, Y = (np.asarray(x, dtype=np.float64) - x0) / a, (np.asarray(y, dtype=np.float64) - y0) / a
    if not south:
        Y = -Y
    tp = np.hypot(X, Y) / _akm1(es, lat_ts, k0)
    phi = np.pi / 2 - 2 * np.arctan(tp)
    for _ in range(12):
        s = e * np.sin(phi)
        phi = np.pi / 2 - 2 * np.arctan(tp * ((1 - s) / (1 + s)) ** (0.5 * e))
    if south:
        phi = -phi
    lam = np.arctan2(X, Y) + np.radians(lon0)
    lam = (lam + np.pi) % (2 * np.pi) - np.pi
    return np.degrees(lam), np.degrees(phi)


def _tsfn(phi, e):
    s = e * np.sin(phi)
    return np.tan(0.5 * (np.pi / 2 - phi)) / ((1 - s) / (1 + s)) ** (0.5 * e)


def _phi2(ts, e):
    phi = np.pi / 2 - 2 * np.arctan(ts)
    for _ in range(12 if e else 0):
        s = e * np.sin(phi)
        phi = np.pi / 2 - 2 * np.arctan(ts * ((1 - s) / (1 + s)) ** (0.5 * e))
    return phi


def _msfn(phi, es):
    return np.cos(phi) / np.sqrt(1 - es * np.sin(phi) ** 2)


def _wrap(lam):
    return (lam + np.pi) % (2 * np.pi) - np.pi


def merc_constants(p):
    es = _es(p['rf'])
    return np.sqrt(es), (_msfn(np.radians(abs(p['lat_ts'])), es) if p.get('lat_ts') else p['k0'])


def merc_forward(lon, lat, **p):
    e, k0 = merc_constants(p)
    lam = _wrap(np.radians(np.asarray(lon, dtype=np.float64)) - np.radians(p['lon0']))
    phi = np.radians(np.asarray(lat, dtype=np.float64))
    return p['a'] * k0 * lam + p['x0'], -p['a'] * k0 * np.log(_tsfn(phi, e)) + p['y0']


def merc_inverse(x, y, **p):
    e, k0 = merc_constants(p)
    X, Y = (np.asarray(x, dtype=np.float64) - p['x0']) / p['a'], (np.asarray(y, dtype=np.float64) - p['y0']) / p['a']
    return np.degrees(_wrap(X / k0 + np.radians(p['lon0']))), np.degrees(_phi2(np.exp(-Y / k0), e))


def lcc_constants(p):
    es = _es(p['rf'])
    e = np.sqrt(es)
    ph1, ph2, ph0 = np.radians(p['lat1']), np.radians(p['lat2']), np.radians(p['lat0'])
    n = np.sin(ph1)
    m1, t1 = _msfn(ph1, es), _tsfn(ph1, e)
    if abs(ph1 - ph2) >= 1e-10:
        n = np.log(m1 / _msfn(ph2, es)) / np.log(t1 / _tsfn(ph2, e))
    c = m1 * t1 ** (-n) / n
    rho0 = 0.0 if abs(abs(ph0) - np.pi / 2) < 1e-10 else c * _tsfn(ph0, e) ** n
    return e, n, c, rho0


def lcc_forward(lon, lat, **p):
    e, n, c, rho0 = lcc_constants(p)
    lam = _wrap(np.radians(np.asarray(lon, dtype=np.float64)) - np.radians(p['lon0']))
    phi = np.radians(np.asarray(lat, dtype=np.float64))
    rho = np.where(np.abs(np.abs(phi) - np.pi / 2) < 1e-10, 0.0, c * _tsfn(phi, e) ** n)
    return p['a'] * p['k0'] * rho * np.sin(n * lam) + p['x0'], p['a'] * p['k0'] * (rho0 - rho * np.cos(n * lam)) + p['y0']


def lcc_inverse(x, y, **p):
    e, n, c, rho0 = lcc_constants(p)
    xx = (np.asarray(x, dtype=np.float64) - p['x0']) / p['a'] / p['k0']
    yy = rho0 - (np.asarray(y, dtype=np.float64) - p['y0']) / p['a'] / p['k0']
    rho = np.hypot(xx, yy)
    if n < 0:
        rho, xx, yy = -rho, -xx, -yy
    with np.errstate(divide='ignore', invalid='ignore'):
        phi = np.where(rho != 0, _phi2((rho / c) ** (1 / n), e), np.pi / 2 * np.sign(n))
        lam = np.where(rho != 0, np.arctan2(xx, yy) / n, 0.0)
    return np.degrees(_wrap(lam + np.radians(p['lon0']))), np.degrees(phi)


class Proj:
    """Callable like pyproj.Proj: p(lon, lat) -> x, y ; p(x, y, inverse=True) -> lon, lat."""

    def __init__(self, proj4):
        self.srs = proj4
        self.params = parse_proj4(proj4)
        self.is_geographic = self.params['kind'] == 'latlong'

    def __call__(self, a, b, inverse=False):
        p = self.params
        if p['kind'] == 'latlong':
            return a, b
        if p['kind'] == 'stere_equit_sphere':
            return (stere_equit_sphere_inverse(a, b, p['a']) if inverse
                    else stere_equit_sphere_forward(a, b, p['a']))
        if p['kind'] == 'merc':
            return merc_inverse(a, b, **p) if inverse else merc_forward(a, b, **p)
        if p['kind'] == 'lcc':
            return lcc_inverse(a, b, **p) if inverse else lcc_forward(a, b, **p)
        return stere_polar_inverse(a, b, **p) if inverse else stere_polar_forward(a, b, **p)
