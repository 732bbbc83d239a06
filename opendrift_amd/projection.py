"""Host-side (NumPy) map projections for set-up work: seeding in reader coordinates, reader
extents, proj4 parsing.  The per-step transforms run on the device (csrc/odr_field.hip.h).

Stands in for pyproj.Proj as used by readers (basereader/__init__.py:119-137,
variables.py:111-143) for the projections on the path: latlong and stereographic
(Snyder, USGS PP 1395, ch. 21).
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
    if name != 'stere':
        raise NotImplementedError('projection +proj=%s is not on the device path' % name)
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
    if abs(abs(lat0) - 90) < 1e-10:
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
        return stere_polar_inverse(a, b, **p) if inverse else stere_polar_forward(a, b, **p)
