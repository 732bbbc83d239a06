"""Host-side (NumPy) map projections for set-up work: seeding in reader coordinates, reader
extents, proj4 parsing.  The per-step transforms run on the device (csrc/odr_field.hip.h).

Stands in for pyproj.Proj as used by readers (basereader/__init__.py:119-137,
variables.py:111-143) for the projections on the path: latlong, stereographic (Snyder, USGS PP 1395, ch. 21: polar,
equatorial and oblique aspects), Mercator (ch. 7), Lambert conformal conic (ch. 15), transverse Mercator / UTM (Krueger's
series to order 6 in the third flattening: Karney 2011), Lambert azimuthal equal-area (ch. 24), sphere or ellipsoid, and the
rotated pole (+proj=ob_tran +o_proj=longlat, Snyder 5-7..5-10b).
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
    if name == 'ob_tran':
        # the rotated pole of HIRLAM / AROME / CMEMS-Arctic files: only geographic coordinates are rotated (o_proj=longlat),
        # reader coordinates are rotated longitude / latitude in DEGREES (variables.py:117-123,136-138)
        if p.get('o_proj') not in ('longlat', 'latlong', 'latlon', 'lonlat'):
            raise NotImplementedError('ob_tran with +o_proj=%s is not on the device path (o_proj=longlat is)' % p.get('o_proj'))
        if 'to_meter' in p or 'o_alpha' in p or 'o_lon_1' in p or 'o_lon_c' in p:
            raise NotImplementedError('ob_tran: only the +o_lat_p / +o_lon_p form without +to_meter is on the device path')
        return dict(kind='ob_tran', a=1.0, rf=0.0, lat0=0.0, lon0=float(p.get('lon_0', 0)), k0=1.0, x0=0.0, y0=0.0,
                    lat1=float(p.get('o_lat_p', 90.0)), lat2=float(p.get('o_lon_p', 0.0)), lat_ts=0.0)
    if name not in ('stere', 'merc', 'lcc', 'tmerc', 'utm', 'laea'):
        raise NotImplementedError('projection +proj=%s is not on the device path (latlong, stere, merc, lcc, tmerc, utm, laea, '
                                  'ob_tran with o_proj=longlat are)' % name)
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
    if name == 'utm':     # PROJ's utm: tmerc on the zone's meridian, k0 = 0.9996, false easting 500 km (northing 10 000 km south)
        if not rf:
            raise ValueError('utm needs an ellipsoid')
        zone = int(p['zone'])
        if not 1 <= zone <= 60:
            raise ValueError('utm: zone %d' % zone)
        out.update(kind='tmerc', lat0=0.0, lon0=6.0 * zone - 183.0, k0=0.9996, x0=500000.0, y0=10000000.0 if 'south' in p else 0.0,
                   lat_ts=0.0)
    elif name == 'tmerc':
        out.update(kind='tmerc', lat_ts=0.0)
    elif name == 'laea':
        out.update(kind='laea', lat_ts=0.0, k0=1.0)
    elif name == 'merc':
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
    elif lat0 == 0 and not rf and out['k0'] == 1.0 and out['x0'] == 0 and out['y0'] == 0:
        out.update(kind='stere_equit_sphere', lat_ts=0.0)
    else:                 # oblique, or equatorial on an ellipsoid / with a scale factor (PROJ ignores +lat_ts here)
        out.update(kind='stere_oblique', lat_ts=0.0)
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


# ---- transverse Mercator (Karney 2011: eqs. 7-9 conformal latitude, 10-11 / 35 forward series, 36 inverse, 19-21 Newton)
def _taup(tau, e):
    t1 = np.hypot(1.0, tau)
    sig = np.sinh(e * np.arctanh(e * tau / t1))
    return np.hypot(1.0, sig) * tau - sig * t1


def _tau_from_taup(taup, e):
    e2m = 1 - e * e
    tau = taup / e2m
    for _ in range(6):
        tp = _taup(tau, e)
        tau = tau + (taup - tp) * (1 + e2m * tau * tau) / (e2m * np.hypot(1.0, tau) * np.hypot(1.0, tp))
    return tau


def tmerc_constants(p):
    es = _es(p['rf'])
    e = np.sqrt(es)
    f = 1 - np.sqrt(1 - es)
    n = f / (2 - f)
    qn = p['k0'] / (1 + n) * (1 + n ** 2 * (1 / 4 + n ** 2 * (1 / 64 + n ** 2 / 256)))
    al = [n / 2 - 2 * n ** 2 / 3 + 5 * n ** 3 / 16 + 41 * n ** 4 / 180 - 127 * n ** 5 / 288 + 7891 * n ** 6 / 37800,
          13 * n ** 2 / 48 - 3 * n ** 3 / 5 + 557 * n ** 4 / 1440 + 281 * n ** 5 / 630 - 1983433 * n ** 6 / 1935360,
          61 * n ** 3 / 240 - 103 * n ** 4 / 140 + 15061 * n ** 5 / 26880 + 167603 * n ** 6 / 181440,
          49561 * n ** 4 / 161280 - 179 * n ** 5 / 168 + 6601661 * n ** 6 / 7257600,
          34729 * n ** 5 / 80640 - 3418889 * n ** 6 / 1995840, 212378941 * n ** 6 / 319334400]
    be = [n / 2 - 2 * n ** 2 / 3 + 37 * n ** 3 / 96 - n ** 4 / 360 - 81 * n ** 5 / 512 + 96199 * n ** 6 / 604800,
          n ** 2 / 48 + n ** 3 / 15 - 437 * n ** 4 / 1440 + 46 * n ** 5 / 105 - 1118711 * n ** 6 / 3870720,
          17 * n ** 3 / 480 - 37 * n ** 4 / 840 - 209 * n ** 5 / 4480 + 5569 * n ** 6 / 90720,
          4397 * n ** 4 / 161280 - 11 * n ** 5 / 504 - 830251 * n ** 6 / 7257600,
          4583 * n ** 5 / 161280 - 108847 * n ** 6 / 3991680, 20648693 * n ** 6 / 638668800]
    xi0p = np.arctan(_taup(np.tan(np.radians(p['lat0'])), e))
    xi0 = xi0p + sum(al[k] * np.sin(2 * (k + 1) * xi0p) for k in range(6))
    return e, qn, al, be, xi0


def tmerc_forward(lon, lat, **p):
    e, qn, al, be, xi0 = tmerc_constants(p)
    lam = _wrap(np.radians(np.asarray(lon, dtype=np.float64)) - np.radians(p['lon0']))
    phi = np.radians(np.asarray(lat, dtype=np.float64))
    taup = _taup(np.tan(phi), e)
    xip, etap = np.arctan2(taup, np.cos(lam)), np.arcsinh(np.sin(lam) / np.hypot(taup, np.cos(lam)))
    xi = xip + sum(al[k] * np.sin(2 * (k + 1) * xip) * np.cosh(2 * (k + 1) * etap) for k in range(6))
    eta = etap + sum(al[k] * np.cos(2 * (k + 1) * xip) * np.sinh(2 * (k + 1) * etap) for k in range(6))
    return p['a'] * qn * eta + p['x0'], p['a'] * qn * (xi - xi0) + p['y0']


def tmerc_inverse(x, y, **p):
    e, qn, al, be, xi0 = tmerc_constants(p)
    eta = (np.asarray(x, dtype=np.float64) - p['x0']) / p['a'] / qn
    xi = (np.asarray(y, dtype=np.float64) - p['y0']) / p['a'] / qn + xi0
    xip = xi - sum(be[k] * np.sin(2 * (k + 1) * xi) * np.cosh(2 * (k + 1) * eta) for k in range(6))
    etap = eta - sum(be[k] * np.cos(2 * (k + 1) * xi) * np.sinh(2 * (k + 1) * eta) for k in range(6))
    sh, c = np.sinh(etap), np.cos(xip)
    phi = np.arctan(_tau_from_taup(np.sin(xip) / np.hypot(sh, c), e))
    return np.degrees(_wrap(np.arctan2(sh, c) + np.radians(p['lon0']))), np.degrees(phi)


# ---- oblique / equatorial stereographic (Snyder 21-2..21-4, 21-14, 21-15; 21-24..21-27, 21-36..21-38 with the conformal latitude)
def _conformal(phi, e):
    s = e * np.sin(phi)
    return 2 * np.arctan(np.tan(0.5 * (np.pi / 2 + phi)) * ((1 - s) / (1 + s)) ** (0.5 * e)) - np.pi / 2


def stere_oblique_forward(lon, lat, **p):
    es = _es(p['rf'])
    e = np.sqrt(es)
    lam = _wrap(np.radians(np.asarray(lon, dtype=np.float64)) - np.radians(p['lon0']))
    phi, ph0 = np.radians(np.asarray(lat, dtype=np.float64)), np.radians(p['lat0'])
    X1 = _conformal(ph0, e) if es else ph0
    X = _conformal(phi, e) if es else phi
    akm1 = 2 * p['k0'] * (np.cos(ph0) / np.sqrt(1 - es * np.sin(ph0) ** 2) / np.cos(X1) if es else 1.0)
    A = akm1 / (1 + np.sin(X1) * np.sin(X) + np.cos(X1) * np.cos(X) * np.cos(lam))
    return (p['a'] * A * np.cos(X) * np.sin(lam) + p['x0'],
            p['a'] * A * (np.cos(X1) * np.sin(X) - np.sin(X1) * np.cos(X) * np.cos(lam)) + p['y0'])


def stere_oblique_inverse(x, y, **p):
    es = _es(p['rf'])
    e = np.sqrt(es)
    ph0 = np.radians(p['lat0'])
    X1 = _conformal(ph0, e) if es else ph0
    akm1 = 2 * p['k0'] * (np.cos(ph0) / np.sqrt(1 - es * np.sin(ph0) ** 2) / np.cos(X1) if es else 1.0)
    xx, yy = (np.asarray(x, dtype=np.float64) - p['x0']) / p['a'], (np.asarray(y, dtype=np.float64) - p['y0']) / p['a']
    rho = np.hypot(xx, yy)
    c = 2 * np.arctan2(rho, akm1)
    with np.errstate(invalid='ignore', divide='ignore'):
        X = np.where(rho == 0, X1, np.arcsin(np.clip(np.cos(c) * np.sin(X1) + yy * np.sin(c) * np.cos(X1) / np.where(rho == 0, 1, rho), -1, 1)))
    lam = np.arctan2(xx * np.sin(c), rho * np.cos(X1) * np.cos(c) - yy * np.sin(X1) * np.sin(c))
    phi = X
    if es:
        tp = np.tan(0.5 * (np.pi / 2 + X))
        for _ in range(12):
            s = e * np.sin(phi)
            phi = 2 * np.arctan(tp * ((1 + s) / (1 - s)) ** (0.5 * e)) - np.pi / 2
    return np.degrees(_wrap(lam + np.radians(p['lon0']))), np.degrees(phi)


# ---- Lambert azimuthal equal-area (Snyder 24-2..24-4, 24-13..24-16; 24-17..24-26 with the authalic latitude 3-11, 3-12, 3-16)
def _qsfn(sinphi, e):
    if e < 1e-7:
        return 2 * sinphi
    con = e * sinphi
    return (1 - e * e) * (sinphi / (1 - con * con) - 0.5 / e * np.log((1 - con) / (1 + con)))


def _authalic(phi, e, qp):
    return np.arcsin(np.clip(_qsfn(np.sin(phi), e) / qp, -1, 1))


def laea_forward(lon, lat, **p):
    es = _es(p['rf'])
    e = np.sqrt(es)
    lam = _wrap(np.radians(np.asarray(lon, dtype=np.float64)) - np.radians(p['lon0']))
    phi, ph0 = np.radians(np.asarray(lat, dtype=np.float64)), np.radians(p['lat0'])
    qp = _qsfn(1.0, e)
    rq = np.sqrt(0.5 * qp)
    b, b1 = (_authalic(phi, e, qp), _authalic(ph0, e, qp)) if es else (phi, ph0)
    if abs(abs(ph0) - np.pi / 2) < 1e-10:          # polar aspects: rho = a sqrt(qp -+ q)
        sgn = 1.0 if ph0 > 0 else -1.0
        rho = np.sqrt(np.maximum(qp - sgn * qp * np.sin(b), 0.0))
        return p['a'] * rho * np.sin(lam) + p['x0'], -sgn * p['a'] * rho * np.cos(lam) + p['y0']
    d = np.cos(ph0) / (np.sqrt(1 - es * np.sin(ph0) ** 2) * rq * np.cos(b1)) if es else 1.0
    bb = rq * np.sqrt(2 / (1 + np.sin(b1) * np.sin(b) + np.cos(b1) * np.cos(b) * np.cos(lam)))
    return (p['a'] * bb * d * np.cos(b) * np.sin(lam) + p['x0'],
            p['a'] * bb / d * (np.cos(b1) * np.sin(b) - np.sin(b1) * np.cos(b) * np.cos(lam)) + p['y0'])


def laea_inverse(x, y, **p):
    es = _es(p['rf'])
    e = np.sqrt(es)
    ph0 = np.radians(p['lat0'])
    qp = _qsfn(1.0, e)
    rq = np.sqrt(0.5 * qp)
    xx, yy = (np.asarray(x, dtype=np.float64) - p['x0']) / p['a'], (np.asarray(y, dtype=np.float64) - p['y0']) / p['a']
    if abs(abs(ph0) - np.pi / 2) < 1e-10:
        sgn = 1.0 if ph0 > 0 else -1.0
        sinb = sgn * (1 - (xx * xx + yy * yy) / qp)
        lam = np.arctan2(xx, -sgn * yy)
    else:
        b1 = _authalic(ph0, e, qp) if es else ph0
        d = np.cos(ph0) / (np.sqrt(1 - es * np.sin(ph0) ** 2) * rq * np.cos(b1)) if es else 1.0
        xx, yy = xx / d, yy * d
        rho = np.hypot(xx, yy)
        ce = 2 * np.arcsin(np.clip(0.5 * rho / rq, -1, 1))
        with np.errstate(invalid='ignore', divide='ignore'):
            sinb = np.where(rho == 0, np.sin(b1), np.cos(ce) * np.sin(b1) + yy * np.sin(ce) * np.cos(b1) / np.where(rho == 0, 1, rho))
        lam = np.arctan2(xx * np.sin(ce), rho * np.cos(b1) * np.cos(ce) - yy * np.sin(b1) * np.sin(ce))
    sinb = np.clip(sinb, -1, 1)
    phi = np.arcsin(sinb)
    if es:
        for _ in range(12):    # Snyder 3-16
            sp, cp = np.sin(phi), np.cos(phi)
            w = 1 - es * sp * sp
            with np.errstate(invalid='ignore', divide='ignore'):
                phi = phi + np.where(np.abs(cp) > 1e-12, w * w / (2 * cp) * (qp * sinb / (1 - es) - sp / w
                                                                             + 0.5 / e * np.log((1 - e * sp) / (1 + e * sp))), 0.0)
    return np.degrees(_wrap(lam + np.radians(p['lon0']))), np.degrees(phi)


# ---- rotated pole (PROJ's ob_tran o_forward / o_inverse with o_proj=longlat): reader coordinates in DEGREES
def ob_tran_forward(lon, lat, **p):
    lam = _wrap(np.radians(np.asarray(lon, dtype=np.float64)) - np.radians(p['lon0']))
    phi = np.radians(np.asarray(lat, dtype=np.float64))
    sp, cp, lamp = np.sin(np.radians(p['lat1'])), np.cos(np.radians(p['lat1'])), np.radians(p['lat2'])
    x = _wrap(np.arctan2(np.cos(phi) * np.sin(lam), sp * np.cos(phi) * np.cos(lam) + cp * np.sin(phi)) + lamp)
    y = np.arcsin(np.clip(sp * np.sin(phi) - cp * np.cos(phi) * np.cos(lam), -1, 1))
    return np.degrees(x), np.degrees(y)


def ob_tran_inverse(x, y, **p):
    sp, cp, lamp = np.sin(np.radians(p['lat1'])), np.cos(np.radians(p['lat1'])), np.radians(p['lat2'])
    l, ph = np.radians(np.asarray(x, dtype=np.float64)) - lamp, np.radians(np.asarray(y, dtype=np.float64))
    phi = np.arcsin(np.clip(sp * np.sin(ph) + cp * np.cos(ph) * np.cos(l), -1, 1))
    lam = np.arctan2(np.cos(ph) * np.sin(l), sp * np.cos(ph) * np.cos(l) - cp * np.sin(ph))
    return np.degrees(_wrap(lam + np.radians(p['lon0']))), np.degrees(phi)


class Proj:
    """Callable like pyproj.Proj: p(lon, lat) -> x, y ; p(x, y, inverse=True) -> lon, lat.  (A rotated-pole reader's
    coordinates are degrees here -- the reference converts pyproj's radians itself, variables.py:117-123,136-138.)"""

    def __init__(self, proj4):
        self.srs = proj4
        self.params = parse_proj4(proj4)
        self.is_geographic = self.params['kind'] in ('latlong', 'ob_tran')   # (pyproj: a derived geographic CRS)

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
        if p['kind'] == 'tmerc':
            return tmerc_inverse(a, b, **p) if inverse else tmerc_forward(a, b, **p)
        if p['kind'] == 'laea':
            return laea_inverse(a, b, **p) if inverse else laea_forward(a, b, **p)
        if p['kind'] == 'stere_oblique':
            return stere_oblique_inverse(a, b, **p) if inverse else stere_oblique_forward(a, b, **p)
        if p['kind'] == 'ob_tran':
            return ob_tran_inverse(a, b, **p) if inverse else ob_tran_forward(a, b, **p)
        return stere_polar_inverse(a, b, **p) if inverse else stere_polar_forward(a, b, **p)
