"""TEST INFRASTRUCTURE ONLY.  Pins the map projections and the geodesic-inverse azimuth of oracle/proj.c, oracle/geodesic.c,
opendrift_amd/projection.py and the device (csrc/odr_field.hip.h) on an INDEPENDENT computation.

pyproj / PROJ are un-vendored dependencies of the reference (pyproject.toml:18-19; the reference hands every proj4 string to
pyproj, readers/basereader/variables.py:111-143) and are not installed here.  Round 5 pinned the restated formulas on Snyder's
printed examples (0.06 m) -- coarser than the 1e-7 deg the trajectories are held to, and blind to an error oracle and device
share.  Here every projection is evaluated with mpmath at 40 digits from its DEFINITION, not from the series or the
operation order of proj.c:

  merc      isometric latitude psi = asinh(tan phi) - e atanh(e sin phi)                             (Snyder 7-7)
  lcc       rho = a F t^n from m(phi), t(phi) in closed form                                           (Snyder 15-1 ... 15-11)
  stere     polar / oblique / equatorial through the conformal latitude chi in closed form           (Snyder 21-24 ... 21-40)
  laea      through the authalic latitude beta from q(phi) in closed form                            (Snyder 24-1 ... 24-26)
  tmerc     the transverse Mercator as what it IS: northing + i easting = k0 M(Phi), where M is the meridian arc length
            continued to complex latitude and Phi solves psi(Phi) = psi(phi) + i (lambda - lambda0) -- a complex Newton
            solve and a complex quadrature (Gauss-Krueger's definition; no Krueger series, no truncation order)
  ob_tran   the rotation of the sphere PROJ's ob_tran makes for o_proj=longlat (o_lat_p, o_lon_p, lon_0), degrees
  geodesic inverse azimuth: the exact direct problem (oracle/validate_geodesic.py: closed-form auxiliary-sphere integrals by
            quadrature) run forward from a known azimuth -- the inverse's answer is then known without solving it

Run:  python oracle/validate_projections.py      (writes tests/golden/proj_kat.npz; ~2 minutes)
"""
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
mp.mp.dps = 40

WGS84 = (mp.mpf(6378137), 1 / mp.mpf('298.257223563'))
GRS80 = (mp.mpf(6378137), 1 / mp.mpf('298.257222101'))
CLARKE66 = (mp.mpf('6378206.4'), 1 / mp.mpf('294.978698213898'))


def _e(f):
    es = f * (2 - f)
    return es, mp.sqrt(es)


def rad(d):
    return mp.radians(mp.mpf(float(d)))


def psi(phi, e):          # isometric latitude (complex arguments allowed)
    return mp.asinh(mp.tan(phi)) - e * mp.atanh(e * mp.sin(phi))


def m_(phi, es):
    return mp.cos(phi) / mp.sqrt(1 - es * mp.sin(phi) ** 2)


def t_(phi, e):
    s = mp.sin(phi)
    return mp.tan(mp.pi / 4 - phi / 2) / ((1 - e * s) / (1 + e * s)) ** (e / 2)


def chi_(phi, e):         # conformal latitude
    s = mp.sin(phi)
    return 2 * mp.atan(mp.tan(mp.pi / 4 + phi / 2) * ((1 - e * s) / (1 + e * s)) ** (e / 2)) - mp.pi / 2


def q_(phi, es, e):
    s = mp.sin(phi)
    return (1 - es) * (s / (1 - es * s * s) - mp.log((1 - e * s) / (1 + e * s)) / (2 * e))


def wrap(lam):
    return (lam + mp.pi) % (2 * mp.pi) - mp.pi


def merc(lon, lat, a, f, lon0, lat_ts, k0=1, x0=0, y0=0):
    es, e = _e(f)
    k = k0 * m_(rad(lat_ts), es)
    return x0 + a * k * wrap(rad(lon) - rad(lon0)), y0 + a * k * psi(rad(lat), e)


def lcc(lon, lat, a, f, lat0, lon0, lat1, lat2, k0=1, x0=0, y0=0):
    es, e = _e(f)
    p1, p2, p0 = rad(lat1), rad(lat2), rad(lat0)
    if lat1 == lat2:
        n = mp.sin(p1)
    else:
        n = mp.log(m_(p1, es) / m_(p2, es)) / mp.log(t_(p1, e) / t_(p2, e))
    F = m_(p1, es) / (n * t_(p1, e) ** n)
    rho0 = a * k0 * F * t_(p0, e) ** n
    rho = a * k0 * F * t_(rad(lat), e) ** n
    th = n * wrap(rad(lon) - rad(lon0))
    return x0 + rho * mp.sin(th), y0 + rho0 - rho * mp.cos(th)


def stere_polar(lon, lat, a, f, lat0, lon0, lat_ts, k0=1, x0=0, y0=0):
    es, e = _e(f)
    south = lat0 < 0
    phi, lam = rad(lat), wrap(rad(lon) - rad(lon0))
    pts = rad(abs(lat_ts))
    if south:
        phi = -phi
    t = t_(phi, e)
    if abs(lat_ts) == 90:
        rho = 2 * a * k0 * t / mp.sqrt((1 + e) ** (1 + e) * (1 - e) ** (1 - e))
    else:
        rho = a * k0 * m_(pts, es) * t / t_(pts, e)
    x, y = rho * mp.sin(lam), -rho * mp.cos(lam)
    if south:
        y = -y
    return x0 + x, y0 + y


def stere_oblique(lon, lat, a, f, lat0, lon0, k0=1, x0=0, y0=0):
    es, e = _e(f)
    p1 = rad(lat0)
    c1, c = chi_(p1, e), chi_(rad(lat), e)
    dl = wrap(rad(lon) - rad(lon0))
    A = 2 * a * k0 * m_(p1, es) / (mp.cos(c1) * (1 + mp.sin(c1) * mp.sin(c) + mp.cos(c1) * mp.cos(c) * mp.cos(dl)))
    return x0 + A * mp.cos(c) * mp.sin(dl), y0 + A * (mp.cos(c1) * mp.sin(c) - mp.sin(c1) * mp.cos(c) * mp.cos(dl))


def laea(lon, lat, a, f, lat0, lon0, x0=0, y0=0):
    es, e = _e(f)
    qp = q_(mp.pi / 2, es, e)
    p1 = rad(lat0)
    b1, b = mp.asin(q_(p1, es, e) / qp), mp.asin(q_(rad(lat), es, e) / qp)
    dl = wrap(rad(lon) - rad(lon0))
    Rq = a * mp.sqrt(qp / 2)
    if abs(lat0) == 90:        # polar aspect (Snyder 24-23 ... 24-25)
        sgn = 1 if lat0 > 0 else -1
        rho = a * mp.sqrt(qp - sgn * q_(rad(lat), es, e))
        return x0 + rho * mp.sin(dl), y0 - sgn * rho * mp.cos(dl)
    D = a * m_(p1, es) / (Rq * mp.cos(b1))
    B = Rq * mp.sqrt(2 / (1 + mp.sin(b1) * mp.sin(b) + mp.cos(b1) * mp.cos(b) * mp.cos(dl)))
    return x0 + B * D * mp.cos(b) * mp.sin(dl), y0 + (B / D) * (mp.cos(b1) * mp.sin(b) - mp.sin(b1) * mp.cos(b) * mp.cos(dl))


def _meridian(Phi, a, es):
    """a (1 - e^2) int_0^Phi (1 - e^2 sin^2 t)^(-3/2) dt along the straight path to the complex latitude Phi"""
    return a * (1 - es) * mp.quad(lambda s: Phi * (1 - es * mp.sin(Phi * s) ** 2) ** mp.mpf(-1.5), [0, 1])


def tmerc(lon, lat, a, f, lat0, lon0, k0=1, x0=0, y0=0):
    es, e = _e(f)
    zeta = psi(rad(lat), e) + 1j * wrap(rad(lon) - rad(lon0))
    guess = 2 * mp.atan(mp.exp(zeta)) - mp.pi / 2          # the sphere's answer (complex Gudermannian)
    Phi = mp.findroot(lambda P: psi(P, e) - zeta, guess, tol=mp.mpf(10) ** -34)
    w = k0 * (_meridian(Phi, a, es) - _meridian(rad(lat0), a, es))
    return x0 + mp.im(w), y0 + mp.re(w)


def ob_tran(lon, lat, lon0, o_lat_p, o_lon_p):
    """degrees in, degrees out (the reference converts pyproj's radians itself, variables.py:117-123)"""
    lam, phi = wrap(rad(lon) - rad(lon0)), rad(lat)
    phip, lamp = rad(o_lat_p), rad(o_lon_p)
    lam2 = wrap(mp.atan2(mp.cos(phi) * mp.sin(lam), mp.sin(phip) * mp.cos(phi) * mp.cos(lam) + mp.cos(phip) * mp.sin(phi)) + lamp)
    phi2 = mp.asin(mp.sin(phip) * mp.sin(phi) - mp.cos(phip) * mp.cos(phi) * mp.cos(lam))
    return mp.degrees(lam2), mp.degrees(phi2)


CASES = {
    # tag: (function, kwargs as plain floats, point box lon_min, lon_max, lat_min, lat_max)
    'merc_wgs84': (merc, dict(ell='wgs84', lon0=10.0, lat_ts=30.0), (-170, 190, -80, 84)),
    'lcc_wgs84': (lcc, dict(ell='wgs84', lat0=40.0, lon0=10.0, lat1=30.0, lat2=50.0), (-30, 50, 15, 75)),
    'lcc_1sp': (lcc, dict(ell='grs80', lat0=63.0, lon0=15.0, lat1=63.0, lat2=63.0, k0=0.99, x0=1000.0, y0=-2000.0), (-10, 40, 50, 78)),
    'stere_north': (stere_polar, dict(ell='wgs84', lat0=90.0, lon0=70.0, lat_ts=60.0, x0=1e5, y0=-2e5), (-180, 180, 50, 89.9)),
    'stere_north_ts90': (stere_polar, dict(ell='wgs84', lat0=90.0, lon0=0.0, lat_ts=90.0, k0=0.994), (-180, 180, 55, 89.9)),
    'stere_south': (stere_polar, dict(ell='wgs84', lat0=-90.0, lon0=-20.0, lat_ts=-71.0), (-180, 180, -89.9, -50)),
    'stere_oblique': (stere_oblique, dict(ell='wgs84', lat0=52.0, lon0=5.0, k0=0.9999, x0=155000.0, y0=463000.0), (-20, 30, 35, 70)),
    'stere_equatorial': (stere_oblique, dict(ell='grs80', lat0=0.0, lon0=-100.0, k0=1.0), (-130, -70, -30, 30)),
    'laea_europe': (laea, dict(ell='grs80', lat0=52.0, lon0=10.0, x0=4321000.0, y0=3210000.0), (-25, 45, 30, 72)),
    'laea_north': (laea, dict(ell='wgs84', lat0=90.0, lon0=-100.0), (-180, 180, 40, 89.9)),
    'utm33': (tmerc, dict(ell='wgs84', lat0=0.0, lon0=15.0, k0=0.9996, x0=500000.0), (9, 21, 0, 84)),
    'tmerc_wide': (tmerc, dict(ell='grs80', lat0=58.0, lon0=18.0, k0=1.0, x0=100.0, y0=-50.0), (0, 36, 45, 75)),
    'rotated_pole': (ob_tran, dict(lon0=-40.0, o_lat_p=25.0, o_lon_p=0.0), (-80, 0, 50, 85)),
}
ELL = {'wgs84': WGS84, 'grs80': GRS80, 'clarke66': CLARKE66}


def geodesic_inverse_cases(n, rng):
    """(lat1, lon1, lat2, lon2) as float64 numbers -> azimuth at point 1 of the geodesic between THOSE numbers: the exact direct
    problem run forward from a chosen azimuth gives point 2, point 2 is rounded to float64, and the inverse of the rounded pair is
    solved at working precision (Newton on the exact direct problem)."""
    sys.path.insert(0, HERE)
    import validate_geodesic as vg
    rows = []
    for k in range(n):
        lat1 = float(rng.uniform(-85, 85))
        lon1 = float(rng.uniform(-179, 179))
        azi = float(rng.uniform(-180, 180))
        s12 = float(10 ** rng.uniform(1.0, 4.2))     # 10 m (the line of rotate_vectors) ... 16 km (a 0.1-degree line)
        if k % 4 == 0:
            azi, s12 = (0.0 if k % 8 else 90.0), 10.0   # the reference's own line for projected readers (variables.py:85-97)
        lat2, lon2, _ = vg.exact_direct_mp(lat1, lon1, azi, s12)
        lat2f, lon2f = float(lat2), float(lon2)
        az, s = vg.exact_inverse_mp(lat1, lon1, lat2f, lon2f, azi, s12)
        rows.append((lat1, lon1, lat2f, lon2f, float(az), float(s)))
    return rows


def main():
    rng = np.random.default_rng(20260)
    out = {}
    n = 200
    for tag, (fn, kw, box) in CASES.items():
        kw = dict(kw)
        ell = kw.pop('ell', None)
        lon = rng.uniform(box[0], box[1], n)
        lat = rng.uniform(box[2], box[3], n)
        xs, ys = np.empty(n), np.empty(n)
        for i in range(n):
            if ell is not None:
                a, f = ELL[ell]
                x, y = fn(lon[i], lat[i], a, f, **kw)
            else:
                x, y = fn(lon[i], lat[i], **kw)
            xs[i], ys[i] = float(x), float(y)
        out[tag + '_lon'], out[tag + '_lat'], out[tag + '_x'], out[tag + '_y'] = lon, lat, xs, ys
        out[tag + '_kw'] = np.array(repr(dict(kw, **({'a': float(ELL[ell][0]), 'rf': float(1 / ELL[ell][1])} if ell else {}))))
        print(tag, 'x range %.3g .. %.3g' % (xs.min(), xs.max()), flush=True)
    rows = geodesic_inverse_cases(200, rng)
    out['inv_lat1'] = np.array([r[0] for r in rows]); out['inv_lon1'] = np.array([r[1] for r in rows])
    out['inv_lat2'] = np.array([r[2] for r in rows]); out['inv_lon2'] = np.array([r[3] for r in rows])
    out['inv_azi1'] = np.array([r[4] for r in rows]); out['inv_s12'] = np.array([r[5] for r in rows])
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'proj_kat.npz'), **out)
    print('wrote tests/golden/proj_kat.npz')


if __name__ == '__main__':
    main()
