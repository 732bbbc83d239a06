#!/usr/bin/env python
"""Derives the Legendre series of the direct geodesic problem used by csrc/odr_geodesic.hip.h (geod_local_*) and checks
its truncation error against the full Karney solution of the CPU oracle.

The geodesic equations on the ellipsoid, with t = tan(phi), eta^2 = e'^2 cos^2(phi), V^2 = 1 + eta^2, N = c / V:
    dphi/ds = cos(alpha) V^2 / N,   dlam/ds = sin(alpha) / (N cos phi),   dalpha/ds = sin(alpha) t / N
are differentiated repeatedly (sympy) along the geodesic; the k-th derivative is homogeneous of degree k in
(cos alpha, sin alpha), so with u = s cos(alpha) / N, v = s sin(alpha) / N the Taylor series in s becomes a polynomial
in (u, v) whose coefficients depend on the start latitude only.

    python tools/derive_geodesic_series.py            # print the coefficients to 5th order
    python tools/derive_geodesic_series.py --check    # truncation error of orders 3, 4, 5 vs the oracle
"""
import sys

import numpy as np


def derive(order=5):
    import sympy as sp
    t, h, N, c, C, S = sp.symbols('t eta2 N c C S')
    V2 = 1 + h
    fphi, flam, fal = C * V2 / N, S / (N * c), S * t / N

    def D(g):   # d/ds along the geodesic: dt/dphi = 1 + t^2, d eta2/dphi = -2 eta2 t, dc/dphi = -t c, dN/dphi = N eta2 t / V^2
        dphi = sp.diff(g, t) * (1 + t**2) + sp.diff(g, h) * (-2 * h * t) + sp.diff(g, c) * (-t * c) + sp.diff(g, N) * (N * h * t / V2)
        dal = sp.diff(g, C) * (-S) + sp.diff(g, S) * C
        return sp.together(dphi * fphi + dal * fal)
    dphi, dlam = [fphi], [flam]
    for _ in range(1, order):
        dphi.append(sp.factor(D(dphi[-1])))
        dlam.append(sp.factor(D(dlam[-1])))
    out = {}
    for name, lst in (('phi', dphi), ('lam', dlam)):
        for k, g in enumerate(lst, 1):
            P = sp.Poly(sp.expand(sp.simplify(g * N**k / sp.factorial(k))), C, S)
            for (i, j), co in P.terms():
                co = sp.simplify(co * c) if name == 'lam' else sp.simplify(co / V2)   # dlam carries 1/cos(phi), dphi carries V^2
                out[(name, i, j)] = sp.factor(co)
    return out


def series(lon, lat, az, s, order=4):
    """NumPy evaluation of the series (the device code is csrc/odr_geodesic.hip.h:geod_local_move)."""
    a, f = 6378137.0, 1 / 298.257223563
    e2 = f * (2 - f)
    ep2 = e2 / (1 - e2)
    phi = np.radians(lat)
    t, c = np.tan(phi), np.cos(phi)
    h = ep2 * c * c
    N = a / np.sqrt(1 - e2 * np.sin(phi)**2)
    u, v = s * np.cos(np.radians(az)) / N, s * np.sin(np.radians(az)) / N
    p = u - 1.5 * h * t * u * u - 0.5 * t * v * v + h * (5 * h * t * t - h + t * t - 1) / 2 * u**3 + (9 * h * t * t - h - 3 * t * t - 1) / 6 * u * v * v
    l = v + t * u * v + (h + 3 * t * t + 1) / 3 * u * u * v - t * t / 3 * v**3
    if order >= 4:
        p += (-h * t * (35 * h * h * t * t - 19 * h * h + 15 * h * t * t - 23 * h - 4) / 8 * u**4
              - t * (45 * h * h * t * t - 17 * h * h - 9 * h * t * t - 13 * h + 6 * t * t + 4) / 12 * u * u * v * v
              - t * (9 * h * t * t - h - 3 * t * t - 1) / 24 * v**4)
        l += t * (-h * h + h + 3 * t * t + 2) / 3 * u**3 * v - t * (h + 3 * t * t + 1) / 3 * u * v**3
    if order >= 5:
        p += (h * (315 * h**3 * t**4 - 314 * h**3 * t**2 + 19 * h**3 + 210 * h**2 * t**4 - 452 * h**2 * t**2 + 42 * h**2 + 15 * h * t**4 - 142 * h * t**2 + 27 * h - 4 * t**2 + 4) / 40 * u**5
              + (525 * h**3 * t**4 - 402 * h**3 * t**2 + 17 * h**3 - 354 * h**2 * t**2 + 30 * h**2 + 45 * h * t**4 + 18 * h * t**2 + 9 * h - 30 * t**4 - 30 * t**2 - 4) / 60 * u**3 * v**2
              + (225 * h**2 * t**4 - 102 * h**2 * t**2 + h**2 - 90 * h * t**4 - 72 * h * t**2 + 2 * h + 45 * t**4 + 30 * t**2 + 1) / 120 * u * v**4)
        l += ((6 * h**3 * t**2 - h**3 - 3 * h**2 * t**2 + 6 * h * t**2 + 3 * h + 15 * t**4 + 15 * t**2 + 2) / 15 * u**4 * v
              - (-7 * h**2 * t**2 + h**2 + 13 * h * t**2 + 2 * h + 30 * t**4 + 20 * t**2 + 1) / 15 * u**2 * v**3 + t**2 * (h + 3 * t**2 + 1) / 15 * v**5)
    return lon + np.degrees(l / c), lat + np.degrees((1 + h) * p)


def check():
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    n = 20000
    for latmax in (60, 80, 88):
        for smax in (100., 1000., 5000., 20000.):
            lon, lat = rng.uniform(-180, 180, n), rng.uniform(-latmax, latmax, n)
            az, s = rng.uniform(-180, 180, n), rng.uniform(0, smax, n)
            lo, la, _ = orc.geod_fwd(lon, lat, az, s)
            row = []
            for order in (3, 4, 5):
                l2, a2 = series(lon, lat, az, s, order)
                dl = (l2 - lo + 180) % 360 - 180
                row.append('order %d: %.1e / %.1e' % (order, np.abs(dl * np.cos(np.radians(lat))).max(), np.abs(a2 - la).max()))
            print('|lat| < %d, s < %6.0f m: max |dlon cos(lat)| / |dlat| [deg]  ' % (latmax, smax) + '   '.join(row))
    # the validity criterion of the device code: q = (s / N) max(1, |tan phi|) <= 4e-3
    lat = rng.uniform(-88.9, 88.9, 200000)
    lon, az = rng.uniform(-180, 180, lat.size), rng.uniform(-180, 180, lat.size)
    N = 6378137.0 / np.sqrt(1 - 0.00669437999014 * np.sin(np.radians(lat))**2)
    s = rng.uniform(0.5, 1.0, lat.size) * 2.5e-3 * N / np.maximum(1.0, np.abs(np.tan(np.radians(lat))))
    lo, la, _ = orc.geod_fwd(lon, lat, az, s)
    l2, a2 = series(lon, lat, az, s, 4)
    dl = (l2 - lo + 180) % 360 - 180
    print('at 0.5..1 x the validity limit (q <= 2.5e-3): max |dlon cos(lat)| %.2e deg, |dlat| %.2e deg, steps %.0f..%.0f m'
          % (np.abs(dl * np.cos(np.radians(lat))).max(), np.abs(a2 - la).max(), s.min(), s.max()))


if __name__ == '__main__':
    if '--check' in sys.argv:
        check()
    else:
        for k, v in derive().items():
            print('%s u^%d v^%d: %s' % (k + (v,)))
