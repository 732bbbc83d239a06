"""TEST INFRASTRUCTURE ONLY.  Pins oracle/geodesic.c (Karney direct) without pyproj.

pyproj/PROJ are un-vendored dependencies of the reference (pyproject.toml:18-19)
and are not installed here, so the restated series are pinned against two
independent high-precision computations (mpmath, 30+ digits):

 (A) the closed-form auxiliary-sphere integrals of Karney (2013) eqs. (7)-(8)
     evaluated by numerical quadrature + root finding  -> checks every series
     coefficient (A1, C1, C1', A3, C3);
 (B) direct integration of the geodesic ODE on the ellipsoid in (phi, lambda,
     alpha) -> checks the auxiliary-sphere formulation itself.

Run:  python oracle/validate_geodesic.py      (writes tests/golden/geodesic_kat.npz)
"""
import ctypes
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

mp.mp.dps = 34
A = mp.mpf(6378137)
F = 1 / mp.mpf('298.257223563')
E2 = F * (2 - F)
EP2 = E2 / (1 - F) ** 2
B = A * (1 - F)


def exact_direct(lat1, lon1, azi1, s12):
    """Closed-form integrals, Karney (2013) eqs. (7), (8), (10)-(12)."""
    return tuple(float(v) for v in exact_direct_mp(lat1, lon1, azi1, s12))


def exact_direct_mp(lat1, lon1, azi1, s12):
    """The same at working precision (mpmath numbers: degrees)."""
    phi1 = mp.radians(mp.mpf(lat1))
    al1 = mp.radians(mp.mpf(azi1))
    bet1 = mp.atan((1 - F) * mp.tan(phi1))
    sal0 = mp.sin(al1) * mp.cos(bet1)
    cal0 = mp.hypot(mp.cos(al1), mp.sin(al1) * mp.sin(bet1))
    sig1 = mp.atan2(mp.sin(bet1), mp.cos(al1) * mp.cos(bet1))
    omg1 = mp.atan2(sal0 * mp.sin(bet1), mp.cos(al1) * mp.cos(bet1))
    k2 = EP2 * cal0 ** 2

    def dist(sig2):
        return B * mp.quad(lambda s: mp.sqrt(1 + k2 * mp.sin(s) ** 2), [sig1, sig2])

    s12 = mp.mpf(s12)
    sig2 = mp.findroot(lambda s: dist(s) - s12, sig1 + s12 / B, tol=mp.mpf(10) ** -28)
    bet2 = mp.asin(cal0 * mp.sin(sig2))
    omg2 = mp.atan2(sal0 * mp.sin(sig2), mp.cos(sig2))
    I3 = mp.quad(lambda s: (2 - F) / (1 + (1 - F) * mp.sqrt(1 + k2 * mp.sin(s) ** 2)),
                 [sig1, sig2])
    lam12 = (omg2 - omg1) - F * sal0 * I3
    phi2 = mp.atan(mp.tan(bet2) / (1 - F))
    al2 = mp.atan2(sal0, cal0 * mp.cos(sig2))
    return (mp.degrees(phi2), mp.degrees(mp.radians(mp.mpf(lon1)) + lam12), mp.degrees(al2))


def exact_inverse_mp(lat1, lon1, lat2, lon2, azi_guess, s_guess):
    """Azimuth at point 1 and length of the geodesic between two points GIVEN AS float64 numbers: Newton on the exact direct
    problem from a nearby guess (short lines: the guess is the line the pair was made from)."""
    t1, t2 = mp.mpf(lat2), mp.mpf(lon2)

    def F(az, s):
        la, lo, _ = exact_direct_mp(lat1, lon1, az, s)
        return la - t1, (lo - t2) * mp.cos(mp.radians(t1))
    az, s = mp.findroot(F, (mp.mpf(azi_guess), mp.mpf(s_guess)), tol=mp.mpf(10) ** -26, maxsteps=30)
    return az, s


def ode_direct(lat1, lon1, azi1, s12):
    """Geodesic ODE d(phi,lam,alpha)/d(s/a) on the ellipsoid of revolution."""
    y0 = [mp.radians(mp.mpf(lat1)), mp.radians(mp.mpf(lon1)), mp.radians(mp.mpf(azi1))]

    def rhs(t, y):
        phi, lam, al = y
        w = mp.sqrt(1 - E2 * mp.sin(phi) ** 2)
        M = (1 - E2) / w ** 3
        N = 1 / w
        return [mp.cos(al) / M, mp.sin(al) / (N * mp.cos(phi)), mp.sin(al) * mp.tan(phi) / N]

    sol = mp.odefun(rhs, 0, y0, tol=mp.mpf(10) ** -26)
    y = sol(mp.mpf(s12) / A)
    return tuple(float(mp.degrees(v)) for v in y)


def wrap(d):
    return d - 360.0 * round(d / 360.0)


def main():
    from oracle import oracle as orc  # noqa: E402  (ctypes wrapper around the C oracle)
    rng = np.random.default_rng(20260925)
    rows = []
    worst_a = worst_b = 0.0
    # (A) 40 random cases incl. the regimes the hot path uses (cm .. 100 km) and long lines
    for k in range(40):
        lat = rng.uniform(-85, 85)
        lon = rng.uniform(-179, 179)
        az = rng.uniform(-180, 180)
        s = 10 ** rng.uniform(-2, 7.2)
        lo, la, a2 = orc.geod_fwd(lon, lat, az, s)
        ela, elo, ea2 = exact_direct(lat, lon, az, s)
        err = max(abs(la[0] - ela), abs(wrap(lo[0] - elo)))
        worst_a = max(worst_a, err)
        rows.append((lat, lon, az, s, ela, wrap(elo), wrap(ea2)))
        print(f"A s={s:14.3f} lat={lat:7.2f} az={az:8.2f} dlat={la[0]-ela: .2e} "
              f"dlon={wrap(lo[0]-elo): .2e} daz={wrap(a2[0]-ea2): .2e}")
    # (B) ODE cross-check of the formulation on a handful of short/medium lines
    for k in range(6):
        lat = rng.uniform(-75, 75)
        lon = rng.uniform(-170, 170)
        az = rng.uniform(-180, 180)
        s = 10 ** rng.uniform(1, 6)
        lo, la, a2 = orc.geod_fwd(lon, lat, az, s)
        ola, olo, oa2 = ode_direct(lat, lon, az, s)
        err = max(abs(la[0] - ola), abs(wrap(lo[0] - olo)))
        worst_b = max(worst_b, err)
        print(f"B s={s:14.3f} lat={lat:7.2f} az={az:8.2f} dlat={la[0]-ola: .2e} "
              f"dlon={wrap(lo[0]-olo): .2e}")
    print("worst |dlat|,|dlon| vs closed-form integrals [deg]:", worst_a)
    print("worst |dlat|,|dlon| vs ODE integration      [deg]:", worst_b)
    out = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'geodesic_kat.npz')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez(out, rows=np.array(rows))
    print("wrote", out)


if __name__ == '__main__':
    main()
