"""The float64 device routines of csrc/odr_geodesic.hip.h, compiled for the CPU (tests/geod_host.cpp, g++ -ffp-contract=off), against
libm / mpmath and the oracle's complete geodesic (oracle/geodesic.c, itself pinned by the 34-digit KAT of tests/golden/geodesic_kat.npz):

* the range-limited sine / cosine (sincos_pi, sincosd), logarithm (log_pos), exponential (exp_small) and arctan2 (atan2_fin) the
  projections, the movers, the Leeway ladder and Box-Muller are built on -- a few ulp;
* the Legendre-series move (the update_positions of every particle-step, basemodel/__init__.py:4631-4657) against the complete
  solution inside its validity radius, and the fall-back outside it;
* the start point of a SECOND move formed from the first one's sine / cosine (geod_local_origin_next: advect_wind -> stokes_drift
  -> horizontal_diffusion, Leeway's two moves) against the one formed from scratch.

No GPU: the arithmetic is the same source the kernels compile (the hardware's reciprocal seeds replaced by exact operations, FMA
contraction off); tests/test_gpu_parity.py and tests/test_gpu_movers.py hold the device build to the same bounds."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope='module')
def gh():
    out = os.path.join(ROOT, 'oracle', '_build', 'geod_host.so')
    src = os.path.join(HERE, 'geod_host.cpp')
    deps = [src, os.path.join(HERE, 'hostshim', 'hip', 'hip_runtime.h'), os.path.join(ROOT, 'opendrift_amd', 'csrc', 'odr_geodesic.hip.h')]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(['g++', '-O1', '-std=c++17', '-ffp-contract=off', '-w', '-shared', '-fPIC', '-I' + os.path.join(HERE, 'hostshim'),
                               '-o', out, src])
    return C.CDLL(out)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _call1(lib, name, *ins, nout=1):
    ins = [np.ascontiguousarray(v, dtype=np.float64) for v in ins]
    outs = [np.empty_like(ins[0]) for _ in range(nout)]
    getattr(lib, name)(C.c_longlong(ins[0].size), *[_p(v) for v in ins], *[_p(v) for v in outs])
    return outs if nout > 1 else outs[0]


def _ulps(a, b):
    return np.abs(a - b) / np.spacing(np.maximum(np.abs(b), 1e-300))


def test_range_limited_elementary_functions(gh):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-3.3, 3.3, 200000), [0.0, -0.0, np.pi, -np.pi, np.pi / 2, np.pi / 4, 1e-300, -1e-9]])
    s, c = _call1(gh, 'gh_sincos_pi', x, nout=2)
    # absolute error below one ulp of 1 (relative accuracy near a zero of the function is limited by pi's two-term split: 1e-33)
    assert np.abs(s - np.sin(x)).max() < 2.3e-16 and np.abs(c - np.cos(x)).max() < 2.3e-16
    import mpmath as mp
    mp.mp.dps = 30
    d = np.concatenate([rng.uniform(-540, 540, 4000), [0.0, 90.0, -90.0, 180.0, 45.0, 30.0, 1e-12]])
    s, c = _call1(gh, 'gh_sincosd', d, nout=2)
    ref = [(mp.sin(mp.mpf(float(v)) * mp.pi / 180), mp.cos(mp.mpf(float(v)) * mp.pi / 180)) for v in d]     # (NumPy's radians() rounds first)
    assert max(abs(float(mp.mpf(float(a)) - r[0])) for a, r in zip(s, ref)) < 2.3e-16
    assert max(abs(float(mp.mpf(float(a)) - r[1])) for a, r in zip(c, ref)) < 2.3e-16
    assert s[-7] == 0 and c[-6] == 0 and s[-4] == 0          # exact quadrant reduction: sin 0, cos 90, sin 180
    u = np.concatenate([rng.uniform(0, 1, 200000), 2.0 ** -rng.uniform(0, 53, 100000), rng.uniform(1, 1e6, 1000), [1.0, 2.0 ** -53, 0.5, 0.70710678118654752]])
    assert _ulps(_call1(gh, "gh_log_pos", u), np.log(u))[np.log(u) != 0].max() <= 3.0     # (e ln 2 and ln m cancel by up to a factor 2 for x in [1/2, 0.71))
    assert _call1(gh, 'gh_log_pos', np.array([1.0]))[0] == 0.0
    e = rng.uniform(-0.0101, 0.0101, 200000)
    assert _ulps(_call1(gh, 'gh_exp_small', e), np.exp(e)).max() <= 1.0
    yy, xx = rng.normal(size=200000) * 10.0 ** rng.uniform(-6, 6, 200000), rng.normal(size=200000) * 10.0 ** rng.uniform(-6, 6, 200000)
    assert _ulps(_call1(gh, 'gh_atan2_fin', yy, xx), np.arctan2(yy, xx)).max() <= 2.0


def _move(gh, lat, lon, x, y, full=0):
    lat, lon, x, y = [np.ascontiguousarray(v, dtype=np.float64) for v in (lat, lon, x, y)]
    la, lo, ser = np.empty_like(lat), np.empty_like(lat), np.empty(lat.size, dtype=np.int32)
    gh.gh_move(C.c_longlong(lat.size), _p(lat), _p(lon), _p(x), _p(y), C.c_int(full), _p(la), _p(lo), _p(ser))
    return la, lo, ser


def _dlon(a, b):
    return np.abs((a - b + 180.0) % 360.0 - 180.0)


def test_series_move_equals_the_complete_geodesic(gh):
    """geod.fwd(lon, lat, degrees(arctan2(x, y)), hypot(x, y)) for steps of millimetres to tens of kilometres at all latitudes:
    the series where it claims validity (q = (s / N) max(1, |tan phi|) <= 2.5e-3: < 3e-12 deg), the complete solution otherwise."""
    rng = np.random.default_rng(11)
    n = 100000
    lat = np.concatenate([rng.uniform(-89.9, 89.9, n - 6), [0.0, 89.5, -89.5, 60.0, 1e-9, -45.0]])
    lon = rng.uniform(-360, 360, n)
    s = 10.0 ** rng.uniform(-3, 4.6, n)
    az = rng.uniform(-180, 180, n)
    x, y = s * np.sin(np.radians(az)), s * np.cos(np.radians(az))
    la, lo, ser = _move(gh, lat, lon, x, y)
    lo_o, la_o, _ = orc.geod_fwd(lon, lat, np.degrees(np.arctan2(x, y)), np.hypot(x, y))
    assert ser.mean() > 0.5 and (ser == 0).any()            # both lanes exercised
    inside = ser == 1
    q = s / 6.36e6 * np.maximum(1.0, np.abs(np.tan(np.radians(lat))))
    assert (q[inside] <= 2.6e-3).all() and ((q[~inside] >= 2.4e-3) | (np.abs(lat[~inside]) >= 89.0)).all()
    assert np.abs(la - la_o)[inside].max() < 3e-12 and (_dlon(lo, lo_o) * np.cos(np.radians(lat)))[inside].max() < 3e-12
    assert np.abs(la - la_o)[~inside].max() < 1e-11 and (_dlon(lo, lo_o) * np.cos(np.radians(lat)))[~inside].max() < 1e-11
    # a step of length zero leaves the element where it is (longitude normalised), as the kernels' early-out tests assume
    la0, lo0, _ = _move(gh, lat[:1000], lon[:1000], np.zeros(1000), np.zeros(1000))
    assert np.array_equal(la0, lat[:1000]) and (_dlon(lo0, lon[:1000]) < 1e-13).all()


def test_chained_start_point_equals_the_one_formed_from_scratch(gh):
    """advect_wind -> stokes_drift -> horizontal_diffusion: the second move starts where the first one ended.  Its coefficients from
    the first start point's sine / cosine (addition theorem) against sincosd of the new latitude: the same end point to the last
    bit or two of float64."""
    rng = np.random.default_rng(12)
    n = 200000
    lat, lon = rng.uniform(-88.5, 88.5, n), rng.uniform(-180, 180, n)
    x1, y1, x2, y2 = [rng.normal(size=n) * 10.0 ** rng.uniform(-2, 3.7, n) for _ in range(4)]
    args = [np.ascontiguousarray(v) for v in (lat, lon, x1, y1, x2, y2)]
    res = []
    for chained in (0, 1):
        la, lo = np.empty(n), np.empty(n)
        gh.gh_two_moves(C.c_longlong(n), *[_p(v) for v in args], C.c_int(chained), _p(la), _p(lo))
        res.append((la, lo))
    (la0, lo0), (la1, lo1) = res
    assert (la0 == la1).mean() > 0.98 and (lo0 == lo1).mean() > 0.98
    assert np.abs(la0 - la1).max() <= 3e-14 and _dlon(lo0, lo1).max() <= 6e-14
