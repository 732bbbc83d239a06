"""The generic kernels (any mix of readers behind a priority list: k_env_group, k_advect, k_vmix, ordered
compaction) and the single-gridded-reader fast paths (k_env_grid, k_step_grid / k_advect_grid, k_vmix_col,
in-place compaction) must give the same bits: ODR_NO_FAST_PATH / ODR_ORDERED_COMPACT switch between them."""
import os

import numpy as np
import pytest

from scenarios import Scenario
from opendrift_amd import synthetic as synth

pytestmark = pytest.mark.gpu

U, V = 'x_sea_water_velocity', 'y_sea_water_velocity'
W, KZ = 'upward_sea_water_velocity', 'ocean_vertical_diffusivity'
DEPTH, SSH, LAND = 'sea_floor_depth_below_sea_level', 'sea_surface_height', 'land_binary_mask'
XW, YW = 'x_wind', 'y_wind'
SX, SY = 'sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity'


class generic:
    def __enter__(self):
        os.environ['ODR_NO_FAST_PATH'] = '1'

    def __exit__(self, *a):
        os.environ.pop('ODR_NO_FAST_PATH', None)


def _eq(a, b):
    return np.array_equal(a, b, equal_nan=True)


def _both(ctx, lon, lat, z, fn):
    """Run fn(P) on two identical particle sets, once per path; returns the two downloads."""
    out = []
    for use_generic in (False, True):
        P = ctx.particles(len(lon))
        P.append(lon, lat, z=z)
        if use_generic:
            with generic():
                extra = fn(P)
        else:
            extra = fn(P)
        out.append((P.download(), extra))
    return out


def test_grid3d_sample_advect_mix_same_bits(ctx):
    g = synth.grid3d(nx=96, ny=80, nz=8, nt=3, seed=5)
    names = [U, V, W, KZ, DEPTH, LAND]
    levels = [(float(g['t'][k]), {n: g[n][k] for n in names}) for k in range(3)]
    Scenario([('grid', dict(x=g['x'], y=g['y'], z=g['z'], levels=levels))],
             fallbacks={U: 0.0, V: 0.0, W: 0.0, KZ: 0.0, DEPTH: 10000.0, SSH: 0.0}).device(ctx)
    rng = np.random.default_rng(1)
    n = 40000
    lon = rng.uniform(g['x'][0] - 0.03, g['x'][-1] + 0.03, n)
    lat = rng.uniform(g['y'][0] - 0.03, g['y'][-1] + 0.03, n)
    z = -rng.uniform(0, 100, n)
    uni = rng.uniform(size=(10, n))

    def fn(P):
        env = {}
        for k, (t, scheme) in enumerate([(0.0, 'runge-kutta4'), (1500.0, 'runge-kutta'), (3600.0, 'euler'), (4000.0, 'runge-kutta4')]):
            env[k] = P.env_sample([U, V, W, KZ, DEPTH, SSH, LAND], t, download=True)
            P.advect(scheme, t, 600.0)
            P.vmix(t, 600.0, 60.0, uniforms=uni, fuse_vertical_advection=False)
        return env

    (a, ea), (b, eb) = _both(ctx, lon, lat, z, fn)
    for q in ('lon', 'lat', 'z'):
        assert _eq(a[q], b[q]), q
    for k in ea:
        for v in ea[k]:
            assert _eq(ea[k][v], eb[k][v]), (k, v)


def test_stere2d_sample_advect_same_bits(ctx):
    from oracle import oracle as orc
    g = synth.grid_stere(nx=120, ny=90, nt=3, seed=9)
    names = [U, V, XW, YW, SX, SY, LAND]
    levels = [(float(g['t'][k]), {n: g[n][k] for n in names}) for k in range(3)]
    Scenario([('grid', dict(x=g['x'], y=g['y'], proj=synth.NORKYST_PROJ, levels=levels))],
             fallbacks={U: 0.0, V: 0.0, XW: 0.0, YW: 0.0, SX: 0.0, SY: 0.0}).device(ctx)
    rng = np.random.default_rng(2)
    n = 30000
    p = orc.make_proj(orc.PROJ_STERE_POLAR, a=6371000.0, es=(2 - 1 / 298.257223563) / 298.257223563,
                      lat0=90.0, lon0=70.0, lat_ts=60.0)
    lon, lat = orc.proj_inv(p, rng.uniform(g['x'][0] - 300, g['x'][-1] + 300, n), rng.uniform(g['y'][0] - 300, g['y'][-1] + 300, n))

    def fn(P):
        env = {}
        for k, (t, scheme) in enumerate([(0.0, 'runge-kutta4'), (900.0, 'runge-kutta'), (3600.0, 'runge-kutta4')]):
            env[k] = P.env_sample(names, t, download=True)
            P.advect(scheme, t, 900.0)
        return env

    (a, ea), (b, eb) = _both(ctx, lon, lat, np.zeros(n), fn)
    assert _eq(a['lon'], b['lon']) and _eq(a['lat'], b['lat'])
    for k in ea:
        for v in ea[k]:
            assert _eq(ea[k][v], eb[k][v]), (k, v)


def test_ordered_and_in_place_compaction_keep_the_same_elements(ctx):
    rng = np.random.default_rng(3)
    n = 60000
    lon, lat = rng.uniform(0, 10, n), rng.uniform(60, 66, n)
    kill = rng.uniform(size=n) < 0.2
    res = []
    for ordered in (False, True):
        P = ctx.particles(n)
        P.append(lon, lat)
        P.env_upload(U, np.arange(n, dtype=np.float32))
        P.deactivate(kill, 2)
        if ordered:
            os.environ['ODR_ORDERED_COMPACT'] = '1'
        try:
            P.compact()
        finally:
            os.environ.pop('ODR_ORDERED_COMPACT', None)
        d, dead, u = P.download(), P.download_deactivated(), P.env_download(U)
        o = np.argsort(d['ID'])
        res.append((d['ID'][o], d['lon'][o], u[o], dead['ID'], dead['lon']))
    for x, y in zip(*res):
        assert _eq(x, y)
    assert (res[1][0] == np.nonzero(~kill)[0]).all()


def test_priority_list_of_two_grids_and_a_constant_matches_oracle(ctx):
    """Environment.get_environment with a priority list: a small high-resolution reader first (with a NaN coast
    and limited time coverage), a large coarse reader second, a constant reader third, then the fallback --
    the group semantics of environment.py:597-791 on the device against the oracle, bit for bit; and RK4 across
    the reader boundary."""
    from oracle import oracle as orc
    rng = np.random.default_rng(12)
    gi = synth.grid3d(nx=64, ny=48, nz=6, nt=3, seed=1, lon0=3.0, lon1=4.0, lat0=60.5, lat1=61.0)
    go = synth.grid3d(nx=40, ny=36, nz=6, nt=3, seed=2, lon0=1.0, lon1=7.0, lat0=59.0, lat1=63.0)
    li = [(float(gi['t'][k]), {U: gi[U][k], V: gi[V][k]}) for k in range(2)]       # inner: only two time levels
    lo = [(float(go['t'][k]), {U: go[U][k], V: go[V][k], W: go[W][k]}) for k in range(3)]
    sc = Scenario([('grid', dict(x=gi['x'], y=gi['y'], z=gi['z'], levels=li, time_coverage=(float(gi['t'][0]), float(gi['t'][1])))),
                   ('grid', dict(x=go['x'], y=go['y'], z=go['z'], levels=lo)),
                   ('constant', {U: 0.11, V: -0.07, XW: 4.0})],
                  fallbacks={U: 0.0, V: 0.0, W: 0.0, XW: 1.0, YW: 2.0})
    sc.device(ctx)
    n = 30000
    lon, lat, z = rng.uniform(0.5, 7.5, n), rng.uniform(58.8, 63.2, n), -rng.uniform(0, 60, n)
    P = ctx.particles(n)
    P.append(lon, lat, z=z)
    names = [U, V, W, XW, YW]
    for t in (0.0, 1800.0, float(gi['t'][1]) + 600.0):      # the last one is outside the inner reader's coverage
        w = sc.oracle_world()
        got = P.env_sample(names, t, download=True)
        ref = orc.get_environment(w, [orc.VAR[k] for k in names], lon, lat, z, t)
        for k, r in zip(names, ref):
            assert _eq(got[k], r), (k, t, int((~((got[k] == r) | (np.isnan(got[k]) & np.isnan(r)))).sum()))
    w = sc.oracle_world()
    lo_, la_ = lon.copy(), lat.copy()
    ue, ve = orc.get_environment(w, [0, 1], lo_, la_, z, 900.0)
    P.env_sample([U, V], 900.0)
    P.advect('runge-kutta4', 900.0, 600.0)
    orc.advect_ocean_current(w, 2, lo_, la_, z, np.ones(n, np.int32), np.ones(n, np.float32), ue, ve, 900.0, 600.0)
    d = P.download()
    assert np.abs(d['lon'] - lo_).max() < 1e-10 and np.abs(d['lat'] - la_).max() < 1e-10


def test_double_gyre_fast_path_against_generic(ctx):
    """The analytic double-gyre fast path (trig-free vector rotation, reduced-range sin / cos) against the generic
    source_sample() walk: environment within one float32 ulp (the formulas agree to float64 round-off, the cast may
    flip in rare cases), RK4 positions to 1e-12 degrees."""
    from opendrift_amd.projection import stere_equit_sphere_inverse
    sid = ctx.add_double_gyre(A=0.1, epsilon=0.25, omega=0.628, t0=0.0)
    ctx.bind(U, [sid], 0.0)
    ctx.bind(V, [sid], 0.0)
    ctx.bind(LAND, [sid], np.nan)
    rng = np.random.default_rng(3)
    n = 50000
    lon, lat = stere_equit_sphere_inverse(rng.uniform(-0.05, 2.05, n), rng.uniform(-0.05, 1.05, n), 6.371e6)

    def fn(P):
        e = P.env_sample([U, V, LAND], 3.3, download=True)
        P.env_sample([U, V], 3.3)
        P.advect('runge-kutta4', 3.3, 0.1)
        P.env_sample([U, V], 3.4)
        P.advect('runge-kutta', 3.4, 0.1)
        return e

    (a, ea), (b, eb) = _both(ctx, lon, lat, np.zeros(n), fn)
    assert _eq(ea[LAND], eb[LAND]) and np.isnan(ea[LAND]).any() and (ea[LAND] == 0).any()
    for k in (U, V):
        ulp = np.spacing(np.abs(eb[k]).astype(np.float32))
        assert (np.abs(ea[k] - eb[k]) <= ulp).all(), k
        assert (ea[k] != eb[k]).mean() < 0.01
    assert np.abs(a['lon'] - b['lon']).max() < 1e-12 and np.abs(a['lat'] - b['lat']).max() < 1e-12
