"""odr_movers: advect_wind -> stokes_drift -> horizontal_diffusion of one step in ONE launch (k_movers) against the three
entry points called one after the other (physics_methods.py:712-791, :793-848, basemodel/__init__.py:1746-1772): the same
positions (bit-identical for a single mover, to the last bit or two of float64 for a chain) for every subset of the movers, with host-drawn and device-drawn normals, on a polar-stereographic reader with
wind, Stokes drift and a constant diffusivity (the C4 shape); a mover whose global early-out holds (calm wind, no Stokes
drift, zero diffusivity) must be skipped as a whole in both lanes."""
import numpy as np
import pytest

from opendrift_amd import synthetic as synth
from opendrift_amd.device import Context

pytestmark = pytest.mark.gpu

U, V, LAND = 'x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask'
XW, YW = 'x_wind', 'y_wind'
SX, SY = 'sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity'
HD = 'horizontal_diffusivity'


def _setup(calm=False, no_stokes=False, hd=10.0, n=30000, seed=4):
    from opendrift_amd.projection import stere_polar_inverse
    g = synth.grid_stere(nx=260, ny=90, nt=3, seed=0)
    if calm:
        g[XW][:] = 0
        g[YW][:] = 0
    if no_stokes:
        g[SX][:] = 0
        g[SY][:] = 0
    names = [U, V, XW, YW, SX, SY, LAND]
    ctx = Context(seed=3)
    sid = ctx.add_grid(g['x'], g['y'], proj=synth.NORKYST_PROJ)
    for k in range(3):
        ctx.upload_block(sid, k, float(g['t'][k]), {nm: g[nm][k] for nm in names})
    for nm in names:
        ctx.bind(nm, [sid], np.nan if nm == LAND else 0.0)
    cs = ctx.add_constant({HD: hd})
    ctx.bind(HD, [cs], 0.0)
    rng = np.random.default_rng(seed)
    x = rng.uniform(g['x'][4], g['x'][-5], n)
    y = rng.uniform(g['y'][4], g['y'][-5], n)
    lon, lat = stere_polar_inverse(x, y, **synth.NORKYST_PROJ)
    z = np.where(rng.random(n) < 0.5, 0.0, -rng.uniform(0, 3, n))
    return ctx, names + [HD], lon, lat, z


def _pair(ctx, names, lon, lat, z, which, host_normals, steps=3, dt=900.0):
    n = len(lon)
    P, Q = ctx.particles(n), ctx.particles(n)
    rng = np.random.default_rng(77)
    for X in (P, Q):
        X.append(lon, lat, z=z)
    for k in range(steps):
        t = 900.0 * k
        normals = (rng.standard_normal(n), rng.standard_normal(n)) if host_normals else None
        for X in (P, Q):
            X.env_sample(names, t)
        if 'wind' in which:
            P.advect_wind(dt, wind_drift_depth=0.1)
        if 'stokes' in which:
            P.stokes_drift(dt, profile=2, hs_mode=1, tp_mode=1)
        if 'hdiff' in which:
            P.hdiffusion(dt, step=k, normals=normals)
        Q.movers(dt, wind=dict(wind_drift_depth=0.1) if 'wind' in which else None,
                 stokes=dict(profile=2, hs_mode=1, tp_mode=1) if 'stokes' in which else None,
                 hdiffusion=dict(step=k, normals=normals) if 'hdiff' in which else None)
        a, b = P.download(), Q.download()
        for q in ('lon', 'lat', 'z', 'ID'):
            eq = (a[q] == b[q]) | ((a[q] != a[q]) & (b[q] != b[q]))
            if q in ('lon', 'lat') and len(which) > 1:
                # the second / third move of the one launch forms its start-point coefficients from the previous start
                # latitude (move_f64_chain): equal to float64 round-off -- the odd last bit of a position
                worst = np.abs(a[q] - b[q])[~eq].max() if (~eq).any() else 0.0    # (a calm wind makes the wave height 0 and the Stokes drift NaN in both lanes)
                assert eq.mean() > 0.98 and worst <= 6e-14, (which, k, q, int((~eq).sum()), worst)
                continue
            assert eq.all(), (which, k, q, int((~eq).sum()))
    moved = bool((a['lon'] != lon).any())
    P.close()
    Q.close()
    return moved


@pytest.mark.parametrize('host_normals', [False, True])
@pytest.mark.parametrize('which', [('wind', 'stokes', 'hdiff'), ('wind', 'stokes'), ('stokes', 'hdiff'), ('wind',), ('hdiff',)])
def test_movers_in_one_launch_equal_the_separate_calls(which, host_normals):
    ctx, names, lon, lat, z = _setup()
    assert _pair(ctx, names, lon, lat, z, which, host_normals)
    ctx.close()


@pytest.mark.parametrize('case', ['calm', 'no_stokes', 'no_diffusivity', 'nothing'])
def test_movers_keep_the_global_early_outs(case):
    """wind_speed.max() == 0, stokes max == 0, D.max() == 0: the mover returns before update_positions (which would
    renormalise the longitude) -- in the fused launch as in the separate calls."""
    ctx, names, lon, lat, z = _setup(calm=case in ('calm', 'nothing'), no_stokes=case in ('no_stokes', 'nothing'),
                                     hd=0.0 if case in ('no_diffusivity', 'nothing') else 10.0, n=8000)
    moved = _pair(ctx, names, lon, lat, z, ('wind', 'stokes', 'hdiff'), False, steps=2)
    assert moved == (case != 'nothing')
    ctx.close()


@pytest.mark.parametrize('case', ['all', 'calm', 'no_stokes', 'no_diffusivity', 'nothing', 'deep', 'relative_wind'])
def test_movers_tests_formed_by_the_step_launch(case, monkeypatch):
    """odr_ctx_set_step_reduce: the launch of odr_env_coast_advect forms the movers' global tests (no element at the surface,
    wind_drift_factor / wind speed / Stokes drift / diffusivity identically zero) from the values it holds in registers;
    the movers that follow -- a compaction in between -- take them from there instead of from a pass over the arrays
    (k_reduce).  Same bits as without the setting, in every combination of tests that hold (calm, no Stokes drift, zero
    diffusivity, every element below the wind drift depth, wind relative to the current), with stranding in the launch."""
    ctx, names, lon, lat, z = _setup(calm=case in ('calm', 'nothing'), no_stokes=case in ('no_stokes', 'nothing'),
                                     hd=0.0 if case in ('no_diffusivity', 'nothing') else 10.0, n=20000)
    if case == 'deep':
        z = z - 5.0
    rel = case == 'relative_wind'
    n = len(lon)
    out = []
    launches = []
    for on in (False, True):
        ctx.set_step_reduce(on, wind_drift_depth=0.1, relative_wind=rel)
        P = ctx.particles(n)
        P.append(lon, lat, z=z)
        for k in range(3):
            P.env_coast_advect(names, 900.0 * k, 'runge-kutta4', 900.0, coastline='stranding', stranded_code=1, store_previous=False)
            P.compact()
            P.movers(900.0, wind=dict(wind_drift_depth=0.1, relative_wind=rel), stokes=dict(profile=2, hs_mode=1, tp_mode=1),
                     hdiffusion=dict(step=k))
        d = P.download()
        o = np.argsort(d['ID'])
        out.append({q: d[q][o] for q in ('lon', 'lat', 'z', 'ID')})
        P.close()
    ctx.set_step_reduce(False)
    for q in out[0]:
        assert np.array_equal(out[0][q], out[1][q], equal_nan=True), (case, q)
    assert len(out[0]['ID']) < n       # elements stranded and were compacted away between the launch and the movers
    ctx.close()


def test_a_different_wind_drift_depth_or_mixing_in_between_falls_back_to_the_pass(monkeypatch):
    """The tests of the launch are keyed by (wind_drift_depth, relative_wind) and by the state they were formed on: movers
    called with another depth, or after something changed z, make the pass over the arrays as before -- same results."""
    ctx, names, lon, lat, z = _setup(n=12000)
    n = len(lon)
    res = []
    for on in (False, True):
        ctx.set_step_reduce(on, wind_drift_depth=0.1)
        P = ctx.particles(n)
        P.append(lon, lat, z=z)
        P.env_coast_advect(names, 0.0, 'euler', 900.0, coastline='none', store_previous=False)
        P.movers(900.0, wind=dict(wind_drift_depth=2.0), stokes=dict(profile=2, hs_mode=1, tp_mode=1), hdiffusion=dict(step=0))
        P.env_coast_advect(names, 900.0, 'euler', 900.0, coastline='none', store_previous=False)
        P.advect_wind(900.0, wind_drift_depth=0.1)
        d = P.download()
        res.append((d['lon'][np.argsort(d['ID'])], d['lat'][np.argsort(d['ID'])]))
        P.close()
    ctx.set_step_reduce(False)
    assert np.array_equal(res[0][0], res[1][0], equal_nan=True) and np.array_equal(res[0][1], res[1][1], equal_nan=True)
    ctx.close()


@pytest.mark.parametrize('what', ['wind', 'stokes'])
def test_one_nonzero_element_in_any_lane_keeps_the_mover_alive(what):
    """The launch reduces over the 64 lanes of a wave with DPP moves (row shifts, row broadcasts): a single element with
    wind (or Stokes drift) -- in lane 0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 62 or 63 of its wave, everything else calm --
    must keep the mover from returning early, exactly as the pass over the arrays decides."""
    from opendrift_amd.projection import stere_polar_inverse
    g = synth.grid_stere(nx=260, ny=90, nt=2, seed=0)
    for nm in (XW, YW, SX, SY, U, V, LAND):
        g[nm][:] = 0
    names = [U, V, XW, YW, SX, SY, LAND]
    n = 64 * 7 + 13
    jj, ii = np.divmod(np.arange(n), 100)
    jj, ii = 10 + 3 * jj, 20 + 2 * ii                      # every element on a node of its own
    xs, ys = g['x'][ii].astype(np.float64), g['y'][jj].astype(np.float64)
    lon, lat = stere_polar_inverse(xs, ys, **synth.NORKYST_PROJ)
    for lane in (0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 62, 63):
        k = 64 * 3 + lane
        f = {nm: g[nm].copy() for nm in names}
        for nm in ((XW, YW) if what == 'wind' else (SX, SY)):
            f[nm][:, jj[k], ii[k]] = 4.0
        res = []
        for on in (False, True):
            ctx = Context(seed=3)
            sid = ctx.add_grid(g['x'], g['y'], proj=synth.NORKYST_PROJ)
            for t in range(2):
                ctx.upload_block(sid, t, float(g['t'][t]), {nm: f[nm][t] for nm in names})
            for nm in names:
                ctx.bind(nm, [sid], 0.0)
            ctx.set_step_reduce(on, wind_drift_depth=0.1)
            P = ctx.particles(n)
            P.append(lon, lat, z=np.zeros(n))
            P.env_coast_advect(names, 0.0, 'euler', 600.0, coastline='none', store_previous=False)
            P.movers(600.0, wind=dict(wind_drift_depth=0.1), stokes=dict(profile=0, hs_mode=2, tp_mode=2))
            d = P.download()
            res.append((d['lon'].copy(), d['lat'].copy()))
            P.close()
            ctx.close()
        assert np.array_equal(res[0][0], res[1][0], equal_nan=True) and np.array_equal(res[0][1], res[1][1], equal_nan=True), lane
        moved = np.flatnonzero((res[1][0] != lon) | (res[1][1] != lat))
        assert k in moved, (lane, moved[:5])
