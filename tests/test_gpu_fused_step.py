"""odr_env_coast_advect (one launch: get_environment -> interact_with_coastline ->
update_previous_state -> advect_ocean_current) must equal the four separate C-ABI calls bit for
bit, on every path: fused kernel (gridded reader, lat/lon 3-D and polar-stereographic 2-D) and the
sequential fallback (analytic reader)."""
import numpy as np
import pytest

from scenarios import Scenario
from opendrift_amd import synthetic as synth
from opendrift_amd.device import Context

pytestmark = pytest.mark.gpu

U, V = 'x_sea_water_velocity', 'y_sea_water_velocity'
W, KZ = 'upward_sea_water_velocity', 'ocean_vertical_diffusivity'
DEPTH, SSH, LAND = 'sea_floor_depth_below_sea_level', 'sea_surface_height', 'land_binary_mask'
XW, YW = 'x_wind', 'y_wind'


def _same(a, b, what):
    for k in ('lon', 'lat', 'z', 'status', 'moving', 'ID'):
        eq = (a[k] == b[k]) | ((a[k] != a[k]) & (b[k] != b[k]))
        assert eq.all(), (what, k, int((~eq).sum()))


def _run_pair(ctx, lon, lat, z, variables, times, dt, coastline, codes, schemes):
    n = len(lon)
    P, Q = ctx.particles(n), ctx.particles(n)
    for X in (P, Q):
        X.append(lon, lat, z=z)
        X.store_previous()
    for k, t in enumerate(times):
        scheme = schemes[k % len(schemes)]
        P.env_sample(variables, t)              # the run() loop order of the reference
        n1 = P.coastline(coastline, **codes)
        P.compact()
        P.store_previous()
        P.advect(scheme, t, dt)
        n2 = Q.env_coast_advect(variables, t, scheme, dt, coastline=coastline, store_previous=True, **codes)
        Q.compact()
        assert n1 == n2 and len(P) == len(Q), (k, n1, n2, len(P), len(Q))
        _same(P.download(), Q.download(), 'step %d' % k)
        da, db = P.download_deactivated(), Q.download_deactivated()
        for q in ('lon', 'lat', 'z', 'status', 'ID'):
            assert (da[q] == db[q]).all(), (k, 'deactivated', q)
        for v in variables:
            a, b = P.env_download(v), Q.env_download(v)
            assert ((a == b) | (np.isnan(a) & np.isnan(b))).all(), (k, v)
        for X in (P, Q):
            X.increase_age(dt)
    return n1


def test_fused_equals_separate_latlon_3d_previous(ctx):
    g = synth.grid3d(nx=96, ny=80, nz=8, nt=3, seed=5)
    names = [U, V, W, KZ, DEPTH, LAND]
    levels = [(float(g['t'][k]), {n: g[n][k] for n in names}) for k in range(3)]
    sc = Scenario([('grid', dict(x=g['x'], y=g['y'], z=g['z'], levels=levels)), ('constant', {XW: 3.0, YW: -2.0})],
                  fallbacks={U: 0.0, V: 0.0, W: 0.0, KZ: 0.0, DEPTH: 10000.0, SSH: 0.0})
    sc.device(ctx)
    rng = np.random.default_rng(3)
    n = 50000
    lon = rng.uniform(g['x'][0] - 0.02, g['x'][-1] + 0.02, n)
    lat = rng.uniform(g['y'][0] - 0.02, g['y'][-1] + 0.02, n)
    z = -rng.uniform(0, 80, n)
    hits = _run_pair(ctx, lon, lat, z, [U, V, W, DEPTH, SSH, LAND, XW, YW], [0.0, 600.0, 1800.0, 3000.0, 3600.0, 4200.0],
                     600.0, 'previous', dict(seeded_on_land_code=5), ['runge-kutta4', 'runge-kutta', 'euler'])
    assert hits >= 0


def test_fused_equals_separate_stere_2d_stranding(ctx):
    from oracle import oracle as orc
    g = synth.grid_stere(nx=120, ny=90, nt=3, seed=9)
    names = [U, V, XW, YW, LAND]
    levels = [(float(g['t'][k]), {n: g[n][k] for n in names}) for k in range(3)]
    sc = Scenario([('grid', dict(x=g['x'], y=g['y'], proj=synth.NORKYST_PROJ, levels=levels))],
                  fallbacks={U: 0.0, V: 0.0, XW: 0.0, YW: 0.0})
    sc.device(ctx)
    rng = np.random.default_rng(4)
    n = 40000
    p = orc.make_proj(orc.PROJ_STERE_POLAR, a=6371000.0, es=(2 - 1 / 298.257223563) / 298.257223563,
                      lat0=90.0, lon0=70.0, lat_ts=60.0)
    lon, lat = orc.proj_inv(p, rng.uniform(g['x'][1], g['x'][-2], n), rng.uniform(g['y'][1], g['y'][-2], n))
    hits = _run_pair(ctx, lon, lat, np.zeros(n), [U, V, XW, YW, LAND], [0.0, 900.0, 1800.0, 3600.0], 900.0,
                     'stranding', dict(stranded_code=3), ['runge-kutta4', 'runge-kutta'])
    assert hits > 0      # the synthetic coast strands some elements: the flagged ones must not have moved


def test_fallback_path_equals_separate_analytic(ctx):
    sid = ctx.add_double_gyre(A=0.1, epsilon=0.25, omega=0.628, t0=0.0)
    ctx.bind(U, [sid], 0.0)
    ctx.bind(V, [sid], 0.0)
    from opendrift_amd.projection import stere_equit_sphere_inverse
    rng = np.random.default_rng(6)
    n = 20000
    lon, lat = stere_equit_sphere_inverse(rng.uniform(0.05, 1.95, n), rng.uniform(0.05, 0.95, n), 6.371e6)
    _run_pair(ctx, lon, lat, np.zeros(n), [U, V], [0.0, 0.1, 0.2], 0.1, 'none', {}, ['runge-kutta4'])


def test_missing_current_is_an_error(ctx):
    P = ctx.particles(4)
    P.append(np.zeros(4), np.zeros(4))
    with pytest.raises(ValueError):
        P.env_coast_advect([XW, YW], 0.0, 'euler', 600.0)


def test_fused_with_seafloor_and_age_equals_separate(ctx):
    """The optional bookkeeping of the loop body inside the fused launch: interact_with_seafloor (lift_to_seafloor)
    and increase_age_and_retire between the coastline and update_previous_state -- same bits as the separate calls
    in the reference's order, retired / seeded-on-land elements do not move."""
    g = synth.grid3d(nx=96, ny=80, nz=8, nt=3, seed=5)
    names = [U, V, W, KZ, DEPTH, LAND]
    arrays = {n: g[n] for n in names}
    arrays[DEPTH] = (np.asarray(g[DEPTH]) * 0 + 25.0).astype(np.float32)      # shallow: many elements start below the floor
    levels = [(float(g['t'][k]), {n: arrays[n][k] for n in names}) for k in range(3)]
    Scenario([('grid', dict(x=g['x'], y=g['y'], z=g['z'], levels=levels))],
             fallbacks={U: 0.0, V: 0.0, W: 0.0, KZ: 0.0, DEPTH: 10000.0, SSH: 0.0}).device(ctx)
    rng = np.random.default_rng(9)
    n = 40000
    lon = rng.uniform(g['x'][2], g['x'][-3], n)
    lat = rng.uniform(g['y'][2], g['y'][-3], n)
    z = -rng.uniform(0, 60, n)
    variables = [U, V, W, DEPTH, SSH, LAND]
    P, Q = ctx.particles(n), ctx.particles(n)
    for X in (P, Q):
        X.append(lon, lat, z=z)
        X.store_previous()
    dt, max_age = 600.0, 1500.0
    for k, t in enumerate([0.0, 600.0, 1200.0, 1800.0]):
        P.env_sample(variables, t)
        P.coastline('previous', seeded_on_land_code=5)
        P.seafloor()
        P.increase_age(dt, max_age, 7)
        P.compact()
        P.store_previous()
        P.advect('runge-kutta4', t, dt)
        Q.env_coast_advect(variables, t, 'runge-kutta4', dt, coastline='previous', seeded_on_land_code=5, store_previous=True,
                           seafloor=True, age_dt=dt, max_age_seconds=max_age, retired_code=7)
        Q.compact()
        assert len(P) == len(Q), (k, len(P), len(Q))
        _same(P.download(), Q.download(), 'step %d' % k)
        da, db = P.download_deactivated(), Q.download_deactivated()
        for q in ('lon', 'lat', 'z', 'status', 'ID'):
            assert (da[q] == db[q]).all(), (k, 'deactivated', q)
    assert len(P) == 0 or (P.download()['z'] >= -25.0 - 1e-9).all()
    assert (Q.download_deactivated()['status'] == 7).any()


@pytest.mark.parametrize('scheme', ['euler', 'runge-kutta4'])
@pytest.mark.parametrize('vadv', [None, False, True])
def test_mixing_inside_the_step_launch_gives_the_same_bits_as_two_calls(monkeypatch, scheme, vadv):
    """odr_step_extras.vmix: OceanDrift.vertical_mixing (+ vertical_advection) inside k_step_grid<..., MIXQ> -- K column
    gathered at the sample position while the particle is in registers -- against odr_env_coast_advect followed by
    odr_vmix (device RNG keyed by element ID and step): bit-identical lon / lat / z / status, also with the sea floor
    in reach, a coastline and a reader time level crossed between the stages."""
    g = synth.grid3d(nx=96, ny=80, nz=8, nt=3, seed=5)
    g[DEPTH][:] = np.minimum(g[DEPTH], 60.0 + 100.0 * np.linspace(0, 1, 96)[None, None, :]).astype(np.float32)
    names = [U, V, W, KZ, DEPTH, LAND]
    rng = np.random.default_rng(12)
    n = 30000
    lon = rng.uniform(g['x'][3], g['x'][-4], n)
    lat = rng.uniform(g['y'][3], g['y'][-4], n)
    z = -rng.uniform(0, 70, n)
    z[:2000] = 0.0
    res = []
    for fused in (True, False):
        if fused:
            monkeypatch.setenv('ODR_FUSED_MIX', '1')
            monkeypatch.delenv('ODR_NO_FUSED_MIX', raising=False)
        else:
            monkeypatch.setenv('ODR_NO_FUSED_MIX', '1')
        ctx = Context(seed=4)
        sid = ctx.add_grid(g['x'], g['y'], z=g['z'])
        for k in range(3):
            ctx.upload_block(sid, k, float(g['t'][k]), {nm: g[nm][k] for nm in names})
        for nm in names:
            ctx.bind(nm, [sid], {LAND: np.nan, DEPTH: 10000.0}.get(nm, 0.0))
        ctx.bind(SSH, [], 0.0)
        P = ctx.particles(n)
        P.append(lon, lat, z=z, terminal_velocity=np.where(np.arange(n) % 3 == 0, -0.004, 0.001).astype(np.float32))
        P.sort_by_cell(sid)
        for k, t in enumerate((0.0, 900.0, 3300.0, 3600.0)):
            P.env_coast_advect([U, V, W, DEPTH, SSH, LAND], t, scheme, 600.0, coastline='previous', count=False, seafloor=True,
                               age_dt=600.0, vmix=dict(dt_mix=60.0, step=k, vertical_advection=vadv))
        d = P.download()
        o = np.argsort(d['ID'])
        res.append(tuple(d[q][o] for q in ('lon', 'lat', 'z', 'status', 'moving')))
        P.close()
        ctx.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    assert np.abs(res[0][2] - np.sort(z)[::1][0] * 0).max() > 1.0 and (res[0][2] <= 0).all()


@pytest.mark.parametrize('lanes', [2, 5])
def test_step_in_lanes_gives_the_same_bits_as_one_launch(monkeypatch, lanes):
    """ODR_LANES: the step + mixing of contiguous particle windows on streams of their own (step_in_lanes, odr_step.hip),
    so that the mixing kernel of one window and the step kernel of the next are resident together.  Per-particle kernels,
    RNG keyed by element ID: bit-identical lon / lat / z / status / environment to the single launch, also with a last
    window that is not a multiple of the workgroup size."""
    g = synth.grid3d(nx=96, ny=80, nz=8, nt=3, seed=5)
    g[DEPTH][:] = np.minimum(g[DEPTH], 60.0 + 100.0 * np.linspace(0, 1, 96)[None, None, :]).astype(np.float32)
    names = [U, V, W, KZ, DEPTH, LAND]
    rng = np.random.default_rng(13)
    n = 30011
    lon = rng.uniform(g['x'][3], g['x'][-4], n)
    lat = rng.uniform(g['y'][3], g['y'][-4], n)
    z = -rng.uniform(0, 70, n)
    monkeypatch.setenv('ODR_LANES_MIN_N', '0')
    res = []
    for nl in (lanes, 1):
        monkeypatch.setenv('ODR_LANES', str(nl))
        ctx = Context(seed=4)
        sid = ctx.add_grid(g['x'], g['y'], z=g['z'])
        for k in range(3):
            ctx.upload_block(sid, k, float(g['t'][k]), {nm: g[nm][k] for nm in names})
        for nm in names:
            ctx.bind(nm, [sid], {LAND: np.nan, DEPTH: 10000.0}.get(nm, 0.0))
        ctx.bind(SSH, [], 0.0)
        P = ctx.particles(n)
        P.append(lon, lat, z=z, terminal_velocity=np.where(np.arange(n) % 3 == 0, -0.004, 0.001).astype(np.float32))
        P.sort_by_cell(sid)
        for k, t in enumerate((0.0, 900.0, 3300.0, 3600.0)):
            P.env_coast_advect([U, V, W, DEPTH, SSH, LAND], t, 'runge-kutta4', 600.0, coastline='previous', count=False,
                               seafloor=True, age_dt=600.0, vmix=dict(dt_mix=60.0, step=k, vertical_advection=True))
            if k == 1:
                P.sort_by_cell(sid, keep_environment=False)
        d = P.download()
        o = np.argsort(d['ID'])
        res.append(tuple(d[q][o] for q in ('lon', 'lat', 'z', 'status', 'moving')) + (P.download_f32('age_seconds')[o],) +
                   tuple(P.env_download(q)[o] for q in (U, W, DEPTH)))
        P.close()
        ctx.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b, equal_nan=True)


def _leeway_props(P, n, seed):
    r = np.random.default_rng(seed)
    ori = (np.arange(n) % 2).astype(np.float32)
    for slot, val in enumerate([np.full(n, 0.96), np.where(ori == 0, 0.54, -0.54), np.zeros(n), np.zeros(n),
                                np.abs(r.standard_normal(n)) * 12.0, r.standard_normal(n) * 9.4, np.full(n, 0.04), ori,
                                (r.uniform(size=n) < 0.2).astype(np.float64)]):
        P.set_property(slot, val.astype(np.float32))


@pytest.mark.parametrize('layout', ['stere', 'latlon', 'constant_wind'])
def test_leeway_step_in_one_launch_equals_the_separate_calls(ctx, layout):
    """odr_env_coast_leeway (k_step_leeway: sample + current / wind uncertainty + coastline + Leeway.update in one launch)
    == env_sample, env_add_noise x 2, coastline, compact, leeway, bit for bit: positions, status, the sampled (perturbed)
    environment, the jibed crosswind slopes -- on a polar-stereographic reader (C5), on a lat / lon reader, and through the
    split lane when the wind comes from another reader than the current."""
    rng = np.random.default_rng(21)
    n = 5000
    if layout == 'stere':
        g = synth.grid_stere(nx=120, ny=90, nt=3, seed=3)
        names = [U, V, XW, YW, LAND]
        levels = [(float(g['t'][k]), {nm: g[nm][k] for nm in names}) for k in range(3)]
        sc = Scenario([('grid', dict(x=g['x'], y=g['y'], levels=levels, proj=synth.NORKYST_PROJ))], fallbacks={U: 0.0, V: 0.0, XW: 0.0, YW: 0.0})
        sc.device(ctx)
        from opendrift_amd.projection import stere_polar_inverse
        x = rng.uniform(g['x'][4], g['x'][-5], n)
        y = rng.uniform(g['y'][4], g['y'][-5], n)
        lon, lat = stere_polar_inverse(x, y, **synth.NORKYST_PROJ)
    else:
        g = synth.grid3d(nx=96, ny=80, nz=1, nt=3, seed=5) if False else synth.grid_stere(nx=96, ny=80, nt=3, seed=4)
        # a lat / lon grid with the stere generator's fields: same arrays, geographic axes
        gx, gy = np.linspace(3.0, 9.0, 96), np.linspace(59.0, 63.0, 80)
        names = [U, V, LAND] if layout == 'constant_wind' else [U, V, XW, YW, LAND]
        levels = [(float(g['t'][k]), {nm: g[nm][k] for nm in names}) for k in range(3)]
        srcs = [('grid', dict(x=gx, y=gy, levels=levels))]
        if layout == 'constant_wind':
            srcs.append(('constant', {XW: 7.0, YW: -3.0}))
        sc = Scenario(srcs, fallbacks={U: 0.0, V: 0.0, XW: 0.0, YW: 0.0})
        sc.device(ctx)
        lon, lat = rng.uniform(gx[3], gx[-4], n), rng.uniform(gy[3], gy[-4], n)
    variables = [XW, YW, U, V, LAND]
    P, Q = ctx.particles(n), ctx.particles(n)
    for X in (P, Q):
        X.append(lon, lat)
        _leeway_props(X, n, 9)
    dt = 600.0
    hits = 0
    for k in range(5):
        t = 300.0 + 600.0 * k
        P.env_sample(variables, t)
        P.env_add_noise(U, V, 0.1, step=k)
        P.env_add_noise(XW, YW, 2.0, step=k)
        n1 = P.coastline('stranding', stranded_code=1)
        P.compact()
        P.leeway(dt, 0.4, step=k)
        n2 = Q.env_coast_leeway(variables, t, dt, 0.4, coastline='stranding', stranded_code=1, current_uncertainty=0.1,
                                wind_uncertainty=2.0, step=k)
        Q.compact()
        hits += n1
        assert n1 == n2 and len(P) == len(Q), (k, n1, n2, len(P), len(Q))
        _same(P.download(), Q.download(), 'step %d' % k)
        for v in variables:
            a, b = P.env_download(v), Q.env_download(v)
            assert ((a == b) | (np.isnan(a) & np.isnan(b))).all(), (k, v)
        for slot in (1, 7):     # crosswind slope and orientation: the jibes
            assert np.array_equal(P.get_property(slot), Q.get_property(slot)), (k, slot)
        da, db = P.download_deactivated(), Q.download_deactivated()
        for q in ('lon', 'lat', 'status', 'ID'):
            assert (da[q] == db[q]).all(), (k, 'deactivated', q)
    assert hits > 0 or layout != 'stere'        # the stere scenario has a coast
