"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs, and against golden vectors produced by the reference itself (tests/golden/*.npz).

Tolerances (stated, north star: trajectories within 1e-6 deg of the CPU reference):
  * float32 environment values from gridded blocks: bit-exact vs the oracle (scipy-exact bilinear);
  * positions vs the oracle: <= 1e-10 deg per step (float64 geodesic, FMA contraction differences);
  * positions vs the reference golden vectors: <= 1e-7 deg over the stored windows -- NumPy's
    float32 arctan2 on the generating host is 1 ulp off the correctly rounded value in ~38 % of the
    cases (oracle/step.c azimuth_f32), which moves a particle by <= |step| * 2.4e-7.
"""
import numpy as np
import pytest

from conftest import golden
from scenarios import Scenario
from opendrift_amd import synthetic as synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

U, V = 'x_sea_water_velocity', 'y_sea_water_velocity'


def _maxerr(a, b):
    return float(np.nanmax(np.abs(np.asarray(a) - np.asarray(b)))) if np.size(a) else 0.0


# ------------------------------------------------------------------ a13: update_positions / geodesic
def test_update_positions_matches_oracle_geodesic(ctx):
    rng = np.random.default_rng(1)
    n = 20000
    lon = rng.uniform(-180, 180, n)
    lat = rng.uniform(-89, 89, n)
    speed = 10 ** rng.uniform(-4, 2.5, n)
    ang = rng.uniform(-np.pi, np.pi, n)
    u, v = speed * np.sin(ang), speed * np.cos(ang)
    moving = (rng.uniform(size=n) > 0.05).astype(np.int32)
    for dtype, dt in ((np.float64, 3600.0), (np.float32, 900.0), (np.float64, -3600.0)):
        P = ctx.particles(n)
        P.append(lon, lat, moving=moving)
        uu, vv = u.astype(dtype), v.astype(dtype)
        P.update_positions(uu, vv, dt)
        got = P.download()
        lo, la = lon.copy(), lat.copy()
        orc.update_positions(lo, la, uu, vv, moving, dt)
        dlon = np.abs(got['lon'] - lo)
        dlon = np.minimum(dlon, 360 - dlon)
        assert dlon.max() < 1e-11 and _maxerr(got['lat'], la) < 1e-11
        P.close()


# ------------------------------------------------------------------ C1 / C2 golden (reference itself)
def test_c1_constant_euler_golden(ctx):
    g = golden('c1_constant_euler.npz')
    sc = Scenario([('constant', {U: 0.3, V: 0.2})], fallbacks={'land_binary_mask': 0.0})
    sc.device(ctx)
    w = sc.oracle_world()
    lon0, lat0 = g['lon'][0], g['lat'][0]
    n = lon0.size
    P = ctx.particles(n)
    P.append(lon0, lat0)
    lo, la, z = lon0.copy(), lat0.copy(), np.zeros(n)
    worst_ref = worst_orc = 0.0
    for k in range(g['lon'].shape[0] - 1):
        t = k * 3600.0
        P.env_sample([U, V], t)
        P.advect('euler', t, 3600.0)
        got = P.download()
        ue, ve = orc.get_environment(w, [0, 1], lo, la, z, t)
        orc.advect_ocean_current(w, 0, lo, la, z, np.ones(n, np.int32), np.ones(n, np.float32), ue, ve, t, 3600.0)
        worst_orc = max(worst_orc, _maxerr(got['lon'], lo), _maxerr(got['lat'], la))
        worst_ref = max(worst_ref, _maxerr(got['lon'], g['lon'][k + 1]), _maxerr(got['lat'], g['lat'][k + 1]))
    assert worst_orc < 1e-10, worst_orc
    assert worst_ref < 1e-7, worst_ref


@pytest.mark.parametrize('name,scheme', [('euler', 'euler'), ('rungekutta', 'runge-kutta'),
                                         ('rungekutta4', 'runge-kutta4')])
def test_c2_double_gyre_golden(ctx, name, scheme):
    g = golden('c2_double_gyre_%s.npz' % name)
    prm = dict(A=float(g['A']), epsilon=float(g['epsilon']), omega=float(g['omega']), t0=0.0)
    sc = Scenario([('double_gyre', prm)], fallbacks={'land_binary_mask': 0.0})
    sc.device(ctx)
    w = sc.oracle_world()
    n = g['lon'].shape[1]
    dt = float(g['dt'])
    P = ctx.particles(n)
    P.append(g['lon'][0], g['lat'][0])
    lo, la, z = g['lon'][0].copy(), g['lat'][0].copy(), np.zeros(n)
    isch = {'euler': 0, 'runge-kutta': 1, 'runge-kutta4': 2}[scheme]
    worst_ref = worst_orc = 0.0
    for k in range(g['lon'].shape[0] - 1):
        t = k * dt
        P.env_sample([U, V], t)
        P.advect(scheme, t, dt)
        got = P.download()
        ue, ve = orc.get_environment(w, [0, 1], lo, la, z, t)
        orc.advect_ocean_current(w, isch, lo, la, z, np.ones(n, np.int32), np.ones(n, np.float32), ue, ve, t, dt)
        worst_orc = max(worst_orc, _maxerr(got['lon'], lo), _maxerr(got['lat'], la))
        worst_ref = max(worst_ref, _maxerr(got['lon'], g['lon'][k + 1]), _maxerr(got['lat'], g['lat'][k + 1]))
    # the whole gyre spans 1.8e-5 deg: also report in reader metres
    p = orc.make_proj(orc.PROJ_STERE_EQUIT_SPHERE, a=6.371e6, es=0.0)
    x, y = orc.proj_fwd(p, got['lon'], got['lat'])
    xr, yr = orc.proj_fwd(p, g['lon'][-1], g['lat'][-1])
    print('c2 %s: vs oracle %.2e deg, vs reference %.2e deg = %.2e m' %
          (scheme, worst_orc, worst_ref, max(_maxerr(x, xr), _maxerr(y, yr))))
    assert worst_orc < 1e-9, worst_orc   # chaotic flow amplifies 1e-16 differences
    assert worst_ref < 1e-9, worst_ref
    assert max(_maxerr(x, xr), _maxerr(y, yr)) < 1e-3


# ------------------------------------------------------------------ a5-a9: gridded blocks
def _grid3d_scenario(g, nlev=3):
    names = [U, V, 'upward_sea_water_velocity', 'ocean_vertical_diffusivity',
             'sea_floor_depth_below_sea_level', 'land_binary_mask']
    levels = [(float(g['t'][k]), {n: g[n][k] for n in names}) for k in range(nlev)]
    return Scenario([('grid', dict(x=g['x'], y=g['y'], z=g['z'], levels=levels))],
                    fallbacks={U: 0.0, V: 0.0, 'upward_sea_water_velocity': 0.0,
                               'ocean_vertical_diffusivity': 0.0, 'sea_floor_depth_below_sea_level': 10000.0})


def test_grid3d_environment_bit_exact(ctx):
    g = synth.grid3d(nx=96, ny=80, nz=8, nt=3, seed=5)
    sc = _grid3d_scenario(g)
    sc.device(ctx)
    w = sc.oracle_world()
    rng = np.random.default_rng(7)
    n = 30000
    lon = rng.uniform(g['x'][0] - 0.05, g['x'][-1] + 0.05, n)   # a few outside the domain
    lat = rng.uniform(g['y'][0] - 0.05, g['y'][-1] + 0.05, n)
    z = -rng.uniform(0, 120, n)
    lon[:50] = g['x'][rng.integers(0, 96, 50)]                   # exactly on grid nodes / edges
    lat[:50] = g['y'][rng.integers(0, 80, 50)]
    lon[50:60], lat[50:60] = g['x'][-1], g['y'][-1]
    names = [U, V, 'upward_sea_water_velocity', 'ocean_vertical_diffusivity',
             'sea_floor_depth_below_sea_level', 'land_binary_mask']
    P = ctx.particles(n)
    P.append(lon, lat, z=z)
    for t in (0.0, 1234.5, 3600.0, 5000.0):
        got = P.env_sample(names, t, download=True)
        ref = orc.get_environment(w, [orc.VAR[k] for k in names], lon, lat, z, t)
        for k, r in zip(names, ref):
            a = got[k]
            same = (a == r) | (np.isnan(a) & np.isnan(r))
            assert same.all(), (k, t, int((~same).sum()), a[~same][:4], r[~same][:4])
        w = sc.oracle_world()   # the oracle's blocks are mutated by the NaN dilation: rebuild


def test_grid_stere_environment(ctx):
    g = synth.grid_stere(nx=120, ny=90, nt=3, seed=9)
    names = [U, V, 'x_wind', 'y_wind', 'sea_surface_wave_stokes_drift_x_velocity',
             'sea_surface_wave_stokes_drift_y_velocity', 'land_binary_mask']
    levels = [(float(g['t'][k]), {n: g[n][k] for n in names}) for k in range(3)]
    sc = Scenario([('grid', dict(x=g['x'], y=g['y'], proj=synth.NORKYST_PROJ, levels=levels))],
                  fallbacks={U: 0.0, V: 0.0})
    sc.device(ctx)
    w = sc.oracle_world()
    rng = np.random.default_rng(11)
    n = 20000
    p = orc.make_proj(orc.PROJ_STERE_POLAR, a=6371000.0, es=(2 - 1 / 298.257223563) / 298.257223563,
                      lat0=90.0, lon0=70.0, lat_ts=60.0)
    lon, lat = orc.proj_inv(p, rng.uniform(g['x'][0] - 500, g['x'][-1] + 500, n),
                            rng.uniform(g['y'][0] - 500, g['y'][-1] + 500, n))
    P = ctx.particles(n)
    P.append(lon, lat)
    for t in (0.0, 900.0, 3600.0):
        got = P.env_sample(names, t, download=True)
        ref = orc.get_environment(w, [orc.VAR[k] for k in names], lon, lat, np.zeros(n), t)
        for k, r in zip(names, ref):
            a = got[k]
            assert (np.isnan(a) == np.isnan(r)).all(), k
            assert _maxerr(a, r) < 2e-6, (k, _maxerr(a, r))   # rotated in f64, rounded to f32: <= 1 ulp of ~10 m/s
        w = sc.oracle_world()


def test_rk4_on_grid3d_matches_oracle(ctx):
    g = synth.grid3d(nx=96, ny=80, nz=8, nt=3, seed=5)
    sc = _grid3d_scenario(g)
    sc.device(ctx)
    rng = np.random.default_rng(13)
    n = 20000
    lon = rng.uniform(g['x'][3], g['x'][-4], n)
    lat = rng.uniform(g['y'][3], g['y'][-4], n)
    z = -rng.uniform(0, 60, n)
    cdf = rng.uniform(0.8, 1.0, n).astype(np.float32)
    P = ctx.particles(n)
    P.append(lon, lat, z=z, current_drift_factor=cdf)
    lo, la = lon.copy(), lat.copy()
    mv = np.ones(n, np.int32)
    for k, scheme in enumerate(('euler', 'runge-kutta', 'runge-kutta4', 'runge-kutta4')):
        t = 600.0 * k + 2700
        w = sc.oracle_world()
        P.env_sample([U, V], t)
        P.advect(scheme, t, 600.0)
        ue, ve = orc.get_environment(w, [0, 1], lo, la, z, t)
        orc.advect_ocean_current(w, {'euler': 0, 'runge-kutta': 1, 'runge-kutta4': 2}[scheme], lo, la, z, mv,
                                 cdf, ue, ve, t, 600.0)
        got = P.download()
        assert _maxerr(got['lon'], lo) < 1e-10 and _maxerr(got['lat'], la) < 1e-10, (scheme, _maxerr(got['lon'], lo))


# ------------------------------------------------------------------ a12 / a14 / a15 / a16
def test_wind_stokes_hdiff_match_oracle(ctx):
    rng = np.random.default_rng(17)
    n = 20000
    lon = rng.uniform(3, 6, n)
    lat = rng.uniform(59, 62, n)
    z = np.where(rng.uniform(size=n) < 0.5, 0.0, -rng.uniform(0, 0.3, n))
    wdf = rng.uniform(0, 0.04, n).astype(np.float32)
    env = {
        'x_wind': rng.normal(5, 3, n).astype(np.float32), 'y_wind': rng.normal(-2, 3, n).astype(np.float32),
        U: rng.normal(0, 0.3, n).astype(np.float32), V: rng.normal(0, 0.3, n).astype(np.float32),
        'sea_surface_wave_stokes_drift_x_velocity': rng.normal(0.05, 0.03, n).astype(np.float32),
        'sea_surface_wave_stokes_drift_y_velocity': rng.normal(0.0, 0.03, n).astype(np.float32),
        'sea_surface_wave_significant_height': rng.uniform(0.5, 3, n).astype(np.float32),
        'sea_surface_wave_period_at_variance_spectral_density_maximum': rng.uniform(4, 12, n).astype(np.float32),
        'horizontal_diffusivity': rng.uniform(0, 20, n).astype(np.float32),
    }
    mv = np.ones(n, np.int32)
    P = ctx.particles(n)
    P.append(lon, lat, z=z, wind_drift_factor=wdf)
    for k, v in env.items():
        P.env_upload(k, v)
    lo, la = lon.copy(), lat.copy()
    # wind, relative and absolute
    for rel in (0, 1):
        P.advect_wind(900.0, wind_drift_depth=0.1, relative_wind=bool(rel))
        orc.advect_wind(lo, la, z, mv, wdf, env['x_wind'], env['y_wind'], env[U], env[V], 0.1, rel, 1.0, 900.0)
        got = P.download()
        assert _maxerr(got['lon'], lo) < 1e-11 and _maxerr(got['lat'], la) < 1e-11, ('wind', rel)
    # Stokes drift: the three Breivik profiles x the Hs/Tp provenance modes
    for profile in (0, 1, 2):
        for hs_mode, tp_mode in ((0, 0), (1, 1), (2, 2), (0, 1)):
            P.stokes_drift(900.0, profile=profile, hs_mode=hs_mode, tp_mode=tp_mode)
            orc.stokes_drift(lo, la, z, mv, env['sea_surface_wave_stokes_drift_x_velocity'],
                             env['sea_surface_wave_stokes_drift_y_velocity'],
                             env['sea_surface_wave_significant_height'],
                             env['sea_surface_wave_period_at_variance_spectral_density_maximum'],
                             env['x_wind'], env['y_wind'], hs_mode, tp_mode, profile, 1.0, 900.0)
            got = P.download()
            assert _maxerr(got['lon'], lo) < 1e-10 and _maxerr(got['lat'], la) < 1e-10, ('stokes', profile, hs_mode)
    # horizontal diffusion with np.random normals handed over (parity mode)
    np.random.seed(0)
    nx, ny = np.random.normal(scale=1, size=n), np.random.normal(scale=1, size=n)
    P.hdiffusion(900.0, normals=(nx, ny))
    orc.horizontal_diffusion(lo, la, mv, env['horizontal_diffusivity'], nx, ny, 900.0)
    got = P.download()
    assert _maxerr(got['lon'], lo) < 1e-11 and _maxerr(got['lat'], la) < 1e-11
    # device Philox mode: statistics of the displacement
    before = P.download()
    P.hdiffusion(900.0, step=3)
    after = P.download()
    dy = (after['lat'] - before['lat']) * 111e3
    sig = np.sqrt(2 * env['horizontal_diffusivity'] * 900.0)
    zscore = dy[sig > 1] / sig[sig > 1]
    assert abs(zscore.mean()) < 0.03 and abs(zscore.std() - 1) < 0.03


def test_early_outs_leave_positions_untouched(ctx):
    n = 1000
    lon = np.linspace(185, 190, n)     # outside [-180,180]: an executed geodesic would renormalise it
    P = ctx.particles(n)
    P.append(lon, np.full(n, 60.0))
    for k in ('x_wind', 'y_wind', U, V, 'sea_surface_wave_stokes_drift_x_velocity',
              'sea_surface_wave_stokes_drift_y_velocity', 'horizontal_diffusivity'):
        P.env_upload(k, np.zeros(n, np.float32))
    P.advect_wind(900.0)
    P.stokes_drift(900.0, hs_mode=2, tp_mode=2)
    P.hdiffusion(900.0)
    assert (P.download()['lon'] == lon).all()


def test_vertical_mixing_matches_oracle(ctx):
    g = synth.grid3d(nx=64, ny=48, nz=8, nt=3, seed=21, coast=False)
    sc = _grid3d_scenario(g)
    sc.device(ctx)
    w = sc.oracle_world()
    rng = np.random.default_rng(23)
    n = 5000
    lon = rng.uniform(g['x'][2], g['x'][-3], n)
    lat = rng.uniform(g['y'][2], g['y'][-3], n)
    z0 = -rng.uniform(0, 90, n)
    z0[:200] = 0.0
    tv = rng.normal(0, 0.002, n).astype(np.float32)
    t, dt, dt_mix = 1800.0, 600.0, 60.0
    names = ['sea_floor_depth_below_sea_level', 'sea_surface_height', 'upward_sea_water_velocity']
    for mix_at_surface in (False, True):
        P = ctx.particles(n)
        P.append(lon, lat, z=z0, terminal_velocity=tv)
        env = P.env_sample(names, t, download=True)
        uni = np.random.default_rng(3).uniform(size=(10, n))
        P.vmix(t, dt, dt_mix, mix_at_surface=mix_at_surface, uniforms=uni)
        P.vertical_advection(dt)
        got = P.download()
        Kp = orc.get_profile(w, orc.VAR['ocean_vertical_diffusivity'], lon, lat, t, len(g['z']))
        zz = z0.copy()
        orc.vertical_mixing(zz, np.ones(n, np.int32), tv, env[names[0]], env[names[1]], g['z'], Kp, dt, dt_mix,
                            int(mix_at_surface), uni)
        orc.vertical_advection(zz, np.ones(n, np.int32), env[names[2]], dt)
        assert _maxerr(got['z'], zz) < 1e-9, _maxerr(got['z'], zz)
        P.close()
    # Philox mode: reproducible and identical regardless of particle order (counter = ID)
    P = ctx.particles(n)
    P.append(lon, lat, z=z0, terminal_velocity=tv)
    P.env_sample(names, t)
    P.vmix(t, dt, dt_mix, step=5)
    a = P.download()['z']
    perm = rng.permutation(n)
    Q = ctx.particles(n)
    Q.append(lon[perm], lat[perm], z=z0[perm], terminal_velocity=tv[perm], id=perm.astype(np.int32))
    Q.env_sample(names, t)
    Q.vmix(t, dt, dt_mix, step=5)
    b = Q.download()['z']
    assert (a[perm] == b).all()
    assert a.min() >= -env['sea_floor_depth_below_sea_level'].max() - 1 and a.max() <= 0


def test_coastline_and_compaction(ctx):
    rng = np.random.default_rng(29)
    n = 100000
    lon, lat = rng.uniform(0, 10, n), rng.uniform(60, 66, n)
    z = np.where(rng.uniform(size=n) < 0.9, 0.0, 1.0)
    land = (rng.uniform(size=n) < 0.3).astype(np.float32)
    P = ctx.particles(n)
    P.append(lon, lat, z=z)
    P.env_upload('land_binary_mask', land)
    P.env_upload(U, np.arange(n, dtype=np.float32))
    hit = P.coastline('stranding', stranded_code=2)
    assert hit == int(land.sum())
    st = np.zeros(n, np.int32)
    mv = np.ones(n, np.int32)
    orc.coastline(1, land, lon.copy(), lat.copy(), z, lon, lat, st, mv, 2)
    got = P.download()
    assert (got['status'] == st).all() and (got['moving'] == mv).all()
    kept = P.compact()
    assert kept == int((st == 0).sum())
    got = P.download()
    dead = P.download_deactivated()
    # in-place compaction: holes are filled from the tail, so the survivors are a permutation
    # (elements are identified by ID); the deactivated store keeps the index order
    assert (np.sort(got['ID']) == np.nonzero(st == 0)[0]).all()
    assert (dead['ID'] == np.nonzero(st != 0)[0]).all()
    assert (got['lon'] == lon[got['ID']]).all() and (got['lat'] == lat[got['ID']]).all()
    assert (dead['lat'] == lat[st != 0]).all()
    assert (P.env_download(U) == got['ID'].astype(np.float32)).all()   # environment follows its element
    stay = np.nonzero(st[:kept] == 0)[0]
    assert (got['ID'][stay] == stay).all()                          # only the holes were refilled
    assert P.compact() == kept                                     # nothing more to remove
    # 'previous': back to the position of the last environment sample
    Q = ctx.particles(n)
    Q.append(lon, lat)
    Q.env_upload('land_binary_mask', land)
    Q.store_previous()
    Q.update_positions(np.full(n, 0.5), np.full(n, 0.5), 600.0)
    Q.coastline('previous')
    g2 = Q.download()
    assert (g2['lon'][land == 1] == lon[land == 1]).all() and (g2['lon'][land == 0] != lon[land == 0]).all()


def test_sort_by_cell_is_layout_only(ctx):
    """odr_sort_particles re-orders memory only: per-ID state and per-ID results are unchanged."""
    g = synth.grid3d(nx=96, ny=80, nz=8, nt=3, seed=5)
    sc = _grid3d_scenario(g)
    sc.device(ctx)
    rng = np.random.default_rng(31)
    n = 50000
    lon = rng.uniform(g['x'][0] - 0.02, g['x'][-1] + 0.02, n)
    lat = rng.uniform(g['y'][0] - 0.02, g['y'][-1] + 0.02, n)
    z = -rng.uniform(0, 60, n)
    tv = rng.normal(0, 0.002, n).astype(np.float32)
    names = [U, V, 'upward_sea_water_velocity', 'sea_floor_depth_below_sea_level', 'sea_surface_height']

    def run(sort):
        P = ctx.particles(n)
        P.append(lon, lat, z=z, terminal_velocity=tv)
        if sort:
            P.sort_by_cell(0)
            d = P.download()
            ix = np.clip(np.floor((d['lon'] - g['x'][0]) / (g['x'][-1] - g['x'][0]) * 95), 0, 95).astype(int)
            assert sorted(d['ID'].tolist()) == list(range(n))
            assert (d['lon'] == lon[d['ID']]).all() and (d['z'] == z[d['ID']]).all()
        for k in range(3):
            t = 600.0 * k
            P.env_sample(names, t)
            P.advect('runge-kutta4', t, 600.0)
            P.vmix(t, 600.0, 60.0, step=k)
            P.vertical_advection(600.0)
            if sort and k == 1:
                P.sort_by_cell(0)
        d = P.download()
        o = np.argsort(d['ID'])
        P.close()
        return d['lon'][o], d['lat'][o], d['z'][o]

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert (x == y).all()


# ------------------------------------------------------------------ C3 / C4 golden (reference itself)
def _states_close(a, b, tol_pos, tol_z):
    for (lo1, la1, z1, s1), (lo2, la2, z2, s2) in zip(a, b):
        assert (s1 == s2).all()
        assert _maxerr(lo1, lo2) < tol_pos and _maxerr(la1, la2) < tol_pos and _maxerr(z1, z2) < tol_z, \
            (_maxerr(lo1, lo2), _maxerr(la1, la2), _maxerr(z1, z2))


def test_c3_golden_device(ctx):
    """RK4 + 3D interpolation + vertical mixing + vertical advection + coastline 'previous' +
    seeded_on_land deactivation + compaction: device vs the reference's golden vectors and vs the oracle."""
    import replay
    g = golden('c3_grid3d_rk4_vmix.npz')
    nst = g['lon'].shape[0] - 1
    D = replay.DeviceBackend(replay.scenario_c3(g), ctx, g['lon'][0], g['lat'][0], g['z'][0])
    dev = replay.replay_c3(D, g, nst)
    worst = replay.compare(dev, g, tol_pos=1e-7, tol_z=1e-5)
    O = replay.OracleBackend(replay.scenario_c3(g), g['lon'][0], g['lat'][0], g['z'][0])
    _states_close(dev, replay.replay_c3(O, g, nst), 1e-10, 1e-8)
    print('c3 device vs reference:', worst)


def test_c4_golden_device(ctx):
    """Polar-stereographic reader (projection + vector rotation), RK4 + wind + Stokes + horizontal
    diffusion + stranding + compaction + RK stages beyond the reader's time coverage."""
    import replay
    g = golden('c4_stere_rk4_hdiff_strand.npz')
    D = replay.DeviceBackend(replay.scenario_c4(g), ctx, g['lon'][0], g['lat'][0], g['z'][0], wdf=float(g['wdf']))
    dev = replay.replay_c4(D, g, 10)    # the 10th step: everything 'missing_data' (reader time coverage ended)
    worst = replay.compare(dev, g, tol_pos=1e-7)
    O = replay.OracleBackend(replay.scenario_c4(g), g['lon'][0], g['lat'][0], g['z'][0], wdf=float(g['wdf']))
    _states_close(dev, replay.replay_c4(O, g, 10), 2e-9, 1e-12)
    print('c4 device vs reference:', worst)


@pytest.mark.parametrize('tag', ['lcc_sphere', 'lcc_wgs84', 'merc_wgs84'])
def test_c20_lambert_and_mercator_device_vs_oracle(ctx, tag):
    """Readers on Lambert conformal conic / Mercator grids (PROJ_LCC / PROJ_MERC in proj_fwd, proj_inv and the vector
    rotation): the device against the reference's own runs (1e-7 deg, the reference's float32 longitude
    modulation of a run's first get_environment included: odr_ctx_set_position_class) and against the oracle at the tolerance of the polar-stereographic case (2e-9 deg)."""
    import replay
    g = golden('c20_lcc_merc_rk4.npz')
    sub = {k: g['%s_%s' % (tag, k)] for k in ('lon', 'lat', 'z', 'status')}
    nst = sub['lon'].shape[0] - 1
    D = replay.DeviceBackend(replay.scenario_c20(g, tag), ctx, sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=float(g['wdf']))
    dev = replay.replay_c20(D, g, tag, nst)
    worst = replay.compare(dev, sub, tol_pos=1e-7)
    O = replay.OracleBackend(replay.scenario_c20(g, tag), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=float(g['wdf']))
    _states_close(dev, replay.replay_c20(O, g, tag, nst), 2e-9, 1e-12)
    print('c20', tag, 'device vs reference:', worst)


@pytest.mark.parametrize('tag', ['utm33', 'laea_grs80', 'stere_oblique', 'rotated_pole'])
def test_c23_round5_projections_device_vs_oracle(ctx, tag):
    """Readers on a UTM (transverse Mercator), an ETRS89-LAEA, an oblique stereographic and a rotated-pole grid (PROJ_TMERC /
    PROJ_LAEA / PROJ_STERE_OBLIQUE / PROJ_OB_TRAN in proj_fwd, proj_inv and the vector rotation -- for the rotated pole the
    geodesic-inverse azimuth of the reference's 0.1-degree line): the device against the reference's own runs
    (oracle/gen_golden_proj2.py, 1e-7 deg) and against the oracle at the tolerance of the polar-stereographic case (2e-9 deg)."""
    import replay
    g = golden('c23_proj_rk4.npz')
    sub = {k: g['%s_%s' % (tag, k)] for k in ('lon', 'lat', 'z', 'status')}
    nst = sub['lon'].shape[0] - 1
    D = replay.DeviceBackend(replay.scenario_c23(g, tag), ctx, sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=float(g['wdf']))
    dev = replay.replay_c20(D, g, tag, nst)
    worst = replay.compare(dev, sub, tol_pos=1e-7)
    O = replay.OracleBackend(replay.scenario_c23(g, tag), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=float(g['wdf']))
    _states_close(dev, replay.replay_c20(O, g, tag, nst), 2e-9, 1e-12)
    print('c23', tag, 'device vs reference:', worst)


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_c24_profile_paths_device_vs_oracle(ctx, tag):
    """(a) truncation of the ocean model at 20 m together with mixing on reader diffusivity profiles (odr_particles_truncate_z
    around the sampling calls, the K columns whole); (b) an ensemble diffusivity: the element mixes on the column of the member
    of its main-loop sample (k_kmember -> k_vmix) -- against the reference's own runs and the oracle."""
    import replay
    g = golden('c24_profiles.npz')
    sub = {k: g['%s_%s' % (tag, k)] for k in ('lon', 'lat', 'z', 'status')}
    nst = sub['lon'].shape[0] - 1
    trunc = float(g['truncate']) if tag == 'a' else None
    D = replay.DeviceBackend(replay.scenario_c24(g, tag), ctx, sub['lon'][0], sub['lat'][0], sub['z'][0])
    dev = replay.replay_c24(D, g, tag, nst, truncate=trunc)
    worst = replay.compare(dev, sub, tol_pos=1e-7, tol_z=1e-5)
    O = replay.OracleBackend(replay.scenario_c24(g, tag), sub['lon'][0], sub['lat'][0], sub['z'][0])
    _states_close(dev, replay.replay_c24(O, g, tag, nst, truncate=trunc), 2e-9, 1e-9)
    print('c24' + tag, 'device vs reference:', worst)


def test_c24c_truncation_on_a_reader_that_cuts_its_block_device_vs_oracle(ctx):
    """odr_vmix_set_profile_levels: the K columns end where a reader that hands out the levels asked for cut its block (golden c24c:
    the reference's own run on such a reader) -- the generic mixing kernel's level search, clamp and np.gradient edge on the cut grid."""
    import replay
    g = golden('c24_profiles.npz')
    sub = {k: g['c_%s' % k] for k in ('lon', 'lat', 'z', 'status')}
    nst = sub['lon'].shape[0] - 1
    trunc = float(g['truncate'])
    D = replay.DeviceBackend(replay.scenario_c24(g, 'a'), ctx, sub['lon'][0], sub['lat'][0], sub['z'][0])
    dev = replay.replay_c24(D, g, 'c', nst, truncate=trunc, gtag='a', cut_levels=True)
    worst = replay.compare(dev, sub, tol_pos=1e-7, tol_z=1e-5)
    O = replay.OracleBackend(replay.scenario_c24(g, 'a'), sub['lon'][0], sub['lat'][0], sub['z'][0])
    _states_close(dev, replay.replay_c24(O, g, 'c', nst, truncate=trunc, gtag='a', cut_levels=True), 2e-9, 1e-9)
    print('c24c device vs reference:', worst)


def test_c5_leeway_golden_device(ctx):
    """Leeway.update kernel + environment uncertainty (host-drawn normals) + jibing vs the reference's Leeway."""
    import replay
    g = golden('c5_leeway_stere.npz')
    nst = g['lon'].shape[0] - 1
    D = replay.DeviceBackend(replay.scenario_c5(g), ctx, g['lon'][0], g['lat'][0], g['z'][0])
    dev = replay.replay_c5(D, g, nst)
    worst = replay.compare(dev, g, tol_pos=1e-7)
    O = replay.OracleBackend(replay.scenario_c5(g), g['lon'][0], g['lat'][0], g['z'][0])
    _states_close(dev, replay.replay_c5(O, g, nst), 2e-9, 1e-12)
    cw = D.P.get_property(1)
    assert (np.sign(cw) != np.sign(g['p_crosswind_slope'][D.P.download()['ID']])).any()    # some elements jibed
    print('c5 device vs reference:', worst)


def test_c5b_leeway_capsizing_golden_device(ctx):
    """processes:capsizing on the device (k_capsize before k_leeway) vs the reference Leeway's own run."""
    import replay
    g = golden('c5b_leeway_capsizing.npz')
    nst = g['lon'].shape[0] - 1
    D = replay.DeviceBackend(replay.scenario_c5(g), ctx, g['lon'][0], g['lat'][0], g['z'][0])
    dev = replay.replay_c5(D, g, nst)
    worst = replay.compare(dev, g, tol_pos=1e-7)
    ids, cap = D.P.download()['ID'], D.P.get_property(8)
    ref = np.full(g['lon'].shape[1], -1.0)
    ref[g['ID_final']] = g['capsized_final']
    assert (cap == ref[ids]).all() and cap.sum() > 50        # the same elements capsized
    print('c5b device vs reference:', worst)


@pytest.mark.parametrize('action', ['deactivate', 'previous'])
def test_c8_seafloor_actions_device(ctx, action):
    """general:seafloor_action 'deactivate' / 'previous' (odr_seafloor_action): device vs the reference's runs and
    bit for bit vs the oracle."""
    import replay
    g = golden('c8_seafloor_actions.npz')
    sub = {k: g[action + '_' + k] for k in ('lon', 'lat', 'z', 'status')}
    D = replay.DeviceBackend(replay.scenario_c8(g), ctx, sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=0.0)
    dev = replay.replay_c8(D, g, sub, action, 8)
    replay.compare(dev, sub, tol_pos=1e-7, tol_z=2e-5)      # z: first-step float32 depth, DESIGN.md 2.1
    O = replay.OracleBackend(replay.scenario_c8(g), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=0.0)
    _states_close(dev, replay.replay_c8(O, g, sub, action, 8), 1e-10, 1e-12)


KZ, DEPTH, SSH = 'ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level', 'sea_surface_height'


@pytest.mark.parametrize('case', range(6))
def test_c11_reference_known_answers_of_an_isolated_mixing_step(ctx, case):
    """The reference's own known-answer test of one mixing time step (tests/models/test_run.py:359-410,
    test_vertical_mixing_profiles: hand-made diffusivity profiles on 15 levels, 120 sub-steps, mixing at the surface),
    executed on the reference itself for the golden (oracle/gen_golden.py:c11): the device kernel reproduces the final
    depths with the recorded np.random draws, and with them the published min / max / mean (one decimal)."""
    g = golden('c11_mixing_profiles.npz')
    vt, K, Kb, T, zmin, zmax, zmean = g['cases'][case]
    n = 100
    zl = g['z_levels'].astype(np.float64)
    x, y = np.array([3.0, 5.0]), np.array([59.0, 61.0])
    prof = g['K_%d' % case].astype(np.float32)
    arr = np.ascontiguousarray(np.broadcast_to(prof[:, None, None], (len(zl), 2, 2)))
    gs = ctx.add_grid(x, y, z=zl)
    ctx.upload_block(gs, 0, 0.0, {KZ: arr})
    ctx.bind(KZ, [gs], 0.0)
    cs = ctx.add_constant({DEPTH: 100.0, SSH: 0.0})
    ctx.bind(DEPTH, [cs], 10000.0)
    ctx.bind(SSH, [cs], 0.0)
    P = ctx.particles(n)
    P.append(np.full(n, 4.0), np.full(n, 60.0), z=np.full(n, -10.0), terminal_velocity=np.full(n, vt, np.float32))
    P.env_sample([DEPTH, SSH], 0.0)
    P.vmix(0.0, 7200.0, float(T), mix_at_surface=True, uniforms=g['uniforms_%d' % case])
    z = P.download()['z']
    # the test's hand-made profile is float64 (K = 0.01), field blocks on the device are float32 like the arrays file
    # readers hand out: 2e-8 relative in K, a 1e-7 m random walk over 120 sub-steps -- against the
    # reference 1e-6 m, against the oracle fed with the same float32 profile 1e-9 m
    assert np.abs(z - g['z_final_%d' % case]).max() < 1e-6
    zo = np.full(n, -10.0)
    orc.vertical_mixing(zo, np.ones(n, np.int32), np.full(n, vt, np.float32), np.full(n, 100, np.float32),
                        np.zeros(n, np.float32), zl, np.ascontiguousarray(np.tile(prof.astype(np.float64), (n, 1)).T),
                        7200.0, float(T), 1, g['uniforms_%d' % case])
    assert np.abs(z - zo).max() < 1e-9
    assert abs(z.min() - zmin) < 0.05 and abs(z.max() - zmax) < 0.05 and abs(z.mean() - zmean) < 0.05
    P.close()


@pytest.mark.parametrize('case', [2, 3, 5])
def test_c11_known_answers_through_the_model_api(case):
    """the same through OceanDrift.run() with the reference's configuration calls and np.random (seed 0)"""
    from datetime import datetime
    from opendrift_amd import readers
    from opendrift_amd.oceandrift import OceanDrift
    g = golden('c11_mixing_profiles.npz')
    vt, K, Kb, T, zmin, zmax, zmean = g['cases'][case]
    zl = g['z_levels'].astype(np.float64)
    t0 = datetime(2020, 1, 1)
    from datetime import timedelta
    prof = g['K_%d' % case].astype(np.float32)
    arr = np.ascontiguousarray(np.broadcast_to(prof[None, :, None, None], (2, len(zl), 2, 2)))
    o = OceanDrift(loglevel=50, seed=0, rng='numpy')
    o.add_reader(readers.GridReader(np.array([3.0, 5.0]), np.array([59.0, 61.0]), [t0, t0 + timedelta(days=1)],
                                    {KZ: arr}, z=zl))
    o.set_config('drift:vertical_mixing', True)
    o.set_config('drift:vertical_mixing_at_surface', True)
    o.set_config('drift:vertical_advection_at_surface', True)
    o.set_config('vertical_mixing:diffusivitymodel', 'environment')
    o.set_config('vertical_mixing:timestep', float(T))
    o.set_config('environment:fallback:land_binary_mask', 0)
    o.set_config('environment:fallback:sea_floor_depth_below_sea_level', 100)
    o.seed_elements(lon=4, lat=60, z=-10, time=t0, number=100, terminal_velocity=vt)
    np.random.seed(0)
    o.run(time_step=7200, steps=1)
    z = np.empty(100)
    z[o.elements.ID] = o.elements.z
    assert abs(z.min() - zmin) < 0.05 and abs(z.max() - zmax) < 0.05 and abs(z.mean() - zmean) < 0.05
    assert np.abs(z - g['z_final_%d' % case]).max() < 1e-6


def test_windsea_swell_stokes_profile_device_vs_oracle(ctx):
    """odr_stokes_drift with profile 3 = drift:stokes_drift_profile 'windsea_swell' (physics_methods.py:418-456, 831-841):
    swell part (monochromatic profile) + wind-sea part (Phillips profile) from six float32 environment variables, against
    the CPU oracle (itself pinned by the reference's own function, golden c18).  Tolerance 1e-9 deg: device and libm
    cosf / sinf may differ by an ulp, amplified where the two directions are nearly parallel."""
    from conftest import golden
    g = golden('c18_windsea_swell_profile.npz')
    n = len(g['sx'])
    rng = np.random.default_rng(3)
    lon, lat = rng.uniform(3, 6, n), rng.uniform(58, 62, n)
    names = {'sea_surface_wave_stokes_drift_x_velocity': g['sx'], 'sea_surface_wave_stokes_drift_y_velocity': g['sy'],
             'sea_surface_swell_wave_to_direction': g['swell_dir'],
             'sea_surface_swell_wave_peak_period_from_variance_spectral_density': g['swell_tp'],
             'sea_surface_swell_wave_significant_height': g['swell_hs'], 'sea_surface_wind_wave_to_direction': g['ww_dir'],
             'sea_surface_wind_wave_mean_period': g['ww_tm'], 'sea_surface_wind_wave_significant_height': g['ww_hs']}
    P = ctx.particles(n)
    P.append(lon, lat, z=g['z'])
    for k, v in names.items():
        P.env_upload(k, v.astype(np.float32))
    P.stokes_drift(900.0, profile=3)
    got = P.download()
    lo, la = lon.copy(), lat.copy()
    orc.stokes_drift_windsea_swell(lo, la, g['z'], np.ones(n, np.int32), g['sx'], g['sy'], g['swell_dir'], g['swell_tp'],
                                   g['swell_hs'], g['ww_dir'], g['ww_tm'], g['ww_hs'], 1.0, 900.0)
    o = np.argsort(got['ID'])
    cond = np.abs(np.sin(np.radians(g['swell_dir'].astype(float) - g['ww_dir'].astype(float))))
    d = np.maximum(np.abs(got['lon'][o] - lo), np.abs(got['lat'][o] - la))
    assert d.max() < 1e-9 and d[cond > 0.5].max() < 1e-10 and np.median(d) < 1e-13
    assert np.abs(lo - lon).max() > 1e-4          # it moved


@pytest.mark.parametrize('tag', ['merc_wgs84', 'lcc_wgs84', 'lcc_1sp', 'stere_north', 'stere_north_ts90', 'stere_south', 'stere_oblique',
                                 'stere_equatorial', 'laea_europe', 'laea_north', 'utm33', 'tmerc_wide', 'rotated_pole'])
def test_projection_known_answers_on_the_device(ctx, tag):
    """The device's forward projections (proj_fwd in csrc/odr_field.hip.h, reached through odr_source_lonlat2xy) against known
    answers computed independently of oracle and device -- mpmath, 40 digits, from each projection's definition
    (oracle/validate_projections.py; tests/test_proj_kat.py holds the oracle and the host mirror to the same file)."""
    import ast
    import test_proj_kat as K
    from opendrift_amd import projection
    g = np.load(K.GOLDEN)
    kw = ast.literal_eval(str(g[tag + '_kw']))
    lon, lat, x, y = (g['%s_%s' % (tag, k)] for k in ('lon', 'lat', 'x', 'y'))
    proj = projection.parse_proj4(K.proj4_of(tag, kw))
    pad = 1.0 if tag == 'rotated_pole' else 1e5
    gx, gy = np.linspace(x.min() - pad, x.max() + pad, 8), np.linspace(y.min() - pad, y.max() + pad, 6)
    sid = ctx.add_grid(gx, gy, proj=proj, lon_mode=1)
    dx, dy = ctx.lonlat2xy(sid, lon, lat)
    deg = tag == 'rotated_pole'
    err = max(np.abs(K._dlon(dx, x) if deg else dx - x).max(), np.abs(dy - y).max())
    print('device forward', tag, '%.2e' % err, 'deg' if deg else 'm')
    assert err < (1e-12 if deg else 1e-5 if tag == 'laea_north' else K.TOL_FWD)
