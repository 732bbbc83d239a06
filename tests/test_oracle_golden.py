"""CPU: the oracle (oracle/*.c) against golden vectors written by the REFERENCE ITSELF
(oracle/gen_golden.py runs OpenDrift v1.14.10's own update()/get_environment()/ReaderBlock code)
and against the un-vendored arithmetic it restates (SciPy directly, mpmath for the geodesic).

Position tolerance vs the reference: NumPy's float32 arctan2 on the generating host is 1 ulp off
the correctly rounded value in ~38 % of the calls (<= 2.4e-7 rad in azimuth), which displaces a
particle by <= 2.4e-7 * |step|; over the stored windows this stays below 1e-7 deg, two orders
inside the 1e-6 deg north-star tolerance.
"""
import numpy as np
import pytest

from conftest import golden
import replay
from oracle import oracle as orc


def test_geodesic_known_answers():
    """oracle/geodesic.c vs 34-digit mpmath evaluation of Karney's closed-form integrals
    (oracle/validate_geodesic.py wrote tests/golden/geodesic_kat.npz)."""
    rows = golden('geodesic_kat.npz')['rows']
    lo, la, a2 = orc.geod_fwd(rows[:, 1], rows[:, 0], rows[:, 2], rows[:, 3])
    dlon = lo - rows[:, 5]
    dlon -= 360 * np.round(dlon / 360)
    assert np.abs(la - rows[:, 4]).max() < 5e-14 and np.abs(dlon).max() < 5e-14


def test_geodesic_reference_test_values():
    """Coarse known answers of the reference's own tests: test_models.py:61-64 (1 m/s north-east
    components for 2 h) and test_environment.py:30-51 (1 m/s east for 1 h from 3E,60N -> 3.0645E)."""
    lo, la, _ = orc.geod_fwd(3.0, 60.0, 90.0, 3600.0)
    assert abs(lo[0] - 3.0645) < 5e-4 and abs(la[0] - 60.0) < 1e-3
    lo, la, _ = orc.geod_fwd(4.0, 60.0, 0.0, 7200.0)
    assert abs(la[0] - 60.0646) < 1e-3


def test_geodesic_inverse_roundtrip():
    rng = np.random.default_rng(0)
    n = 300
    lon, lat = rng.uniform(-170, 170, n), rng.uniform(-80, 80, n)
    az, s = rng.uniform(-180, 180, n), 10 ** rng.uniform(0, 4.5, n)
    lo, la, _ = orc.geod_fwd(lon, lat, az, s)
    az2, s2 = orc.geod_inv(lon, lat, lo, la)
    d = az2 - az
    d -= 360 * np.round(d / 360)
    assert np.abs(s2 - s).max() < 1e-6 and np.abs(d * s).max() < 1e-4   # metres / metre-degrees


def test_bilinear_matches_scipy_bitwise():
    from scipy.ndimage import map_coordinates
    rng = np.random.default_rng(1)
    a = rng.standard_normal((37, 53)).astype(np.float32)
    a[rng.uniform(size=a.shape) < 0.08] = np.nan
    n = 20000
    yi, xi = rng.uniform(-0.5, 36.5, n), rng.uniform(-0.5, 52.5, n)
    yi[:30], xi[:30] = np.round(yi[:30]).clip(0, 36), np.round(xi[:30]).clip(0, 52)   # exact nodes incl. edges
    ref = map_coordinates(a, [yi, xi], cval=np.nan, order=1)
    lib = orc.lib()
    import ctypes as C
    lib.orc_bilinear_f32.restype = C.c_float
    got = np.array([lib.orc_bilinear_f32(a.ctypes.data_as(C.POINTER(C.c_float)), 37, 53, C.c_double(y), C.c_double(x), 0)
                    for y, x in zip(yi[:3000], xi[:3000])], np.float32)
    r = ref[:3000]
    assert ((got == r) | (np.isnan(got) & np.isnan(r))).all()
    ref2 = map_coordinates(a, [yi, xi], cval=np.nan, order=1, mode='nearest')
    got2 = np.array([lib.orc_bilinear_f32(a.ctypes.data_as(C.POINTER(C.c_float)), 37, 53, C.c_double(y), C.c_double(x), 1)
                     for y, x in zip(yi[:3000], xi[:3000])], np.float32)
    r2 = ref2[:3000]
    assert ((got2 == r2) | (np.isnan(got2) & np.isnan(r2))).all()


def test_dilation_matches_scipy():
    from scipy.ndimage import grey_dilation
    rng = np.random.default_rng(2)
    a = rng.standard_normal((40, 30)).astype(np.float32)
    a[rng.uniform(size=a.shape) < 0.4] = np.nan
    a[:6, :5] = np.nan
    b = a.copy()
    # expand_numpy_array, interpolators.py:9-20
    mask = ~np.isfinite(b)
    minval = np.finfo(np.float32).min
    b[mask] = minval
    b[mask] = grey_dilation(b, size=3)[mask]
    b[b == minval] = np.nan
    orc.dilate_nan_once(a)
    assert ((a == b) | (np.isnan(a) & np.isnan(b))).all()


def test_ten_fold_predilation_equals_on_demand_dilation():
    """sampling a block dilated 10x up front with clamped coordinates gives the values
    of Linear2DInterpolator's stateful dilate-and-retry loop (interpolators.py:127-137)."""
    rng = np.random.default_rng(3)
    a = rng.standard_normal((60, 80)).astype(np.float32)
    a[:, 60:] = np.nan
    a[20:28, 30:36] = np.nan
    n = 5000
    yi, xi = rng.uniform(-0.3, 59.3, n), rng.uniform(-0.3, 72.0, n)   # up to 12 cells inland
    stateful = orc.linear2d_call(a.copy(), yi, xi)
    pre = a.copy()
    for _ in range(10):
        orc.dilate_nan_once(pre)
    import ctypes as C
    lib = orc.lib()
    lib.orc_bilinear_f32.restype = C.c_float
    got = np.array([lib.orc_bilinear_f32(pre.ctypes.data_as(C.POINTER(C.c_float)), 60, 80, C.c_double(y), C.c_double(x), 1)
                    for y, x in zip(yi, xi)], np.float32)
    assert ((got == stateful) | (np.isnan(got) & np.isnan(stateful))).all()
    assert np.isnan(stateful).sum() > 0      # deeper than 10 cells inland stays NaN in both


def test_c1_c2_golden_vs_oracle():
    from scenarios import Scenario
    g = golden('c1_constant_euler.npz')
    w = Scenario([('constant', {replay.U: 0.3, replay.VV: 0.2})]).oracle_world()
    lon, lat = g['lon'][0].copy(), g['lat'][0].copy()
    n = lon.size
    z, mv, cdf = np.zeros(n), np.ones(n, np.int32), np.ones(n, np.float32)
    for k in range(24):
        u, v = orc.get_environment(w, [0, 1], lon, lat, z, k * 3600.0)
        orc.advect_ocean_current(w, 0, lon, lat, z, mv, cdf, u, v, k * 3600.0, 3600.0)
        assert max(np.abs(lon - g['lon'][k + 1]).max(), np.abs(lat - g['lat'][k + 1]).max()) < 1e-7
    for name, scheme in (('euler', 0), ('rungekutta', 1), ('rungekutta4', 2)):
        g = golden('c2_double_gyre_%s.npz' % name)
        w = Scenario([('double_gyre', dict(A=0.1, epsilon=0.25, omega=0.628, t0=0.0))]).oracle_world()
        lon, lat = g['lon'][0].copy(), g['lat'][0].copy()
        n = lon.size
        z, mv, cdf = np.zeros(n), np.ones(n, np.int32), np.ones(n, np.float32)
        for k in range(g['lon'].shape[0] - 1):
            u, v = orc.get_environment(w, [0, 1], lon, lat, z, k * 0.1)
            orc.advect_ocean_current(w, scheme, lon, lat, z, mv, cdf, u, v, k * 0.1, 0.1)
        assert max(np.abs(lon - g['lon'][-1]).max(), np.abs(lat - g['lat'][-1]).max()) < 1e-9


def test_c3_golden_vs_oracle():
    g = golden('c3_grid3d_rk4_vmix.npz')
    B = replay.OracleBackend(replay.scenario_c3(g), g['lon'][0], g['lat'][0], g['z'][0])
    worst = replay.compare(replay.replay_c3(B, g, g['lon'].shape[0] - 1), g, tol_pos=1e-7, tol_z=1e-6)
    print('c3 oracle vs reference:', worst)


def test_c4_golden_vs_oracle():
    g = golden('c4_stere_rk4_hdiff_strand.npz')
    B = replay.OracleBackend(replay.scenario_c4(g), g['lon'][0], g['lat'][0], g['z'][0], wdf=float(g['wdf']))
    # step 8 exercises the uncovered RK sub-stage times (fallback velocity); in the last step the model time has left
    # the reader's time coverage: land_binary_mask (no fallback) is NaN and the reference deactivates every element
    # as 'missing_data' (report_missing_variables, basemodel/__init__.py:2501-2515)
    worst = replay.compare(replay.replay_c4(B, g, 10), g, tol_pos=1e-7)
    assert (g['status'][10] != 0).all()
    print('c4 oracle vs reference:', worst)


def test_c5_leeway_golden_vs_oracle():
    g = golden('c5_leeway_stere.npz')
    B = replay.OracleBackend(replay.scenario_c5(g), g['lon'][0], g['lat'][0], g['z'][0])
    worst = replay.compare(replay.replay_c5(B, g, g['lon'].shape[0] - 1), g, tol_pos=1e-7)
    print('c5 oracle vs reference:', worst)


def test_c5b_leeway_capsizing_golden_vs_oracle():
    """processes:capsizing (leeway.py:438-455) against the reference Leeway's own run: positions and the final set of
    capsized elements."""
    g = golden('c5b_leeway_capsizing.npz')
    B = replay.OracleBackend(replay.scenario_c5(g), g['lon'][0], g['lat'][0], g['z'][0])
    worst = replay.compare(replay.replay_c5(B, g, g['lon'].shape[0] - 1), g, tol_pos=1e-7)
    cap = np.zeros(g['lon'].shape[1])
    cap[B.ID] = B.aux[8]
    ref = np.zeros(g['lon'].shape[1])
    ref[g['ID_final']] = g['capsized_final']
    assert (cap[g['ID_final']] == ref[g['ID_final']]).all() and ref.sum() > 50
    print('c5b oracle vs reference:', worst)


@pytest.mark.parametrize('action', ['deactivate', 'previous'])
def test_c8_seafloor_actions_vs_oracle(action):
    """general:seafloor_action 'deactivate' / 'previous' (basemodel/__init__.py:748-783) against the reference's runs."""
    g = golden('c8_seafloor_actions.npz')
    sub = {k: g[action + '_' + k] for k in ('lon', 'lat', 'z', 'status')}
    B = replay.OracleBackend(replay.scenario_c8(g), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=0.0)
    # z: elements put on the sea floor during the FIRST step carry the reference's first-step float32 index arithmetic
    # in their depth (deviation 2): one float32 ulp of a ~250 m depth = 1.5e-5 m
    worst = replay.compare(replay.replay_c8(B, g, sub, action, 8), sub, tol_pos=1e-7, tol_z=2e-5)
    if action == 'deactivate':
        assert list(g['deactivate_categories']) == ['active', 'seafloor'] and (sub['status'][8] != 0).sum() == 37
    print('c8', action, worst)


@pytest.mark.parametrize('case', range(6))
def test_c11_reference_known_answers_of_an_isolated_mixing_step(case):
    """tests/models/test_run.py:359-410 (test_vertical_mixing_profiles) executed on the reference itself
    (oracle/gen_golden.py:c11): the oracle reproduces the final depths, and with them the test's published
    min / max / mean."""
    g = golden('c11_mixing_profiles.npz')
    vt, K, Kb, T, zmin, zmax, zmean = g['cases'][case]
    n = 100
    z = np.full(n, -10.0)
    zl = g['z_levels'].astype(np.float64)
    Kp = np.ascontiguousarray(np.tile(g['K_%d' % case], (n, 1)).T)
    orc.vertical_mixing(z, np.ones(n, np.int32), np.full(n, vt, np.float32), np.full(n, 100, np.float32),
                        np.zeros(n, np.float32), zl, Kp, 7200.0, float(T), 1, g['uniforms_%d' % case])
    assert np.abs(z - g['z_final_%d' % case]).max() < 1e-9
    assert abs(z.min() - zmin) < 0.05 and abs(z.max() - zmax) < 0.05 and abs(z.mean() - zmean) < 0.05


def test_reference_known_answers_of_the_vertical_interpolator():
    """tests/readers/test_interpolation.py:261-283 (test_interpolation_vertical, Linear1DInterpolator): [0.5, 2, 2.857]
    between the levels 0, 1, 3, 10 and [0, 2.2, 3] with clamping outside the levels 1, 3, 5, 10 -- reproduced by the
    oracle's world sampling of a 3D variable that equals its level number.  (The horizontal known answer of that file,
    :199-214, belongs to the optional nearest-neighbour 'ndimage' interpolator, which is not on the path.)"""
    for zgrid, z, want in (([0, -1, -3, -10], [-.5, -3, -9], [0.5, 2, 2.85714286]),
                           ([-1, -3, -5, -10], [-.5, -6, -12], [0.0, 2.2, 3])):
        wb = orc.WorldBuilder()
        lev = np.ascontiguousarray(np.broadcast_to(np.arange(4, dtype=np.float32)[:, None, None], (4, 2, 2)))
        wb.add_grid(orc.make_proj(), np.array([3.0, 5.0]), np.array([59.0, 61.0]), [(0.0, {orc.VAR['x_sea_water_velocity']: lev})],
                    z=np.array(zgrid, dtype=np.float64))
        w = wb.finish()
        n = len(z)
        (u,) = orc.get_environment(w, [orc.VAR['x_sea_water_velocity']], np.full(n, 4.0), np.full(n, 60.0), np.array(z, dtype=np.float64), 0.0)
        assert np.allclose(u, want, atol=1e-6), (u, want)


def test_c12_kelvin_temperatures_become_celsius_like_the_reference():
    """Environment.get_environment's unit check (environment.py:829-838) on a reader that is in Kelvin over half of its
    domain: golden c12 holds the reference's float32 environment right after get_environment (step 0: float32
    positions of the seeding; step 1: no current, same positions in float64)."""
    g = golden('c12_kelvin_environment.npz')
    wb = orc.WorldBuilder()
    levels = [(float(g['g_t'][k]), {orc.VAR['sea_water_temperature']: g['g_T'][k]}) for k in range(len(g['g_t']))]
    wb.add_grid(orc.make_proj(), g['g_x'], g['g_y'], levels)
    w = wb.finish()
    n = len(g['lon'])
    for k, key in enumerate(('T_env_step0', 'T_env_step1')):
        (T,) = orc.get_environment(w, [orc.VAR['sea_water_temperature']], g['lon'], g['lat'], np.zeros(n), k * float(g['dt']))
        assert T.dtype == np.float32 and (T < 100).all()
        tol = 2e-3 if k == 0 else 0.0          # first step: float32 coordinates in the reference, times the 273 K jump across one cell
        assert np.abs(T - g[key]).max() <= tol, (k, np.abs(T - g[key]).max())
    assert (g['g_T'][0] > 100).any() and (g['g_T'][0] < 100).any()


@pytest.mark.parametrize('tag', ['rk2', 'rk4'])
def test_c13_uncertainty_in_the_runge_kutta_stage_calls_vs_oracle(tag):
    """drift:current_uncertainty (+ _uniform) and drift:wind_uncertainty as the reference applies them: in the main
    get_environment call AND in every Runge-Kutta stage call (environment.py:869-891 inside physics_methods.py:638-670);
    golden c13 = the reference's own OceanDrift runs with every np.random draw recorded."""
    g = golden('c13_noise_rk.npz')
    sub = {k: g[tag + '_' + k] for k in ('lon', 'lat', 'z', 'status')}
    B = replay.OracleBackend(replay.scenario_c13(g), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=0.02)
    worst = replay.compare(replay.replay_c13(B, g, tag, sub['lon'].shape[0] - 1), sub, tol_pos=1e-7, tol_z=1e-5)
    assert list(g[tag + '_categories']) == ['active', 'seeded_on_land']
    print('c13', tag, 'oracle vs reference:', worst)
    # the stage noise matters: without it the run leaves the golden by far more than the tolerance
    B0 = replay.OracleBackend(replay.scenario_c13(g), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=0.02)
    g0 = dict(g)
    g0[tag + '_stage_noise'] = np.zeros_like(g[tag + '_stage_noise'])
    st = replay.replay_c13(B0, g0, tag, 2)
    assert np.nanmax(np.abs(st[1][0] - sub['lon'][2])) > 1e-5


def test_c14_openoil_defaults_vs_oracle():
    """The reference's own OpenOil at its default uncertainties with 'runge-kutta4' (golden c14)."""
    g = golden('c14_openoil_defaults.npz')
    for start, tol_pos, tol_z in ((0, 1e-6, 1e-4), (1, 1e-7, 1e-6)):    # from seeding: first-step float32 positions
        B = replay.OracleBackend(replay.scenario_c9(g), g['lon'][start], g['lat'][start], g['z'][start], wdf=g['wdf'])
        B.set_oil(g['diameter'][start].astype(np.float32), float(g['oil_density']), float(g['oil_viscosity']), g['film'])
        states = replay.replay_c14(B, g, 6, start=start)
        for k, (lon, lat, z, status, oil) in enumerate(states, start):
            assert np.abs(lon - g['lon'][k + 1]).max() < tol_pos and np.abs(lat - g['lat'][k + 1]).max() < tol_pos, \
                (k, np.abs(lon - g['lon'][k + 1]).max(), np.abs(lat - g['lat'][k + 1]).max())
            assert np.abs(z - g['z'][k + 1]).max() < tol_z, (k, np.abs(z - g['z'][k + 1]).max())


def test_c16_openoil_in_sea_ice_vs_oracle():
    """OpenOil.advect_oil with the Nordam / Arneborg ice factors per element (openoil.py:1179-1216): golden c16 = the
    reference's own OpenOil on a polar-stereographic reader with an ice edge across the domain (half of the elements in
    pack ice, a third in the transition zone), RK4, windage, Phillips Stokes profile, ice drift."""
    g = golden('c16_openoil_sea_ice.npz')
    B = replay.OracleBackend(replay.scenario_c16(g), g['lon'][0], g['lat'][0], g['z'][0], wdf=g['wdf'])
    states = replay.replay_c16(B, g, g['lon'].shape[0] - 1)
    worst = 0.0
    for k, (lon, lat, z, status) in enumerate(states):
        d = max(np.abs(lon - g['lon'][k + 1]).max(), np.abs(lat - g['lat'][k + 1]).max())
        worst = max(worst, d)
        assert d < 1e-7, (k, d)
    print('c16 oracle vs reference:', worst)
    # the ice matters: the same run without the factors leaves the golden by orders of magnitude more
    B0 = replay.OracleBackend(replay.scenario_c16(g), g['lon'][0], g['lat'][0], g['z'][0], wdf=g['wdf'])
    dt = float(g['dt'])
    B0.sample([replay.U, replay.VV, replay.XW, replay.YW, replay.SX, replay.SY], 0.0)
    B0.advect('runge-kutta4', 0.0, dt)
    B0.wind(dt, wdd=float(g['wind_drift_depth']))
    B0.stokes(dt, profile=2, hs_mode=1, tp_mode=3)
    assert np.abs(B0.lon - g['lon'][1]).max() > 1e-4


@pytest.mark.parametrize('tag', ['2d', '3d', 'partial'])
def test_c17_ensemble_members_of_a_reader_vs_oracle(tag):
    """ReaderBlock ensembles (readers/interpolation/structured.py:119-135): golden c17 = the reference's own OceanDrift on
    a reader that hands the current out as a list of three member arrays, RK4 + stranding.  'partial': a quarter of the
    elements start outside the reader's domain (fallback current) -- the members are numbered among the elements handed to
    the block, i.e. the covered ones (variables.py:747-765)."""
    g = golden('c17_ensemble_reader.npz')
    sub = {k: g['%s_%s' % (tag, k)] for k in ('lon', 'lat', 'z', 'status')}
    B = replay.OracleBackend(replay.scenario_c17(g, tag), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=0.0)
    worst = replay.compare(replay.replay_c17(B, g, tag, sub['lon'].shape[0] - 1), sub, tol_pos=1e-7, tol_z=1e-5)
    print('c17', tag, 'oracle vs reference:', worst)
    assert (sub['status'][-1] > 0).sum() > 10


def test_c18_windsea_swell_stokes_profile_vs_reference_function():
    """stokes_drift_profile_windsea_swell (physics_methods.py:418-456) evaluated by the reference itself on float32
    environment-like arrays (golden c18).  Agreement is exact to float64 round-off wherever NumPy's float32 cos / sin
    (SIMD, not correctly rounded: 17 % of the values differ from libm's by one ulp) and libm agree; a one-ulp difference of
    a unit vector is amplified where swell and wind sea run nearly parallel (the split divides by the sine of their angle)."""
    g = golden('c18_windsea_swell_profile.npz')
    u, v = orc.stokes_windsea_swell(g['z'], g['sx'], g['sy'], g['swell_dir'], g['swell_tp'], g['swell_hs'], g['ww_dir'],
                                    g['ww_tm'], g['ww_hs'])
    err = np.hypot(u - g['stokes_u'], v - g['stokes_v'])
    assert np.isfinite(err).all() and err.max() < 1e-6 and np.median(err) < 1e-15
    assert (u[:20] == 0).all() and (v[:20] == 0).all()              # zero surface drift stays zero
    cond = np.abs(np.sin(np.radians(g['swell_dir'].astype(float) - g['ww_dir'].astype(float))))
    assert err[cond > 0.5].max() < 5e-8


def test_mercator_and_lambert_reproduce_snyders_numerical_examples():
    """oracle/proj.c (and the host-side NumPy projections of opendrift_amd/projection.py) against the worked examples of
    Snyder, Map Projections -- A Working Manual (USGS PP 1395): Mercator pp. 266-267 (sphere, Clarke 1866 ellipsoid),
    Lambert conformal conic pp. 295-297 (sphere, Clarke 1866), forward and inverse.  PROJ itself is not in this image: these
    are the known answers the two projections are pinned on."""
    from oracle import oracle as orc
    from opendrift_amd.projection import Proj
    es = 0.00676866                      # Clarke 1866 (Snyder's examples), a = 6378206.4 m
    rf = float(1 / (1 - np.sqrt(1 - es)))
    cases = [
        (orc.make_proj(orc.PROJ_LCC, a=6378206.4, es=es, lat0=23, lon0=-96, lat1=33, lat2=45),
         '+proj=lcc +lat_1=33 +lat_2=45 +lat_0=23 +lon_0=-96 +a=6378206.4 +rf=%.12f' % rf, 1894410.9, 1564649.5, 0.06),
        (orc.make_proj(orc.PROJ_MERC, a=6378206.4, es=es, lon0=-180, lat_ts=0.0),
         '+proj=merc +lon_0=-180 +a=6378206.4 +rf=%.12f' % rf, 11688673.7, 4139145.6, 0.06),
        (orc.make_proj(orc.PROJ_LCC, a=1.0, es=0.0, lat0=23, lon0=-96, lat1=33, lat2=45),
         '+proj=lcc +lat_1=33 +lat_2=45 +lat_0=23 +lon_0=-96 +R=1', 0.2966785, 0.2462112, 6e-8),
        (orc.make_proj(orc.PROJ_MERC, a=1.0, es=0.0, lon0=-180, lat_ts=0.0), '+proj=merc +lon_0=-180 +R=1', 1.8325957, 0.6528366, 6e-8),
    ]
    for op, proj4, x_want, y_want, tol in cases:
        x, y = orc.proj_fwd(op, -75.0, 35.0)
        assert abs(x[0] - x_want) < tol and abs(y[0] - y_want) < tol, (proj4, x, y)
        lon, lat = orc.proj_inv(op, x, y)
        assert abs(lon[0] + 75.0) < 1e-12 and abs(lat[0] - 35.0) < 1e-12
        hx, hy = Proj(proj4)(-75.0, 35.0)
        assert abs(hx - x[0]) < 1e-6 * max(1.0, abs(x_want) * 1e-6) and abs(hy - y[0]) < 1e-6 * max(1.0, abs(y_want) * 1e-6)
        hl, hp = Proj(proj4)(hx, hy, inverse=True)
        assert abs(hl + 75.0) < 1e-10 and abs(hp - 35.0) < 1e-10


@pytest.mark.parametrize('tag', ['lcc_sphere', 'lcc_wgs84', 'merc_wgs84'])
def test_c20_lambert_and_mercator_golden_vs_oracle(tag):
    """The C oracle (oracle/step.c + proj.c: lonlat2xy through the projection, vectors rotated by the azimuth of the reader's
    +y axis from the 10 m finite difference and the WGS84 geodesic inverse) replays the reference's own runs on Lambert
    conformal conic and Mercator grids (oracle/gen_golden_proj.py).  lcc_wgs84 lies at negative longitudes: the reference's
    first step modulates the still-float32 longitudes in float32 (variables.py:259-280) -- 2.7e-7 deg on the worst element,
    see tests/test_gpu_model_api.py::test_c20_* -- the oracle, like the device, works in float64."""
    g = golden('c20_lcc_merc_rk4.npz')
    sub = {k: g['%s_%s' % (tag, k)] for k in ('lon', 'lat', 'z', 'status')}
    nst = sub['lon'].shape[0] - 1
    B = replay.OracleBackend(replay.scenario_c20(g, tag), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=float(g['wdf']))
    worst = replay.compare(replay.replay_c20(B, g, tag, nst), sub, tol_pos=1e-7)
    assert (sub['status'][nst] != 0).sum() > 5
    print('c20', tag, 'oracle vs reference:', worst)


def test_mercator_and_lambert_round_trips_and_hemispheres():
    """forward then inverse is the identity (< 1e-9 deg) for the oracle's and the host's Mercator / Lambert conformal conic
    over wide domains, including a cone of the SOUTHERN hemisphere (negative cone constant: the sign handling of Snyder
    14-10 / 14-11), a tangent cone, a true-scale latitude on the ellipsoid, false origins and the poles of the cone; oracle
    and host agree to 1e-6 m."""
    from oracle import oracle as orc
    from opendrift_amd.projection import Proj, parse_proj4
    rng = np.random.default_rng(12)
    cases = ['+proj=lcc +lat_1=-30 +lat_2=-60 +lat_0=-45 +lon_0=140 +x_0=1000 +y_0=-2000 +ellps=WGS84',
             '+proj=lcc +lat_1=63.3 +lat_0=63.3 +lon_0=15 +R=6371000',
             '+proj=lcc +lat_1=25 +lat_2=47 +lat_0=36 +lon_0=-100 +ellps=GRS80',
             '+proj=merc +lon_0=100 +lat_ts=-41 +ellps=WGS84 +x_0=5e5',
             '+proj=merc +lon_0=0 +k_0=0.9996 +R=6371229']
    for proj4 in cases:
        pr = parse_proj4(proj4)
        f = 0.0 if not pr['rf'] else 1.0 / pr['rf']
        op = orc.make_proj(orc.PROJ_LCC if pr['kind'] == 'lcc' else orc.PROJ_MERC, a=pr['a'], es=f * (2 - f), lat0=pr['lat0'], lon0=pr['lon0'],
                           lat_ts=pr.get('lat_ts', 0.0), k0=pr['k0'], x0=pr['x0'], y0=pr['y0'], lat1=pr.get('lat1', 0.0), lat2=pr.get('lat2'))
        south = pr['kind'] == 'lcc' and pr['lat1'] < 0
        lon = pr['lon0'] + rng.uniform(-170, 170, 3000)
        lat = rng.uniform(-85, 20, 3000) if south else (rng.uniform(-20, 85, 3000) if pr['kind'] == 'lcc' else rng.uniform(-84, 84, 3000))
        x, y = orc.proj_fwd(op, lon, lat)
        lo, la = orc.proj_inv(op, x, y)
        dlon = (lo - lon + 180) % 360 - 180
        assert np.abs(dlon).max() < 1e-9 and np.abs(la - lat).max() < 1e-9, (proj4, np.abs(dlon).max(), np.abs(la - lat).max())
        hx, hy = Proj(proj4)(lon, lat)
        scale = max(1.0, float(np.abs(x).max()) * 1e-12)
        assert np.abs(hx - x).max() < 1e-6 * max(1.0, scale) + 1e-6 and np.abs(hy - y).max() < 1e-6 * max(1.0, scale) + 1e-6, proj4
        hl, hp = Proj(proj4)(hx, hy, inverse=True)
        assert np.abs((hl - lon + 180) % 360 - 180).max() < 1e-9 and np.abs(hp - lat).max() < 1e-9
    # the apex of a cone maps to its pole
    op = orc.make_proj(orc.PROJ_LCC, a=6378137.0, es=0.00669438, lat0=40, lon0=10, lat1=30, lat2=50)
    x, y = orc.proj_fwd(op, 77.0, 90.0)
    lo, la = orc.proj_inv(op, x, y)
    assert abs(la[0] - 90.0) < 1e-9


# ------------------------------------------------------------------ round 5: tmerc / utm, laea, oblique stere, rotated pole
def test_round5_projections_reproduce_snyders_numerical_examples():
    """oracle/proj.c (and the host's NumPy restatement, opendrift_amd/projection.py) against the numerical examples of Snyder,
    Map Projections -- A Working Manual (USGS PP 1395), Appendix A: transverse Mercator (sphere p. 268, Clarke 1866 p. 269),
    stereographic oblique (sphere p. 312, Clarke 1866 p. 313), Lambert azimuthal equal-area (sphere p. 332, Clarke 1866 oblique
    p. 333, International polar p. 334) -- forward to the digits printed, inverse to 5e-6 deg (the printed coordinates are
    rounded to 0.1 m, or to 7 digits of a unit sphere); PROJ itself is not in this image."""
    from opendrift_amd.projection import Proj
    a, es = 6378206.4, 0.00676866          # Clarke 1866
    rf = float(1 / (1 - np.sqrt(1 - es)))
    ell = '+a=%r +rf=%r' % (a, rf)
    cases = [
        (orc.make_proj(orc.PROJ_TMERC, a=1.0, es=0.0, lat0=0, lon0=-75, k0=1.0), '+proj=tmerc +lon_0=-75 +R=1',
         (-73.5, 40.5), (0.0199077, 0.7070276), 5e-8),
        (orc.make_proj(orc.PROJ_TMERC, a=a, es=es, lat0=0, lon0=-75, k0=0.9996), '+proj=tmerc +lon_0=-75 +k=0.9996 ' + ell,
         (-73.5, 40.5), (127106.5, 4484124.4), 0.06),
        (orc.make_proj(orc.PROJ_STERE_OBLIQUE, a=1.0, es=0.0, lat0=40, lon0=-100, k0=1.0), '+proj=stere +lat_0=40 +lon_0=-100 +R=1',
         (-75.0, 30.0), (0.3807224, -0.1263802), 5e-8),
        (orc.make_proj(orc.PROJ_STERE_OBLIQUE, a=a, es=es, lat0=40, lon0=-100, k0=0.9999), '+proj=stere +lat_0=40 +lon_0=-100 +k=0.9999 ' + ell,
         (-90.0, 30.0), (971630.8, -1063049.3), 0.06),
        (orc.make_proj(orc.PROJ_LAEA, a=3.0, es=0.0, lat0=40, lon0=-100), '+proj=laea +lat_0=40 +lon_0=-100 +R=3',
         (100.0, -20.0), (-4.2339303, 4.0257775), 5e-8),
        (orc.make_proj(orc.PROJ_LAEA, a=a, es=es, lat0=40, lon0=-100), '+proj=laea +lat_0=40 +lon_0=-100 ' + ell,
         (-110.0, 30.0), (-965932.1, -1056814.9), 0.06),
        (orc.make_proj(orc.PROJ_LAEA, a=6378388.0, es=0.00672267, lat0=90, lon0=-100),
         '+proj=laea +lat_0=90 +lon_0=-100 +a=6378388.0 +rf=%r' % float(1 / (1 - np.sqrt(1 - 0.00672267))),
         (5.0, 80.0), (1077459.7, 288704.5), 0.06),
    ]
    for op, proj4, (lon, lat), (xw, yw), tol in cases:
        x, y = orc.proj_fwd(op, lon, lat)
        assert abs(x[0] - xw) < tol and abs(y[0] - yw) < tol, (proj4, x, y)
        lo, la = orc.proj_inv(op, xw, yw)
        assert abs((lo[0] - lon + 180) % 360 - 180) < 5e-6 and abs(la[0] - lat) < 5e-6, (proj4, lo, la)
        lo, la = orc.proj_inv(op, x, y)
        assert abs((lo[0] - lon + 180) % 360 - 180) < 1e-11 and abs(la[0] - lat) < 1e-11
        hx, hy = Proj(proj4)(lon, lat)
        assert abs(hx - x[0]) < 1e-8 * max(1.0, abs(xw)) and abs(hy - y[0]) < 1e-8 * max(1.0, abs(yw)), (proj4, hx, hy)
    # UTM: zone 33 has its central meridian at 15 E; the northing of 60 N on it is k0 times the meridian arc
    hx, hy = Proj('+proj=utm +zone=33 +ellps=WGS84')(15.0, 60.0)
    assert abs(hx - 500000.0) < 1e-6 and abs(hy - 6651411.190) < 2e-3
    hx, hy = Proj('+proj=utm +zone=33 +south +ellps=WGS84')(15.0, -60.0)
    assert abs(hx - 500000.0) < 1e-6 and abs(hy - (10000000.0 - 6651411.190)) < 2e-3


def test_transverse_mercator_series_against_the_meridian_arc_and_conformality():
    """Independent of Snyder's printed numbers: on the central meridian the northing of Krueger's series is k0 times the
    meridian arc (numerical quadrature of M(phi) = a (1 - e^2) / (1 - e^2 sin^2 phi)^(3/2), scipy), and the mapping is conformal
    -- the Cauchy-Riemann equations hold for finite differences in isometric latitude -- 3 degrees off it; the rotated pole
    against the same rotation written as two matrix products."""
    from scipy.integrate import quad
    a, rf = 6378137.0, 298.257223563
    f = 1 / rf
    es = f * (2 - f)
    op = orc.make_proj(orc.PROJ_TMERC, a=a, es=es, lat0=0, lon0=9, k0=0.9996, x0=500000.0)
    for lat in (5.0, 37.0, 61.5, 84.0):
        arc = quad(lambda ph: a * (1 - es) / (1 - es * np.sin(ph) ** 2) ** 1.5, 0, np.radians(lat), epsabs=1e-6)[0]
        x, y = orc.proj_fwd(op, 9.0, lat)
        assert abs(x[0] - 500000.0) < 1e-6 and abs(y[0] - 0.9996 * arc) < 2e-5, (lat, y[0] - 0.9996 * arc)
    # conformality: d(x + i y) / d(lambda + i psi) is one complex number -- x_lam = y_psi, y_lam = -x_psi (psi = isometric latitude)
    e = np.sqrt(es)
    psi = lambda ph: np.arctanh(np.sin(ph)) - e * np.arctanh(e * np.sin(ph))
    for lon, lat in ((12.0, 60.0), (6.5, 45.0)):
        h = 1e-5
        ph = np.radians(lat)
        dpsi = psi(ph + np.radians(h)) - psi(ph - np.radians(h))
        xe, ye = orc.proj_fwd(op, [lon + h, lon - h], [lat, lat])
        xn, yn = orc.proj_fwd(op, [lon, lon], [lat + h, lat - h])
        x_lam, y_lam = (xe[0] - xe[1]) / np.radians(2 * h), (ye[0] - ye[1]) / np.radians(2 * h)
        x_psi, y_psi = (xn[0] - xn[1]) / dpsi, (yn[0] - yn[1]) / dpsi
        assert abs(x_lam - y_psi) < 2e-3 * 1e-3 * abs(x_lam) + 1e-2 and abs(y_lam + x_psi) < 1e-2, (x_lam, y_psi, y_lam, x_psi)
    # rotated pole: (lon, lat) -> unit vector -> rotate about z by -lon_0, about y by (90 - o_lat_p) -> rotated lon / lat
    lon0, latp = -40.0, 22.0
    op = orc.make_proj(orc.PROJ_OB_TRAN, lon0=lon0, lat1=latp, lat2=0.0)
    rng = np.random.default_rng(5)
    lon, lat = rng.uniform(-20, 40, 500), rng.uniform(50, 80, 500)
    lam, phi = np.radians(lon - lon0), np.radians(lat)
    v = np.stack([np.cos(phi) * np.cos(lam), np.cos(phi) * np.sin(lam), np.sin(phi)])
    t = np.radians(90.0 - latp)
    R = np.array([[np.cos(t), 0, np.sin(t)], [0, 1, 0], [-np.sin(t), 0, np.cos(t)]])
    w = R @ v
    x, y = orc.proj_fwd(op, lon, lat)

    assert np.abs(np.degrees(np.arcsin(w[2])) - y).max() < 1e-11
    d = (np.degrees(np.arctan2(w[1], w[0])) - x + 180) % 360 - 180
    assert np.abs(d).max() < 1e-11, np.abs(d).max()
    lo, la = orc.proj_inv(op, x, y)
    assert np.abs(lo - lon).max() < 1e-11 and np.abs(la - lat).max() < 1e-11


@pytest.mark.parametrize('tag', ['utm33', 'laea_grs80', 'stere_oblique', 'rotated_pole'])
def test_c23_round5_projections_golden_vs_oracle(tag):
    """The C oracle replays the reference's own RK4 + wind + Stokes drift + stranding runs on a UTM grid, an ETRS89-LAEA grid, an
    oblique stereographic grid and a rotated-pole grid (oracle/gen_golden_proj2.py): lonlat2xy through the projection, vectors
    rotated by the azimuth of the reader's +y axis (10 m line; 0.1 degree for the rotated pole)."""
    g = golden('c23_proj_rk4.npz')
    sub = {k: g['%s_%s' % (tag, k)] for k in ('lon', 'lat', 'z', 'status')}
    nst = sub['lon'].shape[0] - 1
    B = replay.OracleBackend(replay.scenario_c23(g, tag), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=float(g['wdf']))
    worst = replay.compare(replay.replay_c20(B, g, tag, nst), sub, tol_pos=1e-7)
    assert (sub['status'][nst] != 0).sum() > 5
    print('c23', tag, 'oracle vs reference:', worst)


def test_c24c_truncation_on_a_reader_that_cuts_its_block_golden_vs_oracle():
    """drift:truncate_ocean_model_below_m with diffusivity profiles from a reader that hands out the LEVELS ASKED FOR (what the
    reference's file readers do: reader_netCDF_CF_generic.py:414-423): the block ends one level + verticalbuffer below the
    truncation depth (5 of 8 levels here) and the elements below mix on K and dK/dz of its last level -- the reference's own
    run (golden c24c, oracle/gen_golden_profiles.py: 7 m away in z from the run on whole columns, c24a)."""
    g = golden('c24_profiles.npz')
    sub = {k: g['c_%s' % k] for k in ('lon', 'lat', 'z', 'status')}
    nst = sub['lon'].shape[0] - 1
    assert list(g['c_levels_handed_out']) == [5] and replay.cf_reader_levels(g['a_g_z'], float(g['truncate'])) == 5
    assert np.nanmax(np.abs(g['c_z'][-1] - g['a_z'][-1])) > 1.0                       # the cut matters
    B = replay.OracleBackend(replay.scenario_c24(g, 'a'), sub['lon'][0], sub['lat'][0], sub['z'][0])
    worst = replay.compare(replay.replay_c24(B, g, 'c', nst, truncate=float(g['truncate']), gtag='a', cut_levels=True), sub,
                           tol_pos=1e-7, tol_z=1e-5)
    print('c24c oracle vs reference:', worst)


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_c24_profile_paths_golden_vs_oracle(tag):
    """The C oracle replays the reference's own runs (oracle/gen_golden_profiles.py) of (a) drift:truncate_ocean_model_below_m
    TOGETHER with vertical mixing on reader diffusivity profiles -- every sampling call sees max(z, -20 m), the columns are the
    reader's, whole -- and (b) an ensemble reader whose ocean_vertical_diffusivity is a list of three members: element j of the
    call mixes on the column of member j % 3 (readers/interpolation/structured.py:119-135)."""
    g = golden('c24_profiles.npz')
    sub = {k: g['%s_%s' % (tag, k)] for k in ('lon', 'lat', 'z', 'status')}
    nst = sub['lon'].shape[0] - 1
    B = replay.OracleBackend(replay.scenario_c24(g, tag), sub['lon'][0], sub['lat'][0], sub['z'][0])
    worst = replay.compare(replay.replay_c24(B, g, tag, nst, truncate=float(g['truncate']) if tag == 'a' else None), sub,
                           tol_pos=1e-7, tol_z=1e-5)
    print('c24' + tag, 'oracle vs reference:', worst)
    if tag == 'a':      # the option matters, and half of the elements are below the truncation depth
        assert np.nanmax(np.abs(g['a_lon'][-1] - g['a0_lon'][-1])) > 1e-3 and (g['a_z'][0] < -20).sum() > 100
    else:               # the members differ by far more than anything else does: mixing on the wrong member's column is seen
        dz = sub['z'][-1] - sub['z'][0]
        rms = [np.sqrt(np.nanmean(dz[m::3] ** 2)) for m in range(3)]
        assert rms[2] > 1.8 * rms[0] > 1.8 * 1.5 * rms[1]


def test_closed_form_vector_rotation_of_conformal_readers_against_the_geodesic_inverse():
    """The FAST stage samples of polar stereographic, Lambert and Mercator readers take cos / sin of rotate_vectors' angle
    (variables.py:59-109: minus the WGS84 azimuth of the 10 m line (x, y) -> (x, y + 10)) from closed forms (csrc/odr_field.hip.h:
    proj_fwd_near, rot_closed_form).  The formulas, restated in NumPy, against the reference's own recipe
    evaluated with the oracle (projection inverse + geodesic inverse): <= 1e-9 rad where the projection's ellipsoid is the
    geodesic's; on a SPHERE the map keeps the sphere's azimuths and the closed form is off by ~2e-3 rad (the device keeps
    rotation_angle there: rot_same_ellipsoid)."""
    rng = np.random.default_rng(1)
    es = 0.00669437999014

    def reference(p, lon, lat):
        x, y = orc.proj_fwd(p, lon, lat)
        lo2, la2 = orc.proj_inv(p, x, y + 10.0)
        az, _ = orc.geod_inv(lon, lat, lo2, la2)
        return -np.radians(az), x, y

    def off(r, az1):
        return np.abs(np.angle(np.exp(1j * (r + az1)))).max()

    n = 2000
    # polar stereographic, both hemispheres
    for south, lat0, lat_ts, lon0 in ((False, 90.0, 60.0, 70.0), (True, -90.0, -70.0, -30.0)):
        p = orc.make_proj(orc.PROJ_STERE_POLAR, a=6378137.0, es=es, lat0=lat0, lon0=lon0, lat_ts=lat_ts, x0=1e5, y0=-2e5)
        lon, lat = rng.uniform(-180, 180, n), rng.uniform(55, 89, n) * (-1 if south else 1)
        r, x, y = reference(p, lon, lat)
        D = np.radians(((lon - lon0 + 180) % 360) - 180)
        q = 5.0 * np.sin(D) / np.hypot(x - 1e5, y + 2e5) * (1 - np.abs(np.sin(np.radians(lat))))
        cs, sn = (np.cos(D) + q * np.sin(D), np.sin(D) - q * np.cos(D)) if south else (np.cos(D) - q * np.sin(D), -np.sin(D) - q * np.cos(D))
        assert np.abs(cs - np.cos(r)).max() < 1e-9 and np.abs(sn - np.sin(r)).max() < 1e-9
        assert 1e-7 < off(r, -D if south else D) < 1e-6          # (without the chord term)
    # Lambert conformal conic, northern and southern cone; Mercator
    for kw in (dict(a=6378137.0, es=es, lat0=63.3, lon0=15.0, lat1=63.3, lat2=63.3, x0=1e5, y0=2e5, k0=1.0),
               dict(a=6378137.0, es=es, lat0=-40.0, lon0=140.0, lat1=-30.0, lat2=-50.0, x0=0.0, y0=0.0, k0=0.9996)):
        p = orc.make_proj(orc.PROJ_LCC, **kw)
        lon = kw['lon0'] + rng.uniform(-40, 40, n)
        lat = rng.uniform(35, 80, n) * (-1 if kw['lat0'] < 0 else 1)
        r, x, y = reference(p, lon, lat)
        X, Yr = (x - kw['x0']) / (p.a * p.k0), p.rho0 - (y - kw['y0']) / (p.a * p.k0)
        ir = (1.0 if p.n > 0 else -1.0) / np.hypot(X, Yr)
        st, ct = X * ir, Yr * ir
        c = 5.0 / (p.a * p.k0) * st * ir * (1.0 - np.sin(np.radians(lat)) / p.n)
        assert np.abs(ct - c * st - np.cos(r)).max() < 1e-9 and np.abs(-(st + c * ct) - np.sin(r)).max() < 1e-9
    p = orc.make_proj(orc.PROJ_MERC, a=6378137.0, es=es, lat0=0.0, lon0=10.0, lat_ts=30.0)
    r, _, _ = reference(p, 10.0 + rng.uniform(-60, 60, n), rng.uniform(-70, 70, n))
    assert np.abs(r).max() < 1e-9                                # no rotation at all
    # a conformal map of a SPHERE keeps the sphere's azimuths, not WGS84's
    p = orc.make_proj(orc.PROJ_LCC, a=6371000.0, es=0.0, lat0=60.0, lon0=-20.0, lat1=50.0, lat2=70.0)
    lon, lat = -20.0 + rng.uniform(-40, 40, n), rng.uniform(35, 80, n)
    r, x, y = reference(p, lon, lat)
    assert off(r, p.n * np.radians(lon + 20.0)) > 1e-3
