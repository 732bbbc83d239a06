"""CPU: host-side mirror of the reference interface (config validation, reader surface, seeding
rules) -- no device calls."""
import os
from datetime import datetime, timedelta

import numpy as np
import pytest

from opendrift_amd.oceandrift import OceanDrift, WrongMode
from opendrift_amd import readers, projection, synthetic


def test_config_validation_like_reference():
    o = OceanDrift(loglevel=50)
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    assert o.get_config('drift:advection_scheme') == 'runge-kutta4'
    with pytest.raises(ValueError):
        o.set_config('drift:advection_scheme', 'rk45')            # enum
    with pytest.raises(ValueError):
        o.set_config('vertical_mixing:timestep', 1e6)             # max
    with pytest.raises(ValueError):
        o.set_config('drift:no_such_key', 1)
    with pytest.raises(ValueError):
        o.set_config('drift:vertical_mixing', 'yes')              # bool
    assert o.get_config('environment:fallback:ocean_mixed_layer_thickness') == 50
    assert o.get_config('environment:fallback:land_binary_mask') is None


def test_seeding_rules_and_modes():
    o = OceanDrift(loglevel=50)
    t = datetime(2020, 1, 1)
    with pytest.raises(ValueError):
        o.seed_elements(lon=[4, 5], lat=[60], time=t)
    with pytest.raises(ValueError):
        o.seed_elements(lon=4, lat=95, time=t)
    with pytest.raises(ValueError):
        o.seed_elements(lon=[4, 5, 6], lat=[60, 61, 62], number=4, time=t)
    o.seed_elements(lon=[4, 5], lat=[60, 61], number=6, time=[t, t + timedelta(hours=5)], z=-3.2)
    assert o.num_elements_total() == 6 and o.mode == 'Ready'
    assert (o._sched['lon'] == np.array([4, 4, 4, 5, 5, 5.])).all()
    assert o._sched['time'][1] == t + timedelta(hours=1)
    assert o._sched['z'][0] == np.float64(np.float32(-3.2))           # float32 at seeding (elements.py:71-88)
    with pytest.raises(WrongMode):
        o.set_config('drift:advection_scheme', 'euler')               # config only in mode Config
    with pytest.raises(ValueError):
        o.run(steps=3, duration=timedelta(hours=1))
    with pytest.raises(TypeError):
        OceanDrift(loglevel=50).add_reader(object())


def test_reader_surface():
    r = readers.DoubleGyreReader(initial_time=datetime(2000, 1, 1), epsilon=0.25, omega=0.628, A=0.1)
    x, y = np.array([0.3, 1.2]), np.array([0.4, 0.8])
    lon, lat = r.xy2lonlat(x, y)
    x2, y2 = r.lonlat2xy(lon, lat)
    assert np.abs(x2 - x).max() < 1e-9 and np.abs(y2 - y).max() < 1e-9
    v = r.get_variables(r.variables, datetime(2000, 1, 1, 0, 0, 5), x, y)
    assert set(v) >= {'x_sea_water_velocity', 'y_sea_water_velocity', 'x', 'y', 'time'}
    g = synthetic.grid3d(nx=16, ny=12, nz=4, nt=3, seed=0)
    times = [datetime(2020, 1, 1) + timedelta(seconds=float(t)) for t in g['t']]
    gr = readers.GridReader(g['x'], g['y'], times, {'x_sea_water_velocity': g['x_sea_water_velocity']}, z=g['z'])
    blk = gr.get_variables(['x_sea_water_velocity'], times[1])
    assert blk['x_sea_water_velocity'].shape == (4, 12, 16) and blk['time'] == times[1]
    assert gr.nearest_time(times[0] + timedelta(minutes=30)) == (0, 1) and gr.nearest_time(times[1]) == (1, 1)
    assert gr.covers_time(times[2]) and not gr.covers_time(times[2] + timedelta(seconds=1))
    p = projection.parse_proj4(synthetic.NORKYST_PROJ4)
    assert p['kind'] == 'stere_polar' and p['lat_ts'] == 60 and p['lon0'] == 70 and abs(p['rf'] - 298.257223563) < 1e-9
    q = projection.parse_proj4('+proj=lcc +lat_1=49.5 +lon_0=10 +R=6371000')     # tangent cone: lat_0 = lat_2 = lat_1
    assert q['kind'] == 'lcc' and q['lat1'] == q['lat2'] == q['lat0'] == 49.5 and q['rf'] == 0.0
    q = projection.parse_proj4('+proj=ob_tran +o_proj=longlat +o_lat_p=22 +lon_0=-40')      # rotated pole: round 5
    assert q['kind'] == 'ob_tran' and q['lat1'] == 22 and q['lat2'] == 0 and q['lon0'] == -40
    q = projection.parse_proj4('+proj=utm +zone=33 +ellps=WGS84')
    assert q['kind'] == 'tmerc' and q['lon0'] == 15 and q['k0'] == 0.9996 and q['x0'] == 500000 and q['y0'] == 0
    assert projection.parse_proj4('+proj=stere +lat_0=52 +lon_0=5 +ellps=WGS84')['kind'] == 'stere_oblique'
    assert projection.parse_proj4('+proj=stere +lat_0=0 +lon_0=0 +lat_ts=0 +a=6.371e6 +e=0')['kind'] == 'stere_equit_sphere'
    with pytest.raises(NotImplementedError):
        projection.parse_proj4('+proj=ob_tran +o_proj=longlat +o_lat_p=22 +lon_0=-40 +to_meter=0.0174532925199433')
    with pytest.raises(NotImplementedError):
        projection.parse_proj4('+proj=aea +lat_1=50 +lat_2=70')


def test_openoil_host_interface():
    """OpenOil mirror: its own required variables and defaults (openoil.py:221-296, :493-499), the oil given as numbers,
    droplet sizes of elements seeded below the surface drawn like the reference (:1659-1700), keep_droplet_diameter."""
    from opendrift_amd.openoil import OpenOil
    t = datetime(2020, 1, 1)
    o = OpenOil(loglevel=50)
    assert o.required_variables['x_wind']['fallback'] is None and o.required_variables['sea_water_temperature']['fallback'] == 10
    assert o.get_config('drift:vertical_mixing') is True and o.get_config('drift:wind_uncertainty') == 0.5
    assert o.get_config('wave_entrainment:droplet_size_distribution') == 'Johansen et al. (2015)'
    with pytest.raises(ValueError):
        o.set_config('wave_entrainment:droplet_size_distribution', 'Delvigne')
    with pytest.raises(NotImplementedError):
        o.set_config('processes:evaporation', True)
    with pytest.raises(ValueError, match='ADIOS'):
        o.set_oiltype('GENERIC BUNKER C')
    with pytest.raises(ValueError, match='Unknown oil properties'):
        o.set_oiltype({'density': 900., 'pour_point': 3.})
    with pytest.raises(ValueError, match='deprecated'):
        o.seed_elements(lon=4.0, lat=60.0, time=t, oiltype='x')
    # surface seeding: no droplet sizes drawn, diameter 0, oil properties from the dict
    np.random.seed(0)
    state = np.random.get_state()[1].copy()
    o.seed_elements(lon=4.0, lat=60.0, number=5, time=t, oil_type={'density': 920.0, 'viscosity': 0.01,
                                                                 'oil_water_interfacial_tension': 0.025})
    assert (np.random.get_state()[1] == state).all() and o.keep_droplet_diameter is False
    assert (o._sched['diameter'] == 0).all() and (o._sched['density'] == np.float32(920)).all()
    assert o.oil_water_interfacial_tension == 0.025 and o._sched['oil_film_thickness'].dtype == np.float32
    # sub-surface seeding: np.random.uniform(min_subsea, max_subsea, number)
    np.random.seed(1)
    want = np.random.uniform(0.0005, 0.005, 4)
    o2 = OpenOil(loglevel=50)      # the constructor seeds np.random (seed=0), like the reference's
    np.random.seed(1)
    o2.seed_elements(lon=4.0, lat=60.0, z=-10.0, number=4, time=t)
    assert np.array_equal(o2._sched['diameter'], want.astype(np.float32)) and o2.keep_droplet_diameter is False
    assert o2.oiltype['density'] == 880.0                       # Oil element defaults when no oil is named
    o3 = OpenOil(loglevel=50)
    o3.seed_elements(lon=[4.0, 4.1], lat=[60.0, 60.1], z=[-5.0, 0.0], time=t, diameter=2e-4)
    assert o3.keep_droplet_diameter is True and (o3._sched['diameter'] == np.float32(2e-4)).all()
    with pytest.raises(ValueError, match='diameter has length'):
        OpenOil(loglevel=50).seed_elements(lon=[4.0, 4.1], lat=[60.0, 60.1], time=t, diameter=[1e-4, 2e-4, 3e-4])
    assert OpenOil.aux_properties[:4] == ['diameter', 'density', 'viscosity', 'oil_film_thickness']


def test_landmask_raster_reader_surface():
    r = readers.LandmaskRasterReader(3.0, 59.0, 0.5, 0.25, np.array([[0, 1, 0], [1, 1, 0]]))
    assert r.variables == ['land_binary_mask'] and r.name == 'global_landmask' and r.device_kind == 'landmask'
    out = r.get_variables(['land_binary_mask'], None, np.array([3.6, 3.6 + 360, 2.9, 3.2, 4.51]), np.array([59.1, 59.3, 59.1, 59.3, 59.3]))
    assert out['land_binary_mask'].tolist() == [True, True, False, True, False]
    f = readers.FailingReader()
    with pytest.raises(ValueError):
        f.get_variables(['x_wind'])
    o = OceanDrift(loglevel=50)
    assert o.discarded_readers == {}


def test_read_objectprop_layout(tmp_path):
    """OBJECTPROP.DAT as the reference reads it (leeway.py:186-219): key line, description line, nine numbers; the
    first blank line ends the table."""
    from opendrift_amd.leeway import read_objectprop
    f = tmp_path / 'OBJECTPROP.DAT'
    f.write_text(' PIW-1                        1\n Person-in-water (PIW), unknown state (mean values)\n'
                 '       0.96     0.00     12.00      0.54      0.00      9.40     -0.54      0.00      9.40\n'
                 ' PIW-2                        2\n >PIW, vertical PFD type III conscious\n'
                 '       0.48     0.00      8.30      0.15      0.00      6.70     -0.15      0.00      6.70\n'
                 '\n ignored\n')
    t = read_objectprop(str(f))
    assert list(t) == [1, 2] and t[1]['OBJKEY'] == 'PIW-1' and t[2]['Description'] == '>PIW, vertical PFD type III conscious'
    assert (t[1]['DWSLOPE'], t[1]['DWSTD'], t[1]['CWLSLOPE'], t[2]['CWRSTD']) == (0.96, 12.0, -0.54, 6.7)


def test_constant_reader_values_per_element_id_on_the_host():
    """reader_constant with 'element_ID' (reader_constant.py:42-80): the listed IDs get their values -- one value for all of
    them or one each --, everything else NaN (the next reader's turn)."""
    import numpy as np
    from opendrift_amd import readers
    r = readers.ConstantReader({'x_wind': np.array([1.0, 2.0, 3.0]), 'y_wind': 5.0, 'element_ID': [4, 9, 2]})
    assert getattr(r, 'device_kind', 'constant') is None          # evaluated on the host at the element positions
    r._element_ID = np.array([0, 2, 4, 7, 9])
    out = r.get_variables(['x_wind', 'y_wind'], None, np.zeros(5), np.zeros(5), np.zeros(5))
    assert np.array_equal(out['x_wind'], [np.nan, 3.0, 1.0, np.nan, 2.0], equal_nan=True)
    assert np.array_equal(out['y_wind'], [np.nan, 5.0, 5.0, np.nan, 5.0], equal_nan=True)
    plain = readers.ConstantReader({'x_wind': 3.0})
    assert plain.device_kind == 'constant' and np.all(plain.get_variables(['x_wind'], None, np.zeros(3))['x_wind'] == 3.0)


def test_grid_reader_hands_out_ensemble_members_as_a_list():
    """basereader/structured.py:125-147: a variable may be a list of member arrays; the window cut applies to every member."""
    import numpy as np
    from datetime import datetime, timedelta
    from opendrift_amd import readers
    x, y = np.linspace(0, 9, 10), np.linspace(50, 57, 8)
    t = [datetime(2020, 1, 1) + timedelta(hours=k) for k in range(2)]
    members = [np.full((2, 8, 10), float(m), np.float32) + np.arange(10, dtype=np.float32) for m in range(3)]
    r = readers.GridReader(x, y, t, {'x_sea_water_velocity': members, 'land_binary_mask': np.zeros((2, 8, 10), np.float32)})
    b = r.get_variables(['x_sea_water_velocity', 'land_binary_mask'], t[1], np.array([4.2, 4.6]), np.array([53.1, 53.2]))
    assert isinstance(b['x_sea_water_velocity'], list) and len(b['x_sea_water_velocity']) == 3
    assert all(m.shape == b['land_binary_mask'].shape == (len(b['y']), len(b['x'])) for m in b['x_sea_water_velocity'])
    assert len(b['x']) < 10 and b['x_sea_water_velocity'][2][0, 0] == 2.0 + b['x'][0]


def test_tsprofiles_is_refused_loudly_and_leeway_seeding_keeps_the_random_stream():
    """vertical_mixing:TSprofiles = True: accepted by OceanDrift, whose update_terminal_velocity hook is empty and which does not
    require salinity (oceandrift.py:285-297, 461-462: no effect in the reference either); refused loudly by OpenOil, whose use of
    the T / S profiles is not built.  Leeway.seed_elements draws the
    downwind perturbations in batches that consume np.random exactly like the reference's element-by-element loop with
    rejection (leeway.py:331-339): same values, same generator state afterwards."""
    from opendrift_amd.oceandrift import OceanDrift
    from opendrift_amd.leeway import Leeway
    o = OceanDrift(loglevel=50)
    o.set_config('vertical_mixing:TSprofiles', False)
    o.set_config('vertical_mixing:TSprofiles', True)
    assert o.get_config('vertical_mixing:TSprofiles') is True
    from opendrift_amd.openoil import OpenOil
    oo = OpenOil(loglevel=50)
    oo.set_config('vertical_mixing:TSprofiles', False)
    with pytest.raises(NotImplementedError):
        oo.set_config('vertical_mixing:TSprofiles', True)
    c = dict(DWSLOPE=0.05, DWOFFSET=1.0, DWSTD=12.0, CWRSLOPE=0.5, CWROFFSET=1.0, CWRSTD=5.0, CWLSLOPE=-0.5, CWLOFFSET=-1.0, CWLSTD=5.0)
    n = 4001
    np.random.seed(7)
    want = np.zeros(n)
    for i in range(n):
        want[i] = np.random.randn(1)[0] * c['DWSTD']
        while c['DWSLOPE'] + want[i] / 20.0 < 0.0:
            want[i] = np.random.randn(1)[0] * c['DWSTD']
    rcw, tail = np.random.randn(n), np.random.randn(2)
    lw = Leeway(loglevel=50)
    np.random.seed(7)
    lw.seed_elements(lon=np.full(n, 4.0), lat=np.full(n, 60.0), time=datetime(2020, 1, 1), leeway_coefficients=c)
    assert np.array_equal(lw._sched['downwind_eps'], want.astype(np.float32))
    assert (want != np.random.RandomState(7).randn(n) * c['DWSTD']).any()          # some draws were rejected
    assert np.array_equal(np.random.randn(2), tail)


def test_leeway_object_classes_come_from_the_table_or_raise(objectprop_path, monkeypatch):
    """leeway.py:186-232,292-400: `Leeway(d=path)` reads the object-class table, `seed_elements(object_type=k)` and
    `seed:object_type` perturb that class's nine coefficients -- the element properties of golden C5 (written by the
    reference with object_type=1 behind np.random.seed(0)) bit for bit; without a table the request RAISES (round 5 accepted
    `object_type` and ignored it)."""
    from opendrift_amd.leeway import Leeway
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'c5_leeway_stere.npz'))
    keys = ('downwind_slope', 'crosswind_slope', 'downwind_offset', 'crosswind_offset', 'downwind_eps', 'crosswind_eps',
            'orientation')
    t0 = datetime(2020, 1, 1)
    for how in ('argument', 'config', 'environment'):
        if how == 'environment':
            monkeypatch.setenv('ODR_OBJECTPROP', objectprop_path)
            lw = Leeway(loglevel=50)
        else:
            lw = Leeway(objectprop_path, loglevel=50)
        assert sorted(lw.leewayprop) == [1, 2] and lw.leewayprop[2]['OBJKEY'] == 'PIW-2'
        assert lw.get_config('seed:object_type') == 'Person-in-water (PIW), unknown state (mean values)'
        np.random.seed(0)
        if how == 'config':
            lw.seed_elements(lon=g['lon'][0], lat=g['lat'][0], time=t0)            # the default class of seed:object_type
        else:
            lw.seed_elements(lon=g['lon'][0], lat=g['lat'][0], time=t0, object_type=1)
        for k in keys:
            assert np.array_equal(lw._sched[k], g['p_' + k].astype(np.float32)), (how, k)
    monkeypatch.delenv('ODR_OBJECTPROP')
    lw = Leeway(objectprop_path, loglevel=50)
    with pytest.raises(ValueError):
        lw.set_config('seed:object_type', 'no such object')
    lw.set_config('seed:object_type', '>PIW, vertical PFD type III conscious')
    np.random.seed(0)
    lw.seed_elements(lon=4.0, lat=60.0, time=t0, number=10)
    assert np.allclose(lw._sched['downwind_slope'], 0.48) and np.allclose(np.abs(lw._sched['crosswind_slope']), 0.15)
    with pytest.raises(KeyError):
        lw.seed_elements(lon=4.0, lat=60.0, time=t0, object_type=77)
    # the draws of the class come BEFORE the base class draws the seeding radius (leeway.py:327-346 before :386)
    a, b = Leeway(objectprop_path, loglevel=50), Leeway(objectprop_path, loglevel=50)
    monkeypatch.setattr(Leeway, '_geod_fwd', lambda self, lon, lat, az, dist: (lon + 1e-5 * dist, lat))   # (the device's geodesic)
    np.random.seed(3)
    a.seed_elements(lon=4.0, lat=60.0, time=t0, number=50, radius=1000.0, object_type=2)
    np.random.seed(3)
    b.seed_elements(lon=4.0, lat=60.0, time=t0, number=50, object_type=2)
    assert np.array_equal(a._sched['downwind_eps'], b._sched['downwind_eps']) and a._sched['lon'].std() > 0
    # no table: a class cannot be looked up -> raise; coefficients or explicit arrays still work
    monkeypatch.setattr('opendrift_amd.leeway.find_objectprop', lambda d=None: None)
    bare = Leeway(loglevel=50)
    assert bare.leewayprop is None
    with pytest.raises(FileNotFoundError):
        bare.seed_elements(lon=4.0, lat=60.0, time=t0, object_type=1)
    with pytest.raises(FileNotFoundError):
        bare.seed_elements(lon=4.0, lat=60.0, time=t0)
    bare.seed_elements(lon=4.0, lat=60.0, time=t0, downwind_slope=1.0, crosswind_slope=0.5)
    with pytest.raises(FileNotFoundError):
        Leeway('/no/such/OBJECTPROP.DAT', loglevel=50)


def test_static_variables_and_content_ids_without_a_gpu():
    """GridReader declares once which 2-D variables are the same array at every time level (what the device then gathers at one
    level: odr_block_set_content_ids); ContentIds assigns ids by bitwise comparison with the last array seen: equal -> same
    id, changed -> new id, 3-D variables and lists -> 0."""
    from opendrift_amd.device import ContentIds
    g = synthetic.grid3d(nx=20, ny=16, nz=4, nt=3, seed=1)
    names = ['x_sea_water_velocity', 'sea_floor_depth_below_sea_level', 'land_binary_mask']
    times = [datetime(2020, 1, 1) + timedelta(seconds=float(t)) for t in g['t']]
    r = readers.GridReader(g['x'], g['y'], times, {k: g[k] for k in names}, z=g['z'])
    assert sorted(r.static_variables) == ['land_binary_mask', 'sea_floor_depth_below_sea_level']
    moving = g['sea_floor_depth_below_sea_level'].copy()
    moving[2] += 1.0
    r2 = readers.GridReader(g['x'], g['y'], times, {'sea_floor_depth_below_sea_level': moving, 'land_binary_mask': g['land_binary_mask']})
    assert r2.static_variables == ['land_binary_mask']
    c = ContentIds()
    depth = g['sea_floor_depth_below_sea_level']
    a = c.assign(names, {'x_sea_water_velocity': g['x_sea_water_velocity'][0], 'sea_floor_depth_below_sea_level': depth[0],
                         'land_binary_mask': [g['land_binary_mask'][0]] * 2})
    assert a['x_sea_water_velocity'] == 0 and a['land_binary_mask'] == 0 and a['sea_floor_depth_below_sea_level'] > 0
    b = c.assign(names[1:2], {'sea_floor_depth_below_sea_level': depth[1].copy()})
    assert b['sea_floor_depth_below_sea_level'] == a['sea_floor_depth_below_sea_level']
    d = c.assign(names[1:2], {'sea_floor_depth_below_sea_level': depth[1] + np.float32(0.5)})
    assert d['sea_floor_depth_below_sea_level'] not in (0, a['sea_floor_depth_below_sea_level'])
    nanny = depth[0].copy()
    nanny[3, 4] = np.nan
    e = c.assign(names[1:2], {'sea_floor_depth_below_sea_level': nanny})
    f = c.assign(names[1:2], {'sea_floor_depth_below_sea_level': nanny.copy()})
    assert e['sea_floor_depth_below_sea_level'] == f['sea_floor_depth_below_sea_level'] != d['sea_floor_depth_below_sea_level']


def test_which_steps_need_all_rank_reductions_is_decided_on_the_host():
    """Sharded runs: the step's collective carries the 16 reduction slots -- and cannot be finished behind
    update() -- only when a mover of the step consults all-rank maxima.  A mover whose input no reader delivers and whose fallback
    is 0 returns early on every rank without looking (advect_wind: `wind_speed.max() == 0`, physics_methods.py:771-780; stokes_drift
    :799-804; horizontal_diffusion, basemodel/__init__.py:1754)."""
    class FakeBinding:
        sid = 0

    def model(**cfg):
        o = OceanDrift(loglevel=50)
        for k, v in cfg.items():
            o.set_config(k.replace('__', ':'), v)
        o.readers, o.priority_list = {}, {}
        return o

    o = model()
    assert o._calm_everywhere() and o._identically_zero('horizontal_diffusivity')
    assert not o._needs_reductions()                     # nothing but fallback zeros: no mover looks at the other ranks
    o = model(drift__relative_wind=True)
    assert not o._calm_everywhere() and o._needs_reductions()      # the wind relative to the current is not identically zero
    o = model(environment__fallback__x_wind=3.0)
    assert o._needs_reductions()
    o = model(environment__fallback__horizontal_diffusivity=10.0)
    assert o._needs_reductions()
    o = model()
    o.readers['w'] = FakeBinding()
    o.priority_list['y_wind'] = ['w']                    # a reader delivers the wind
    assert not o._calm_everywhere() and o._needs_reductions()
    o = model()
    o.priority_list['x_wind'] = ['gone']                 # listed, but discarded / not on the device: still identically zero
    assert o._calm_everywhere()
    o = model(vertical_mixing__diffusivitymodel='windspeed_Large1994', drift__vertical_mixing=True)
    assert o._needs_reductions()                         # the analytic profiles take the deepest mixed layer of ALL elements
