"""CPU: host-side mirror of the reference interface (config validation, reader surface, seeding
rules) -- no device calls."""
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
    with pytest.raises(NotImplementedError):
        projection.parse_proj4('+proj=lcc +lat_1=49.5')
