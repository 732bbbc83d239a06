"""GPU: vertical mixing with wind-parameterised diffusivity profiles (Large et al. 1994 -- also the default
'environment' model when no reader provides ocean_vertical_diffusivity -- and Sundby 1983; oceandrift.py:385-395,
425-458, physics_methods.py:203-250) against the reference's own runs (tests/golden/c7_wind_diffusivity.npz) and
the CPU oracle fed with the NumPy restatement of the profiles.

Tolerances: device vs oracle 1e-9 m on z (the only non-identical operation is sigma**3: libm pow in NumPy, an
error-compensated cube on the device); vs the reference's run 1e-9 m from its state after step 1, 1e-5 m from the
seeding state (first-step float32 positions)."""
from datetime import datetime, timedelta

import numpy as np
import pytest

import replay
from conftest import golden
from opendrift_amd import readers
from opendrift_amd.oceandrift import OceanDrift

pytestmark = pytest.mark.gpu
T0 = datetime(2020, 1, 1)
CASES = [('large', 'windspeed_Large1994'), ('sundby', 'windspeed_Sundby1983')]


def _sub(g, tag, start=0):
    sub = {k: g[tag + '_' + k][start:] for k in ('lon', 'lat', 'z', 'status')}
    sub['uniforms'] = g[tag + '_uniforms']
    return sub


def _device(ctx, g, sub):
    D = replay.DeviceBackend(replay.scenario_c7(g), ctx, sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=0.0)
    D.P.upload(terminal_velocity=g['tv'].astype(np.float32))
    return D


@pytest.mark.parametrize('tag,model', CASES)
def test_c7_device_vs_oracle_and_reference(ctx, tag, model):
    g = golden('c7_wind_diffusivity.npz')
    bg = float(g[tag + '_bg'])
    sub = _sub(g, tag)
    dev = replay.replay_c7(_device(ctx, g, sub), g, sub, model, bg, 6)
    replay.compare(dev, sub, 1e-8, 1e-5)
    O = replay.OracleBackend(replay.scenario_c7(g), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=0.0)
    O.tv = g['tv'].astype(np.float32)
    orc = replay.replay_c7(O, g, sub, model, bg, 6)
    for (lo1, la1, z1, s1), (lo2, la2, z2, s2) in zip(dev, orc):
        assert (s1 == s2).all()
        assert np.nanmax(np.abs(lo1 - lo2)) < 1e-10 and np.nanmax(np.abs(la1 - la2)) < 1e-10
        assert np.nanmax(np.abs(z1 - z2)) < 1e-9
    assert np.nanmin(dev[-1][2]) < -50 and (dev[-1][2] == 0).sum() >= 40     # deep mixing happened, slick stayed


@pytest.mark.parametrize('tag,model', CASES)
def test_c7_device_from_the_references_second_state(tag, model):
    from opendrift_amd.device import Context
    g = golden('c7_wind_diffusivity.npz')
    sub = _sub(g, tag, start=1)
    dev = replay.replay_c7(_device(Context(seed=0), g, sub), g, sub, model, float(g[tag + '_bg']), 6, start=1)
    worst = replay.compare(dev, sub, 1e-8, 1e-9)
    assert worst['z'] < 1e-9


def _reader(g):
    times = [T0 + timedelta(seconds=float(t)) for t in g['g_t']]
    names = ['x_wind', 'y_wind', 'ocean_mixed_layer_thickness', 'sea_floor_depth_below_sea_level',
             'x_sea_water_velocity', 'y_sea_water_velocity']
    return readers.GridReader(g['g_x'], g['g_y'], times, {k: g['g_' + k] for k in names})


@pytest.mark.parametrize('tag,model,config_model', [('large', 'windspeed_Large1994', 'environment'),
                                                    ('large', 'windspeed_Large1994', 'windspeed_Large1994'),
                                                    ('sundby', 'windspeed_Sundby1983', 'windspeed_Sundby1983')])
def test_c7_model_run(tag, model, config_model):
    """OceanDrift.run() with the reference's configuration calls (the default 'environment' model has no diffusivity
    reader here, so it takes the Large et al. profile like the reference does)."""
    g = golden('c7_wind_diffusivity.npz')
    n = g[tag + '_lon'].shape[1]
    o = OceanDrift(loglevel=50, seed=0, rng='numpy')
    o.add_reader(_reader(g))
    o.set_config('environment:fallback:land_binary_mask', 0)
    o.set_config('drift:advection_scheme', 'euler')
    o.set_config('drift:vertical_mixing', True)
    o.set_config('vertical_mixing:timestep', 60)
    o.set_config('vertical_mixing:diffusivitymodel', config_model)
    o.set_config('vertical_mixing:background_diffusivity', float(g[tag + '_bg']))
    o.set_config('drift:stokes_drift', False)
    np.random.seed(0)
    o.seed_elements(lon=g[tag + '_lon'][0], lat=g[tag + '_lat'][0], z=g[tag + '_z'][0], time=T0,
                    terminal_velocity=g['tv'], wind_drift_factor=0.0)
    o.run(time_step=900, steps=6)
    lon, lat, z = np.full(n, np.nan), np.full(n, np.nan), np.full(n, np.nan)
    for d in (o.elements, o.elements_deactivated):
        lon[d.ID], lat[d.ID], z[d.ID] = d.lon, d.lat, d.z
    assert np.abs(lon - g[tag + '_lon'][6]).max() < 1e-7 and np.abs(lat - g[tag + '_lat'][6]).max() < 1e-7
    assert np.abs(z - g[tag + '_z'][6]).max() < 1e-5


def test_unknown_model_raises_like_the_reference():
    g = golden('c7_wind_diffusivity.npz')
    o = OceanDrift(loglevel=50, seed=0)
    o.add_reader(_reader(g))
    o.set_config('environment:fallback:land_binary_mask', 0)
    o.set_config('drift:vertical_mixing', True)
    o.set_config('vertical_mixing:diffusivitymodel', 'stepfunction')
    o.seed_elements(lon=4.0, lat=60.0, z=-5.0, number=10, time=T0)
    with pytest.raises(ValueError, match='Unknown diffusivity model'):
        o.run(time_step=900, steps=1, stop_on_error=True)
