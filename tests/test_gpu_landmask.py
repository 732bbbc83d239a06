"""GPU: the landmask raster source (device image of reader_global_landmask.Reader) and coastline_crossing
(general:coastline_approximation_precision; basemodel/__init__.py:81-134, 694-746) against the reference's own
function and runs on a synthetic raster (tests/golden/c10_landmask_crossing.npz) and the NumPy oracle.
The crossing points are linspace samples: they must equal the reference's bit for bit."""
from datetime import datetime

import numpy as np
import pytest

import replay
from conftest import golden
from opendrift_amd import readers
from opendrift_amd.device import Context
from oracle import landmask

pytestmark = pytest.mark.gpu
T0 = datetime(2020, 1, 1)
LAND = 'land_binary_mask'


def test_raster_lookup_equals_oracle():
    g = golden('c10_landmask_crossing.npz')
    m = landmask.RasterMask.from_golden(g)
    rng = np.random.default_rng(0)
    n = 200000
    lon = rng.uniform(2.5, 7.5, n)
    lat = rng.uniform(58.5, 62.5, n)
    lon[:1000] += 360.0                  # modulate_longitude
    lon[1000:2000] -= 360.0
    ctx = Context(seed=0)
    sid = ctx.add_landmask(m.lon0, m.lat0, m.dlon, m.dlat, m.cells)
    ctx.bind(LAND, [sid], np.nan)
    P = ctx.particles(n)
    P.append(lon, lat)
    got = P.env_sample([LAND], 0.0, download=True)[LAND]
    want = m.land_binary_mask(lon, lat)
    assert np.array_equal(got, want) and 0.2 < want.mean() < 0.8
    P.close()
    ctx.close()


@pytest.mark.parametrize('side', [True, False])
def test_coastline_crossing_equals_the_references_function(side):
    g = golden('c10_landmask_crossing.npz')
    m = landmask.RasterMask.from_golden(g)
    n = len(g['fn_lon1'])
    ctx = Context(seed=0)
    sid = ctx.add_landmask(m.lon0, m.lat0, m.dlon, m.dlat, m.cells)
    P = ctx.particles(n)
    P.append(g['fn_lon1'], g['fn_lat1'])
    P.store_previous()
    P.upload(lon=g['fn_lon2'], lat=g['fn_lat2'])
    P.env_upload(LAND, np.ones(n, np.float32))
    hit = P.coastline_crossing('stranding' if side else 'previous', float(g['precision']), sid)
    d = P.download()
    assert hit == n
    assert np.array_equal(d['lon'], g['fn_lon_c_%s' % side]) and np.array_equal(d['lat'], g['fn_lat_c_%s' % side])
    assert (P.env_download(LAND) == 0).all()
    assert ((d['status'] != 0) == side).all()       # 'stranding' deactivates the elements at the surface
    P.close()
    ctx.close()


@pytest.mark.parametrize('action', ['stranding', 'previous'])
def test_c10_device_vs_oracle_and_reference(action):
    g = golden('c10_landmask_crossing.npz')
    m = landmask.RasterMask.from_golden(g)
    lon0, lat0, z0 = g[action + '_lon'][0], g[action + '_lat'][0], g[action + '_z'][0]
    D = replay.DeviceBackend(replay.scenario_c10(g, device_raster=m), Context(seed=0), lon0, lat0, z0, wdf=0.0)
    dev = replay.replay_c10(D, g, action, 14, m)
    O = replay.OracleBackend(replay.scenario_c10(g), lon0, lat0, z0, wdf=0.0)
    orc = replay.replay_c10(O, g, action, 14, m)
    for k, ((lo1, la1, z1, s1), (lo2, la2, z2, s2)) in enumerate(zip(dev, orc)):
        assert (s1 == s2).all() and (s1 == g[action + '_status'][k + 1]).all(), k
        assert np.nanmax(np.abs(lo1 - lo2)) < 1e-10 and np.nanmax(np.abs(la1 - la2)) < 1e-10
        tol = 1e-6 if k == 0 else 2e-7       # first-step float32 positions
        assert np.nanmax(np.abs(lo1 - g[action + '_lon'][k + 1])) < tol and np.nanmax(np.abs(la1 - g[action + '_lat'][k + 1])) < tol
    if action == 'stranding':
        assert (dev[-1][3] == 1).sum() > 100


@pytest.mark.parametrize('action', ['stranding', 'previous'])
def test_c10_model_run_with_auto_landmask(action):
    """OceanDrift.run() with general:use_auto_landmask and general:coastline_approximation_precision, the raster
    standing in for the GSHHG data."""
    from opendrift_amd.oceandrift import OceanDrift
    g = golden('c10_landmask_crossing.npz')
    m = landmask.RasterMask.from_golden(g)
    n = g[action + '_lon'].shape[1]
    o = OceanDrift(loglevel=50, seed=0)
    o.set_config('general:use_auto_landmask', True)
    o.set_config('general:coastline_action', action)
    o.set_config('general:coastline_approximation_precision', float(g['precision']))
    o.set_config('drift:advection_scheme', 'euler')
    o.set_config('drift:stokes_drift', False)
    o.set_config('drift:vertical_mixing', False)
    o.set_config('drift:vertical_advection', False)
    o.add_reader(readers.LandmaskRasterReader(m.lon0, m.lat0, m.dlon, m.dlat, m.cells))
    o.add_reader(readers.ConstantReader({'x_sea_water_velocity': float(g['u']), 'y_sea_water_velocity': float(g['v']),
                                         'x_wind': 0.0, 'y_wind': 0.0}))
    o.seed_elements(lon=g[action + '_lon'][0], lat=g[action + '_lat'][0], z=g[action + '_z'][0], time=T0, wind_drift_factor=0.0)
    o.run(time_step=900, steps=14)
    lon, lat, status = np.full(n, np.nan), np.full(n, np.nan), np.full(n, -1)
    for d in (o.elements, o.elements_deactivated):
        lon[d.ID], lat[d.ID], status[d.ID] = d.lon, d.lat, d.status
    # run() ends with interact_with_coastline(final=True) (basemodel/__init__.py:2310), which the step-by-step golden
    # does not contain: the expectation is the oracle replay of the golden (checked against it in
    # test_c10_device_vs_oracle_and_reference) followed by that one call
    O = replay.OracleBackend(replay.scenario_c10(g), g[action + '_lon'][0], g[action + '_lat'][0], g[action + '_z'][0], wdf=0.0)
    replay.replay_c10(O, g, action, 14, m)
    before = O.state(n)
    O.sample_landmask(m)
    moved_ids = O.ID[O.env[LAND] == 1]
    O.coast_crossing(action, float(g['precision']), m)
    lon_e, lat_e, _, st_e = O.state(n)
    assert len(moved_ids) > 3 and (np.abs(before[0] - lon_e)[moved_ids] > 0).any()
    assert (status == st_e).all()
    # a final position that differs by 1e-9 deg can put the crossing on the neighbouring 0.001 deg sample
    tol = np.full(n, 1e-6)
    tol[moved_ids] = 2e-3
    assert (np.abs(lon - lon_e) < tol).all() and (np.abs(lat - lat_e) < tol).all()
    assert (np.abs(lon - lon_e)[moved_ids] < 1e-6).mean() > 0.5


def test_precision_without_a_landmask_raster_raises():
    from opendrift_amd.oceandrift import OceanDrift
    o = OceanDrift(loglevel=50, seed=0)
    o.set_config('general:coastline_approximation_precision', 0.001)
    o.set_config('environment:constant:land_binary_mask', 1)
    o.add_reader(readers.ConstantReader({'x_sea_water_velocity': 0.1, 'y_sea_water_velocity': 0.0}))
    o.seed_elements(lon=4.0, lat=60.0, number=10, time=T0)
    with pytest.raises(NotImplementedError, match='landmask'):
        o.run(time_step=900, steps=1, stop_on_error=True)
