"""CPU: the landmask raster lookup and coastline_crossing oracle (oracle/landmask.py) against the reference's own
function and runs (tests/golden/c10_landmask_crossing.npz, written by oracle/gen_golden_landmask.py: the reference's
global-landmask Reader and coastline_crossing on a synthetic raster in place of the GSHHG data)."""
import numpy as np
import pytest

import replay
from conftest import golden
from oracle import landmask


@pytest.mark.parametrize('side', [True, False])
def test_coastline_crossing_equals_the_references_function(side):
    g = golden('c10_landmask_crossing.npz')
    m = landmask.RasterMask.from_golden(g)
    lc, la = landmask.coastline_crossing(m, g['fn_lon1'], g['fn_lat1'], g['fn_lon2'], g['fn_lat2'], float(g['precision']),
                                         land_side=side)
    assert np.array_equal(lc, g['fn_lon_c_%s' % side]) and np.array_equal(la, g['fn_lat_c_%s' % side])
    moved = (lc != (g['fn_lon2'] if side else g['fn_lon1']))
    assert 50 < moved.sum() < 500


@pytest.mark.parametrize('action', ['stranding', 'previous'])
def test_c10_oracle_replays_the_reference(action):
    g = golden('c10_landmask_crossing.npz')
    m = landmask.RasterMask.from_golden(g)
    B = replay.OracleBackend(replay.scenario_c10(g), g[action + '_lon'][0], g[action + '_lat'][0], g[action + '_z'][0],
                             wdf=0.0)
    out = replay.replay_c10(B, g, action, 14, m)
    for k, (lon, lat, z, status) in enumerate(out):
        tol = 1e-6 if k == 0 else 2e-7       # first-step float32 positions
        assert (status == g[action + '_status'][k + 1]).all(), k
        assert np.nanmax(np.abs(lon - g[action + '_lon'][k + 1])) < tol and np.nanmax(np.abs(lat - g[action + '_lat'][k + 1])) < tol
    if action == 'stranding':
        assert (out[-1][3] == 1).sum() > 100
