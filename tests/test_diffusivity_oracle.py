"""CPU: wind-parameterised diffusivity profiles -- the NumPy restatement (oracle/diffusivity.py) against the
reference's own functions (golden c7), and the oracle's vertical_mixing driven by these profiles against the
reference's runs (default 'environment' model without a diffusivity reader -> Large et al. 1994; Sundby 1983)."""
import numpy as np
import pytest

from conftest import golden
from oracle import diffusivity as dif
from replay import OracleBackend, replay_c7, scenario_c7, compare


def test_functions_equal_the_reference_bit_for_bit():
    g = golden('c7_wind_diffusivity.npz')
    wind, depth = np.meshgrid(np.arange(0, 20, 5), np.arange(0, 80, 5))
    assert np.array_equal(dif.large1994(wind, depth), g['kat_large'])
    assert np.array_equal(dif.sundby1983(wind, depth), g['kat_sundby'])
    # the reference's known answers (tests/models/test_physics.py:50-60)
    assert abs(g['kat_large'].max() - 0.2017) < 1e-3 and abs(g['kat_sundby'].max() - 0.0585) < 1e-3
    assert g['kat_large'].min() == 0 and g['kat_sundby'].min() == 0
    W, D = np.meshgrid(g['fn_wind'], g['fn_depths'])
    assert np.array_equal(dif.large1994(W, D, g['fn_mld'], 1e-4), g['fn_large'])
    assert np.array_equal(dif.sundby1983(W, D, g['fn_mld'], 1e-4), g['fn_sundby'])


def _sub(g, tag, start=0):
    sub = {k: g[tag + '_' + k][start:] for k in ('lon', 'lat', 'z', 'status')}
    sub['uniforms'] = g[tag + '_uniforms']
    return sub


@pytest.mark.parametrize('tag,model', [('large', 'windspeed_Large1994'), ('sundby', 'windspeed_Sundby1983')])
def test_oracle_mixing_with_analytic_profiles_reproduces_reference_run(tag, model):
    """From the SEEDING state: the first step reproduces the reference bit for bit (z after steps 1 and 2 identical) -- during the
    first get_environment of a run the reference's positions are float32 arrays (elements.py:71-88): the longitude modulation
    (variables.py:259-280) and, for this geographic reader with float32 coordinate arrays, the index maps of the interpolators
    (interpolators.py:32-37,110-111) are float32 arithmetic, which orc_set_position_class + orc_source.xy_f32 restate (round 5:
    1e-5 m from the seeding state, "Deviation 2").  Later steps: one element in the Sundby run 1.1e-6 m (the float32 arctan2 of
    the golden-writing host).  Steps 2..6 from the reference's state after step 1: z to 1e-9 m."""
    g = golden('c7_wind_diffusivity.npz')
    bg = float(g[tag + '_bg'])
    sub = _sub(g, tag)
    B = OracleBackend(scenario_c7(g), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=0.0)
    B.tv = g['tv'].astype(np.float32)
    states = replay_c7(B, g, sub, model, bg, 6)
    assert np.array_equal(states[0][2], sub['z'][1]) and np.array_equal(states[1][2], sub['z'][2])      # bit for bit
    worst = compare(states, sub, 1e-8, 2e-6)
    sub = _sub(g, tag, start=1)
    B = OracleBackend(scenario_c7(g), sub['lon'][0], sub['lat'][0], sub['z'][0], wdf=0.0)
    B.tv = g['tv'].astype(np.float32)
    worst = compare(replay_c7(B, g, sub, model, bg, 6, start=1), sub, 1e-8, 1e-9)
    assert worst['z'] < 1e-9
