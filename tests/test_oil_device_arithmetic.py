"""CPU: the per-element arithmetic of the DEVICE code of the oil physics (opendrift_amd/csrc/odr_oil.hip.h, compiled
for the host by tests/oil_host.cpp with the HIP rounding intrinsics as IEEE single operations) against the NumPy
oracle and the reference's own numbers (tests/golden/c9_openoil_mixing.npz).  The float32 chains (sea water density
and viscosity, wave height, breaking fraction) must agree bit for bit; float64 results to the last bits (libm pow /
exp / log of glibc in both).  The kernels around this arithmetic (reductions, scan, the mixing loop) run only on the
GPU: tests/test_gpu_oil.py."""
import numpy as np
import pytest

import oil_host
import replay
from conftest import golden
from oracle import oil

CASES = [('johansen', 'Johansen et al. (2015)', 1), ('li', 'Li et al. (2017)', 2)]


def _state(g, tag, k):
    """environment and element state the reference had at the start of step k (k >= 1: float64 positions)"""
    B = replay.OracleBackend(replay.scenario_c9(g), g[tag + '_lon'][k], g[tag + '_lat'][k], g[tag + '_z'][k], wdf=0.0)
    B.sample([replay.XW, replay.YW, replay.TEMP, replay.SALT], k * float(g['dt']))
    return B.env


@pytest.mark.parametrize('tag,dist,code', CASES)
def test_device_arithmetic_equals_oracle_and_reference(tag, dist, code):
    g = golden('c9_openoil_mixing.npz')
    k = 2
    e = _state(g, tag, k)
    n = len(e[replay.XW])
    d_now = g[tag + '_diameter'][k].astype(np.float32)
    d_if = g[tag + '_diameter_if_entrained'][k + 1].astype(np.float32)
    rho, nu, sig = float(g['oil_density']), float(g['oil_viscosity']), float(g['interfacial_tension'])
    out = oil_host.elements(e[replay.XW], e[replay.YW], e[replay.TEMP], e[replay.SALT], d_now, rho, nu, g['film'], d_if,
                            sig, float(oil.RHO_W_DEFAULT), 60.0, code)
    # against the reference itself
    assert np.allclose(out['prob'], g[tag + '_probability'][k + 1], rtol=1e-12, atol=1e-16)
    # against the oracle
    T = e[replay.TEMP].copy()
    T[T < 100] += 273.15
    rho64, nu64 = np.full(n, rho), np.full(n, nu)
    assert np.allclose(out['w_now'], oil.terminal_velocity(d_now, rho64, T, e[replay.SALT]), rtol=1e-14, atol=0)
    assert np.allclose(out['w_if'], oil.terminal_velocity(d_if, rho64, T, e[replay.SALT]), rtol=1e-14, atol=0)
    T0 = T - 273.15
    nyw = oil.seawater_dynamic_viscosity(T0, e[replay.SALT]) / oil.sea_water_density(T0, e[replay.SALT])
    assert nyw.dtype == np.float32 and np.array_equal(out['nyw'], nyw)                        # float32 chain: bit for bit
    hs = oil.significant_wave_height(e[replay.XW], e[replay.YW])
    assert np.array_equal(out['zb'], 1.5 * hs)
    if code == 1:
        Sd = np.log(10) * 0.4
        want = oil.droplet_median_johansen2015(rho64[:1], nu64[:1], g['film'][:1], hs[:1], sig)   # one element: its own value
        assert abs(out['dv50'][0] / float(want) - 1) < 1e-14
        assert abs(out['dv50'].mean() / float(oil.droplet_median_johansen2015(rho64, nu64, g['film'], hs, sig)) - 1) < 1e-14
        assert Sd > 0
    else:
        assert abs(out['dv50'].mean() / float(oil.droplet_median_li2017(rho64, nu64, hs, sig)) - 1) < 1e-14


@pytest.mark.parametrize('tag,dist,code', CASES)
def test_device_spectrum_lookup_equals_np_random_choice(tag, dist, code):
    """k_oil_choice's table (blocked summation) and binary search against the reference's np.random.choice draws:
    the same grid point except where a uniform falls within the summation-order difference of a cell boundary."""
    g = golden('c9_openoil_mixing.npz')
    k = 2
    e = _state(g, tag, k)
    n = len(e[replay.XW])
    hs = oil.significant_wave_height(e[replay.XW], e[replay.YW])
    rho64, nu64 = np.full(n, float(g['oil_density'])), np.full(n, float(g['oil_viscosity']))
    sig = float(g['interfacial_tension'])
    dv50 = oil.droplet_median_johansen2015(rho64, nu64, g['film'], hs, sig) if code == 1 else \
        oil.droplet_median_li2017(rho64, nu64, hs, sig)
    d, idx = oil_host.choice(float(dv50), g[tag + '_u_diameter'][k])
    want = g[tag + '_idx_diameter'][k]
    assert np.abs(idx - want).max() <= 1 and (idx != want).mean() < 0.02
    assert np.abs(d - g[tag + '_diameter_if_entrained'][k + 1]).max() < 3.1e-9 + 1e-10     # float32 storage of <= 3 mm


def test_wave_modes_and_kelvin_switch():
    """wave height / period from the environment (modes 0), period from the wind in float64 (mode 1, a model without the
    variable) and without the Kelvin conversion of a backward run."""
    rng = np.random.default_rng(1)
    n = 64
    xw, yw = rng.uniform(-15, 15, n).astype(np.float32), rng.uniform(-15, 15, n).astype(np.float32)
    hs, tp = rng.uniform(0.5, 5, n).astype(np.float32), rng.uniform(3, 12, n).astype(np.float32)
    T, S = rng.uniform(275, 290, n).astype(np.float32), rng.uniform(20, 35, n).astype(np.float32)
    rho, nu = np.full(n, 950.0), np.full(n, float(np.float32(0.02)))
    out = oil_host.elements(xw, yw, T, S, 1e-4, 950.0, nu[0], 0.001, 2e-4, 0.025, float(oil.RHO_W_DEFAULT), 30.0, 1,
                            hs=hs, tp=tp, hs_mode=0, tp_mode=0, to_kelvin=True)
    wbf = 0.032 * (oil.wind_speed(xw, yw) - 5) / tp
    wbf[wbf < 0] = 0
    assert np.allclose(out['prob'], oil.entrainment_probability(rho, nu, 0.025, hs, wbf, 30.0), rtol=1e-12, atol=1e-16)
    assert np.allclose(out['w_now'], oil.terminal_velocity(np.full(n, 1e-4, np.float32), rho, T, S), rtol=1e-14, atol=0)
    out = oil_host.elements(xw, yw, T, S, 1e-4, 950.0, nu[0], 0.001, 2e-4, 0.025, float(oil.RHO_W_DEFAULT), 30.0, 1,
                            hs_mode=1, tp_mode=1)
    wbf = 0.032 * (oil.wind_speed(xw, yw) - 5) / oil.wave_period(xw, yw, in_environment=False)
    wbf[wbf < 0] = 0
    assert np.allclose(out['prob'], oil.entrainment_probability(rho, nu, 0.025, oil.significant_wave_height(xw, yw), wbf,
                                                                30.0), rtol=1e-12, atol=1e-16)
