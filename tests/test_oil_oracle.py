"""CPU: the oil-physics oracle (oracle/oil.py) against the reference's own OpenOil runs
(tests/golden/c9_openoil_mixing.npz, written by oracle/gen_golden_oil.py): terminal velocities, entrainment
probabilities, droplet spectra / np.random.choice, slick formation and wave entrainment inside the mixing loop.

Tolerances: the oracle follows NumPy's dtypes and operation order, so everything except libm-level differences is
identical: z to 1e-9 m from the reference's second state (1e-5 m from the seeding state: first-step float32
positions), diameters exactly (within one cell of the spectrum grid from the seeding state), probabilities to 1e-14."""
import numpy as np
import pytest

import replay
from conftest import golden

CASES = [('johansen', 'Johansen et al. (2015)'), ('li', 'Li et al. (2017)')]


def _backend(g, tag, start=0):
    B = replay.OracleBackend(replay.scenario_c9(g), g[tag + '_lon'][start], g[tag + '_lat'][start], g[tag + '_z'][start],
                             wdf=0.0)
    B.set_oil(g[tag + '_diameter'][start].astype(np.float32), float(g['oil_density']), float(g['oil_viscosity']), g['film'])
    return B


@pytest.mark.parametrize('start', [0, 1])
@pytest.mark.parametrize('tag,dist', CASES)
def test_c9_oracle_replays_the_references_openoil(tag, dist, start):
    g = golden('c9_openoil_mixing.npz')
    out = replay.replay_c9(_backend(g, tag, start), g, tag, 6, dist, start=start)
    tol_pos, tol_z = (1e-6, 1e-4) if start == 0 else (1e-9, 1e-8)   # start 0: first-step float32 positions
    for k, (lon, lat, z, status, oil) in enumerate(out, start):
        assert np.abs(lon - g[tag + '_lon'][k + 1]).max() < tol_pos and np.abs(lat - g[tag + '_lat'][k + 1]).max() < tol_pos
        assert np.abs(z - g[tag + '_z'][k + 1]).max() < tol_z, (k, np.abs(z - g[tag + '_z'][k + 1]).max())
        cell = 3.1e-9     # spectrum grid spacing (3e-3 - 1e-6) / 999999: a last-bit change of dV_50 (positions differ by 1e-10 deg) moves a draw by one cell
        assert np.abs(oil['diameter'] - g[tag + '_diameter'][k + 1].astype(np.float32)).max() <= cell
        assert np.abs(oil['diameter_if_entrained'] - g[tag + '_diameter_if_entrained'][k + 1]).max() <= cell
        assert np.allclose(oil['terminal_velocity'], g[tag + '_terminal_velocity'][k + 1], rtol=1e-12, atol=0)
    assert (out[-1][2] == 0).sum() > 100 and np.nanmin(out[-1][2]) < -20     # a slick and deep droplets


@pytest.mark.parametrize('tag,dist', CASES)
def test_c9_probability_and_mean_intrusion_depth(tag, dist):
    g = golden('c9_openoil_mixing.npz')
    B = _backend(g, tag, 1)           # from the reference's second state (float64 positions)
    out = replay.replay_c9(B, g, tag, 2, dist, start=1)
    assert np.allclose(B.probability, g[tag + '_probability'][2], rtol=1e-14, atol=0)
    m = g[tag + '_mean_zb'][1]
    assert float(B.mean_zb) == m[np.isfinite(m)][0]
    assert len(out) == 1


def test_np_random_choice_restatement():
    """np.random.choice(a, size, p=p) == a[cumsum(p)/sum .searchsorted(random_sample(size), 'right')]: the form the
    device kernel evaluates (numpy/random/mtrand.pyx, legacy RandomState.choice)."""
    from oracle import oil
    d, cdf = oil.droplet_spectrum_cdf(3.1e-4)
    pdf = np.diff(np.concatenate([[0.0], cdf]))
    np.random.seed(3)
    want = np.random.choice(d, size=1000, p=pdf / pdf.sum())
    np.random.seed(3)
    got = d[(np.cumsum(pdf / pdf.sum()) / np.cumsum(pdf / pdf.sum())[-1]).searchsorted(np.random.random(1000), side='right')]
    assert np.array_equal(want, got)
