"""oracle/roms.py (restatement of roppy sdepth + multi_zslice + the ROMS reader's NaN rules) against the golden
vectors produced by the reference's own depth.py (oracle/gen_golden_roms.py): bit for bit."""
import numpy as np

from conftest import golden
from oracle import roms


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_sdepth_and_zslice_match_reference_bit_for_bit():
    g = golden('roms_sigma2z.npz')
    for vt in (1, 2):
        zr = roms.z_rho(g['H'], g['zeta'], float(g['Hc']), g['Cs'], Vtransform=vt)
        assert zr.dtype == np.float64 and _same(zr, g['zrho_vt%d' % vt]), vt
        assert np.isnan(zr).any()                        # the wet/dry rule was exercised
        R = roms.zslice(g['F_vt%d' % vt], g['zrho_vt%d' % vt], g['Z'])
        assert R.dtype == np.float64 and _same(R, g['R_vt%d' % vt]), vt
        assert np.isnan(R[:, 5:8, 10:12]).all()
