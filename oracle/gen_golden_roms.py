"""Golden vectors for the sigma -> z regridding: runs the REFERENCE's own pure-NumPy module
/root/reference/opendrift/readers/roppy/depth.py (imported by file path) on seeded inputs and stores inputs
and outputs in tests/golden/roms_sigma2z.npz.  Run in the build container only (needs /root/reference)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('ref_depth', '/root/reference/opendrift/readers/roppy/depth.py')
depth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(depth)


def stretching(N, theta_s=6.0, theta_b=0.3):
    """Song & Haidvogel (1994) Cs_r at rho points (the usual content of a ROMS file's Cs_r)."""
    s = -1.0 + (np.arange(N) + 0.5) / N
    return (1 - theta_b) * np.sinh(theta_s * s) / np.sinh(theta_s) + theta_b * (np.tanh(theta_s * (s + 0.5)) / (2 * np.tanh(0.5 * theta_s)) - 0.5)


def main():
    rng = np.random.default_rng(0)
    N, ny, nx = 12, 18, 23
    H = rng.uniform(8.0, 900.0, (ny, nx))
    zeta = rng.uniform(-0.6, 0.9, (ny, nx))
    zeta[3, 4] = -(H[3, 4] + 1.0)          # a dry cell (surface below the bottom): z_rho > 0 -> NaN (:535-538)
    Cs = stretching(N)
    Hc = 20.0
    Z = np.array([0, -.5, -1, -3, -5, -10, -25, -50, -75, -100, -150, -200, -250, -300, -400, -500], float)
    out = dict(H=H, zeta=zeta, Cs=Cs, Hc=Hc, Z=Z)
    for vt in (1, 2):
        zr = depth.sdepth(H, zeta, Hc, Cs, Vtransform=vt, Vstretching=1)
        zr -= np.asarray(zeta)[np.newaxis]                     # reader_ROMS_native.py:518,530
        if (np.nanmax(zr) > 0).any():                          # :535-538
            zr[zr > 0] = np.nan
        F = (rng.standard_normal((N, ny, nx)) * 0.4).astype(np.float32)
        F[:, 5:8, 10:12] = np.nan                              # land-masked columns (:607-615)
        F[2, 0, 0] = np.float32(3e9)                           # a fill value that must not survive (:683-684)
        R, A, C, I, kmax = depth.multi_zslice(F.copy(), zr.copy(), Z)
        R[R > 1e+9] = np.nan
        out['zrho_vt%d' % vt] = zr
        out['F_vt%d' % vt] = F
        out['R_vt%d' % vt] = R
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'roms_sigma2z.npz'), **out)
    print('written', {k: getattr(v, 'shape', v) for k, v in out.items()})


if __name__ == '__main__':
    main()
