"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the sigma -> z regridding the ROMS-native reader applies to
every 3-D block (tests/, bench cpu_baseline only; the product path never imports oracle/).

Restates, column by column and in the same float64 operation order,
  * opendrift/readers/roppy/depth.py:31-113  (sdepth, stagger 'rho', Vtransform 1 and 2, default S = -1 + (k+0.5)/N)
  * opendrift/readers/reader_ROMS_native.py:512-538 (z_rho -= zeta; positive z_rho -> NaN)
  * opendrift/readers/roppy/depth.py:213-284 (multi_zslice: C = #(S < Z) clipped to [1, N-1],
    A = (Z - S[C-1]) / (S[C] - S[C-1]) clipped to [0, 1], R = (1 - A) F[C-1] + A F[C])
  * opendrift/readers/reader_ROMS_native.py:683-684 (R > 1e9 -> NaN).
Pinned against the reference's own depth.py run by oracle/gen_golden_roms.py (tests/golden/roms_sigma2z.npz).
"""
import numpy as np


def s_levels(N):
    return -1.0 + (np.arange(N, dtype=np.float64) + 0.5) / N


def z_rho(H, zeta, Hc, C, Vtransform=1, S=None):
    """Depth of the rho s-levels relative to the surface, [N, ny, nx] float64."""
    H = np.asarray(H, np.float64)
    zeta = np.zeros_like(H) if zeta is None else np.asarray(zeta, np.float64)
    C = np.asarray(C, np.float64)
    N = len(C)
    S = s_levels(N) if S is None else np.asarray(S, np.float64)
    out = np.empty((N,) + H.shape)
    for k in range(N):
        if Vtransform == 1:
            zo = Hc * (S[k] - C[k]) + C[k] * H
        elif Vtransform == 2:
            zo = (Hc * S[k] + C[k] * H) / (1.0 + Hc / H)
        else:
            raise ValueError('Unknown Vtransform')
        out[k] = (zo + zeta * (1 + zo / H)) - zeta
    with np.errstate(invalid='ignore'):
        if np.nanmax(out) > 0:
            out[out > 0] = np.nan
    return out


def zslice(F, zr, Z):
    """F [N, ny, nx] (float32 or float64) on s-levels with depths zr -> [len(Z), ny, nx] float64."""
    F, zr, Z = np.asarray(F), np.asarray(zr, np.float64), np.atleast_1d(np.asarray(Z, np.float64))
    N = F.shape[0]
    out = np.empty((len(Z),) + F.shape[1:])
    with np.errstate(invalid='ignore', divide='ignore'):
        for j, z in enumerate(Z):
            c = np.clip((zr < z).sum(axis=0), 1, N - 1)
            s0 = np.take_along_axis(zr, (c - 1)[None], 0)[0]
            s1 = np.take_along_axis(zr, c[None], 0)[0]
            f0 = np.take_along_axis(F, (c - 1)[None], 0)[0]
            f1 = np.take_along_axis(F, c[None], 0)[0]
            a = np.clip((z - s0) / (s1 - s0), 0.0, 1.0)
            out[j] = (1 - a) * f0 + a * f1
        out[out > 1e9] = np.nan
    return out
