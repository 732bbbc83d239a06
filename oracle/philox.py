"""TEST INFRASTRUCTURE (checker only: tests/ and __graft_entry__.smoke() may import this; the product never does).

Philox4x32-10 of Salmon, Moraes, Dror and Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11), restated in NumPy
from the paper's round function, and the uniforms libodrift_hip.so draws from it for OceanDrift.vertical_mixing in
ODR_RNG_DEVICE mode (csrc/odr_kernels.hip.h: philox4x32_10 / mix_block / mix_uniform).  The reference has no counterpart
-- its random walk calls np.random.uniform (opendrift/models/oceandrift.py:531), which ODR_RNG_HOST mode reproduces
by taking the caller's draws; this file pins what the device's own stream is, so that a device-mode run can be replayed
on the host bit for bit (tests/test_gpu_vmix_window.py::test_device_stream_*).  Pinned against the known-answer vectors
published with the Random123 library (tests/test_philox.py)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
SH = np.uint64(32)


def philox4x32(counter, key, rounds=10):
    """counter: 4 arrays (or scalars) of 32-bit words, key: 2 words -> 4 uint32 arrays (one block per element).
    `rounds`: 10 is the Random123 default; 7 is the smallest count the paper reports as Crush-resistant."""
    c = [np.atleast_1d(np.asarray(x, dtype=np.uint64)) & MASK for x in counter]
    c = list(np.broadcast_arrays(*c))
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(rounds):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        c = [((p1 >> SH) ^ c[1] ^ np.uint64(k0)) & MASK, p1 & MASK, ((p0 >> SH) ^ c[3] ^ np.uint64(k1)) & MASK, p0 & MASK]
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return [x.astype(np.uint32) for x in c]


def philox4x32_10(counter, key):
    return philox4x32(counter, key, 10)


MIX_TAG = 0x4D495856


def library_rounds():
    """Round count of the mixing stream of the built libodrift_hip.so (csrc/odr_kernels.hip.h: ODR_MIX_ROUNDS), read from
    odr_version() -- the checker follows the library it checks, the known answers of tests/test_philox.py pin both counts."""
    import re
    from opendrift_amd import _abi
    m = re.search(rb'Philox4x32-(\d+)', _abi.load().odr_version())
    return int(m.group(1))


def mixing_uniforms(seed, ids, step, ntimes, rounds=None):
    """[ntimes][n] float64 uniforms of the mixing sub-steps of elements `ids` in step `step` (mix_uniform): block b = it // 5
    from counter {b, step low, id, MIX_TAG ^ step high} and key = seed; sub-step it % 5 takes the upper 24 bits of word
    0..3, the fifth the low bytes of words 0..2; u = (x + 1/2) 2^-24."""
    ids = np.asarray(ids).astype(np.uint32)
    rounds = library_rounds() if rounds is None else rounds
    out = np.empty((ntimes, len(ids)))
    q = None
    for it in range(ntimes):
        b, k = divmod(it, 5)
        if k == 0:
            q = philox4x32((b, step & 0xFFFFFFFF, ids, MIX_TAG ^ (step >> 32)), (seed & 0xFFFFFFFF, seed >> 32), rounds)
        if k < 4:
            x = q[k] >> np.uint32(8)
        else:
            x = ((q[0] & np.uint32(255)) << np.uint32(16)) | ((q[1] & np.uint32(255)) << np.uint32(8)) | (q[2] & np.uint32(255))
        out[it] = (x.astype(np.float64) + 0.5) * 5.9604644775390625e-08
    return out
