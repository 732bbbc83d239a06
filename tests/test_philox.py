"""CPU: oracle/philox.py against the known-answer vectors of Philox4x32-10 published with the Random123 library
(kat_vectors: zero, all-ones and the digits of pi as counter / key), and the layout of the mixing uniforms."""
import numpy as np

from oracle import philox


def test_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox.philox4x32_10(ctr, key)
        assert tuple(int(g[0]) for g in got) == want
    # the same three counters / keys under Philox4x32-7 (kat_vectors of the same library)
    kat7 = [(0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48), (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662),
            (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a)]
    for (ctr, key, _), want in zip(kat, kat7):
        got = philox.philox4x32(ctr, key, 7)
        assert tuple(int(g[0]) for g in got) == want


def test_library_names_its_round_count():
    import __graft_entry__ as g
    g.build()
    assert philox.library_rounds() in (7, 10)


def test_mixing_uniforms_layout():
    ids = np.arange(1000, dtype=np.int32)
    u = philox.mixing_uniforms(1234567890123, ids, 7, 12)
    assert u.shape == (12, 1000) and (u > 0).all() and (u < 1).all()
    # symmetric about 1/2 on the 2^-24 lattice
    x = u / 5.9604644775390625e-08 - 0.5
    assert np.array_equal(x, np.round(x)) and x.max() < 2 ** 24
    # sub-steps 0..4 share block 0, 5..9 block 1; different steps / seeds / elements give different numbers
    q = philox.philox4x32((0, 7, ids.astype(np.uint32), philox.MIX_TAG), (1234567890123 & 0xFFFFFFFF, 1234567890123 >> 32), philox.library_rounds())
    assert np.array_equal(x[1], (q[1] >> np.uint32(8)).astype(np.float64))
    assert not np.array_equal(u, philox.mixing_uniforms(1234567890123, ids, 8, 12))
    assert not np.array_equal(u[:, 1:], u[:, :-1])
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    # the five draws of a block are uncorrelated
    c = np.corrcoef(u[:5])
    assert np.abs(c - np.eye(5)).max() < 0.15
