"""The CPU restatement of state_to_buffer (oracle/history.py) against hand-derived expectations from
opendrift/models/basemodel/__init__.py:2084-2105,2384-2414 (the reference's own buffer needs xarray)."""
import numpy as np

from oracle.history import HistoryOracle


def test_output_step_writes_everything_substep_only_deactivated():
    h = HistoryOracle(5, 3, ['lon', 'status', 'x_wind'])
    ID = np.array([0, 1, 2, 4])
    lon = np.array([4.1, 4.2 + 1e-9, 4.3, 4.5])                     # float64: 4.2+1e-9 is not a float32
    vals = dict(lon=lon, status=np.array([0, 0, 1, 0], np.int32), x_wind=np.array([1, 2, 3, 4], np.float32))
    h.record(0, ID, vals['status'], vals)                           # output time: all elements present
    assert h.buf['lon'].dtype == np.float32 and np.isnan(h.buf['lon'][3]).all()       # ID 3 not seeded yet
    assert h.buf['lon'][1, 0] == np.float32(4.2 + 1e-9) and h.buf['status'][2, 0] == 1.0
    vals2 = dict(lon=lon + 1, status=np.array([0, 2, 1, 0], np.int32), x_wind=vals['x_wind'] * 2)
    h.record(1, ID, vals2['status'], vals2, only_deactivated=True)  # sub-step: deactivated -> next output slot
    assert np.isnan(h.buf['lon'][[0, 4], 1]).all() and h.buf['status'][1, 1] == 2.0 and h.buf['x_wind'][2, 1] == 6.0
    h.record(1, ID[[0, 3]], np.zeros(2, np.int32), {k: v[[0, 3]] for k, v in vals2.items()})   # the output time itself
    assert h.buf['lon'][0, 1] == np.float32(5.1) and h.buf['status'][1, 1] == 2.0               # earlier write kept
    assert np.isnan(h.buf['lon'][:, 2]).all()
    lo, hi = h.minmax('lon')
    assert lo == float(np.float32(4.1)) and hi == float(np.float32(5.5))
    h.reset()
    assert all(np.isnan(a).all() for a in h.buf.values()) and np.isnan(h.minmax('lon')[0])


def test_nothing_to_write_is_a_no_op():
    h = HistoryOracle(3, 2, ['z'])
    h.record(0, np.arange(3), np.zeros(3, np.int32), dict(z=np.zeros(3)), only_deactivated=True)
    assert np.isnan(h.buf['z']).all()


def _replay_slots(g, record, flush):
    """The calls of run()'s _state_to_buffer (opendrift_amd/oceandrift.py) for the recorded states of golden c15: at an
    output step every element present goes to that output time, in between only the deactivated ones go to the NEXT one;
    after export_buffer_length output times the buffer is handed over (flush) and cleared."""
    out_every, nbuf = int(g['out_every']), int(g['export_buffer_length'])
    base = 0
    for i in range(int(g['n_calls'])):
        step = int(g['call%d_step' % i])
        k = step // out_every
        if step % out_every == 0:
            record(i, k - base, False)
            if k - base == nbuf - 1:
                flush()
                base += nbuf
        elif k + 1 - base < nbuf:
            record(i, k + 1 - base, True)


def test_c15_restatement_equals_the_references_own_state_to_buffer():
    """Golden c15 = the reference's OWN state_to_buffer executed on a functional xarray stand-in
    (oracle/gen_golden_history.py): per call the elements present, per buffer the float32 [trajectory, time] arrays right
    before the reference clears them.  The NumPy restatement (which the device buffer is compared with bit for bit,
    tests/test_gpu_history.py) reproduces every buffer exactly, NaN pattern included: row f1 is pinned by reference
    execution."""
    from conftest import golden
    g = golden('c15_state_to_buffer.npz')
    variables = [str(v) for v in g['variables']]
    n, nbuf = int(g['n']), int(g['export_buffer_length'])
    H = HistoryOracle(n, nbuf, variables)
    bufs = []

    def record(i, slot, only_deactivated):
        H.record(slot, g['call%d_ID' % i], g['call%d_status' % i], {v: g['call%d_%s' % (i, v)] for v in variables}, only_deactivated)

    def flush():
        bufs.append({v: H.buf[v].copy() for v in variables})
        H.reset()
    _replay_slots(g, record, flush)
    assert len(bufs) == int(g['n_buffers']) == 2
    for j, b in enumerate(bufs):
        for v in variables:
            assert np.array_equal(b[v], g['buf%d_%s' % (j, v)], equal_nan=True), (j, v)
    # the scenario exercises what it should: deactivations written between output times, late releases, float32 casts
    st = g['buf0_status']
    assert (st[:, 1:] > 0).any() and np.isnan(g['buf0_lon'][:, 0]).any() and not np.isnan(g['buf1_lon'][:40, 0]).all()
    assert g['buf0_lon'].dtype == np.float32 and (g['call0_lon'].dtype == np.float64 or g['call0_lon'].dtype == np.float32)
