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
