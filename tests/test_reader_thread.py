"""CPU: ReadAhead (opendrift_amd/readers.py) -- rank 0 of a sharded run reads the NEXT reader time level on a worker thread
while the steps of the current one run, so that the rank that owns the host Reader (and, one collective per step, every
other rank) does not stop for file I/O.  The block handed out is the reader's own either way."""
import threading
import time

import numpy as np
import pytest

from opendrift_amd.readers import ReadAhead


class SlowReader:
    """get_variables takes 0.15 s (a file read) and records the thread it ran on."""

    def __init__(self, nt=6, fail_at=None):
        self.times = list(range(nt))
        self.calls = []
        self.fail_at = fail_at

    def get_variables(self, variables, t, x, y, z):
        self.calls.append((t, threading.current_thread().name))
        time.sleep(0.15)
        if t == self.fail_at:
            raise OSError('level %d is unreadable' % t)
        return {'time': t, 'x': x, 'y': y, **{v: np.full((2, 2), float(t)) for v in variables}}


def test_next_level_is_read_off_the_calling_thread_and_is_ready_when_due():
    r = SlowReader()
    a = ReadAhead(r, ['u'])
    x, y = np.array([0.0, 1.0]), np.array([2.0, 3.0])
    b0 = a.read(0, x, y)                       # nothing read ahead yet: inline
    assert b0['u'][0, 0] == 0.0 and a.misses == 1 and a.hits == 0
    t0 = time.perf_counter()
    a.start(1, x, y)
    assert time.perf_counter() - t0 < 0.05     # returns at once
    time.sleep(0.3)                            # "the steps of this period"
    t0 = time.perf_counter()
    b1 = a.read(1, x, y)
    assert time.perf_counter() - t0 < 0.05     # no wait: it was read meanwhile
    assert b1['u'][0, 0] == 1.0 and a.hits == 1
    assert r.calls[0][1] == threading.current_thread().name and r.calls[1][1].startswith('odr-reader')
    assert a.worker_s >= 0.14
    a.close()


def test_a_read_ahead_for_another_level_or_window_is_dropped():
    r = SlowReader()
    a = ReadAhead(r, ['u'])
    x, y = np.array([0.0, 1.0]), np.array([2.0, 3.0])
    a.start(1, x, y)
    b = a.read(3, x, y)                        # the run jumped: level 3 is due
    assert b['u'][0, 0] == 3.0 and a.hits == 0 and a.misses == 1
    a.start(4, x, y)
    b = a.read(4, x, y + 1.0)                  # re-cut window
    assert b['u'][0, 0] == 4.0 and np.array_equal(b['y'], y + 1.0) and a.hits == 0 and a.misses == 2
    # never two reads at a time: the dropped one was waited for first
    assert [c[0] for c in r.calls] == [1, 3, 4, 4]
    a.close()


def test_the_readers_exception_surfaces_where_the_inline_read_would_have_raised_it():
    r = SlowReader(fail_at=2)
    a = ReadAhead(r, ['u'])
    a.start(2, None, None)
    time.sleep(0.2)
    with pytest.raises(OSError, match='unreadable'):
        a.read(2, None, None)
    assert a.read(3, None, None)['u'][0, 0] == 3.0      # and the next level is fine
    a.close()


def test_close_waits_for_the_read_in_flight_and_keeps_its_exception_to_itself():
    """run() closes the read-ahead of every binding before it returns (the caller may close the reader's file then); the
    read of a level nobody will ask for -- even a failing one -- must not raise there, and the worker must be gone."""
    r = SlowReader(fail_at=3)
    a = ReadAhead(r, ['u'])
    x, y = np.array([0.0]), np.array([0.0])
    a.start(3, x, y)
    t0 = time.perf_counter()
    a.close()
    assert time.perf_counter() - t0 > 0.1          # waited for get_variables to return
    assert [c[0] for c in r.calls] == [3] and a._pool is None and a._fut is None
    assert not [t for t in threading.enumerate() if t.name.startswith('odr-reader')]
    b = a.read(1, x, y)                            # a later read works (inline; a new worker starts with the next start())
    assert b['u'][0, 0] == 1.0
