"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's result buffer for the device
history (tests/, bench cpu_baseline only; the product path never imports oracle/).

Follows opendrift/models/basemodel/__init__.py:
  * buffer creation  :2084-2105 -- every exported variable is a float32 [trajectory, time] array filled
    with NaN (default_dtype = np.float32, "Allows NaN (in contrast to np.int)");
  * state_to_buffer  :2384-2403 -- at an output time every element present is written at
    (trajectory = ID, time); otherwise only the elements with status != 0, into the next output
    time (`method='backfill'`); the assignment casts float64 / int32 properties to float32;
  * min / max        :2409-2414 -- var.min(skipna=True), var.max(skipna=True);
  * new buffer       :2493-2499 -- all variables reset to NaN.
Pinned by reference execution: oracle/gen_golden_history.py runs the reference's OWN state_to_buffer on a functional
stand-in for the xarray calls it makes (oracle/xarray_standin.py) and stores every full buffer
(tests/golden/c15_state_to_buffer.npz); tests/test_history_oracle.py replays the recorded states through this
restatement and obtains the same float32 arrays, NaN pattern included.
"""
import numpy as np


class HistoryOracle:
    def __init__(self, n_trajectories, n_times, variables):
        self.variables = list(variables)
        self.buf = {v: np.full((n_trajectories, n_times), np.nan, dtype=np.float32) for v in self.variables}

    def record(self, time_index, ID, status, values, only_deactivated=False):
        """values: dict variable -> array over the elements present (any dtype)."""
        ID = np.asarray(ID)
        sel = np.asarray(status) != 0 if only_deactivated else np.ones(len(ID), bool)
        if not sel.any():
            return
        for v in self.variables:
            self.buf[v][ID[sel], time_index] = np.asarray(values[v])[sel]   # NumPy casts to float32 on assignment

    def minmax(self, variable):
        a = self.buf[variable]
        if np.isnan(a).all():
            return np.nan, np.nan
        return float(np.nanmin(a)), float(np.nanmax(a))

    def reset(self):
        for a in self.buf.values():
            a[:] = np.nan
