"""TEST INFRASTRUCTURE ONLY -- a functional stand-in for the handful of xarray calls that the reference's
`OpenDriftSimulation.state_to_buffer` makes (opendrift/models/basemodel/__init__.py:2384-2499), so that the REFERENCE'S OWN
function can be executed here (xarray is not installed) and pin the result-buffer semantics of SURVEY.md section 8 row f1:

    Dataset(coords=, data_vars=, attrs=)         ds.data_vars (iteration, .items())      ds[name], ds.<name>
    ds.time: `t in ds.time`, ds.time[-1], ds.time.sel(time=t, method='backfill').values, == -> .values
    ds[name].loc[{'time': t, 'trajectory': ids}] = values           (label-based scatter, values cast to the array dtype)
    var.min(skipna=True).item() / .max(...)      var.attrs, var.<attr>      var.assign_attrs({...})      ds[name] = var
    ds[name].dims, ds[name][:] = nan             ds.coords['time'] (+ Timedelta, assignment)      ds.sizes

Semantics follow xarray's documented behaviour for exactly these calls (label lookup on a unique DatetimeIndex /
integer index, 'backfill' = the first label >= the key, NaN-skipping reductions); nothing else is implemented.
`snapshots` collects a copy of every variable right before the buffer is cleared ("Initialising new buffer"), which is
how oracle/gen_golden_history.py sees each full buffer.
"""
import numpy as np
import pandas as pd


class _Scalar:
    def __init__(self, v):
        self.values = v

    def item(self):
        return self.values.item() if hasattr(self.values, 'item') else self.values


class _Loc:
    def __init__(self, var):
        self.var = var

    def __setitem__(self, key, value):
        ds = self.var._ds
        ti = ds._time_index.get_loc(pd.Timestamp(key['time']))
        tr = np.asarray(key['trajectory'])
        rows = ds._traj_index.get_indexer(tr)
        assert (rows >= 0).all(), 'unknown trajectory label'
        self.var.values[rows, ti] = value          # NumPy casts float64 / int32 to the array dtype (float32)


class DataArray:
    def __init__(self, ds, name, dims, values, attrs=None):
        self._ds, self.name, self.dims, self.values, self.attrs = ds, name, tuple(dims), values, dict(attrs or {})

    @property
    def loc(self):
        return _Loc(self)

    def __getattr__(self, k):
        if k.startswith('_') or k not in self.__dict__.get('attrs', {}):
            raise AttributeError(k)
        return self.attrs[k]

    def min(self, skipna=True):
        with np.errstate(all='ignore'):
            return _Scalar(np.float32(np.nan) if np.isnan(self.values).all() else np.nanmin(self.values))

    def max(self, skipna=True):
        with np.errstate(all='ignore'):
            return _Scalar(np.float32(np.nan) if np.isnan(self.values).all() else np.nanmax(self.values))

    def assign_attrs(self, d):
        return DataArray(self._ds, self.name, self.dims, self.values, {**self.attrs, **d})

    def __setitem__(self, key, value):
        if key == slice(None) and self._ds is not None:
            self._ds._snapshot(self.name, self.values)
        self.values[key] = value


class _TimeCoord:
    def __init__(self, index):
        self.index = pd.DatetimeIndex(index)

    @property
    def values(self):
        return self.index.values

    def __contains__(self, t):
        return pd.Timestamp(t) in self.index

    def __getitem__(self, k):
        return _TimeLabel(self.index[k])

    def __len__(self):
        return len(self.index)

    def __add__(self, delta):
        return _TimeCoord(self.index + delta)

    def sel(self, time=None, method=None):
        assert method == 'backfill'
        pos = self.index.searchsorted(pd.Timestamp(time), side='left')      # first label >= time
        if pos >= len(self.index):
            raise KeyError(time)
        return _Scalar(self.index[pos].to_datetime64())


class _TimeLabel:
    def __init__(self, ts):
        self.ts = pd.Timestamp(ts)

    def __eq__(self, other):
        return _Scalar(np.bool_(pd.Timestamp(getattr(other, 'ts', other)) == self.ts))

    __req__ = __eq__

    def __hash__(self):
        return hash(self.ts)


class _Coords:
    def __init__(self, ds):
        self._ds = ds

    def __getitem__(self, k):
        assert k == 'time'
        return self._ds.time

    def __setitem__(self, k, v):
        assert k == 'time'
        self._ds._set_time(v.index if isinstance(v, _TimeCoord) else v)


class Dataset:
    def __init__(self, coords=None, data_vars=None, attrs=None):
        self.attrs = dict(attrs or {})
        self._traj_index = pd.Index(np.asarray(coords['trajectory'][1]))
        self._set_time(coords['time'][1])
        self._vars = {}
        for name, spec in (data_vars or {}).items():
            dims, values = spec[0], spec[1]
            self._vars[name] = DataArray(self, name, dims, values, spec[2] if len(spec) > 2 else None)
        self.snapshots = []
        self._pending = {}

    def _set_time(self, index):
        self._time_index = pd.DatetimeIndex(index)
        self.time = _TimeCoord(self._time_index)

    def _snapshot(self, name, values):
        self._pending[name] = values.copy()
        if len(self._pending) == len(self._vars):
            self.snapshots.append((self._time_index.copy(), self._pending))
            self._pending = {}

    @property
    def data_vars(self):
        return self._vars

    @property
    def coords(self):
        return _Coords(self)

    @property
    def sizes(self):
        return {'trajectory': len(self._traj_index), 'time': len(self._time_index)}

    def __getitem__(self, name):
        return self._vars[name]

    def __setitem__(self, name, var):
        self._vars[name] = var

    def __getattr__(self, k):
        if k.startswith('_') or k not in self.__dict__.get('_vars', {}):
            raise AttributeError(k)
        return self._vars[k]
