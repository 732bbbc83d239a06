"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/c15_state_to_buffer.npz by EXECUTING THE REFERENCE'S OWN
`OpenDriftSimulation.state_to_buffer` (opendrift/models/basemodel/__init__.py:2384-2499) on a functional xarray stand-in
(oracle/xarray_standin.py; xarray itself is not installed here).  This pins SURVEY.md section 8 row f1: which elements
are written at which buffer time (every element present at an output time; only the deactivated ones, into the NEXT output
time, in between: `method='backfill'`), the float32 cast of float64 / int32 properties, min / max bookkeeping and the
buffer reset after `export_buffer_length` output times.

Scenario: the reference's OceanDrift on a small lon/lat grid with an eastward current towards a land strip (elements strand
at different calculation steps, also between output times), elements released over the first steps, time_step 600 s,
time_step_output 1200 s, export_buffer_length 3, 10 steps.  The loop body is driven by oracle/refdriver.py with the
reference's state_to_buffer() called where run() calls it (:2255, after interact_with_seafloor).  The result Dataset is
created as run() creates it (:2084-2134).

Stored: the per-call inputs (step, IDs present, status and the exported variables of those elements, float64 / int32 /
float32 as the reference holds them) and the outputs (every full buffer right before it is cleared, the last partial
buffer, min / max attributes).

    python oracle/gen_golden_history.py
"""
import os
import sys
from datetime import timedelta

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (installs the shim)
import xarray_standin as xs  # noqa: E402
from oracle.refdriver import RefStepper  # noqa: E402

EXPORT = ['z', 'age_seconds', 'x_sea_water_velocity', 'land_binary_mask']


def main():
    nx, ny, nt = 60, 40, 3
    x = np.linspace(4.0, 5.0, nx).astype(np.float32)
    y = np.linspace(60.0, 60.5, ny).astype(np.float32)
    X, Y = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    t = np.arange(nt) * 3600.0
    times = [gg.T0 + timedelta(seconds=float(v)) for v in t]
    g = dict(x=x, y=y, t=t)
    g['x_sea_water_velocity'] = np.stack([(0.8 + 0.1 * np.sin(3 * Y + k)) for k in range(nt)]).astype(np.float32)
    g['y_sea_water_velocity'] = np.stack([0.1 * np.sin(5 * X - k) for k in range(nt)]).astype(np.float32)
    land = np.zeros((ny, nx), np.float32)
    land[:, 46:] = 1.0
    g['land_binary_mask'] = np.stack([land] * nt)
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask']
    o = gg._base('euler')
    o.add_reader(gg.GridReader('+proj=latlong', x, y, times, {k: g[k] for k in names}))
    o.set_config('general:coastline_action', 'stranding')
    o.set_config('drift:stokes_drift', False)
    rng = np.random.default_rng(15)
    N0, N1 = 40, 20
    lon = np.concatenate([rng.uniform(4.40, 4.74, N0), rng.uniform(4.55, 4.74, N1)])
    lat = rng.uniform(60.1, 60.4, N0 + N1)
    tt = [gg.T0] * N0 + [gg.T0 + timedelta(seconds=float(s)) for s in rng.uniform(0, 1800, N1)]
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, z=-rng.uniform(0, 5, N0 + N1), time=tt, wind_drift_factor=0.0)
    N = N0 + N1
    steps, dt, out_every, nbuf = 10, 600.0, 2, 3
    st = RefStepper(o, dt, steps)
    o.time_step_output = timedelta(seconds=dt * out_every)
    o.export_buffer_length = nbuf
    o.outfile = None
    # the result Dataset as run() creates it (:2076-2134)
    export_variables = list(set(EXPORT + ['lon', 'lat', 'status']))
    coords = {'trajectory': ('trajectory', np.arange(N), {}),
              'time': ('time', pd.date_range(o.start_time, periods=nbuf, freq=o.time_step_output), {})}
    shape = (N, nbuf)
    dims = ('trajectory', 'time')
    element_vars = {vn: (dims, np.full(shape, np.nan, np.float32), {}) for vn in o.elements.variables
                    if vn != 'ID' and vn in export_variables}
    environment_vars = {vn: (dims, np.full(shape, np.nan, np.float32)) for vn in o.required_variables if vn in export_variables}
    o.result = xs.Dataset(coords=coords, data_vars=element_vars | environment_vars, attrs={})
    variables = list(o.result.data_vars)
    calls = []
    orig = o.interact_with_seafloor

    def hooked():                      # run() calls state_to_buffer right after interact_with_seafloor (:2253-2255)
        orig()
        e = o.elements
        rec = dict(step=o.steps_calculation, ID=np.array(e.ID, copy=True), status=np.array(e.status, copy=True))
        for v in variables:
            src = e if hasattr(e, v) else o.environment
            rec[v] = np.array(getattr(src, v), copy=True) * np.ones(len(e.ID), dtype=np.asarray(getattr(src, v)).dtype)
        calls.append(rec)
        type(o).state_to_buffer(o)     # THE REFERENCE'S OWN FUNCTION
    o.interact_with_seafloor = hooked
    for k in range(steps):
        st.step()
    # the last state (run() writes it with final=True after the loop; the same scatter at an output time)
    o.environment, o.environment_profiles, missing = o.env.get_environment(
        list(o.required_variables), o.time, o.elements.lon, o.elements.lat, o.elements.z, o.required_profiles, o.profiles_depth,
        element_ID=o.elements.ID)
    o.interact_with_seafloor = orig
    e = o.elements
    rec = dict(step=o.steps_calculation, ID=np.array(e.ID, copy=True), status=np.array(e.status, copy=True))
    for v in variables:
        src = e if hasattr(e, v) else o.environment
        rec[v] = np.array(getattr(src, v), copy=True) * np.ones(len(e.ID), dtype=np.asarray(getattr(src, v)).dtype)
    calls.append(rec)
    type(o).state_to_buffer(o)
    ds = o.result
    snaps = list(ds.snapshots)
    if not np.all([np.isnan(ds[v].values).all() for v in variables]):
        snaps.append((ds._time_index.copy(), {v: ds[v].values.copy() for v in variables}))
    out = dict(variables=np.array(variables), n=N, steps=steps, dt=dt, out_every=out_every, export_buffer_length=nbuf,
               n_calls=len(calls), n_buffers=len(snaps), categories=np.array(o.status_categories))
    for i, c in enumerate(calls):
        for k, v in c.items():
            out['call%d_%s' % (i, k)] = v
    for j, (tindex, vals) in enumerate(snaps):
        out['buf%d_time_s' % j] = np.array([(pd.Timestamp(q) - pd.Timestamp(o.start_time)).total_seconds() for q in tindex])
        for v in variables:
            out['buf%d_%s' % (j, v)] = vals[v]
    for v in variables:
        if v != 'status':
            out['minval_' + v] = np.float32(ds[v].attrs.get('minval', np.nan))
            out['maxval_' + v] = np.float32(ds[v].attrs.get('maxval', np.nan))
    print('calls', len(calls), 'buffers', len(snaps), 'deactivated at the end', int((calls[-1]['status'] != 0).sum()) + N - len(calls[-1]['ID']),
          'categories', o.status_categories)
    np.savez_compressed(os.path.join(gg.GOLD, 'c15_state_to_buffer.npz'), **{('g_' + k): v for k, v in g.items()}, **out)


if __name__ == '__main__':
    main()
