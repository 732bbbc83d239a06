#!/usr/bin/env python
"""Measurement of the output path: odr_history_record / odr_history_flush on MI355X
beside the CPU restatement of state_to_buffer (oracle/history.py, 1 core).

    python tools/bench_history.py [--particles N] [--times T] [--reps R]

One JSON line: element-records/s of the record kernel (an element-record = all exported variables of one
element at one output time), its HBM roofline fraction, the flush rate (device -> pinned host, PCIe bound)
and the CPU baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
U, V, XW, YW = 'x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind'
VARS = ['lon', 'lat', 'z', 'status', 'moving', 'age_seconds', U, V, XW, YW]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--particles', type=int, default=10_000_000)
    ap.add_argument('--times', type=int, default=4)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--cpu-particles', type=int, default=2_000_000)
    a = ap.parse_args()
    import __graft_entry__ as G
    G.build()
    from opendrift_amd.device import Context
    ctx = Context(device=0, seed=0)
    n, nt = a.particles, a.times
    rng = np.random.default_rng(0)
    P = ctx.particles(n)
    ids = rng.permutation(n).astype(np.int32)            # device order is not ID order (sorting, compaction)
    P.append(rng.uniform(0, 10, n), rng.uniform(60, 66, n), z=-rng.uniform(0, 50, n), id=ids)
    for v in (U, V, XW, YW):
        P.env_upload(v, rng.standard_normal(n).astype(np.float32))
    H = ctx.history(n, nt, VARS)
    H.record(P, 0)
    ctx.sync()
    ctx.timer_begin()
    for r in range(a.reps):
        H.record(P, r % nt)
    ms = ctx.timer_end() / a.reps
    # algorithmic bytes per element-record: id 4 + status 4 (read for the filter) + the exported variables once in
    # (3 x 8 + 7 x 4 = 52, status shared) + one float32 record out (stride 12 floats = 48)
    bytes_rec = 4 + 52 + 48
    H.flush(0, nt)                                       # first flush allocates the pinned host buffer
    H.wait()
    t0 = time.perf_counter()
    H.flush(0, nt)
    t_launch = time.perf_counter() - t0
    H.wait()
    t_flush = time.perf_counter() - t0
    flush_bytes = len(VARS) * n * nt * 4
    # CPU baseline: the numpy restatement on a bounded sample (1 core)
    from oracle.history import HistoryOracle
    m = a.cpu_particles
    O = HistoryOracle(m, nt, VARS)
    idc = rng.permutation(m)
    vals = {'lon': rng.uniform(0, 10, m), 'lat': rng.uniform(60, 66, m), 'z': -rng.uniform(0, 50, m),
            'status': np.zeros(m, np.int32), 'moving': np.ones(m, np.int32), 'age_seconds': np.zeros(m, np.float32),
            U: np.zeros(m, np.float32), V: np.zeros(m, np.float32), XW: np.zeros(m, np.float32), YW: np.zeros(m, np.float32)}
    O.record(0, idc, vals['status'], vals)
    t0 = time.perf_counter()
    reps_cpu = 0
    while time.perf_counter() - t0 < 10.0 and reps_cpu < 20:
        O.record(reps_cpu % nt, idc, vals['status'], vals)
        reps_cpu += 1
    cpu = m * reps_cpu / (time.perf_counter() - t0)
    out = {
        'metric': 'result-buffer element-records/s (state_to_buffer, %d variables)' % len(VARS),
        'value': n / (ms * 1e-3), 'unit': 'element-records/s', 'n_gpus': 1, 'ms_per_record': ms, 'dtype': 'f32',
        'data': 'synthetic', 'config': {'workload': 'output path: %d elements x %d output times x %d variables' % (n, nt, len(VARS))},
        'roofline': {'bound': 'hbm', 'kernel': 'k_hist_record', 'achieved': bytes_rec * n / (ms * 1e-3) / 1e9, 'peak': 8000.0,
                     'unit': 'GB/s', 'frac': bytes_rec * n / (ms * 1e-3) / 8e12, 'traffic': None,
                     'algorithmic_bytes_per_record': bytes_rec},
        'flush': {'bytes': flush_bytes, 'seconds': t_flush, 'GB_per_s': flush_bytes / t_flush / 1e9,
                  'host_blocked_seconds': t_launch, 'note': 'device -> pinned host over PCIe, asynchronous to the compute stream'},
        'cpu_baseline': {'value': cpu, 'unit': 'element-records/s', 'cores': 1, 'kind': 'port',
                         'sample': '%d elements x %d records, numpy restatement oracle/history.py' % (m, reps_cpu)},
    }
    print(json.dumps(out), flush=True)
    H.close()
    P.close()
    ctx.close()


if __name__ == '__main__':
    main()
