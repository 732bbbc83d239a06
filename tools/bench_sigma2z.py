#!/usr/bin/env python
"""Measurement of the sigma -> z regridding: odr_sgrid_zslice on MI355X beside the CPU
restatement (oracle/roms.py, NumPy, 1 core).  Workload: one 3-D variable of a ROMS-shaped block, N s-levels on
ny x nx nodes regridded to kmax z levels (what reader_ROMS_native does per variable and time level).

    python tools/bench_sigma2z.py [--n 35] [--ny 1024] [--nx 1024] [--kmax 12]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=35)
    ap.add_argument('--ny', type=int, default=1024)
    ap.add_argument('--nx', type=int, default=1024)
    ap.add_argument('--kmax', type=int, default=12)
    ap.add_argument('--reps', type=int, default=20)
    a = ap.parse_args()
    import torch
    import __graft_entry__ as G
    G.build()
    from gen_helpers import stretching
    from opendrift_amd.device import Context, SigmaGrid
    ctx = Context(device=0, seed=0)
    rng = np.random.default_rng(0)
    N, ny, nx, kmax = a.n, a.ny, a.nx, a.kmax
    H = rng.uniform(20.0, 800.0, (ny, nx))
    Cs = stretching(N)
    Z = np.array([0, -.5, -1, -3, -5, -10, -25, -50, -75, -100, -150, -200, -250, -300, -400, -500][:kmax], float)
    F = (rng.standard_normal((N, ny, nx)) * 0.3).astype(np.float32)
    sg = SigmaGrid(ctx, H, 20.0, Cs, Vtransform=2)
    Fd = torch.from_numpy(F).cuda()                      # the raw block already resident (read by the host reader)
    torch.cuda.synchronize()
    sg.zslice(Fd.data_ptr(), Z)
    ctx.sync()
    ctx.timer_begin()
    for _ in range(a.reps):
        sg.zslice(Fd.data_ptr(), Z)
    ms = ctx.timer_end() / a.reps
    M = ny * nx
    # algorithmic bytes per column: the field once (N x 4), the level depths once (N x 8), the result (kmax x 4)
    bytes_col = N * 4 + N * 8 + kmax * 4
    # host-resident field (PCIe inclusive), as the reader would call it
    t0 = time.perf_counter()
    sg.zslice(F, Z)
    ctx.sync()
    t_host = time.perf_counter() - t0
    # CPU baseline on a bounded sample of columns
    from oracle import roms
    ms_rows = max(8, min(ny, int(ny * 4e6 / (M * 1.0))))
    zr = roms.z_rho(H[:ms_rows], None, 20.0, Cs, Vtransform=2)
    t0 = time.perf_counter()
    reps_cpu = 0
    while time.perf_counter() - t0 < 10.0 and reps_cpu < 10:
        roms.zslice(F[:, :ms_rows], zr, Z)
        reps_cpu += 1
    cpu = ms_rows * nx * reps_cpu / (time.perf_counter() - t0)
    out = {
        'metric': 'sigma->z regridded columns/s (N=%d s-levels -> %d z levels)' % (N, kmax), 'value': M / (ms * 1e-3),
        'unit': 'columns/s', 'n_gpus': 1, 'ms_per_variable': ms, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'one 3-D variable, %d x %d x %d -> %d x %d x %d' % (N, ny, nx, kmax, ny, nx)},
        'roofline': {'bound': 'hbm', 'kernel': 'k_roms_zslice', 'achieved': bytes_col * M / (ms * 1e-3) / 1e9, 'peak': 8000.0,
                     'unit': 'GB/s', 'frac': bytes_col * M / (ms * 1e-3) / 8e12, 'traffic': None,
                     'algorithmic_bytes_per_column': bytes_col},
        'host_field_seconds': t_host,
        'cpu_baseline': {'value': cpu, 'unit': 'columns/s', 'cores': 1, 'kind': 'port',
                         'sample': '%d x %d columns x %d repetitions, NumPy restatement oracle/roms.py' % (ms_rows, nx, reps_cpu)},
    }
    print(json.dumps(out), flush=True)
    sg.close()
    ctx.close()


if __name__ == '__main__':
    main()
