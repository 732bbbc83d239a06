#!/usr/bin/env python
"""Experiment (profiles/r06_ab_variants.txt): do the step launch and the mixing launch of C3 overlap when they belong to
DIFFERENT halves of the particle set on two HIP streams?  Both launches are issue-bound with the SIMDs' VALU 50-60 % busy; if
waves of the two kernels co-reside, the bubbles of one could be filled by the other.

  one context, 10 M elements, the bench's C3 sequence                      -> ms per step (baseline)
  two contexts (= two streams) on the same device, 5 M elements each,
  their sequences enqueued alternately by one host thread                  -> ms per step of the pair
"""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from opendrift_amd.device import Context  # noqa: E402

N = int(os.environ.get('N', 10_000_000))
STEPS = int(os.environ.get('STEPS', 96))
fields = bench.make_fields('c3')
rng = np.random.default_rng(1000)
lon, lat, z = bench.seed_particles('c3', fields, N, rng)


def make(lo, hi):
    ctx = Context(device=0, seed=0)
    ctx.set_stage_math('fast')
    wl = bench.Workload('c3', ctx, fields, (0, 0, 1), via_torch=False)
    P = ctx.particles(hi - lo)
    P.append(lon[lo:hi], lat[lo:hi], z=z[lo:hi], id=np.arange(lo, hi, dtype=np.int32))
    return ctx, wl, P


def run(sets, steps, first):
    for c, _, _ in sets:
        c.sync()
    t0 = time.perf_counter()
    for k in range(first, first + steps):
        for _, wl, P in sets:
            wl.step(P, k)
    for c, _, _ in sets:
        c.sync()
    return (time.perf_counter() - t0) / steps * 1e3


one = [make(0, N)]
run(one, 200, 0)
a = [run(one, STEPS, 200 + i * STEPS) for i in range(3)]
print('one stream, %d elements:            ms per step %s' % (N, ' '.join('%.4f' % v for v in a)), flush=True)
for c, _, P in one:
    P.close(); c.close()
two = [make(0, N // 2), make(N // 2, N)]
run(two, 200, 0)
b = [run(two, STEPS, 200 + i * STEPS) for i in range(3)]
print('two streams, %d elements each:      ms per step %s' % (N // 2, ' '.join('%.4f' % v for v in b)), flush=True)
# the same two halves one after the other on their streams (no overlap possible: sync between)
def run_serial(sets, steps, first):
    t0 = time.perf_counter()
    for k in range(first, first + steps):
        for c, wl, P in sets:
            wl.step(P, k)
            c.sync()
    return (time.perf_counter() - t0) / steps * 1e3
s = [run_serial(two, STEPS, 200 + (3 + i) * STEPS) for i in range(2)]
print('the two halves serialised by the host: ms per step %s' % ' '.join('%.4f' % v for v in s), flush=True)
