#!/bin/bash
# Prices the host part of the sharded step on ONE GPU (no 8-GPU node in the pool): two gloo ranks x 5 M particles share the device,
# against one rank x 10 M and one rank x 5 M.  The two processes time-slice the GPU, so "2 x 5 M" is at best the 10 M step: what it
# costs beyond that is the host read + host-side collective per step + the GPU switching between the processes.
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export ODR_BENCH_ONE_MODE=1
timeout 600 python bench.py --steps 96 --no-cpu --no-extras > $O/one_10m.log 2>&1; grep "^{" $O/one_10m.log | tail -1 > $O/one_10m.json
timeout 600 python bench.py --steps 96 --no-cpu --no-extras --particles 5000000 > $O/one_5m.log 2>&1; grep "^{" $O/one_5m.log | tail -1 > $O/one_5m.json
ODR_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 96 --no-cpu --no-extras --particles 5000000 --block-every -1 > $O/two_5m.log 2>&1; grep "^{" $O/two_5m.log | tail -1 > $O/two_5m.json
# the sequence of the sharded step in ONE process (no second process on the device, the collective is a no-op): the status read
# between the two launches with the mixing launch enqueued ahead of it (default) and behind it (ODR_NO_SPECULATION=1)
ODR_BENCH_SHARDED_LOOP=1 timeout 600 python bench.py --steps 96 --no-cpu --no-extras --block-every -1 > $O/seq_spec.log 2>&1; grep "^{" $O/seq_spec.log | tail -1 > $O/seq_spec.json
ODR_NO_SPECULATION=1 ODR_BENCH_SHARDED_LOOP=1 timeout 600 python bench.py --steps 96 --no-cpu --no-extras --block-every -1 > $O/seq_nospec.log 2>&1; grep "^{" $O/seq_nospec.log | tail -1 > $O/seq_nospec.json
python - <<PY
import json
try:
    a, s1, s0 = (json.load(open('$O/%s.json' % n))['ms_per_step'] for n in ('one_10m', 'seq_spec', 'seq_nospec'))
    print('one process, the sharded step sequence: %.4f ms/step with the mixing launch enqueued ahead of the status read (+%.4f), %.4f reading first (+%.4f); plain sequence %.4f' % (s1, s1 - a, s0, s0 - a, a))
except Exception as e:
    print('sequence runs:', e)
PY
# the machinery alone: rank 0 holds the 10 M elements, rank 1 a thousand (no time-slicing between two full workloads); with the
# collective finished behind the mixing launch (default) and blocking at the end of the step (rounds 3-4)
ODR_BENCH_OTHER_RANKS_PARTICLES=1000 ODR_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 96 --no-cpu --no-extras --block-every -1 > $O/uneven.log 2>&1; grep "^{" $O/uneven.log | tail -1 > $O/uneven.json
ODR_BENCH_SYNC_SUMMARY=1 ODR_BENCH_OTHER_RANKS_PARTICLES=1000 ODR_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 96 --no-cpu --no-extras --block-every -1 > $O/uneven_sync.log 2>&1; grep "^{" $O/uneven_sync.log | tail -1 > $O/uneven_sync.json
ODR_BENCH_OTHER_RANKS_PARTICLES=1000 ODR_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 96 --no-cpu --no-extras > $O/uneven_levels.log 2>&1; grep "^{" $O/uneven_levels.log | tail -1 > $O/uneven_levels.json
python - <<PY
import json
try:
    a, u, us, ul = (json.load(open('$O/%s.json' % n)) for n in ('one_10m', 'uneven', 'uneven_sync', 'uneven_levels'))
    print('rank 0 x 10 M + rank 1 x 1000 (gloo, one GPU): %.4f ms/step (collective behind the mixing launch), %.4f (blocking, rounds 3-4), %.4f with a reader level every 6 steps;  1 process: %.4f' % (u['ms_per_step'], us['ms_per_step'], ul['ms_per_step'], a['ms_per_step']))
    print('overhead of the sharded machinery per step: %.4f ms (%.1f %%), blocking: %.4f ms' % (u['ms_per_step'] - a['ms_per_step'], 100 * (u['ms_per_step'] / a['ms_per_step'] - 1), us['ms_per_step'] - a['ms_per_step']))
    print('sharded_loop (uneven):', u.get('sharded_loop'))
except Exception as e:
    print('uneven runs:', e)
PY
python - <<PY
import json
a, b, c = (json.load(open('$O/%s.json' % n)) for n in ('one_10m', 'one_5m', 'two_5m'))
print('1 rank x 10 M: %.4f ms/step   1 rank x 5 M: %.4f   2 gloo ranks x 5 M on one GPU: %.4f ms/step' % (a['ms_per_step'], b['ms_per_step'], c['ms_per_step']))
print('sharded_loop:', c.get('sharded_loop'))
PY
tail -3 $O/two_5m.log | cut -c1-300
