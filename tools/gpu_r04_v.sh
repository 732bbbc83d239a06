#!/bin/bash
# round 4, GPU call V: C3 re-sort interval after the last kernel changes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04v; mkdir -p $O
export ODR_BENCH_ONE_MODE=1
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --workload ${W:-c3} --steps 192 --no-cpu --no-extras 2>&1 | tail -1 > $O/$name.json
  python - <<PY
import json
try:
    d=json.load(open('$O/$name.json'))
    print('%-22s ms/step %.4f kernel_ms %.4f k2 %.4f' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('second_kernel',{}).get('kernel_ms',0)))
except Exception as e:
    print('$name', 'failed', e)
PY
}
run s24
run s32 ODR_SORT_EVERY=32
run s48 ODR_SORT_EVERY=48
run s24b
run s32b ODR_SORT_EVERY=32
W=c4 run c4_s16
W=c4 run c4_s32 ODR_SORT_EVERY=32
