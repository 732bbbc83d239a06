#!/bin/bash
# round 4, GPU call T: where the cost of the in-launch tests sits (what-if builds)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_movers.py tests/test_gpu_fused_step.py -x -q 2>&1 | tail -3
export ODR_BENCH_ONE_MODE=1
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --workload ${W:-c4} --steps 64 --no-cpu --no-extras 2>&1 | tail -1 > $O/$name.json
  python - <<PY
import json
try:
    d=json.load(open('$O/$name.json'))
    print('%-22s ms/step %.4f kernel_ms %.4f' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms']))
except Exception as e:
    print('$name', 'failed', e)
PY
}
run pass ODR_BENCH_REDUCE_PASS=1
run launch


run off_in_kernel ODR_NO_STEP_REDUCE=1
