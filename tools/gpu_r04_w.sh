#!/bin/bash
# round 4, GPU call W: step + mixing of particle windows on separate streams (ODR_LANES), re-measured on the round-4 kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w; mkdir -p $O
export ODR_BENCH_ONE_MODE=1
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --workload c3 --steps 96 --no-cpu --no-extras 2>&1 | tail -1 > $O/$name.json
  python - <<PY
import json
try:
    d=json.load(open('$O/$name.json'))
    print('%-22s ms/step %.4f' % ('$name', d['ms_per_step']))
except Exception as e:
    print('$name', 'failed', e)
PY
}
run lanes1
run lanes2 ODR_LANES=2
run lanes4 ODR_LANES=4
run lanes8 ODR_LANES=8
