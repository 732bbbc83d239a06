#!/bin/bash
# round 4, GPU call U: mixing sub-steps unrolled in groups of five (word of the Philox block known at compile time)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vmix_window.py tests/test_gpu_diffusivity.py tests/test_gpu_fused_step.py tests/test_gpu_parity.py tests/test_gpu_oil.py -x -q 2>&1 | tail -3
export ODR_BENCH_ONE_MODE=1
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --workload ${W:-c3} --steps 96 --no-cpu --no-extras 2>&1 | tail -1 > $O/$name.json
  python - <<PY
import json
try:
    d=json.load(open('$O/$name.json'))
    print('%-22s ms/step %.4f kernel_ms %.4f k2 %.4f' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('second_kernel',{}).get('kernel_ms',0)))
except Exception as e:
    print('$name', 'failed', e)
PY
}
run unrolled1
run rolled1 ODR_LIB=$PWD/tools/_libB.so
run unrolled2
run rolled2 ODR_LIB=$PWD/tools/_libB.so
