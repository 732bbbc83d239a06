#!/bin/bash
# round 4, GPU call X: the mixing launch's first two Philox blocks drawn by the step launch (StepDesc.mix_pre)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vmix_window.py tests/test_gpu_diffusivity.py tests/test_gpu_fused_step.py tests/test_gpu_parity.py tests/test_gpu_oil.py tests/test_gpu_model_api.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -3
export ODR_BENCH_ONE_MODE=1
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --workload c3 --steps 96 --no-cpu --no-extras 2>&1 | tail -1 > $O/$name.json
  python - <<PY
import json
try:
    d=json.load(open('$O/$name.json'))
    print('%-22s ms/step %.4f kernel_ms %.4f k2 %.4f' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('second_kernel',{}).get('kernel_ms',0)))
except Exception as e:
    print('$name', 'failed', e)
PY
}
run pre1
run nopre1 ODR_NO_RNG_PRE=1
run pre2
run nopre2 ODR_NO_RNG_PRE=1
