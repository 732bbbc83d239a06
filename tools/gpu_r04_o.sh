#!/bin/bash
# round 4, GPU call O: slot B/C/D gathers behind slot A's (ODR_BURST_EARLY), first Philox block before the column gathers
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; mkdir -p $O
ODR_LIB=$PWD/tools/_libB.so timeout 900 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_parity.py tests/test_gpu_stage_math.py -x -q 2>&1 | tail -3
ODR_LIB=$PWD/tools/_libC.so timeout 900 python -m pytest tests/test_gpu_diffusivity.py tests/test_gpu_fused_step.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
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
run A1
run B1 ODR_LIB=$PWD/tools/_libB.so
run C1 ODR_LIB=$PWD/tools/_libC.so
run A2
run B2 ODR_LIB=$PWD/tools/_libB.so
run C2 ODR_LIB=$PWD/tools/_libC.so
W=c4 run c4_A
W=c4 run c4_B ODR_LIB=$PWD/tools/_libB.so
W=c5 run c5_A
W=c5 run c5_B ODR_LIB=$PWD/tools/_libB.so
