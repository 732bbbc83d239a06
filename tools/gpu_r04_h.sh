#!/bin/bash
# round 4, GPU call H: FAST gates + occupancy of k_vmix_col (A: unconstrained = 102 registers, B: 5 waves, default: 6 waves)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_stage_math.py tests/test_gpu_movers.py tests/test_gpu_fused_step.py -x -q 2>&1 | tail -8 > $O/pytest.log
cat $O/pytest.log
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
for rep in 1 2; do
run vmix_w4_$rep ODR_LIB=$PWD/tools/_libA.so
run vmix_w5_$rep ODR_LIB=$PWD/tools/_libB.so
run vmix_w6_$rep
done
