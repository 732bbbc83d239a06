#!/bin/bash
# round 4, GPU call P: vertical mixing on a five-level window (k_vmix_win), direct Philox blocks, cost of the generators
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vmix_window.py -x -q 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_gpu_diffusivity.py tests/test_gpu_fused_step.py tests/test_gpu_parity.py tests/test_gpu_model_api.py tests/test_gpu_generic_paths.py tests/test_gpu_oil.py tests/test_gpu_distributed.py -x -q 2>&1 | tail -3
true
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
run col1 ODR_VMIX_WINDOW=0
run win5_1
run win6_1 ODR_LIB=$PWD/tools/_libB.so
run win4_1 ODR_LIB=$PWD/tools/_libC.so
run col2 ODR_VMIX_WINDOW=0
run win5_2
run win6_2 ODR_LIB=$PWD/tools/_libB.so
run win4_2 ODR_LIB=$PWD/tools/_libC.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/st -o st -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 32 --warmup 2 --no-cpu --no-extras > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/st $GRAFT_REPO_ROOT/$O/c3_kernel_stats.txt > /dev/null
rm -rf $GRAFT_REPO_ROOT/$O/st
head -10 $GRAFT_REPO_ROOT/$O/c3_kernel_stats.txt | cut -c1-70,105-170
