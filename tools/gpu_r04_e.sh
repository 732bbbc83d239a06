#!/bin/bash
# round 4, GPU call E: mixing inside the step launch (ODR_FUSED_MIX=1) with the kept records and the fast stage math
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_tile.py -x -q 2>&1 | tail -5 > $O/pytest.log
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
run two_launches_1
run fused_1 ODR_FUSED_MIX=1
run two_launches_2
run fused_2 ODR_FUSED_MIX=1
run fused_exact ODR_FUSED_MIX=1 ODR_STAGE_MATH=exact
run two_exact ODR_STAGE_MATH=exact
cd /tmp && export TMPDIR=/tmp
ODR_FUSED_MIX=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/st -o st -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 32 --warmup 2 --no-cpu --no-extras > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/st $GRAFT_REPO_ROOT/$O/stats_fused.txt > /dev/null
rm -rf $GRAFT_REPO_ROOT/$O/st
head -8 $GRAFT_REPO_ROOT/$O/stats_fused.txt | cut -c1-70,105-170
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  ODR_FUSED_MIX=1 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/$O/pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 6 --warmup 2 --no-cpu --no-extras > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/pmc | grep -E "n=" | grep -E "k_step_grid" >> $GRAFT_REPO_ROOT/$O/pmc.txt
  rm -rf $GRAFT_REPO_ROOT/$O/pmc
done
cat $GRAFT_REPO_ROOT/$O/pmc.txt | cut -c1-30,90-200
