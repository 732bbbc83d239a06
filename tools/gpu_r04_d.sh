#!/bin/bash
# round 4, GPU call D: kept (u,v) records between the samples of a step (UVKeep) -- A = -DODR_NO_KEEP, B = default
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tile.py tests/test_gpu_fused_step.py tests/test_gpu_stage_math.py -x -q 2>&1 | tail -5 > $O/pytest.log
cat $O/pytest.log
export ODR_BENCH_ONE_MODE=1 ODR_TILE=0
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
run nokeep_$rep ODR_LIB=$PWD/tools/_libA.so
run keep_$rep
done
run nokeep_exact ODR_LIB=$PWD/tools/_libA.so ODR_STAGE_MATH=exact
run keep_exact ODR_STAGE_MATH=exact
run keep_tile ODR_TILE=1 ODR_TILE_LDS=51200
W=c4 run c4_nokeep ODR_LIB=$PWD/tools/_libA.so
W=c4 run c4_keep
cd /tmp && export TMPDIR=/tmp
for v in A B; do
  if [ $v = A ]; then export ODR_LIB=$GRAFT_REPO_ROOT/tools/_libA.so; else unset ODR_LIB; fi
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
    rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/$O/pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 6 --warmup 2 --no-cpu --no-extras > /dev/null 2>&1
    echo "variant $v" >> $GRAFT_REPO_ROOT/$O/pmc.txt
    python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/pmc | grep -E "n=" | grep -E "k_step_grid" >> $GRAFT_REPO_ROOT/$O/pmc.txt
    rm -rf $GRAFT_REPO_ROOT/$O/pmc
  done
done
cat $GRAFT_REPO_ROOT/$O/pmc.txt | cut -c1-30,90-200
