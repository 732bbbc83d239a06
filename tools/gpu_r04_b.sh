#!/bin/bash
# round 4, GPU call B: remaining parity tests, sort-interval / LDS variants, counters of k_step_tile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tile.py -q 2>&1 | tail -15 > $O/pytest_tile.log
cat $O/pytest_tile.log
export ODR_BENCH_ONE_MODE=1
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --workload c3 --steps 96 --no-cpu --no-extras 2>&1 | tail -1 > $O/$name.json
  python - <<PY
import json
try:
    d=json.load(open('$O/$name.json'))
    t=d.get('lds_tile') or {}
    print('%-28s ms/step %.4f kernel_ms %.4f vmix %.4f  handed/launch %s cuts %s' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('second_kernel',{}).get('kernel_ms',0), t.get('handed_over_per_launch'), t.get('rectangles_cut')))
except Exception as e:
    print('$name', 'failed', e)
PY
}
run base_s16 ODR_TILE=0
run base_s8 ODR_TILE=0 ODR_SORT_EVERY=8
run tile38_s8 ODR_TILE=1 ODR_SORT_EVERY=8
run tile38_s4 ODR_TILE=1 ODR_SORT_EVERY=4
run tile51_s8 ODR_TILE=1 ODR_TILE_LDS=51200 ODR_SORT_EVERY=8
run tile51_s16 ODR_TILE=1 ODR_TILE_LDS=51200
run tile76_s16 ODR_TILE=1 ODR_TILE_LDS=76800
cd /tmp && export TMPDIR=/tmp
SETS=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
 "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
k=0
for set in "${SETS[@]}"; do
  k=$((k+1))
  ODR_TILE=1 ODR_TILE_LDS=51200 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/$O/pmc_$k -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 6 --warmup 2 --no-cpu --no-extras > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/pmc_$k | grep -E "n=" | grep -E "k_step_tile|k_step_list|k_vmix_col" >> $GRAFT_REPO_ROOT/$O/pmc_tile51.txt
  rm -rf $GRAFT_REPO_ROOT/$O/pmc_$k
done
cat $GRAFT_REPO_ROOT/$O/pmc_tile51.txt | cut -c1-40,90-200
