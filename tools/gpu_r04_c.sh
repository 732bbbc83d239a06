#!/bin/bash
# round 4, GPU call C: depth bands in the sort key (ODR_SORT_ZBANDS) -- do the lanes of a wave share sectors?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
export ODR_BENCH_ONE_MODE=1 ODR_TILE=0
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --workload c3 --steps 96 --no-cpu --no-extras 2>&1 | tail -1 > $O/$name.json
  python - <<PY
import json
try:
    d=json.load(open('$O/$name.json'))
    print('%-22s ms/step %.4f kernel_ms %.4f vmix %.4f' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('second_kernel',{}).get('kernel_ms',0)))
except Exception as e:
    print('$name', 'failed', e)
PY
}
run zb0_s16
run zb1_s16 ODR_SORT_ZBANDS=1
run zb2_s16 ODR_SORT_ZBANDS=2
run zb3_s16 ODR_SORT_ZBANDS=3
run zb1_s8 ODR_SORT_ZBANDS=1 ODR_SORT_EVERY=8
run zb2_s8 ODR_SORT_ZBANDS=2 ODR_SORT_EVERY=8
run zb1_s32 ODR_SORT_ZBANDS=1 ODR_SORT_EVERY=32
cd /tmp && export TMPDIR=/tmp
for zb in 0 1; do
  ODR_SORT_ZBANDS=$zb rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -d $GRAFT_REPO_ROOT/$O/pmc_$zb -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 20 --warmup 2 --no-cpu --no-extras > /dev/null 2>&1
  echo "zbands=$zb" >> $GRAFT_REPO_ROOT/$O/pmc.txt
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/pmc_$zb | grep -E "n=" | grep -E "k_step_grid|k_vmix_col<3, true|k_sort|k_gather" >> $GRAFT_REPO_ROOT/$O/pmc.txt
  rm -rf $GRAFT_REPO_ROOT/$O/pmc_$zb
  ODR_SORT_ZBANDS=$zb rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/st_$zb -o st -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 32 --warmup 2 --no-cpu --no-extras > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/st_$zb $GRAFT_REPO_ROOT/$O/stats_zb$zb.txt > /dev/null
  rm -rf $GRAFT_REPO_ROOT/$O/st_$zb
  head -9 $GRAFT_REPO_ROOT/$O/stats_zb$zb.txt | cut -c1-60,105-170
done
cat $GRAFT_REPO_ROOT/$O/pmc.txt | cut -c1-40,90-200
