#!/bin/bash
# Round profiles: kernel stats for c3/c4/c5 and the PMC passes of the c3 bench (separate passes per counter set).
R=${ROUND:-r01}
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
for w in c3 c4 c5; do
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/stats_$w -o st -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 32 --warmup 3 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_$w.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/stats_$w $GRAFT_REPO_ROOT/gpurun_out/prof/${R}_${w}_kernel_stats.txt > /dev/null
done
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_$tag | grep -E "k_step_grid|k_vmix_col|k_gather|k_sort" | grep -E "n=" >> $GRAFT_REPO_ROOT/gpurun_out/prof/${R}_c3_pmc_raw.txt
done
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
tail -n 1 gpurun_out/prof/bench_c3.log | cut -c1-300
head -12 gpurun_out/prof/${R}_c3_kernel_stats.txt
cat gpurun_out/prof/${R}_c3_pmc_raw.txt
