#!/bin/bash
# Kernel times (rocprofv3 --kernel-trace --stats) and instruction counters (separate --pmc pass) of the C3 bench on the
# GPU box.  Usage (from the repo root, on the box): tools/prof_c3.sh [tag] [workload]   -> gpurun_out/prof_<tag>/
TAG=${1:-x}; W=${2:-c3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o st -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 32 --warmup 3 --no-cpu > $OUT/bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/stats $OUT/kernel_stats.txt > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD -d $OUT/pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/pmc | grep -E "k_step_grid|k_vmix_col|k_gather|k_sort_hist" | grep -E "n=" > $OUT/pmc.txt
cd $GRAFT_REPO_ROOT
tail -n 1 $OUT/bench.log | cut -c1-260
head -14 $OUT/kernel_stats.txt
cat $OUT/pmc.txt | cut -c1-60,100-170
