#!/bin/bash
# tools/gpu_pmc.sh OUT WORKLOAD "COUNTER SET 1" ["COUNTER SET 2" ...]   (GPU box)
# One `rocprofv3 --kernel-trace --pmc <set>` pass of `bench.py --workload W --steps 6 --warmup 2 --no-cpu --no-extras` per counter set
# (counters in their own runs, --kernel-trace only: MI355X_MICROARCH.md); per-dispatch averages of the step's kernels are appended to
# gpurun_out/OUT/pmc_raw.txt.  ODR_STAGE_MATH / ODR_LIB are passed through.
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; W=$2; shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ODR_BENCH_ONE_MODE=1
k=0
for set in "$@"; do
  k=$((k+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$k -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 6 --warmup 2 --no-cpu --no-extras > $OUT/pmc_$k.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/pmc_$k | grep -E "n=" | grep -E "k_step_grid|k_vmix_col|k_step_leeway|k_movers|k_gather_perm|k_sort_hist|k_sort_perm" | tee -a $OUT/pmc_raw.txt | cut -c1-40,90-160
  rm -rf $OUT/pmc_$k
done
