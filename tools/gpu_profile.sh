#!/bin/bash
# Round profiles (run on the GPU box from the repo root: ROUND=r02 tools/gpu_profile.sh):
#   kernel stats (rocprofv3 --kernel-trace --stats) of the c3 / c4 / c5 bench and of OceanDrift.run() on the c3 inputs,
#   PMC passes of the c3 bench, each counter set in its own run with --kernel-trace only (MI355X_MICROARCH.md, HBM section).
# Everything lands under gpurun_out/prof; tools/collect_profiles.py copies the summaries into profiles/.
R=${ROUND:-r02}
P=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
for w in ${WL:-c3 c4 c5}; do
  rocprofv3 --kernel-trace --stats -d $P/stats_$w -o st -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 32 --warmup 3 --no-cpu --no-extras > $P/bench_$w.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $P/stats_$w $P/${R}_${w}_kernel_stats.txt > /dev/null
done
rocprofv3 --kernel-trace --stats -d $P/stats_model -o st -- python $GRAFT_REPO_ROOT/tools/model_time.py 10000000 48 > $P/${R}_c3_model_api_host_profile.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $P/stats_model $P/${R}_c3_model_api_kernel_stats.txt > /dev/null
rm -f $P/${R}_c3_pmc_raw.txt
k=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH"; do
  k=$((k+1))
  rocprofv3 --kernel-trace --pmc $set -d $P/pmc_$k -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 6 --warmup 2 --no-cpu --no-extras > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $P/pmc_$k | grep -E "n=" | grep -E "k_step_grid|k_vmix_col|k_gather_perm|k_sort_hist|k_sort_perm|k_fill_f32" >> $P/${R}_c3_pmc_raw.txt
done
cd $GRAFT_REPO_ROOT
tail -n 1 $P/bench_c3.log | cut -c1-300
head -12 $P/${R}_c3_kernel_stats.txt
cat $P/${R}_c3_pmc_raw.txt | awk '{print substr($0,1,46), $(NF-2), $(NF-1), $NF}'
