#!/bin/bash
# The profiles of a round (run on the GPU box from the repo root: [R=r06] [ODR_STAGE_MATH=exact] tools/gpu_profile_round.sh [workloads]):
#   kernel stats (rocprofv3 --kernel-trace --stats) and PMC passes of the c3 / c4 / c5 bench -- every counter set in its own
#   run with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 sections) -- and the kernel stats of OceanDrift.run() on
#   the c3 inputs.  Everything lands under gpurun_out/prof_$R[_exact]; tools/collect_profiles_round.py copies the summaries into profiles/.
#   Every rocprofv3 pass runs under `timeout`: a counter set rocprofv3 chokes on must not eat the round's GPU minutes.
R=${R:-r06}
SFX=""; [ "${ODR_STAGE_MATH:-fast}" = exact ] && SFX=_exact
P=$GRAFT_REPO_ROOT/gpurun_out/prof_$R$SFX
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
export ODR_BENCH_ONE_MODE=1   # the profiled command runs one stage arithmetic (ODR_STAGE_MATH, default fast)
SETS=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_IFETCH"
 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32"
 "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"
)
for w in ${@:-c3 c4 c5}; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $P/stats_$w -o st -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 32 --warmup 3 --no-cpu --no-extras > $P/bench_$w.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $P/stats_$w $P/${R}_${w}_kernel_stats.txt > /dev/null
  rm -rf $P/stats_$w
  rm -f $P/${R}_${w}_pmc_raw.txt
  k=0
  for set in "${SETS[@]}"; do
    k=$((k+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $set -d $P/pmc_${w}_$k -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 6 --warmup 2 --no-cpu --no-extras > /dev/null 2>&1
    python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $P/pmc_${w}_$k | grep -E "n=" | grep -E "k_step_grid|k_step_tile|k_step_list|k_vmix_col|k_step_leeway|k_leeway|k_env_grid|k_gather_perm|k_sort_hist|k_sort_perm|k_reduce|k_movers|k_cmp_|k_wg_|k_scan_|k_red_init" >> $P/${R}_${w}_pmc_raw.txt
    rm -rf $P/pmc_${w}_$k
  done
done
if [ -z "$SFX" ] && { [ -z "$1" ] || [ "$1" = c3 ]; }; then
  timeout 300 rocprofv3 --kernel-trace --stats -d $P/stats_model -o st -- python $GRAFT_REPO_ROOT/tools/model_time.py 10000000 48 > $P/${R}_c3_model_api_host_profile.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $P/stats_model $P/${R}_c3_model_api_kernel_stats.txt > /dev/null
  rm -rf $P/stats_model
fi
cd $GRAFT_REPO_ROOT
for w in ${@:-c3 c4 c5}; do tail -n 1 $P/bench_$w.log | cut -c1-200; head -8 $P/${R}_${w}_kernel_stats.txt | cut -c1-70,105-160; done
