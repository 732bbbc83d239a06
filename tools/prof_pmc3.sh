#!/bin/bash
# Memory-pipeline and instruction-cache counters of the C3 step kernels (round 3): which unit the dominant kernel waits on.
#   tools/prof_pmc3.sh [tag] [scheme ...]     -> gpurun_out/prof_<tag>/pmc3_<scheme>.txt
# Every counter set is its own rocprofv3 run with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).
TAG=${1:-x}; shift
SCHEMES=${@:-runge-kutta4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=(
 "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
 "TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
 "TA_FLAT_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_sum TCC_READ_sum TCC_WRITE_sum SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH"
)
for sch in $SCHEMES; do
  rm -f $OUT/pmc3_$sch.txt
  k=0
  for set in "${SETS[@]}"; do
    k=$((k+1))
    ODR_BENCH_SCHEME=$sch rocprofv3 --kernel-trace --pmc $set -d $OUT/p3_${sch}_$k -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload ${WL:-c3} --steps 6 --warmup 2 --no-cpu --no-extras > $OUT/p3_${sch}_$k.log 2>&1
    python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/p3_${sch}_$k 2>/dev/null | grep -E "n=" | grep -E "k_step_grid|k_vmix_col|k_step_leeway|k_leeway|k_env_grid" >> $OUT/pmc3_$sch.txt
    rm -rf $OUT/p3_${sch}_$k
  done
  echo "== $sch"; awk '{print substr($0,1,34), $(NF-2), $(NF-1), $NF}' $OUT/pmc3_$sch.txt
done
