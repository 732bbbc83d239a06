#!/bin/bash
# Two more SQ counter passes of the C3 bench (busy cycles per unit; VALU instruction classes).  tools/prof_pmc2.sh [tag]
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
k=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_BRANCH" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"; do
  k=$((k+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/pmcx$k -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/pmcx$k | grep -E "k_step_grid|k_vmix_col<3, true" | grep -E "n=" >> $OUT/pmc2.txt
done
awk '{print substr($0,1,40), $(NF-2), $NF}' $OUT/pmc2.txt
