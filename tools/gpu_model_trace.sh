#!/bin/bash
# OceanDrift.run() on the C3 inputs under rocprofv3 --kernel-trace: where the device waits inside a step of the drop-in loop
# (tools/rocpd_timeline.py) and the kernel statistics.   tools/gpu_model_trace.sh OUT [particles] [steps]
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; N=${2:-10000000}; K=${3:-48}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ODR_MODEL_NOPROFILE=1 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o st -- python $GRAFT_REPO_ROOT/tools/model_time.py $N $K > $OUT/model_host_profile.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/trace $OUT/model_kernel_stats.txt > /dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $OUT/trace $OUT/model_timeline.txt 4 | head -60
rm -rf $OUT/trace
head -1 $OUT/model_host_profile.txt | cut -c1-200
if [ -n "$HOSTPROF" ]; then timeout 400 python $GRAFT_REPO_ROOT/tools/model_time.py $N $K > $OUT/model_host_profile.txt 2>&1; fi
