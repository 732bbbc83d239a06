#!/bin/bash
# On the GPU box: tools/vbench.sh "<TAG ...>" [workload] [reps] [bench args]: bench lines of the variant libraries, interleaved
TAGS=$1; W=${2:-c3}; R=${3:-2}; shift 3
for i in $(seq $R); do
  for v in $TAGS; do
    if [ $v = base ]; then L=$PWD/opendrift_amd/libodrift_hip.so; else L=$PWD/tools/_lib$v.so; fi
    ODR_LIB=$L python bench.py --workload $W --no-cpu --no-extras --steps ${STEPS:-128} "$@" 2>/dev/null | tail -1 | python tools/bl.py $v
  done
done
