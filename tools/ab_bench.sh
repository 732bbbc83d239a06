#!/bin/bash
# A/B comparison of two builds of the library on the same GPU box (box-to-box variation is several per cent):
#   tools/ab_build.sh "<flags A>" "<flags B>"   (in the build container), then on the GPU:
#   tools/ab_bench.sh [workload] [reps]
W=${1:-c3}; R=${2:-3}
for i in $(seq $R); do
  for v in A B; do
    ODR_LIB=$PWD/tools/_lib$v.so python bench.py --workload $W --no-cpu 2>/dev/null | tail -1 | python tools/bl.py $v
  done
done
