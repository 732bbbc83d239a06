#!/bin/bash
# C3 bench with the step + mixing in 1 / 2 / 4 / 8 lanes (ODR_LANES), same box:  tools/lanes_bench.sh [reps]
for i in $(seq ${1:-2}); do
  for L in ${LANES:-1 2 4 8}; do
    ODR_LANES=$L python bench.py --no-cpu --no-extras --steps ${STEPS:-200} 2>/dev/null | tail -1 | python tools/bl.py "lanes $L"
  done
done
