#!/bin/bash
# round 4, GPU call K: profiles of c3 / c4 / c5 (kernel stats + PMC passes) and the bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
bash tools/gpu_profile_r04.sh c3 c4 c5 > $O/profile.log 2>&1
tail -30 $O/profile.log | cut -c1-200
