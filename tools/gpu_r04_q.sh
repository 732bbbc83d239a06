#!/bin/bash
# round 4, GPU call Q: mixing kernels against the number of reader levels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O
timeout 900 python tools/vmix_levels.py 2>&1 | tee $O/vmix_levels.txt
