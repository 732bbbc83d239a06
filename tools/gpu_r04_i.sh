#!/bin/bash
# round 4, GPU call I: suite, sharded bench rehearsal (2 ranks on one GPU over gloo), c3 profile pipeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest_all.log
cat $O/pytest_all.log
ODR_DIST_BACKEND=gloo ODR_BENCH_ONE_MODE=1 timeout 900 python bench.py --gpus 2 --particles 2000000 --steps 24 --no-cpu --no-extras > $O/bench_2ranks_gloo.log 2>&1
tail -1 $O/bench_2ranks_gloo.log | cut -c1-600
grep -o '"sharded_loop": {[^}]*}' $O/bench_2ranks_gloo.log
bash tools/gpu_profile_r04.sh c3 > $O/profile_c3.log 2>&1
tail -12 $O/profile_c3.log
