#!/bin/bash
# round 4, GPU call S: the movers' global tests formed by the step launch (C4)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_movers.py -x -q -rs 2>&1 | tail -4
export ODR_BENCH_ONE_MODE=1
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --workload ${W:-c4} --steps 96 --no-cpu --no-extras 2>&1 | tail -1 > $O/$name.json
  python - <<PY
import json
try:
    d=json.load(open('$O/$name.json'))
    print('%-22s ms/step %.4f kernel_ms %.4f' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms']))
except Exception as e:
    print('$name', 'failed', e)
PY
}
run pass1 ODR_BENCH_REDUCE_PASS=1
run launch1
run pass2 ODR_BENCH_REDUCE_PASS=1
run launch2
W=c3 run c3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/st -o st -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 32 --warmup 2 --no-cpu --no-extras > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/st $GRAFT_REPO_ROOT/$O/c4_kernel_stats.txt > /dev/null
rm -rf $GRAFT_REPO_ROOT/$O/st
head -10 $GRAFT_REPO_ROOT/$O/c4_kernel_stats.txt | cut -c1-70,105-170
