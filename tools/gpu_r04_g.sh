#!/bin/bash
# round 4, GPU call G: the whole GPU suite after the refactors + C4 with the reworked reduction
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_all.log
cat $O/pytest_all.log
export ODR_BENCH_ONE_MODE=1
for rep in 1 2; do
for w in c4 c3; do
  timeout 600 python bench.py --workload $w --steps 96 --no-cpu --no-extras 2>&1 | tail -1 > $O/${w}_$rep.json
  python - <<PY
import json
d=json.load(open('$O/${w}_$rep.json'))
print('$w ms/step %.4f kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))
PY
done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/st -o st -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 32 --warmup 2 --no-cpu --no-extras > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/st $GRAFT_REPO_ROOT/$O/c4_kernel_stats.txt > /dev/null
rm -rf $GRAFT_REPO_ROOT/$O/st
head -8 $GRAFT_REPO_ROOT/$O/c4_kernel_stats.txt | cut -c1-70,105-170
