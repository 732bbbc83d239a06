#!/bin/bash
# round 4, GPU call A: parity of the LDS-tile step + A/B of the C3 step with and without it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tile.py -x -q 2>&1 | tail -15 > $O/pytest_tile.log
cat $O/pytest_tile.log
export ODR_BENCH_ONE_MODE=1
for v in 0 1; do
  ODR_TILE=$v timeout 600 python bench.py --workload c3 --steps 200 --no-cpu --no-extras 2>&1 | tail -1 > $O/c3_tile$v.json
  python - <<PY
import json
d=json.load(open('$O/c3_tile$v.json'))
print('tile=$v ms/step %.4f kernel_ms %.4f second %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('second_kernel',{}).get('kernel_ms',0)), d.get('lds_tile'))
PY
done
for lds in 24576 32768 51200 65536; do
  ODR_TILE=1 ODR_TILE_LDS=$lds timeout 600 python bench.py --workload c3 --steps 100 --no-cpu --no-extras 2>&1 | tail -1 > $O/c3_lds$lds.json
  python - <<PY
import json
d=json.load(open('$O/c3_lds$lds.json'))
print('lds=$lds ms/step %.4f kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']), d.get('lds_tile'))
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o st -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 32 --warmup 3 --no-cpu --no-extras > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/stats $GRAFT_REPO_ROOT/$O/c3_kernel_stats.txt > /dev/null
rm -rf $GRAFT_REPO_ROOT/$O/stats
head -14 $GRAFT_REPO_ROOT/$O/c3_kernel_stats.txt | cut -c1-70,105-170
