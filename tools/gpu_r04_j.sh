#!/bin/bash
# round 4, GPU call J: five 24-bit uniforms per Philox block in the mixing loop: suite + C3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest_all.log
cat $O/pytest_all.log
export ODR_BENCH_ONE_MODE=1
for rep in 1 2 3; do
  timeout 600 python bench.py --workload c3 --steps 96 --no-cpu --no-extras 2>&1 | tail -1 > $O/c3_$rep.json
  python - <<PY
import json
d=json.load(open('$O/c3_$rep.json'))
print('c3 ms/step %.4f kernel_ms %.4f k2 %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('second_kernel',{}).get('kernel_ms',0)))
PY
done
python bench.py --workload c3 --steps 20 --no-cpu --no-extras 2>&1 | tail -1 > $O/c3_steps20.json
python - <<PY
import json
d=json.load(open('$O/c3_steps20.json'))
print('c3 steps 20: ms/step %.4f' % d['ms_per_step'], 'exact', d.get('stage_math_exact',{}).get('ms_per_step'))
PY
