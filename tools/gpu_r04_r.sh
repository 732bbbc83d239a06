#!/bin/bash
# round 4, GPU call R: Leeway one-launch lane in run(), reader thread in the 2-rank run, C3 at re-sort interval 24
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04r; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_model_api.py tests/test_gpu_history.py tests/test_gpu_distributed.py tests/test_gpu_fused_step.py -x -q 2>&1 | tail -12
export ODR_BENCH_ONE_MODE=1
for w in c3 c5; do
  timeout 600 python bench.py --workload $w --steps 96 --no-cpu --no-extras 2>&1 | tail -1 > $O/$w.json
  python - <<PY
import json
d=json.load(open('$O/$w.json'))
print('$w ms/step %.4f kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))
PY
done
