#!/bin/bash
# round 4, GPU call M: the three OceanDrift options against their reference goldens + the whole suite + default bench again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model_api.py -q -k "c21 or c22" 2>&1 | tail -25 > $O/pytest_options.log
cat $O/pytest_options.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest_all.log
cat $O/pytest_all.log
timeout 1200 python bench.py > $O/c3_default.log 2>&1; grep "^{" $O/c3_default.log | tail -1 > $O/c3_default.json
python - <<PY
import json
d=json.load(open('$O/c3_default.json'))
print('default: ms/step %.4f exact %s model_api %s pcie %s' % (d['ms_per_step'], (d.get('stage_math_exact') or {}).get('ms_per_step'), (d.get('model_api') or {}).get('ms_per_step'), (d.get('pcie_inclusive') or {}).get('ms_per_step')))
PY
