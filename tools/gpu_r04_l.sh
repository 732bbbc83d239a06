#!/bin/bash
# round 4, GPU call L: the bench lines (default run as the driver makes it, --steps 20, and the other workloads)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
timeout 1200 python bench.py > $O/c3_default.log 2>&1; grep "^{" $O/c3_default.log | tail -1 > $O/r04_c3_bench_line.json
timeout 600 python bench.py --steps 20 --warmup 3 > $O/c3_steps20.log 2>&1; grep "^{" $O/c3_steps20.log | tail -1 > $O/r04_c3_bench_line_steps20.json
for w in c2 c4 c5; do
  timeout 600 python bench.py --workload $w --no-cpu > $O/$w.log 2>&1; grep "^{" $O/$w.log | tail -1 > $O/r04_${w}_bench_line.json
done
python - <<PY
import json
for f in ('r04_c3_bench_line','r04_c3_bench_line_steps20','r04_c2_bench_line','r04_c4_bench_line','r04_c5_bench_line'):
    try:
        d=json.load(open('$O/%s.json' % f))
        o=d.get('stage_math_exact') or {}
        print(f, 'ms/step %.4f value %.3e' % (d['ms_per_step'], d['value']), 'exact', o.get('ms_per_step'), 'model_api', (d.get('model_api') or {}).get('ms_per_step'), 'pcie', (d.get('pcie_inclusive') or {}).get('ms_per_step'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e:
        print(f, 'failed', e)
PY
