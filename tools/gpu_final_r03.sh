#!/bin/bash
# Round-3 measurement run on the GPU box: profiles (tools/gpu_profile_r03.sh), then the bench lines of the three workloads
# (c3 with every leg: both stage arithmetics, model API, PCIe-inclusive, CPU baseline) -> gpurun_out/prof_r03/r03_*_bench_line.json
cd $GRAFT_REPO_ROOT
tools/gpu_profile_r03.sh > gpurun_out/prof_r03_script.log 2>&1
P=gpurun_out/prof_r03
python bench.py 2> $P/bench_c3.err | tail -1 > $P/r03_c3_bench_line.json
python bench.py --workload c4 --no-extras --cpu-particles 100000 2> $P/bench_c4.err | tail -1 > $P/r03_c4_bench_line.json
python bench.py --workload c5 --no-extras --cpu-particles 100000 2> $P/bench_c5.err | tail -1 > $P/r03_c5_bench_line.json
python bench.py --workload c2 --no-extras --no-cpu 2> $P/bench_c2.err | tail -1 > $P/r03_c2_bench_line.json
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" > $P/smoke.log 2>&1; tail -1 $P/smoke.log
for w in c3 c4 c5 c2; do python - $P/r03_${w}_bench_line.json <<'PY'
import sys, json
d = json.load(open(sys.argv[1]))
print(d['config']['workload'][:3], 'ms/step %.4f' % d['ms_per_step'], 'value %.3e' % d['value'], 'roofline', d['roofline'].get('bound'), '%.3f' % d['roofline'].get('frac', 0),
      'kernel_ms', d['roofline'].get('kernel_ms'), 'other', {k: round(v['ms_per_step'], 4) for k, v in d.items() if k.startswith('stage_math_')},
      'model_api', d.get('model_api', {}).get('ms_per_step'), 'pcie', d.get('pcie_inclusive', {}).get('ms_per_step'), 'cpu', d.get('cpu_baseline', {}).get('value'))
PY
done
