#!/bin/bash
# round 4, final GPU call: full GPU suite, the bench lines that go into profiles/, all profiles after the last kernel changes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/r04_final_gpu_tests.txt
timeout 900 python bench.py > $O/bench_default.log 2>&1; grep "^{" $O/bench_default.log | tail -1 > $O/r04_c3_bench_line.json
timeout 600 python bench.py --steps 20 --no-cpu > $O/bench_s20.log 2>&1; grep "^{" $O/bench_s20.log | tail -1 > $O/r04_c3_bench_line_steps20.json
for w in c2 c4 c5; do
  timeout 600 python bench.py --workload $w --no-cpu > $O/bench_$w.log 2>&1; grep "^{" $O/bench_$w.log | tail -1 > $O/r04_${w}_bench_line.json
done
python - <<'PY'
import json
for n in ('r04_c3_bench_line', 'r04_c3_bench_line_steps20', 'r04_c2_bench_line', 'r04_c4_bench_line', 'r04_c5_bench_line'):
    try:
        d = json.load(open('gpurun_out/r04_final/%s.json' % n))
        print('%-28s value %.4g %s  ms/step %.4f  roofline.frac %.3f  step.frac %s  cpu %s' % (
            n, d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d.get('roofline_step', {}).get('frac'),
            d.get('cpu_baseline', {}).get('value')))
    except Exception as e:
        print(n, 'failed', e)
PY
bash tools/gpu_profile_r04.sh c3 c4 c5 2>&1 | tail -4
