#!/bin/bash
# One GPU-box round trip: parity tests (full log under gpurun_out/pytest.log), then the bench lines of the workloads
# given in $WL (default: c3), printed compactly.  $BENCH_ARGS are passed on.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; tail -4 gpurun_out/pytest.log | cut -c1-300
for w in ${WL:-c3}; do
  python bench.py --workload $w --no-cpu ${BENCH_ARGS} 2>gpurun_out/bench_$w.err | tail -1 | tee gpurun_out/bench_$w.json | python tools/bl.py
done
