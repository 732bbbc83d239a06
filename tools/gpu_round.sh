#!/bin/bash
# One GPU-box round trip: parity tests, then the bench lines of the workloads given in $WL
# (default: c3).  Outputs under gpurun_out/.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for w in ${WL:-c3}; do
  python bench.py --workload $w --no-cpu ${BENCH_ARGS} 2>&1 | tail -1 | tee gpurun_out/bench_$w.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:3], 'ms/step %.3f'%d['ms_per_step'], 'value %.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'frac %.3f'%d['roofline']['frac'])"
done
if [ -n "$PMC" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload ${PMC} --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py gpurun_out/pmc 2>&1 | grep -v "k_blk\|k_scan\|k_cmp\|rocclr" | head -60
fi
if [ -n "$STATS" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/stats -o st -- python $GRAFT_REPO_ROOT/bench.py --workload ${STATS} --steps 20 --warmup 3 --no-cpu > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py gpurun_out/stats 2>&1 | head -30
fi
