#!/bin/bash
# One parametrised A/B script for the GPU box (replaces the round-4 one-offs tools/gpu_r04_[a-x].sh; the record of what they
# measured is profiles/r04_ab_variants.txt):
#
#   tools/gpu_ab.sh OUT [-t "pytest args"] [-w workload] [-s steps] [-b "extra bench args"] RUN [RUN ...]
#
# RUN = name[,VAR=VALUE...]: one `bench.py --workload W --steps S --no-cpu --no-extras` with those variables set (ODR_LIB=tools/_libX.so
# selects a variant library built by tools/vbuild.sh X <flags>); runs execute in the order given -- list a pair twice (A B A B) to see
# the box's drift.  Every run's JSON line lands in gpurun_out/OUT/<name>.json, one summary line per run is printed.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift
mkdir -p $OUT
W=c3; S=96; B=""; T=""
while getopts "t:w:s:b:" o; do
  case $o in
    t) T="$OPTARG";; w) W="$OPTARG";; s) S="$OPTARG";; b) B="$OPTARG";;
  esac
done
shift $((OPTIND-1))
if [ -n "$T" ]; then
  timeout 1500 python -m pytest $T -x -q 2>&1 | tail -5 | tee $OUT/tests.txt
fi
k=0
for run in "$@"; do
  k=$((k+1))
  name=${run%%,*}
  vars=""
  if [ "$run" != "$name" ]; then vars=$(echo "${run#*,}" | tr ',' ' '); fi
  env ODR_BENCH_ONE_MODE=${ODR_BENCH_ONE_MODE-1} $vars timeout 900 python bench.py --workload $W --steps $S --no-cpu --no-extras $B > $OUT/${k}_$name.log 2>&1
  grep "^{" $OUT/${k}_$name.log | tail -1 > $OUT/${k}_$name.json
  python - "$OUT/${k}_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d['roofline']
    o = d.get('stage_math_exact') or d.get('stage_math_fast') or {}
    print('%-24s ms/step %.4f  kernel_ms %.4f  k2 %.4f  other-mode ms/step %s kernel %s' % (
        sys.argv[2], d['ms_per_step'], r['kernel_ms'], r.get('second_kernel', {}).get('kernel_ms', 0),
        ('%.4f' % o['ms_per_step']) if o else '-', ('%.4f' % o['kernel_ms']) if o else '-'))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
done
