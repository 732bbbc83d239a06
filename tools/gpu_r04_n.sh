#!/bin/bash
# round 4, GPU call N: group-wise k_gather_perm; sort interval with the kept records
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_generic_paths.py -x -q 2>&1 | tail -4
export ODR_BENCH_ONE_MODE=1
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --workload ${W:-c3} --steps 96 --no-cpu --no-extras 2>&1 | tail -1 > $O/$name.json
  python - <<PY
import json
try:
    d=json.load(open('$O/$name.json'))
    print('%-22s ms/step %.4f kernel_ms %.4f k2 %.4f' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('second_kernel',{}).get('kernel_ms',0)))
except Exception as e:
    print('$name', 'failed', e)
PY
}
run s16_a
run s12 ODR_SORT_EVERY=12
run s24 ODR_SORT_EVERY=24
run s16_b
W=c5 run c5_s16
W=c5 run c5_s48 ODR_SORT_EVERY=48
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/st -o st -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 32 --warmup 2 --no-cpu --no-extras > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/st $GRAFT_REPO_ROOT/$O/c3_kernel_stats.txt > /dev/null
rm -rf $GRAFT_REPO_ROOT/$O/st
head -9 $GRAFT_REPO_ROOT/$O/c3_kernel_stats.txt | cut -c1-70,105-170
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/$O/pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 17 --warmup 2 --no-cpu --no-extras > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/pmc | grep -E "n=" | grep -E "k_gather_perm|k_sort" >> $GRAFT_REPO_ROOT/$O/pmc.txt
  rm -rf $GRAFT_REPO_ROOT/$O/pmc
done
cat $GRAFT_REPO_ROOT/$O/pmc.txt | cut -c1-40,90-200
