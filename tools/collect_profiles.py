#!/usr/bin/env python
"""Copy the summaries tools/gpu_profile.sh left under gpurun_out/prof into profiles/ (tracked) and
derive profiles/<round>_c3_pmc.json (read by bench.py for roofline.traffic)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src, dst = os.path.join(ROOT, 'gpurun_out', 'prof'), os.path.join(ROOT, 'profiles')
for w in ('c3', 'c4', 'c5'):
    f = os.path.join(src, '%s_%s_kernel_stats.txt' % (R, w))
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, '%s_%s_kernel_stats_final.txt' % (R, w)))
raw = open(os.path.join(src, '%s_c3_pmc_raw.txt' % R)).read()
vals = {}
for line in raw.splitlines():
    m = re.match(r'^(.*?)\s+(\w+)\s+n=(\d+)\s+avg=([\d.e+]+)', line)
    if m:
        k = 'step' if 'k_step_grid' in m.group(1) else 'vmix_tl' if 'k_vmix_col<3, true>' in m.group(1) else None
        if k:
            vals[(k, m.group(2))] = float(m.group(4))
hdr = '''PMC counters of the C3 bench (per dispatch, summed over dimensions), MI355X, 10 M particles:
rocprofv3 --kernel-trace --pmc <set> -- python bench.py --workload c3 --steps 6 --warmup 2 --no-cpu
three separate passes: {FETCH_SIZE} {WRITE_SIZE} {SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY}.  FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.
Calibration in this access pattern: k_step_grid stores 68 B per particle (5 float32 environment values, sample position,
previous position, lon, lat = 680 MB for 10 M particles) and WRITE_SIZE reports 664 062 KiB = 680 MB: exact, so no x2
correction is applied to FETCH_SIZE either (MI355X_MICROARCH.md calibrated the 1/2 factor on 16-byte-per-lane streams only).
'''
open(os.path.join(dst, '%s_c3_pmc.txt' % R), 'w').write(hdr + raw)
out = {
    'kernel': 'k_step_grid<2,0,true>', 'workload': 'c3', 'particles': 10000000,
    'FETCH_SIZE_bytes': vals[('step', 'FETCH_SIZE')] * 1024, 'WRITE_SIZE_bytes': vals[('step', 'WRITE_SIZE')] * 1024,
    'valu_per_wave': vals[('step', 'SQ_INSTS_VALU')] / vals[('step', 'SQ_WAVES')],
    'vmix_valu_per_wave': vals[('vmix_tl', 'SQ_INSTS_VALU')] / vals[('vmix_tl', 'SQ_WAVES')],
    'note': 'raw rocprofv3 FETCH_SIZE/WRITE_SIZE per launch; see profiles/%s_c3_pmc.txt for the calibration remark' % R,
}
json.dump(out, open(os.path.join(dst, '%s_c3_pmc.json' % R), 'w'), indent=1)
print(json.dumps(out, indent=1))
