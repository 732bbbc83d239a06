#!/usr/bin/env python
"""Copy the summaries tools/gpu_profile.sh left under gpurun_out/prof into profiles/ (tracked) and derive
profiles/<round>_c3_pmc.json, which bench.py reads for roofline.traffic / roofline_hbm_counters / roofline_issue.

    python tools/collect_profiles.py r02

FETCH_SIZE: MI355X_MICROARCH.md (HBM section) says gfx950's rocprofv3 reports 1/2 of the bytes of a coalesced streaming
read and asks for a calibration in the kernel's own access pattern.  The calibration here: k_sort_hist reads lon and lat
of every particle once (8-byte lanes, 16 B per particle, nothing else of size), so its FETCH_SIZE x 1024 / (16 B x N) is
the factor for this code's 8-byte-per-lane streams.  It is stored next to the corrected figure."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else 'r02'
N = 10_000_000
src, dst = os.path.join(ROOT, 'gpurun_out', 'prof'), os.path.join(ROOT, 'profiles')
for w in ('c3', 'c4', 'c5', 'c3_model_api'):
    f = os.path.join(src, '%s_%s_kernel_stats.txt' % (R, w))
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, '%s_%s_kernel_stats.txt' % (R, w)))
f = os.path.join(src, '%s_c3_model_api_host_profile.txt' % R)
if os.path.exists(f):
    keep = [ln for ln in open(f).read().splitlines() if not ln.startswith('W2') and 'rocprofv3' not in ln][:70]
    open(os.path.join(dst, '%s_c3_model_api_host_profile.txt' % R), 'w').write('\n'.join(keep) + '\n')
f = os.path.join(src, 'bench_c3.log')
raw = open(os.path.join(src, '%s_c3_pmc_raw.txt' % R)).read()
KEY = [('k_step_grid<2, 0, true', 'step_rk4'), ('k_step_grid<0, 0, true', 'step_euler'), ('k_vmix_col<3, true', 'vmix_tl'),
       ('k_vmix_col<3, false', 'vmix'), ('k_gather_perm', 'gather'), ('k_sort_hist', 'sort_hist'), ('k_sort_perm', 'sort_perm'),
       ('k_fill_f32', 'fill')]
vals = {}
for line in raw.splitlines():
    m = re.match(r'^(.*?)\s+(\w+)\s+n=(\d+)\s+avg=([\d.e+-]+)', line)
    if m:
        for pat, k in KEY:
            if pat in m.group(1):
                vals[(k, m.group(2))] = float(m.group(4))
                break
calib = vals.get(('sort_hist', 'FETCH_SIZE'), float('nan')) * 1024 / (16.0 * N)
# k_gather_perm reads the permutation (4 B) and every byte it writes exactly once -- in permuted order: over-fetch, not a calibration
calib_g = vals.get(('gather', 'FETCH_SIZE'), float('nan')) / (vals.get(('gather', 'WRITE_SIZE'), float('nan')) + 4.0 * N / 1024)
hdr = '''PMC counters of the C3 bench (per dispatch, summed over dimensions; avg over the dispatches of that kernel), MI355X,
10 M particles:  rocprofv3 --kernel-trace --pmc <set> -- python bench.py --workload c3 --steps 6 --warmup 2 --no-cpu --no-extras
four separate passes: {FETCH_SIZE} {WRITE_SIZE} {SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD} {SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH}.
FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; the SQ cycle counters are in quad-cycles summed over waves.
FETCH_SIZE calibration in this code's access pattern: k_sort_hist streams lon and lat (16 B per particle, 8-byte lanes) =
160 MB; FETCH_SIZE x 1024 / 160 MB = %.3f  (1.0 = exact, 0.5 = the guide's halved count; its histogram atomics add a little).
k_gather_perm reads the permutation (4 B per particle) plus every byte it writes, once, but in permuted order: raw FETCH_SIZE /
(WRITE_SIZE + 4 B x N) = %.3f -- whole lines fetched for 4- and 8-byte elements (the first sort starts from random order).
WRITE_SIZE calibration: k_step_grid<RK4> stores 72 B per particle (5 float32 environment values, sample position, previous
position, lon, lat, age = 720 MB; z only where the sea floor lifts an element); WRITE_SIZE x 1024 / 720 MB = %.3f.
''' % (calib, calib_g, vals.get(('step_rk4', 'WRITE_SIZE'), float('nan')) * 1024 / (72.0 * N))
open(os.path.join(dst, '%s_c3_pmc.txt' % R), 'w').write(hdr + raw)
fs = vals[('step_rk4', 'FETCH_SIZE')] * 1024
out = {
    'kernel': 'k_step_grid<2,0,true> (RK4, lon/lat, 3D)', 'workload': 'c3', 'particles': N,
    'FETCH_SIZE_bytes_raw': fs, 'FETCH_SIZE_bytes_x2': 2 * fs,
    'FETCH_SIZE_calibration_8B_stream': calib, 'gather_perm_fetch_over_useful_raw': calib_g,
    'WRITE_SIZE_bytes': vals[('step_rk4', 'WRITE_SIZE')] * 1024,
    'SQ_WAVES': vals[('step_rk4', 'SQ_WAVES')], 'SQ_INSTS_VALU': vals[('step_rk4', 'SQ_INSTS_VALU')],
    'SQ_INSTS_SALU': vals[('step_rk4', 'SQ_INSTS_SALU')], 'SQ_INSTS_VMEM_RD': vals.get(('step_rk4', 'SQ_INSTS_VMEM_RD')),
    'valu_per_wave': vals[('step_rk4', 'SQ_INSTS_VALU')] / vals[('step_rk4', 'SQ_WAVES')],
    'SQ_WAVE_CYCLES': vals.get(('step_rk4', 'SQ_WAVE_CYCLES')), 'SQ_WAIT_INST_ANY': vals.get(('step_rk4', 'SQ_WAIT_INST_ANY')),
    'SQ_ACTIVE_INST_ANY': vals.get(('step_rk4', 'SQ_ACTIVE_INST_ANY')), 'SQ_ACTIVE_INST_VALU': vals.get(('step_rk4', 'SQ_ACTIVE_INST_VALU')),
    'SQ_BUSY_CYCLES': vals.get(('step_rk4', 'SQ_BUSY_CYCLES')),
    'vmix': {'kernel': 'k_vmix_col<3,true>',
             'FETCH_SIZE_bytes_x2': 2 * 1024 * vals.get(('vmix_tl', 'FETCH_SIZE'), float('nan')),
             'WRITE_SIZE_bytes': 1024 * vals.get(('vmix_tl', 'WRITE_SIZE'), float('nan')),
             'SQ_INSTS_VALU': vals.get(('vmix_tl', 'SQ_INSTS_VALU')),
             'valu_per_wave': vals[('vmix_tl', 'SQ_INSTS_VALU')] / vals[('vmix_tl', 'SQ_WAVES')],
             'SQ_ACTIVE_INST_VALU': vals.get(('vmix_tl', 'SQ_ACTIVE_INST_VALU')), 'SQ_BUSY_CYCLES': vals.get(('vmix_tl', 'SQ_BUSY_CYCLES'))},
    'note': 'rocprofv3 per-launch averages; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (raw value and the '
            'calibration on a known 8-byte stream kept beside it); see profiles/%s_c3_pmc.txt' % R,
}
json.dump(out, open(os.path.join(dst, '%s_c3_pmc.json' % R), 'w'), indent=1)
print(json.dumps(out, indent=1))
