#!/usr/bin/env python
"""Copy the summaries tools/gpu_profile_round.sh left under gpurun_out/prof_<R>[_exact] into profiles/ (tracked) and derive
profiles/<R>_<workload>_pmc[_exact].json, which bench.py reads for the roofline objects (PMC counters cannot be collected from
inside the bench process).

    [R=r06] python tools/collect_profiles_round.py [stage_math]

FETCH_SIZE: MI355X_MICROARCH.md (HBM section): gfx950's rocprofv3 reports 1/2 of the bytes of a coalesced streaming read; the
factor is calibrated on this code's own 8-byte streams (k_sort_hist reads lon and lat of every particle once) and stored
next to the doubled figure.  Cycle pricing of a VALU wave-instruction: profiles/r03_rate_bench_raw.txt (float64 arithmetic and
conversions 4, transcendental float64 16, everything else 2 cycles per SIMD)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.environ.get('R', 'r06')
SM = sys.argv[1] if len(sys.argv) > 1 else 'fast'
SFX = '' if SM == 'fast' else '_' + SM
src, dst = os.path.join(ROOT, 'gpurun_out', 'prof_' + R + SFX), os.path.join(ROOT, 'profiles')
N = {'c3': 10_000_000, 'c4': 6_250_000, 'c5': 10_000_000}
DOMINANT = {'c3': 'k_step_grid<2, 0, true', 'c4': 'k_step_grid<2, 2, false', 'c5': 'k_step_leeway<2>'}   # (either stage math: the template's SM argument comes last)
SECOND = {'c3': 'k_vmix_col<3, true'}
for w in ('c3', 'c4', 'c5', 'c3_model_api'):
    f = os.path.join(src, '%s_%s_kernel_stats.txt' % (R, w))
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, '%s_%s_kernel_stats%s.txt' % (R, w, SFX)))
f = os.path.join(src, '%s_c3_model_api_host_profile.txt' % R)
if os.path.exists(f):
    keep = [ln for ln in open(f).read().splitlines() if not ln.startswith('W2') and 'rocprofv3' not in ln][:70]
    open(os.path.join(dst, '%s_c3_model_api_host_profile.txt' % R), 'w').write('\n'.join(keep) + '\n')


def parse(raw):
    vals = {}
    for line in raw.splitlines():
        m = re.match(r'^(.*?)\s+(\w+)\s+n=(\d+)\s+avg=([\d.e+-]+)', line)
        if m:
            vals.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(4))
    return vals


def kernel(vals, pat):
    for name, d in vals.items():
        if pat in name:
            return name, d
    return None, {}


def avg_us(stats_file, pat):
    for ln in open(stats_file).read().splitlines():
        if pat in ln:
            p = ln.split()
            return float(p[-2])
    return None


def block(d, n, us):
    fs = d.get('FETCH_SIZE', float('nan')) * 1024
    f64 = sum(d.get(k, 0) for k in ('SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_CVT'))
    tr = d.get('SQ_INSTS_VALU_TRANS_F64', 0)
    valu = d.get('SQ_INSTS_VALU', 0)
    return {
        'particles': n, 'kernel_us_rocprof': us,
        'FETCH_SIZE_bytes_raw': fs, 'FETCH_SIZE_bytes_x2': 2 * fs, 'WRITE_SIZE_bytes': d.get('WRITE_SIZE', float('nan')) * 1024,
        'SQ_WAVES': d.get('SQ_WAVES'), 'SQ_INSTS_VALU': valu, 'SQ_INSTS_SALU': d.get('SQ_INSTS_SALU'),
        'SQ_INSTS_SMEM': d.get('SQ_INSTS_SMEM'), 'SQ_INSTS_VMEM': d.get('SQ_INSTS_VMEM'), 'SQ_INSTS_LDS': d.get('SQ_INSTS_LDS'),
        'valu_f64_class': f64, 'valu_trans_f64': tr, 'valu_other': valu - f64 - tr,
        'valu_cycles_per_simd_slot': 4 * f64 + 16 * tr + 2 * (valu - f64 - tr),
        'valu_per_wave': valu / d['SQ_WAVES'] if d.get('SQ_WAVES') else None,
        'TCP_TOTAL_CACHE_ACCESSES': d.get('TCP_TOTAL_CACHE_ACCESSES_sum'), 'TA_TA_BUSY': d.get('TA_TA_BUSY_sum'),
        'TCP_TCC_READ_REQ': d.get('TCP_TCC_READ_REQ_sum'), 'TCC_HIT': d.get('TCC_HIT_sum'), 'TCC_MISS': d.get('TCC_MISS_sum'),
        'GRBM_GUI_ACTIVE_8xcd': d.get('GRBM_GUI_ACTIVE'),
        'SQ_WAVE_CYCLES': d.get('SQ_WAVE_CYCLES'), 'SQ_WAIT_ANY': d.get('SQ_WAIT_ANY'), 'SQ_WAIT_INST_ANY': d.get('SQ_WAIT_INST_ANY'),
        'SQ_ACTIVE_INST_VALU': d.get('SQ_ACTIVE_INST_VALU'), 'SQ_BUSY_CU_CYCLES': d.get('SQ_BUSY_CU_CYCLES'),
        'SQC_ICACHE_REQ': d.get('SQC_ICACHE_REQ'), 'SQC_ICACHE_HITS': d.get('SQC_ICACHE_HITS'), 'SQC_ICACHE_MISSES': d.get('SQC_ICACHE_MISSES'),
    }


for w in ('c3', 'c4', 'c5'):
    rawf = os.path.join(src, '%s_%s_pmc_raw.txt' % (R, w))
    stf = os.path.join(src, '%s_%s_kernel_stats.txt' % (R, w))
    if not os.path.exists(rawf):
        continue
    raw = open(rawf).read()
    vals = parse(raw)
    name, d = kernel(vals, DOMINANT[w])
    if not d:
        print('no counters of', DOMINANT[w], 'for', w)
        continue
    _, sh = kernel(vals, 'k_sort_hist')
    calib = sh.get('FETCH_SIZE', float('nan')) * 1024 / (16.0 * N[w]) if sh else None
    out = dict(workload=w, stage_math=SM, kernel=name[:60], **block(d, N[w], avg_us(stf, DOMINANT[w])))
    out['FETCH_SIZE_calibration_8B_stream'] = calib
    if w in SECOND:
        n2, d2 = kernel(vals, SECOND[w])
        if d2:
            out['second'] = dict(kernel=n2[:60], **block(d2, N[w], avg_us(stf, SECOND[w])))
    # HBM bytes of ALL kernels of one step: every kernel's bytes per launch (counters) x its launches per step (kernel trace of
    # a 32-step run; the re-sort kernels run every 16th step) -- bench.py's roofline_step
    calls = {}
    for ln in open(stf).read().splitlines()[1:]:
        q = ln.split()
        if len(q) >= 5 and q[-4].isdigit():
            calls[ln[:110].strip()] = int(q[-4])
    nstep = calls.get(next((k for k in calls if DOMINANT[w] in k), ''), 0)
    per_step, total = {}, 0.0
    if nstep:
        for kname, d2 in vals.items():
            cnt = next((c for k, c in calls.items() if k[:60] == kname[:60]), 0)
            if not cnt or 'FETCH_SIZE' not in d2:
                continue
            byt = (2 * d2['FETCH_SIZE'] + d2.get('WRITE_SIZE', 0.0)) * 1024 * cnt / nstep
            short = re.sub(r'\(.*', '', kname).replace('void odr::', '').replace('odr::', '')[:48]
            per_step[short] = per_step.get(short, 0.0) + byt
            total += byt
        out['step_hbm_bytes'] = total
        out['step_hbm_kernels'] = {k: round(v / 1e6, 1) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])[:8]}
    out['note'] = ('rocprofv3 per-launch averages of `bench.py --workload %s --steps 6 --warmup 2 --no-cpu --no-extras` (stage math %s), '
                   'one --pmc pass per counter set; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; see '
                   'profiles/%s_%s_pmc%s.txt' % (w, SM, R, w, SFX))
    hdr = ('PMC counters of the %s bench (per dispatch, summed over dimensions; avg over the dispatches of a kernel), MI355X, stage math %s:\n'
           'rocprofv3 --kernel-trace --pmc <set> -- python bench.py --workload %s --steps 6 --warmup 2 --no-cpu --no-extras\n'
           'one pass per set (tools/gpu_profile_round.sh).  FETCH_SIZE / WRITE_SIZE in KiB per dispatch; SQ cycle counters in quad-cycles\n'
           'summed over waves; GRBM_GUI_ACTIVE summed over the 8 XCDs.  FETCH_SIZE calibration on k_sort_hist (16 B per particle): %s\n'
           % (w, SM, w, ('%.3f' % calib) if calib else 'n/a'))
    open(os.path.join(dst, '%s_%s_pmc%s.txt' % (R, w, SFX)), 'w').write(hdr + raw)
    json.dump(out, open(os.path.join(dst, '%s_%s_pmc%s.json' % (R, w, SFX)), 'w'), indent=1)
    print(w, json.dumps({k: out[k] for k in ('kernel', 'kernel_us_rocprof', 'valu_per_wave', 'TCP_TOTAL_CACHE_ACCESSES', 'FETCH_SIZE_bytes_x2', 'WRITE_SIZE_bytes')}))
