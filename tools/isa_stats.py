#!/usr/bin/env python
"""Static instruction mix of the device kernels: hipcc -S for gfx950, then count VALU / SALU / memory
instructions, registers, scratch and LDS per kernel.  CPU-only (cross-compile)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ''
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'k.s')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
                               '-S', '--cuda-device-only', '-I', os.path.join(ROOT, 'include'),
                               os.path.join(ROOT, 'opendrift_amd/csrc', os.environ.get('ODR_TU', 'odr_step.hip')), '-o', out] + os.environ.get('ODR_FLAGS', '').split())
        txt = open(out).read()
    cur, stats, meta = None, collections.OrderedDict(), {}
    for line in txt.splitlines():
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1)
            stats[cur] = collections.Counter()
            continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith('.end_amdhsa_kernel') or t.startswith('.Lfunc_end'):
            cur = None if t.startswith('.Lfunc_end') else cur
            continue
        m = re.match(r'^(v_|s_|ds_|global_|buffer_|flat_|scratch_)(\w+)', t)
        if m:
            k = {'v_': 'valu', 's_': 'salu', 'ds_': 'lds', 'global_': 'vmem', 'buffer_': 'vmem', 'flat_': 'vmem',
                 'scratch_': 'scratch'}[m.group(1)]
            stats[cur][k] += 1
            if m.group(1) == 'v_' and ('f64' in t.split()[0]):
                stats[cur]['v_f64'] += 1
    for m in re.finditer(r'\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', txt, re.S):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    for m in re.finditer(r'\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+)', txt, re.S):
        meta[m.group(2)] = meta.get(m.group(2), (0, 0)) + (int(m.group(1)), int(m.group(3)))
    print('%-70s %6s %6s %6s %5s %5s %5s  %s' % ('kernel', 'valu', 'v_f64', 'salu', 'vmem', 'lds', 'scr', 'sgpr,vgpr,lds_bytes,scratch_bytes'))
    for k, c in stats.items():
        if pat and pat not in k:
            continue
        if not c:
            continue
        d = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
        print('%-70s %6d %6d %6d %5d %5d %5d  %s' % (d[:70], c['valu'], c['v_f64'], c['salu'], c['vmem'], c['lds'],
                                                    c['scratch'], meta.get(k, '')))


if __name__ == '__main__':
    main()
