#!/usr/bin/env python
"""Registers, scratch, code size and instruction classes of the kernels in a hipcc -S listing:
   python tools/isa_regs.py k.s [name-substring]"""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
meta = {}
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', txt):
    meta[m.group(1)] = (int(m.group(2)), int(m.group(3)), int(m.group(4)))
lines = txt.splitlines()
for i, l in enumerate(lines):
    m = re.match(r'^(_Z\w+):', l)
    if not m or m.group(1) not in meta or pat not in m.group(1):
        continue
    name = m.group(1)
    body = []
    for ll in lines[i + 1:]:
        if ll.startswith('.Lfunc_end'):
            break
        body.append(ll.strip())
    ins = [b.split()[0] for b in body if re.match(r'^(v_|s_|ds_|global_|buffer_|flat_|scratch_)', b)]
    c = lambda p: sum(1 for x in ins if re.match(p, x))
    size = next((int(re.search(r'(\d+)', ll).group(1)) for ll in lines[i:i + 40000] if 'codeLenInByte' in ll), 0)
    sc, sg, vg = meta[name]
    short = re.sub(r'^_ZN3odr\d+', '', name)[:60]
    print('%-60s vgpr %3d sgpr %3d scratch %3d bytes %6d | valu %5d (f64 %4d cvt %3d pk %3d mov %3d cnd %3d) salu %4d (branch %3d waitcnt %3d) vmem %3d lds %3d' % (
        short, vg, sg, sc, size, c('v_'), c(r'v_\w*f64'), c('v_cvt'), c('v_pk_'), c('v_mov'), c('v_cndmask'), c('s_'), c('s_cbranch|s_branch'), c('s_waitcnt'),
        c('buffer_|global_|flat_'), c('ds_')))
