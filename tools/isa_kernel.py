#!/usr/bin/env python
"""Instruction-class counts of single kernels in a hipcc -S listing: python tools/isa_kernel.py k.s prefix..."""
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
for pref in sys.argv[2:]:
    start = next((i for i, l in enumerate(txt) if l.startswith(pref) and ': ' in l and not l[0].isspace()), None)
    if start is None:
        print(pref, 'not found')
        continue
    body = []
    for l in txt[start + 1:]:
        body.append(l)
        if 's_endpgm' in l:
            break
    c = lambda p: sum(1 for l in body if re.match(r'^\s+' + p, l))
    print(pref[:44], 'valu', c('v_'), 'f64', sum(1 for l in body if re.match(r'^\s+v_\w*f64', l)), 'readlane', c('v_readlane'),
          'writelane', c('v_writelane'), 'lshl_add_u64', c('v_lshl_add_u64'), 'mad_u64', c('v_mad_u64_u32'), 'mul_lo',
          c('v_mul_lo_u32'), 'mul24', c('v_mul_u32_u24') + c('v_mad_u32_u24'), 'loads_saddr',
          sum(1 for l in body if re.match(r'^\s+global_load', l) and re.search(r', s\[\d+:\d+\]', l)), 'loads', c('global_load'),
          'branches', c('s_cbranch'))
