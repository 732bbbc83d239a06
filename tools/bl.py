#!/usr/bin/env python
"""Compact view of a bench.py JSON line on stdin:  python bench.py ... | tail -1 | python tools/bl.py [tag]"""
import json
import sys

d = json.loads(sys.stdin.read())
r = d['roofline']
out = [sys.argv[1] if len(sys.argv) > 1 else '', d['config']['workload'][:2], 'steps %d' % d['steps'], 'ms/step %.4f' % d['ms_per_step'],
       'kernel_ms %.4f (median %.4f)' % (r['kernel_ms'], r.get('kernel_ms_median', float('nan')))]
if 'second_kernel' in r:
    out.append('k2_ms %.4f' % r['second_kernel']['kernel_ms'])
for k in ('pcie_inclusive', 'model_api'):
    if k in d:
        out.append('%s %.3f' % (k, d[k]['ms_per_step']))
print(' '.join(out))
