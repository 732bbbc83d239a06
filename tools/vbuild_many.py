#!/usr/bin/env python
"""Several A/B libraries in one go: the default build once, then every variant's translation units in parallel.

  tools/vbuild_many.py TAG:unit.hip[+unit.hip]:-DFLAG[,-DFLAG...] [TAG:...]

-> tools/_lib<TAG>.so each (selected with ODR_LIB on the GPU box; tools/gpu_ab.sh).  (tools/vbuild_tu.sh builds one variant; two of
them started side by side both rebuild the default objects first and race on them.)"""
import sys
import os
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opendrift_amd import build as b

b.build()
specs = []
for a in sys.argv[1:]:
    tag, units, flags = (a.split(':') + ['', ''])[:3]
    specs.append((tag, units.split('+'), [f for f in flags.split(',') if f]))
orig = b.build
b.build = lambda *a, **k: b.LIB            # the default objects are fresh: the variants must not start it again
with ThreadPoolExecutor(int(os.environ.get('ODR_VBUILD_JOBS', 4))) as ex:
    for lib in ex.map(lambda s: b.build_variant(s[0], s[1], s[2]), specs):
        print(lib)
