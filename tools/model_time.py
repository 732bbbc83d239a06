#!/usr/bin/env python
"""OceanDrift.run() on the C3 inputs of bench.py under cProfile: where the host time of the loop goes.
    python tools/model_time.py [particles] [steps]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
fields = bench.make_fields('c3')
if os.environ.get('ODR_MODEL_NOPROFILE'):     # (the host profile slows the loop it measures: the kernel trace is taken without it)
    r = bench.model_api_leg(fields, n, steps, 0)
    print({k: v for k, v in r.items() if k != 'what'})
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
r = bench.model_api_leg(fields, n, steps, 0)
pr.disable()
print({k: v for k, v in r.items() if k != 'what'})
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
