#!/usr/bin/env python
"""RK4 current advection on a Lambert-conformal reader (the C4 fields re-labelled as an lcc grid, WGS84): the launch of
odr_env_coast_advect with the FAST stage arithmetic.   python tools/bench_lcc.py [particles]
A/B: ODR_LIB=tools/_libX.so built with -DODR_NO_STAGE_ROT_CLOSED_FORM (the stage rotation from rotation_angle, rounds 3-4)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from opendrift_amd import projection  # noqa: E402
from opendrift_amd.device import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
fields = bench.make_fields('c4')
g = fields['g']
x, y = np.asarray(g['x'], np.float64), np.asarray(g['y'], np.float64)
proj = dict(kind='lcc', a=6378137.0, rf=298.257223563, lat0=63.3, lon0=15.0, lat1=63.3, lat2=63.3, k0=1.0,
            x0=0.5 * (x[0] + x[-1]), y0=0.5 * (y[0] + y[-1]))
rng = np.random.default_rng(0)
px = rng.uniform(x[0] + 0.1 * (x[-1] - x[0]), x[-1] - 0.1 * (x[-1] - x[0]), n)
py = rng.uniform(y[0] + 0.1 * (y[-1] - y[0]), y[-1] - 0.1 * (y[-1] - y[0]), n)
lon, lat = projection.lcc_inverse(px, py, **{k: v for k, v in proj.items() if k != 'kind'})
U, V, LAND = 'x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask'
for math in ('fast', 'exact'):
    ctx = Context(0, seed=0)
    ctx.set_stage_math(math)
    sid = ctx.add_grid(x, y, proj=proj)
    for slot in range(3):
        ctx.upload_block(sid, slot, float(g['t'][slot]), {k: g[k][slot] for k in (U, V, LAND)})
    for k in (U, V):
        ctx.bind(k, [sid], 0.0)
    ctx.bind(LAND, [sid], np.nan)
    P = ctx.particles(n)
    P.append(lon, lat)
    P.sort_by_cell(sid, keep_environment=False)
    ms = []
    for k in range(12):
        ctx.timer_begin()
        P.env_coast_advect([U, V, LAND], 600.0 * (k % 5), 'runge-kutta4', 600.0, coastline='previous', store_previous=True, count=False)
        ms.append(ctx.timer_end())
    print('lcc reader, %d particles, RK4 step launch, stage math %s: %.4f ms (median of %d)' % (n, math, float(np.median(ms[2:])), len(ms) - 2))
    P.close()
    ctx.close()
