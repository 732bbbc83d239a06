#!/usr/bin/env python
"""Times the fused RK4 step on the C4-shaped surface fields when the reader has NO projection (2D lon/lat nodes, the
Delaunay lookup of csrc/odr_mesh.h + curvi_locate) next to the same fields read through their polar-stereographic
projection; also the bare lookup (odr_source_lonlat2xy's kernel is not timed separately: host copies dominate it)
and the host preparation of the triangulation.  GPU only.

    python tools/bench_curvilinear.py [--particles 6250000] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opendrift_amd import synthetic as synth          # noqa: E402
from opendrift_amd.device import Context              # noqa: E402
from opendrift_amd.projection import stere_polar_inverse  # noqa: E402

U, V, LAND = 'x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask'


def run(ctx, sid, g, lon, lat, steps, dt=900.0):
    for slot in range(3):
        ctx.upload_block(sid, slot, float(g['t'][slot]), {k: g[k][slot] for k in (U, V, LAND)})
    for k in (U, V):
        ctx.bind(k, [sid], 0.0)
    ctx.bind(LAND, [sid], np.nan)
    P = ctx.particles(len(lon))
    P.append(lon, lat, z=np.zeros(len(lon)))
    P.sort_by_cell(sid)
    out = []
    for rep in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        for k in range(steps):
            P.env_coast_advect([U, V, LAND], (k * dt) % 6000.0, 'runge-kutta4', dt, coastline='previous',
                               store_previous=True, count=False)
        ctx.sync()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    return min(out), len(P)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--particles', type=int, default=6_250_000)
    ap.add_argument('--steps', type=int, default=20)
    a = ap.parse_args()
    g = synth.grid_stere()
    rng = np.random.default_rng(0)
    x = rng.uniform(g['x'][8], g['x'][int(0.9 * len(g['x']))], a.particles)
    y = rng.uniform(g['y'][8], g['y'][-9], a.particles)
    lon, lat = stere_polar_inverse(x, y, **synth.NORKYST_PROJ)
    res = {'particles': a.particles, 'grid': [len(g['y']), len(g['x'])]}

    ctx = Context(seed=0)
    sid = ctx.add_grid(g['x'], g['y'], proj=synth.NORKYST_PROJ)
    res['projected_ms_per_step'], _ = run(ctx, sid, g, lon, lat, a.steps)
    del ctx

    X, Y = np.meshgrid(g['x'].astype(np.float64), g['y'].astype(np.float64))
    lon2d, lat2d = stere_polar_inverse(X, Y, **synth.NORKYST_PROJ)
    ctx = Context(seed=0)
    t0 = time.perf_counter()
    sid = ctx.add_grid_curvilinear(lon2d, lat2d)
    res['triangulation_build_s'] = time.perf_counter() - t0
    res['curvilinear_ms_per_step'], _ = run(ctx, sid, g, lon, lat, a.steps)
    t0 = time.perf_counter()
    qx, qy = ctx.lonlat2xy(sid, lon[:1_000_000], lat[:1_000_000])
    res['lonlat2xy_1M_host_roundtrip_s'] = time.perf_counter() - t0
    # the lookup inverts the mesh: pixel coordinates of the seeding positions (piecewise-linear vs the projection)
    xi = (x[:1_000_000] - g['x'][0]) / 800.0
    yi = (y[:1_000_000] - g['y'][0]) / 800.0
    res['max_pixel_difference_to_projection'] = float(max(np.nanmax(np.abs(qx - xi)), np.nanmax(np.abs(qy - yi))))
    res['not_located'] = int(np.isnan(qx).sum())
    print(json.dumps(res))


if __name__ == '__main__':
    main()
