#!/usr/bin/env python
"""Measurement of the landmask raster source and coastline_crossing on MI355X beside the NumPy
oracle (oracle/landmask.py, 1 core).

    python tools/bench_landmask.py [--particles N] [--reps R]

A global 30 arc-second raster (43200 x 21600 cells; --nx 86400 gives the 15" resolution of the GSHHG bitmap behind roaring_landmask;
synthetic content: smooth pseudo-continents) is bit-packed into 117 MB (466 MB) of HBM.  Measured: land_binary_mask lookups/s
for elements clustered in a 10 x 6 degree coastal region (the C3/C4 situation) and spread over the globe, and the
crossing search for the elements that moved onto land (precision 0.001 deg, displacements of ~0.01 deg).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LAND = 'land_binary_mask'


def raster(nx, ny):
    """smooth pseudo-continents evaluated on a 10x coarser grid and repeated (the lookup cost does not depend on the
    content; evaluating four sines on 3.7e9 cells would take minutes on one core)"""
    cx, cy = nx // 10, ny // 10
    LO = np.radians(-180 + (np.arange(cx, dtype=np.float32) + 0.5) * (360.0 / cx))[None, :]
    LA = np.radians(-90 + (np.arange(cy, dtype=np.float32) + 0.5) * (180.0 / cy))[:, None]
    f = np.sin(3 * LO) * np.cos(2 * LA) + 0.6 * np.sin(7 * LO + 1) * np.sin(5 * LA) + 0.3 * np.sin(23 * LO) * np.cos(19 * LA) \
        + 0.05 * np.sin(211 * LO) * np.sin(173 * LA)
    coarse = (f > 0.35).astype(np.uint8)
    return np.repeat(np.repeat(coarse, 10, axis=0), 10, axis=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--particles', type=int, default=10_000_000)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--nx', type=int, default=43200)   # 30"; 86400 = the 15" GSHHG bitmap resolution (466 MB)
    ap.add_argument('--cpu-particles', type=int, default=1_000_000)
    a = ap.parse_args()
    import __graft_entry__ as G
    G.build()
    from opendrift_amd.device import Context
    from oracle import landmask
    nx, ny = a.nx, a.nx // 2
    t0 = time.perf_counter()
    cells = raster(nx, ny)
    t_gen = time.perf_counter() - t0
    d = 360.0 / nx
    ctx = Context(device=0, seed=0)
    t0 = time.perf_counter()
    sid = ctx.add_landmask(-180.0, -90.0, d, d, cells)
    t_up = time.perf_counter() - t0
    ctx.bind(LAND, [sid], np.nan)
    n = a.particles
    rng = np.random.default_rng(0)
    res = {}
    m = landmask.RasterMask(-180.0, -90.0, d, d, cells)
    # a 10 x 6 degree window with a coast in it (land share 25-50 %)
    per = int(round(1.0 / d))
    best = None
    for lat_w in range(-60, 60, 6):
        for lon_w in range(-180, 170, 10):
            blk = cells[(lat_w + 90) * per:(lat_w + 96) * per:16, (lon_w + 180) * per:(lon_w + 190) * per:16]
            if 0.25 < blk.mean() < 0.5:
                best = (lon_w, lat_w)
                break
        if best:
            break
    lon_w, lat_w = best
    for tag, lon, lat in (('regional', rng.uniform(lon_w, lon_w + 10, n), rng.uniform(lat_w, lat_w + 6, n)),
                          ('global', rng.uniform(-180, 180, n), rng.uniform(-80, 80, n))):
        P = ctx.particles(n)
        P.append(lon, lat)
        P.env_sample([LAND], 0.0)
        ctx.sync()
        ctx.timer_begin()
        for _ in range(a.reps):
            P.env_sample([LAND], 0.0)
        ms = ctx.timer_end() / a.reps
        got = P.env_download(LAND)
        k = a.cpu_particles
        t0 = time.perf_counter()
        want = m.land_binary_mask(lon[:k], lat[:k])
        t_cpu = time.perf_counter() - t0
        assert np.array_equal(got[:k], want)
        res[tag] = dict(ms=ms, lookups_per_s=n / (ms * 1e-3), land_share=float(got.mean()), cpu_lookups_per_s=k / t_cpu)
        if tag == 'regional':   # crossing search: previous position, then a displacement of ~0.01 deg
            P.store_previous()
            lon2, lat2 = lon + rng.uniform(0.002, 0.02, n), lat + rng.uniform(-0.01, 0.01, n)
            P.upload(lon=lon2, lat=lat2)
            P.env_sample([LAND], 0.0)
            hit_mask = P.env_download(LAND) == 1
            ctx.sync()
            ctx.timer_begin()
            hit = P.coastline_crossing('previous', 0.001, sid)
            ms_x = ctx.timer_end()
            kk = np.where(hit_mask)[0][:20000]
            t0 = time.perf_counter()
            lc, la = landmask.coastline_crossing(m, lon[kk], lat[kk], lon2[kk], lat2[kk], 0.001, land_side=False)
            t_cpu_x = time.perf_counter() - t0
            dd = P.download()
            assert np.array_equal(dd['lon'][kk], lc) and np.array_equal(dd['lat'][kk], la)
            res['crossing'] = dict(elements_on_land=int(hit), ms=ms_x, elements_per_s=hit / (ms_x * 1e-3),
                                   cpu_elements_per_s=len(kk) / t_cpu_x)
        P.close()
    print(json.dumps({'metric': 'land_binary_mask lookups/s on a %dx%d bit-packed raster (%.0f MB)' % (nx, ny, nx * ny / 8e6),
                      'particles': n, 'raster_upload_s': t_up, 'raster_generation_s': t_gen, **res,
                      'cpu_baseline': 'oracle/landmask.py (NumPy, 1 core) on %d lookups / 20000 crossings' % a.cpu_particles}))


if __name__ == '__main__':
    main()
