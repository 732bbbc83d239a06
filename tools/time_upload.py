"""Where the time of one block upload goes (developer tool): python tools/time_upload.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opendrift_amd import synthetic as synth
from opendrift_amd.device import Context
U, V, W = 'x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity'
KZ, DEPTH, LAND = 'ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level', 'land_binary_mask'
names = [U, V, W, KZ, DEPTH, LAND]
g = synth.grid3d(nx=1024, ny=1024, nz=12, nt=2, seed=0)
ctx = Context(device=0, seed=0)
sid = ctx.add_grid(g['x'], g['y'], z=g['z'])
host = {k: np.ascontiguousarray(g[k][0]) for k in names}
mb = sum(a.nbytes for a in host.values()) / 1e6
def t(fn, reps=3):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3
print('block %.0f MB' % mb)
print('sync upload, pageable host   %.2f ms' % t(lambda: ctx.upload_block(sid, 0, 0.0, host)))
pin = {k: ctx.pin(a) for k, a in host.items()}
print('sync upload, registered host %.2f ms' % t(lambda: ctx.upload_block(sid, 0, 0.0, pin)))
dev = {k: torch.from_numpy(a).cuda() for k, a in host.items()}
torch.cuda.synchronize()
ptrs = {k: d.data_ptr() for k, d in dev.items()}
nz = {k: (host[k].shape[0] if host[k].ndim == 3 else 1) for k in names}
print('device-resident source       %.2f ms (preparation kernels only)' % t(lambda: ctx.upload_block_device(sid, 0, 0.0, ptrs, nz)))
ph = {k: torch.from_numpy(a).pin_memory() for k, a in host.items()}
t0 = time.perf_counter()
for k in names: dev[k].copy_(ph[k], non_blocking=True)
torch.cuda.synchronize()
print('torch pinned H2D of the raw arrays %.2f ms' % ((time.perf_counter() - t0) * 1e3))
