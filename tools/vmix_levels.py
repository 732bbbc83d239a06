"""MI355X: odr_vmix on readers with 8 .. 40 levels, whole-column kernel (ODR_VMIX_WINDOW=0) against the five-level window
kernel -- decides from how many levels on the window kernel runs (csrc/odr_mix.hip).  4 M particles, sorted by cell."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opendrift_amd.device import Context

U, V = 'x_sea_water_velocity', 'y_sea_water_velocity'
W, KZ = 'upward_sea_water_velocity', 'ocean_vertical_diffusivity'
DEPTH, SSH = 'sea_floor_depth_below_sea_level', 'sea_surface_height'


def field(nz):
    nx, ny, nt = 512, 384, 2
    x = np.linspace(0.0, 10.0, nx).astype(np.float32)
    y = np.linspace(60.0, 66.0, ny).astype(np.float32)
    z = -np.concatenate([[0.0], np.cumsum(np.linspace(3.0, 3.0 + 400.0 / nz, nz - 1))])
    X, Y = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    g = dict(x=x, y=y, z=z, t=3600.0 * np.arange(nt))
    K = np.empty((nt, nz, ny, nx), np.float32)
    for it in range(nt):
        for k in range(nz):
            K[it, k] = 1e-2 * np.exp(z[k] / 30.0) * (1 + 0.5 * np.sin(3 * X + 2 * Y + 0.3 * it)) + 1e-5
    g[KZ] = K
    g[W] = np.zeros_like(K)
    g[DEPTH] = np.full((nt, ny, nx), 500.0, np.float32)
    return g


def run(nz, window, n=4_000_000, reps=12):
    os.environ['ODR_VMIX_WINDOW'] = '1' if window else '0'
    g = field(nz)
    ctx = Context(seed=1)
    sid = ctx.add_grid(g['x'], g['y'], z=g['z'])
    for k in range(2):
        ctx.upload_block(sid, k, float(g['t'][k]), {nm: g[nm][k] for nm in (W, KZ, DEPTH)})
    for nm in (W, KZ, DEPTH):
        ctx.bind(nm, [sid], 0.0)
    ctx.bind(SSH, [], 0.0)
    rng = np.random.default_rng(2)
    P = ctx.particles(n)
    P.append(rng.uniform(1, 9, n), rng.uniform(60.5, 65.5, n), z=-rng.uniform(0, 60, n))
    P.sort_by_cell(sid)
    P.env_sample([W, DEPTH, SSH], 1800.0)
    for k in range(3):
        P.vmix(1800.0, 600.0, 60.0, step=k)
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(reps):
        P.vmix(1800.0, 600.0, 60.0, step=10 + k)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    d = P.download()
    zsum = d['z'][np.argsort(d['ID'])]
    P.close()
    ctx.close()
    return dt * 1e3, zsum


if __name__ == '__main__':
    for nz in (8, 12, 16, 24, 40):
        a, za = run(nz, False)
        b, zb = run(nz, True)
        print('levels %3d   column kernel %.3f ms   window kernel %.3f ms   same result: %s' % (nz, a, b, bool(np.array_equal(za, zb))), flush=True)
