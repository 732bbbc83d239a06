#!/usr/bin/env python
"""One-off check for profiles/r06_ab_variants.txt: the directly evaluated Philox blocks + the range-limited Box-Muller
(rng_block / rng_normal2 / rng_uniform2, csrc/odr_kernels.hip.h) hand out the numbers of rocRAND's state object
(-DODR_RNG_ROCRAND build, tools/_librocrand.so).  Runs the same Leeway steps (current / wind uncertainty, jibing) and
horizontal-diffusion steps with the device generator under both libraries (one subprocess each) and compares.

  python tools/vbuild_many.py rocrand:odrift.hip+odr_step_noise.hip+odr_step_fast_noise.hip+odr_mix.hip:-DODR_RNG_ROCRAND   (here)
  python tools/check_rng_equiv.py                 (on the GPU box)
"""
import os
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def run(out):
    from opendrift_amd import synthetic as synth
    from opendrift_amd.device import Context
    from opendrift_amd.projection import stere_polar_inverse
    U, V, LAND = 'x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask'
    XW, YW, HD = 'x_wind', 'y_wind', 'horizontal_diffusivity'
    g = synth.grid_stere(nx=260, ny=90, nt=3, seed=0)
    names = [U, V, XW, YW, LAND]
    ctx = Context(seed=3)
    sid = ctx.add_grid(g['x'], g['y'], proj=synth.NORKYST_PROJ)
    for k in range(3):
        ctx.upload_block(sid, k, float(g['t'][k]), {nm: g[nm][k] for nm in names})
    for nm in names:
        ctx.bind(nm, [sid], np.nan if nm == LAND else 0.0)
    cs = ctx.add_constant({HD: 10.0})
    ctx.bind(HD, [cs], 0.0)
    rng = np.random.default_rng(4)
    n = 200000
    x = rng.uniform(g['x'][4], g['x'][-5], n)
    y = rng.uniform(g['y'][4], g['y'][-5], n)
    lon, lat = stere_polar_inverse(x, y, **synth.NORKYST_PROJ)
    P = ctx.particles(n)
    P.append(lon, lat)
    r = np.random.default_rng(9)
    ori = (np.arange(n) % 2).astype(np.float32)
    for slot, val in enumerate([np.full(n, 0.96), np.where(ori == 0, 0.54, -0.54), np.zeros(n), np.zeros(n),
                                np.abs(r.standard_normal(n)) * 12.0, r.standard_normal(n) * 9.4, np.full(n, 0.04), ori, np.zeros(n)]):
        P.set_property(slot, val.astype(np.float32))
    res = {}
    for k in range(4):
        P.env_coast_leeway([XW, YW, U, V, LAND], 300.0 + 600.0 * k, 600.0, 0.4, coastline='none', current_uncertainty=0.1,
                           wind_uncertainty=2.0, step=k)
        P.env_sample([HD], 0.0)
        P.hdiffusion(600.0, step=k)
        if k == 0:      # the perturbed environment of the first step: the normal pairs themselves, rounded into float32
            for v in (U, V, XW, YW):
                res['env_' + v] = P.env_download(v)
    d = P.download()
    o = np.argsort(d['ID'])
    res.update(lon=d['lon'][o], lat=d['lat'][o], slope=P.get_property(1)[o])
    np.savez(out, **res)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run(sys.argv[1])
        sys.exit(0)
    outs = []
    for tag, lib in (('default', None), ('rocrand', os.path.join(HERE, '_librocrand.so'))):
        env = dict(os.environ)
        if lib:
            env['ODR_LIB'] = lib
        out = '/tmp/rng_equiv_%s.npz' % tag
        subprocess.check_call([sys.executable, os.path.abspath(__file__), out], env=env, stdin=subprocess.DEVNULL)
        outs.append(np.load(out))
    a, b = outs
    for k in a.files:
        same = (a[k] == b[k]) | (np.isnan(a[k]) & np.isnan(b[k]))
        print('%-28s identical %.6f   worst |difference| %.3e' % (k, same.mean(), np.nanmax(np.abs(a[k].astype(np.float64) - b[k]))))
