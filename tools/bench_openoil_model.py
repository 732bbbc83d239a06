#!/usr/bin/env python
"""OpenOil.run() through the model API at C4 size on one MI355X: NorKyst-800-shaped 2602x902 polar-stereographic
surface block (current, wind, Stokes drift, landmask), 6.25 M oil elements, RK4 + windage + Stokes drift + horizontal
diffusion + stranding, and OpenOil's vertical mixing with the oil physics on the device (droplet rise velocities,
slick, wave entrainment; Large et al. 1994 profiles), wind / current uncertainty as OpenOil's defaults have them.

    python tools/bench_openoil_model.py [--particles N] [--steps K] [--no-mixing]

One JSON line: ms per step of the whole loop body (release, environment, coastline, age, compaction, result buffer,
update) and particle-steps/s.
"""
import argparse
import json
import os
import sys
import time
from datetime import datetime, timedelta

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--particles', type=int, default=6_250_000)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--no-mixing', action='store_true')
    a = ap.parse_args()
    import __graft_entry__ as G
    G.build()
    from opendrift_amd import readers, synthetic as synth
    from opendrift_amd.openoil import OpenOil
    from opendrift_amd.projection import stere_polar_inverse
    g = synth.grid_stere(nt=12)    # hourly levels: 11 h of fields
    t0 = datetime(2020, 1, 1)
    times = [t0 + timedelta(seconds=float(t)) for t in g['t']]
    names = [k for k in g if k not in ('x', 'y', 't')]
    o = OpenOil(loglevel=50, seed=0)
    o.add_reader(readers.GridReader(g['x'], g['y'], times, {k: g[k] for k in names}, proj4=synth.NORKYST_PROJ4))
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    o.set_config('drift:vertical_mixing', not a.no_mixing)
    o.set_config('environment:constant:horizontal_diffusivity', 10)
    o.set_config('general:coastline_action', 'stranding')
    n = a.particles
    rng = np.random.default_rng(0)
    x = rng.uniform(g['x'][8], g['x'][int(0.9 * len(g['x']))], n)
    y = rng.uniform(g['y'][8], g['y'][-9], n)
    lon, lat = stere_polar_inverse(x, y, **synth.NORKYST_PROJ)
    o.seed_elements(lon=lon, lat=lat, z=0.0, time=t0, oil_type={'density': 900.0, 'viscosity': 0.005,
                                                                'oil_water_interfacial_tension': 0.03})
    t_start = time.perf_counter()
    if os.environ.get('ODR_PROFILE'):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
    o.run(time_step=900, steps=a.steps, time_step_output=900 * a.steps, export_variables=['lon', 'lat', 'z', 'status'])
    o.ctx.sync()
    el = time.perf_counter() - t_start
    if os.environ.get('ODR_PROFILE'):
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats('cumulative').print_stats(32)
    e = o.elements
    print(json.dumps({
        'metric': 'particle-steps/s through OpenOil.run() (model API, whole loop body)', 'particles': n, 'steps': a.steps,
        'vertical_mixing_with_oil_physics': not a.no_mixing, 'ms_per_step_including_setup': 1e3 * el / a.steps,
        'timing': {k: v for k, v in getattr(o, 'timing', {}).items()},
        'value': n * a.steps / el, 'unit': 'particle-steps/s', 'active_at_end': int(o.num_elements_active()),
        'stranded': int(o.num_elements_deactivated()), 'share_at_surface': float((e.z == 0).mean()),
        'z_min': float(e.z.min()) if len(e.z) else None, 'status_categories': list(getattr(o, 'status_categories', []))}))


if __name__ == '__main__':
    main()
