#!/usr/bin/env python
"""Measurement of the oil physics inside OpenOil's mixing loop on MI355X beside the NumPy
oracle (oracle/oil.py, 1 core).

    python tools/bench_oil.py [--particles N] [--reps R] [--cpu-particles M]

One step = odr_oil_prepare_mixing (element statistics, 1e6-point droplet spectrum + scan, np.random.choice lookup)
+ the oil variant of the mixing kernel (10 sub-steps of 60 s: terminal velocity, random walk with the Large et al.
1994 profile, slick, wave entrainment), device Philox numbers, wind / temperature / salinity environment already
sampled.  One JSON line: particle-mixing-steps/s, the time of the two parts, the same step without the oil physics
(OceanDrift's kernel) and the CPU baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
XW, YW, MLD, DEPTH, SSH = 'x_wind', 'y_wind', 'ocean_mixed_layer_thickness', 'sea_floor_depth_below_sea_level', 'sea_surface_height'
TEMP, SALT = 'sea_water_temperature', 'sea_water_salinity'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--particles', type=int, default=10_000_000)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--cpu-particles', type=int, default=200_000)
    a = ap.parse_args()
    import __graft_entry__ as G
    G.build()
    from opendrift_amd.device import Context, sea_water_density_default
    n = a.particles
    rng = np.random.default_rng(0)
    ctx = Context(device=0, seed=0)
    P = ctx.particles(n)
    z0 = np.where(rng.uniform(0, 1, n) < 0.5, 0.0, -rng.uniform(0, 40, n))
    P.append(rng.uniform(2, 8, n), rng.uniform(59, 63, n), z=z0)
    env = {XW: rng.uniform(4, 16, n), YW: rng.uniform(-4, 4, n), MLD: rng.uniform(20, 70, n), DEPTH: rng.uniform(80, 300, n),
           SSH: np.zeros(n), TEMP: rng.uniform(4, 12, n), SALT: rng.uniform(30, 35, n)}
    for k, v in env.items():
        P.env_upload(k, v.astype(np.float32))
    d0 = np.where(z0 < 0, rng.uniform(2e-5, 3e-3, n), 0.0).astype(np.float32)
    props = [d0, np.full(n, 900.0, np.float32), np.full(n, 0.005, np.float32), rng.uniform(5e-4, 1.5e-3, n).astype(np.float32)]
    rho_w = sea_water_density_default()

    def reset():
        P.upload(z=z0)
        for slot, v in enumerate(props):
            P.set_property(slot, v)

    def oil_step(k):
        P.oil_prepare_mixing(600.0, 60.0, 0.03, 'Johansen et al. (2015)', rho_w, step=k)
        P.vmix_analytic('windspeed_Large1994', 1.2e-5, 600.0, 60.0, step=k)

    reset()
    oil_step(0)
    ctx.sync()
    ctx.timer_begin()
    for k in range(a.reps):
        P.oil_prepare_mixing(600.0, 60.0, 0.03, 'Johansen et al. (2015)', rho_w, step=k + 1)
    ms_prepare = ctx.timer_end() / a.reps
    ctx.timer_begin()
    for k in range(a.reps):
        oil_step(k + 1)
    ms_step = ctx.timer_end() / a.reps
    zz = P.download()['z']
    share_surface = float((zz == 0).mean())
    reset()
    P.vmix_analytic('windspeed_Large1994', 1.2e-5, 600.0, 60.0, step=0)
    ctx.timer_begin()
    for k in range(a.reps):
        P.vmix_analytic('windspeed_Large1994', 1.2e-5, 600.0, 60.0, step=k + 1)
    ms_plain = ctx.timer_end() / a.reps
    # CPU baseline: the NumPy oracle on a bounded sample, 1 core
    from oracle import diffusivity, oil
    m = a.cpu_particles
    e = {k: v[:m].astype(np.float32) for k, v in env.items()}
    z, d = z0[:m].copy(), d0[:m].copy()
    rho, nu = np.full(m, 900.0), np.full(m, float(np.float32(0.005)))
    t0 = time.perf_counter()
    T = e[TEMP] + np.float32(273.15)
    hs = oil.significant_wave_height(e[XW], e[YW])
    prob = oil.entrainment_probability(rho, nu, 0.03, hs, oil.wave_breaking_fraction(e[XW], e[YW]), 60.0)
    dv = oil.droplet_median_johansen2015(rho, nu, props[3][:m], hs, 0.03)
    dif = oil.droplet_diameters(dv, rng.uniform(0, 1, m))
    zlev, Kp = diffusivity.profiles('windspeed_Large1994', e[XW], e[YW], e[MLD], 1.2e-5)
    u = rng.uniform(0, 1, (3, 10, m))
    oil.vertical_mixing_oil(z, np.ones(m, np.int32), d, rho, T, e[SALT], e[DEPTH], e[SSH], zlev, Kp, 600.0, 60.0, prob, dif,
                            np.mean(1.5 * hs), u[0], u[1], u[2])
    t_cpu = time.perf_counter() - t0
    print(json.dumps({
        'metric': 'particle-mixing-steps/s (OpenOil: prepare_vertical_mixing + 10 sub-steps with oil physics)',
        'value': n / (ms_step * 1e-3), 'unit': 'particle-steps/s', 'particles': n, 'ms_per_step': ms_step,
        'ms_prepare_vertical_mixing': ms_prepare, 'ms_same_step_without_oil_physics': ms_plain,
        'share_at_surface_after': share_surface, 'dtype': 'f64/f32', 'data': 'synthetic',
        'cpu_baseline': {'value': m / t_cpu, 'unit': 'particle-steps/s', 'cores': 1, 'kind': 'port',
                         'sample': '%d particles x 1 step (oracle/oil.py NumPy restatement, %.1f s)' % (m, t_cpu)}}))


if __name__ == '__main__':
    main()
