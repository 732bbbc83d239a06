#!/usr/bin/env python
"""bench.py -- particle-steps/s of the MI355X hot path on BASELINE.json's workloads.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5] [--particles P]

A "step" is one pass of the per-timestep hot path over all particles of this rank with the
inputs already resident in HBM (field blocks uploaded, particle SoA on device):
  c3 (default, "RK4, 3D interp"): spatial re-sort (every 16th step) -> ONE launch for Environment sample
      (u,v,w,depth,ssh,landmask) + coastline 'previous' + seafloor + age + previous state + RK4 advect_ocean_current (3 more 3D field
      evaluations + 4 geodesics) -> vertical_mixing (10 Visser sub-steps from the K profile) + vertical_advection;
      10 M particles, synthetic ROMS-shaped 1024x1024x12 z-level block, 2 time levels interpolated.
  c2: analytic double gyre, 1 M particles, RK4.
  c4: NorKyst-800-shaped 2602x902 polar-stereographic surface block (current, wind, Stokes,
      landmask), RK4 + wind + Stokes + horizontal diffusion + stranding + compaction, 6.25 M particles.
  c5: Leeway ensemble members on the same grid (wind / current uncertainty, stranding), 10 M particles.
One JSON line is printed by rank 0 (metric / roofline / cpu_baseline).  N>1: one process
per GPU (torchrun), particles sharded, field blocks broadcast once from rank 0 over RCCL; weak scaling (per-GPU
work fixed).  --block-every N [--block-async]: a new field time level arrives from host memory every N steps
inside the timed region (PCIe-inclusive rate).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

U, V, W = 'x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity'
KZ, DEPTH, SSH, LAND = ('ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level',
                        'sea_surface_height', 'land_binary_mask')
XW, YW = 'x_wind', 'y_wind'
SX, SY = 'sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity'
HD = 'horizontal_diffusivity'

# algorithmic bytes per particle
BYTES = {
    'c2': dict(step=48, advect=56),
    # fused = k_step_grid: environment sample of the group + coastline + previous + RK4
    'c3': dict(step=908, advect=3 * 128 + 56, fused=(128 + 64 + 32 + 8) + 112 + 3 * 128),   # state: + age r/w, ssh
    'c4': dict(step=436, advect=3 * 64 + 56, fused=(3 * 64 + 8) + 92 + 3 * 64),
    'c5': dict(step=220, advect=88, fused=220 - 8),   # fused = k_step_leeway: the whole step but the compaction's status scan
}
HBM_PEAK = 8.0e12
HBM_ACHIEVABLE = 6.3e12   # what a streaming kernel reaches on MI355X (/opt/skills/guides/MI355X_MICROARCH.md, HBM section)


def make_fields(workload, small=False):
    from opendrift_amd import synthetic as synth
    if workload == 'c3':
        n = (128, 96, 8) if small else (1024, 1024, 12)
        g = synth.grid3d(nx=n[0], ny=n[1], nz=n[2], nt=3, seed=0)
        names = [U, V, W, KZ, DEPTH, LAND]
        return dict(g=g, names=names, proj=None, z=g['z'])
    if workload in ('c4', 'c5'):
        n = (260, 90) if small else (2602, 902)
        g = synth.grid_stere(nx=n[0], ny=n[1], nt=3, seed=0)
        names = [U, V, XW, YW, SX, SY, LAND] if workload == 'c4' else [U, V, XW, YW, LAND]
        return dict(g=g, names=names, proj=synth.NORKYST_PROJ, z=None)
    return None


def seed_particles(workload, fields, n, rng):
    if workload == 'c2':
        from opendrift_amd.projection import stere_equit_sphere_inverse
        x = rng.uniform(0.05, 1.95, n)
        y = rng.uniform(0.05, 0.95, n)
        lon, lat = stere_equit_sphere_inverse(x, y, 6.371e6)
        return lon, lat, np.zeros(n)
    g = fields['g']
    if workload == 'c3':
        lon = rng.uniform(g['x'][8], g['x'][int(0.9 * len(g['x']))], n)
        lat = rng.uniform(g['y'][8], g['y'][-9], n)
        return lon, lat, -rng.uniform(0, 50, n)
    from opendrift_amd.projection import stere_polar_inverse
    x = rng.uniform(g['x'][8], g['x'][int(0.9 * len(g['x']))], n)
    y = rng.uniform(g['y'][8], g['y'][-9], n)
    lon, lat = stere_polar_inverse(x, y, **fields['proj'])
    return lon, lat, np.zeros(n)


class Workload:
    """Device-side step of one workload (the product path: opendrift_amd -> libodrift_hip.so)."""

    def __init__(self, name, ctx, fields, dist_info, via_torch=True):
        from opendrift_amd import distributed as D
        self.name, self.ctx, self.fields = name, ctx, fields
        # re-sort interval: C3 is flat between 12 and 24 steps (1.077 / 1.068 / 1.065 ms per step at 12 / 16 / 24) -> 24: the 0.9 ms
        # re-sort costs 0.038 ms per step; the Leeway members
        # of C5 drift at the surface without vertical shear and stay ordered for longer (0.917 -> 0.888 ms per step at 48)
        self.sort_every = int(os.environ.get('ODR_SORT_EVERY', {'c5': 48, 'c3': 24}.get(name, 16)))
        self.fused = not os.environ.get('ODR_UNFUSED')   # one launch for sample+coastline+previous+advect
        if name == 'c4' and self.fused and not os.environ.get('ODR_BENCH_REDUCE_PASS'):
            # the movers' global tests (calm / no Stokes drift / zero diffusivity / nothing at the surface) come out of the step
            # launch instead of out of a pass over the arrays (odr_ctx_set_step_reduce; ODR_BENCH_REDUCE_PASS=1: the pass, as in round 3)
            ctx.set_step_reduce(True, wind_drift_depth=0.1, relative_wind=False)
        self.scheme = os.environ.get('ODR_BENCH_SCHEME', 'runge-kutta4')   # what-if runs only: the metric is quoted on RK4
        rank, local_rank, world = dist_info
        if name == 'c2':
            sid = ctx.add_double_gyre(A=0.1, epsilon=0.25, omega=0.628, t0=0.0)
            ctx.bind(U, [sid], 0.0)
            ctx.bind(V, [sid], 0.0)
            self.dt, self.tmax = 0.1, 1e9
            return
        g = fields['g']
        sid = ctx.add_grid(g['x'], g['y'], z=fields['z'], proj=fields['proj'])
        self.sid = sid
        names2d = [k for k in fields['names'] if g[k].ndim == 3 and all(np.array_equal(g[k][0], g[k][j]) for j in range(1, g[k].shape[0]))]
        self.static_ids = {k: 1 + fields['names'].index(k) for k in names2d}
        for slot in range(3):
            if not via_torch:   # single process, no torch in it (tests/test_gpu_full_size.py): host arrays straight in
                assert world == 1
                ctx.upload_block(sid, slot, float(g['t'][slot]), {k: g[k][slot] for k in fields['names']}, content_ids=self.static_ids)
                continue
            arrays = {k: g[k][slot] for k in fields['names']} if rank == 0 else None
            shapes = {k: g[k][slot].shape for k in fields['names']}
            if D.backend() == 'rccl':
                # the C-ABI collectives (no torch in the process): ONE broadcast of the level on the upload stream, straight into
                # the staging memory of the block preparation (odr_block_broadcast), then the commit
                ctx.block_broadcast(sid, slot, float(g['t'][slot]), arrays, shapes, root=0, content_ids=self.static_ids)
                ctx.commit_block(sid, slot)
                continue
            # rank 0 owns the host Reader; the block travels to every GPU once per time level (RCCL broadcast), and
            # goes from the received device tensors into the block without touching the other hosts' memory
            tens = D.broadcast_block(arrays, shapes=shapes, src=0)
            import torch
            torch.cuda.synchronize()
            # (the synthetic reader's sea floor depth and land mask are the same array at every level, as a file reader's are:
            # declared with content ids, they are gathered at one of the two bracketing levels)
            ctx.upload_block_device(sid, slot, float(g['t'][slot]), {k: t.data_ptr() for k, t in tens.items()},
                                    {k: (t.shape[0] if t.dim() == 3 else 1) for k, t in tens.items()},
                                    content_ids=self.static_ids)
            del tens
        for k in fields['names']:
            ctx.bind(k, [sid], {LAND: np.nan, DEPTH: 10000.0}.get(k, 0.0))
        ctx.bind(SSH, [], 0.0)
        if name == 'c3':
            self.dt, self.dt_mix, self.tmax = 600.0, 60.0, 2 * 3600.0 - 600.0
            self.vars = [U, V, W, DEPTH, SSH, LAND]
        elif name == 'c5':
            self.dt, self.tmax = 600.0, 2 * 3600.0 - 600.0
            self.vars = [XW, YW, U, V, LAND]
        else:
            cs = ctx.add_constant({HD: 10.0})
            ctx.bind(HD, [cs], 0.0)
            self.dt, self.tmax = 900.0, 2 * 3600.0 - 900.0
            self.vars = [U, V, XW, YW, SX, SY, LAND, HD]

    def time_of(self, k):
        return (k * self.dt) % self.tmax if self.tmax < 1e8 else k * self.dt

    def step(self, P, k, mid=None):
        """One step of the workload.  mid: called where OceanDrift.run() reads the step's status scan -- behind the step launch,
        in front of the compaction and the launches of update() (the sharded loop starts its collective there and finishes it
        behind them; it has made the scan, so the compaction only applies it)."""
        t = self.time_of(k)
        mid_given = mid is not None
        compact = P.compact_apply if mid_given else P.compact
        mid = mid or (lambda *scan: None)      # mid(kept, flags) when the caller of mid has read the scan already, mid() otherwise
        if self.sort_every and self.name != 'c2' and k % self.sort_every == 0:
            P.sort_by_cell(self.sid, keep_environment=False)   # device layout maintenance, part of the timed step
        if self.name == 'c2':
            P.env_sample([U, V], t)
            P.advect('runge-kutta4', t, self.dt)
        elif self.name == 'c3':
            if self.fused and mid_given:   # the two launches as two calls, the status scan between them (OceanDrift.run())
                P.env_coast_advect(self.vars, t, self.scheme, self.dt, coastline='previous', store_previous=True,
                                   count=False, seafloor=True, age_dt=self.dt)
                if not os.environ.get('ODR_NO_SPECULATION') and P.scan_status_begin():
                    # as run() does between output times: the mixing launch enqueued behind the fold of the scan, guarded by its
                    # verdict "every element stays", the scan read afterwards
                    early = getattr(mid, 'early', None)
                    if early is not None:
                        early()                    # (C-ABI collectives: the step's collective leaves behind the fold of the scan)
                    ok = P.vmix(t, self.dt, self.dt_mix, step=k, fuse_vertical_advection=False, guarded=True)
                    kept, flags = P.scan_status_end()
                    mid(kept, flags)
                    if not (ok and kept == len(P)):
                        P.vmix(t, self.dt, self.dt_mix, step=k, fuse_vertical_advection=False)
                else:
                    mid()
                    P.vmix(t, self.dt, self.dt_mix, step=k, fuse_vertical_advection=False)
            elif self.fused:   # one call: the step launch, then the mixing launch
                P.env_coast_advect(self.vars, t, self.scheme, self.dt, coastline='previous', store_previous=True,
                                   count=False, seafloor=True, age_dt=self.dt,
                                   vmix=dict(dt_mix=self.dt_mix, step=k, vertical_advection=False))
            else:
                P.env_sample(self.vars, t)
                P.coastline('previous')
                P.seafloor()
                P.increase_age(self.dt)
                P.store_previous()
                mid()
                P.advect('runge-kutta4', t, self.dt)
                P.vmix(t, self.dt, self.dt_mix, step=k, fuse_vertical_advection=False)
        elif self.name == 'c5':   # Leeway ensemble members: Euler by construction (leeway.py:472-476)
            if self.fused:   # ONE launch: sample + drift:current_uncertainty + drift:wind_uncertainty + coastline + Leeway.update
                P.env_coast_leeway(self.vars, t, self.dt, 0.4, coastline='stranding', stranded_code=1,
                                   current_uncertainty=0.1, wind_uncertainty=2.0, step=k)
                mid()
                compact()
            else:
                P.env_sample(self.vars, t)
                P.env_add_noise(U, V, 0.1, step=k)        # drift:current_uncertainty
                P.env_add_noise(XW, YW, 2.0, step=k)      # drift:wind_uncertainty
                P.coastline('stranding', stranded_code=1)
                mid()
                compact()
                P.leeway(self.dt, 0.4, step=k)
        else:
            if self.fused:
                P.env_coast_advect(self.vars, t, 'runge-kutta4', self.dt, coastline='stranding', stranded_code=1,
                                   store_previous=False)
                mid()
                compact()
            else:
                P.env_sample(self.vars, t)
                P.coastline('stranding', stranded_code=1)
                mid()
                compact()
                P.advect('runge-kutta4', t, self.dt)
            if os.environ.get('ODR_BENCH_SEPARATE_MOVERS'):    # what-if: the three launches of rounds 1-3
                P.advect_wind(self.dt, wind_drift_depth=0.1)
                P.stokes_drift(self.dt, profile=2, hs_mode=1, tp_mode=1)
                P.hdiffusion(self.dt, step=k)
            else:      # advect_wind -> stokes_drift -> horizontal_diffusion in one launch (odr_movers)
                P.movers(self.dt, wind=dict(wind_drift_depth=0.1), stokes=dict(profile=2, hs_mode=1, tp_mode=1), hdiffusion=dict(step=k))

    def dominant_kernel(self, P, k):
        """The dominant launch of the step on its own, with exactly the arguments step() uses (timed with HIP events)."""
        t = self.time_of(k)
        if self.name == 'c5' and self.fused:
            P.env_coast_leeway(self.vars, t, self.dt, 0.4, coastline='stranding', stranded_code=1, current_uncertainty=0.1,
                               wind_uncertainty=2.0, step=k)
        elif self.name == 'c5':
            P.leeway(self.dt, 0.4, step=k)
        elif self.name == 'c3' and self.fused:
            P.env_coast_advect(self.vars, t, self.scheme, self.dt, coastline='previous', store_previous=True,
                               count=False, seafloor=True, age_dt=self.dt)
        elif self.name == 'c4' and self.fused:
            P.env_coast_advect(self.vars, t, 'runge-kutta4', self.dt, coastline='stranding', stranded_code=1,
                               store_previous=False, count=False)
        else:
            P.advect('runge-kutta4', t, self.dt)

    def second_kernel(self, P, k):
        """C3: the vertical-mixing launch on its own."""
        P.vmix(self.time_of(k), self.dt, self.dt_mix, step=k, fuse_vertical_advection=False)


def cpu_baseline(name, fields, n_cpu, rng, small=False):
    """The CPU oracle (C port of the reference path) on a bounded sample of the workload: one core (the reference is
    single-threaded by design, docs/source/performance.rst:22), and all host cores as independent simulations with
    n_cpu particles each (the quasi-parallel mode the reference's documentation suggests, performance.rst:36)."""
    import threading
    step1 = _cpu_stepper(name, fields, n_cpu, rng)
    nsteps, t0 = 0, time.perf_counter()
    while nsteps < 3 or (time.perf_counter() - t0 < 12.0 and nsteps < 50):
        step1(1 + nsteps)
        nsteps += 1
    el = time.perf_counter() - t0
    out = dict(value=n_cpu * nsteps / el, unit='particle-steps/s', cores=1, kind='port',
               sample='%d particles x %d steps of the same workload (oracle/*.c, gcc -O2, 1 thread, %.1f s)'
                      % (n_cpu, nsteps, el))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    host_cores = cores
    cores = min(cores, int(os.environ.get('ODR_CPU_THREADS', cores)))
    if cores > 1 and not os.environ.get('ODR_CPU_ONE_CORE'):
        # ALL host cores, one simulation per core (the quasi-parallel mode of performance.rst:36), the field blocks SHARED: the
        # world is built once here and the simulations are forked from this process -- its 0.9 GB of blocks are mapped
        # copy-on-write into every child (a child only writes the pages its own particles make the oracle dilate, as the
        # reference's cached blocks are dilated lazily); round 5 gave every simulation its own copy and stopped at 64 cores.
        # The children never touch the GPU runtime and leave through os._exit.
        # ... forked from a HELPER process (`bench.py --cpu-all-cores-helper`), not from this one: this process holds the HIP
        # runtime and its threads, and a child forked while one of them holds a lock (malloc's) never wakes up.  The helper has
        # one thread, builds the same world from the same seeds and forks the simulations.
        try:
            import subprocess
            # (a quarter of the single-core sample per simulation: with every core of the box gathering from the same 0.9 GB of blocks
            # a step of 200 000 particles takes ~50 s -- the leg is memory-bound, which is its result -- and the default run should
            # stay within minutes)
            n_all = max(20000, n_cpu // 4)
            cmd = [sys.executable, os.path.abspath(__file__), '--cpu-all-cores-helper', '--workload', name, '--cpu-particles', str(n_all),
                   '--cpu-cores', str(cores)] + (['--small'] if small else [])
            env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=float(os.environ.get('ODR_CPU_ALL_TIMEOUT', 240)), env=env)
            line = [ln for ln in pr.stdout.splitlines() if ln.startswith('{')]
            if pr.returncode != 0 or not line:
                raise RuntimeError('helper failed: ' + pr.stderr[-300:])
            out['all_cores'] = json.loads(line[-1])
        except Exception as e:      # noqa: BLE001 -- a reported side leg must not take the bench line with it
            out['all_cores'] = dict(value=None, cores=0, host_cores=host_cores, note='not measured: %r' % (e,))
    return out


def cpu_all_cores_helper(a):
    """`bench.py --cpu-all-cores-helper`: the all-cores leg of cpu_baseline in a process of its own (one thread, no GPU runtime)."""
    fields = make_fields(a.workload, a.small)
    step1 = _cpu_stepper(a.workload, fields, a.cpu_particles, np.random.default_rng(5))
    host_cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    print(json.dumps(_cpu_all_cores(a.workload, fields, a.cpu_particles, step1.keep_alive, a.cpu_cores or host_cores, host_cores)), flush=True)
    return 0


def _cpu_all_cores(name, fields, n_cpu, world, cores, host_cores):
    """ALL host cores, one oracle simulation per core, the field blocks shared (forked from this process, copy-on-write)."""
    import struct
    seconds = float(os.environ.get('ODR_CPU_ALL_SECONDS', 8.0))
    start_at = time.time() + 0.02 * cores + 0.5          # every child starts stepping at the same wall-clock instant
    pipes = []
    for j in range(cores):
        r, wfd = os.pipe()
        pid = os.fork()
        if pid == 0:
            try:
                os.close(r)
                # (the warm-up step of every simulation is untimed, as the single-core figure's is: it performs the reference's lazy
                # NaN dilation of the cached blocks -- here on the child's copy-on-write pages, 30 s with 256 children at once)
                stp = _cpu_stepper(name, fields, n_cpu, np.random.default_rng(100 + j), world=world, warm=True)
                while time.time() < start_at:
                    time.sleep(0.001)
                k, t0 = 0, time.perf_counter()
                while time.perf_counter() - t0 < seconds or k < 1:
                    stp(1000 + k)
                    k += 1
                os.write(wfd, struct.pack('qd', k, time.perf_counter() - t0))
            finally:
                os._exit(0)
        os.close(wfd)
        pipes.append((pid, r))
    counts, spans = [], []
    for pid, r in pipes:
        buf = os.read(r, 16)
        os.close(r)
        os.waitpid(pid, 0)
        if len(buf) == 16:
            k, el = struct.unpack('qd', buf)
            counts.append(k)
            spans.append(el)
    if not counts:
        raise RuntimeError('no simulation reported back')
    rate = sum(n_cpu * k / el for k, el in zip(counts, spans))     # every child over its own span (all overlap fully)
    return dict(value=rate, cores=len(counts), host_cores=host_cores,
                note='%d simulations on the %d host cores, field blocks shared (forked, copy-on-write)' % (len(counts), host_cores),
                sample='%d independent simulations x %d particles, %d steps in total, %.1f s each'
                       % (len(counts), n_cpu, sum(counts), max(spans)))


def _cpu_stepper(name, fields, n_cpu, rng, world=None, warm=True):
    """One oracle simulation (its own particles; its own world unless one is handed over); returns step(k) after the warm-up
    step."""
    from oracle import oracle as orc
    if world is not None:
        return _cpu_stepper_on(name, fields, n_cpu, rng, world[0], world[1], world[2], warm)
    wb = orc.WorldBuilder()
    if name == 'c2':
        wb.add_double_gyre(A=0.1, epsilon=0.25, omega=0.628, t0=0.0)
        dt = 0.1
    else:
        g = fields['g']
        proj = orc.make_proj()
        if fields['proj']:
            p = fields['proj']
            f = 1.0 / p['rf']
            proj = orc.make_proj(orc.PROJ_STERE_POLAR, a=p['a'], es=f * (2 - f), lat0=p['lat0'], lon0=p['lon0'],
                                 lat_ts=p['lat_ts'])
        levels = [(float(g['t'][k]), {orc.VAR[n]: g[n][k] for n in fields['names']}) for k in range(3)]
        wb.add_grid(proj, g['x'], g['y'], levels, z=fields['z'])
        for n in fields['names']:
            wb.set_fallback(orc.VAR[n], {LAND: np.nan, DEPTH: 10000.0}.get(n, 0.0))
        wb.set_fallback(orc.VAR[SSH], 0.0)
        if name == 'c4':
            wb.add_constant({orc.VAR[HD]: 10.0})
        dt = 900.0 if name == 'c4' else 600.0
    w = wb.finish()
    return _cpu_stepper_on(name, fields, n_cpu, rng, wb, w, dt, warm)


def _cpu_stepper_on(name, fields, n_cpu, rng, wb, w, dt, warm=True):
    from oracle import oracle as orc
    lon, lat, z = seed_particles(name, fields, n_cpu, rng)
    mv, cdf = np.ones(n_cpu, np.int32), np.ones(n_cpu, np.float32)
    wdf = np.full(n_cpu, 0.02, np.float32)
    if name == 'c5':   # the same PIW-like LeewayObj coefficients as the device run
        r5 = np.random.default_rng(7)
        ori = (np.arange(n_cpu) % 2).astype(np.float32)
        aux = [a.astype(np.float32) for a in (np.full(n_cpu, 0.96), np.where(ori == 0, 0.54, -0.54), np.zeros(n_cpu),
                                              np.zeros(n_cpu), np.abs(r5.standard_normal(n_cpu)) * 12.0,
                                              r5.standard_normal(n_cpu) * 9.4, np.full(n_cpu, 0.04), ori, np.zeros(n_cpu))]

    def step(k):
        t = (k * dt) % 6000.0
        if name == 'c2':
            u, v = orc.get_environment(w, [0, 1], lon, lat, z, t)
            orc.advect_ocean_current(w, 2, lon, lat, z, mv, cdf, u, v, t, dt)
        elif name == 'c3':
            ids = [orc.VAR[n] for n in (U, V, W, DEPTH, SSH, LAND)]
            u, v, ww, dep, ssh, land = orc.get_environment(w, ids, lon, lat, z, t)
            Kp = orc.get_profile(w, orc.VAR[KZ], lon, lat, t, len(fields['z']))
            orc.advect_ocean_current(w, 2, lon, lat, z, mv, cdf, u, v, t, dt)
            uni = np.random.default_rng(k).uniform(size=(10, n_cpu))
            orc.vertical_mixing(z, mv, np.zeros(n_cpu, np.float32), dep, ssh, fields['z'], Kp, dt, 60.0, 0, uni)
            orc.vertical_advection(z, mv, ww, dt)
        elif name == 'c5':
            ids = [orc.VAR[n] for n in (XW, YW, U, V, LAND)]
            xw, yw, u, v, land = orc.get_environment(w, ids, lon, lat, z, t)
            r = np.random.default_rng(k)
            u = (u.astype(np.float64) + r.normal(0, 0.1, n_cpu)).astype(np.float32)      # drift:current_uncertainty
            v = (v.astype(np.float64) + r.normal(0, 0.1, n_cpu)).astype(np.float32)
            xw = (xw.astype(np.float64) + r.normal(0, 2.0, n_cpu)).astype(np.float32)    # drift:wind_uncertainty
            yw = (yw.astype(np.float64) + r.normal(0, 2.0, n_cpu)).astype(np.float32)
            orc.leeway(lon, lat, mv, aux, xw, yw, u, v, dt, 0.4, r.uniform(size=n_cpu))
        else:
            ids = [orc.VAR[n] for n in (U, V, XW, YW, SX, SY, LAND, HD)]
            u, v, xw, yw, sx, sy, land, hd = orc.get_environment(w, ids, lon, lat, z, t)
            orc.advect_ocean_current(w, 2, lon, lat, z, mv, cdf, u, v, t, dt)
            orc.advect_wind(lon, lat, z, mv, wdf, xw, yw, u, v, 0.1, 0, 1.0, dt)
            orc.stokes_drift(lon, lat, z, mv, sx, sy, sx, sy, xw, yw, 1, 1, 2, 1.0, dt)
            r = np.random.default_rng(k)
            orc.horizontal_diffusion(lon, lat, mv, hd, r.standard_normal(n_cpu), r.standard_normal(n_cpu), dt)

    if warm:
        step(0)  # warm-up: also performs the reference's one-off NaN dilation of the cached blocks
    step.keep_alive = (wb, w, dt)   # the world points into buffers owned by the builder
    return step


def model_api_leg(fields, n, steps, device):
    """The drop-in surface on the same inputs: OceanDrift.run() (opendrift_amd/oceandrift.py) with a GridReader holding
    the C3 fields; steady-state ms per step of the main loop (first step -- seeding, first uploads, first sort --
    excluded), lon / lat / status as the only export variables, one output at the end."""
    from datetime import datetime, timedelta
    from opendrift_amd import readers
    from opendrift_amd.oceandrift import OceanDrift
    g = fields['g']
    t0 = datetime(2020, 1, 1)
    times = [t0 + timedelta(seconds=float(t)) for t in g['t']]
    nlev = len(g['t'])
    hours = int(np.ceil(steps * 600.0 / 3600.0)) + 2

    class CyclingReader(readers.GridReader):   # hourly levels for the whole run: the three synthetic levels in turn
        def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
            it = self.times.index(time) % nlev
            out = {'x': self.x, 'y': self.y, 'time': time, 'z': self.z}
            for v in requested_variables:
                out[v] = self.arrays[v][it]
            return out
    times = [t0 + timedelta(hours=k) for k in range(hours)]
    o = OceanDrift(loglevel=50, seed=0, device=device)
    o.add_reader(CyclingReader(g['x'], g['y'], times, {k: g[k] for k in fields['names']}, z=fields['z']))
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    o.set_config('drift:vertical_mixing', True)
    o.set_config('vertical_mixing:timestep', 60)
    o.set_config('drift:vertical_advection', True)
    o.set_config('general:coastline_action', 'previous')
    o.set_config('drift:stokes_drift', False)
    rng = np.random.default_rng(1000)
    lon, lat, z = seed_particles('c3', fields, n, rng)
    o.seed_elements(lon=lon, lat=lat, z=z, time=t0, wind_drift_factor=0.0)
    if os.environ.get('ODR_BENCH_PROFILE_MODEL'):     # developer switch: where the host time of the leg goes (stderr)
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        o.run(time_step=600, steps=steps, time_step_output=600 * steps, export_variables=['z'])
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats('cumulative').print_stats(30)
    else:
        o.run(time_step=600, steps=steps, time_step_output=600 * steps, export_variables=['z'])
    tm = o.timing
    return dict(ms_per_step=tm['steady_ms_per_step'], value=n * 1e3 / tm['steady_ms_per_step'], unit='particle-steps/s',
                steps=tm['steps'], host_phases_ms_per_step=tm.get('host_phases_ms_per_step'),
                what='OceanDrift.run(): the full loop body per step (release, the required variables sampled -- '
                'the export-only sample of the diffusivity at the element only when it is recorded --, deactivation checks, result '
                'buffer, age, compaction, update(): RK4 + wind + vertical mixing + vertical advection, horizontal diffusion early-out)')


class ShardedLoop:
    """What a sharded OceanDrift.run() does around the device step of every rank (opendrift_amd/oceandrift.py, DESIGN.md
    section 6), inside the timed region of `bench.py --gpus N`:
      (i) ONE collective per step -- every rank's step summary (kept count, eight status-reason flags, its 16 raw reduction
          slots) all-gathered (distributed.allgather_vector; RCCL under nccl): OceanDrift._step_summary;
      (ii) a new reader time level every `block_every` steps: rank 0 holds the host Reader's arrays, the broadcast of the
          NEXT level is started one period ahead (distributed.start_broadcast_block: RCCL broadcast straight into device
          memory) and waited for when the level is due; `install(tensors, j)` hands it to the device
          (odr_block_upload_device): DeviceReaderBinding._prefetch_dist / _upload.
    Counters: collectives, collective_s, levels, level_stall_s (host time spent waiting for a level that was due)."""

    def __init__(self, fields, names, block_every, install=None, rccl=None):
        from opendrift_amd import distributed as D
        self.D, self.rank, _, self.world = D, *D.env_world()
        self.fields, self.names, self.block_every, self.install = fields, names, int(block_every), install
        self.rccl = rccl          # (ctx, source id, content ids): reader levels by odr_block_broadcast, staged one period ahead
        if rccl is not None and self.rank == 0 and fields is not None:
            # rank 0's reader arrays page-locked once: the level's copy into the staging memory is a DMA transfer behind which the
            # host does not wait (pageable arrays go through the library's 8 MB bounce buffer, synchronously: 20-40 ms per level)
            for kk in names:
                try:
                    rccl[0].pin(np.ascontiguousarray(fields['g'][kk]))
                except Exception:
                    pass
        self.collectives, self.collective_s, self.levels, self.level_stall_s = 0, 0.0, 0, 0.0
        self.pending = None
        self.last = None

    def _start(self, j):
        g = self.fields['g']
        nlev = len(g['t'])
        arrays = {k: g[k][j % nlev] for k in self.names} if self.rank == 0 else None
        shapes = {k: g[k][j % nlev].shape for k in self.names}
        if self.rccl is not None:
            ctx, sid, cids = self.rccl
            ctx.block_broadcast(sid, j % 3, float(g['t'][j % nlev]), arrays, shapes, root=0, content_ids=cids)
            self.pending = (j, None, None)
            return
        self.pending = (j, ) + self.D.start_broadcast_block(arrays, shapes, src=0)

    def before_step(self, k):
        if not self.block_every or self.fields is None or k % self.block_every:
            return
        j = k // self.block_every
        if self.pending is None or self.pending[0] != j:
            self._start(j)                     # (first level of the loop: nothing was started ahead)
        t0 = time.perf_counter()
        _, tens, works = self.pending
        if self.rccl is not None:              # staged a period ago on the upload stream: the commit makes it current
            self.rccl[0].commit_block(self.rccl[1], j % 3)
            self.level_stall_s += time.perf_counter() - t0
            self.levels += 1
            self._start(j + 1)
            return
        self.D.finish_broadcast(works)
        self.level_stall_s += time.perf_counter() - t0
        self.levels += 1
        if self.install is not None:
            self.install(tens, j)
        self.last = tens
        self._start(j + 1)                     # the next level travels while this one is in use

    def start_summary(self, kept, flags=0, raw16=None, from_scan_ctx=None):
        """The step's ONE collective, started where run() has read the status scan (distributed.start_allgather_vector) ...
        from_scan_ctx (C-ABI collectives): started BEHIND THE FOLD of the scan instead, ahead of the host's read -- the count and
        the flags of the row are taken on the device (odr_comm_allgather_begin, from_scan)."""
        row = np.concatenate([[float(kept)], [float(flags >> b & 1) for b in range(8)], np.zeros(16) if raw16 is None else raw16])
        t0 = time.perf_counter()
        h = self.D.start_allgather_vector(row, from_scan_ctx=from_scan_ctx) if from_scan_ctx is not None else \
            self.D.start_allgather_vector(row)
        self.collective_s += time.perf_counter() - t0
        self.collectives += 1
        return h

    def finish_summary(self, handle):
        """... and finished behind the launches of the step that do not depend on the other ranks (OceanDrift.run(): update()).
        collective_s: the host time of both halves -- what the step still waits for the other ranks."""
        t0 = time.perf_counter()
        rows = self.D.finish_allgather_vector(handle)
        self.collective_s += time.perf_counter() - t0
        return int(round(rows[:, 0].sum()))

    def after_step(self, kept, flags=0, raw16=None):
        return self.finish_summary(self.start_summary(kept, flags, raw16))

    def finish(self):
        if self.pending is not None:           # the level started ahead of the loop's end: no collective may stay open
            if self.rccl is not None:
                self.rccl[0].commit_block(self.rccl[1], self.pending[0] % 3)
            else:
                self.D.finish_broadcast(self.pending[2])
            self.pending = None

    def report(self, steps):
        return dict(collectives=self.collectives, collectives_per_step=self.collectives / max(1, steps), collective_s=self.collective_s,
                    reader_levels=self.levels, block_every=self.block_every, reader_level_stall_s=self.level_stall_s)


def spawn_ranks(a):
    """`python bench.py --gpus N` on its own (no RANK / WORLD_SIZE in the environment) starts its N ranks itself: this
    process becomes the launcher of `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same
    arguments>` (one process per GPU, rendezvous on 127.0.0.1, a free port), rank 0 prints the one JSON line.  Under the
    driver's own torchrun launch WORLD_SIZE is set and nothing is spawned."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # the host driver only supports dmabuf IPC (RCCL over xGMI)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def plumbing_only(a):
    """--plumbing-only: the N-rank machinery of this script WITHOUT the device path -- rendezvous, the ID shards, one field
    block broadcast from rank 0, the barrier-bracketed loop, max / sum over the ranks, rank 0's single line.  What a box
    without N GPUs can check (tests/test_bench_spawn.py, gloo); `value` is null: it measures nothing."""
    from opendrift_amd import distributed as D
    rank, local_rank, world = D.init(backend=os.environ.get('ODR_DIST_BACKEND') or 'torch')   # (the rehearsal layer: tensors on the host)
    n = a.particles or 1000
    fields = make_fields(a.workload, True)
    lo, hi = D.shard_range(n * world, rank, world)
    ok = hi - lo == n
    if fields is not None:
        g = fields['g']
        names = fields['names']
        arrays = {k: g[k][0] for k in names} if rank == 0 else None
        tens = D.broadcast_block(arrays, shapes={k: g[k][0].shape for k in names}, src=0)
        ok = ok and all(np.array_equal(tens[k].cpu().numpy(), g[k][0], equal_nan=True) for k in names)
    # the sharded loop's communication (one collective per step, a reader level every block_every steps) without the device step
    be = a.block_every or (6 if world > 1 and fields is not None else 0)
    sh = ShardedLoop(fields, fields['names'] if fields is not None else [], be)
    D.barrier()
    t0 = time.perf_counter()
    g_kept = n * world
    for k in range(a.steps):
        sh.before_step(k)
        g_kept = sh.after_step(n)
    sh.finish()
    el = time.perf_counter() - t0
    D.barrier()
    ok = ok and g_kept == n * world
    if fields is not None and sh.last is not None:     # the last level that arrived is rank 0's array of that level
        g = fields['g']
        j = (a.steps - 1) // be if be else 0
        ok = ok and all(np.array_equal(sh.last[k].cpu().numpy(), g[k][j % len(g['t'])], equal_nan=True) for k in fields['names'])
    el_max = float(D.allreduce_scalars([el], 'max')[0])
    units = float(D.allreduce_scalars([float(n * a.steps)], 'sum')[0])
    all_ok = float(D.allreduce_scalars([1.0 if ok else 0.0], 'min')[0]) == 1.0
    if rank == 0:
        print(json.dumps({'metric': 'particle-steps/sec (plumbing only: nothing measured)', 'value': None, 'unit': 'particle-steps/s',
                          'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': None, 'higher_is_better': True,
                          'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic', 'plumbing_only': True,
                          'comm': D.comm_info(), 'shards_ok': all_ok, 'units_all_ranks': units, 'loop_s_max_over_ranks': el_max, 'sharded_loop': sh.report(a.steps),
                          'config': {'workload': a.workload, 'particles_per_gpu': n, 'particles_total': n * world}}), flush=True)
    D.shutdown()
    return 0 if all_ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)   # default 400: >= 0.45 s timed region; 25 re-sorts (every 16th step) fall inside it
    ap.add_argument('--warmup', type=int, default=None)  # default 3 (c2: see below)
    ap.add_argument('--workload', default=os.environ.get('ODR_WORKLOAD', 'c3'), choices=['c2', 'c3', 'c4', 'c5'])
    ap.add_argument('--particles', type=int, default=0, help='particles per GPU (default: the config size)')
    ap.add_argument('--small', action='store_true', help='small field block (debug)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the model_api and pcie_inclusive legs')
    ap.add_argument('--host-sort', action='store_true', help='experiment: seed particles already sorted by grid cell')
    ap.add_argument('--cpu-particles', type=int, default=200000)
    ap.add_argument('--cpu-all-cores-helper', action='store_true', help='(internal) the all-cores CPU leg in a process of its own')
    ap.add_argument('--cpu-cores', type=int, default=0)
    ap.add_argument('--block-every', type=int, default=0,
                    help='re-upload one field time level from host memory every N steps inside the timed region '
                         '(PCIe-inclusive rate, DESIGN.md section 5; 0 = inputs resident, the headline)')
    ap.add_argument('--block-async', action='store_true', help='with --block-every: odr_block_upload_async from pinned arrays')
    ap.add_argument('--stage-math', default=os.environ.get('ODR_STAGE_MATH', 'fast'), choices=['fast', 'exact'],
                    help="arithmetic of the Runge-Kutta stage evaluations (odr_ctx_set_stage_math): 'fast' is the headline -- its "
                         "parity gate is tests/test_gpu_stage_math.py -- 'exact' keeps every float32 rounding point of the "
                         "reference inside a stage; the line carries the other mode's rate as `stage_math_exact` / `_fast`")
    ap.add_argument('--plumbing-only', action='store_true',
                    help='rendezvous, shards, block broadcast and the reductions of the N-rank run without the device path '
                         '(no GPU needed; value is null)')
    a = ap.parse_args()
    if a.cpu_all_cores_helper:
        sys.exit(cpu_all_cores_helper(a))
    # C2 (1 M particles, two launches of 50 + 15 us per step): 400 steps are 30-70 ms -- inside the time the GPU takes to leave
    # its idle clocks after the host-side set-up (the first timed loop of the process measured 0.16-0.18 ms per step, the
    # second 0.074).  Its defaults are long enough to measure the steady state; the other workloads keep 400 / 3.
    if a.steps is None:
        a.steps = 4000 if a.workload == 'c2' else 400
    if a.warmup is None:
        a.warmup = 2000 if a.workload == 'c2' else 3
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(a))      # this process is the launcher; its ranks re-enter main() with RANK / WORLD_SIZE set
    if int(os.environ.get('WORLD_SIZE', 1)) != a.gpus:
        print('bench.py: --gpus %d but WORLD_SIZE=%s: the launcher decides, running %s rank(s)'
              % (a.gpus, os.environ.get('WORLD_SIZE', '1'), os.environ.get('WORLD_SIZE', '1')), file=sys.stderr)
    if a.plumbing_only:
        sys.exit(plumbing_only(a))

    from opendrift_amd import distributed as D
    if int(os.environ.get('WORLD_SIZE', 1)) > 1 and D._choose_backend(None)[0] == 'torch':
        # the rehearsal layer (ODR_DIST_BACKEND=gloo | nccl): torch brings its own HIP runtime, and it has to be the FIRST one the
        # process loads -- the device library loaded ahead of it left this rank with "no ROCm-capable device" (two builds of
        # libamdhip64 under one name).  The RCCL path through the C ABI never imports torch.
        import torch  # noqa: F401
    from opendrift_amd.device import Context
    import __graft_entry__ as G
    if int(os.environ.get('RANK', 0)) == 0:
        G.build()                       # (objects are fresh on the GPU box: this loads the library)
    rank, local_rank, world = D.init()  # world > 1: RCCL through the C ABI (odr_comm_*), or torch.distributed on request (ODR_DIST_BACKEND)
    if world == 1 and os.environ.get('ODR_BENCH_RCCL_WORLD1'):
        # rehearsal on ONE GPU of what N ranks do: a one-rank RCCL communicator through the C ABI, the sharded loop on it (with
        # ODR_BENCH_SHARDED_LOOP=1): reader levels by odr_block_broadcast, the step's all-gather from the device scan
        D.init_rccl(world1=True)
    D.barrier()
    use_torch = D.backend() == 'torch'
    if use_torch:
        import torch
        assert torch.cuda.is_available(), 'bench.py needs a GPU: the product path has no CPU fallback'
        dev = local_rank % torch.cuda.device_count()    # == local_rank on a node with one GPU per rank (the driver's launch)
        torch.cuda.set_device(dev)
    else:                               # no torch in this process: one HIP runtime, the device library's
        ndev = D.device_count()
        assert ndev > 0, 'bench.py needs a GPU: the product path has no CPU fallback'
        dev = local_rank % ndev

    def device_synchronize():
        if use_torch:
            torch.cuda.synchronize()

    n = a.particles or {'c2': 1_000_000, 'c3': 10_000_000, 'c4': 6_250_000, 'c5': 10_000_000}[a.workload]
    fields = make_fields(a.workload, a.small)
    ctx = Context(device=dev, seed=0)
    ctx.set_stage_math(a.stage_math)
    wl = Workload(a.workload, ctx, fields, (rank, local_rank, world), via_torch=use_torch or world > 1 or D.backend() == 'rccl')
    rng = np.random.default_rng(1000 + rank)
    lon, lat, z = seed_particles(a.workload, fields, n, rng)
    if a.host_sort and fields is not None:
        g = fields['g']
        if a.workload == 'c3':
            ix = np.searchsorted(g['x'], lon); iy = np.searchsorted(g['y'], lat)
            o = np.argsort((iy // 8) * 100000 + (ix // 8) * 64 + (iy % 8) * 8 + ix % 8, kind='stable')
            lon, lat, z = lon[o], lat[o], z[o]
    lo, hi = D.shard_range(n * world, rank, world)     # global particle IDs of this shard
    n_other = int(os.environ.get('ODR_BENCH_OTHER_RANKS_PARTICLES', 0))
    if n_other and world > 1:
        # pricing run on ONE GPU (tools/gpu_sharded_price.sh): rank 0 holds the whole workload, the other ranks a handful of
        # elements -- the sharded step's machinery (host read, collective, second process) is all there, but the device is not
        # time-sliced between two full workloads: ms_per_step minus the single-process figure is what the machinery costs
        if rank > 0:
            lon, lat, z = lon[:n_other], lat[:n_other], z[:n_other]
        lo = 0 if rank == 0 else n + (rank - 1) * n_other
        hi = lo + len(lon)
        n = len(lon)
    P = ctx.particles(n)
    P.append(lon, lat, z=z, id=np.arange(lo, hi, dtype=np.int32))
    if a.workload == 'c5':   # LeewayObj coefficients of a PIW-like class, perturbed per element (leeway.py:318-372)
        r5 = np.random.default_rng(7 + rank)
        ori = (np.arange(n) % 2).astype(np.float32)
        for slot, val in enumerate([np.full(n, 0.96), np.where(ori == 0, 0.54, -0.54), np.zeros(n), np.zeros(n),
                                    np.abs(r5.standard_normal(n)) * 12.0, r5.standard_normal(n) * 9.4,
                                    np.full(n, 0.04), ori, np.zeros(n)]):
            P.set_property(slot, val.astype(np.float32))

    # N > 1: the timed loop is the SHARDED step -- besides the device step of this rank's particles, the one collective per
    # step and the reader levels arriving from rank 0 by RCCL broadcast (ShardedLoop; weak scaling: every rank steps n particles)
    sharded = None
    # ODR_BENCH_SHARDED_LOOP=1 with one process: the sharded step's sequence (status read between the two launches, the summary
    # row) without a second process -- what the sequence itself costs on a device that is not shared (tools/gpu_sharded_price.sh)
    if (world > 1 or os.environ.get('ODR_BENCH_SHARDED_LOOP')) and a.workload != 'c2':
        def install(tens, j):
            g = fields['g']
            slot = j % 3
            if next(iter(tens.values())).is_cuda:     # RCCL: the level arrived in device memory
                ctx.upload_block_device(wl.sid, slot, float(g['t'][slot]), {kk: t.data_ptr() for kk, t in tens.items()},
                                        {kk: (t.shape[0] if t.dim() == 3 else 1) for kk, t in tens.items()}, content_ids=wl.static_ids)
            else:                                     # gloo rehearsal (ODR_DIST_BACKEND=gloo): host tensors
                ctx.upload_block(wl.sid, slot, float(g['t'][slot]), {kk: t.numpy() for kk, t in tens.items()}, content_ids=wl.static_ids)
        # --block-every -1: no reader levels inside the sharded loop (prices the per-step collective alone: over gloo a 210 MB
        # level travels through the loopback interface, tools/gpu_sharded_price.sh)
        sharded = ShardedLoop(fields, fields['names'], 0 if a.block_every < 0 else (a.block_every or 6), install,
                              rccl=(ctx, wl.sid, wl.static_ids) if D.backend() == 'rccl' else None)

    def timed_loop(steps, first, block_every=0, block_async=False, pinned=None):
        """`steps` steps between barrier + synchronize on both sides; returns (seconds, particle-steps of this rank)"""
        ctx.sync()
        device_synchronize()
        D.barrier()
        n0 = len(P)
        t0 = time.perf_counter()
        for k in range(steps):
            if sharded is not None:
                sharded.before_step(k)
                if os.environ.get('ODR_BENCH_SYNC_SUMMARY'):      # A/B: rounds 3-4 -- the scan and a blocking collective close the step
                    wl.step(P, first + k)
                    sharded.after_step(*P.scan_status())
                    continue
                h = []
                # the ONE host read of a sharded step (the status scan) and its ONE collective, where OceanDrift.run() makes them

                def mid(*scan):
                    if not h:
                        h.append(sharded.start_summary(*(scan or P.scan_status())))
                if D.backend() == 'rccl':      # ... the collective behind the fold of the scan, ahead of the host's read
                    mid.early = lambda: h.append(sharded.start_summary(0, 0, None, from_scan_ctx=ctx))
                wl.step(P, first + k, mid=mid)
                if not h:          # (a workload without deactivations: the summary closes the step)
                    h.append(sharded.start_summary(*P.scan_status()))
                sharded.finish_summary(h[0])
                continue
            if block_every and a.workload != 'c2':   # a new reader time level arrives every block_every steps
                g = fields['g']
                j = k // block_every
                if block_async:     # staged one period ahead on the upload stream from pinned arrays, committed when due
                    if k % block_every == 0:
                        if j > 0:
                            ctx.commit_block(wl.sid, (j - 1) % 3)
                        ctx.upload_block_async(wl.sid, j % 3, float(g['t'][j % 3]), {kk: pinned[kk][j % 3] for kk in fields['names']},
                                               content_ids=wl.static_ids)
                elif k % block_every == 0:
                    ctx.upload_block(wl.sid, j % 3, float(g['t'][j % 3]), {kk: g[kk][j % 3] for kk in fields['names']}, content_ids=wl.static_ids)
            wl.step(P, first + k)
        if sharded is not None:
            sharded.finish()
        ctx.sync()
        device_synchronize()
        el = time.perf_counter() - t0
        D.barrier()
        return el, 0.5 * (n0 + len(P)) * steps

    def warm_uploads(pinned):   # untimed: scratch pools and recyclable blocks of the upload pipeline exist
        g = fields['g']
        for rep in range(2):
            for slot in range(3):
                arrs = {kk: (pinned[kk][slot] if pinned else g[kk][slot]) for kk in fields['names']}
                ctx.upload_block_async(wl.sid, slot, float(g['t'][slot]), arrs, content_ids=wl.static_ids)
                ctx.commit_block(wl.sid, slot)
                wl.step(P, a.warmup)

    pinned = None
    if a.block_every > 0 and a.block_async and a.workload != 'c2':
        pinned = {kk: ctx.pin(np.ascontiguousarray(fields['g'][kk])) for kk in fields['names']}
    # spin-up (untimed, before the W warm-up steps, the same count on every rank): ~0.25 s of the workload's own steps, so that a
    # short timed region (the driver's --steps 20 is 25 ms) does not fall into the GPU's ramp from its idle clocks after the
    # host-side set-up (measured on C2: 0.17 vs 0.06 ms per step); c2 spins up through its own default warm-up
    spin = 0 if a.workload == 'c2' or os.environ.get('ODR_BENCH_NO_SPINUP') else 200
    for k in range(spin):
        wl.step(P, k)
    ctx.sync()
    for k in range(a.warmup):
        wl.step(P, spin + k)
    if a.block_every > 0 and a.workload != 'c2':
        warm_uploads(pinned)
    if sharded is not None:      # untimed: RCCL channels, staging buffers and the recyclable blocks exist before the timed region
        timed_loop(2 * max(sharded.block_every, 3) + 1, spin + a.warmup)
        sharded.collectives, sharded.collective_s, sharded.levels, sharded.level_stall_s = 0, 0.0, 0, 0.0
    el, units_rank = timed_loop(a.steps, spin + a.warmup, max(a.block_every, 0), a.block_async, pinned)
    sharded_report = sharded.report(a.steps) if sharded is not None else None
    sharded = None               # the legs below (kernel timings, the other stage arithmetic) time this rank's device step alone
    el_max = float(D.allreduce_scalars([el], 'max')[0])
    units = float(D.allreduce_scalars([units_rank], 'sum')[0])

    # the two kernels of the step on their own, HIP events on the context stream, exactly the step's launches
    reps = 20
    kfused = wl.fused and a.workload in ('c3', 'c4', 'c5')
    if a.workload != 'c2' and wl.sort_every:
        P.sort_by_cell(wl.sid)      # the layout right after a re-sort, as in 1 of every 16 steps (the kernels drift apart by < 3 % in between)
    if not kfused:
        P.env_sample([U, V], wl.time_of(0))
    # every launch between its own pair of events (one pair around all 20 launches also times the host whenever it
    # falls behind the device: seen as 2.5 instead of 0.84 ms at --steps 200 while rocprofv3 reported 0.84 ms)
    def launch_ms(fn):
        fn(P, 0)
        ctx.sync()
        out = []
        for k in range(reps):
            ctx.timer_begin()
            fn(P, k)
            out.append(ctx.timer_end())
        return float(np.mean(out)), float(np.median(out))
    k_ms, k_ms_median = launch_ms(wl.dominant_kernel)
    k2_ms = None
    if a.workload == 'c3':
        k2_ms, _ = launch_ms(wl.second_kernel)
    nact = len(P)
    kbytes = BYTES[a.workload]['fused' if kfused else 'advect']
    ach = kbytes * nact / (k_ms * 1e-3)

    # the other arithmetic of the stage evaluations on the same particles (a quarter of the steps, same bracketing): the
    # headline is `a.stage_math`; both modes have a parity gate (tests/test_gpu_parity.py ... / tests/test_gpu_stage_math.py)
    other_mode, other = ('exact' if a.stage_math == 'fast' else 'fast'), None
    if a.workload != 'c5' and not a.block_every and not os.environ.get('ODR_BENCH_ONE_MODE'):
        ctx.set_stage_math(other_mode)
        # a measurement of its own, whatever --steps says: >= 48 steps that span two re-sorts (a whole number of re-sort
        # intervals, so that the share of the re-sort in the mean is the steady one), after 8 untimed steps in that mode
        # (round 4 timed 8 steps without a re-sort among them: 11 % off the 100-step figure)
        se = wl.sort_every if (wl.sort_every and a.workload != 'c2') else 1
        nst_o = max(48, a.steps // 4)
        nst_o = -(-nst_o // se) * se
        first_o = spin + a.warmup + a.steps
        for k in range(8):
            wl.step(P, first_o + k)
        el_o, units_o = timed_loop(nst_o, first_o + 8)
        el_o = float(D.allreduce_scalars([el_o], 'max')[0])
        units_o = float(D.allreduce_scalars([units_o], 'sum')[0])
        if a.workload != 'c2' and wl.sort_every:
            P.sort_by_cell(wl.sid)
        ko_ms, ko_med = launch_ms(wl.dominant_kernel)
        other = dict(value=units_o / el_o, unit='particle-steps/s', ms_per_step=1e3 * el_o / nst_o, steps=nst_o, untimed_steps_before=8,
                     re_sorts_inside=(nst_o // se if se > 1 else 0), kernel_ms=ko_ms, kernel_ms_median=ko_med)
        if a.workload == 'c3':
            other['second_kernel_ms'] = launch_ms(wl.second_kernel)[0]
        ctx.set_stage_math(a.stage_math)

    # counters of the same command from the committed rocprofv3 passes (profiles/r06_<workload>_pmc[_exact].json, written by
    # tools/gpu_profile_round.sh + tools/collect_profiles_round.py from this tree): PMC counters cannot be collected from inside
    # this process.  Used only when they belong to this size and stage math.
    def load_pmc(mode):
        for r in ('r06', 'r05', 'r04'):
            f = os.path.join('profiles', '%s_%s_pmc%s.json' % (r, a.workload, '' if mode == 'fast' else '_' + mode))
            if os.path.exists(os.path.join(ROOT, f)):
                pm = json.load(open(os.path.join(ROOT, f)))
                if pm.get('particles') == n and pm.get('stage_math', mode) == mode:
                    return pm, f
        return None, None
    pmc, pmc_file = load_pmc(a.stage_math)
    pmc_other, pmc_other_file = load_pmc(other_mode)
    extras = {}
    if rank == 0 and world == 1 and not a.no_extras and a.workload == 'c3' and not a.small and not a.block_every:
        # (0) the drop-in surface: OceanDrift.run() on the same inputs, in a context of its own (first of the extra legs: it
        # came out 20 % slower when it ran after the upload leg below; the bare sequence's context and particles stay open)
        extras['model_api'] = model_api_leg(fields, n, 48, dev)
        extras['model_api']['vs_bare_sequence'] = extras['model_api']['ms_per_step'] / (1e3 * el_max / a.steps)
    if rank == 0 and world == 1 and not a.no_extras and a.workload == 'c3' and not a.block_every and not a.small and \
            not os.environ.get('ODR_BENCH_SKIP_PCIE'):
        # (i) PCIe-inclusive: a new reader time level (hourly fields, 10-minute steps) arrives from pinned host memory
        # every 6 steps on the upload stream inside the timed region
        pinned = {kk: ctx.pin(np.ascontiguousarray(fields['g'][kk])) for kk in fields['names']}
        warm_uploads(pinned)
        nst = 120     # 20 level periods whatever --steps says (the driver's 20 steps held 4 uploads and no steady state; 48 still read 8 % high)
        el_p, up = timed_loop(nst, a.warmup, 6, True, pinned)
        extras['pcie_inclusive'] = dict(ms_per_step=1e3 * el_p / nst, value=up / el_p, unit='particle-steps/s', steps=nst,
                                        what='one 210 MB time level uploaded every 6 steps (odr_block_upload_async from '
                                             'page-locked arrays, committed one period later) inside the timed region')
        for kk in fields['names']:
            ctx.unpin(pinned[kk])
        pinned = None
    if rank == 0:
        traffic = None if pmc is None else (pmc['FETCH_SIZE_bytes_x2'] + pmc['WRITE_SIZE_bytes']) / 1e9
        out = {
            'metric': {'c3': 'particle-steps/sec (RK4, 3D interp)', 'c2': 'particle-steps/sec (RK4, analytic field)',
                       'c4': 'particle-steps/sec (RK4, 2D interp + wind + Stokes + diffusion)',
                       'c5': 'particle-steps/sec (Leeway, Euler)'}[a.workload], 'value': units / el_max, 'unit': 'particle-steps/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * el_max / a.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': {'c2': 'C2: OceanDrift, analytic double gyre, RK4',
                                    'c3': 'C3: OceanDrift 3D, synthetic ROMS-shaped z-level grid 1024x1024x12 (u,v,w,K), '
                                          'RK4 + vertical_mixing(60 s) + vertical_advection',
                                    'c4': 'C4: OpenOil-advection on NorKyst-800-shaped 2602x902 stere grid, RK4 + wind + '
                                          'Stokes + horizontal diffusion + stranding',
                                    'c5': 'C5: Leeway ensemble members (2 x 5 M per GPU) on the NorKyst-800-shaped grid, '
                                          'wind/current uncertainty, stranding -- the C-ABI sequence (odr_env_coast_leeway = one '
                                          'launch per step), the launch Leeway.run() of the host mirror makes with the device RNG (tests/test_gpu_model_api.py)'}[a.workload],
                       'particles_per_gpu': n, 'particles_total': n * world, 'time_step_s': wl.dt, 'stage_math': a.stage_math, 'spin_up_steps': spin,
                       'block_every': a.block_every, 'inputs': 'resident in HBM' if a.block_every <= 0 else 'uploaded in the timed region',
                       'parallelism': 'particle-sharded x%d, field block broadcast once per time level' % world},
        }
        kname = ('k_step_leeway' if wl.fused else 'k_leeway') if a.workload == 'c5' else ('k_step_grid<RK4>' if kfused else 'k_advect<RK4>')
        # (1) SURVEY.md 8(d): algorithmic bytes (4 B per field corner touched + state once in / once out) over the launch time.
        # Counts cache-served corners: a throughput figure in the survey's unit, NOT a physical bandwidth (it can exceed 1).
        algorithmic = {'bound': 'hbm', 'kernel': kname, 'achieved': ach / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                       'frac': ach / HBM_PEAK, 'algorithmic_gb_per_launch': kbytes * nact / 1e9,
                       'algorithmic_bytes_per_particle': kbytes, 'step_bytes_per_particle': BYTES[a.workload]['step'],
                       'step_frac': BYTES[a.workload]['step'] * (units / el_max) / world / HBM_PEAK,
                       'note': 'SURVEY 8(d) accounting: counts corners served by L1 / L2 / Infinity Cache; not a physical bandwidth'}
        out['roofline_algorithmic'] = algorithmic
        if pmc is not None:
            # (2) roofline of the dominant kernel: achieved = its HBM bytes per launch from the counters (FETCH_SIZE x 2 +
            # WRITE_SIZE, MI355X_MICROARCH.md) over the launch time measured HERE with HIP events; peak = 8 TB/s; frac =
            # achieved / peak -- a fraction of a physical peak, whichever unit binds.  `units` carries the other units of the
            # launch the same way, achieved / measured peak: L1 lookups per clock per CU against the 2.03 of
            # tools/gather_bench.hip's `nodes` shape (profiles/r03_gather_bench.txt); VALU issue cycles (every wave64
            # instruction priced by class as measured by tools/rate_bench.hip: float64 arithmetic / conversions 4,
            # transcendental float64 16, other 2) against 1 024 SIMDs; and, as a diagnostic only, the share of the launch the
            # texture addresser reported busy (TA_TA_BUSY: includes the cycles it waits on the caches -- not a roofline).
            def units_of(q, ms):
                hb = q['FETCH_SIZE_bytes_x2'] + q['WRITE_SIZE_bytes']
                clk_ = (q.get('GRBM_GUI_ACTIVE_8xcd') or 0) / 8.0 / (q['kernel_us_rocprof'] * 1e-6) if q.get('GRBM_GUI_ACTIVE_8xcd') else 2.4e9
                sec = ms * 1e-3
                return hb, clk_, {
                    'hbm': hb / sec / HBM_PEAK,
                    'l1_lookups': (q.get('TCP_TOTAL_CACHE_ACCESSES') or 0) / 256.0 / (clk_ * q['kernel_us_rocprof'] * 1e-6) / 2.03,
                    'valu_issue': q['valu_cycles_per_simd_slot'] / (1024 * clk_) / sec,
                    'ta_busy_share_diagnostic': (q.get('TA_TA_BUSY') or 0) / 256.0 / clk_ / sec}
            hbm_bytes, clk, un = units_of(pmc, k_ms)
            out['roofline'] = {
                'bound': 'hbm', 'kernel': kname, 'kernel_ms': k_ms, 'kernel_ms_median': k_ms_median,
                'achieved': hbm_bytes / (k_ms * 1e-3) / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': un['hbm'],
                'frac_of_achievable': hbm_bytes / (k_ms * 1e-3) / HBM_ACHIEVABLE, 'achievable': HBM_ACHIEVABLE / 1e9,
                'traffic': hbm_bytes / 1e9, 'traffic_unit': 'GB of HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE)',
                'algorithmic_gb_per_launch': kbytes * nact / 1e9,
                'units': un, 'binding_unit': max((k for k in un if not k.endswith('diagnostic')), key=lambda k: un[k]),
                'clock_ghz': clk / 1e9,
                'source': pmc_file + ' (counters of the same command and tree; this line times the launch itself)',
                'note': 'frac = HBM bytes of the launch (counters) / launch time / 8 TB/s; the SURVEY 8(d) figure is roofline_algorithmic'}
            if 'second' in pmc and k2_ms is not None:
                hb2, _, un2 = units_of(pmc['second'], k2_ms)
                out['roofline']['second_kernel'] = {'kernel': 'k_vmix_col', 'kernel_ms': k2_ms, 'frac': un2['hbm'], 'traffic': hb2 / 1e9,
                                                    'units': un2}
            if pmc.get('step_hbm_bytes'):
                # every kernel of a step (re-sort share included), real HBM bytes over the measured step time
                sb = pmc['step_hbm_bytes']
                out['roofline_step'] = {'bound': 'hbm', 'achieved': sb / (el_max / a.steps) / 1e9,
                                        'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': sb / (el_max / a.steps) / HBM_PEAK,
                                        'frac_of_achievable': sb / (el_max / a.steps) / HBM_ACHIEVABLE,
                                        'traffic': sb / 1e9, 'kernels': pmc.get('step_hbm_kernels'),
                                        'note': 'HBM bytes of ALL kernels of one step (counters, calls per step from the kernel trace) / ms_per_step / 8 TB/s'}
        else:
            out['roofline'] = dict(algorithmic, kernel_ms=k_ms, kernel_ms_median=k_ms_median, traffic=None,
                                   note='no counter file for this size / stage math: the SURVEY 8(d) algorithmic figure only')
            if k2_ms is not None:
                out['roofline']['second_kernel'] = {'kernel': 'k_vmix_col', 'kernel_ms': k2_ms}
        out['comm'] = D.comm_info()     # which layer carried the collectives and how many ranks IT saw (N > 1: RCCL through the C ABI)
        out['torch_in_process'] = 'torch' in sys.modules
        if sharded_report is not None:
            out['sharded_loop'] = dict(sharded_report, what='the timed region holds, per step: the device step of this rank, one host read '
                                       '(odr_scan_status) and ONE all_gather of the step summaries; every block_every steps a reader level '
                                       'from rank 0, staged one period ahead (C-ABI collectives: odr_block_broadcast, ONE RCCL broadcast inside '
                                       'the upload pipeline; torch layer: broadcast into tensors -> odr_block_upload_device)')
        if a.workload in ('c3', 'c4'):
            ts = P.tile_stats()     # the LDS-tile step (csrc/odr_tile.hip.h): launches on that path / elements it handed to the HBM path
            out['lds_tile'] = dict(ts, handed_over_per_launch=(ts['handed_over'] / ts['launches'] if ts['launches'] else None))
        if other is not None:
            # the other arithmetic's launch against the same peak: its own counter file when one was collected
            # (profiles/r05_<workload>_pmc_<mode>.json), else the bytes of the headline mode's launch -- the two arithmetics
            # read and write the same particle state and node records (the counters of both agree to < 1 %)
            src_pm, src_f = (pmc_other, pmc_other_file) if pmc_other is not None else (pmc, pmc_file)
            if src_pm is not None:
                hb_o = src_pm['FETCH_SIZE_bytes_x2'] + src_pm['WRITE_SIZE_bytes']
                other['roofline'] = {'bound': 'hbm', 'kernel': kname, 'kernel_ms': other['kernel_ms'], 'traffic': hb_o / 1e9,
                                     'achieved': hb_o / (other['kernel_ms'] * 1e-3) / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                                     'frac': hb_o / (other['kernel_ms'] * 1e-3) / HBM_PEAK,
                                     'frac_of_achievable': hb_o / (other['kernel_ms'] * 1e-3) / HBM_ACHIEVABLE, 'source': src_f}
            out['stage_math_' + other_mode] = other
        out.update(extras)
        if not a.no_cpu and world == 1:
            out['cpu_baseline'] = cpu_baseline(a.workload, fields, a.cpu_particles, np.random.default_rng(5), small=a.small)
            refs = [os.path.join(ROOT, 'profiles', '%s_cpu_reference_numpy.json' % r) for r in ('r06', 'r02')]   # newest first
            ref = next((f for f in refs if os.path.exists(f)), refs[-1])
            if a.workload == 'c3' and os.path.exists(ref):   # the reference's own NumPy path, timed where /root/reference exists
                rj = json.load(open(ref))
                best = max(rj['runs'], key=lambda r: r['particle_steps_per_s'])
                out['cpu_baseline']['reference_numpy'] = dict(
                    value=best['particle_steps_per_s'], unit='particle-steps/s', cores=1, kind='reference',
                    sample='%d particles x %d steps, median' % (best['particles'], best['steps_timed']),
                    measured_on='%s (%s, %d cores) -- not this host; /root/reference is not on the GPU box'
                                % (rj['measured_on'], rj['host_cpu'], rj['host_cores']), source='profiles/' + os.path.basename(ref))
        print(json.dumps(out), flush=True)
    if P is not None:
        P.close()
    D.shutdown()          # the communicator goes before the context does
    ctx.close()


if __name__ == '__main__':
    main()
