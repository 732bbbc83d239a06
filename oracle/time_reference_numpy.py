"""TEST INFRASTRUCTURE ONLY -- times the REFERENCE'S OWN NumPy path on the C3 workload (SURVEY.md
section 8d): OceanDrift (runge-kutta4, vertical mixing with 60 s sub-steps, vertical advection) on the synthetic
ROMS-shaped z-level grid of bench.py, the loop body driven by oracle/refdriver.py (run() itself needs xarray), pyproj
replaced by the C build of the same geodesic (oracle/refshim.py) so that the reference is not penalised for a Python
geodesic.  One core (the reference is single-threaded by design, docs/source/performance.rst:22).

    python oracle/time_reference_numpy.py [N ...]      # default 100000 1000000

Writes profiles/<ODR_ROUND, default r06>_cpu_reference_numpy.json (host CPU model and core count stated).  Runs in the build container only:
/root/reference does not exist on the GPU box; bench.py carries the stored numbers as cpu_baseline.reference_numpy.
"""
import json
import os
import sys
import time
from datetime import timedelta

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
os.environ.setdefault('OMP_NUM_THREADS', '1')     # performance.rst:7
import gen_golden as gg  # noqa: E402  (installs the shim)
from oracle.refdriver import RefStepper  # noqa: E402
from opendrift_amd import synthetic as synth  # noqa: E402


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def time_c3(n, g, steps=5):
    times = [gg.T0 + timedelta(seconds=float(t)) for t in g['t']]
    names = ('x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity', 'ocean_vertical_diffusivity',
             'sea_floor_depth_below_sea_level', 'land_binary_mask')
    o = gg._base('runge-kutta4')
    o.add_reader(gg.GridReader('+proj=latlong', g['x'], g['y'], times, {k: g[k] for k in names}, z=g['z']))
    o.set_config('drift:vertical_mixing', True)
    o.set_config('vertical_mixing:timestep', 60)
    o.set_config('vertical_mixing:diffusivitymodel', 'environment')
    o.set_config('drift:vertical_advection', True)
    o.set_config('general:coastline_action', 'previous')
    o.set_config('drift:stokes_drift', False)
    rng = np.random.default_rng(1000)
    lon = rng.uniform(g['x'][8], g['x'][int(0.9 * len(g['x']))], n)
    lat = rng.uniform(g['y'][8], g['y'][-9], n)
    z = -rng.uniform(0, 50, n)
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, z=z, time=gg.T0, wind_drift_factor=0.0)
    st = RefStepper(o, 600.0, steps + 1)
    st.step()                                  # warm-up: reader blocks cached, NaN dilation done
    per = []
    for _ in range(steps):
        t0 = time.perf_counter()
        st.step()
        per.append(time.perf_counter() - t0)
    med = float(np.median(per))
    return dict(particles=n, steps_timed=steps, s_per_step_median=med, s_per_step_all=per,
                particle_steps_per_s=n / med, active_at_end=int(o.num_elements_active()))


def main():
    ns = [int(a) for a in sys.argv[1:]] or [100000, 1000000]
    g = synth.grid3d(nx=1024, ny=1024, nz=12, nt=3, seed=0)
    out = dict(workload='C3: OceanDrift 3D, synthetic ROMS-shaped z-level grid 1024x1024x12 (u,v,w,K), RK4 + '
                        'vertical_mixing(60 s) + vertical_advection -- the inputs of bench.py --workload c3',
               what="the reference's own NumPy code (OpenDrift v1.14.10: Environment.get_environment, ReaderBlock, "
                    "advect_ocean_current, vertical_mixing) driven step by step by oracle/refdriver.py; pyproj = C build of "
                    "the same geodesic (oracle/refshim.py); no result buffer, no netCDF",
               cores=1, host_cpu=cpu_model(), host_cores=os.cpu_count(), measured_on='build container (no GPU)',
               numpy=np.__version__, runs=[])
    for n in ns:
        r = time_c3(n, g)
        print(n, 'particles: %.3f s/step -> %.3e particle-steps/s' % (r['s_per_step_median'], r['particle_steps_per_s']), flush=True)
        out['runs'].append(r)
    with open(os.path.join(ROOT, 'profiles', '%s_cpu_reference_numpy.json' % os.environ.get('ODR_ROUND', 'r06')), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
