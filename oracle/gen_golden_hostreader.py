"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/c19_host_reader_rk.npz from the REFERENCE ITSELF.

A user-defined ContinuousReader (basereader/continuous.py:20-46) is called with the exact element positions; under a
Runge-Kutta scheme advect_ocean_current (physics_methods.py:611-680) calls get_environment again at every stage position
and stage time.  Scenarios (the reference's OceanDrift, lon/lat, no wind / Stokes / diffusion):

  a_rk2, a_rk4   the current comes from a time-dependent analytic reader alone (a shear + an oscillation whose period
                 makes the stage times t + dt/2 and t + dt matter)
  b_rk4          the same reader restricted to a box, FIRST in the priority list, a gridded reader behind it: elements
                 inside the box take the analytic current, the others the interpolated one -- per stage position

Stored: the reader parameters, the grid, the float64 state per step.

    python oracle/gen_golden_hostreader.py
"""
import os
import sys
from datetime import timedelta

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (installs the shim)
from opendrift.readers.basereader.continuous import ContinuousReader  # noqa: E402

PERIOD = 5400.0


def field(lon, lat, seconds):
    """the analytic current both sides evaluate (float64): shear in latitude + oscillation in time + a cell pattern"""
    ph = 2 * np.pi * seconds / PERIOD
    u = 0.4 * (lat - 60.0) + 0.3 * np.sin(ph) + 0.2 * np.sin(3.0 * (lon - 4.0))
    v = 0.25 * np.cos(ph) * np.cos(2.0 * (lon - 4.0)) - 0.1 * (lat - 60.0)
    return u, v


class Analytic(ContinuousReader):
    def __init__(self, box=None):
        self.variables = ['x_sea_water_velocity', 'y_sea_water_velocity']
        self.proj4 = '+proj=latlong +datum=WGS84'
        self.xmin, self.xmax, self.ymin, self.ymax = box if box else (-180, 180, -90, 90)
        self.start_time = self.end_time = self.time_step = None
        self.name = 'analytic_shear'
        super().__init__()

    def get_variables(self, requestedVariables, time=None, x=None, y=None, z=None):
        u, v = field(np.asarray(x, np.float64), np.asarray(y, np.float64), (time - gg.T0).total_seconds())
        return {'time': time, 'x': x, 'y': y, 'z': z, 'x_sea_water_velocity': u, 'y_sea_water_velocity': v}


def scenario(tag, scheme, box=None, grid=None, steps=8, dt=900):
    o = gg._base(scheme)
    o.add_reader(Analytic(box))
    if grid is not None:
        o.add_reader(grid)
    o.set_config('environment:constant:land_binary_mask', 0)
    o.set_config('drift:stokes_drift', False)
    rng = np.random.default_rng(19)
    N = 300
    lon, lat = rng.uniform(4.2, 5.2, N), rng.uniform(59.6, 60.6, N)
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, time=gg.T0, wind_drift_factor=0.0)
    res, _ = gg._run(o, dt, steps)
    print(tag, 'moved', np.abs(res['lon'][-1] - res['lon'][0]).max(), 'deg', o.status_categories)
    return {('%s_%s' % (tag, k)): v for k, v in res.items()}


def main():
    out = dict(period=PERIOD, dt=900.0)
    out.update(scenario('a_rk2', 'runge-kutta'))
    out.update(scenario('a_rk4', 'runge-kutta4'))
    nx, ny, nt = 40, 30, 4
    x = np.linspace(3.5, 6.0, nx).astype(np.float32)
    y = np.linspace(59.0, 61.2, ny).astype(np.float32)
    X, Y = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    t = np.arange(nt) * 3600.0
    times = [gg.T0 + timedelta(seconds=float(v)) for v in t]
    u = np.stack([(-0.5 + 0.1 * k) * np.cos(3 * Y + 0.2 * k) for k in range(nt)]).astype(np.float32)
    v = np.stack([(0.4 - 0.05 * k) * np.sin(4 * X) for k in range(nt)]).astype(np.float32)
    box = (4.5, 4.9, 59.9, 60.3)
    grid = gg.GridReader('+proj=latlong', x, y, times, {'x_sea_water_velocity': u, 'y_sea_water_velocity': v})
    out.update(scenario('b_rk4', 'runge-kutta4', box=box, grid=grid))
    inside = ((out['b_rk4_lon'][0] > box[0]) & (out['b_rk4_lon'][0] < box[1]) & (out['b_rk4_lat'][0] > box[2]) & (out['b_rk4_lat'][0] < box[3]))
    print('b: elements starting inside the box', int(inside.sum()))
    assert 10 < inside.sum() < 290
    out.update(box=np.array(box), g_x=x, g_y=y, g_t=t, g_u=u, g_v=v)
    np.savez_compressed(os.path.join(gg.GOLD, 'c19_host_reader_rk.npz'), **out)


if __name__ == '__main__':
    main()
