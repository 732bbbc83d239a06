"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/c17_ensemble_reader.npz from the REFERENCE ITSELF.

A StructuredReader may hand out a variable as a LIST of arrays, one per ensemble member (basereader/structured.py:125-147,
reader_netCDF_CF_generic.py); ReaderBlock.interpolate (readers/interpolation/structured.py:119-135) then gives element
number j OF THE CALL the member j % M -- the position in the arrays of that get_environment call, i.e. the rank among the
active elements, which shifts whenever an element is removed.  Ensemble layers are not filled towards the sea floor
(:58-60).

Scenario: the reference's OceanDrift on a lon/lat grid whose current comes in M = 3 members (clearly different fields),
plain wind-free run, 'runge-kutta4' (the stage calls use the same member mapping), land strip with stranding (so that the
ranks shift during the run), 2D and 3D member arrays.  Stored: inputs and the live float64 state per step.

    python oracle/gen_golden_ensemble.py
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
from oracle.refdriver import RefStepper  # noqa: E402


class EnsembleGridReader(gg.GridReader):
    """arrays[var] = list over members of [nt, (nz,) ny, nx] -> get_variables hands out a list of member arrays"""

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        it = self.times.index(time)
        out = {'x': self.x, 'y': self.y, 'time': time, 'z': self.z if self.z is not None else 0}
        for v in requested_variables:
            a = self.arrays[v]
            out[v] = [np.array(m[it], copy=True) for m in a] if isinstance(a, list) else np.array(a[it], copy=True)
        return out


def run(tag, three_d, partial=False):
    """partial: a quarter of the elements start WEST of the reader's domain (fallback current carries them in): the block
    numbers only the elements handed to it -- the covered ones (variables.py:747-765 ind_covered)."""
    nx, ny, nt, M = 48, 36, 3, 3
    x = np.linspace(3.0, 6.0, nx).astype(np.float32)
    y = np.linspace(59.0, 61.0, ny).astype(np.float32)
    X, Y = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    t = np.arange(nt) * 3600.0
    times = [gg.T0 + timedelta(seconds=float(v)) for v in t]
    zlev = np.array([0.0, -10.0, -30.0]) if three_d else None
    g = dict(x=x, y=y, t=t)

    def field(f):
        a = np.stack([f(k) for k in range(nt)]).astype(np.float32)          # [nt, ny, nx]
        if three_d:
            a = np.stack([a * s for s in (1.0, 0.6, 0.2)], axis=1)          # [nt, nz, ny, nx]
        return a
    us = [field(lambda k, m=m: (0.5 + 0.25 * m) * np.cos(2 * Y + 0.3 * k + m) + 0.9) for m in range(M)]
    vs = [field(lambda k, m=m: (0.3 - 0.2 * m) * np.sin(3 * X - 0.2 * k) + 0.1 * m) for m in range(M)]
    land = np.zeros((ny, nx), np.float32)
    land[:, 37:] = 1.0
    for m in range(M):
        g['u%d' % m], g['v%d' % m] = us[m], vs[m]
    g['land_binary_mask'] = np.stack([land] * nt)
    o = gg._base('runge-kutta4')
    arrays = {'x_sea_water_velocity': us, 'y_sea_water_velocity': vs, 'land_binary_mask': g['land_binary_mask']}
    o.add_reader(EnsembleGridReader('+proj=latlong', x, y, times, arrays, z=zlev))
    o.set_config('general:coastline_action', 'stranding')
    o.set_config('general:coastline_approximation_precision', None)
    o.set_config('drift:stokes_drift', False)
    if partial:
        o.set_config('environment:fallback:x_sea_water_velocity', 1.5)
        o.set_config('environment:fallback:y_sea_water_velocity', 0.1)
        o.set_config('environment:fallback:land_binary_mask', 0)
    rng = np.random.default_rng(17)
    N = 200
    lon = rng.uniform(4.4, 5.34, N)
    if partial:
        lon[::4] = rng.uniform(2.93, 2.999, len(lon[::4]))      # outside [3, 6]: 1.5 m/s eastward brings them in within a few steps
    lat = rng.uniform(59.3, 60.7, N)
    zz = -rng.uniform(0, 25, N) if three_d else np.zeros(N)
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, z=zz, time=gg.T0, wind_drift_factor=0.0)
    res, _ = gg._run(o, 900, 8)
    stranded = int((res['status'][-1] > 0).sum())
    print(tag, 'stranded', stranded, 'categories', o.status_categories, 'lon range', np.nanmin(res['lon'][-1]), np.nanmax(res['lon'][-1]))
    assert stranded > 10, stranded
    return {('%s_%s' % (tag, k)): v for k, v in res.items()}, g, zlev


def main():
    out = {}
    for tag, three_d in (('2d', False), ('3d', True), ('partial', False)):
        res, g, zlev = run(tag, three_d, partial=tag == 'partial')
        out.update(res)
        out.update({('%s_g_%s' % (tag, k)): v for k, v in g.items()})
        if zlev is not None:
            out[tag + '_g_z'] = zlev
    np.savez_compressed(os.path.join(gg.GOLD, 'c17_ensemble_reader.npz'), dt=900.0, members=3, **out)


if __name__ == '__main__':
    main()
