"""TEST INFRASTRUCTURE ONLY -- golden vectors of two OceanDrift options, written by the reference itself (round 4):

  c21a  drift:water_column_stretching (models/oceandrift.py:299-313): sea_surface_height varies in space and time; the first
        call of update() moves z by delta_zeta * z / depth with the sea_surface_height of the previous step
        (update_previous_state, basemodel/__init__.py:642-656, restated here without xarray);
  c21b  drift:truncate_ocean_model_below_m (models/basemodel/environment.py:554-566): every get_environment call -- the
        Runge-Kutta stage calls of advect_ocean_current included -- samples the readers at max(z, -20 m).

    python oracle/gen_golden_options.py              ->  tests/golden/c21_options.npz
    python oracle/gen_golden_options.py ocean_only   ->  tests/golden/c22_ocean_only.npz  (seed:ocean_only, closest_ocean_points)
"""
import os
import sys
from datetime import timedelta
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import refshim  # noqa: E402,F401  (installs the stand-ins, puts /root/reference on the path)
from oracle import gen_golden as gg  # noqa: E402
from oracle.refdriver import RefStepper  # noqa: E402
from opendrift_amd import synthetic as synth  # noqa: E402


class _EnvPrevious:
    """environment_previous of the reference (an xarray Dataset selected by trajectory) as plain arrays"""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __contains__(self, k):
        return k in self.__dict__


class StepperWithPreviousEnvironment(RefStepper):
    """RefStepper + the environment half of update_previous_state (basemodel/__init__.py:642-656)"""

    def step(self):
        o = self.o
        self._release()
        o.environment, o.environment_profiles, missing = o.env.get_environment(
            list(o.required_variables), o.time, o.elements.lon, o.elements.lat, o.elements.z,
            o.required_profiles, o.profiles_depth, element_ID=o.elements.ID)
        o.calculate_missing_environment_variables()
        o.report_missing_variables(missing)
        o.deactivate_outside()
        o.interact_with_coastline()
        o.interact_with_seafloor()
        o.increase_age_and_retire()
        o.remove_deactivated_elements()
        if self._store_prev:
            o._elements_previous.lon[o.elements.ID] = o.elements.lon
            o._elements_previous.lat[o.elements.ID] = o.elements.lat
        if getattr(self, '_ssh_by_id', None) is None:
            self._ssh_by_id = np.full(self.n_total, np.nan, np.float32)
        ids = np.asarray(o.elements.ID, dtype=int)
        ssh = np.asarray(o.environment.sea_surface_height, dtype=np.float32)
        prev = self._ssh_by_id[ids]
        # (a masked array: the reference's environment_previous is an xarray selection, and water_column_stretching ends
        # with `self.elements.z = self.elements.z.data`, which needs an array wrapper to unwrap)
        o.environment_previous = _EnvPrevious(sea_surface_height=np.ma.array(np.where(np.isnan(prev), ssh, prev).astype(np.float32)))
        self._ssh_by_id[ids] = ssh
        if o.num_elements_active() > 0:
            o.update()
        o.horizontal_diffusion()
        o.time = o.time + o.time_step
        o.steps_calculation += 1


def _fields():
    g = synth.grid3d(nx=48, ny=40, nz=8, nt=3, seed=21, coast=False)
    X, Y = np.meshgrid(np.linspace(0, 1, 48), np.linspace(0, 1, 40))
    g['sea_surface_height'] = np.stack([(0.6 * np.sin(3 * X + 0.9 * k) * np.cos(2 * Y) + 0.25 * k) for k in range(3)]).astype(np.float32)
    return g


def _run(o, stepper_cls, lon, lat, zz, steps, dt):
    o.seed_elements(lon=lon, lat=lat, z=zz, time=gg.T0, wind_drift_factor=0.0)
    st = stepper_cls(o, dt, steps)
    N = len(lon)
    res = {k: np.full((steps + 1, N), np.nan) for k in ('lon', 'lat', 'z')}
    sch = o.elements_scheduled
    res['lon'][0], res['lat'][0], res['z'][0] = sch.lon, sch.lat, np.atleast_1d(sch.z) * np.ones(N)
    for k in range(steps):
        st.step()
        res['lon'][k + 1], res['lat'][k + 1], res['z'][k + 1], _ = st.state()
    return res


def main():
    g = _fields()
    times = [gg.T0 + timedelta(seconds=float(t)) for t in g['t']]
    names = ('x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity', 'sea_floor_depth_below_sea_level',
             'sea_surface_height', 'land_binary_mask')
    arrays = {k: g[k] for k in names}
    rng = np.random.default_rng(21)
    N = 300
    lon = rng.uniform(g['x'][4], g['x'][-5], N)
    lat = rng.uniform(g['y'][4], g['y'][-5], N)
    zz = -rng.uniform(0, 70, N)
    zz[:40] = 0.0
    out = {}
    # a: water column stretching (RK2; vertical advection on, mixing off)
    o = gg._base('runge-kutta')
    o.add_reader(gg.GridReader('+proj=latlong', g['x'], g['y'], times, arrays, z=g['z']))
    o.set_config('drift:water_column_stretching', True)
    o.set_config('drift:vertical_mixing', False)
    o.set_config('drift:vertical_advection', True)
    o.set_config('drift:stokes_drift', False)
    o.set_config('general:coastline_action', 'previous')
    res = _run(o, StepperWithPreviousEnvironment, lon, lat, zz, 8, 600.0)
    out.update({'a_' + k: v for k, v in res.items()})
    # (the same run without the option: the test checks that the option matters)
    o = gg._base('runge-kutta')
    o.add_reader(gg.GridReader('+proj=latlong', g['x'], g['y'], times, arrays, z=g['z']))
    o.set_config('drift:vertical_mixing', False)
    o.set_config('drift:vertical_advection', True)
    o.set_config('drift:stokes_drift', False)
    o.set_config('general:coastline_action', 'previous')
    res0 = _run(o, RefStepper, lon, lat, zz, 8, 600.0)
    out['a_z_without'] = res0['z']
    print('water_column_stretching: max |dz| against the run without it: %.3f m' % np.nanmax(np.abs(res['z'][-1] - res0['z'][-1])))
    # b: truncation at 20 m (RK4)
    for tag, trunc in (('b', 20.0), ('b0', None)):
        o = gg._base('runge-kutta4')
        o.add_reader(gg.GridReader('+proj=latlong', g['x'], g['y'], times, arrays, z=g['z']))
        if trunc is not None:
            o.set_config('drift:truncate_ocean_model_below_m', trunc)
        o.set_config('drift:vertical_mixing', False)
        o.set_config('drift:vertical_advection', True)
        o.set_config('drift:stokes_drift', False)
        o.set_config('general:coastline_action', 'previous')
        r = _run(o, RefStepper, lon, lat, zz, 8, 600.0)
        out.update({tag + '_' + k: v for k, v in r.items()})
    print('truncation: max |dlon| against the run without it: %.2e deg' % np.nanmax(np.abs(out['b_lon'][-1] - out['b0_lon'][-1])))
    np.savez_compressed(os.path.join(gg.GOLD, 'c21_options.npz'), dt=600.0, truncate=20.0,
                        **{('g_' + k): v for k, v in g.items()}, **out)




def ocean_only():
    """c22: seed:ocean_only -- the reference's own closest_ocean_points (basemodel/__init__.py:936-1031) on seeds of which a
    part lies on the land strip of the synthetic grid, land_binary_mask from a gridded reader."""
    g = synth.grid3d(nx=64, ny=48, nz=2, nt=2, seed=22, lon0=3.0, lon1=4.2, lat0=60.0, lat1=60.7)
    times = [gg.T0 + timedelta(seconds=float(t)) for t in g['t']]
    names = ('x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask')
    o = gg._base('euler')
    o.add_reader(gg.GridReader('+proj=latlong', g['x'], g['y'], times, {k: g[k][:, 0] if g[k].ndim == 4 else g[k] for k in names}))
    rng = np.random.default_rng(22)
    N = 400
    lon = rng.uniform(4.02, 4.19, N)          # the land strip starts around X = 0.94 (lon ~ 4.13)
    lat = rng.uniform(60.05, 60.65, N)
    o.seed_elements(lon=lon, lat=lat, time=gg.T0)
    sch = o.elements_scheduled
    lon0, lat0 = np.array(sch.lon, dtype=np.float64), np.array(sch.lat, dtype=np.float64)
    # closest_ocean_points reads self.elements.ID for get_environment's element_ID argument
    from opendrift.models.basemodel import Mode
    o.env.finalize(start=gg.T0, end=gg.T0 + timedelta(hours=1))
    lo, la, idx = o.closest_ocean_points(sch.lon, sch.lat)
    print('ocean_only: %d of %d seeds moved, max shift %.3f deg' % (len(idx), N, np.abs(np.asarray(lo, np.float64) - lon0).max()))
    np.savez_compressed(os.path.join(gg.GOLD, 'c22_ocean_only.npz'), lon0=lon0, lat0=lat0, lon=np.asarray(lo, np.float64),
                        lat=np.asarray(la, np.float64), moved=np.asarray(idx, np.int64),
                        **{('g_' + k): v for k, v in g.items()})


if __name__ == '__main__':
    if 'ocean_only' in sys.argv:
        ocean_only()
    else:
        main()
