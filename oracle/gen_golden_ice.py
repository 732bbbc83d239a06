"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/c16_openoil_sea_ice.npz from the REFERENCE ITSELF.

OpenOil.advect_oil in sea ice (models/openoil/openoil.py:1179-1216; Nordam et al. 2019, Arneborg 2017):
k_ice = clip((A - 0.3) / 0.5, 0, 1) per element from sea_ice_area_fraction, advect_ocean_current(factor=1 - k_ice),
advect_wind(factor=1 - k_ice), stokes_drift((0.7 - A) / 0.7, 0 above 0.7) and advect_with_sea_ice(factor=k_ice) with the
reader's sea_ice_x/y_velocity (physics_methods.py:693-710) -- a vector pair that is rotated from the reader's projection
like the current (basereader/consts.py:27-36).

Scenario: the reference's own OpenOil (stub oil, weathering / mixing / uncertainties off) on the C4 polar-stereographic
grid with an ice edge across the domain (A from 0 to 1), 'runge-kutta4', windage, Phillips Stokes profile; surface and
submerged elements.  Stored: inputs and the live float64 state per step.

    python oracle/gen_golden_ice.py
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
import gen_golden_oil as go  # noqa: E402
from oracle.refdriver import RefStepper  # noqa: E402
from opendrift_amd import synthetic as synth  # noqa: E402


def main():
    oo = go.oo
    g = synth.grid_stere(nx=70, ny=50, nt=3, seed=16)
    nt, (ny, nx) = 3, g['land_binary_mask'].shape[1:]
    X, Y = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    g['land_binary_mask'][:] = 0
    for k in ('x_sea_water_velocity', 'y_sea_water_velocity'):
        g[k] = np.nan_to_num(g[k], nan=0.1).astype(np.float32)
    # ice edge: open water in the south-west, pack ice in the north-east, moving in time
    g['sea_ice_area_fraction'] = np.stack([np.clip(1.6 * (X + 0.6 * Y) - 0.5 + 0.05 * k, 0, 1) for k in range(nt)]).astype(np.float32)
    g['sea_ice_x_velocity'] = np.stack([0.12 * np.cos(2 * Y + 0.3 * k) for k in range(nt)]).astype(np.float32)
    g['sea_ice_y_velocity'] = np.stack([0.08 * np.sin(3 * X - 0.2 * k) for k in range(nt)]).astype(np.float32)
    times = [gg.T0 + timedelta(seconds=float(t)) for t in g['t']]
    names = [k for k in g if k not in ('x', 'y', 't')]

    oo.adios.get_oil_names = lambda location=None: ['STUB OIL']
    oo.Density = lambda oil: go._Const(go.OIL_DENSITY)
    oo.KinematicViscosity = lambda oil: go._Const(go.OIL_VISCOSITY)
    o = oo.OpenOil(loglevel=50)
    o.oiltype = go._StubOil()
    o.oil_name = 'STUB OIL'
    o.store_oil_seed_metadata = lambda **kw: None
    r = gg.GridReader(synth.NORKYST_PROJ4, g['x'], g['y'], times, {k: g[k] for k in names})
    o.add_reader(r)
    o.set_config('general:use_auto_landmask', False)
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    for p in ('evaporation', 'emulsification', 'dispersion', 'biodegradation'):
        o.set_config('processes:' + p, False)
    o.set_config('drift:vertical_mixing', False)
    o.set_config('drift:current_uncertainty', 0)
    o.set_config('drift:wind_uncertainty', 0)
    o.set_config('drift:stokes_drift', True)
    rng = np.random.default_rng(16)
    N = 300
    x = rng.uniform(g['x'][6], g['x'][-7], N)
    y = rng.uniform(g['y'][6], g['y'][-7], N)
    lon, lat = r.xy2lonlat(x, y)
    zz = np.zeros(N)
    zz[200:] = -rng.uniform(0.5, 8, 100)
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, z=zz, time=gg.T0, oil_film_thickness=0.001)
    wdf = np.array(o.elements_scheduled.wind_drift_factor * np.ones(N), dtype=np.float32)
    steps, dt = 6, 900.0
    st = RefStepper(o, dt, steps)
    res = {k: np.full((steps + 1, N), np.nan) for k in ('lon', 'lat', 'z', 'status')}
    sch = o.elements_scheduled
    res['lon'][0], res['lat'][0], res['z'][0], res['status'][0] = sch.lon, sch.lat, np.atleast_1d(sch.z) * np.ones(N), 0
    ice = np.full((steps, N), np.nan)
    for k in range(steps):
        st.step()
        assert o.num_elements_active() == N and (np.diff(o.elements.ID) > 0).all()
        lo, la, z_, s_ = st.state()
        res['lon'][k + 1], res['lat'][k + 1], res['z'][k + 1], res['status'][k + 1] = lo, la, z_, s_
        ice[k] = o.environment.sea_ice_area_fraction
    kfrac = [(ice < 0.3).mean(), ((ice >= 0.3) & (ice <= 0.8)).mean(), (ice > 0.8).mean()]
    print('c16: ice fraction classes (open / transition / pack):', np.round(kfrac, 2), 'stokes profile', o.get_config('drift:stokes_drift_profile'))
    np.savez_compressed(os.path.join(gg.GOLD, 'c16_openoil_sea_ice.npz'), dt=dt, wdf=wdf, ice_fraction=ice,
                        stokes_profile=str(o.get_config('drift:stokes_drift_profile')),
                        wind_drift_depth=float(o.get_config('drift:wind_drift_depth')),
                        **{('g_' + k): v for k, v in g.items()}, **res)


if __name__ == '__main__':
    main()
