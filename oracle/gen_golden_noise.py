"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/c13_noise_rk.npz and c14_openoil_defaults.npz from the REFERENCE ITSELF.

`drift:current_uncertainty`, `drift:current_uncertainty_uniform` and `drift:wind_uncertainty` are applied inside
Environment.get_environment (environment.py:869-891) -- in EVERY call whose variables hold the current, i.e. also in
the one (RK2) / three (RK4) stage calls of advect_ocean_current (physics_methods.py:638-670).  OpenOil's defaults are
current_uncertainty = 0.05 and wind_uncertainty = 0.5 (openoil.py:497-498), so a default OpenOil run with a
Runge-Kutta scheme draws 2 + 2 + 3 x 2 normal arrays per step besides the mixing draws.

  c13  OceanDrift on a 3-D lon/lat grid + a constant wind: (a) 'runge-kutta' with current_uncertainty 0.05 and
       wind_uncertainty 1.0, (b) 'runge-kutta4' with additionally current_uncertainty_uniform 0.03.
  c14  the reference's own OpenOil (stub oil, weathering off) with its DEFAULT uncertainties, 'runge-kutta4', vertical
       mixing with wave entrainment (wind-parameterised diffusivity), windage.

Stored: inputs, every np.random draw in call order sorted into (main sample | stage calls | mixing), the live float64
state per step.

    python oracle/gen_golden_noise.py [c13] [c14]
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
from opendrift.readers import reader_constant  # noqa: E402


class Recorder:
    """np.random.random / uniform / normal / choice in call order (choice restated as in gen_golden_oil.Recorder)."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._orig = (np.random.random, np.random.uniform, np.random.choice, np.random.normal)
        rec = self

        def random(size=None):
            r = rec._orig[0](size)
            rec.draws.append(('random', np.array(r, copy=True)))
            return r

        def uniform(low=0.0, high=1.0, size=None):
            r = rec._orig[1](low, high, size)
            rec.draws.append(('uniform', np.array(r, copy=True), float(low), float(high)))
            return r

        def normal(loc=0.0, scale=1.0, size=None):
            r = rec._orig[3](loc, scale, size)
            rec.draws.append(('normal', np.array(r, copy=True), float(loc), float(scale)))
            return r

        def choice(a, size=None, replace=True, p=None):
            state = np.random.get_state()
            want = rec._orig[2](a, size=size, replace=replace, p=p)
            np.random.set_state(state)
            u = rec._orig[0](size)
            cdf = np.cumsum(p)
            cdf /= cdf[-1]
            idx = cdf.searchsorted(u, side='right')
            got = np.asarray(a)[idx]
            assert np.array_equal(got, want), 'np.random.choice restatement differs'
            rec.draws.append(('choice', np.array(u, copy=True), idx.astype(np.int64)))
            return got

        np.random.random, np.random.uniform, np.random.choice, np.random.normal = random, uniform, choice, normal
        return self

    def __exit__(self, *a):
        np.random.random, np.random.uniform, np.random.choice, np.random.normal = self._orig


def _pad(a, n):
    out = np.full(n, np.nan)
    out[:len(a)] = a
    return out


def c13_noise_rk():
    g = synth.grid3d(nx=48, ny=40, nz=8, nt=3, seed=13)
    times = [gg.T0 + timedelta(seconds=float(t)) for t in g['t']]
    names = ('x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity',
             'sea_floor_depth_below_sea_level', 'land_binary_mask')
    arrays = {k: g[k] for k in names}
    rng = np.random.default_rng(13)
    N = 300
    lon = rng.uniform(g['x'][4], g['x'][-5], N)
    lat = rng.uniform(g['y'][4], g['y'][-5], N)
    zz = rng.uniform(-40, 0, N)
    zz[:60] = 0.0
    out = {}
    for tag, scheme, std_u in (('rk2', 'runge-kutta', 0.0), ('rk4', 'runge-kutta4', 0.03)):
        o = gg._base(scheme)
        o.add_reader(gg.GridReader('+proj=latlong', g['x'], g['y'], times, arrays, z=g['z']))
        o.add_reader(reader_constant.Reader({'x_wind': 7.0, 'y_wind': -4.0}))
        o.set_config('drift:current_uncertainty', 0.05)
        o.set_config('drift:current_uncertainty_uniform', std_u)
        o.set_config('drift:wind_uncertainty', 1.0)
        o.set_config('drift:vertical_mixing', False)
        o.set_config('drift:vertical_advection', True)
        o.set_config('drift:stokes_drift', False)
        o.set_config('general:coastline_action', 'previous')
        np.random.seed(0)
        o.seed_elements(lon=lon, lat=lat, z=zz, time=gg.T0, wind_drift_factor=0.02)
        steps, dt = 6, 600.0
        nstage = 1 if scheme == 'runge-kutta' else 3
        ncomp = 4 if std_u > 0 else 2
        st = RefStepper(o, dt, steps)
        res = {k: np.full((steps + 1, N), np.nan) for k in ('lon', 'lat', 'z')}
        res['status'] = np.zeros((steps + 1, N), np.int32)
        sch = o.elements_scheduled
        res['lon'][0], res['lat'][0], res['z'][0] = sch.lon, sch.lat, np.atleast_1d(sch.z) * np.ones(N)
        main = np.full((steps, ncomp + 2, N), np.nan)        # current normal x, y [, uniform x, y], wind normal x, y
        stage = np.full((steps, nstage, ncomp, N), np.nan)
        for k in range(steps):
            with Recorder() as rr:
                st.step()
            d = rr.draws
            kinds = [x[0] for x in d]
            want = (['normal', 'normal'] + (['uniform', 'uniform'] if std_u > 0 else [])) + ['normal', 'normal'] + \
                (['normal', 'normal'] + (['uniform', 'uniform'] if std_u > 0 else [])) * nstage
            assert kinds == want, (kinds, want)
            for j in range(ncomp + 2):
                main[k, j] = _pad(d[j][1], N)
            j = ncomp + 2
            for s in range(nstage):
                for c in range(ncomp):
                    stage[k, s, c] = _pad(d[j][1], N)
                    j += 1
            res['lon'][k + 1], res['lat'][k + 1], res['z'][k + 1], res['status'][k + 1] = st.state()
        out.update({tag + '_' + k: v for k, v in res.items()})
        out[tag + '_main_noise'], out[tag + '_stage_noise'] = main, stage
        out[tag + '_categories'] = np.array(o.status_categories)
        print(tag, 'active at the end', o.num_elements_active(), o.status_categories)
    np.savez_compressed(os.path.join(gg.GOLD, 'c13_noise_rk.npz'), dt=600.0, wind=np.array([7.0, -4.0]),
                        current_uncertainty=0.05, current_uncertainty_uniform=0.03, wind_uncertainty=1.0,
                        **{('g_' + k): v for k, v in g.items()}, **out)


def c14_openoil_defaults():
    oo = go.oo
    rng = np.random.default_rng(14)
    nx, ny, nt = 40, 32, 3
    x = np.linspace(2, 8, nx).astype(np.float32)
    y = np.linspace(59, 63, ny).astype(np.float32)
    X, Y = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    t = np.arange(nt) * 3600.0
    times = [gg.T0 + timedelta(seconds=float(v)) for v in t]
    g = dict(x=x, y=y, t=t)
    g['x_wind'] = np.stack([(10 + 6 * np.sin(3 * X + k)) for k in range(nt)]).astype(np.float32)
    g['y_wind'] = np.stack([(4 * np.cos(4 * Y - k)) for k in range(nt)]).astype(np.float32)
    g['ocean_mixed_layer_thickness'] = np.stack([(30 + 22.5 * (1 + np.sin(2 * X + 3 * Y)))] * nt).astype(np.float32)
    g['sea_floor_depth_below_sea_level'] = np.stack([(25 + 150 * X)] * nt).astype(np.float32)
    g['x_sea_water_velocity'] = np.stack([0.3 * np.cos(3 * Y + k) for k in range(nt)]).astype(np.float32)
    g['y_sea_water_velocity'] = np.stack([0.3 * np.sin(3 * X - k) for k in range(nt)]).astype(np.float32)
    g['sea_water_temperature'] = np.stack([(4 + 9 * Y + 0.5 * k) for k in range(nt)]).astype(np.float32)
    g['sea_water_salinity'] = np.stack([(30 + 5 * X)] * nt).astype(np.float32)
    names = [k for k in g if k not in ('x', 'y', 't')]
    N = 300
    lon = rng.uniform(x[3], x[-4], N)
    lat = rng.uniform(y[3], y[-4], N)
    zz = -rng.uniform(1, 40, N)
    zz[:170] = 0.0
    diam = rng.uniform(2e-5, 3e-3, N)
    diam[:170] = 0.0

    oo.adios.get_oil_names = lambda location=None: ['STUB OIL']
    oo.Density = lambda oil: go._Const(go.OIL_DENSITY)
    oo.KinematicViscosity = lambda oil: go._Const(go.OIL_VISCOSITY)
    o = oo.OpenOil(loglevel=50)
    o.oiltype = go._StubOil()
    o.oil_name = 'STUB OIL'
    o.store_oil_seed_metadata = lambda **kw: None
    o.add_reader(gg.GridReader('+proj=latlong', x, y, times, {k: g[k] for k in names}))
    o.set_config('general:use_auto_landmask', False)
    o.set_config('environment:fallback:land_binary_mask', 0)
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    for p in ('evaporation', 'emulsification', 'dispersion', 'biodegradation'):
        o.set_config('processes:' + p, False)
    # OpenOil's defaults, untouched (openoil.py:493-499)
    assert o.get_config('drift:current_uncertainty') == 0.05 and o.get_config('drift:wind_uncertainty') == 0.5
    assert o.get_config('drift:vertical_mixing') is True and o.get_config('drift:current_uncertainty_uniform') == 0
    o.set_config('vertical_mixing:timestep', 60)
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, z=zz, time=gg.T0, oil_film_thickness=0.001)
    wdf = np.array(o.elements_scheduled.wind_drift_factor * np.ones(N), dtype=np.float32)
    o.elements_scheduled.diameter = diam.astype(np.float32)
    o.elements_scheduled.oil_film_thickness = (0.0005 + 0.001 * rng.uniform(0, 1, N)).astype(np.float32)
    film = np.array(o.elements_scheduled.oil_film_thickness, dtype=np.float32)

    entrained_log = []
    orig_swm = o.surface_wave_mixing

    def swm(dt_):
        before = np.array(o.elements.z, copy=True)
        orig_swm(dt_)
        entrained_log.append((before >= 0) & (o.elements.z < 0))
    o.surface_wave_mixing = swm

    steps, dt, nsub = 6, 600.0, 10
    st = RefStepper(o, dt, steps)
    assert st.n_total == N
    keys = ('lon', 'lat', 'z', 'status', 'diameter')
    res = {k: np.full((steps + 1, N), np.nan) for k in keys}
    sch = o.elements_scheduled
    res['lon'][0], res['lat'][0], res['z'][0], res['status'][0] = sch.lon, sch.lat, np.atleast_1d(sch.z) * np.ones(N), 0
    res['diameter'][0] = diam.astype(np.float32)
    main = np.empty((steps, 4, N))                      # current normal x, y; wind normal x, y
    stage = np.empty((steps, 3, 2, N))
    u_mix = np.empty((steps, nsub, N))
    u_ent = np.empty((steps, nsub, N))
    u_int = np.full((steps, nsub, N), np.nan)
    u_dia = np.empty((steps, N))
    for k in range(steps):
        del entrained_log[:]
        with Recorder() as rr:
            st.step()
        assert o.num_elements_active() == N and (np.diff(o.elements.ID) > 0).all()
        d = rr.draws
        assert [x[0] for x in d[:4]] == ['normal'] * 4
        for j in range(4):
            main[k, j] = d[j][1]
        assert (d[0][3], d[2][3]) == (0.05, 0.5)
        assert d[4][0] == 'choice'
        u_dia[k] = d[4][1]
        j = 5
        for s in range(nsub):
            assert d[j][0] == 'random' and d[j + 1][0] == 'uniform' and d[j + 1][2:] == (0.0, 1.0)
            u_mix[k, s], u_ent[k, s] = d[j][1], d[j + 1][1]
            j += 2
            m = entrained_log[s]
            if m.sum() > 0:
                assert d[j][0] == 'uniform' and len(d[j][1]) == m.sum() and d[j][2] == 0.0
                u_int[k, s, m] = d[j][1] / d[j][3]
                j += 1
        for s in range(3):
            assert d[j][0] == 'normal' and d[j + 1][0] == 'normal' and d[j][3] == 0.05
            stage[k, s, 0], stage[k, s, 1] = d[j][1], d[j + 1][1]
            j += 2
        assert j == len(d), (j, len(d), [x[0] for x in d[j:]])
        lo, la, z_, s_ = st.state()
        res['lon'][k + 1], res['lat'][k + 1], res['z'][k + 1], res['status'][k + 1] = lo, la, z_, s_
        res['diameter'][k + 1] = o.elements.diameter
    print('c14: surface at end', int((res['z'][-1] == 0).sum()), 'z range', np.nanmin(res['z']))
    np.savez_compressed(os.path.join(gg.GOLD, 'c14_openoil_defaults.npz'), dt=dt, dt_mix=60.0,
                        background_diffusivity=1.2e-5, oil_density=go.OIL_DENSITY, oil_viscosity=go.OIL_VISCOSITY,
                        interfacial_tension=go.INTERFACIAL_TENSION, film=film, wdf=wdf,
                        main_noise=main, stage_noise=stage, u_mix=u_mix, u_entrain=u_ent, u_intrusion=u_int,
                        u_diameter=u_dia, **{('g_' + k): v for k, v in g.items()}, **res)


if __name__ == '__main__':
    which = sys.argv[1:] or ['c13', 'c14']
    if 'c13' in which:
        c13_noise_rk()
    if 'c14' in which:
        c14_openoil_defaults()
