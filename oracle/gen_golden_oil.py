"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/c9_openoil_mixing.npz from the REFERENCE'S OWN OpenOil.

SURVEY.md section 8 (f4): the per-particle oil physics that sits INSIDE the vertical-mixing loop --
`OpenOil.update_terminal_velocity` (openoil.py:922-998), `prepare_vertical_mixing` (:1017-1031: entrainment
probability after Li et al. 2017, physics_methods.py:115-137, and the droplet diameters drawn from the Johansen
et al. 2015 / Li et al. 2017 spectra, :1072-1172), `surface_stick` (:1056-1061) and `surface_wave_mixing`
(:1033-1054), called from `OceanDrift.vertical_mixing` (oceandrift.py:509,553-554).

The reference's OpenOil class is imported through oracle/refshim.py.  The ADIOS oil database (`adios_db`, not
installed) is replaced by a stub oil with a constant density / kinematic viscosity / interfacial tension: what the
database contributes to this path is three numbers per oil.  Weathering processes are switched off (they are
chemistry outside the path); `oil_weathering_noaa` still runs and does what it does in that case: converts the
temperature to Kelvin in place and re-derives density and viscosity (unchanged with a constant-property oil).

Recorded per step: every np.random draw in call order (mixing uniforms, entrainment uniforms, intrusion depths,
the uniforms behind np.random.choice), which elements were entrained in which sub-step, the live float64 state, the
droplet diameters, the entrainment probabilities and the diameters-if-entrained.

    python oracle/gen_golden_oil.py
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
from opendrift.models.openoil import openoil as oo  # noqa: E402
from opendrift.readers import reader_constant  # noqa: E402

OIL_DENSITY = 900.0                          # kg/m3, float32-representable
OIL_VISCOSITY = float(np.float32(0.005))     # m2/s
INTERFACIAL_TENSION = 0.03                   # N/m


class _Const:
    def __init__(self, v):
        self.v = v

    def at_temp(self, T):
        return self.v if np.ndim(T) == 0 else np.full(len(T), self.v)


class _StubOil:
    name = 'STUB OIL'
    oil = None
    mass_fraction = [1.0]
    bulltime = -999
    bullwinkle = 0.0
    emulsion_water_fraction_max = 0.9
    gnome_oil = {'emulsion_water_fraction_max': 0.9}

    def oil_water_surface_tension(self):
        return INTERFACIAL_TENSION

    def valid(self):
        return True


class Recorder:
    """np.random.random / uniform / choice in call order.  np.random.choice(a, size, p=p) draws `size` uniforms
    from the same stream and looks them up in the normalised cumulative sum (legacy RandomState.choice); it is
    replaced by exactly that so that the uniforms can be stored, and checked against the original on a copy of
    the generator state."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._orig = (np.random.random, np.random.uniform, np.random.choice)
        rec = self

        def random(size=None):
            r = rec._orig[0](size)
            rec.draws.append(('random', np.array(r, copy=True)))
            return r

        def uniform(low=0.0, high=1.0, size=None):
            r = rec._orig[1](low, high, size)
            rec.draws.append(('uniform', np.array(r, copy=True), float(low), float(high)))
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

        np.random.random, np.random.uniform, np.random.choice = random, uniform, choice
        return self

    def __exit__(self, *a):
        np.random.random, np.random.uniform, np.random.choice = self._orig


def c9_openoil(distribution='Johansen et al. (2015)', tag='johansen'):
    rng = np.random.default_rng(9)
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
    g['x_sea_water_velocity'] = np.stack([0.2 * np.cos(3 * Y + k) for k in range(nt)]).astype(np.float32)
    g['y_sea_water_velocity'] = np.stack([0.2 * np.sin(3 * X - k) for k in range(nt)]).astype(np.float32)
    g['sea_water_temperature'] = np.stack([(4 + 9 * Y + 0.5 * k) for k in range(nt)]).astype(np.float32)
    g['sea_water_salinity'] = np.stack([(30 + 5 * X)] * nt).astype(np.float32)
    names = [k for k in g if k not in ('x', 'y', 't')]
    N = 300
    lon = rng.uniform(x[3], x[-4], N)
    lat = rng.uniform(y[3], y[-4], N)
    zz = -rng.uniform(1, 40, N)
    zz[:170] = 0.0
    diam = rng.uniform(2e-5, 3e-3, N)            # the droplets seeded below the surface
    diam[:170] = 0.0

    oo.adios.get_oil_names = lambda location=None: ['STUB OIL']
    oo.Density = lambda oil: _Const(OIL_DENSITY)
    oo.KinematicViscosity = lambda oil: _Const(OIL_VISCOSITY)
    o = oo.OpenOil(loglevel=50)
    o.oiltype = _StubOil()
    o.oil_name = 'STUB OIL'
    o.store_oil_seed_metadata = lambda **kw: None
    o.add_reader(gg.GridReader('+proj=latlong', x, y, times, {k: g[k] for k in names}))
    o.set_config('general:use_auto_landmask', False)
    o.set_config('environment:fallback:land_binary_mask', 0)
    o.set_config('drift:advection_scheme', 'euler')
    for p in ('evaporation', 'emulsification', 'dispersion', 'biodegradation'):
        o.set_config('processes:' + p, False)
    o.set_config('drift:current_uncertainty', 0)
    o.set_config('drift:wind_uncertainty', 0)
    o.set_config('drift:stokes_drift', False)
    o.set_config('vertical_mixing:timestep', 60)
    assert o.get_config('vertical_mixing:background_diffusivity') == 1.2e-5
    o.set_config('wave_entrainment:droplet_size_distribution', distribution)
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, z=zz, time=gg.T0, wind_drift_factor=0.0, oil_film_thickness=0.001)
    assert o.keep_droplet_diameter is False
    # the droplets of the elements seeded below the surface (seed_elements draws them itself from
    # seed:droplet_diameter_min/max_subsea; fixed here so that the fixture does not depend on that draw)
    o.elements_scheduled.diameter = diam.astype(np.float32)
    o.elements_scheduled.oil_film_thickness = (0.0005 + 0.001 * rng.uniform(0, 1, N)).astype(np.float32)

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
    keys = ('lon', 'lat', 'z', 'status', 'diameter', 'terminal_velocity', 'probability', 'diameter_if_entrained')
    res = {k: np.full((steps + 1, N), np.nan) for k in keys}
    sch = o.elements_scheduled                       # float32 positions at seeding, like oracle/gen_golden.py:_run
    res['lon'][0], res['lat'][0], res['z'][0], res['status'][0] = sch.lon, sch.lat, np.atleast_1d(sch.z) * np.ones(N), 0
    res['diameter'][0] = diam.astype(np.float32)
    u_mix = np.empty((steps, nsub, N))
    u_ent = np.empty((steps, nsub, N))
    u_int = np.full((steps, nsub, N), np.nan)        # expanded to element positions
    u_dia = np.empty((steps, N))
    idx_dia = np.empty((steps, N), np.int64)
    ent = np.zeros((steps, nsub, N), bool)
    mean_zb = np.full((steps, nsub), np.nan)
    for k in range(steps):
        del entrained_log[:]
        with Recorder() as rr:
            st.step()
        assert o.num_elements_active() == N and (np.diff(o.elements.ID) > 0).all()
        d = rr.draws
        assert d[0][0] == 'choice'
        u_dia[k], idx_dia[k] = d[0][1], d[0][2]
        j = 1
        for s in range(nsub):
            assert d[j][0] == 'random' and d[j + 1][0] == 'uniform' and d[j + 1][2:] == (0.0, 1.0)
            u_mix[k, s], u_ent[k, s] = d[j][1], d[j + 1][1]
            j += 2
            m = entrained_log[s]
            ent[k, s] = m
            if m.sum() > 0:
                assert d[j][0] == 'uniform' and len(d[j][1]) == m.sum() and d[j][2] == 0.0
                mean_zb[k, s] = d[j][3]
                u_int[k, s, m] = d[j][1] / d[j][3]    # np.random.uniform(0, h, k) = h * random_sample(k)
                j += 1
        assert j == len(d)
        lo, la, z_, s_ = st.state()
        res['lon'][k + 1], res['lat'][k + 1], res['z'][k + 1], res['status'][k + 1] = lo, la, z_, s_
        res['diameter'][k + 1] = o.elements.diameter
        res['terminal_velocity'][k + 1] = o.elements.terminal_velocity
        res['probability'][k + 1] = o.oil_entrainment_probability
        res['diameter_if_entrained'][k + 1] = o.droplet_diameter_if_entrained
    print(tag, 'entrained per step', ent.sum(axis=(1, 2)), 'z range', np.nanmin(res['z']), 'surface at end',
          int((res['z'][-1] == 0).sum()), 'density dtype', o.elements.density.dtype)
    out = {tag + '_' + k: v for k, v in res.items()}
    out.update({tag + '_u_mix': u_mix, tag + '_u_entrain': u_ent, tag + '_u_intrusion': u_int, tag + '_u_diameter': u_dia,
                tag + '_idx_diameter': idx_dia, tag + '_entrained': ent, tag + '_mean_zb': mean_zb})
    return g, out, dict(film=np.asarray(o.elements.oil_film_thickness, dtype=np.float32))


def c12_kelvin():
    """Environment.get_environment's unit check (environment.py:829-838): sea_water_temperature above 100 is taken as
    Kelvin and converted to Celsius per element.  A reader whose temperature field is in Kelvin in the western half
    of the domain and in Celsius in the eastern half; the float32 environment right after the reference's
    get_environment is stored."""
    rng = np.random.default_rng(12)
    nx, ny, nt = 40, 32, 2
    x = np.linspace(2, 8, nx).astype(np.float32)
    y = np.linspace(59, 63, ny).astype(np.float32)
    X, Y = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    times = [gg.T0 + timedelta(seconds=3600.0 * k) for k in range(nt)]
    T = np.stack([(4 + 9 * Y + 0.5 * k + np.where(X < 0.5, 273.15, 0.0)) for k in range(nt)]).astype(np.float32)
    g = dict(x=x, y=y, t=np.arange(nt) * 3600.0, sea_water_temperature=T,
             x_sea_water_velocity=np.zeros_like(T), y_sea_water_velocity=np.zeros_like(T),
             x_wind=np.zeros_like(T), y_wind=np.zeros_like(T))
    oo.adios.get_oil_names = lambda location=None: ['STUB OIL']
    oo.Density = lambda oil: _Const(OIL_DENSITY)
    oo.KinematicViscosity = lambda oil: _Const(OIL_VISCOSITY)
    o = oo.OpenOil(loglevel=50)
    o.oiltype = _StubOil()
    o.oil_name = 'STUB OIL'
    o.store_oil_seed_metadata = lambda **kw: None
    o.add_reader(gg.GridReader('+proj=latlong', x, y, times, {k: v for k, v in g.items() if k not in ('x', 'y', 't')}))
    o.set_config('general:use_auto_landmask', False)
    o.set_config('environment:fallback:land_binary_mask', 0)
    for p in ('evaporation', 'emulsification', 'dispersion', 'biodegradation'):
        o.set_config('processes:' + p, False)
    o.set_config('drift:vertical_mixing', False)
    o.set_config('drift:stokes_drift', False)
    o.set_config('drift:current_uncertainty', 0)
    o.set_config('drift:wind_uncertainty', 0)
    N = 200
    lon, lat = rng.uniform(x[2], x[-3], N), rng.uniform(y[2], y[-3], N)
    o.seed_elements(lon=lon, lat=lat, z=0.0, time=gg.T0, wind_drift_factor=0.0)
    st = RefStepper(o, 1800.0, 2)
    seen = []
    orig = o.calculate_missing_environment_variables

    def hook():
        seen.append(np.array(o.environment.sea_water_temperature, copy=True))
        orig()
    o.calculate_missing_environment_variables = hook
    st.step()
    st.step()
    sch_lon, sch_lat = np.array(lon, dtype=np.float32).astype(np.float64), np.array(lat, dtype=np.float32).astype(np.float64)
    assert seen[0].dtype == np.float32 and (seen[0] < 100).all() and (T[0][:, :nx // 2] > 100).all()
    np.savez_compressed(os.path.join(gg.GOLD, 'c12_kelvin_environment.npz'), g_x=x, g_y=y, g_t=g['t'], g_T=T,
                        lon=sch_lon, lat=sch_lat, T_env_step0=seen[0], T_env_step1=seen[1], dt=1800.0)


def main():
    out = {}
    for dist, tag in (('Johansen et al. (2015)', 'johansen'), ('Li et al. (2017)', 'li')):
        g, o1, extra = c9_openoil(dist, tag)
        out.update(o1)
    np.savez_compressed(os.path.join(gg.GOLD, 'c9_openoil_mixing.npz'), dt=600.0, dt_mix=60.0,
                        background_diffusivity=1.2e-5, oil_density=OIL_DENSITY, oil_viscosity=OIL_VISCOSITY, interfacial_tension=INTERFACIAL_TENSION,
                        film=extra['film'], **{('g_' + k): v for k, v in g.items()}, **out)


if __name__ == '__main__':
    if 'c12' in sys.argv[1:]:
        c12_kelvin()
    else:
        main()
