"""Replays the reference loop body (basemodel/__init__.py:2193-2284 + OceanDrift.update,
oceandrift.py:185-211) on two interchangeable back ends:

  OracleBackend -- NumPy arrays + the C oracle (CPU checker)
  DeviceBackend -- opendrift_amd.device (the product path)

so that the same scenario script is compared with the golden vectors written by the
reference itself (oracle/gen_golden.py).
"""
import numpy as np

from oracle import oracle as orc

V = orc.VAR
U, VV, W = 'x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity'
KZ, DEPTH, SSH, LAND = ('ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level',
                        'sea_surface_height', 'land_binary_mask')
XW, YW, MLD = 'x_wind', 'y_wind', 'ocean_mixed_layer_thickness'
SX, SY = 'sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity'
HS, HD = 'sea_surface_wave_significant_height', 'horizontal_diffusivity'
TEMP, SALT = 'sea_water_temperature', 'sea_water_salinity'
ICE_A, ICE_U, ICE_V = 'sea_ice_area_fraction', 'sea_ice_x_velocity', 'sea_ice_y_velocity'
OIL_PROPS = ['diameter', 'density', 'viscosity', 'oil_film_thickness', 'diameter_if_entrained']   # property slots
LEEWAY_PROPS = ['downwind_slope', 'crosswind_slope', 'downwind_offset', 'crosswind_offset', 'downwind_eps',
                'crosswind_eps', 'jibe_probability', 'orientation', 'capsized']


def _is_float32_state(lon, lat):
    lon, lat = np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64)
    return bool(np.array_equal(lon, lon.astype(np.float32)) and np.array_equal(lat, lat.astype(np.float32)))


class OracleBackend:
    def __init__(self, scenario, lon, lat, z, wdf=0.02):
        self.sc = scenario
        n = len(lon)
        self.lon, self.lat, self.z = lon.copy(), lat.copy(), np.array(z, dtype=np.float64) * np.ones(n)
        self.ID = np.arange(n, dtype=np.int32)
        self.status = np.zeros(n, np.int32)
        self.moving = np.ones(n, np.int32)
        self.wdf = np.full(n, wdf, np.float32)
        self.cdf = np.ones(n, np.float32)
        self.tv = np.zeros(n, np.float32)
        self.age = np.zeros(n, np.float32)
        self.plon, self.plat = lon.copy(), lat.copy()
        self.dead = dict(ID=[], lon=[], lat=[], z=[], status=[])
        self.env = {}

    def sample(self, names, t, profile=None, nzp=0):
        w = self.sc.oracle_world()  # fresh blocks: the oracle's NaN dilation is stateful
        # the FIRST get_environment of a reference run works on float32 element arrays (elements.py:71-88 -> modulate_longitude
        # in float32, variables.py:259-280): a replay that starts from the seeded state (float32 values) does the same
        first = not getattr(self, '_sampled_once', False) and _is_float32_state(self.lon, self.lat)
        self._sampled_once = True
        orc.set_position_class(first)
        try:
            vals = orc.get_environment(w, [V[k] for k in names], self.lon, self.lat, self.z, t)
            self.env = dict(zip(names, vals))
            if profile:
                self.Kp = orc.get_profile(w, V[profile], self.lon, self.lat, t, nzp)
        finally:
            orc.set_position_class(False)
        self._w = w

    def truncate(self, depth):   # drift:truncate_ocean_model_below_m (environment.py:554-566): the sampling calls see max(z, -depth)
        self._z_true = self.z
        self.z = np.maximum(self.z, -float(depth))

    def restore(self):
        self.z = self._z_true

    def report_missing(self, names, code):   # basemodel/__init__.py:2501-2515, environment.py:903-908
        missing = np.zeros(len(self.lon), bool)
        for k in names:
            missing |= np.isnan(self.env[k])
        self.status[missing & (self.status == 0)] = code
        self.moving[missing] = 0

    def coast(self, action, code=1, seeded_code=0):
        orc.coastline({'stranding': 1, 'previous': 2}[action], self.env[LAND], self.lon, self.lat, self.z,
                      self.plon, self.plat, self.status, self.moving, code, self.age, seeded_code)

    def sample_landmask(self, mask):
        """land_binary_mask from the raster landmask reader (exact at the element positions)"""
        self.env[LAND] = mask.land_binary_mask(self.lon, self.lat)

    def coast_crossing(self, action, precision, mask, code=1):
        """interact_with_coastline with general:coastline_approximation_precision (basemodel/__init__.py:694-746)"""
        from oracle import landmask
        land = self.env[LAND] == 1
        on = np.where(land)[0]
        if len(on) == 0:
            return
        if action == 'stranding':
            hit = land & (self.z <= 0)
            self.status[hit & (self.status == 0)] = code
            self.moving[hit] = 0
        lc, la = landmask.coastline_crossing(mask, self.plon[on], self.plat[on], self.lon[on], self.lat[on], precision,
                                             land_side=(action == 'stranding'))
        self.lon[on], self.lat[on] = lc, la
        self.env[LAND][on] = 0

    def increase_age(self, dt):
        self.age = (self.age + np.float32(dt)).astype(np.float32)

    def seafloor(self, action='lift_to_seafloor', code=1):   # basemodel/__init__.py:748-783
        floor = -(self.env[DEPTH] + self.env.get(SSH, np.float32(0)))
        below = self.z < floor
        if action == 'previous':
            self.lon[below], self.lat[below] = self.plon[below], self.plat[below]
            return
        if action == 'deactivate':
            self.status[below & (self.status == 0)] = code
            self.moving[below] = 0
        self.z[below] = floor[below]

    def compact(self):
        keep = self.status == 0
        if keep.all():
            return
        for k in ('ID', 'lon', 'lat', 'z', 'status'):
            self.dead[k].append(getattr(self, k)[~keep])
        for k in ('lon', 'lat', 'z', 'ID', 'status', 'moving', 'wdf', 'cdf', 'tv', 'age', 'plon', 'plat'):
            setattr(self, k, getattr(self, k)[keep])
        self.env = {k: v[keep] for k, v in self.env.items()}
        if hasattr(self, 'Kp'):
            self.Kp = np.ascontiguousarray(self.Kp[:, keep])
        if hasattr(self, 'aux'):
            self.aux = [np.ascontiguousarray(a[keep]) for a in self.aux]

    def store_previous(self):
        self.plon, self.plat = self.lon.copy(), self.lat.copy()

    def advect(self, scheme, t, dt, stage_noise=None, stds=None, ice=False):
        """stage_noise [nstage][ncomp][n]: np.random draws of the Runge-Kutta stage calls (current uncertainty);
        ice: factor = 1 - k_ice per element (OpenOil.advect_oil, openoil.py:1207)"""
        w = self.sc.oracle_world()
        factor = 1 - orc.ice_factors(self.env[ICE_A])[0] if ice else 1.0
        orc.advect_ocean_current(w, {'euler': 0, 'runge-kutta': 1, 'runge-kutta4': 2}[scheme], self.lon, self.lat,
                                 self.z, self.moving, self.cdf, self.env[U], self.env[VV], t, dt, factor=factor,
                                 stage_noise=stage_noise)

    def vadvect(self, dt):
        orc.vertical_advection(self.z, self.moving, self.env[W], dt)

    def wind(self, dt, wdd=0.1, relative=False, ice=False):
        factor = 1 - orc.ice_factors(self.env[ICE_A])[0] if ice else 1.0
        orc.advect_wind(self.lon, self.lat, self.z, self.moving, self.wdf, self.env[XW], self.env[YW],
                        self.env[U], self.env[VV], wdd, int(relative), factor, dt)

    def stokes(self, dt, profile=2, hs_mode=1, tp_mode=1, ice=False):
        z = np.zeros_like(self.env[SX])
        factor = orc.ice_factors(self.env[ICE_A])[1] if ice else 1.0
        orc.stokes_drift(self.lon, self.lat, self.z, self.moving, self.env[SX], self.env[SY],
                         self.env.get(HS, z), z, self.env[XW], self.env[YW], hs_mode, tp_mode, profile, factor, dt)

    def ice_drift(self, dt):   # advect_with_sea_ice(factor=k_ice), physics_methods.py:693-697: float32 products
        k = orc.ice_factors(self.env[ICE_A])[0]
        orc.update_positions(self.lon, self.lat, k * self.env[ICE_U], k * self.env[ICE_V], self.moving, dt)

    def hdiff(self, dt, normals):
        n = len(self.lon)
        orc.horizontal_diffusion(self.lon, self.lat, self.moving, self.env[HD], normals[0][:n], normals[1][:n], dt)

    def set_leeway(self, props):
        self.aux = [np.array(props[k], dtype=np.float32) for k in LEEWAY_PROPS]

    def noise(self, vx, vy, nx, ny, uniform=False):   # environment.py:869-891: float32 += float64
        n = len(self.lon)
        self.env[vx] = (self.env[vx].astype(np.float64) + nx[:n]).astype(np.float32)
        self.env[vy] = (self.env[vy].astype(np.float64) + ny[:n]).astype(np.float32)

    def leeway(self, dt, uniforms, frac=0.4, cap_uniforms=None, thr=30.0, sig=5.0):
        n = len(self.lon)
        cu = None
        if cap_uniforms is not None:
            cu = cap_uniforms[:int((self.aux[8] == (0.0 if dt >= 0 else 1.0)).sum())]
        orc.leeway(self.lon, self.lat, self.moving, self.aux, self.env[XW], self.env[YW], self.env[U], self.env[VV],
                   dt, frac, uniforms[:n], cap_uniforms=cu, wind_threshold=thr, wind_threshold_sigma=sig)

    def vmix(self, t, dt, dt_mix, zlevels, uniforms, vadv=True, levels=0):
        # levels: the reader's block was cut there (a reader that hands out the levels asked for): the column the mixing sees
        zl, Kp = (zlevels, self.Kp) if not levels else (np.ascontiguousarray(zlevels[:levels]), np.ascontiguousarray(self.Kp[:levels]))
        orc.vertical_mixing(self.z, self.moving, self.tv, self.env[DEPTH], self.env[SSH], zl, Kp, dt,
                            dt_mix, 0, uniforms)
        if vadv:
            orc.vertical_advection(self.z, self.moving, self.env[W], dt)

    def vbuoy(self, dt, action='lift_to_seafloor', code=1):   # vertical_buoyancy, oceandrift.py:352-368
        oc = self.z < 0
        self.z[oc] = np.minimum(0, self.z[oc] + self.tv[oc] * dt)
        Zmin = -1. * (self.env[DEPTH] + self.env[SSH])
        if (self.z < Zmin).any() and action != 'none':
            self.seafloor(action, code)      # interact_with_seafloor() again, inside update()

    def vmix_analytic(self, model, background, dt, dt_mix, uniforms):
        from oracle import diffusivity
        zlev, Kp = diffusivity.profiles(model, self.env[XW], self.env[YW], self.env[MLD], background)
        orc.vertical_mixing(self.z, self.moving, self.tv, self.env[DEPTH], self.env[SSH], zlev,
                            np.ascontiguousarray(Kp), dt, dt_mix, 0, uniforms)

    def set_oil(self, diameter, density, viscosity, film):
        n = len(self.lon)
        self.oil = dict(diameter=np.asarray(diameter, np.float32) * np.ones(n, np.float32),
                        density=np.asarray(density, np.float64) * np.ones(n),        # float64 after oil_weathering_noaa
                        viscosity=np.asarray(viscosity, np.float64) * np.ones(n),
                        film=np.asarray(film, np.float32) * np.ones(n, np.float32))

    def vmix_oil(self, model, background, dt, dt_mix, interfacial_tension, distribution, uniforms, t=0.0, zlevels=None):
        """OpenOil: oil_weathering_noaa's Kelvin conversion (openoil.py:722-724), prepare_vertical_mixing and the
        mixing loop with the oil physics (oracle/oil.py); no wave height / period from readers (from the wind)."""
        from oracle import diffusivity, oil
        o, e = self.oil, self.env
        T = e[TEMP].copy()
        T[T < 100] += 273.15
        hs = oil.significant_wave_height(e[XW], e[YW])
        wbf = oil.wave_breaking_fraction(e[XW], e[YW])
        self.probability = oil.entrainment_probability(o['density'], o['viscosity'], interfacial_tension, hs, wbf, dt_mix)
        if distribution == 'Johansen et al. (2015)':
            self.dV_50 = oil.droplet_median_johansen2015(o['density'], o['viscosity'], o['film'], hs, interfacial_tension)
        else:
            self.dV_50 = oil.droplet_median_li2017(o['density'], o['viscosity'], hs, interfacial_tension)
        self.diameter_if_entrained = oil.droplet_diameters(self.dV_50, uniforms['diameter'])
        self.mean_zb = np.mean(1.5 * hs)
        if model == 'environment':      # profiles from a reader (B.sample(..., profile=KZ, nzp=...))
            zlev, Kp = np.asarray(zlevels, dtype=np.float64), np.ascontiguousarray(self.Kp)
        else:
            zlev, Kp = diffusivity.profiles(model, e[XW], e[YW], e[MLD], background)
        self.w = oil.vertical_mixing_oil(self.z, self.moving, o['diameter'], o['density'], T, e[SALT], e[DEPTH], e[SSH],
                                         zlev, Kp, dt, dt_mix, self.probability, self.diameter_if_entrained,
                                         self.mean_zb, uniforms['mix'], uniforms['entrain'], uniforms['intrusion'])

    def oil_state(self):
        return dict(diameter=self.oil['diameter'].copy(), diameter_if_entrained=self.diameter_if_entrained.copy(),
                    terminal_velocity=self.w.copy())

    def state(self, n_total):
        lon, lat, z = np.full(n_total, np.nan), np.full(n_total, np.nan), np.full(n_total, np.nan)
        status = np.full(n_total, -1, np.int32)
        lon[self.ID], lat[self.ID], z[self.ID], status[self.ID] = self.lon, self.lat, self.z, self.status
        for ID, lo, la, zz, st in zip(*[self.dead[k] for k in ('ID', 'lon', 'lat', 'z', 'status')]):
            lon[ID], lat[ID], z[ID], status[ID] = lo, la, zz, st
        return lon, lat, z, status


class DeviceBackend:
    def __init__(self, scenario, ctx, lon, lat, z, wdf=0.02):
        self.sc, self.ctx = scenario, ctx
        scenario.device(ctx)
        n = len(lon)
        self.P = ctx.particles(n)
        self.P.append(lon, lat, z=np.array(z, dtype=np.float64) * np.ones(n), wind_drift_factor=np.full(n, wdf, np.float32))
        self._f32_state = _is_float32_state(lon, lat)

    def sample(self, names, t, profile=None, nzp=0):
        first = not getattr(self, '_sampled_once', False) and self._f32_state      # (see OracleBackend.sample)
        self._sampled_once = True
        self.ctx.set_position_class(first)
        try:
            self.P.env_sample(names, t)
        finally:
            self.ctx.set_position_class(False)

    def truncate(self, depth):
        self.P.truncate_z(depth)

    def restore(self):
        self.P.restore_z()

    def report_missing(self, names, code):
        self.P.deactivate_missing(names, code)

    def coast(self, action, code=1, seeded_code=0):
        self.P.coastline(action, stranded_code=code, seeded_on_land_code=seeded_code)

    def sample_landmask(self, mask):
        pass      # the raster is a source of the device context (scenario_c10): sampled with the other variables

    def coast_crossing(self, action, precision, mask, code=1):
        self.P.coastline_crossing(action, precision, self.ctx.landmask_sid, stranded_code=code)

    def increase_age(self, dt):
        self.P.increase_age(dt)

    def seafloor(self, action='lift_to_seafloor', code=1):
        self.P.seafloor(action, code)

    def compact(self):
        self.P.compact()

    def store_previous(self):
        self.P.store_previous()

    def advect(self, scheme, t, dt, stage_noise=None, stds=None, ice=False):
        if stage_noise is not None:
            self.P.set_advect_noise(stds[0], stds[1], stage_draws=np.asarray(stage_noise)[..., :len(self.P)])
        self.P.set_element_factor('ice_current' if ice else None)
        self.P.advect(scheme, t, dt)
        self.P.set_element_factor(None)

    def vadvect(self, dt):
        self.P.vertical_advection(dt)

    def wind(self, dt, wdd=0.1, relative=False, ice=False):
        self.P.set_element_factor('ice_current' if ice else None)
        self.P.advect_wind(dt, wind_drift_depth=wdd, relative_wind=relative)
        self.P.set_element_factor(None)

    def stokes(self, dt, profile=2, hs_mode=1, tp_mode=1, ice=False):
        self.P.set_element_factor('ice_stokes' if ice else None)
        self.P.stokes_drift(dt, profile=profile, hs_mode=hs_mode, tp_mode=tp_mode)
        self.P.set_element_factor(None)

    def ice_drift(self, dt):
        self.P.set_element_factor('ice_drift')
        self.P.advect_sea_ice(dt)
        self.P.set_element_factor(None)

    def hdiff(self, dt, normals):
        n = len(self.P)
        self.P.hdiffusion(dt, normals=(normals[0][:n], normals[1][:n]))

    def set_leeway(self, props):
        for slot, k in enumerate(LEEWAY_PROPS):
            self.P.set_property(slot, np.asarray(props[k], dtype=np.float32))

    def noise(self, vx, vy, nx, ny, uniform=False):
        n = len(self.P)
        self.P.env_add_noise(vx, vy, 1.0, normals=(nx[:n], ny[:n]), uniform=uniform)

    def leeway(self, dt, uniforms, frac=0.4, cap_uniforms=None, thr=30.0, sig=5.0):
        if cap_uniforms is not None:
            self.P.leeway_capsize(dt, thr, sig, uniforms=cap_uniforms)
        self.P.leeway(dt, frac, uniforms=uniforms[:len(self.P)])

    def vmix(self, t, dt, dt_mix, zlevels, uniforms, vadv=True, levels=0):
        self.P.vmix(t, dt, dt_mix, uniforms=uniforms, fuse_vertical_advection=False if vadv else None, profile_levels=levels)

    def vbuoy(self, dt, action='lift_to_seafloor', code=1):
        self.ctx.set_seafloor_action(action, code)
        self.P.vertical_buoyancy(dt)
        self.ctx.set_seafloor_action('lift_to_seafloor')

    def vmix_analytic(self, model, background, dt, dt_mix, uniforms):
        self.P.vmix_analytic(model, background, dt, dt_mix, uniforms=uniforms)

    def set_oil(self, diameter, density, viscosity, film):
        n = len(self.P)
        for slot, v in enumerate((diameter, density, viscosity, film)):
            self.P.set_property(slot, np.asarray(v, np.float32) * np.ones(n, np.float32))

    def vmix_oil(self, model, background, dt, dt_mix, interfacial_tension, distribution, uniforms, t=0.0, zlevels=None):
        self.P.vmix_oil(model, background, dt, dt_mix, interfacial_tension, distribution, uniforms=uniforms, t_epoch=t)

    def oil_state(self):
        return dict(diameter=self.P.get_property(0), diameter_if_entrained=self.P.get_property(4),
                    terminal_velocity=self.P.download_f32('terminal_velocity'))

    def state(self, n_total):
        a, d = self.P.download(), self.P.download_deactivated()
        lon, lat, z = np.full(n_total, np.nan), np.full(n_total, np.nan), np.full(n_total, np.nan)
        status = np.full(n_total, -1, np.int32)
        for s in (a, d):
            lon[s['ID']], lat[s['ID']], z[s['ID']], status[s['ID']] = s['lon'], s['lat'], s['z'], s['status']
        return lon, lat, z, status


def replay_c3(B, g, nsteps):
    """C3-shaped golden: RK4 + vertical mixing + vertical advection, coastline 'previous'."""
    dt, dt_mix = float(g['dt']), float(g['dt_mix'])
    n = g['lon'].shape[1]
    out = []
    names = [U, VV, W, DEPTH, SSH, LAND]
    for k in range(nsteps):
        t = k * dt
        B.sample(names, t, profile=KZ, nzp=len(g['g_z']))
        B.coast('previous', seeded_code=1)   # status_categories: ['active', 'seeded_on_land']
        B.seafloor()
        B.increase_age(dt)
        B.compact()
        B.store_previous()
        B.advect('runge-kutta4', t, dt)
        B.vmix(t, dt, dt_mix, g['g_z'], g['uniforms'][k])
        out.append(B.state(n))
    return out


def scenario_c24(g, tag):
    """c24a: the C3-shaped fields; c24b: the same with ocean_vertical_diffusivity as a list of three members (oracle/gen_golden_profiles.py)"""
    from scenarios import Scenario
    names = [U, VV, W, KZ, DEPTH, LAND]
    t = g[tag + '_g_t']
    levels = []
    for k in range(len(t)):
        arrays = {nm: g['%s_g_%s' % (tag, nm)][k] for nm in names}
        if tag == 'b':
            arrays[KZ] = [g['b_g_K%d' % m][k] for m in range(int(g['members']))]
        levels.append((float(t[k]), arrays))
    return Scenario([('grid', dict(x=g[tag + '_g_x'], y=g[tag + '_g_y'], z=g[tag + '_g_z'], levels=levels))],
                    fallbacks={U: 0.0, VV: 0.0, W: 0.0, KZ: 0.0, DEPTH: 10000.0, SSH: 0.0})


def cf_reader_levels(zlev, depth, verticalbuffer=1):
    """Levels a reader that hands out the levels asked for returns for a request reaching `depth` from the surface
    (reader_netCDF_CF_generic.py:414-423: the span of the request, one more level, plus verticalbuffer; descending z)."""
    d = -np.asarray(zlev, dtype=np.float64)
    return int(min(len(d), np.searchsorted(d, depth) + 1 + verticalbuffer))


def replay_c24(B, g, tag, nsteps, truncate=None, gtag=None, cut_levels=False):
    """RK4 + vertical mixing on reader diffusivity profiles + vertical advection, coastline 'previous'; truncate: every
    sampling call -- the main one and the Runge-Kutta stage calls -- sees max(z, -truncate) (environment.py:554-566).
    cut_levels (golden c24c): the reader cut its block at the depth asked of it -- min(profiles_depth, truncate) and the
    deepest truncated element -- and the mixing runs on that column (elements below: K and dK/dz of its last level)."""
    dt, dt_mix = float(g['dt']), float(g['dt_mix'])
    n = g[tag + '_lon'].shape[1]
    zlev = g[(gtag or tag) + '_g_z']
    levels = 0
    if cut_levels:
        # some element is below the truncation depth in every step of the golden: the request reaches `truncate`
        levels = cf_reader_levels(zlev, truncate, int(g['c_verticalbuffer']))
        assert levels in g['c_levels_handed_out']
    out = []
    names = [U, VV, W, DEPTH, SSH, LAND]
    for k in range(nsteps):
        t = k * dt
        if truncate is not None:
            B.truncate(truncate)
        B.sample(names, t, profile=KZ, nzp=len(zlev))
        if truncate is not None:
            B.restore()
        B.coast('previous', seeded_code=1)
        B.seafloor()
        B.increase_age(dt)
        B.compact()
        B.store_previous()
        if truncate is not None:
            B.truncate(truncate)
        B.advect('runge-kutta4', t, dt)
        if truncate is not None:
            B.restore()
        B.vmix(t, dt, dt_mix, zlev, g[tag + '_uniforms'][k], levels=levels)
        out.append(B.state(n))
    return out


def replay_c4(B, g, nsteps):
    """C4-shaped golden: stere grid, RK4 + wind + Stokes + horizontal diffusion + stranding."""
    dt = float(g['dt'])
    n = g['lon'].shape[1]
    out = []
    names = [U, VV, XW, YW, SX, SY, LAND, HD, HS]
    for k in range(nsteps):
        t = k * dt
        B.sample(names, t)
        B.report_missing(names, 2)           # land_binary_mask has no fallback: NaN outside the reader's coverage
        B.coast('stranding', code=1)         # status_categories: ['active', 'stranded', 'missing_data']
        B.increase_age(dt)
        B.compact()
        B.store_previous()
        B.advect('runge-kutta4', t, dt)
        B.wind(dt, wdd=0.1)
        B.stokes(dt, profile=2, hs_mode=1, tp_mode=1)
        B.hdiff(dt, g['normals'][k])
        out.append(B.state(n))
    return out


def replay_c5(B, g, nsteps):
    """C5-shaped golden: Leeway on a stere grid, wind / current uncertainty, stranding, jibing."""
    dt = float(g['dt'])
    n = g['lon'].shape[1]
    B.set_leeway({k: g['p_' + k] for k in LEEWAY_PROPS})
    out = []
    names = [XW, YW, U, VV, LAND]
    for k in range(nsteps):
        t = k * dt
        B.sample(names, t)
        B.noise(U, VV, g['normals'][k][0], g['normals'][k][1])
        B.noise(XW, YW, g['normals'][k][2], g['normals'][k][3])
        B.coast('stranding', code=1)
        B.increase_age(dt)
        B.compact()
        B.store_previous()
        if 'cap_uniforms' in g:    # processes:capsizing golden (c5b)
            B.leeway(dt, g['uniforms'][k], cap_uniforms=g['cap_uniforms'][k], thr=float(g['wind_threshold']),
                     sig=float(g['wind_threshold_sigma']))
        else:
            B.leeway(dt, g['uniforms'][k])
        out.append(B.state(n))
    return out


def replay_c13(B, g, tag, nsteps):
    """c13 golden: OceanDrift, 3-D lon/lat grid + constant wind, 'runge-kutta' (tag 'rk2') / 'runge-kutta4' ('rk4') with
    drift:current_uncertainty (+ _uniform for rk4) and drift:wind_uncertainty: noise in the main sample AND in every
    Runge-Kutta stage call (environment.py:869-891, physics_methods.py:638-670)."""
    dt = float(g['dt'])
    n = g[tag + '_lon'].shape[1]
    scheme = {'rk2': 'runge-kutta', 'rk4': 'runge-kutta4'}[tag]
    main, stage = g[tag + '_main_noise'], g[tag + '_stage_noise']
    ncomp = stage.shape[2]
    stds = (float(g['current_uncertainty']), float(g['current_uncertainty_uniform']) if ncomp == 4 else 0.0)
    out = []
    names = [U, VV, W, XW, YW, DEPTH, SSH, LAND]
    for k in range(nsteps):
        t = k * dt
        B.sample(names, t)
        B.noise(U, VV, main[k][0], main[k][1])
        if ncomp == 4:
            B.noise(U, VV, main[k][2], main[k][3], uniform=True)
        B.noise(XW, YW, main[k][ncomp], main[k][ncomp + 1])
        B.coast('previous', seeded_code=1)   # status_categories: ['active', 'seeded_on_land']
        B.seafloor()
        B.increase_age(dt)
        B.compact()
        B.store_previous()
        B.advect(scheme, t, dt, stage_noise=stage[k], stds=stds)
        B.wind(dt, wdd=0.1)
        B.vbuoy(dt)
        B.vadvect(dt)
        out.append(B.state(n))
    return out


def scenario_c13(g):
    from scenarios import Scenario
    names = [U, VV, W, DEPTH, LAND]
    levels = [(float(g['g_t'][k]), {nm: g['g_' + nm][k] for nm in names}) for k in range(len(g['g_t']))]
    return Scenario([('grid', dict(x=g['g_x'], y=g['g_y'], z=g['g_z'], levels=levels)),
                     ('constant', {XW: float(g['wind'][0]), YW: float(g['wind'][1])})],
                    fallbacks={U: 0.0, VV: 0.0, W: 0.0, XW: 0.0, YW: 0.0, DEPTH: 10000.0, SSH: 0.0})


def replay_c14(B, g, nsteps, start=0):
    """c14 golden: the reference's OpenOil at its DEFAULT uncertainties (current 0.05, wind 0.5; openoil.py:497-498) with
    'runge-kutta4': OpenOil.update = oil_weathering (Kelvin) -> vertical mixing with the oil physics -> advect_oil =
    RK4 current (noise in the three stage calls) + windage."""
    dt, dt_mix = float(g['dt']), float(g['dt_mix'])
    n = g['lon'].shape[1]
    out = []
    names = [U, VV, XW, YW, MLD, DEPTH, SSH, LAND, TEMP, SALT]
    for k in range(start, nsteps):
        t = k * dt
        B.sample(names, t)
        B.noise(U, VV, g['main_noise'][k][0], g['main_noise'][k][1])
        B.noise(XW, YW, g['main_noise'][k][2], g['main_noise'][k][3])
        B.seafloor()
        B.increase_age(dt)
        B.store_previous()
        uni = dict(mix=g['u_mix'][k], entrain=g['u_entrain'][k], intrusion=np.nan_to_num(g['u_intrusion'][k], nan=0.5),
                   diameter=g['u_diameter'][k])
        B.vmix_oil('windspeed_Large1994', float(g['background_diffusivity']), dt, dt_mix, float(g['interfacial_tension']),
                   'Johansen et al. (2015)', uni)
        B.advect('runge-kutta4', t, dt, stage_noise=g['stage_noise'][k], stds=(0.05, 0.0))
        B.wind(dt, wdd=0.1)
        out.append(B.state(n) + (B.oil_state(),))
    return out


def replay_c16(B, g, nsteps):
    """c16 golden: the reference's OpenOil in sea ice (openoil.py:1179-1216): RK4 current and windage scaled by 1 - k_ice,
    Stokes drift by (0.7 - A) / 0.7, drift with the ice velocity scaled by k_ice; no mixing, no uncertainties."""
    dt = float(g['dt'])
    n = g['lon'].shape[1]
    out = []
    names = [U, VV, XW, YW, SX, SY, LAND, ICE_A, ICE_U, ICE_V]
    for k in range(nsteps):
        t = k * dt
        B.sample(names, t)
        B.increase_age(dt)
        B.store_previous()
        B.advect('runge-kutta4', t, dt, ice=True)
        B.wind(dt, wdd=float(g['wind_drift_depth']), ice=True)
        B.stokes(dt, profile=2, hs_mode=1, tp_mode=3, ice=True)
        B.ice_drift(dt)
        out.append(B.state(n))
    return out


def scenario_c16(g):
    from scenarios import Scenario
    from opendrift_amd import synthetic as synth
    names = [U, VV, XW, YW, SX, SY, LAND, ICE_A, ICE_U, ICE_V]
    levels = [(float(t), {k: g['g_' + k][i] for k in names}) for i, t in enumerate(g['g_t'])]
    return Scenario([('grid', dict(x=g['g_x'], y=g['g_y'], levels=levels, proj=synth.NORKYST_PROJ))],
                    fallbacks={k: 0.0 for k in names + [HS]})


def replay_c17(B, g, tag, nsteps):
    """c17 golden: a reader whose current comes as a list of ensemble members -- element j of a call takes member j % M
    (readers/interpolation/structured.py:119-135); RK4 (the stage calls map the same way), stranding (the ranks shift)."""
    dt = float(g['dt'])
    n = g[tag + '_lon'].shape[1]
    out = []
    names = [U, VV, LAND]
    for k in range(nsteps):
        t = k * dt
        B.sample(names, t)
        B.coast('stranding', code=1)
        B.increase_age(dt)
        B.compact()
        B.store_previous()
        B.advect('runge-kutta4', t, dt)
        out.append(B.state(n))
    return out


def scenario_c17(g, tag):
    from scenarios import Scenario
    M = int(g['members'])
    q = lambda k: g['%s_g_%s' % (tag, k)]
    levels = [(float(t), {U: [q('u%d' % m)[i] for m in range(M)], VV: [q('v%d' % m)[i] for m in range(M)],
                          LAND: q('land_binary_mask')[i]}) for i, t in enumerate(q('t'))]
    z = g[tag + '_g_z'] if (tag + '_g_z') in g else None
    # 'partial': elements start west of the reader's domain; the fallback current of the golden's run carries them in
    fb = {U: 1.5, VV: 0.1, LAND: 0.0} if tag == 'partial' else {U: 0.0, VV: 0.0}
    return Scenario([('grid', dict(x=q('x'), y=q('y'), z=z, levels=levels))], fallbacks=fb)


def replay_c7(B, g, sub, model, background, nsteps, start=0):
    """c7 golden: Euler current + vertical mixing with a wind-parameterised diffusivity profile.  `start`: first
    step to replay (the back end then holds golden row `start`)."""
    dt, dt_mix = float(g['dt']), float(g['dt_mix'])
    n = sub['lon'].shape[1]
    out = []
    names = [U, VV, XW, YW, MLD, DEPTH, SSH, LAND]
    for k in range(start, nsteps):
        t = k * dt
        B.sample(names, t)
        B.seafloor()
        B.increase_age(dt)
        B.store_previous()
        B.advect('euler', t, dt)
        B.vmix_analytic(model, background, dt, dt_mix, sub['uniforms'][k])
        out.append(B.state(n))
    return out


def replay_c9(B, g, tag, nsteps, distribution, start=0):
    """c9 golden: the reference's OpenOil with a constant-property oil: per step oil_weathering (Kelvin), vertical
    mixing with terminal velocities / slick / wave entrainment, then advect_oil (Euler current; no windage, no Stokes
    drift) -- OpenOil.update, openoil.py:1218-1239.  Returns [(lon, lat, z, status, oil_state)] per step."""
    dt, dt_mix = float(g['dt']), float(g['dt_mix'])
    n = g[tag + '_lon'].shape[1]
    out = []
    names = [U, VV, XW, YW, MLD, DEPTH, SSH, LAND, TEMP, SALT]
    for k in range(start, nsteps):
        t = k * dt
        B.sample(names, t)
        B.seafloor()
        B.increase_age(dt)
        B.store_previous()
        uni = dict(mix=g[tag + '_u_mix'][k], entrain=g[tag + '_u_entrain'][k],
                   intrusion=np.nan_to_num(g[tag + '_u_intrusion'][k], nan=0.5), diameter=g[tag + '_u_diameter'][k])
        B.vmix_oil('windspeed_Large1994', float(g['background_diffusivity']), dt, dt_mix, float(g['interfacial_tension']), distribution, uni)
        B.advect('euler', t, dt)
        out.append(B.state(n) + (B.oil_state(),))
    return out


def scenario_c9(g):
    from scenarios import Scenario
    names = [U, VV, XW, YW, MLD, DEPTH, TEMP, SALT]
    levels = [(float(g['g_t'][k]), {nm: g['g_' + nm][k] for nm in names}) for k in range(len(g['g_t']))]
    return Scenario([('grid', dict(x=g['g_x'], y=g['g_y'], levels=levels))],
                    fallbacks={U: 0.0, VV: 0.0, XW: 0.0, YW: 0.0, MLD: 50.0, DEPTH: 10000.0, SSH: 0.0, LAND: 0.0,
                               TEMP: 10.0, SALT: 34.0})


def replay_c10(B, g, action, nsteps, mask):
    """c10 golden: constant current towards a raster coast, general:coastline_approximation_precision set."""
    dt, prec = float(g['dt']), float(g['precision'])
    n = g[action + '_lon'].shape[1]
    out = []
    for k in range(nsteps):
        t = k * dt
        B.sample([U, VV, LAND], t)
        B.sample_landmask(mask)
        B.coast_crossing(action, prec, mask)
        B.increase_age(dt)
        B.compact()
        B.store_previous()
        B.advect('euler', t, dt)
        out.append(B.state(n))
    return out


def scenario_c10(g, device_raster=None):
    """constant current; the landmask raster is a device source (device_raster = RasterMask) or is sampled by the
    oracle back end in NumPy (fallback 0 here)"""
    from scenarios import Scenario
    src = [('constant', {U: float(g['u']), VV: float(g['v'])})]
    if device_raster is not None:
        m = device_raster
        src.append(('landmask', dict(lon0=m.lon0, lat0=m.lat0, dlon=m.dlon, dlat=m.dlat, cells=m.cells)))
        return Scenario(src)
    return Scenario(src, fallbacks={LAND: 0.0})


def replay_c8(B, g, sub, action, nsteps):
    """c8 golden: Euler current into shoaling water, general:seafloor_action 'deactivate' / 'previous'."""
    dt = float(g['dt'])
    n = sub['lon'].shape[1]
    out = []
    names = [U, VV, DEPTH, SSH, LAND]
    for k in range(nsteps):
        t = k * dt
        B.sample(names, t)
        B.seafloor(action, code=1)
        B.increase_age(dt)
        B.compact()
        B.store_previous()
        B.advect('euler', t, dt)
        B.vbuoy(dt, action, code=1)      # OceanDrift.update without mixing: vertical_buoyancy checks the sea floor again
        out.append(B.state(n))
    return out


def scenario_c8(g):
    from scenarios import Scenario
    names = [U, VV, DEPTH]
    levels = [(float(g['g_t'][k]), {nm: g['g_' + nm][k] for nm in names}) for k in range(len(g['g_t']))]
    return Scenario([('grid', dict(x=g['g_x'], y=g['g_y'], levels=levels))],
                    fallbacks={U: 0.0, VV: 0.0, DEPTH: 10000.0, SSH: 0.0, LAND: 0.0})


def scenario_c7(g):
    from scenarios import Scenario
    names = [U, VV, XW, YW, MLD, DEPTH]
    levels = [(float(g['g_t'][k]), {nm: g['g_' + nm][k] for nm in names}) for k in range(len(g['g_t']))]
    return Scenario([('grid', dict(x=g['g_x'], y=g['g_y'], levels=levels))],
                    fallbacks={U: 0.0, VV: 0.0, XW: 0.0, YW: 0.0, MLD: 50.0, DEPTH: 10000.0, SSH: 0.0, LAND: 0.0})


def scenario_c5(g):
    from scenarios import Scenario
    from opendrift_amd import synthetic as synth
    names = [U, VV, XW, YW, LAND]
    levels = [(float(g['g_t'][k]), {nm: g['g_' + nm][k] for nm in names}) for k in range(len(g['g_t']))]
    return Scenario([('grid', dict(x=g['g_x'], y=g['g_y'], proj=synth.NORKYST_PROJ, levels=levels,
                                   time_coverage=(float(g['g_t'][0]), float(g['g_t'][-1]))))])


def scenario_c3(g):
    from scenarios import Scenario
    names = [U, VV, W, KZ, DEPTH, LAND]
    levels = [(float(g['g_t'][k]), {nm: g['g_' + nm][k] for nm in names}) for k in range(len(g['g_t']))]
    return Scenario([('grid', dict(x=g['g_x'], y=g['g_y'], z=g['g_z'], levels=levels))],
                    fallbacks={U: 0.0, VV: 0.0, W: 0.0, KZ: 0.0, DEPTH: 10000.0, SSH: 0.0})


def scenario_c4(g):
    from scenarios import Scenario
    from opendrift_amd import synthetic as synth
    names = [U, VV, XW, YW, SX, SY, LAND]
    levels = [(float(g['g_t'][k]), {nm: g['g_' + nm][k] for nm in names}) for k in range(len(g['g_t']))]
    return Scenario([('grid', dict(x=g['g_x'], y=g['g_y'], proj=synth.NORKYST_PROJ, levels=levels,
                                   time_coverage=(float(g['g_t'][0]), float(g['g_t'][-1])))),
                     ('constant', {HD: 10.0})],
                    fallbacks={U: 0.0, VV: 0.0, XW: 0.0, YW: 0.0, SX: 0.0, SY: 0.0, HD: 0.0, HS: 0.0})


def scenario_c20(g, tag):
    """c20: the C4-shaped fields on a Lambert conformal conic / Mercator grid (oracle/gen_golden_proj.py)"""
    from scenarios import Scenario
    from opendrift_amd import projection
    names = [U, VV, XW, YW, SX, SY, LAND]
    t = g[tag + '_g_t']
    levels = [(float(t[k]), {nm: g['%s_g_%s' % (tag, nm)][k] for nm in names}) for k in range(len(t))]
    return Scenario([('grid', dict(x=g[tag + '_g_x'], y=g[tag + '_g_y'], proj=projection.parse_proj4(str(g[tag + '_proj4'])),
                                   levels=levels, time_coverage=(float(t[0]), float(t[-1]))))],
                    fallbacks={U: 0.0, VV: 0.0, XW: 0.0, YW: 0.0, SX: 0.0, SY: 0.0, HS: 0.0})


def scenario_c23(g, tag):
    """c23: the same fields on a UTM / LAEA / oblique-stereographic grid and on a rotated pole (oracle/gen_golden_proj2.py).
    The rotated-pole domain lies at positive true longitudes: modulate_longitude (variables.py:259-280) takes np.mod(lon, 360)."""
    sc = scenario_c20(g, tag)
    if tag == 'rotated_pole':
        sc.sources[0][1]['lon_mode'] = 2
    return sc


def replay_c20(B, g, tag, nsteps):
    """RK4 + wind + Stokes drift + stranding, no random terms"""
    dt = float(g['dt'])
    out = []
    names = [U, VV, XW, YW, SX, SY, LAND, HS]
    n = g[tag + '_lon'].shape[1]
    for k in range(nsteps):
        t = k * dt
        B.sample(names, t)
        B.coast('stranding', code=1)
        B.increase_age(dt)
        B.compact()
        B.store_previous()
        B.advect('runge-kutta4', t, dt)
        B.wind(dt, wdd=0.1)
        B.stokes(dt, profile=2, hs_mode=1, tp_mode=1)
        out.append(B.state(n))
    return out


def compare(states, g, tol_pos, tol_z=None):
    worst = dict(lon=0.0, lat=0.0, z=0.0)
    for k, (lon, lat, z, status) in enumerate(states):
        ref = {q: g[q][k + 1] for q in ('lon', 'lat', 'z', 'status')}
        assert (np.isnan(lon) == np.isnan(ref['lon'])).all()
        assert ((status != 0) == (ref['status'] != 0)).all(), 'step %d: deactivation sets differ' % k
        for q, a in (('lon', lon), ('lat', lat), ('z', z)):
            worst[q] = max(worst[q], float(np.nanmax(np.abs(a - ref[q]))))
    assert worst['lon'] < tol_pos and worst['lat'] < tol_pos, worst
    if tol_z is not None:
        assert worst['z'] < tol_z, worst
    return worst
