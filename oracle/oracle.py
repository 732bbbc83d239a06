"""TEST INFRASTRUCTURE ONLY -- ctypes front end of the CPU oracle (oracle/*.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (opendrift_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, '_build', 'liboracle.so')

NVAR = 26
MAXLEVELS = 4
VAR = dict(x_sea_water_velocity=0, y_sea_water_velocity=1, x_wind=2, y_wind=3,
           upward_sea_water_velocity=4, ocean_vertical_diffusivity=5,
           sea_surface_wave_stokes_drift_x_velocity=6,
           sea_surface_wave_stokes_drift_y_velocity=7, land_binary_mask=8,
           sea_floor_depth_below_sea_level=9, sea_surface_height=10,
           horizontal_diffusivity=11, sea_surface_wave_significant_height=12,
           sea_surface_wave_period_at_variance_spectral_density_maximum=13,
           ocean_mixed_layer_thickness=14, sea_water_temperature=15, sea_water_salinity=16,
           sea_ice_area_fraction=17, sea_ice_x_velocity=18, sea_ice_y_velocity=19,
           sea_surface_swell_wave_to_direction=20,
           sea_surface_swell_wave_peak_period_from_variance_spectral_density=21,
           sea_surface_swell_wave_significant_height=22, sea_surface_wind_wave_to_direction=23,
           sea_surface_wind_wave_mean_period=24, sea_surface_wind_wave_significant_height=25)
PROJ_LATLONG, PROJ_STERE_EQUIT_SPHERE, PROJ_STERE_POLAR = 0, 1, 2
PROJ_MERC, PROJ_LCC = 4, 5
PROJ_TMERC, PROJ_LAEA, PROJ_STERE_OBLIQUE, PROJ_OB_TRAN = 6, 7, 8, 9
SRC_CONSTANT, SRC_DOUBLE_GYRE, SRC_OSCILLATING, SRC_GRID = 0, 1, 2, 3


def build(force=False):
    """Compile oracle/*.c with gcc (make).  Building the checker is not using it."""
    srcs = [os.path.join(HERE, f) for f in ('geodesic.c', 'proj.c', 'interp.c', 'step.c',
                                            'oracle.h', 'geodesic.h')]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    subprocess.check_call(['make', '-C', HERE, '-B', '-s'])
    return LIB_PATH


class Proj(C.Structure):
    _fields_ = [('kind', C.c_int), ('south', C.c_int)] + \
        [(k, C.c_double) for k in ('a', 'es', 'e', 'lon0', 'lat0', 'x0', 'y0', 'k0', 'akm1', 'n', 'c', 'rho0')] + \
        [('mode', C.c_int), ('pad', C.c_int), ('q', C.c_double * 16)]


class Block(C.Structure):
    _fields_ = [('nz', C.c_int), ('ny', C.c_int), ('nx', C.c_int),
                ('x0', C.c_double), ('xspan', C.c_double), ('y0', C.c_double), ('yspan', C.c_double),
                ('xmin', C.c_double), ('xrange', C.c_double), ('ymin', C.c_double), ('yrange', C.c_double),
                ('z', C.POINTER(C.c_double)), ('t', C.c_double),
                ('data', C.POINTER(C.c_float) * NVAR), ('var_nz', C.c_int * NVAR), ('members', C.c_int * NVAR)]


class Source(C.Structure):
    _fields_ = [('kind', C.c_int), ('proj', Proj),
                ('xmin', C.c_double), ('xmax', C.c_double), ('ymin', C.c_double), ('ymax', C.c_double),
                ('zmin', C.c_double), ('zmax', C.c_double),
                ('lon_mode', C.c_int), ('mod360_x', C.c_int), ('has_var', C.c_int * NVAR),
                ('const_val', C.c_double * NVAR), ('params', C.c_double * 8),
                ('nlevels', C.c_int), ('level', Block * MAXLEVELS), ('always_valid', C.c_int),
                ('tmin', C.c_double), ('tmax', C.c_double), ('xy_f32', C.c_int)]


class World(C.Structure):
    _fields_ = [('nsrc', C.c_int), ('src', C.POINTER(Source)),
                ('nlist', C.c_int * NVAR), ('list', (C.c_int * 4) * NVAR),
                ('fallback', C.c_float * NVAR)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def geod_fwd(lon, lat, az, dist):
    """pyproj.Geod(ellps='WGS84').fwd stand-in: returns lon2, lat2, forward azimuth at 2."""
    lon, lat, az, dist = np.broadcast_arrays(*[np.atleast_1d(_d(v)) for v in (lon, lat, az, dist)])
    lon, lat, az, dist = _d(lon), _d(lat), _d(az), _d(dist)
    n = lon.size
    o1, o2, o3 = np.empty(n), np.empty(n), np.empty(n)
    lib().orc_wgs84_direct_n(C.c_long(n), _p(lon, C.c_double), _p(lat, C.c_double), _p(az, C.c_double),
                             _p(dist, C.c_double), _p(o1, C.c_double), _p(o2, C.c_double), _p(o3, C.c_double))
    return o1, o2, o3


def geod_inv(lon1, lat1, lon2, lat2):
    """Geod.inv stand-in: forward azimuth at point 1 and distance (back azimuth not computed)."""
    a = [np.atleast_1d(_d(v)) for v in (lon1, lat1, lon2, lat2)]
    a = [_d(v) for v in np.broadcast_arrays(*a)]
    n = a[0].size
    az, s = np.empty(n), np.empty(n)
    lib().orc_wgs84_inverse_n(C.c_long(n), *[_p(v, C.c_double) for v in a], _p(az, C.c_double),
                              None, _p(s, C.c_double))
    return az, s


def make_proj(kind=PROJ_LATLONG, a=6378137.0, es=0.0, lat0=0.0, lon0=0.0, lat_ts=90.0, k0=1.0,
              x0=0.0, y0=0.0, lat1=0.0, lat2=None):
    p = Proj()
    if kind in (PROJ_TMERC, PROJ_LAEA, PROJ_STERE_OBLIQUE, PROJ_OB_TRAN):   # ob_tran: lat1 = o_lat_p, lat2 = o_lon_p
        lib().orc_proj_init_ext(C.byref(p), C.c_int(kind), C.c_double(a), C.c_double(es), C.c_double(lat0), C.c_double(lon0),
                                C.c_double(k0), C.c_double(x0), C.c_double(y0), C.c_double(lat1),
                                C.c_double(0.0 if lat2 is None else lat2))
        return p
    if kind in (PROJ_MERC, PROJ_LCC):
        lib().orc_proj_init_conic(C.byref(p), C.c_int(kind), C.c_double(a), C.c_double(es), C.c_double(lat0),
                                  C.c_double(lon0), C.c_double(lat_ts), C.c_double(k0), C.c_double(x0), C.c_double(y0),
                                  C.c_double(lat1), C.c_double(lat1 if lat2 is None else lat2))
        return p
    lib().orc_proj_init(C.byref(p), C.c_int(kind), C.c_double(a), C.c_double(es), C.c_double(lat0),
                        C.c_double(lon0), C.c_double(lat_ts), C.c_double(k0), C.c_double(x0),
                        C.c_double(y0))
    return p


def proj_fwd(p, lon, lat):
    lon, lat = np.atleast_1d(_d(lon)), np.atleast_1d(_d(lat))
    x, y = np.empty_like(lon), np.empty_like(lon)
    cx, cy = C.c_double(), C.c_double()
    f = lib().orc_proj_fwd
    for i in range(lon.size):
        f(C.byref(p), C.c_double(lon[i]), C.c_double(lat[i]), C.byref(cx), C.byref(cy))
        x[i], y[i] = cx.value, cy.value
    return x, y


def proj_inv(p, x, y):
    x, y = np.atleast_1d(_d(x)), np.atleast_1d(_d(y))
    lon, lat = np.empty_like(x), np.empty_like(x)
    cx, cy = C.c_double(), C.c_double()
    f = lib().orc_proj_inv
    for i in range(x.size):
        f(C.byref(p), C.c_double(x[i]), C.c_double(y[i]), C.byref(cx), C.byref(cy))
        lon[i], lat[i] = cx.value, cy.value
    return lon, lat


def linear2d_call(array2d, yi, xi):
    """Linear2DInterpolator.__call__ on fractional indices; MUTATES array2d (float32) in place."""
    assert array2d.dtype == np.float32 and array2d.flags.c_contiguous
    yi, xi = _d(yi), _d(xi)
    out = np.empty(yi.size, np.float32)
    lib().orc_linear2d_call(_p(array2d, C.c_float), C.c_int(array2d.shape[0]), C.c_int(array2d.shape[1]),
                            C.c_long(yi.size), _p(yi, C.c_double), _p(xi, C.c_double), _p(out, C.c_float))
    return out


def dilate_nan_once(array2d):
    assert array2d.dtype == np.float32 and array2d.flags.c_contiguous
    lib().orc_dilate_nan_once(_p(array2d, C.c_float), C.c_int(array2d.shape[0]), C.c_int(array2d.shape[1]))


class WorldBuilder:
    """Assembles an orc_world: sources + per-variable priority lists + fallbacks."""

    def __init__(self):
        self.sources = []
        self.keep = []  # keep numpy buffers alive
        self.lists = {v: [] for v in range(NVAR)}
        self.fallback = np.full(NVAR, np.nan, np.float32)

    def _new(self, kind, proj, domain, lon_mode, variables):
        s = Source()
        s.kind = kind
        s.proj = proj if proj is not None else make_proj()
        s.xmin, s.xmax, s.ymin, s.ymax = domain[:4]
        s.zmin = domain[4] if len(domain) > 4 else -np.inf
        s.zmax = domain[5] if len(domain) > 5 else np.inf
        s.lon_mode = lon_mode
        s.tmin, s.tmax = -np.inf, np.inf
        for v in variables:
            s.has_var[v] = 1
        self.sources.append(s)
        idx = len(self.sources) - 1
        for v in variables:
            self.lists[v].append(idx)
        return idx, s

    def add_constant(self, values):
        """reader_constant.Reader({...}) / environment:constant:<var>"""
        idx, s = self._new(SRC_CONSTANT, None, (-180, 180, -90, 90), 1, list(values.keys()))
        for v, val in values.items():
            s.const_val[v] = float(val)
        return idx

    def add_double_gyre(self, A=0.25, epsilon=0.1, omega=0.628, t0=0.0):
        p = make_proj(PROJ_STERE_EQUIT_SPHERE, a=6.371e6, es=0.0, lat0=0.0, lon0=0.0, lat_ts=0.0)
        idx, s = self._new(SRC_DOUBLE_GYRE, p, (0., 2., 0., 1.), 2, [0, 1, VAR['land_binary_mask']])
        s.params[0], s.params[1], s.params[2], s.params[3] = A, epsilon, omega, t0
        return idx

    def add_oscillating(self, var, amplitude, period_s, t0):
        idx, s = self._new(SRC_OSCILLATING, None, (-180, 180, -90, 90), 1, [var])
        s.params[0], s.params[1], s.params[2], s.params[3] = var, amplitude, period_s, t0
        return idx

    def add_grid(self, proj, x, y, levels, z=None, lon_mode=1, mod360_x=0, time_coverage=None):
        """levels: list of (t_epoch, {var_id: float32 array [ny,nx] or [nz,ny,nx]}); x, y in the
        dtype the reader hands out (float32 for the file readers)."""
        variables = sorted(levels[0][1].keys())
        dom = (float(x.min()), float(x.max()), float(y.min()), float(y.max()))
        idx, s = self._new(SRC_GRID, proj, dom, lon_mode, variables)
        s.mod360_x = mod360_x
        s.xy_f32 = (1 if np.asarray(x).dtype == np.float32 else 0) | (2 if np.asarray(y).dtype == np.float32 else 0)
        if time_coverage is not None:
            s.tmin, s.tmax = float(time_coverage[0]), float(time_coverage[1])
        s.nlevels = len(levels)
        zz = None
        if z is not None and np.size(z) > 1:
            zz = _d(z)
            self.keep.append(zz)
        for k, (t, arrays) in enumerate(levels):
            b = s.level[k]
            b.ny, b.nx = len(y), len(x)
            b.nz = zz.size if zz is not None else 1
            # Linear2DInterpolator: (x - xgrid[0])/(xgrid[-1]-xgrid[0]) -- the span is formed in x's dtype
            b.x0, b.xspan = float(x[0]), float(x[-1] - x[0])
            b.y0, b.yspan = float(y[0]), float(y[-1] - y[0])
            b.xmin, b.xrange = float(x.min()), float(x.max() - x.min())
            b.ymin, b.yrange = float(y.min()), float(y.max() - y.min())
            b.t = float(t)
            if zz is not None:
                b.z = _p(zz, C.c_double)
            for v, arr in arrays.items():
                if isinstance(arr, (list, tuple)):      # ensemble members (structured.py:125-147): one after the other
                    b.members[v] = len(arr)
                    b.var_nz[v] = arr[0].shape[0] if np.ndim(arr[0]) == 3 else 1
                    arr = np.stack([np.asarray(a, dtype=np.float32) for a in arr])
                    arr = np.array(arr.reshape((-1,) + arr.shape[-2:]), dtype=np.float32, order='C', copy=True)
                else:
                    arr = np.array(arr, dtype=np.float32, order='C', copy=True)
                    b.var_nz[v] = arr.shape[0] if arr.ndim == 3 else 1
                self.keep.append(arr)
                b.data[v] = _p(arr, C.c_float)
        return idx

    def set_fallback(self, var, value):
        self.fallback[var] = value

    def set_priority(self, var, source_ids):
        self.lists[var] = list(source_ids)

    def finish(self):
        w = World()
        arr = (Source * len(self.sources))(*self.sources)
        self.keep.append(arr)
        w.nsrc = len(self.sources)
        w.src = C.cast(arr, C.POINTER(Source))
        for v in range(NVAR):
            w.nlist[v] = len(self.lists[v])
            for k, sid in enumerate(self.lists[v][:4]):
                w.list[v][k] = sid
            w.fallback[v] = self.fallback[v]
        self.world = w
        return w


def set_position_class(f32):
    """The get_environment / get_profile calls that follow treat the positions as the reference's float32 element arrays of the
    first step of a run (modulate_longitude in float32, variables.py:259-280 with elements.py:71-88); False ends it."""
    lib().orc_set_position_class(C.c_int(1 if f32 else 0))


def get_environment(world, variables, lon, lat, z, t):
    lon, lat, z = _d(lon), _d(lat), _d(np.broadcast_to(z, np.shape(lon)))
    n = lon.size
    vars_ = (C.c_int * len(variables))(*variables)
    outs = [np.empty(n, np.float32) for _ in variables]
    ptrs = (C.POINTER(C.c_float) * len(variables))(*[_p(o, C.c_float) for o in outs])
    lib().orc_get_environment(C.byref(world), C.c_int(len(variables)), vars_, C.c_long(n),
                              _p(lon, C.c_double), _p(lat, C.c_double), _p(z, C.c_double),
                              C.c_double(t), ptrs)
    return outs


def get_profile(world, var, lon, lat, t, nz_prof):
    lon, lat = _d(lon), _d(lat)
    out = np.empty((nz_prof, lon.size))
    lib().orc_get_profile(C.byref(world), C.c_int(var), C.c_long(lon.size), _p(lon, C.c_double),
                          _p(lat, C.c_double), C.c_double(t), C.c_int(nz_prof), _p(out, C.c_double))
    return out


def update_positions(lon, lat, u, v, moving, dt):
    """In place on float64 lon/lat; dispatches on the velocity dtype like NumPy does."""
    n = lon.size
    moving = _i(moving)
    if u.dtype == np.float32:
        lib().orc_update_positions_f32(C.c_long(n), _p(lon, C.c_double), _p(lat, C.c_double),
                                       _p(_f(u), C.c_float), _p(_f(v), C.c_float), _p(moving, C.c_int),
                                       C.c_double(dt))
    else:
        lib().orc_update_positions_f64(C.c_long(n), _p(lon, C.c_double), _p(lat, C.c_double),
                                       _p(_d(u), C.c_double), _p(_d(v), C.c_double), _p(moving, C.c_int),
                                       C.c_double(dt))


def advect_ocean_current(world, scheme, lon, lat, z, moving, cdf, u_env, v_env, t, dt, factor=1.0, stage_noise=None):
    """stage_noise: [nstage][ncomp][n] float64 -- the np.random draws of the Runge-Kutta stage get_environment calls
    (drift:current_uncertainty normal x, y; then drift:current_uncertainty_uniform x, y), or None"""
    n = lon.size
    z, moving, cdf = _d(np.broadcast_to(z, lon.shape)), _i(moving), _f(np.broadcast_to(cdf, lon.shape))
    if np.ndim(factor):      # per-element float32 factor: factor*cdf is float32 * float32 (physics_methods.py:622)
        cdf, factor = _f(np.asarray(factor, dtype=np.float32) * cdf), 1.0
    ncomp, sn = 0, None
    if stage_noise is not None and scheme > 0:
        sn = _d(np.asarray(stage_noise)[..., :n])
        assert sn.ndim == 3 and sn.shape[0] == (1 if scheme == 1 else 3) and sn.shape[1] in (2, 4), sn.shape
        ncomp = sn.shape[1]
    lib().orc_advect_ocean_current_noise(C.byref(world), C.c_int(scheme), C.c_long(n), _p(lon, C.c_double),
                                         _p(lat, C.c_double), _p(z, C.c_double), _p(moving, C.c_int),
                                         _p(cdf, C.c_float), _p(_f(u_env), C.c_float), _p(_f(v_env), C.c_float),
                                         C.c_double(t), C.c_double(dt), C.c_double(factor), C.c_int(ncomp),
                                         _p(sn, C.c_double) if sn is not None else None)


def ice_factors(A):
    """OpenOil.advect_oil (openoil.py:1182-1201) on the float32 sea_ice_area_fraction: (k_ice, factor_stokes), float32
    like NumPy computes them (float32 array with python scalars)."""
    A = np.asarray(A, dtype=np.float32)
    k_ice = (A - 0.3) / (0.8 - 0.3)
    k_ice[A < 0.3] = 0
    k_ice[A > 0.8] = 1
    factor_stokes = (0.7 - A) / 0.7
    factor_stokes[A > 0.7] = 0
    assert k_ice.dtype == np.float32 and factor_stokes.dtype == np.float32
    return k_ice, factor_stokes


def advect_wind(lon, lat, z, moving, wdf, xwind, ywind, u_env, v_env, wind_drift_depth, relative_wind,
                factor, dt):
    """factor: python scalar, or a float32 array (per element)"""
    n = lon.size
    ef = _f(factor) if np.ndim(factor) else None
    lib().orc_advect_wind_ef(C.c_long(n), _p(lon, C.c_double), _p(lat, C.c_double), _p(_d(z), C.c_double),
                             _p(_i(moving), C.c_int), _p(_f(wdf), C.c_float), _p(_f(xwind), C.c_float),
                             _p(_f(ywind), C.c_float), _p(_f(u_env), C.c_float), _p(_f(v_env), C.c_float),
                             C.c_double(wind_drift_depth), C.c_int(relative_wind),
                             C.c_double(1.0 if ef is not None else factor), _p(ef, C.c_float) if ef is not None else None,
                             C.c_double(dt))


def stokes_drift(lon, lat, z, moving, sx, sy, hs, tp, xwind, ywind, hs_mode, tp_mode, profile, factor, dt):
    n = lon.size
    ef = _f(factor) if np.ndim(factor) else None
    lib().orc_stokes_drift_ef(C.c_long(n), _p(lon, C.c_double), _p(lat, C.c_double), _p(_d(z), C.c_double),
                              _p(_i(moving), C.c_int), _p(_f(sx), C.c_float), _p(_f(sy), C.c_float),
                              _p(_f(hs), C.c_float), _p(_f(tp), C.c_float), _p(_f(xwind), C.c_float),
                              _p(_f(ywind), C.c_float), C.c_int(hs_mode), C.c_int(tp_mode), C.c_int(profile),
                              C.c_double(1.0 if ef is not None else factor),
                              _p(ef, C.c_float) if ef is not None else None, C.c_double(dt))


def stokes_windsea_swell(z, sx, sy, swell_dir, swell_tp, swell_hs, ww_dir, ww_tm, ww_hs):
    """stokes_drift_profile_windsea_swell (physics_methods.py:418-456): (stokes_u, stokes_v), float64"""
    n = np.size(sx)
    u, v = np.empty(n), np.empty(n)
    lib().orc_stokes_windsea_swell(C.c_long(n), _p(_d(z), C.c_double), *[_p(_f(a), C.c_float) for a in (
        sx, sy, swell_dir, swell_tp, swell_hs, ww_dir, ww_tm, ww_hs)], _p(u, C.c_double), _p(v, C.c_double))
    return u, v


def stokes_drift_windsea_swell(lon, lat, z, moving, sx, sy, swell_dir, swell_tp, swell_hs, ww_dir, ww_tm, ww_hs, factor, dt):
    n = lon.size
    lib().orc_stokes_drift_windsea_swell(C.c_long(n), _p(lon, C.c_double), _p(lat, C.c_double), _p(_d(z), C.c_double),
                                         _p(_i(moving), C.c_int), *[_p(_f(a), C.c_float) for a in (
                                             sx, sy, swell_dir, swell_tp, swell_hs, ww_dir, ww_tm, ww_hs)],
                                         C.c_double(factor), C.c_double(dt))


def horizontal_diffusion(lon, lat, moving, D, nx, ny, dt):
    n = lon.size
    lib().orc_horizontal_diffusion(C.c_long(n), _p(lon, C.c_double), _p(lat, C.c_double),
                                   _p(_i(moving), C.c_int), _p(_f(D), C.c_float), _p(_d(nx), C.c_double),
                                   _p(_d(ny), C.c_double), C.c_double(dt))


def vertical_mixing(z, moving, tv, depth, ssh, zp, Kprof, dt, dt_mix, mix_at_surface, uniforms):
    n = z.size
    zp, Kprof, uniforms = _d(zp), _d(Kprof), _d(uniforms)
    lib().orc_vertical_mixing(C.c_long(n), _p(z, C.c_double), _p(_i(moving), C.c_int), _p(_f(tv), C.c_float),
                              _p(_f(depth), C.c_float), _p(_f(ssh), C.c_float), C.c_int(zp.size),
                              _p(zp, C.c_double), _p(Kprof, C.c_double), C.c_double(dt), C.c_double(dt_mix),
                              C.c_int(mix_at_surface), _p(uniforms, C.c_double))


def vertical_advection(z, moving, w, dt, at_surface=0):
    lib().orc_vertical_advection(C.c_long(z.size), _p(z, C.c_double), _p(_i(moving), C.c_int),
                                 _p(_f(w), C.c_float), C.c_double(dt), C.c_int(at_surface))


def coastline(action, land, lon, lat, z, prev_lon, prev_lat, status, moving, stranded_code, age_seconds=None,
              seeded_on_land_code=0):
    age = _f(age_seconds) if age_seconds is not None else None
    lib().orc_coastline(C.c_long(lon.size), C.c_int(action), _p(_f(land), C.c_float), _p(lon, C.c_double),
                        _p(lat, C.c_double), _p(_d(z), C.c_double), _p(_d(prev_lon), C.c_double),
                        _p(_d(prev_lat), C.c_double), _p(status, C.c_int), _p(moving, C.c_int),
                        C.c_int(stranded_code), _p(age, C.c_float) if age is not None else None,
                        C.c_int(seeded_on_land_code))


def leeway(lon, lat, moving, aux, xwind, ywind, u, v, dt, capsize_fraction, uniforms, cap_uniforms=None,
           wind_threshold=30.0, wind_threshold_sigma=5.0):
    """aux: list of 9 float32 arrays (mutated: crosswind_slope / orientation flip on jibing, capsized).
    cap_uniforms: np.random.rand(len(can_be_capsized)) of processes:capsizing, or None (capsizing off)."""
    n = lon.size
    ptrs = (C.POINTER(C.c_float) * 9)(*[_p(a, C.c_float) for a in aux])
    cu = None if cap_uniforms is None else _d(np.concatenate([np.asarray(cap_uniforms, dtype=np.float64), [0.0]]))
    lib().orc_leeway(C.c_long(n), _p(lon, C.c_double), _p(lat, C.c_double), _p(_i(moving), C.c_int), ptrs,
                     _p(_f(xwind), C.c_float), _p(_f(ywind), C.c_float), _p(_f(u), C.c_float), _p(_f(v), C.c_float),
                     C.c_double(dt), C.c_double(capsize_fraction), _p(_d(uniforms), C.c_double),
                     None if cu is None else _p(cu, C.c_double), C.c_double(wind_threshold), C.c_double(wind_threshold_sigma))
