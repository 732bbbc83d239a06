"""Synthetic forcing fields shaped like BASELINE.json's configs (NumPy only, input generation).

Shapes follow SURVEY.md section 8(d): C3 = z-level lon/lat grid (u, v, w, K, depth, landmask),
C4 = NorKyst-800-shaped polar-stereographic surface grid (current, wind, Stokes, landmask).
Block coordinates are float32 like the reference's file readers hand them out
(reader_netCDF_CF_generic.py:586-587, reader_ROMS_native.py:734-735).
"""
import numpy as np

NORKYST_PROJ4 = ('+proj=stere +lat_0=90 +lon_0=70 +lat_ts=60 +a=6371000 '
                 '+rf=298.257223563 +units=m +no_defs')
# the same projection as keyword arguments for the device / oracle projection structs
NORKYST_PROJ = dict(kind='stere_polar', a=6371000.0, rf=298.257223563, lat0=90.0, lon0=70.0,
                    lat_ts=60.0, k0=1.0, x0=0.0, y0=0.0)


def grid3d(nx=1024, ny=1024, nz=12, nt=3, seed=0, lon0=0.0, lon1=10.0, lat0=60.0, lat1=66.0,
           dt_level=3600.0, coast=True):
    """C3: rectilinear lon/lat grid with z levels; eddy field decaying with depth."""
    rng = np.random.default_rng(seed)
    x = np.linspace(lon0, lon1, nx).astype(np.float32)
    y = np.linspace(lat0, lat1, ny).astype(np.float32)
    zfull = np.array([0, -5, -10, -20, -30, -50, -75, -100, -150, -200, -300, -500, -750, -1000],
                     dtype=np.float64)
    z = zfull[:nz]
    X, Y = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    t = np.arange(nt) * dt_level
    shape2 = (ny, nx)
    u = np.empty((nt, nz) + shape2, np.float32)
    v = np.empty((nt, nz) + shape2, np.float32)
    w = np.empty((nt, nz) + shape2, np.float32)
    K = np.empty((nt, nz) + shape2, np.float32)
    for it in range(nt):
        ph = 0.3 * it
        psi_u = -np.sin(2 * np.pi * (X + 0.05 * ph)) * np.cos(2 * np.pi * Y) * 0.6
        psi_v = np.cos(2 * np.pi * (X + 0.05 * ph)) * np.sin(2 * np.pi * Y) * 0.6
        for k in range(nz):
            dec = np.exp(z[k] / 100.0)
            u[it, k] = (psi_u * dec + 0.05 * np.sin(7 * X + it)).astype(np.float32)
            v[it, k] = (psi_v * dec + 0.05 * np.cos(5 * Y - it)).astype(np.float32)
            w[it, k] = (1e-3 * np.sin(2 * np.pi * X) * np.sin(2 * np.pi * Y) * dec).astype(np.float32)
            K[it, k] = (1e-2 * np.exp(z[k] / 30.0) * (1 + 0.5 * np.sin(3 * X + 2 * Y + ph)) + 1e-5).astype(np.float32)
    depth = (50 + 450 * (0.5 + 0.5 * np.sin(2 * X + 1.0) * np.cos(1.5 * Y))).astype(np.float32)
    land = np.zeros(shape2, np.float32)
    if coast:
        strip = X > (0.94 + 0.03 * np.sin(9 * Y))
        land[strip] = 1.0
        u[:, :, strip] = np.nan
        v[:, :, strip] = np.nan
        w[:, :, strip] = np.nan
        K[:, :, strip] = np.nan
    noise = (0.02 * rng.standard_normal(shape2)).astype(np.float32)
    u[:, 0] += noise
    return dict(x=x, y=y, z=z, t=t, x_sea_water_velocity=u, y_sea_water_velocity=v,
                upward_sea_water_velocity=w, ocean_vertical_diffusivity=K,
                sea_floor_depth_below_sea_level=np.broadcast_to(depth, (nt,) + shape2).copy(),
                land_binary_mask=np.broadcast_to(land, (nt,) + shape2).copy())


def grid_stere(nx=2602, ny=902, nt=3, seed=0, dx=800.0, xc=-2.8e6, yc=-1.26e6, dt_level=3600.0):
    """C4: NorKyst-800-shaped surface fields on a polar-stereographic grid (metres)."""
    rng = np.random.default_rng(seed)
    x = (xc + dx * (np.arange(nx) - nx // 2)).astype(np.float32)
    y = (yc + dx * (np.arange(ny) - ny // 2)).astype(np.float32)
    X, Y = np.meshgrid(np.linspace(0, 2, nx), np.linspace(0, 1, ny))
    t = np.arange(nt) * dt_level
    shape2 = (ny, nx)
    land = np.zeros(shape2, np.float32)
    band = X > (1.86 + 0.05 * np.sin(11 * Y) + 0.03 * np.sin(37 * Y))
    land[band] = 1.0
    out = dict(x=x, y=y, t=t)
    u = np.empty((nt,) + shape2, np.float32)
    v = np.empty_like(u)
    xw = np.empty_like(u)
    yw = np.empty_like(u)
    for it in range(nt):
        a = 0.25 * np.sin(0.5 * it)
        f = a * X * X + (1 - 2 * a) * X
        uu = -np.pi * 0.25 * np.sin(np.pi * f) * np.cos(np.pi * Y)
        vv = np.pi * 0.25 * np.cos(np.pi * f) * np.sin(np.pi * Y) * (2 * a * X + 1 - 2 * a)
        u[it] = (uu + 0.05 * rng.standard_normal(shape2)).astype(np.float32)
        v[it] = (vv + 0.05 * rng.standard_normal(shape2)).astype(np.float32)
        xw[it] = (8 + 2 * np.sin(3 * X + it)).astype(np.float32)
        yw[it] = (2 * np.cos(4 * Y - it)).astype(np.float32)
        u[it][band] = np.nan
        v[it][band] = np.nan
    out['x_sea_water_velocity'] = u
    out['y_sea_water_velocity'] = v
    out['x_wind'] = xw
    out['y_wind'] = yw
    out['sea_surface_wave_stokes_drift_x_velocity'] = (0.015 * xw).astype(np.float32)
    out['sea_surface_wave_stokes_drift_y_velocity'] = (0.015 * yw).astype(np.float32)
    out['land_binary_mask'] = np.broadcast_to(land, (nt,) + shape2).copy()
    return out
