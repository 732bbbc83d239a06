"""odr_movers: advect_wind -> stokes_drift -> horizontal_diffusion of one step in ONE launch (k_movers) against the three
entry points called one after the other (physics_methods.py:712-791, :793-848, basemodel/__init__.py:1746-1772): bit-identical
positions for every subset of the movers, with host-drawn and device-drawn normals, on a polar-stereographic reader with
wind, Stokes drift and a constant diffusivity (the C4 shape); a mover whose global early-out holds (calm wind, no Stokes
drift, zero diffusivity) must be skipped as a whole in both lanes."""
import numpy as np
import pytest

from opendrift_amd import synthetic as synth
from opendrift_amd.device import Context

pytestmark = pytest.mark.gpu

U, V, LAND = 'x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask'
XW, YW = 'x_wind', 'y_wind'
SX, SY = 'sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity'
HD = 'horizontal_diffusivity'


def _setup(calm=False, no_stokes=False, hd=10.0, n=30000, seed=4):
    from opendrift_amd.projection import stere_polar_inverse
    g = synth.grid_stere(nx=260, ny=90, nt=3, seed=0)
    if calm:
        g[XW][:] = 0
        g[YW][:] = 0
    if no_stokes:
        g[SX][:] = 0
        g[SY][:] = 0
    names = [U, V, XW, YW, SX, SY, LAND]
    ctx = Context(seed=3)
    sid = ctx.add_grid(g['x'], g['y'], proj=synth.NORKYST_PROJ)
    for k in range(3):
        ctx.upload_block(sid, k, float(g['t'][k]), {nm: g[nm][k] for nm in names})
    for nm in names:
        ctx.bind(nm, [sid], np.nan if nm == LAND else 0.0)
    cs = ctx.add_constant({HD: hd})
    ctx.bind(HD, [cs], 0.0)
    rng = np.random.default_rng(seed)
    x = rng.uniform(g['x'][4], g['x'][-5], n)
    y = rng.uniform(g['y'][4], g['y'][-5], n)
    lon, lat = stere_polar_inverse(x, y, **synth.NORKYST_PROJ)
    z = np.where(rng.random(n) < 0.5, 0.0, -rng.uniform(0, 3, n))
    return ctx, names + [HD], lon, lat, z


def _pair(ctx, names, lon, lat, z, which, host_normals, steps=3, dt=900.0):
    n = len(lon)
    P, Q = ctx.particles(n), ctx.particles(n)
    rng = np.random.default_rng(77)
    for X in (P, Q):
        X.append(lon, lat, z=z)
    for k in range(steps):
        t = 900.0 * k
        normals = (rng.standard_normal(n), rng.standard_normal(n)) if host_normals else None
        for X in (P, Q):
            X.env_sample(names, t)
        if 'wind' in which:
            P.advect_wind(dt, wind_drift_depth=0.1)
        if 'stokes' in which:
            P.stokes_drift(dt, profile=2, hs_mode=1, tp_mode=1)
        if 'hdiff' in which:
            P.hdiffusion(dt, step=k, normals=normals)
        Q.movers(dt, wind=dict(wind_drift_depth=0.1) if 'wind' in which else None,
                 stokes=dict(profile=2, hs_mode=1, tp_mode=1) if 'stokes' in which else None,
                 hdiffusion=dict(step=k, normals=normals) if 'hdiff' in which else None)
        a, b = P.download(), Q.download()
        for q in ('lon', 'lat', 'z', 'ID'):
            eq = (a[q] == b[q]) | ((a[q] != a[q]) & (b[q] != b[q]))
            assert eq.all(), (which, k, q, int((~eq).sum()))
    moved = bool((a['lon'] != lon).any())
    P.close()
    Q.close()
    return moved


@pytest.mark.parametrize('host_normals', [False, True])
@pytest.mark.parametrize('which', [('wind', 'stokes', 'hdiff'), ('wind', 'stokes'), ('stokes', 'hdiff'), ('wind',), ('hdiff',)])
def test_movers_in_one_launch_equal_the_separate_calls(which, host_normals):
    ctx, names, lon, lat, z = _setup()
    assert _pair(ctx, names, lon, lat, z, which, host_normals)
    ctx.close()


@pytest.mark.parametrize('case', ['calm', 'no_stokes', 'no_diffusivity', 'nothing'])
def test_movers_keep_the_global_early_outs(case):
    """wind_speed.max() == 0, stokes max == 0, D.max() == 0: the mover returns before update_positions (which would
    renormalise the longitude) -- in the fused launch as in the separate calls."""
    ctx, names, lon, lat, z = _setup(calm=case in ('calm', 'nothing'), no_stokes=case in ('no_stokes', 'nothing'),
                                     hd=0.0 if case in ('no_diffusivity', 'nothing') else 10.0, n=8000)
    moved = _pair(ctx, names, lon, lat, z, ('wind', 'stokes', 'hdiff'), False, steps=2)
    assert moved == (case != 'nothing')
    ctx.close()
