"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/c20_lcc_merc_rk4.npz from the REFERENCE ITSELF.

Readers whose proj4 is a Lambert conformal conic (MEPS / AROME-Arctic / NORA3 style: tangent cone on a sphere, and a
two-parallel cone on WGS84) or a Mercator projection (true-scale latitude on WGS84): lonlat2xy through the projection
(variables.py:111-143), vectors rotated from the reader's axes to east / north by the azimuth of its +y axis
(rotate_vectors, variables.py:59-109).  pyproj is not installed here: the reference runs on the shim of oracle/refshim.py
whose merc / lcc arithmetic is oracle/proj.c (Snyder ch. 7 / 15, pinned on Snyder's numerical examples) -- for these two
projections the golden pins the DEVICE against the reference's control flow and the oracle's projection, not against PROJ.

Scenario per projection: C4-shaped surface fields (current, wind, Stokes drift, land mask) on an 800 m grid, RK4 +
wind drift + Stokes drift + stranding, no random terms.

    python oracle/gen_golden_proj.py
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
from opendrift_amd import synthetic as synth  # noqa: E402

PROJ4 = {
    'lcc_sphere': '+proj=lcc +lat_0=63.3 +lon_0=15 +lat_1=63.3 +lat_2=63.3 +R=6371000 +no_defs',
    'lcc_wgs84': '+proj=lcc +lat_0=66 +lon_0=-20 +lat_1=60 +lat_2=72 +x_0=400000 +y_0=-150000 +ellps=WGS84 +no_defs',
    'merc_wgs84': '+proj=merc +lon_0=5 +lat_ts=62 +ellps=WGS84 +no_defs',
}
CENTRE = {'lcc_sphere': (4.5, 60.5), 'lcc_wgs84': (-14.0, 64.8), 'merc_wgs84': (8.0, 66.0)}
NAMES = ('x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind', 'sea_surface_wave_stokes_drift_x_velocity',
         'sea_surface_wave_stokes_drift_y_velocity', 'land_binary_mask')


def run(tag):
    import pyproj
    p = pyproj.Proj(PROJ4[tag])
    xc, yc = p(*CENTRE[tag])
    g = synth.grid_stere(nx=70, ny=50, nt=3, seed=20, xc=float(np.round(xc, -2)), yc=float(np.round(yc, -2)))
    times = [gg.T0 + timedelta(seconds=float(t)) for t in g['t']]
    o = gg._base('runge-kutta4')
    r = gg.GridReader(PROJ4[tag], g['x'], g['y'], times, {k: g[k] for k in NAMES})
    o.add_reader(r)
    o.set_config('general:coastline_action', 'stranding')
    o.set_config('general:coastline_approximation_precision', None)
    o.set_config('drift:stokes_drift', True)
    rng = np.random.default_rng(21)
    N = 300
    x = rng.uniform(g['x'][5], g['x'][-6], N)
    y = rng.uniform(g['y'][5], g['y'][-6], N)
    lon, lat = r.xy2lonlat(x, y)
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, time=gg.T0, wind_drift_factor=0.03)
    res, _ = gg._run(o, 900, 8)
    print(tag, 'stranded', int((res['status'][-1] > 0).sum()), o.status_categories,
          'moved %.4f deg' % np.nanmax(np.abs(res['lon'][-1] - res['lon'][0])))
    out = {('%s_%s' % (tag, k)): v for k, v in res.items()}
    out.update({('%s_g_%s' % (tag, k)): v for k, v in g.items()})
    return out


def main():
    out = dict(dt=900.0, wdf=0.03)
    for tag in PROJ4:
        out.update(run(tag))
        out[tag + '_proj4'] = PROJ4[tag]
    np.savez_compressed(os.path.join(gg.GOLD, 'c20_lcc_merc_rk4.npz'), **out)


if __name__ == '__main__':
    main()
