"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/c10_landmask_crossing.npz from the REFERENCE ITSELF.

SURVEY.md section 8 (f3): the global landmask lookup (`reader_global_landmask.Reader`,
readers/reader_global_landmask.py:201-255) and `coastline_crossing` -- the per-stranded-element transect search of
`interact_with_coastline` when `general:coastline_approximation_precision` is set
(models/basemodel/__init__.py:81-134, 694-746).

The GSHHG data behind `roaring_landmask` (a Rust extension with a 15" bitmap + polygons) is not vendored and not
installed, so the DATA is replaced by a synthetic lon/lat raster (`RasterMask`: a wiggly coast, a fjord and two
islands on a 0.005 deg grid) with the same `contains_many(x, y)` interface; everything that is reference CODE runs
unmodified on it: `get_mask()` hands the raster to the reference's own Reader (auto landmask, ContinuousReader path:
exact values at the element positions, modulate_longitude) and `opendrift.models.basemodel.rl` is the same object, so
`coastline_crossing` samples it along its transect rectangle exactly as it samples GSHHG.

    python oracle/gen_golden_landmask.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (installs the shim)
import opendrift.models.basemodel as bm  # noqa: E402
from opendrift.readers import reader_constant, reader_global_landmask  # noqa: E402


class RasterMask:
    """cells[iy, ix] covers lon0 + [ix, ix+1) dlon x lat0 + [iy, iy+1) dlat; ocean outside the raster"""

    def __init__(self, lon0, lat0, dlon, dlat, cells):
        self.lon0, self.lat0, self.dlon, self.dlat, self.cells = lon0, lat0, dlon, dlat, cells

    def contains_many(self, x, y):
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        ix = np.floor((x - self.lon0) / self.dlon).astype(np.int64)
        iy = np.floor((y - self.lat0) / self.dlat).astype(np.int64)
        ny, nx = self.cells.shape
        ok = (ix >= 0) & (ix < nx) & (iy >= 0) & (iy < ny)
        out = np.zeros(x.shape, bool)
        out[ok] = self.cells[iy[ok], ix[ok]] != 0
        return out


def make_raster():
    lon0, lat0, d = 3.0, 59.0, 0.005
    nx, ny = 800, 600
    lon = lon0 + (np.arange(nx) + 0.5) * d
    lat = lat0 + (np.arange(ny) + 0.5) * d
    LON, LAT = np.meshgrid(lon, lat)
    coast = 5.6 + 0.15 * np.sin(9 * LAT) + 0.05 * np.sin(41 * LAT)           # land east of a wiggly coast
    cells = (LON > coast)
    cells &= ~((np.abs(LAT - 60.6) < 0.02 + 0.0 * LON) & (LON < 6.2))          # a fjord cut into the land
    cells |= (LON - 5.1)**2 + (2 * (LAT - 60.2))**2 < 0.06**2                  # islands
    cells |= (LON - 5.3)**2 + (2 * (LAT - 61.1))**2 < 0.03**2
    return lon0, lat0, d, d, cells.astype(np.uint8)


def run(action, raster, precision, seed_pos):
    mask = RasterMask(*raster)
    reader_global_landmask.__roaring_mask__ = mask          # get_mask() (reader_global_landmask.py:36-47)
    bm.rl = mask                                            # coastline_crossing's landmask (basemodel/__init__.py:79)
    o = gg.OceanDrift(loglevel=50)
    o.set_config('general:use_auto_landmask', True)
    o.set_config('general:coastline_action', action)
    o.set_config('general:coastline_approximation_precision', precision)
    o.set_config('drift:advection_scheme', 'euler')
    o.set_config('drift:stokes_drift', False)
    o.set_config('drift:vertical_mixing', False)
    o.set_config('drift:vertical_advection', False)
    o.add_reader(reader_constant.Reader({'x_sea_water_velocity': 1.1, 'y_sea_water_velocity': 0.35,
                                         'x_wind': 0.0, 'y_wind': 0.0}))
    lon, lat, z = seed_pos
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, z=z, time=gg.T0, wind_drift_factor=0.0)
    res, _ = gg._run(o, 900, 14)
    assert type(o.env.readers['global_landmask']).__module__ == 'opendrift.readers.reader_global_landmask'
    return res, np.array(o.status_categories)


def main():
    from opendrift.models.basemodel import coastline_crossing
    raster = make_raster()
    rng = np.random.default_rng(10)
    N = 400
    lon = rng.uniform(5.0, 5.6, N)
    lat = rng.uniform(59.6, 61.6, N)
    z = np.where(np.arange(N) % 7 == 0, -3.0, 0.0)
    on_land = RasterMask(*raster).contains_many(lon, lat)
    lon[on_land] -= 0.5                                      # nothing seeded on land
    out = {}
    for action in ('stranding', 'previous'):
        res, cats = run(action, raster, 0.001, (lon, lat, z))
        out.update({action + '_' + k: v for k, v in res.items()})
        out[action + '_categories'] = cats
        print(action, 'stranded/deactivated at the end:', int((res['status'][-1] > 0).sum()), cats)
    # the function itself (basemodel/__init__.py:81-134) on random transects, both sides, incl. degenerate ones
    mask = RasterMask(*raster)
    bm.rl = mask
    M = 500
    lon1 = rng.uniform(4.8, 5.5, M)
    lat1 = rng.uniform(59.6, 61.6, M)
    lon2 = lon1 + rng.uniform(-0.02, 0.35, M)
    lat2 = lat1 + rng.uniform(-0.08, 0.08, M)
    lon2[:10], lat2[:10] = lon1[:10], lat1[:10]              # no displacement
    lat2[10:20] = lat1[10:20]                                # zonal transects (one y sample)
    lon2[20:30] = lon1[20:30] + 0.0004                       # shorter than the step in x
    for side in (True, False):
        lc, la = coastline_crossing(lon1.copy(), lat1.copy(), lon2.copy(), lat2.copy(), 0.001, land_side=side)
        out['fn_lon_c_%s' % side], out['fn_lat_c_%s' % side] = lc, la
    out.update(fn_lon1=lon1, fn_lat1=lat1, fn_lon2=lon2, fn_lat2=lat2)
    np.savez_compressed(os.path.join(gg.GOLD, 'c10_landmask_crossing.npz'), dt=900.0, precision=0.001, u=1.1, v=0.35,
                        raster_lon0=raster[0], raster_lat0=raster[1], raster_dlon=raster[2], raster_dlat=raster[3],
                        raster_cells=np.packbits(raster[4], axis=1), raster_nx=raster[4].shape[1], **out)


if __name__ == '__main__':
    main()
