"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the landmask raster lookup and of coastline_crossing.

NumPy restatement of
  reader_global_landmask.Reader._on_land / get_variables   opendrift/readers/reader_global_landmask.py:226-255
      (the landmask DATA -- GSHHG through roaring_landmask -- is replaced by a lon/lat raster with the same
       contains_many interface, see oracle/gen_golden_landmask.py)
  coastline_crossing                                        opendrift/models/basemodel/__init__.py:81-134
  interact_with_coastline with general:coastline_approximation_precision   opendrift/models/basemodel/__init__.py:694-746
Pinned by tests/golden/c10_landmask_crossing.npz (the reference's own function and two reference runs).
"""
import numpy as np


class RasterMask:
    def __init__(self, lon0, lat0, dlon, dlat, cells):
        self.lon0, self.lat0, self.dlon, self.dlat, self.cells = float(lon0), float(lat0), float(dlon), float(dlat), cells

    @classmethod
    def from_golden(cls, g):
        cells = np.unpackbits(g['raster_cells'], axis=1)[:, :int(g['raster_nx'])]
        return cls(g['raster_lon0'], g['raster_lat0'], g['raster_dlon'], g['raster_dlat'], cells)

    def contains_many(self, x, y):
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        ix = np.floor((x - self.lon0) / self.dlon).astype(np.int64)
        iy = np.floor((y - self.lat0) / self.dlat).astype(np.int64)
        ny, nx = self.cells.shape
        ok = (ix >= 0) & (ix < nx) & (iy >= 0) & (iy < ny)
        out = np.zeros(x.shape, bool)
        out[ok] = self.cells[iy[ok], ix[ok]] != 0
        return out

    def land_binary_mask(self, lon, lat):
        """Reader._on_land: longitudes modulated to [-180, 180) (basereader/variables.py modulate_longitude of a reader
        with xmin = -180), float32 like every environment variable"""
        return self.contains_many(np.mod(np.asarray(lon) + 180, 360) - 180, lat).astype(np.float32)


def coastline_crossing(mask, lon1, lat1, lon2, lat2, step_degrees, land_side=True):
    lon1, lat1, lon2, lat2 = (np.atleast_1d(np.array(a, dtype=np.float64)) for a in (lon1, lat1, lon2, lat2))
    lon_c, lat_c = (lon2.copy(), lat2.copy()) if land_side else (lon1.copy(), lat1.copy())
    for i, (a1, b1, a2, b2) in enumerate(zip(lon1, lat1, lon2, lat2)):
        xd, yd = np.abs(a2 - a1), np.abs(b2 - b1)
        if xd == 0 and yd == 0:
            continue
        if xd > 180:                      # crossing the dateline
            if a1 < 0:
                a2 = a2 - 360
                xd = np.abs(a2 - a1)
        xs = np.floor(xd / step_degrees).astype(np.int64) if xd > step_degrees else 1
        ys = np.floor(yd / step_degrees).astype(np.int64) if yd > step_degrees else 1
        xx, yy = np.meshgrid(np.linspace(a1, a2, xs), np.linspace(b1, b2, ys))
        xx, yy = xx.ravel(), yy.ravel()
        m = mask.contains_many(xx, yy)
        if np.any(m):
            index = np.argmax(m)
            if land_side is False:
                index = np.maximum(0, index - 1)
            lon_c[i], lat_c[i] = xx[index], yy[index]
    return lon_c, lat_c
