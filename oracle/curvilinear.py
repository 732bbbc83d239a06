"""TEST INFRASTRUCTURE ONLY -- CPU oracle of StructuredReader.lonlat2xy for readers WITHOUT a projection.

Reference: opendrift/readers/basereader/structured.py:44-113 builds

    spl_x = LinearNDInterpolator((lon.ravel(), lat.ravel()), block_x.ravel(), fill_value=nan)
    spl_y = deepcopy(spl_x); spl_y.values[:, 0] = block_y.ravel()

and lonlat2xy (:438-472) evaluates them.  scipy (the reference's pinned dependency, scipy>=1.9; 1.15.3 here) is
installed, so `lonlat2xy_reference` below IS that construction.  Pinned by tests/golden/c6_curvilinear_rk4.npz
(qlon, qlat -> qx, qy written by the reference's own StructuredReader, oracle/gen_golden.py::c6_curvilinear).

The product builds the same triangulation from the structured mesh (opendrift_amd/csrc/odr_mesh.h: Delaunay diagonal
per cell + Lawson edge flips) instead of from a point cloud; tests/test_curvilinear_mesh.py compiles that header with
g++ and checks on the CPU that every triangle it makes is one of qhull's and that its interpolation agrees with the
functions below; tests/test_gpu_curvilinear.py checks the device kernel against both.
"""
import copy

import numpy as np


def interpolators(lon2d, lat2d):
    from scipy.interpolate import LinearNDInterpolator
    ny, nx = lon2d.shape
    block_x, block_y = np.mgrid[0:nx, 0:ny]          # structured.py:86-88 with xmin = ymin = 0
    block_x, block_y = block_x.T, block_y.T
    spl_x = LinearNDInterpolator((lon2d.ravel(), lat2d.ravel()), block_x.ravel(), fill_value=np.nan)
    spl_y = copy.deepcopy(spl_x)
    spl_y.values[:, 0] = block_y.ravel()
    return spl_x, spl_y


def lonlat2xy_reference(lon2d, lat2d, lon, lat, spl=None):
    spl_x, spl_y = spl or interpolators(lon2d, lat2d)
    return spl_x(lon, lat), spl_y(lon, lat)


def inside_outline(lon2d, lat2d, lon, lat):
    """True where (lon, lat) lies inside the outline polygon of the mesh (even-odd rule).  qhull's triangulation
    covers the CONVEX HULL of the nodes: between the outline and the hull it has slivers joining boundary nodes,
    where the reference returns finite (meaningless) pixel indices and the product returns 'not covered'."""
    ol = np.concatenate([lon2d[0, :-1], lon2d[:-1, -1], lon2d[-1, :0:-1], lon2d[:0:-1, 0]])
    oa = np.concatenate([lat2d[0, :-1], lat2d[:-1, -1], lat2d[-1, :0:-1], lat2d[:0:-1, 0]])
    x1, y1, x2, y2 = ol, oa, np.roll(ol, -1), np.roll(oa, -1)
    lon, lat = np.atleast_1d(lon)[:, None], np.atleast_1d(lat)[:, None]
    with np.errstate(divide='ignore', invalid='ignore'):
        cross = ((y1 > lat) != (y2 > lat)) & (lon < (x2 - x1) * (lat - y1) / (y2 - y1) + x1)
    return (cross.sum(axis=1) % 2) == 1
