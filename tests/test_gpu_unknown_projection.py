"""A reader whose proj4 the device has no closed form for (here a rotated-pole grid, `+proj=ob_tran`): GridReader(...,
lon=, lat=) serves it through the node-array lookup with the vector pairs rotated at the nodes (opendrift_amd/readers.py:
NodeLookupGridReader; the reference hands such a string to pyproj, basereader/variables.py:59-143).

pyproj is not available in this environment, so the reference cannot be run on this case; the checker below is the
reference's algorithm with the projection evaluated EXACTLY (a rotated pole is a rotation of the sphere: two matrix
products): lonlat2xy -> fractional pixel -> bilinear interpolation of the grid-relative components (float32 block
values, float64 weights, as basereader/interpolation) -> rotate_vectors AT THE ELEMENT with the reference's recipe
(variables.py:80-108: 0.1 degree along the reader's y axis, WGS84 azimuth) -> geod.fwd (oracle).  The device path
differs by construction in second order of the cell size: the node lookup is the reference's lookup for readers
WITHOUT projection -- piecewise linear over the triangulated nodes (structured.py:438-472) -- which puts an element up
to 5e-4 cells from its exact pixel position on this 0.05-degree (5.5 km) mesh at 60-70 N, and the vectors are rotated
at the nodes instead of at the element.  Measured on MI355X: 3.1e-6 degrees (0.33 m) after 12 Euler steps in a field
with 0.03 m/s of shear per cell -- NOT inside the north-star 1e-6 degrees: this lane is a served approximation for
projections without a closed form on the device, bounded here at 1e-5 degrees, and DESIGN.md says so."""
from datetime import datetime, timedelta

import numpy as np
import pytest

from opendrift_amd import readers
from opendrift_amd.oceandrift import OceanDrift
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
T0 = datetime(2020, 1, 1)
POLE_LON, POLE_LAT = -40.0, 25.0       # position of the rotated north pole


def _matrix():
    a, b = np.radians(POLE_LON), np.radians(90.0 - POLE_LAT)
    rz = np.array([[np.cos(a), np.sin(a), 0], [-np.sin(a), np.cos(a), 0], [0, 0, 1]])
    ry = np.array([[np.cos(b), 0, -np.sin(b)], [0, 1, 0], [np.sin(b), 0, np.cos(b)]])
    return ry @ rz          # geographic unit vector -> rotated frame (pole -> +z)


def _to_rotated(lon, lat):
    lo, la = np.radians(lon), np.radians(lat)
    v = np.stack([np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)])
    w = np.tensordot(_matrix(), v, axes=1)
    return np.degrees(np.arctan2(w[1], w[0])), np.degrees(np.arcsin(np.clip(w[2], -1, 1)))


def _from_rotated(rlon, rlat):
    lo, la = np.radians(rlon), np.radians(rlat)
    w = np.stack([np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)])
    v = np.tensordot(_matrix().T, w, axes=1)
    return np.degrees(np.arctan2(v[1], v[0])), np.degrees(np.arcsin(np.clip(v[2], -1, 1)))


def _grid():
    x = np.arange(-6.0, 6.0001, 0.05)          # rotated longitude / latitude of the nodes (degrees): ~5.5 km cells
    y = np.arange(-4.0, 4.0001, 0.05)
    X, Y = np.meshgrid(x, y)
    lon2d, lat2d = _from_rotated(X, Y)
    nt = 3
    u = np.empty((nt,) + X.shape, np.float32)
    v = np.empty_like(u)
    for k in range(nt):       # components along the grid's own x / y axes
        u[k] = 0.6 * np.sin(0.9 * X + 0.2 * k) * np.cos(0.7 * Y) + 0.15
        v[k] = -0.5 * np.cos(0.8 * X) * np.sin(1.1 * Y - 0.1 * k) + 0.1
    return x, y, lon2d, lat2d, u, v


def _exact_velocity(x, y, u, v, lon, lat, wt):
    """reference algorithm with the exact transform: (east, north) float32 velocity at lon / lat, time weight wt in [0, 2]."""
    rx, ry = _to_rotated(lon, lat)
    fx, fy = (rx - x[0]) / (x[-1] - x[0]) * (len(x) - 1), (ry - y[0]) / (y[-1] - y[0]) * (len(y) - 1)
    i0 = np.clip(np.floor(fx).astype(int), 0, len(x) - 2)
    j0 = np.clip(np.floor(fy).astype(int), 0, len(y) - 2)
    tx, ty = fx - i0, fy - j0

    def bil(a):
        return ((a[j0, i0] * (1 - ty) + a[j0 + 1, i0] * ty) * (1 - tx) + (a[j0, i0 + 1] * (1 - ty) + a[j0 + 1, i0 + 1] * ty) * tx).astype(np.float32)
    k0 = min(int(np.floor(wt)), u.shape[0] - 2)
    w = wt - k0
    ug = (bil(u[k0]).astype(np.float64) * (1 - w) + bil(u[k0 + 1]).astype(np.float64) * w).astype(np.float32)
    vg = (bil(v[k0]).astype(np.float64) * (1 - w) + bil(v[k0 + 1]).astype(np.float64) * w).astype(np.float32)
    lon2, lat2 = _from_rotated(rx, ry + 0.1)                 # rotate_vectors: delta_y = .1 for a geographic CRS (:80-83)
    az, _ = orc.geod_inv(lon, lat, lon2, lat2)
    rot = -np.radians(az)
    return (ug * np.cos(rot) - vg * np.sin(rot)).astype(np.float32), (ug * np.sin(rot) + vg * np.cos(rot)).astype(np.float32)


def test_refused_without_node_coordinates():
    x, y, lon2d, lat2d, u, v = _grid()
    with pytest.raises(NotImplementedError, match='lon=, lat='):
        readers.GridReader(x, y, [T0], {'x_sea_water_velocity': u[:1], 'y_sea_water_velocity': v[:1]},
                           proj4='+proj=ob_tran +o_proj=longlat +lon_0=-40 +o_lat_p=25 +R=6.371e+06 +no_defs')


def test_rotated_pole_reader_through_the_node_lookup_follows_the_exact_transform():
    x, y, lon2d, lat2d, u, v = _grid()
    times = [T0 + timedelta(hours=k) for k in range(3)]
    r = readers.GridReader(x, y, times, {'x_sea_water_velocity': u, 'y_sea_water_velocity': v},
                           proj4='+proj=ob_tran +o_proj=longlat +lon_0=-40 +o_lat_p=25 +R=6.371e+06 +no_defs',
                           lon=lon2d, lat=lat2d)
    assert isinstance(r, readers.NodeLookupGridReader) and not r.projected
    # the grid's y axis is turned against north by up to 15 degrees here: the rotation matters
    turn = readers.node_y_azimuth(lon2d, lat2d)
    assert np.abs(turn).max() > 10.0
    o = OceanDrift(loglevel=50, seed=0)
    o.add_reader(r)
    o.set_config('environment:constant:land_binary_mask', 0)
    o.set_config('drift:advection_scheme', 'euler')
    rng = np.random.default_rng(3)
    n = 3000
    lon0, lat0 = _from_rotated(rng.uniform(-5.0, 5.0, n), rng.uniform(-3.0, 3.0, n))
    o.seed_elements(lon=lon0, lat=lat0, time=T0)
    steps, dt = 12, 600.0
    o.run(time_step=dt, steps=steps)
    e = o.elements
    assert len(e.ID) == n
    lon, lat = lon0.copy(), lat0.copy()
    moving = np.ones(n, np.int32)
    for k in range(steps):
        ue, ve = _exact_velocity(x, y, u, v, lon, lat, k * dt / 3600.0)
        orc.update_positions(lon, lat, ue, ve, moving, dt)
    dlon, dlat = np.abs(e.lon - lon[e.ID]).max(), np.abs(e.lat - lat[e.ID]).max()
    moved = np.hypot(lon - lon0, lat - lat0).max()
    print('rotated pole, 12 Euler steps: max deviation from the exact-transform path %.2e / %.2e deg (moved up to %.3f deg)' % (dlon, dlat, moved))
    assert moved > 0.03
    assert dlon < 1e-5 and dlat < 1e-5
    # without the rotation the same run is far off: the check above is sensitive to it
    ue, ve = _exact_velocity(x, y, u, v, lon0, lat0, 0.0)
    rx, ry = _to_rotated(lon0, lat0)
    assert np.abs(ue - (0.6 * np.sin(0.9 * rx) * np.cos(0.7 * ry) + 0.15)).max() > 0.05
