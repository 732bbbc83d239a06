"""The rotated pole (`+proj=ob_tran +o_proj=longlat`) on the device, and the node-lookup lane for proj4 strings the device has
no closed form for.

pyproj is not available in this environment; the checker below is the reference's algorithm with the projection evaluated
EXACTLY and INDEPENDENTLY of oracle/proj.c (a rotated pole is a rotation of the sphere: two matrix products): lonlat2xy ->
fractional pixel -> bilinear interpolation of the grid-relative components (float32 block values, float64 weights, as
basereader/interpolation) -> rotate_vectors AT THE ELEMENT with the reference's recipe (variables.py:80-108: 0.1 degree
along the reader's y axis, WGS84 azimuth) -> geod.fwd (oracle), Euler and the reference's RK4 (physics_methods.py:611-691).

 * Round 5: the device evaluates the rotated pole in closed form (PROJ_OB_TRAN: proj_fwd / proj_inv / the geodesic-inverse
   azimuth of the 0.1-degree line) -- the exact lane: 1e-8 degrees after 12 RK4 steps (the reference's own run on such a
   reader is golden c23, tests/test_gpu_model_api.py).
 * A string parse_proj4 refuses (here: the `+to_meter` form of ob_tran, whose coordinates PROJ hands out in degrees) is still
   served through the node-array lookup with the vector pairs rotated at the nodes (readers.NodeLookupGridReader) when the
   reader brings its 2-D lon / lat: a second-order approximation in the cell size -- piecewise linear over the triangulated
   nodes (structured.py:438-472), rotation at the nodes instead of at the element.  On a 0.02-degree (2.2 km) mesh: inside the
   north-star 1e-6 degrees after 12 RK4 steps.  (Round 4 quoted 3.1e-6 degrees for a 0.05-degree mesh: most of that was the
   checker starting from the float64 seed positions while the model, like the reference, keeps seeds as float32.)"""
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


def _grid(d=0.05):
    x = np.arange(-6.0, 6.0001, d)          # rotated longitude / latitude of the nodes (degrees): ~5.5 km cells at d = 0.05
    y = np.arange(-4.0, 4.0001, d)
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


# PROJ puts the new pole at (lon_0 + 180, o_lat_p) and counts the rotated longitude from the meridian that runs from the new
# pole AWAY from the old one; the matrices above count it from the meridian towards the old pole: o_lon_p = 180
PROJ4 = '+proj=ob_tran +o_proj=longlat +lon_0=%r +o_lat_p=%r +o_lon_p=180 +R=6.371e+06 +no_defs' % (POLE_LON + 180.0, POLE_LAT)
PROJ4_REFUSED = PROJ4 + ' +to_meter=0.0174532925199433'


def _reference_path(x, y, u, v, lon0, lat0, scheme, steps, dt, f32_first=None):
    """the reference's Euler / RK4 (physics_methods.py:611-691) on the exact-transform velocity"""
    lon, lat = lon0.copy(), lat0.copy()
    moving = np.ones(len(lon), np.int32)
    for k in range(steps):
        t = k * dt / 3600.0
        lon_s = lon
        if k == 0 and f32_first is not None:
            # the first get_environment of a run works on float32 element arrays: modulate_longitude in float32
            # (variables.py:259-280 with elements.py:71-88; branch 1: a corner of the reader's domain has a negative longitude)
            l32 = lon.astype(np.float32)
            lon_s = ((np.mod(l32 + np.float32(180), np.float32(360)) - np.float32(180)) if f32_first == 1
                     else np.mod(l32, np.float32(360))).astype(np.float64)
        u1, v1 = _exact_velocity(x, y, u, v, lon_s, lat, t)
        if scheme == 'runge-kutta4':
            def stage(uu, vv):
                lo, la = lon.copy(), lat.copy()
                orc.update_positions(lo, la, uu, vv, moving, dt * 0.5)
                return lo, la
            lo, la = stage(u1, v1)
            u2, v2 = _exact_velocity(x, y, u, v, lo, la, t + dt / 7200.0)
            lo, la = stage(u2, v2)
            u3, v3 = _exact_velocity(x, y, u, v, lo, la, t + dt / 7200.0)
            lo, la = stage(u3, v3)
            u4, v4 = _exact_velocity(x, y, u, v, lo, la, t + dt / 3600.0)
            u1 = ((u1 + np.float32(2) * u2 + np.float32(2) * u3 + u4) / np.float32(6.0)).astype(np.float32)
            v1 = ((v1 + np.float32(2) * v2 + np.float32(2) * v3 + v4) / np.float32(6.0)).astype(np.float32)
        orc.update_positions(lon, lat, u1, v1, moving, dt)
    return lon, lat


def test_the_independent_transform_is_the_oracles_rotated_pole():
    """(CPU work inside a GPU module: the matrix form above and oracle/proj.c's ob_tran agree, so the checker is the same
    projection the golden c23 was written with)"""
    op = orc.make_proj(orc.PROJ_OB_TRAN, lon0=POLE_LON + 180.0, lat1=POLE_LAT, lat2=180.0)
    rng = np.random.default_rng(0)
    lon, lat = _from_rotated(rng.uniform(-6, 6, 400), rng.uniform(-4, 4, 400))
    rx, ry = _to_rotated(lon, lat)
    ox, oy = orc.proj_fwd(op, lon, lat)
    assert np.abs((ox - rx + 180) % 360 - 180).max() < 1e-11 and np.abs(oy - ry).max() < 1e-11


@pytest.mark.parametrize('scheme', ['euler', 'runge-kutta4'])
def test_rotated_pole_reader_in_closed_form_on_the_device(scheme):
    x, y, lon2d, lat2d, u, v = _grid()
    times = [T0 + timedelta(hours=k) for k in range(3)]
    r = readers.GridReader(x, y, times, {'x_sea_water_velocity': u, 'y_sea_water_velocity': v}, proj4=PROJ4)
    assert type(r) is readers.GridReader and r.projected
    o = OceanDrift(loglevel=50, seed=0, stage_math='exact')
    o.add_reader(r)
    o.set_config('environment:constant:land_binary_mask', 0)
    o.set_config('drift:advection_scheme', scheme)
    rng = np.random.default_rng(3)
    n = 3000
    lon0, lat0 = _from_rotated(rng.uniform(-5.0, 5.0, n), rng.uniform(-3.0, 3.0, n))
    # seed_elements keeps positions as float32 (elements.py:71-88), here as in the reference: the checker starts where the model does
    lon0, lat0 = lon0.astype(np.float32).astype(np.float64), lat0.astype(np.float32).astype(np.float64)
    o.seed_elements(lon=lon0, lat=lat0, time=T0)
    steps, dt = 12, 600.0
    o.run(time_step=dt, steps=steps)
    e = o.elements
    assert len(e.ID) == n
    exlons, _ = r.xy2lonlat(np.array([r.xmin, r.xmin, r.xmax, r.xmax]), np.array([r.ymin, r.ymax, r.ymax, r.ymin]))
    lon, lat = _reference_path(x, y, u, v, lon0, lat0, scheme, steps, dt, f32_first=1 if np.min(exlons) < 0 else 2)
    dlon, dlat = np.abs(e.lon - lon[e.ID]).max(), np.abs(e.lat - lat[e.ID]).max()
    print('rotated pole in closed form, 12 %s steps: max deviation from the exact-transform path %.2e / %.2e deg' % (scheme, dlon, dlat))
    assert np.hypot(lon - lon0, lat - lat0).max() > 0.03
    # (the float32 environment differs in its last bit now and then between the two evaluations of the same formulas)
    assert dlon < 1e-8 and dlat < 1e-8


def test_refused_without_node_coordinates():
    x, y, lon2d, lat2d, u, v = _grid()
    with pytest.raises(NotImplementedError, match='lon=, lat='):
        readers.GridReader(x, y, [T0], {'x_sea_water_velocity': u[:1], 'y_sea_water_velocity': v[:1]}, proj4=PROJ4_REFUSED)


def test_unknown_projection_through_the_node_lookup_follows_the_exact_transform():
    x, y, lon2d, lat2d, u, v = _grid(0.02)
    times = [T0 + timedelta(hours=k) for k in range(3)]
    r = readers.GridReader(x, y, times, {'x_sea_water_velocity': u, 'y_sea_water_velocity': v}, proj4=PROJ4_REFUSED,
                           lon=lon2d, lat=lat2d)
    assert isinstance(r, readers.NodeLookupGridReader) and not r.projected
    # the grid's y axis is turned against north by up to 15 degrees here: the rotation matters
    turn = readers.node_y_azimuth(lon2d, lat2d)
    assert np.abs(turn).max() > 10.0
    o = OceanDrift(loglevel=50, seed=0, stage_math='exact')
    o.add_reader(r)
    o.set_config('environment:constant:land_binary_mask', 0)
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    rng = np.random.default_rng(3)
    n = 3000
    lon0, lat0 = _from_rotated(rng.uniform(-5.0, 5.0, n), rng.uniform(-3.0, 3.0, n))
    # seed_elements keeps positions as float32 (elements.py:71-88), here as in the reference: the checker starts where the model does
    lon0, lat0 = lon0.astype(np.float32).astype(np.float64), lat0.astype(np.float32).astype(np.float64)
    o.seed_elements(lon=lon0, lat=lat0, time=T0)
    steps, dt = 12, 600.0
    o.run(time_step=dt, steps=steps)
    e = o.elements
    assert len(e.ID) == n
    lon, lat = _reference_path(x, y, u, v, lon0, lat0, 'runge-kutta4', steps, dt)
    dlon, dlat = np.abs(e.lon - lon[e.ID]).max(), np.abs(e.lat - lat[e.ID]).max()
    moved = np.hypot(lon - lon0, lat - lat0).max()
    print('node lookup, 0.02-degree mesh, 12 RK4 steps: max deviation from the exact-transform path %.2e / %.2e deg (moved up to %.3f deg)' % (dlon, dlat, moved))
    assert moved > 0.03
    assert dlon < 1e-6 and dlat < 1e-6
    # without the rotation the same run is far off: the check above is sensitive to it
    ue, ve = _exact_velocity(x, y, u, v, lon0, lat0, 0.0)
    rx, ry = _to_rotated(lon0, lat0)
    assert np.abs(ue - (0.6 * np.sin(0.9 * rx) * np.cos(0.7 * ry) + 0.15)).max() > 0.05
