"""CPU: the host side of readers with a projection the device has no closed form for (opendrift_amd/readers.py:
NodeLookupGridReader): the WGS84 forward azimuth (Vincenty) against the oracle's geodesic inverse (Karney, oracle/geodesic.c),
the azimuth of the mesh's y axis at the nodes, and the rotation of the vector pairs of a block to east / north
(variables.py:59-108: rot = -azimuth; u' = u cos rot - v sin rot, v' = u sin rot + v cos rot)."""
from datetime import datetime, timedelta

import numpy as np
import pytest

from opendrift_amd import readers
from oracle import oracle as orc

T0 = datetime(2020, 1, 1)


def test_forward_azimuth_equals_the_geodesic_inverse():
    rng = np.random.default_rng(0)
    n = 5000
    lon1, lat1 = rng.uniform(-180, 180, n), rng.uniform(-85, 85, n)
    d = rng.uniform(1e-4, 0.5, n)                       # neighbouring grid nodes: metres to tens of kilometres
    th = rng.uniform(0, 2 * np.pi, n)
    lon2, lat2 = lon1 + d * np.sin(th) / np.cos(np.radians(lat1)), np.clip(lat1 + d * np.cos(th), -89.9, 89.9)
    az, _ = orc.geod_inv(lon1, lat1, lon2, lat2)
    mine = readers.wgs84_forward_azimuth(lon1, lat1, lon2, lat2)
    diff = (mine - az + 180.0) % 360.0 - 180.0
    assert np.abs(diff).max() < 1e-8


def test_y_axis_azimuth_of_simple_meshes():
    lon, lat = np.meshgrid(np.linspace(0, 5, 21), np.linspace(60, 62, 11))
    az = readers.node_y_azimuth(lon, lat)               # a lon / lat mesh: y is due north everywhere, last row included
    assert az.shape == lon.shape and np.abs(az).max() < 1e-9
    az = readers.node_y_azimuth(lon[::-1], lat[::-1])   # rows running south: 180 degrees
    assert np.abs(np.abs(az) - 180.0).max() < 1e-9
    az = readers.node_y_azimuth(lat.T * 0 + lon.T, lat.T)   # transposed: rows run east
    assert np.abs(az - 90.0).max() < 1.0 and np.abs(az - 90.0).max() > 1e-3      # (the parallel is not a geodesic)


def test_unknown_projection_without_node_arrays_is_refused_and_with_them_rotates_the_vector_pairs():
    x, y = np.linspace(-1, 1, 9), np.linspace(-1, 1, 7)
    X, Y = np.meshgrid(x, y)
    th = np.radians(25.0)                                # mesh turned 25 degrees clockwise against north, near the equator
    lon2d, lat2d = 10.0 + 0.1 * (X * np.cos(th) + Y * np.sin(th)), 0.1 * (-X * np.sin(th) + Y * np.cos(th))
    u = np.ones((2,) + X.shape, np.float32)              # one unit along the mesh's x axis
    v = np.zeros_like(u)
    arrays = {'x_sea_water_velocity': u, 'y_sea_water_velocity': v, 'sea_surface_height': 2 * u}
    times = [T0, T0 + timedelta(hours=1)]
    with pytest.raises(NotImplementedError, match='lon=, lat='):
        readers.GridReader(x, y, times, arrays, proj4='+proj=aea +lat_1=50 +lat_2=70')     # (Albers: no closed form on the device)
    r = readers.GridReader(x, y, times, arrays, proj4='+proj=aea +lat_1=50 +lat_2=70', lon=lon2d, lat=lat2d)
    assert isinstance(r, readers.NodeLookupGridReader) and not r.projected and r.native_proj4 == '+proj=aea +lat_1=50 +lat_2=70'
    b = r.get_variables(['x_sea_water_velocity', 'y_sea_water_velocity', 'sea_surface_height'], T0)
    # the x axis points 25 degrees south of east: east component cos 25, north component -sin 25
    # (on the ellipsoid a degree of latitude is 0.7 % shorter than a degree of longitude at the equator: 25.15 degrees)
    assert np.abs(b["x_sea_water_velocity"] - np.cos(th)).max() < 4e-3 and np.abs(b["y_sea_water_velocity"] + np.sin(th)).max() < 4e-3
    assert b['x_sea_water_velocity'].dtype == np.float32 and np.array_equal(b['sea_surface_height'], 2 * u[0])
    with pytest.raises(ValueError, match='rotated together'):
        r.get_variables(['x_sea_water_velocity'], T0)
    # masked cells (land / missing data of a netCDF block) come out as NaN, not as the fill value turned into a velocity
    um = np.ma.masked_array(u.copy(), mask=np.zeros(u.shape, bool))
    um.mask[:, 2, 3] = True
    rm = readers.GridReader(x, y, times, {'x_sea_water_velocity': um, 'y_sea_water_velocity': np.ma.masked_array(v.copy(), mask=um.mask)},
                            proj4='+proj=aea +lat_1=50 +lat_2=70', lon=lon2d, lat=lat2d)
    bm = rm.get_variables(['x_sea_water_velocity', 'y_sea_water_velocity'], T0)
    assert np.isnan(bm['x_sea_water_velocity'][2, 3]) and np.isnan(bm['y_sea_water_velocity'][2, 3])
    assert np.isfinite(np.delete(bm['x_sea_water_velocity'].ravel(), 2 * len(x) + 3)).all() and np.nanmax(np.abs(bm['x_sea_water_velocity'])) < 2
    # utm, tmerc, laea, every aspect of stere and the rotated pole have closed forms on the device since round 5
    for p4 in ('+proj=utm +zone=33 +ellps=WGS84', '+proj=laea +lat_0=52 +lon_0=10 +ellps=GRS80', '+proj=stere +lat_0=52 +lon_0=5 +ellps=WGS84',
               '+proj=ob_tran +o_proj=longlat +lon_0=-40 +o_lat_p=22 +R=6.371e+06'):
        assert type(readers.GridReader(x, y, times, arrays, proj4=p4)) is readers.GridReader
    # a known projection is untouched by lon= / lat=
    assert type(readers.GridReader(x, y, times, arrays, proj4='+proj=latlong', lon=lon2d, lat=lat2d)) is readers.GridReader
