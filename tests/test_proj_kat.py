"""CPU: the projections and the geodesic-inverse azimuth of the oracle (oracle/proj.c, geodesic.c) and of the host mirror
(opendrift_amd/projection.py) against known answers computed INDEPENDENTLY of both -- mpmath, 40 digits, every projection from
its definition (oracle/validate_projections.py -> tests/golden/proj_kat.npz; the transverse Mercator as the meridian arc
continued to complex latitude, no Krueger series).  pyproj / PROJ, which the reference calls (variables.py:111-143), are not in
this image: round 5 had Snyder's printed examples at 0.06 m; these hold forward and inverse at the float64 round-off of the
formulas, 200 points per projection.  tests/test_gpu_parity.py::test_projection_known_answers_on_the_device holds the device to
the same file."""
import ast
import os

import numpy as np
import pytest

from oracle import oracle as orc
from opendrift_amd import projection

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'proj_kat.npz')
TAGS = ['merc_wgs84', 'lcc_wgs84', 'lcc_1sp', 'stere_north', 'stere_north_ts90', 'stere_south', 'stere_oblique',
        'stere_equatorial', 'laea_europe', 'laea_north', 'utm33', 'tmerc_wide', 'rotated_pole']
# forward: metres (degrees for the rotated pole); inverse: degrees.  What the float64 formulas reach, with a margin of ~4
# (the measured worst cases are printed by the tests); north_star's 1e-6 deg is ~0.1 m.
TOL_FWD = 2e-8
TOL_INV = 1e-12


def oracle_proj(tag, kw):
    rf = kw.get('rf')
    es = (2 - 1 / rf) / rf if rf else 0.0
    a = kw.get('a', 6378137.0)
    k0, x0, y0 = kw.get('k0', 1.0), kw.get('x0', 0.0), kw.get('y0', 0.0)
    if tag.startswith('merc'):
        return orc.make_proj(orc.PROJ_MERC, a=a, es=es, lat0=0.0, lon0=kw['lon0'], lat_ts=kw['lat_ts'], k0=k0, x0=x0, y0=y0)
    if tag.startswith('lcc'):
        return orc.make_proj(orc.PROJ_LCC, a=a, es=es, lat0=kw['lat0'], lon0=kw['lon0'], lat1=kw['lat1'], lat2=kw['lat2'],
                             k0=k0, x0=x0, y0=y0)
    if tag in ('stere_north', 'stere_north_ts90', 'stere_south'):
        return orc.make_proj(orc.PROJ_STERE_POLAR, a=a, es=es, lat0=kw['lat0'], lon0=kw['lon0'], lat_ts=kw['lat_ts'], k0=k0,
                             x0=x0, y0=y0)
    if tag.startswith('stere'):
        return orc.make_proj(orc.PROJ_STERE_OBLIQUE, a=a, es=es, lat0=kw['lat0'], lon0=kw['lon0'], k0=k0, x0=x0, y0=y0)
    if tag.startswith('laea'):
        return orc.make_proj(orc.PROJ_LAEA, a=a, es=es, lat0=kw['lat0'], lon0=kw['lon0'], x0=x0, y0=y0)
    if tag == 'rotated_pole':
        return orc.make_proj(orc.PROJ_OB_TRAN, lon0=kw['lon0'], lat1=kw['o_lat_p'], lat2=kw['o_lon_p'])
    return orc.make_proj(orc.PROJ_TMERC, a=a, es=es, lat0=kw['lat0'], lon0=kw['lon0'], k0=k0, x0=x0, y0=y0)


def proj4_of(tag, kw):
    ell = ' +a=%r +rf=%r' % (kw['a'], kw['rf']) if 'rf' in kw else ''
    tail = ''.join(' +%s=%r' % (k, kw[k]) for k in ('x_0', 'y_0') if False)
    off = ' +x_0=%r +y_0=%r' % (kw.get('x0', 0.0), kw.get('y0', 0.0))
    if tag.startswith('merc'):
        return '+proj=merc +lon_0=%r +lat_ts=%r%s%s' % (kw['lon0'], kw['lat_ts'], off, ell) + tail
    if tag.startswith('lcc'):
        return '+proj=lcc +lat_0=%r +lon_0=%r +lat_1=%r +lat_2=%r +k_0=%r%s%s' % (kw['lat0'], kw['lon0'], kw['lat1'], kw['lat2'],
                                                                                kw.get('k0', 1.0), off, ell)
    if tag in ('stere_north', 'stere_north_ts90', 'stere_south'):
        return '+proj=stere +lat_0=%r +lon_0=%r +lat_ts=%r +k_0=%r%s%s' % (kw['lat0'], kw['lon0'], kw['lat_ts'], kw.get('k0', 1.0), off, ell)
    if tag.startswith('stere'):
        return '+proj=stere +lat_0=%r +lon_0=%r +k_0=%r%s%s' % (kw['lat0'], kw['lon0'], kw.get('k0', 1.0), off, ell)
    if tag.startswith('laea'):
        return '+proj=laea +lat_0=%r +lon_0=%r%s%s' % (kw['lat0'], kw['lon0'], off, ell)
    if tag == 'rotated_pole':
        return '+proj=ob_tran +o_proj=longlat +lon_0=%r +o_lat_p=%r +o_lon_p=%r' % (kw['lon0'], kw['o_lat_p'], kw['o_lon_p'])
    return '+proj=tmerc +lat_0=%r +lon_0=%r +k_0=%r%s%s' % (kw['lat0'], kw['lon0'], kw.get('k0', 1.0), off, ell)


def _dlon(a, b):
    return (a - b + 180.0) % 360.0 - 180.0


@pytest.mark.parametrize('tag', TAGS)
def test_projection_known_answers_oracle_and_host_mirror(tag):
    g = np.load(GOLDEN)
    kw = ast.literal_eval(str(g[tag + '_kw']))
    lon, lat, x, y = (g['%s_%s' % (tag, k)] for k in ('lon', 'lat', 'x', 'y'))
    p = oracle_proj(tag, kw)
    ox, oy = orc.proj_fwd(p, lon, lat)
    olon, olat = orc.proj_inv(p, x, y)
    H = projection.Proj(proj4_of(tag, kw))
    hx, hy = H(lon, lat)
    hlon, hlat = H(x, y, inverse=True)
    deg = tag == 'rotated_pole'
    scale = 1.0 if not deg else 1.0          # (degrees for the rotated pole: 2e-8 deg would be far too loose -> own bound below)
    fwd = max(np.abs((_dlon(ox, x) if deg else ox - x)).max(), np.abs(oy - y).max())
    fwd_h = max(np.abs((_dlon(hx, x) if deg else hx - x)).max(), np.abs(hy - y).max())
    inv = max(np.abs(_dlon(olon, lon) * np.cos(np.radians(lat))).max(), np.abs(olat - lat).max())
    inv_h = max(np.abs(_dlon(hlon, lon) * np.cos(np.radians(lat))).max(), np.abs(hlat - lat).max())
    print('%-18s forward: oracle %.2e host %.2e %s   inverse: oracle %.2e host %.2e deg' % (tag, fwd, fwd_h, 'deg' if deg else 'm', inv, inv_h))
    tol_f, tol_i = (1e-12 if deg else TOL_FWD * scale), TOL_INV
    if tag == 'laea_north':
        # the polar aspect forms rho = a sqrt(qp - q) (Snyder 24-23; PROJ does the same): a difference of nearly equal numbers
        # next to the pole -- 2e-16 / (2 rho / a) * a = 4e-7 m at 0.1 degrees from it.  Conditioning of the formula, not an error.
        tol_f, tol_i = 1e-5, 2e-11
    assert fwd < tol_f and fwd_h < tol_f
    assert inv < tol_i and inv_h < tol_i


def test_geodesic_inverse_azimuth_known_answers():
    """Geod.inv's forward azimuth, which the reference uses for the vector rotation of projected (10 m lines, variables.py:85-97)
    and rotated-pole readers (0.1-degree lines): the exact azimuth of the geodesic between two float64 points, solved at 40
    digits on the exact direct problem.  The oracle shoots on its float64 direct routine: what it can reach on a line of length s
    is the routine's position accuracy over s -- a few 1e-9 m / s."""
    g = np.load(GOLDEN)
    az, s = orc.geod_inv(g['inv_lon1'], g['inv_lat1'], g['inv_lon2'], g['inv_lat2'])
    d = np.radians(np.abs(_dlon(az, g['inv_azi1'])))
    bound = 2e-8 / g['inv_s12'] + 1e-13
    print('geodesic inverse azimuth: worst %.2e rad (10 m lines: %.2e), worst |ds| %.2e m' %
          (d.max(), d[g['inv_s12'] < 10.5].max(), np.abs(s - g['inv_s12']).max()))
    assert (d < bound).all(), (d / bound).max()
    assert np.abs(s - g['inv_s12']).max() < 2e-8
