"""GPU: error behaviour of the C ABI as seen through the ctypes binding -- bad arguments raise ValueError
(ODR_ERR_INVALID), wrong call order / exhausted capacity raise OdrError, like the reference raises
ValueError / WrongMode for bad config and ordering."""
import numpy as np
import pytest

from opendrift_amd._abi import OdrError

pytestmark = pytest.mark.gpu
U, V = 'x_sea_water_velocity', 'y_sea_water_velocity'


def test_capacity_and_ordering_errors(ctx):
    P = ctx.particles(10)
    with pytest.raises(OdrError):                       # capacity
        P.append(np.zeros(11), np.zeros(11))
    P.append(np.linspace(3, 4, 10), np.full(10, 60.0))
    with pytest.raises(OdrError):                       # advect before the current has been sampled
        P.advect('euler', 0.0, 60.0)
    with pytest.raises(ValueError):
        P.advect('runge-kutta5', 0.0, 60.0)             # "Drift scheme not recognised" (physics_methods.py:689-691)
    with pytest.raises(OdrError):
        P.advect_wind(60.0)
    with pytest.raises(OdrError):
        P.vmix(0.0, 600.0, 60.0)
    with pytest.raises(OdrError):
        P.coastline('stranding')
    with pytest.raises(OdrError):
        P.leeway(600.0)
    assert P.coastline('none') == 0                     # action 'none' needs nothing


def test_bad_source_arguments(ctx):
    with pytest.raises(ValueError):
        ctx.bind(U, [3], 0.0)                           # unknown source
    sid = ctx.add_constant({U: 0.1, V: 0.2})
    with pytest.raises(ValueError):
        ctx.drop_block(sid, 9)                          # slot out of range
    x, y = np.linspace(0, 1, 8, dtype=np.float32), np.linspace(60, 61, 6, dtype=np.float32)
    g = ctx.add_grid(x, y, z=np.array([0., -10., -20.]))
    with pytest.raises(ValueError):                     # 4 levels in a 3-level source
        ctx.upload_block(g, 0, 0.0, {U: np.zeros((4, 6, 8), np.float32)})
    with pytest.raises(ValueError):
        ctx.upload_block(g, 7, 0.0, {U: np.zeros((3, 6, 8), np.float32)})   # slot out of range
    for k in range(16 - 2):
        ctx.add_constant({U: 0.0})
    with pytest.raises(OdrError):                       # at most 16 sources (MAXSRC)
        ctx.add_constant({U: 0.0})


def test_empty_particle_set_is_harmless(ctx):
    sid = ctx.add_constant({U: 0.1, V: 0.2})
    ctx.bind(U, [sid], 0.0)
    ctx.bind(V, [sid], 0.0)
    P = ctx.particles(4)
    P.env_sample([U, V], 0.0)
    P.advect('runge-kutta4', 0.0, 60.0)
    assert P.compact() == 0 and len(P) == 0
    assert P.reduce_scalars()['n_active'] == 0
    P.append([4.0], [60.0])
    P.env_sample([U, V], 0.0)
    P.advect('runge-kutta4', 0.0, 60.0)
    d = P.download()
    assert d['lon'][0] > 4.0 and d['lat'][0] > 60.0
