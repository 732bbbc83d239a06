"""odr_source_release: a source that is no longer used gives its id back -- a gridded reader whose blocks are re-cut to a
new window (DeviceReaderBinding.recut) registers a new device source every time, and the context holds 16 at a time."""
import numpy as np
import pytest

from opendrift_amd import synthetic as synth
from opendrift_amd.device import Context

pytestmark = pytest.mark.gpu
U, V = 'x_sea_water_velocity', 'y_sea_water_velocity'


def test_released_source_ids_are_reused_and_leave_the_priority_lists():
    g = synth.grid3d(nx=40, ny=32, nz=4, nt=3, seed=1)
    ctx = Context(seed=0)
    n = 500
    rng = np.random.default_rng(0)
    lon, lat = rng.uniform(g['x'][2], g['x'][-3], n), rng.uniform(g['y'][2], g['y'][-3], n)
    P = ctx.particles(n)
    P.append(lon, lat, z=-rng.uniform(0, 5, n))
    seen, want = set(), None
    for k in range(40):       # far more re-registrations than the 16 sources a context holds
        sid = ctx.add_grid(g['x'], g['y'], z=g['z'])
        seen.add(sid)
        ctx.upload_block(sid, 0, 0.0, {U: g[U][0], V: g[V][0]})
        ctx.bind(U, [sid], 0.0)
        ctx.bind(V, [sid], 0.0)
        u = P.env_sample([U, V], 0.0, download=True)[U]
        if want is None:
            want = u.copy()
        assert np.array_equal(u, want)              # the re-registered source serves the same values
        ctx.release_source(sid)
        assert (P.env_sample([U, V], 0.0, download=True)[U] == 0).all()     # released: out of the lists, the fallback serves
    assert len(seen) <= 2
    P.close()
    ctx.close()
