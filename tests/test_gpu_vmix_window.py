"""odr_vmix on reader profiles runs k_vmix_win: the diffusivity of the five levels around the element's starting level in
registers, and a second loop that fetches three levels per sub-step for an element whose random walk left them
(csrc/odr_kernels.hip.h).  It makes the same float operations in the same order as the whole-column kernel
(ODR_VMIX_WINDOW=0, the round 1-3 path, itself checked against the oracle in test_gpu_parity / test_gpu_diffusivity):
bit-identical z / status / moving / lon / lat -- on and between reader time levels, uniform and stretched levels, 3 to
40 levels, with the sea floor in reach (all seafloor actions), surface elements, host uniforms, with the vertical
advection folded in, and with a diffusivity that sends nearly every element out of its window.
The device's own random stream (ODR_RNG_DEVICE) is pinned too: a device-mode run equals a host-mode run fed with
oracle/philox.py's restatement of Philox4x32-10 and of the block / word layout, bit for bit."""
import numpy as np
import pytest

from opendrift_amd.device import Context
from oracle import philox

pytestmark = pytest.mark.gpu

U, V = 'x_sea_water_velocity', 'y_sea_water_velocity'
W, KZ = 'upward_sea_water_velocity', 'ocean_vertical_diffusivity'
DEPTH, SSH, LAND = 'sea_floor_depth_below_sea_level', 'sea_surface_height', 'land_binary_mask'


def _field(nz, seed, stretched=False, shallow=False):
    """lon/lat grid with nz levels (any number), three time levels an hour apart: smooth K with a subsurface maximum."""
    nx, ny, nt = 64, 48, 3
    rng = np.random.default_rng(seed)
    x = np.linspace(0.0, 3.0, nx).astype(np.float32)
    y = np.linspace(60.0, 62.0, ny).astype(np.float32)
    if stretched:
        z = -np.concatenate([[0.0], np.cumsum(np.linspace(2.0, 14.0, nz - 1))])
    else:
        z = -6.0 * np.arange(nz, dtype=np.float64)
    X, Y = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    g = dict(x=x, y=y, z=z, t=3600.0 * np.arange(nt))
    K = np.empty((nt, nz, ny, nx), np.float32)
    w = np.empty_like(K)
    for it in range(nt):
        for k in range(nz):
            K[it, k] = (2e-2 * (0.2 + (-z[k] / 25.0)) * np.exp(z[k] / 25.0) * (1 + 0.5 * np.sin(3 * X + 2 * Y + 0.3 * it)) + 1e-5)
            w[it, k] = 1e-3 * np.sin(2 * np.pi * X) * np.sin(2 * np.pi * Y) * np.exp(z[k] / 100.0)
    K += (1e-4 * rng.random(K.shape)).astype(np.float32)
    g[KZ] = K
    g[W] = w
    g[U] = np.zeros_like(K)
    g[V] = np.zeros_like(K)
    depth = (30 + 400 * (0.5 + 0.5 * np.sin(2 * X + 1.0) * np.cos(1.5 * Y))).astype(np.float32)
    if shallow:
        depth = np.minimum(depth, 25.0 + 60.0 * np.linspace(0, 1, nx)[None, :]).astype(np.float32)
    g[DEPTH] = np.broadcast_to(depth, (nt, ny, nx)).copy()
    return g


def _mix(monkeypatch, g, window, n, times, seed, dt=600.0, dt_mix=60.0, vadv=None, seafloor=None, uniforms=None,
         mix_at_surface=False, kz_scale=1.0, step0=0):
    monkeypatch.setenv('ODR_VMIX_WINDOW', '1' if window else '0')
    names = [U, V, W, KZ, DEPTH]
    ctx = Context(seed=seed)
    sid = ctx.add_grid(g['x'], g['y'], z=g['z'])
    for k in range(3):
        blk = {nm: g[nm][k] for nm in names}
        blk[KZ] = (blk[KZ] * kz_scale).astype(np.float32)
        ctx.upload_block(sid, k, float(g['t'][k]), blk)
    for nm in names:
        ctx.bind(nm, [sid], {DEPTH: 10000.0}.get(nm, 0.0))
    ctx.bind(SSH, [], 0.0)
    if seafloor:
        ctx.set_seafloor_action(seafloor, 7)
    rng = np.random.default_rng(seed + 1)
    lon = rng.uniform(g['x'][2], g['x'][-3], n)
    lat = rng.uniform(g['y'][2], g['y'][-3], n)
    zmax = float(-g['z'][-1]) * 1.1
    z = -rng.uniform(0, zmax, n)
    z[: n // 16] = 0.0
    P = ctx.particles(n)
    P.append(lon, lat, z=z, terminal_velocity=np.where(np.arange(n) % 3 == 0, -0.004, 0.002).astype(np.float32))
    ids = P.download()['ID']
    for k, t in enumerate(times):
        P.env_sample([U, V, W, DEPTH, SSH], t)
        P.store_previous()
        u = None
        if uniforms == 'philox':      # the device stream, drawn on the host
            u = philox.mixing_uniforms(seed, ids, k + step0, abs(int(dt / dt_mix)))
        elif uniforms is not None:
            u = np.random.default_rng(100 + k).uniform(0, 1, (abs(int(dt / dt_mix)), n))
        P.vmix(t, dt, dt_mix, mix_at_surface=mix_at_surface, step=k + step0, uniforms=u, fuse_vertical_advection=vadv)
    d = P.download()
    o = np.argsort(d['ID'])
    res = {q: d[q][o] for q in ('lon', 'lat', 'z', 'status', 'moving', 'ID')}
    P.close()
    ctx.close()
    res['z0'] = z
    return res


def _compare(monkeypatch, g, **kw):
    a = _mix(monkeypatch, g, True, **kw)
    b = _mix(monkeypatch, g, False, **kw)
    for q in a:
        assert np.array_equal(a[q], b[q], equal_nan=True), q
    return a


@pytest.mark.parametrize('nz,stretched', [(12, False), (12, True), (3, False), (5, True), (16, False), (40, True)])
def test_window_plus_list_equals_the_column_kernel(monkeypatch, nz, stretched):
    g = _field(nz, seed=7 + nz, stretched=stretched)
    t0, t1 = float(g['t'][0]), float(g['t'][1])
    a = _compare(monkeypatch, g, n=40000, seed=3, times=(t0, t0 + 0.37 * (t1 - t0), t1, t1 + 600.0))
    assert (a['z'] <= 0).all() and np.ptp(a['z']) > 1.0 and np.abs(a['z'] - a['z0']).max() > 1.0


def test_strong_mixing_sends_most_elements_out_of_their_window(monkeypatch):
    """Diffusivity x 30: a sub-step spans several levels, nearly every element leaves its window -- the second loop does
    most of the sub-steps and the result is still the column kernel's."""
    g = _field(12, seed=2)
    a = _compare(monkeypatch, g, n=20000, seed=5, times=(0.0, 600.0), kz_scale=30.0)
    assert np.median(np.abs(a['z'] - a['z0'])) > 6.0      # more than a level spacing


@pytest.mark.parametrize('action', ['lift_to_seafloor', 'deactivate', 'previous'])
@pytest.mark.parametrize('vadv', [None, True, False])
def test_sea_floor_and_vertical_advection(monkeypatch, action, vadv):
    g = _field(8, seed=11, shallow=True)
    a = _compare(monkeypatch, g, n=30000, seed=9, times=(0.0, 900.0, 3600.0), seafloor=action, vadv=vadv)
    if action == 'deactivate':
        assert (a['status'] == 7).any() and (a['moving'] == 0).any()


def test_host_uniforms_and_mixing_at_the_surface(monkeypatch):
    g = _field(12, seed=4)
    _compare(monkeypatch, g, n=5000, seed=1, times=(0.0, 1800.0), uniforms=True, mix_at_surface=True)
    _compare(monkeypatch, g, n=5000, seed=1, times=(0.0, 1800.0), uniforms=True, dt=-600.0)


@pytest.mark.parametrize('window', [True, False])
def test_device_stream_equals_the_oracle_philox_handed_over_as_host_uniforms(monkeypatch, window):
    """ODR_RNG_DEVICE draws block it // 5 of Philox4x32-10 with counter {block, step, ID, tag} and key = seed, 24 bits per
    sub-step; oracle/philox.py restates that on the host.  Feeding its numbers through ODR_RNG_HOST gives the same bits --
    12 sub-steps (three blocks), a seed and a step above 2^32, in both kernels."""
    g = _field(12, seed=31)
    kw = dict(n=6000, seed=(7 << 40) + 12345, times=(0.0, 600.0, 1200.0), dt=600.0, dt_mix=50.0, step0=(1 << 33) + 5)
    dev = _mix(monkeypatch, g, window, **kw)
    host = _mix(monkeypatch, g, window, uniforms='philox', **kw)
    for q in dev:
        assert np.array_equal(dev[q], host[q]), q
    assert np.abs(dev['z'] - dev['z0']).max() > 1.0
