"""Asynchronous block upload (odr_block_upload_async + odr_block_commit on the upload stream) gives the same
environment as the synchronous upload, the replaced block stays readable until the commit, and pinned host
arrays can be uploaded while the simulation runs."""
import numpy as np
import pytest

from opendrift_amd import synthetic as synth

pytestmark = pytest.mark.gpu

U, V, W = 'x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity'
KZ, DEPTH, LAND = 'ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level', 'land_binary_mask'


def _eq(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_async_upload_equals_sync_and_overlaps(ctx):
    g = synth.grid3d(nx=128, ny=96, nz=8, nt=3, seed=2)
    names = [U, V, W, KZ, DEPTH, LAND]
    sa = ctx.add_grid(g['x'], g['y'], z=g['z'])          # async
    ss = ctx.add_grid(g['x'], g['y'], z=g['z'])          # sync
    rng = np.random.default_rng(0)
    n = 30000
    lon, lat, z = rng.uniform(g['x'][2], g['x'][-3], n), rng.uniform(g['y'][2], g['y'][-3], n), -rng.uniform(0, 80, n)
    P = ctx.particles(n)
    P.append(lon, lat, z=z)
    pinned = {k: ctx.pin(np.ascontiguousarray(g[k])) for k in names}     # page-locked once, uploaded level by level

    def sample(sid, t):
        for k in names:
            ctx.bind(k, [sid], {LAND: np.nan, DEPTH: 10000.0}.get(k, 0.0))
        return P.env_sample(names, t, download=True)

    for slot in (0, 1):
        ctx.upload_block(ss, slot, float(g['t'][slot]), {k: g[k][slot] for k in names})
        ctx.upload_block_async(sa, slot, float(g['t'][slot]), {k: pinned[k][slot] for k in names})
        ctx.commit_block(sa, slot)
    a, b = sample(sa, 1000.0), sample(ss, 1000.0)
    for k in names:
        assert _eq(a[k], b[k]), k
    # stage level 2 into slot 0 while kernels that read the old slot 0 are running; the old content stays in force
    ctx.upload_block_async(sa, 0, float(g['t'][2]), {k: pinned[k][2] for k in names})
    before = sample(sa, 1000.0)
    for k in names:
        assert _eq(before[k], b[k]), k                   # not committed yet: still levels 0 and 1
    P.advect('runge-kutta4', 1000.0, 600.0)              # compute stream busy while the upload stream works
    ctx.commit_block(sa, 0)
    ctx.upload_block(ss, 0, float(g['t'][2]), {k: g[k][2] for k in names})
    t = float(g['t'][1]) + 700.0
    a, b = sample(sa, t), sample(ss, t)
    for k in names:
        assert _eq(a[k], b[k]), k
    for k in names:
        ctx.unpin(pinned[k])


def test_commit_without_staging_is_an_error(ctx):
    from opendrift_amd._abi import OdrError
    g = synth.grid3d(nx=32, ny=24, nz=4, nt=1, seed=1)
    sid = ctx.add_grid(g['x'], g['y'], z=g['z'])
    with pytest.raises(OdrError):
        ctx.commit_block(sid, 0)


def test_model_run_with_prefetch_equals_run_without():
    """run(): the next reader time level is staged on the upload stream while the steps of the current one execute
    (DeviceReaderBinding prefetch); trajectories must not depend on it."""
    from datetime import datetime, timedelta
    from opendrift_amd import readers
    from opendrift_amd.oceandrift import OceanDrift
    g = synth.grid3d(nx=96, ny=80, nz=6, nt=6, seed=4)
    T0 = datetime(2020, 1, 1)
    times = [T0 + timedelta(seconds=float(t)) for t in g['t']]
    names = [U, V, W, DEPTH, LAND]

    def run(prefetch):
        readers.DeviceReaderBinding.PREFETCH = prefetch
        o = OceanDrift(loglevel=50, seed=1)
        o.add_reader(readers.GridReader(g['x'], g['y'], times, {k: np.ascontiguousarray(g[k]) for k in names}, z=g['z']))
        o.set_config('drift:advection_scheme', 'runge-kutta4')
        o.set_config('general:coastline_action', 'previous')
        r = np.random.default_rng(0)
        n = 20000
        o.seed_elements(lon=r.uniform(g['x'][5], g['x'][-6], n), lat=r.uniform(g['y'][5], g['y'][-6], n),
                        z=-r.uniform(0, 30, n), time=T0)
        nsteps = int((g['t'][-1] - g['t'][0]) // 600) - 1
        o.run(time_step=600, steps=nsteps)
        e = o.elements
        k = np.argsort(e.ID)
        staged = sum(len(b.staged) + len(b.slots) for b in o.readers.values())
        assert all(b.prefetch == prefetch for b in o.readers.values())
        return e.lon[k], e.lat[k], e.z[k], staged

    try:
        a, b = run(True), run(False)
    finally:
        readers.DeviceReaderBinding.PREFETCH = True
    for p, q in zip(a[:3], b[:3]):
        assert _eq(p, q)
    assert a[3] >= 2


@pytest.mark.parametrize('nx', [33, 34, 36])
def test_row_dilation_sweeps_equal_the_plain_ones(ctx, monkeypatch, nx):
    """The ten grey_dilation sweeps of a new block (expand_numpy_array, interpolators.py:9-20): the row kernels that only
    look at the cells still NaN in their ping-pong target (k_blk_dilate_row, 4 / 2 / 1 cells per thread by the parity
    of nx) against the plain sweeps (ODR_PLAIN_DILATE): the same samples bit for bit, on fields with NaN blobs wider
    than ten cells (so that NaN survives), NaN on the block edges and NaN-free variables."""
    ny, nz = 29, 5
    rng = np.random.default_rng(nx)
    x, y, z = np.linspace(3.0, 5.0, nx), np.linspace(59.0, 61.0, ny), -np.linspace(0, 40, nz)
    Y, X = np.meshgrid(np.arange(ny), np.arange(nx), indexing='ij')
    blob = ((X - 8) ** 2 + (Y - 9) ** 2 < 36) | ((X - nx + 3) ** 2 + (Y - 20) ** 2 < 150) | (rng.uniform(size=(ny, nx)) < 0.08)
    blob[0, :5] = True
    blob[-1, -4:] = True
    fields = {}
    for k, nm in enumerate((U, V, KZ)):
        f = rng.normal(size=(nz, ny, nx)).astype(np.float32)
        if nm != KZ:
            f[:, blob] = np.nan
            f[3:, (X + Y) % 7 == 0] = np.nan          # deeper layers: more NaN (fill towards the sea floor first)
        fields[nm] = f
    fields[DEPTH] = np.where(blob, np.nan, 100.0).astype(np.float32)
    names = [U, V, KZ, DEPTH]
    n = 40000
    lon, lat, zz = rng.uniform(x[0], x[-1], n), rng.uniform(y[0], y[-1], n), -rng.uniform(0, 40, n)
    P = ctx.particles(n)
    P.append(lon, lat, z=zz)
    res = []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv('ODR_PLAIN_DILATE', '1')
        else:
            monkeypatch.delenv('ODR_PLAIN_DILATE', raising=False)
        sid = ctx.add_grid(x, y, z=z)
        ctx.upload_block(sid, 0, 0.0, fields)
        for nm in names:
            ctx.bind(nm, [sid], np.nan)
        res.append(P.env_sample(names, 0.0, download=True))
    for nm in names:
        assert _eq(res[0][nm], res[1][nm]), nm
    assert np.isnan(res[0][U]).any() and np.isfinite(res[0][U]).sum() > n // 2


@pytest.mark.parametrize('nx,ny,old', [(150, 101, 'ODR_PLAIN_DILATE'), (89, 133, 'ODR_ROW_DILATE'), (44, 45, 'ODR_PLAIN_DILATE')])
def test_tiled_block_preparation_equals_the_sweeps(ctx, monkeypatch, nx, ny, old):
    """mask + sea-floor fill + ten grey_dilation sweeps of a new block (ReaderBlock.__init__, expand_numpy_array):
    k_blk_mask_fill + k_blk_dilate_tile (44x44-cell tiles with a 10-cell halo in LDS, flagged tiles only) + the merging
    record writer against the whole-variable sweeps of rounds 1-2 -- the same samples bit for bit on grids of several
    tiles with NaN regions deeper than ten cells, NaN across tile seams and on the block edges, values beyond 1e9, inf,
    a layer that is NaN everywhere, a NaN-free variable, a 2-D variable and the (undilated) land mask."""
    nz = 4
    rng = np.random.default_rng(nx)
    x, y, z = np.linspace(3.0, 5.0, nx), np.linspace(59.0, 61.0, ny), -np.linspace(0, 30, nz)
    Y, X = np.meshgrid(np.arange(ny), np.arange(nx), indexing='ij')
    blob = ((X - 44) ** 2 + (Y - 44) ** 2 < 300) | ((X - nx + 3) ** 2 + (Y - 20) ** 2 < 150) | (rng.uniform(size=(ny, nx)) < 0.05) \
        | ((X > 80) & (X < 97) & (Y > 30))
    blob[0, :5] = True
    blob[-1, -4:] = True
    blob[:, 43:45] |= (Y[:, 43:45] % 3 == 0)           # a dotted line along a tile seam
    fields = {}
    for k, nm in enumerate((U, V, KZ)):
        f = rng.normal(size=(nz, ny, nx)).astype(np.float32)
        if nm != KZ:
            f[:, blob] = np.nan
            f[2:, (X + Y) % 7 == 0] = np.nan
            f[1, (X * Y) % 11 == 0] = 3e9 if nm == U else np.inf
        fields[nm] = f
    fields[V][0] = np.nan                                # a whole layer without a value (sea-floor fill leaves layer 0 alone)
    fields[DEPTH] = np.where(blob, np.nan, 100.0 + X).astype(np.float32)
    fields[LAND] = blob.astype(np.float32)
    names = [U, V, KZ, DEPTH, LAND]
    n = 60000
    lon, lat, zz = rng.uniform(x[0], x[-1], n), rng.uniform(y[0], y[-1], n), -rng.uniform(0, 30, n)
    P = ctx.particles(n)
    P.append(lon, lat, z=zz)
    res = []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv(old, '1')
        sid = ctx.add_grid(x, y, z=z)
        ctx.upload_block(sid, 0, 0.0, {k: v.copy() for k, v in fields.items()})
        for nm in names:
            ctx.bind(nm, [sid], np.nan)
        res.append(P.env_sample(names, 0.0, download=True))
    for nm in names:
        assert _eq(res[0][nm], res[1][nm]), nm
    assert np.isnan(res[0][U]).any() and np.isfinite(res[0][U]).sum() > n // 2


def test_variables_identical_at_both_time_levels_are_gathered_once(ctx, monkeypatch):
    """odr_block_set_content_ids (ids by value comparison here: device.ContentIds; the reader bindings use a reader's
    `static_variables`): sea floor depth and land mask that the reader hands out unchanged with every block are read at ONE
    of the two bracketing levels (G.ps_static) -- the same samples bit for bit as with the skip disabled; a depth that DOES
    change between the levels gets a new id per level and keeps both gathers (and its time interpolation)."""
    from opendrift_amd.device import ContentIds
    g = synth.grid3d(nx=64, ny=48, nz=6, nt=3, seed=2)
    names = [U, V, W, DEPTH, LAND]
    rng = np.random.default_rng(9)
    n = 30000
    lon, lat, zz = rng.uniform(g['x'][2], g['x'][-3], n), rng.uniform(g['y'][2], g['y'][-3], n), -rng.uniform(0, 40, n)
    P = ctx.particles(n)
    P.append(lon, lat, z=zz)
    out = {}
    for moving_floor in (False, True):
        for skip in (True, False):
            if skip:
                monkeypatch.delenv('ODR_NO_STATIC_SKIP', raising=False)
            else:
                monkeypatch.setenv('ODR_NO_STATIC_SKIP', '1')
            sid = ctx.add_grid(g['x'], g['y'], z=g['z'])
            cids = ContentIds()
            for k in range(3):
                f = {nm: g[nm][k] for nm in names}
                if moving_floor:
                    f[DEPTH] = (g[DEPTH][k] + 7.0 * k).astype(np.float32)
                ids = cids.assign(names, f)
                assert ids[LAND] != 0 and ids[U] == 0 and (k == 0 or (ids[DEPTH] == first[DEPTH]) == (not moving_floor))
                first = ids if k == 0 else first
                ctx.upload_block(sid, k, float(g['t'][k]), f, content_ids=ids)
            for nm in names:
                ctx.bind(nm, [sid], np.nan)
            out[moving_floor, skip] = P.env_sample(names, float(g['t'][0]) + 1234.5, download=True)
    for mf in (False, True):
        for nm in names:
            assert _eq(out[mf, True][nm], out[mf, False][nm]), (mf, nm)
    assert not _eq(out[False, True][DEPTH], out[True, True][DEPTH])
    d = out[True, True][DEPTH] - out[False, True][DEPTH]
    assert np.nanmax(np.abs(d - 7.0 * 1234.5 / 3600.0)) < 1e-3          # interpolated in time between the two floors
