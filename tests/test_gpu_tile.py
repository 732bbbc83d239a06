"""k_step_tile (csrc/odr_tile.hip.h): the fused step with the node records of each workgroup's rectangle staged in LDS by
LDS-DMA must give the SAME BITS as k_step_grid, which gathers the same records from the blocks in HBM -- whatever the
rectangle covers: a freshly sorted set (the tile serves everyone), a set that drifted since its sort, elements moved into
holes by the in-place compaction, stage positions several cells away from the element (long time step), an LDS budget
too small for the workgroups' rectangles, steps on / between / across reader time levels.  Both stage arithmetics
(ODR_STAGE_EXACT, ODR_STAGE_FAST), RK2 and RK4, a 3-D lon/lat reader and a 2-D polar-stereographic one.
The reference semantics both kernels implement: readers/interpolation/interpolators.py:105-139 (bilinear footprint),
models/physics_methods.py:611-691 (advect_ocean_current)."""
import numpy as np
import pytest

from opendrift_amd import synthetic as synth
from opendrift_amd.device import Context

pytestmark = pytest.mark.gpu

U, V = 'x_sea_water_velocity', 'y_sea_water_velocity'
W, KZ = 'upward_sea_water_velocity', 'ocean_vertical_diffusivity'
DEPTH, SSH, LAND = 'sea_floor_depth_below_sea_level', 'sea_surface_height', 'land_binary_mask'
XW, YW = 'x_wind', 'y_wind'


def _run(monkeypatch, tile, g, names, proj, lon, lat, z, scheme, math, times, dt, lds=None, resort_every=0, strand=False,
         seafloor=False):
    monkeypatch.setenv('ODR_TILE', '1' if tile else '0')
    monkeypatch.setenv('ODR_TILE_MIN_N', '1')
    if lds:
        monkeypatch.setenv('ODR_TILE_LDS', str(lds))
    else:
        monkeypatch.delenv('ODR_TILE_LDS', raising=False)
    ctx = Context(seed=0)
    ctx.set_stage_math(math)
    sid = ctx.add_grid(g['x'], g['y'], z=g.get('z') if proj is None else None, proj=proj)
    for k in range(3):
        ctx.upload_block(sid, k, float(g['t'][k]), {nm: g[nm][k] for nm in names})
    for nm in names:
        ctx.bind(nm, [sid], {LAND: np.nan, DEPTH: 10000.0}.get(nm, 0.0))
    n = len(lon)
    P = ctx.particles(n)
    P.append(lon, lat, z=z)
    P.store_previous()
    P.sort_by_cell(sid)
    for k, t in enumerate(times):
        if resort_every and k and k % resort_every == 0:
            P.sort_by_cell(sid, keep_environment=False)
        P.env_coast_advect(names, t, scheme, dt, coastline='stranding' if strand else 'previous', stranded_code=1,
                           store_previous=True, count=False, seafloor=seafloor, age_dt=dt)
        if strand:
            P.compact()       # in place: elements of the tail fill the holes, far from their new neighbours
    st = P.tile_stats()
    d = P.download()
    o = np.argsort(d['ID'])
    dd = P.download_deactivated()
    od = np.argsort(dd['ID'])
    out = {k: d[k][o] for k in ('lon', 'lat', 'z', 'status', 'ID')}
    out['env_u'] = P.env_download(U)[o]
    out['dead'] = {k: dd[k][od] for k in ('lon', 'lat', 'ID', 'status')}
    P.close()
    ctx.close()
    return out, st


def _same(a, b):
    for k in ('ID', 'lon', 'lat', 'z', 'status', 'env_u'):
        eq = (a[k] == b[k]) | ((a[k] != a[k]) & (b[k] != b[k]))
        assert eq.all(), (k, int((~eq).sum()), len(eq))
    for k in ('ID', 'lon', 'lat', 'status'):
        assert np.array_equal(a['dead'][k], b['dead'][k]), ('deactivated', k)


def _c3(n=60000, seed=11):
    g = synth.grid3d(nx=128, ny=96, nz=8, nt=3, seed=5)
    rng = np.random.default_rng(seed)
    lon = rng.uniform(g['x'][3], g['x'][-4], n)
    lat = rng.uniform(g['y'][3], g['y'][-4], n)
    z = -rng.uniform(0, 60, n)
    return g, [U, V, W, DEPTH, LAND], lon, lat, z


@pytest.mark.parametrize('math', ['exact', 'fast'])
@pytest.mark.parametrize('scheme', ['runge-kutta', 'runge-kutta4'])
def test_tile_step_equals_global_step_on_a_sorted_set(monkeypatch, math, scheme):
    """Eight steps after one sort (the set drifts away from its sort tiles), on a time level (t = 0), between levels and
    with the full-step stage on the next level (t + dt = 3600); sea floor lift and ages in the launch."""
    g, names, lon, lat, z = _c3()
    times = [0.0, 600.0, 1200.0, 1800.0, 2400.0, 3000.0, 3600.0, 4200.0]
    a, st = _run(monkeypatch, True, g, names, None, lon, lat, z, scheme, math, times, 600.0, seafloor=True)
    b, _ = _run(monkeypatch, False, g, names, None, lon, lat, z, scheme, math, times, 600.0, seafloor=True)
    _same(a, b)
    assert st['launches'] == len(times) and st['ranges'] > 0
    assert st['handed_over'] < 0.02 * len(lon) * len(times), st     # the rectangles serve (nearly) everyone


@pytest.mark.parametrize('math', ['exact', 'fast'])
def test_tile_step_with_stage_positions_far_from_the_element(monkeypatch, math):
    """dt = 50 min: a stage position is up to ~4 cells from the element -- outside the rectangle's one-node margin: those
    samples come from the blocks in HBM, one by one, inside the tile launch."""
    g, names, lon, lat, z = _c3(n=30000, seed=3)
    times = [0.0, 100.0]
    a, st = _run(monkeypatch, True, g, names, None, lon, lat, z, 'runge-kutta4', math, times, 3000.0)
    b, _ = _run(monkeypatch, False, g, names, None, lon, lat, z, 'runge-kutta4', math, times, 3000.0)
    _same(a, b)
    assert st['launches'] == len(times)


@pytest.mark.parametrize('math', ['exact', 'fast'])
def test_tile_step_with_a_small_lds_budget_and_compaction(monkeypatch, math):
    """16 KiB of LDS per workgroup (rectangles cut to the capacity: many elements are handed to k_step_list), stranding
    with in-place compaction between the steps (tail elements fill holes anywhere in the set), a re-sort in between."""
    g, names, lon, lat, z = _c3(n=50000, seed=7)
    z[:] = 0.0
    times = [300.0, 900.0, 1500.0, 2100.0, 2700.0]
    kw = dict(lds=16 * 1024, resort_every=3, strand=True)
    a, st = _run(monkeypatch, True, g, names, None, lon, lat, z, 'runge-kutta4', math, times, 600.0, **kw)
    b, _ = _run(monkeypatch, False, g, names, None, lon, lat, z, 'runge-kutta4', math, times, 600.0, **kw)
    _same(a, b)
    assert st['launches'] == len(times)
    assert len(a['dead']['ID']) > 0                       # something stranded, the set was compacted
    assert st['rectangles_cut'] > 0 and st['handed_over'] > 0, st


@pytest.mark.parametrize('math', ['exact', 'fast'])
def test_tile_step_on_a_polar_stereographic_reader(monkeypatch, math):
    """C4-shaped: 2-D current + wind + land mask on a polar-stereographic grid (vector rotation in the samples)."""
    from opendrift_amd.projection import stere_polar_inverse
    g = synth.grid_stere(nx=260, ny=90, nt=3, seed=0)
    names = [U, V, XW, YW, LAND]
    rng = np.random.default_rng(5)
    n = 40000
    x = rng.uniform(g['x'][4], g['x'][-5], n)
    y = rng.uniform(g['y'][4], g['y'][-5], n)
    lon, lat = stere_polar_inverse(x, y, **synth.NORKYST_PROJ)
    times = [0.0, 900.0, 1800.0, 2700.0]
    a, st = _run(monkeypatch, True, g, names, synth.NORKYST_PROJ, lon, lat, np.zeros(n), 'runge-kutta4', math, times, 900.0,
                 strand=True)
    b, _ = _run(monkeypatch, False, g, names, synth.NORKYST_PROJ, lon, lat, np.zeros(n), 'runge-kutta4', math, times, 900.0,
                strand=True)
    _same(a, b)
    assert st['launches'] == len(times)


def test_tile_step_is_not_used_after_an_append(monkeypatch):
    """Elements appended after the sort are in no workgroup range: the table is invalid until the next sort."""
    g, names, lon, lat, z = _c3(n=20000, seed=9)
    monkeypatch.setenv('ODR_TILE', '1')
    monkeypatch.setenv('ODR_TILE_MIN_N', '1')
    ctx = Context(seed=0)
    sid = ctx.add_grid(g['x'], g['y'], z=g['z'])
    for k in range(3):
        ctx.upload_block(sid, k, float(g['t'][k]), {nm: g[nm][k] for nm in names})
    for nm in names:
        ctx.bind(nm, [sid], {LAND: np.nan, DEPTH: 10000.0}.get(nm, 0.0))
    P = ctx.particles(len(lon) + 100)
    P.append(lon, lat, z=z)
    P.sort_by_cell(sid)
    P.env_coast_advect(names, 0.0, 'runge-kutta4', 600.0, coastline='previous', count=False)
    assert P.tile_stats()['launches'] == 1
    P.append(lon[:100], lat[:100], z=z[:100])
    P.env_coast_advect(names, 600.0, 'runge-kutta4', 600.0, coastline='previous', count=False)
    assert P.tile_stats()['launches'] == 1
    P.sort_by_cell(sid)
    P.env_coast_advect(names, 1200.0, 'runge-kutta4', 600.0, coastline='previous', count=False)
    assert P.tile_stats()['launches'] == 2
    P.close()
    ctx.close()
