"""Device sigma -> z regridding (odr_sgrid_*) against the reference's own roppy output (golden vectors) and the
restatement oracle/roms.py: float64 results bit for bit; the float32 block it feeds gives the same environment as
the host-regridded block."""
import numpy as np
import pytest

from conftest import golden
from oracle import roms
from opendrift_amd.device import SigmaGrid

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_device_regridding_equals_reference_golden(ctx):
    g = golden('roms_sigma2z.npz')
    for vt in (1, 2):
        sg = SigmaGrid(ctx, g['H'], float(g['Hc']), g['Cs'], zeta=g['zeta'], Vtransform=vt)
        assert _same(sg.z_rho(), g['zrho_vt%d' % vt]), vt
        _, R = sg.zslice(g['F_vt%d' % vt], g['Z'], want_float64=True)
        assert _same(R, g['R_vt%d' % vt]), vt
        _, R64 = sg.zslice(g['F_vt%d' % vt].astype(np.float64), g['Z'], want_float64=True)   # float64 input path
        assert _same(R64, g['R_vt%d' % vt]), vt
        sg.close()


def test_device_regridding_equals_oracle_and_feeds_the_block(ctx):
    from opendrift_amd import synthetic as synth
    from gen_helpers import stretching
    rng = np.random.default_rng(5)
    N, ny, nx = 20, 96, 128
    H = rng.uniform(15.0, 600.0, (ny, nx))
    Cs = stretching(N)
    Z = np.array([0, -1, -3, -5, -10, -25, -50, -75, -100, -150, -200, -300], float)
    sg = SigmaGrid(ctx, H, 15.0, Cs, Vtransform=2)
    zr = roms.z_rho(H, None, 15.0, Cs, Vtransform=2)
    assert _same(sg.z_rho(), zr)
    U, V = 'x_sea_water_velocity', 'y_sea_water_velocity'
    fields = {U: (rng.standard_normal((N, ny, nx)) * 0.3).astype(np.float32),
              V: (rng.standard_normal((N, ny, nx)) * 0.3).astype(np.float32)}
    fields[U][:, 40:50, 60:70] = np.nan
    fields[V][:, 40:50, 60:70] = np.nan
    x, y = np.linspace(2, 6, nx), np.linspace(60, 63, ny)
    sid_d = ctx.add_grid(x, y, z=Z)
    sid_h = ctx.add_grid(x, y, z=Z)
    ptrs, host = {}, {}
    for slot, (k, f) in enumerate(fields.items()):
        p, R = sg.zslice(f, Z, want_float64=True, slot=slot)
        assert _same(R, roms.zslice(f, zr, Z)), k
        ptrs[k], host[k] = p, R.astype(np.float32)
    ctx.upload_block_device(sid_d, 0, 0.0, ptrs, {k: len(Z) for k in ptrs})      # device -> device
    ctx.upload_block(sid_h, 0, 0.0, host)                                        # host-regridded
    n = 20000
    lon, lat, z = rng.uniform(2.1, 5.9, n), rng.uniform(60.1, 62.9, n), -rng.uniform(0, 250, n)
    out = []
    for sid in (sid_d, sid_h):
        ctx.bind(U, [sid], 0.0)
        ctx.bind(V, [sid], 0.0)
        P = ctx.particles(n)
        P.append(lon, lat, z=z)
        out.append(P.env_sample([U, V], 0.0, download=True))
    for k in (U, V):
        assert _same(out[0][k], out[1][k]), k
    sg.close()


def test_sgrid_errors(ctx):
    with pytest.raises(ValueError):
        SigmaGrid(ctx, np.ones((4, 4)), 10.0, np.linspace(-1, 0, 5), Vtransform=3)


def test_model_run_with_sigma_reader_equals_z_level_reader():
    """OceanDrift on a ROMS-type reader (s-levels regridded on the device per time level) = the same run on a
    z-level reader holding the reference-restated regridding (oracle/roms.py) of the same fields."""
    from datetime import datetime, timedelta
    from gen_helpers import stretching
    from opendrift_amd import readers
    from opendrift_amd.oceandrift import OceanDrift
    rng = np.random.default_rng(11)
    N, ny, nx, nt = 16, 60, 72, 3
    x, y = np.linspace(3, 5, nx), np.linspace(60, 61, ny)
    H = 40.0 + 200.0 * rng.uniform(size=(ny, nx))
    Cs = stretching(N)
    T0 = datetime(2020, 1, 1)
    times = [T0 + timedelta(hours=k) for k in range(nt)]
    U, V, LAND = 'x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask'
    s3 = {U: (rng.standard_normal((nt, N, ny, nx)) * 0.2 + 0.3).astype(np.float32),
          V: (rng.standard_normal((nt, N, ny, nx)) * 0.2).astype(np.float32)}
    land = np.zeros((nt, ny, nx), np.float32)
    sr = readers.SigmaGridReader(x, y, times, s3, {LAND: land}, H, 10.0, Cs, Vtransform=2)
    zr = roms.z_rho(H, None, 10.0, Cs, Vtransform=2)
    z3 = {k: np.stack([roms.zslice(a[it], zr, sr.z).astype(np.float32) for it in range(nt)]) for k, a in s3.items()}
    zreader = readers.GridReader(x, y, times, dict(z3, **{LAND: land}), z=sr.z)

    def run(reader):
        o = OceanDrift(loglevel=50, seed=3)
        o.add_reader(reader)
        o.set_config('drift:advection_scheme', 'runge-kutta4')
        o.set_config('drift:vertical_mixing', False)
        n = 5000
        r = np.random.default_rng(2)
        o.seed_elements(lon=r.uniform(3.3, 4.7, n), lat=r.uniform(60.2, 60.8, n), z=-r.uniform(0, 30, n), time=T0)
        o.run(time_step=600, steps=9)
        e = o.elements
        order = np.argsort(e.ID)
        return e.lon[order], e.lat[order], e.z[order]

    a, b = run(sr), run(zreader)
    for p, q in zip(a, b):
        assert np.array_equal(p, q)
    assert np.abs(a[0] - np.sort(a[0])).max() >= 0    # (ran)
