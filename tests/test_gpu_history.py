"""Device result buffer (odr_history_*: state_to_buffer on the GPU) against the CPU restatement
oracle/history.py: bit-exact float32 [trajectory, time] arrays, NaN where nothing was written."""
import numpy as np
import pytest

from oracle.history import HistoryOracle

pytestmark = pytest.mark.gpu

U, XW = 'x_sea_water_velocity', 'x_wind'


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_history_matches_oracle_with_deactivation_and_permutation(ctx):
    rng = np.random.default_rng(3)
    n, nt = 50000, 5
    names = ['lon', 'lat', 'z', 'status', 'age_seconds', U, XW, ('property', 1)]
    onames = ['lon', 'lat', 'z', 'status', 'age_seconds', U, XW, 'prop1']
    P = ctx.particles(n + 1000)
    lon, lat = rng.uniform(0, 10, n) + 1e-9, rng.uniform(60, 66, n)
    P.append(lon, lat, z=-rng.uniform(0, 50, n))
    P.set_property(1, rng.standard_normal(n).astype(np.float32))
    H = ctx.history(n + 1000, nt, names)
    O = HistoryOracle(n + 1000, nt, onames)


    def record(tindex, only_deactivated):
        d = P.download()
        vals = {'lon': d['lon'], 'lat': d['lat'], 'z': d['z'], 'status': d['status'],
                'age_seconds': age[d['ID']], U: P.env_download(U), XW: P.env_download(XW), 'prop1': P.get_property(1)}
        H.record(P, tindex, only_deactivated)
        O.record(tindex, d['ID'], d['status'], vals, only_deactivated)

    age = np.zeros(n + 1000, np.float32)
    step = 0
    for tindex in range(nt):
        for sub in range(2):                                     # two calculation steps per output step
            m = len(P)
            P.env_upload(U, rng.standard_normal(m).astype(np.float32))
            P.env_upload(XW, (rng.standard_normal(m) * 5).astype(np.float32))
            P.update_positions(rng.standard_normal(m) * 0.3, rng.standard_normal(m) * 0.3, 600.0)
            kill = rng.uniform(size=m) < 0.03
            P.deactivate(kill, 1 + step % 3)
            record(tindex if sub == 0 else min(tindex + 1, nt - 1), only_deactivated=(sub != 0))
            P.increase_age(600.0)
            age[P.download()['ID']] += np.float32(600.0)
            P.compact()                                          # in place: permutes the survivors
            step += 1
        if tindex == 1:                                          # elements released later get the next IDs
            k = 1000
            P.append(rng.uniform(0, 10, k), rng.uniform(60, 66, k), z=np.zeros(k), id=np.arange(n, n + k, dtype=np.int32))
            P.set_property(1, np.concatenate([P.get_property(1)[:len(P) - k], np.ones(k, np.float32)]))
    H.flush()
    H.wait()
    for v, ov in zip(names, onames):
        a = H.array(v)
        assert a.shape == (n + 1000, nt) and a.dtype == np.float32
        assert _same(a, O.buf[ov]), (v, int((~((a == O.buf[ov]) | (np.isnan(a) & np.isnan(O.buf[ov])))).sum()))
        lo, hi = H.minmax(v)
        olo, ohi = O.minmax(ov)
        assert (lo == olo and hi == ohi), (v, lo, olo, hi, ohi)
    assert np.isnan(H.array('lon')).any() and (H.array('status') > 0).any()
    # partial flush of two time slots, then a new buffer
    H.flush(1, 2)
    H.wait()
    assert _same(H.array('lat'), O.buf['lat'][:, 1:3])
    H.reset()
    H.flush()
    H.wait()
    assert np.isnan(H.array('lon')).all() and np.isnan(H.minmax('lon')[0])
    H.close()


def test_history_errors(ctx):
    from opendrift_amd._abi import OdrError
    P = ctx.particles(8)
    P.append(np.zeros(8), np.zeros(8))
    H = ctx.history(8, 2, ['lon', XW])
    with pytest.raises(OdrError):
        H.record(P, 0)                       # x_wind has not been sampled
    with pytest.raises(ValueError):
        H.record(P, 5)                       # outside the buffer
    with pytest.raises(OdrError):
        H.array('lon')                       # nothing flushed yet
    with pytest.raises(ValueError):
        ctx.history(8, 2, ['lon'] * 41)      # more than the 40 variables of a record
    H.close()
