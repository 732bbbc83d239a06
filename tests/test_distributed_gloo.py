"""CPU, world_size 2, gloo: the N>1 plumbing of the particle-sharded path (opendrift_amd.distributed):
block broadcast from the rank that owns the host Reader, shard partition, scalar all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from opendrift_amd import distributed as D, synthetic
    r, lr, w = D.init(backend='gloo')
    assert (r, w) == (rank, world)
    g = synthetic.grid3d(nx=24, ny=16, nz=4, nt=2, seed=3)
    names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask']
    arrays = {k: g[k][0] for k in names} if rank == 0 else None      # only rank 0 "reads" the block
    tens = D.broadcast_block(arrays, src=0)
    ok = all(np.array_equal(tens[k].numpy(), g[k][0], equal_nan=True) for k in names)
    lo, hi = D.shard_range(1001, rank, world)
    tot = D.allreduce_scalars([hi - lo, float(rank)], 'sum')
    mx = D.allreduce_scalars([float(lo)], 'max')
    D.barrier()
    q.put((rank, ok, lo, hi, tot.tolist(), mx.tolist()))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_broadcast_shard_allreduce_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, ok0, lo0, hi0, tot0, mx0), (r1, ok1, lo1, hi1, tot1, mx1) = res
    assert ok0 and ok1                                   # every rank holds the identical block
    assert (lo0, hi0, lo1, hi1) == (0, 501, 501, 1001)   # contiguous, exhaustive, disjoint
    assert tot0 == tot1 == [1001.0, 1.0] and mx0 == mx1 == [501.0]


def test_shard_range_properties():
    from opendrift_amd.distributed import shard_range
    for n in (0, 1, 7, 1000, 12345677):
        for w in (1, 2, 3, 8):
            edges = [shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[k][1] == edges[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def _combine_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from opendrift_amd import distributed as D
    D.init(backend='gloo')
    # raw reduction slots of one rank (odr_reduce_local): counts in slots 0 and 11, maxima elsewhere (minima negated)
    raw = np.full(16, -np.inf)
    raw[0], raw[11] = 100.0 + rank, 7.0 * (rank + 1)
    raw[1], raw[2] = -(4.0 + rank), 9.0 - rank           # lon_min (negated), lon_max
    raw[7] = 0.0 if rank == 0 else 12.5                  # D.max(): only the other rank has diffusivity
    got = D.combine_reductions(raw)
    # a reader block read by rank 0 only
    blk = dict(x=np.arange(5, dtype=np.float32), y=np.arange(4, dtype=np.float32), z=None, time=None,
               u=np.arange(20, dtype=np.float32).reshape(4, 5)) if rank == 0 else None
    meta, tens, works = D.broadcast_reader_block(blk, ['u'], content_ids={'u': 7} if rank == 0 else None)
    assert works == [] and tens['__cid__'].tolist() == [7]     # the level's content ids arrive with it (device.ContentIds)
    q.put((rank, got.tolist(), meta['x'].tolist(), tens['u'].numpy().tolist()))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_reduction_combine_and_reader_block_broadcast_gloo_world2():
    """What makes a sharded run independent of the number of ranks: counts summed, maxima maximised (the early-outs of
    advect_wind / stokes_drift / horizontal_diffusion and MLD.max() see all elements); the reader block of the rank
    that reads arrives on every rank with its coordinate metadata."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_combine_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, got, x, u in res:
        assert got[0] == 201.0 and got[11] == 21.0 and got[1] == -4.0 and got[2] == 9.0 and got[7] == 12.5
        assert got[3] == -np.inf
        assert x == [0, 1, 2, 3, 4] and u[3][4] == 19.0


def _worker_async_summary(rank, world, port, q):
    """The step's collective started and finished in two halves, one rank blocking (a rank whose elements carry a new reason
    waits for the category at once, OceanDrift.run()) and the other finishing it later: same rows on both."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from opendrift_amd import distributed as D
    D.init(backend='gloo')
    out = []
    for step in range(4):
        row = np.array([100.0 * rank + step, float(rank == 1 and step == 2)] + [0.5 * rank] * 23)
        if rank == 0 or step % 2:
            h = D.start_allgather_vector(row)
            row[:] = -1.0                      # the caller's buffer may change while the collective travels
            busy = sum(k * k for k in range(20000))   # (what the loop does in between)
            rows = D.finish_allgather_vector(h)
        else:
            rows = D.allgather_vector(row)
        out.append(rows.tolist())
    D.barrier()
    q.put((rank, out))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_step_summary_in_two_halves_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_async_summary, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1]
    for step in range(4):
        rows = np.array(res[0][step])
        assert rows.shape == (2, 25)
        assert rows[:, 0].tolist() == [float(step), 100.0 + step] and rows[1, 1] == float(step == 2)
        assert rows[0, 2:].tolist() == [0.0] * 23 and rows[1, 2:].tolist() == [0.5] * 23


def test_staging_buffers_of_a_variable_are_bounded(monkeypatch):
    """distributed._staged_to_device: the page-locked staging buffers rank 0 sends a reader level through are reused per
    (variable, shape) and a variable keeps those of its two most recent shapes only -- a reader whose window is re-cut again and
    again must not leave every old window's buffers page-locked for the life of the process.  CPU: pinning and
    events replaced by stand-ins."""
    from opendrift_amd import distributed as D

    class Ev:
        def __init__(self):
            self.waited = 0

        def record(self, stream=None):
            pass

        def synchronize(self):
            self.waited += 1

    monkeypatch.setattr(torch.Tensor, 'pin_memory', lambda self: self)
    monkeypatch.setattr(torch.cuda, 'Event', Ev)
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda dev=None: None)
    monkeypatch.setattr(D, '_STAGING', {})
    monkeypatch.setattr(D, '_STAGING_SHAPES', {})
    dev = torch.device('cpu')
    shapes = [(4, 6), (4, 6), (5, 6), (4, 6), (7, 3), (8, 3), (8, 3)]
    for k, shp in enumerate(shapes):
        a = np.full(shp, float(k), np.float32)
        t = D._staged_to_device(('u', shp), a, dev)
        assert t.shape == shp and float(t[0, 0]) == k
        assert len([key for key in D._STAGING if key[0] == 'u']) <= D._STAGING_KEEP
    assert sorted(key[1] for key in D._STAGING) == [(7, 3), (8, 3)] and D._STAGING_SHAPES['u'] == [(7, 3), (8, 3)]
    D._staged_to_device(('v', (4, 6)), np.zeros((4, 6), np.float32), dev)       # another variable has buffers of its own
    assert len(D._STAGING) == 3
    # the two buffers of a shape are used in turn, and a buffer is written again only after its last copy is done
    st = D._STAGING[('u', (8, 3))]
    assert st[2] in (0, 1) and all(e is not None for e in st[1])
    before = [e.waited for e in st[1]]
    D._staged_to_device(('u', (8, 3)), np.ones((8, 3), np.float32), dev)
    assert sum(e.waited for e in st[1] if e is not None) >= sum(before)


def test_communicator_id_travels_through_a_file_without_a_collective(tmp_path, monkeypatch):
    """distributed._exchange_unique_id: rank 0 writes the RCCL id atomically, the other ranks poll for a COMPLETE file of this
    launch (key: MASTER_ADDR / MASTER_PORT + the launcher's pid, or ODR_COMM_KEY); a file left behind by a dead job is ignored."""
    import threading
    import time
    from opendrift_amd import distributed as D
    monkeypatch.setenv('ODR_COMM_DIR', str(tmp_path))
    monkeypatch.setenv('ODR_COMM_KEY', 'job-a')
    got = {}
    th = threading.Thread(target=lambda: got.setdefault('id', D._exchange_unique_id(1, None, 256, timeout=20)[0]))
    th.start()
    time.sleep(0.2)
    ident, path = D._exchange_unique_id(0, lambda: bytes(range(256)), 256)
    th.join()
    assert got['id'] == ident == bytes(range(256)) and os.path.dirname(path) == str(tmp_path)
    # a stale file (older than ODR_COMM_ID_MAX_AGE) is not this job's
    monkeypatch.setenv('ODR_COMM_KEY', 'job-b')
    stale = os.path.join(str(tmp_path), 'odr_comm_id_job-b')
    with open(stale, 'wb') as f:
        f.write(b'x' * 256)
    os.utime(stale, (time.time() - 3600, time.time() - 3600))
    with pytest.raises(TimeoutError):
        D._exchange_unique_id(1, None, 256, timeout=0.5)


def test_rccl_probe_answers_without_a_gpu():
    """distributed._rccl_usable: every rank asks the device library for an RCCL id before any of them waits for another one; where
    that cannot work (no GPU / no librccl: this container) the answer is False -- never an exception -- and init() takes the
    torch layer (with a warning) unless ODR_DIST_BACKEND=rccl insists."""
    from opendrift_amd import distributed as D
    first = D._rccl_usable()
    assert first in (False, True) and D._rccl_usable() is first        # asked once (what RCCL answers without a device depends
    #                                                                    on which HIP runtime the process loaded first)
