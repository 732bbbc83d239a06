"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The path shards by particle: particles are independent within a step, every
GPU holds the full field block.  The only data-path exchange is the broadcast of a new field
block when a reader time level advances (rank 0 runs the host Reader), plus an all-reduce of a few
scalars (active counts, min/max for logging).  No particle migration, no halo exchange.
"""
import os

import numpy as np


COUNT_SLOTS = (0, 11)      # odr_reduce_local: n_active, n_surface are sums; every other slot is a maximum


def env_world():
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)),
            int(os.environ.get('WORLD_SIZE', 1)))


def init(backend=None):
    """Initialise torch.distributed from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torchrun env)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:   # ODR_DIST_BACKEND=gloo: rehearsal of the N-rank flow on a box with fewer GPUs than ranks
            backend = os.environ.get('ODR_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_total, rank, world):
    """Contiguous block partition of particle IDs [0, n_total): rank r owns [lo, hi)."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_block(arrays, shapes=None, src=0, device=None):
    """Broadcast one field block {variable: float32 ndarray} from `src` to every rank.

    Returns {variable: torch.Tensor} on `device` (cuda:<local_rank> under nccl, cpu under gloo).
    Non-source ranks pass arrays=None and shapes={variable: shape}."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if device is None:
        device = torch.device('cuda', local_rank) if (world > 1 and dist.get_backend() == 'nccl') or \
            (world == 1 and torch.cuda.is_available()) else torch.device('cpu')
    if world > 1:
        meta = [None]
        if rank == src:
            meta = [{k: tuple(np.shape(v)) for k, v in arrays.items()}]
        dist.broadcast_object_list(meta, src=src)
        shapes = meta[0]
    elif shapes is None:
        shapes = {k: tuple(np.shape(v)) for k, v in arrays.items()}
    out = {}
    for k, shp in shapes.items():
        if rank == src:
            t = torch.from_numpy(np.ascontiguousarray(arrays[k], dtype=np.float32)).to(device)
        else:
            t = torch.empty(shp, dtype=torch.float32, device=device)
        if world > 1:
            dist.broadcast(t, src=src)
        out[k] = t
    if device.type == 'cuda':
        # the collective runs on RCCL's stream, the block is consumed on the library's upload stream: the tensors must
        # be complete before their pointers are handed over (odr_block_upload_device)
        torch.cuda.synchronize(device)
    return out


def allreduce_scalars(values, op='sum'):
    """All-reduce a handful of float64 scalars (counts: sum, extents: min/max)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    v = np.asarray(values, dtype=np.float64)
    if world == 1 or not dist.is_initialized():
        return v
    dev = torch.device('cuda', local_rank) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.from_numpy(v.copy()).to(dev)
    dist.all_reduce(t, op={'sum': dist.ReduceOp.SUM, 'min': dist.ReduceOp.MIN, 'max': dist.ReduceOp.MAX}[op])
    return t.cpu().numpy()


def allgather_vector(values):
    """ONE collective for everything a step needs from the other ranks: every rank's float64 vector, as rows of a
    [world, n] array (the caller sums / maximises the columns itself -- counts and maxima travel together)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    v = np.ascontiguousarray(values, dtype=np.float64)
    if world == 1 or not dist.is_initialized():
        return v[None, :]
    dev = torch.device('cuda', local_rank) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.from_numpy(v).to(dev)
    if dist.get_backend() == 'nccl':
        out = torch.empty((world, v.size), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy()
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return torch.stack(parts).numpy()


def start_allgather_vector(values):
    """allgather_vector started now and finished later (finish_allgather_vector): the step's collective travels while the
    device runs the launches that do not depend on it (OceanDrift.run(): the mixing launch of the step).  Returns a handle."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    v = np.ascontiguousarray(values, dtype=np.float64)
    if world == 1 or not dist.is_initialized():
        return ('done', v[None, :].copy())
    if dist.get_backend() == 'nccl':
        dev = torch.device('cuda', local_rank)
        t = torch.from_numpy(v).to(dev, non_blocking=True)
        out = torch.empty((world, v.size), dtype=torch.float64, device=dev)
        return ('nccl', dist.all_gather_into_tensor(out, t, async_op=True), out, t)
    t = torch.from_numpy(v.copy())
    parts = [torch.empty_like(t) for _ in range(world)]
    return ('gloo', dist.all_gather(parts, t, async_op=True), parts, t)


def finish_allgather_vector(handle):
    """The rows of every rank ([world, n]) of a collective begun with start_allgather_vector."""
    import torch
    if handle[0] == 'done':
        return handle[1]
    handle[1].wait()
    if handle[0] == 'nccl':
        return handle[2].cpu().numpy()
    return torch.stack(handle[2]).numpy()


def combine_rows(rows):
    """Rows of raw reduction slots (one per rank) -> the all-rank reductions: COUNT_SLOTS summed, the others maximised."""
    rows = np.asarray(rows, dtype=np.float64)
    out = rows.max(axis=0)
    for k in COUNT_SLOTS:
        out[k] = rows[:, k].sum()
    return out


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()




def combine_reductions(raw16):
    """The 16 raw reduction slots of every rank -> what one process holding all elements would have computed."""
    raw = np.asarray(raw16, dtype=np.float64)
    rank, local_rank, world = env_world()
    if world == 1:
        return raw
    mx = allreduce_scalars(raw, 'max')
    sm = allreduce_scalars([raw[k] for k in COUNT_SLOTS], 'sum')
    out = mx.copy()
    for j, k in enumerate(COUNT_SLOTS):
        out[k] = sm[j]
    return out


class RemoteReaderError(RuntimeError):
    """The reader of the rank that reads failed: raised on EVERY rank, so that all of them count the failure alike."""


def broadcast_reader_block(block_or_none, variables, src=0, error=None, shapes=None, async_op=False, content_ids=None):
    """One reader time level from the rank that runs the host Reader to every rank:
      * a one-number header (1 = the level follows, 0 = the reader failed: EVERY rank raises RemoteReaderError -- nobody is
        left waiting in a collective, and all ranks count the failure alike);
      * `shapes` None (first level of a reader): the coordinate metadata and the array shapes as a Python object; later
        levels pass the shapes they know and skip this (the metadata of a reader does not change);
      * the arrays as tensors (RCCL broadcast into device memory under nccl).
    async_op: the array broadcasts are only STARTED -- the prefetch of the next level while the current one is in use; all
    ranks step in lockstep, so the order of the collectives is the same everywhere.  Returns (meta or None, {variable:
    tensor}, [work handles]); finish_broadcast(works) before the tensors are used."""
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    meta, arrays = None, None
    if rank == src and error is None:
        b = block_or_none
        def one(a):
            return np.ascontiguousarray(np.ma.filled(a, np.nan) if isinstance(a, np.ma.MaskedArray) else a, dtype=np.float32)
        arrays, members = {}, {}
        for v in variables:
            if isinstance(b[v], (list, tuple)):     # ensemble members: stacked along the layer axis, [members x nz, ny, nx]
                m = np.stack([one(a) for a in b[v]])
                arrays[v], members[v] = np.ascontiguousarray(m.reshape((-1,) + m.shape[-2:])), len(b[v])
            else:
                arrays[v] = one(b[v])
        if shapes is None:
            meta = {k: (np.asarray(b[k]) if k in ('x', 'y', 'z') and b.get(k) is not None else b.get(k))
                    for k in ('x', 'y', 'z', 'time', 's_level_variables') if k in b}
            meta['__shapes__'] = {v: tuple(a.shape) for v, a in arrays.items()}
            meta['__members__'] = members
    if world > 1:
        ok = allreduce_scalars([0.0 if (rank == src and error is not None) else 1.0], 'min')[0]
        if ok < 1:
            raise RemoteReaderError('the reader failed on rank %d%s' % (src, (': %r' % (error,)) if error is not None else ''))
        if shapes is None:
            box = [meta]
            dist.broadcast_object_list(box, src=src)
            meta = box[0]
    elif error is not None:
        raise error
    if shapes is None:
        shapes = meta.pop('__shapes__')
    tens, works = start_broadcast_block(arrays, shapes, src)
    # the content ids of the level's variables (device.ContentIds, assigned on `src` where the host arrays are) travel with it
    import torch
    ids = torch.zeros(len(variables), dtype=torch.int64)
    if rank == src and content_ids:
        ids = torch.tensor([int(content_ids.get(v, 0)) for v in variables], dtype=torch.int64)
    ids = ids.to(_device())
    if world > 1:
        works.append(dist.broadcast(ids, src=src, async_op=True))
    tens['__cid__'] = ids
    if not async_op:
        finish_broadcast(works)
        works = []
    return meta, tens, works


def _device():
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    return torch.device('cuda', local_rank) if (world > 1 and dist.get_backend() == 'nccl') or \
        (world == 1 and torch.cuda.is_available()) else torch.device('cpu')


_STAGING = {}      # (variable, shape) -> [two page-locked host tensors, the events of their last copies, turn]
_STAGING_SHAPES = {}   # variable -> the shapes in use, most recent last (at most _STAGING_KEEP: the others are let go)
_STAGING_KEEP = 2      # two readers may deliver the same variable on different grids; a window that is re-cut changes the shape


def _staged_to_device(key, a, dev):
    """Host array -> device tensor through a page-locked staging buffer that is REUSED: two per (variable, shape), in turn
    (hipHostMalloc of a 210 MB level on the critical path every few steps otherwise); a buffer is overwritten only after
    the copy that last read it has completed (its event).  A variable keeps the buffers of its two most recent shapes: a
    reader whose window is re-cut again and again does not leave every old window's buffers page-locked for the life of the
    process."""
    import torch
    name, shape = key
    order = _STAGING_SHAPES.setdefault(name, [])
    if shape in order:
        order.remove(shape)
    order.append(shape)
    while len(order) > _STAGING_KEEP:
        old = _STAGING.pop((name, order.pop(0)), None)
        if old is not None:
            for ev in old[1]:
                if ev is not None:
                    ev.synchronize()       # (its last copy may still be reading the buffer)
    st = _STAGING.get(key)
    if st is None:
        st = _STAGING[key] = [[torch.empty(shape, dtype=torch.float32).pin_memory() for _ in range(2)], [None, None], 0]
    bufs, evs, turn = st[0], st[1], st[2]
    st[2] = 1 - turn
    if evs[turn] is not None:
        evs[turn].synchronize()
    bufs[turn].copy_(torch.from_numpy(a))
    t = bufs[turn].to(dev, non_blocking=True)
    evs[turn] = torch.cuda.Event()
    evs[turn].record(torch.cuda.current_stream(dev))
    return t


def start_broadcast_block(arrays, shapes, src=0):
    """Start the broadcasts of one block whose shapes every rank knows; returns ({variable: tensor}, [work handles])."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    dev = _device()
    tens, works = {}, []
    for k, shp in shapes.items():
        if rank == src:
            a = np.ascontiguousarray(arrays[k], dtype=np.float32)
            t = _staged_to_device((k, tuple(a.shape)), a, dev) if dev.type == 'cuda' else torch.from_numpy(a)
        else:
            t = torch.empty(tuple(shp), dtype=torch.float32, device=dev)
        if world > 1:
            works.append(dist.broadcast(t, src=src, async_op=True))
        tens[k] = t
    return tens, works


def finish_broadcast(works):
    import torch
    for w in works:
        w.wait()
    dev = _device()
    if dev.type == 'cuda':
        torch.cuda.synchronize(dev)    # the tensors are consumed on the library's upload stream, not on torch's


def broadcast_block_known(arrays, shapes, src=0):
    tens, works = start_broadcast_block(arrays, shapes, src)
    finish_broadcast(works)
    return tens
