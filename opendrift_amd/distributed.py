"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The path shards by particle: particles are independent within a step, every
GPU holds the full field block.  The only data-path exchange is the broadcast of a new field
block when a reader time level advances (rank 0 runs the host Reader), plus an all-reduce of a few
scalars (active counts, min/max for logging).  No particle migration, no halo exchange.
"""
import os

import numpy as np


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


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


COUNT_SLOTS = (0, 11)      # odr_reduce_local: n_active, n_surface are sums; every other slot is a maximum


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


def broadcast_reader_block(block_or_none, variables, src=0):
    """One reader time level from the rank that runs the host Reader to every rank: the coordinate metadata as a Python
    object, the arrays as tensors (RCCL broadcast into device memory under nccl).  Returns (meta, {variable: tensor})."""
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    meta = [None]
    arrays = None
    if rank == src:
        b = block_or_none
        meta = [{k: (np.asarray(b[k]) if k in ('x', 'y', 'z') and b.get(k) is not None else b.get(k))
                 for k in ('x', 'y', 'z', 'time', 's_level_variables') if k in b}]
        arrays = {v: np.ascontiguousarray(np.ma.filled(b[v], np.nan) if isinstance(b[v], np.ma.MaskedArray) else b[v],
                                          dtype=np.float32) for v in variables}
    if world > 1:
        dist.broadcast_object_list(meta, src=src)
    tens = broadcast_block(arrays, src=src)
    return meta[0], tens
