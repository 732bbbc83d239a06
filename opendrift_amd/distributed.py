"""Multi-GPU plumbing: one process per GPU; the collectives go over RCCL / xGMI through the C ABI of libodrift_hip.so
(odr_comm_*: no torch in the process), or over torch.distributed (gloo rehearsal on CPU boxes).

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


# ---------------------------------------------------------------------------------------------------------------------
# Two backends behind the same functions.
#   'rccl'  -- the product path: the collectives are entry points of libodrift_hip.so (csrc/odr_comm.hip: librccl through
#              the C ABI, include/odrift.h "communication"); NO torch in the process, one HIP runtime.  Default when the
#              process sees a GPU.
#   'torch' -- torch.distributed (gloo on CPU boxes: the rehearsal of the N-rank flow in tests/; nccl on request:
#              ODR_DIST_BACKEND=nccl).
_BACKEND = None        # None until init(): 'rccl' | 'torch'
_COMM = None           # rccl: the Context the communicator was made on


def backend():
    return _BACKEND


def comm_info():
    """What the bench line prints as `comm` (the driver's scaling run shows from it that RCCL saw N ranks)."""
    rank, local_rank, world = env_world()
    if _BACKEND == 'rccl':
        import ctypes as C
        from . import _abi
        lib = _abi.load()
        r, n, h, v, k = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_int32(), C.c_uint64()
        _abi.check(lib.odr_comm_info(C.byref(r), C.byref(n), C.byref(h), C.byref(v), C.byref(k)))
        return dict(backend='rccl (libodrift_hip.so: odr_comm_*)', nranks_seen=int(n.value), rank=int(r.value),
                    unique_id_hash='%016x' % h.value, rccl_version=int(v.value), collectives=int(k.value))
    if _BACKEND == 'torch':
        import torch.distributed as dist
        return dict(backend='torch.distributed/' + dist.get_backend(), nranks_seen=dist.get_world_size(), rank=dist.get_rank(),
                    unique_id_hash=None)
    return dict(backend=None, nranks_seen=1, rank=0, unique_id_hash=None)


def _choose_backend(asked=None):
    asked = asked or os.environ.get('ODR_DIST_BACKEND')
    if asked in ('gloo', 'nccl', 'torch'):
        return 'torch', (None if asked == 'torch' else asked)
    if asked == 'rccl' or (asked is None and os.path.exists('/dev/kfd')):
        return 'rccl', None
    return 'torch', None


def _exchange_unique_id(rank, make, nbytes, timeout=300.0):
    """The communicator id from rank 0 to every rank of the job WITHOUT a collective (there is none yet) and without torch:
    a file every process of the node can see.  Key: MASTER_ADDR / MASTER_PORT of the launcher + the launcher's pid (the ranks
    of one torchrun are children of one agent) or ODR_COMM_KEY; directory ODR_COMM_DIR (default: the temporary directory).
    Rank 0 removes the file once its communicator is up (init_rccl), i.e. once every rank has read it."""
    import tempfile
    import time
    key = os.environ.get('ODR_COMM_KEY') or '%s_%s_%d' % (os.environ.get('MASTER_ADDR', 'local'),
                                                         os.environ.get('MASTER_PORT', '0'), os.getppid())
    key = ''.join(ch if ch.isalnum() or ch in '._-' else '_' for ch in key)
    path = os.path.join(os.environ.get('ODR_COMM_DIR', tempfile.gettempdir()), 'odr_comm_id_' + key)
    if rank == 0:
        ident = make()
        tmp = path + '.%d.tmp' % os.getpid()
        with open(tmp, 'wb') as f:
            f.write(ident)
        os.replace(tmp, path)          # atomic: a reader sees the whole id or no file
        return ident, path
    t0 = time.time()
    while True:
        try:
            # (a file left behind by a job that died before its rank 0 removed it is not this job's: the ranks of one launch start
            # within seconds of each other)
            if os.path.getmtime(path) > t0 - float(os.environ.get('ODR_COMM_ID_MAX_AGE', 600.0)):
                with open(path, 'rb') as f:
                    ident = f.read()
                if len(ident) == nbytes:
                    return ident, path
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError('rank %d: no communicator id at %s after %.0f s (is rank 0 running?)' % (rank, path, timeout))
        time.sleep(0.01)


def init_rccl(ctx=None, world1=False):
    """One RCCL communicator pair for this process through the C ABI (odr_comm_init).  ctx: the Context of this rank's GPU
    (default: one on device LOCAL_RANK % device count).  world1: make the communicator in a one-rank job as well (tests)."""
    global _BACKEND, _COMM
    import ctypes as C
    from . import _abi
    rank, local_rank, world = env_world()
    if _BACKEND == 'rccl':
        return rank, local_rank, world
    if world == 1 and not world1:
        return rank, local_rank, world
    lib = _abi.load()
    if ctx is None:
        from .device import Context
        n = C.c_int32()
        _abi.check(lib.odr_device_count(C.byref(n)))
        ctx = Context(device=local_rank % max(1, n.value), seed=0)

    def make():
        buf = (C.c_uint8 * _abi.COMM_ID_BYTES)()
        _abi.check(lib.odr_comm_unique_id(buf))
        return bytes(buf)
    ident, path = _exchange_unique_id(rank, make, _abi.COMM_ID_BYTES)
    buf = (C.c_uint8 * _abi.COMM_ID_BYTES).from_buffer_copy(ident)
    try:
        _abi.check(lib.odr_comm_init(ctx.h, buf, rank, world))
    finally:
        if rank == 0:        # every rank has joined (ncclCommInitRank returns when all of them have) or the job is lost anyway
            try:
                os.unlink(path)
            except OSError:
                pass
    _BACKEND, _COMM = 'rccl', ctx
    return rank, local_rank, world


def shutdown():
    """End of the process: the communicator goes before the contexts do."""
    global _BACKEND, _COMM
    if _BACKEND == 'rccl':
        from . import _abi
        _abi.check(_abi.load().odr_comm_destroy())
        _BACKEND, _COMM = None, None
    elif _BACKEND == 'torch':
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
        _BACKEND = None


def device_count():
    """GPUs this process sees, asked of the device library (no torch)."""
    import ctypes as C
    from . import _abi
    n = C.c_int32()
    _abi.check(_abi.load().odr_device_count(C.byref(n)))
    return int(n.value)


def _rccl_allreduce(values, op):
    import ctypes as C
    from . import _abi
    v = np.ascontiguousarray(values, dtype=np.float64).copy()
    _abi.check(_abi.load().odr_allreduce_scalars(_COMM.h, v.ctypes.data_as(C.POINTER(C.c_double)), int(v.size),
                                                 {'sum': 0, 'min': 1, 'max': 2}[op]))
    return v


def broadcast_object(obj, src=0):
    """A small Python object (metadata of a reader level) from `src` to every rank."""
    rank, local_rank, world = env_world()
    if world == 1 and _BACKEND != 'rccl':
        return obj
    if _BACKEND == 'rccl':
        import ctypes as C
        import pickle
        from . import _abi
        lib = _abi.load()
        data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL) if rank == src else b''
        n = int(_rccl_allreduce([float(len(data))], 'max')[0])      # the length first: every rank passes the same byte count
        buf = (C.c_uint8 * n).from_buffer_copy(data) if rank == src else (C.c_uint8 * n)()
        _abi.check(lib.odr_comm_broadcast_bytes(_COMM.h, C.cast(buf, C.c_void_p), n, src))
        return obj if rank == src else pickle.loads(bytes(buf))
    import torch.distributed as dist
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


_RCCL_USABLE = None


def _rccl_usable():
    """Can this process make RCCL ids through the C ABI (librccl loads, the device library answers)?  Asked by EVERY rank before
    any of them waits for another one: the answer is a property of the node's software, so the ranks agree on it."""
    global _RCCL_USABLE
    if _RCCL_USABLE is None:            # (asked once: a library that failed to initialise answers anything afterwards)
        import ctypes as C
        try:
            from . import _abi
            buf = (C.c_uint8 * _abi.COMM_ID_BYTES)()
            _RCCL_USABLE = _abi.load().odr_comm_unique_id(buf) == 0
        except Exception:
            _RCCL_USABLE = False
    return _RCCL_USABLE


def init(backend=None):
    """The communication layer of a sharded run from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun's environment):
    RCCL through the C ABI where there is a GPU (no torch in the process), torch.distributed otherwise / on request
    (ODR_DIST_BACKEND = rccl | gloo | nccl)."""
    global _BACKEND
    rank, local_rank, world = env_world()
    kind, torch_backend = _choose_backend(backend)
    if world > 1 and kind == 'rccl' and _BACKEND != 'torch':
        if _rccl_usable() or os.environ.get('ODR_DIST_BACKEND') == 'rccl':
            return init_rccl()
        # the RCCL library cannot be loaded on this node (the same for every rank of the job: one image): the torch layer
        import warnings
        warnings.warn('opendrift_amd.distributed: librccl is not usable through the C ABI here, falling back to torch.distributed')
    if world == 1:
        return rank, local_rank, world
    backend = torch_backend
    import torch
    import torch.distributed as dist
    _BACKEND = 'torch'
    if world > 1 and not dist.is_initialized():
        if backend is None:   # (no GPU in sight: the rehearsal of the N-rank flow over gloo)
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
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
    if _BACKEND == 'rccl':
        return _rccl_allreduce(values, op)
    if _BACKEND is None:
        return np.asarray(values, dtype=np.float64)
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
    if _BACKEND == 'rccl':
        return finish_allgather_vector(start_allgather_vector(values))
    if _BACKEND is None:
        return np.ascontiguousarray(values, dtype=np.float64)[None, :]
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


def start_allgather_vector(values, from_scan_ctx=None):
    """allgather_vector started now and finished later (finish_allgather_vector): the step's collective travels while the
    device runs the launches that do not depend on it (OceanDrift.run(): the mixing launch of the step).  Returns a handle.
    from_scan_ctx (rccl): the Context whose status scan is open (between scan_status_begin and _end) -- entries 0 .. 8 of the
    row (elements that stay, eight status flags) are then taken from the fold ON THE DEVICE, the host has not read them yet."""
    if _BACKEND == 'rccl':
        import ctypes as C
        from . import _abi
        v = np.ascontiguousarray(values, dtype=np.float64)
        c = from_scan_ctx if from_scan_ctx is not None else _COMM
        _abi.check(_abi.load().odr_comm_allgather_begin(c.h, v.ctypes.data_as(C.POINTER(C.c_double)), int(v.size),
                                                       1 if from_scan_ctx is not None else 0))
        return ('rccl', int(v.size))
    if _BACKEND is None:
        return ('done', np.ascontiguousarray(values, dtype=np.float64)[None, :].copy())
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
    if handle[0] == 'done':
        return handle[1]
    if handle[0] == 'rccl':
        import ctypes as C
        from . import _abi
        rank, local_rank, world = env_world()
        out = np.empty((world, handle[1]), np.float64)
        _abi.check(_abi.load().odr_comm_allgather_end(_COMM.h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out
    import torch
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
    if _BACKEND == 'rccl':
        from . import _abi
        _abi.check(_abi.load().odr_comm_barrier(_COMM.h))
        return
    if _BACKEND is None:
        return
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()




def combine_reductions(raw16):
    """The 16 raw reduction slots of every rank -> what one process holding all elements would have computed."""
    raw = np.asarray(raw16, dtype=np.float64)
    rank, local_rank, world = env_world()
    if world == 1 and _BACKEND != 'rccl':
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
