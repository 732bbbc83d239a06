"""Thin NumPy-facing wrapper over the C ABI (include/odrift.h).

`Context` is the device image of OpenDrift's Environment (readers + priority lists);
`Particles` is the device image of a LagrangianArray.  All arithmetic happens in
libodrift_hip.so; this file only marshals arrays.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import VARIABLES, check

_dp, _fp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)


def _vid(v):
    return v if isinstance(v, (int, np.integer)) else VARIABLES[v]


def _d(a, n=None):
    if a is None:
        return None, None
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (n,)) if n is not None
                             else np.asarray(a, dtype=np.float64))
    return a, a.ctypes.data_as(_dp)


def _f(a, n=None):
    if a is None:
        return None, None
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float32), (n,)) if n is not None
                             else np.asarray(a, dtype=np.float32))
    return a, a.ctypes.data_as(_fp)


def _i(a, n=None):
    if a is None:
        return None, None
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.int32), (n,)) if n is not None
                             else np.asarray(a, dtype=np.int32))
    return a, a.ctypes.data_as(_ip)


def proj_desc(proj):
    """dict(kind='latlong'|'stere_equit_sphere'|'stere_polar'|'merc'|'lcc'|'tmerc'|'laea'|'stere_oblique'|'ob_tran', a, rf|es, lat0,
    lon0, lat_ts, k0, x0, y0, lat1, lat2) -- projection.parse_proj4's output (ob_tran: lat1 = o_lat_p, lat2 = o_lon_p)"""
    if proj is None or proj.get('kind', 'latlong') == 'latlong':
        return None
    kind = {'stere_equit_sphere': _abi.PROJ_STERE_EQUIT_SPHERE, 'stere_polar': _abi.PROJ_STERE_POLAR,
            'merc': _abi.PROJ_MERC, 'lcc': _abi.PROJ_LCC, 'tmerc': _abi.PROJ_TMERC, 'laea': _abi.PROJ_LAEA,
            'stere_oblique': _abi.PROJ_STERE_OBLIQUE, 'ob_tran': _abi.PROJ_OB_TRAN}[proj['kind']]
    if 'es' in proj:
        es = proj['es']
    else:
        rf = proj.get('rf', 0.0)
        f = 0.0 if not rf else 1.0 / rf
        es = f * (2 - f)
    return _abi.ProjDesc(kind, proj.get('a', 6378137.0), es, proj.get('lat0', 0.0), proj.get('lon0', 0.0),
                         proj.get('lat_ts', 90.0), proj.get('k0', 1.0), proj.get('x0', 0.0), proj.get('y0', 0.0),
                         proj.get('lat1', 0.0), proj.get('lat2', proj.get('lat1', 0.0)))


class ContentIds:
    """Content ids of the 2-D variables of successive time levels of one reader (odr_block_set_content_ids) BY COMPARISON: a
    variable whose array equals, bit for bit, the one last seen for it keeps its id -- the reader re-reads the sea floor
    depth and the land mask with every block (structured.py:15-94), and the samplers then gather such a variable at one of
    the two bracketing levels.  Compared against a private copy (the caller may reuse its buffers); costs a pass over the
    array per level, so the reader bindings prefer a reader's own declaration (`static_variables`).  3-D variables,
    ensemble lists and device pointers get id 0 (unknown)."""
    _counter = 0

    def __init__(self):
        self.last = {}

    def assign(self, names, arrays):
        out = {}
        for k in names:
            a = arrays.get(k) if hasattr(arrays, 'get') else None
            if not isinstance(a, np.ndarray) or a.ndim != 2:
                out[k] = 0
                continue
            prev = self.last.get(k)
            a = np.ascontiguousarray(a, dtype=np.float32)
            if prev is None or prev[1].shape != a.shape or not np.array_equal(prev[1].view(np.uint32), a.view(np.uint32)):
                ContentIds._counter += 1
                prev = self.last[k] = (ContentIds._counter, np.array(a, copy=True))
            out[k] = prev[0]
        return out


def sea_water_density_default():
    """PhysicsMethods.sea_water_density() with its default arguments T=10., S=35. (physics_methods.py:574-608): the
    UNESCO 1983 one-atmosphere equation of state, a float64 constant of the oil formulae."""
    T, S = 10., 35.
    R1 = ((((6.536332E-09 * T - 1.120083E-06) * T + 1.001685E-04) * T - 9.095290E-03) * T + 6.793952E-02) * T - 28.263737
    R2 = (((5.3875E-09 * T - 8.2467E-07) * T + 7.6438E-05) * T - 4.0899E-03) * T + 8.24493E-01
    R3 = (-1.6546E-06 * T + 1.0227E-04) * T - 5.72466E-03
    SIG = R1 + (4.8314E-04 * S + R3 * np.sqrt(S) + R2) * S
    return float(SIG + 28.106331 + 1000.)


class Context:
    def __init__(self, device=0, seed=0):
        self.lib = _abi.load()
        self.h = C.c_void_p()
        check(self.lib.odr_ctx_create(device, seed, C.byref(self.h)))
        self.device = device
        self._grids = {}
        self.stage_math = 'exact'
        if os.environ.get('ODR_STAGE_MATH'):     # what-if runs of whole test / bench sessions: 'exact' | 'fast'
            self.set_stage_math(os.environ['ODR_STAGE_MATH'])

    def close(self):
        if self.h:
            self.lib.odr_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.lib.odr_sync(self.h))

    def set_stream(self, stream_ptr):
        check(self.lib.odr_set_stream(self.h, C.c_void_p(stream_ptr)))

    # ---- sources ----
    def add_constant(self, values):
        ids, pi = _i([_vid(k) for k in values])
        vals, pv = _d([float(v) for v in values.values()])
        sid = C.c_int32()
        check(self.lib.odr_source_constant(self.h, len(ids), pi, pv, C.byref(sid)))
        return sid.value

    def add_landmask(self, lon0, lat0, dlon, dlat, cells):
        """Landmask raster source (reader_global_landmask.Reader on the device): cells[iy, ix] != 0 is land."""
        cells = np.ascontiguousarray(np.asarray(cells) != 0, dtype=np.uint8)
        ny, nx = cells.shape
        sid = C.c_int32()
        check(self.lib.odr_source_landmask(self.h, nx, ny, float(lon0), float(lat0), float(dlon), float(dlat),
                                           cells.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(sid)))
        return sid.value

    def add_double_gyre(self, A=0.25, epsilon=0.1, omega=0.628, t0=0.0):
        prm, pp = _d([A, epsilon, omega, t0])
        sid = C.c_int32()
        check(self.lib.odr_source_analytic(self.h, _abi.ANALYTIC_DOUBLE_GYRE, pp, 4, C.byref(sid)))
        return sid.value

    def add_oscillating(self, variable, amplitude, period_s, t0):
        prm, pp = _d([_vid(variable), amplitude, period_s, t0])
        sid = C.c_int32()
        check(self.lib.odr_source_analytic(self.h, _abi.ANALYTIC_OSCILLATING, pp, 4, C.byref(sid)))
        return sid.value

    def add_grid(self, x, y, z=None, proj=None, lon_mode=1, mod360_x=0, domain=None):
        """x, y: the reader's coordinate arrays in the dtype it hands out (float32 for file readers)."""
        x, y = np.asarray(x), np.asarray(y)
        if domain is None:
            domain = (float(x.min()), float(x.max()), float(y.min()), float(y.max()), -np.inf, np.inf)
        dom, pd = _d(domain)
        nz = 0 if z is None else int(np.size(z))
        zz, pz = _d(np.atleast_1d(z)) if nz > 1 else (None, None)
        desc = proj_desc(proj)
        sid = C.c_int32()
        check(self.lib.odr_source_grid(self.h, C.byref(desc) if desc is not None else None, pd, lon_mode,
                                       mod360_x, nz, pz, C.byref(sid)))
        # index maps with the spans formed in the coordinate dtype (interpolators.py:32-33,110-111)
        xy8 = np.array([float(x[0]), float(x[-1] - x[0]), float(y[0]), float(y[-1] - y[0]),
                        float(x.min()), float(x.max() - x.min()), float(y.min()), float(y.max() - y.min())])
        self._grids[sid.value] = dict(xy8=xy8, ny=len(y), nx=len(x), nz=max(nz, 1))
        # float32 coordinate arrays: the index maps of a geographic reader are float32 arithmetic in the first get_environment of
        # a run (odr_ctx_set_position_class; interpolators.py:110-111 with elements.py:71-88)
        check(self.lib.odr_source_set_coordinate_dtype(self.h, sid.value, int(x.dtype == np.float32), int(y.dtype == np.float32)))
        return sid.value

    def add_grid_curvilinear(self, lon2d, lat2d, z=None, lon_mode=None, domain=None):
        """Reader without a projection (basereader/structured.py:44-113): 2D lon/lat node arrays, x/y = pixel indices."""
        lon2d = np.ascontiguousarray(lon2d, dtype=np.float64)
        lat2d = np.ascontiguousarray(lat2d, dtype=np.float64)
        assert lon2d.ndim == 2 and lon2d.shape == lat2d.shape
        ny, nx = lon2d.shape
        if lon_mode is None:   # modulate_longitude, variables.py:259-280: from the corner longitudes
            lon_mode = 1 if min(lon2d[0, 0], lon2d[0, -1], lon2d[-1, 0], lon2d[-1, -1]) < 0 else 2
        if domain is None:
            domain = (0.0, nx - 1.0, 0.0, ny - 1.0, -np.inf, np.inf)
        dom, pd = _d(domain)
        nz = 0 if z is None else int(np.size(z))
        zz, pz = _d(np.atleast_1d(z)) if nz > 1 else (None, None)
        sid = C.c_int32()
        check(self.lib.odr_source_grid_curvilinear(self.h, lon2d.ctypes.data_as(_dp), lat2d.ctypes.data_as(_dp), ny, nx,
                                                   pd, int(lon_mode), nz, pz, C.byref(sid)))
        # x = arange(nx), y = arange(ny): the index maps of interpolators.py:32-33,110-111
        xy8 = np.array([0.0, nx - 1.0, 0.0, ny - 1.0, 0.0, nx - 1.0, 0.0, ny - 1.0])
        self._grids[sid.value] = dict(xy8=xy8, ny=ny, nx=nx, nz=max(nz, 1))
        return sid.value

    def lonlat2xy(self, sid, lon, lat):
        """Variables.lonlat2xy of one source for host positions (variables.py:111-143)."""
        lon, pl = _d(np.atleast_1d(lon))
        lat, pa = _d(np.atleast_1d(lat))
        x, y = np.empty(len(lon)), np.empty(len(lon))
        check(self.lib.odr_source_lonlat2xy(self.h, sid, len(lon), pl, pa, x.ctypes.data_as(_dp), y.ctypes.data_as(_dp)))
        return x, y

    def _ensemble_arrays(self, sid, arrays):
        """A variable given as a LIST of member arrays (ensemble data, basereader/structured.py:125-147): declared on the
        source (odr_source_set_members) and stacked along the layer axis, [members x nz, ny, nx]."""
        out = {}
        for k, v in arrays.items():
            if isinstance(v, (list, tuple)):
                m = np.stack([np.ma.filled(a, np.nan) if isinstance(a, np.ma.MaskedArray) else np.asarray(a) for a in v]).astype(np.float32)
                known = self._grids[sid].setdefault('members', {})
                if known.get(k) != len(v):
                    check(self.lib.odr_source_set_members(self.h, sid, _vid(k), len(v)))
                    known[k] = len(v)
                v = np.ascontiguousarray(m.reshape((-1,) + m.shape[-2:]))
            out[k] = v
        return out

    def declare_members(self, sid, variable, members):
        """The variable of this source arrives as `members` ensemble members stacked along the layer axis (what
        _ensemble_arrays does for a list of member arrays; a sharded run receives the stack itself)."""
        known = self._grids[sid].setdefault('members', {})
        if known.get(variable) != members:
            check(self.lib.odr_source_set_members(self.h, sid, _vid(variable), int(members)))
            known[variable] = members

    def _content_ids(self, sid, slot, given):
        """odr_block_set_content_ids: {variable: id} for the level just uploaded (0 / missing: unknown).  The ids come from
        whoever knows the data -- a reader that declares `static_variables`, or ContentIds.assign where a comparison by
        value is affordable; nothing is compared here (a 1 M-node comparison per variable and level on the host costs more
        than the gathers it saves)."""
        given = {k: v for k, v in (given or {}).items() if v}
        if given:
            (va, pv), ia = _i([_vid(k) for k in given]), np.asarray(list(given.values()), dtype=np.uint64)
            check(self.lib.odr_block_set_content_ids(self.h, sid, slot, len(given), pv, ia.ctypes.data_as(C.POINTER(C.c_uint64))))

    def upload_block(self, sid, slot, t_epoch, arrays, content_ids=None):
        """arrays: {variable: float32 [ny,nx] or [nz,ny,nx], or a list of such arrays (ensemble members)} -- one
        ReaderBlock / time level."""
        g = self._grids[sid]
        arrays = self._ensemble_arrays(sid, arrays)
        names = list(arrays)
        ids, pi = _i([_vid(k) for k in names])
        keep = [np.ascontiguousarray(np.ma.filled(arrays[k], np.nan) if isinstance(arrays[k], np.ma.MaskedArray)
                                     else arrays[k], dtype=np.float32) for k in names]
        for a in keep:
            assert a.shape[-2:] == (g['ny'], g['nx']), 'block shape %s does not match grid' % (a.shape,)
        nzs, pn = _i([a.shape[0] if a.ndim == 3 else 1 for a in keep])
        ptrs = (_fp * len(keep))(*[a.ctypes.data_as(_fp) for a in keep])
        xy8, px = _d(g['xy8'])
        check(self.lib.odr_block_upload(self.h, sid, slot, float(t_epoch), len(keep), pi, ptrs, pn, g['ny'],
                                        g['nx'], px))
        self._content_ids(sid, slot, content_ids)

    def upload_block_device(self, sid, slot, t_epoch, dev_ptrs, var_nz, content_ids=None):
        """dev_ptrs: {variable: device pointer (int) of a float32 array already in HBM, or a host NumPy array};
        var_nz: levels per variable (needed for the device pointers); content_ids: {variable: id} assigned by
        ContentIds.assign where the host arrays were (the rank that read the level)."""
        g = self._grids[sid]
        dev_ptrs = self._ensemble_arrays(sid, dev_ptrs)
        names = list(dev_ptrs)
        ids, pi = _i([_vid(k) for k in names])
        keep = {k: np.ascontiguousarray(np.ma.filled(v, np.nan) if isinstance(v, np.ma.MaskedArray) else v, dtype=np.float32)
                for k, v in dev_ptrs.items() if not isinstance(v, (int, np.integer))}
        nzs, pn = _i([(keep[k].shape[0] if keep[k].ndim == 3 else 1) if k in keep else var_nz[k] for k in names])
        ptrs = (C.c_void_p * len(names))(*[C.c_void_p(keep[k].ctypes.data if k in keep else int(dev_ptrs[k])) for k in names])
        xy8, px = _d(g['xy8'])
        check(self.lib.odr_block_upload_device(self.h, sid, slot, float(t_epoch), len(names), pi, ptrs, pn,
                                               g['ny'], g['nx'], px))
        self._content_ids(sid, slot, content_ids)

    def upload_block_async(self, sid, slot, t_epoch, arrays, var_nz=None, content_ids=None):
        """Enqueue the upload of one time level on the upload stream and return (the simulation continues); the
        block becomes visible with commit_block().  arrays: {variable: float32 host array (pin it with pin() for a true
        DMA transfer) or device pointer (int)}.  The arrays are kept referenced until the commit."""
        g = self._grids[sid]
        arrays = self._ensemble_arrays(sid, arrays)
        names = list(arrays)
        ids, pi = _i([_vid(k) for k in names])
        keep = {k: np.ascontiguousarray(np.ma.filled(v, np.nan) if isinstance(v, np.ma.MaskedArray) else v, dtype=np.float32)
                for k, v in arrays.items() if not isinstance(v, (int, np.integer))}
        nzs, pn = _i([(keep[k].shape[0] if keep[k].ndim == 3 else 1) if k in keep else var_nz[k] for k in names])
        ptrs = (C.c_void_p * len(names))(*[C.c_void_p(keep[k].ctypes.data if k in keep else int(arrays[k])) for k in names])
        xy8, px = _d(g['xy8'])
        check(self.lib.odr_block_upload_async(self.h, sid, slot, float(t_epoch), len(names), pi, ptrs, pn, g['ny'], g['nx'], px))
        self._content_ids(sid, slot, content_ids)      # (on the staged level: it becomes the slot's with the commit)
        self._staged_refs = getattr(self, '_staged_refs', {})
        self._staged_refs[(sid, slot)] = keep

    def block_broadcast(self, sid, slot, t_epoch, arrays, shapes, root=0, content_ids=None):
        """One reader time level from rank `root` to every rank over RCCL (odr_block_broadcast): staged on the upload stream
        like upload_block_async -- commit_block() makes it current.  arrays: {variable: float32 host array} on `root`, None
        elsewhere; shapes: {variable: shape} on every rank (the order of its keys is the order of the level's variables)."""
        g = self._grids[sid]
        names = list(shapes)
        ids, pi = _i([_vid(k) for k in names])
        nzs, pn = _i([(int(shapes[k][0]) if len(shapes[k]) == 3 else 1) for k in names])
        xy8, px = _d(g['xy8'])
        keep, ptrs = None, None
        if arrays is not None:
            keep = {k: np.ascontiguousarray(np.ma.filled(arrays[k], np.nan) if isinstance(arrays[k], np.ma.MaskedArray) else arrays[k],
                                            dtype=np.float32) for k in names}
            for k in names:
                if tuple(keep[k].shape) != tuple(shapes[k]):
                    raise ValueError('block_broadcast: %s has shape %s, announced %s' % (k, keep[k].shape, tuple(shapes[k])))
            ptrs = (C.c_void_p * len(names))(*[C.c_void_p(keep[k].ctypes.data) for k in names])
        check(self.lib.odr_block_broadcast(self.h, sid, slot, float(t_epoch), len(names), pi, ptrs, pn, g['ny'], g['nx'], px, int(root)))
        self._content_ids(sid, slot, content_ids)
        self._staged_refs = getattr(self, '_staged_refs', {})
        self._staged_refs[(sid, slot)] = keep

    def commit_block(self, sid, slot):
        check(self.lib.odr_block_commit(self.h, sid, slot))
        getattr(self, '_staged_refs', {}).pop((sid, slot), None)

    def pin(self, array):
        """Page-lock a host array (hipHostRegister) so that uploads from it are asynchronous DMA transfers."""
        a = np.asarray(array)
        assert a.flags['C_CONTIGUOUS']
        check(self.lib.odr_host_register(self.h, C.c_void_p(a.ctypes.data), a.nbytes))
        self._pinned = getattr(self, '_pinned', [])
        self._pinned.append(a)
        return a

    def unpin(self, array):
        check(self.lib.odr_host_unregister(self.h, C.c_void_p(np.asarray(array).ctypes.data)))

    def set_position_class(self, float32):
        """True: the main-loop samples that follow see the reference's float32 element arrays of the first get_environment of a run
        (odr_ctx_set_position_class: modulate_longitude in float32); False ends it."""
        check(self.lib.odr_ctx_set_position_class(self.h, 1 if float32 else 0))

    def set_stage_math(self, mode):
        """'exact' | 'fast': arithmetic of the Runge-Kutta stage evaluations (odr_ctx_set_stage_math, include/odrift.h)"""
        check(self.lib.odr_ctx_set_stage_math(self.h, _abi.STAGE_MATH[mode]))
        self.stage_math = mode

    def set_step_reduce(self, on=True, wind_drift_depth=0.1, relative_wind=False):
        """The movers' global early-out tests formed by the env_coast_advect launch instead of by a pass of their own
        (odr_ctx_set_step_reduce)."""
        check(self.lib.odr_ctx_set_step_reduce(self.h, int(bool(on)), float(wind_drift_depth), int(bool(relative_wind))))

    def set_seafloor_action(self, action, status_code=0):
        """general:seafloor_action for the sea floor checks inside update() (vertical_buoyancy, vertical_mixing)."""
        a = {'none': 0, 'lift_to_seafloor': 1, 'deactivate': 2, 'previous': 3}[action]
        check(self.lib.odr_set_seafloor_action(self.h, a, int(status_code)))

    def set_time_coverage(self, sid, t_start, t_end, always_valid=False):
        check(self.lib.odr_source_time_coverage(self.h, sid, float(t_start), float(t_end), int(always_valid)))

    def drop_block(self, sid, slot):
        check(self.lib.odr_block_drop(self.h, sid, slot))

    def release_source(self, sid):
        """The source is no longer used: its blocks are dropped, its id is free for the next add_* (odr_source_release)."""
        check(self.lib.odr_source_release(self.h, int(sid)))
        self._grids.pop(sid, None)

    def bind(self, variable, source_ids, fallback=np.nan):
        ids, pi = _i(list(source_ids)) if len(source_ids) else (None, None)
        check(self.lib.odr_env_bind(self.h, _vid(variable), len(source_ids), pi,
                                    np.nan if fallback is None else float(fallback)))

    def history(self, n_trajectories, n_times, variables, id_base=0):
        return History(self, n_trajectories, n_times, variables, id_base=id_base)

    def particles(self, capacity):
        return Particles(self, capacity)

    def timer_begin(self):
        check(self.lib.odr_timer_begin(self.h))

    def timer_end(self):
        ms = C.c_float()
        check(self.lib.odr_timer_end(self.h, C.byref(ms)))
        return ms.value


class Particles:
    def __init__(self, ctx, capacity):
        self.ctx, self.lib = ctx, ctx.lib
        self.h = C.c_void_p()
        check(self.lib.odr_particles_create(ctx.h, int(capacity), C.byref(self.h)))

    def close(self):
        if self.h and self.ctx.h:
            self.lib.odr_particles_destroy(self.ctx.h, self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.count()[0]

    def count(self):
        a, d = C.c_int64(), C.c_int64()
        check(self.lib.odr_particles_count(self.ctx.h, self.h, C.byref(a), C.byref(d)))
        return a.value, d.value

    def append(self, lon, lat, z=None, id=None, moving=None, wind_drift_factor=None,
               current_drift_factor=None, terminal_velocity=None):
        lon = np.atleast_1d(np.asarray(lon, dtype=np.float64))
        n = lon.size
        a = [_d(lon, n), _d(lat, n), _d(z, n), _i(id, n), _i(moving, n), _f(wind_drift_factor, n),
             _f(current_drift_factor, n), _f(terminal_velocity, n)]
        if id is None:                       # the device numbers them (active + deactivated so far) + 0 .. n-1
            act, dead = self.count()
            ids = act + dead + np.arange(n, dtype=np.int64)
        else:
            ids = np.atleast_1d(np.asarray(id, dtype=np.int64))
        check(self.lib.odr_particles_append(self.ctx.h, self.h, n, *[p for _, p in a]))
        # release sequence of every ID: the reference keeps its arrays in release order (move_elements appends the
        # scheduled elements as they are released, elements.py:197-228), which differs from ID order when the seed times
        # are not monotonic in ID; host-drawn random numbers (rng='numpy') arrive in that order (_host_order)
        seq = getattr(self, '_release_seq', None)
        top = int(ids.max()) + 1 if n else 0
        if seq is None or seq.size < top:
            new = np.full(max(top, 2 * (seq.size if seq is not None else 0), 1024), -1, np.int64)
            if seq is not None:
                new[:seq.size] = seq
            seq = self._release_seq = new
        k0 = getattr(self, '_release_count', 0)
        seq[ids] = k0 + np.arange(n)
        self._release_count = k0 + n

    def upload(self, lon=None, lat=None, z=None, moving=None, wind_drift_factor=None,
               current_drift_factor=None, terminal_velocity=None):
        n = len(self)
        a = [_d(lon, n), _d(lat, n), _d(z, n), _i(moving, n), _f(wind_drift_factor, n),
             _f(current_drift_factor, n), _f(terminal_velocity, n)]
        check(self.lib.odr_particles_upload(self.ctx.h, self.h, *[p for _, p in a]))

    def download(self):
        n = len(self)
        out = dict(lon=np.empty(n), lat=np.empty(n), z=np.empty(n), ID=np.empty(n, np.int32),
                   status=np.empty(n, np.int32), moving=np.empty(n, np.int32))
        check(self.lib.odr_particles_download(
            self.ctx.h, self.h, out['lon'].ctypes.data_as(_dp), out['lat'].ctypes.data_as(_dp),
            out['z'].ctypes.data_as(_dp), out['ID'].ctypes.data_as(_ip), out['status'].ctypes.data_as(_ip),
            out['moving'].ctypes.data_as(_ip)))
        return out

    def download_f32(self, name):
        """wind_drift_factor / current_drift_factor / terminal_velocity / age_seconds of the active elements"""
        out = np.empty(len(self), np.float32)
        check(self.lib.odr_particles_download_f32(self.ctx.h, self.h, name.encode(), out.ctypes.data_as(_fp)))
        return out

    def download_deactivated(self):
        n = self.count()[1]
        out = dict(lon=np.empty(n), lat=np.empty(n), z=np.empty(n), ID=np.empty(n, np.int32),
                   status=np.empty(n, np.int32))
        check(self.lib.odr_particles_download_deactivated(
            self.ctx.h, self.h, out['lon'].ctypes.data_as(_dp), out['lat'].ctypes.data_as(_dp),
            out['z'].ctypes.data_as(_dp), out['ID'].ctypes.data_as(_ip), out['status'].ctypes.data_as(_ip)))
        return out

    def device_ptr(self, name):
        p = C.c_void_p()
        check(self.lib.odr_particles_device_ptr(self.ctx.h, self.h, name.encode(), C.byref(p)))
        return p.value

    # ---- hot-path stages ----
    def env_sample(self, variables, t_epoch, download=False):
        ids, pi = _i([_vid(v) for v in variables])
        if download:
            n = len(self)
            outs = [np.empty(n, np.float32) for _ in variables]
            ptrs = (_fp * len(outs))(*[o.ctypes.data_as(_fp) for o in outs])
            check(self.lib.odr_env_sample(self.ctx.h, self.h, len(ids), pi, float(t_epoch), ptrs))
            return dict(zip(variables, outs))
        check(self.lib.odr_env_sample(self.ctx.h, self.h, len(ids), pi, float(t_epoch), None))

    def env_download(self, variable):
        out = np.empty(len(self), np.float32)
        check(self.lib.odr_env_download(self.ctx.h, self.h, _vid(variable), out.ctypes.data_as(_fp)))
        return out

    def env_upload(self, variable, values):
        a, p = _f(values, len(self))
        check(self.lib.odr_env_upload(self.ctx.h, self.h, _vid(variable), p))

    def env_add_noise(self, var_x, var_y, std, step=0, normals=None, uniform=False):
        """drift:current_uncertainty / wind_uncertainty (normal) or drift:current_uncertainty_uniform (uniform=True) on
        the last sample (environment.py:869-891).  normals: the caller's np.random draws (x, y), already scaled."""
        n = len(self)
        dist = _abi.NOISE_UNIFORM if uniform else _abi.NOISE_NORMAL
        if normals is not None:
            (ax, px), (ay, py) = _d(self._host_order(normals[0]), n), _d(self._host_order(normals[1]), n)
            check(self.lib.odr_env_add_noise(self.ctx.h, self.h, _vid(var_x), _vid(var_y), float(std), dist,
                                             _abi.RNG_HOST, px, py, step))
        else:
            check(self.lib.odr_env_add_noise(self.ctx.h, self.h, _vid(var_x), _vid(var_y), float(std), dist,
                                             _abi.RNG_DEVICE, None, None, step))

    def set_advect_noise(self, std_normal=0.0, std_uniform=0.0, step=0, stage_draws=None, main_draws=None):
        """drift:current_uncertainty(_uniform) inside the next advect() / env_coast_advect() (odr_advect_set_noise): the
        reference adds them in the Runge-Kutta stage get_environment calls too (physics_methods.py:638-670).
        stage_draws [nstage][ncomp][n] / main_draws [ncomp][n]: np.random draws in the reference's order (parity mode);
        None: Philox on the device."""
        if stage_draws is None and main_draws is None:
            check(self.lib.odr_advect_set_noise(self.ctx.h, self.h, float(std_normal), float(std_uniform), _abi.RNG_DEVICE,
                                                None, None, 0, step))
            return
        ps = pm = None
        nstage = 0
        if stage_draws is not None:
            sd, ps = _d(np.ascontiguousarray(self._host_order(np.asarray(stage_draws, dtype=np.float64))))
            nstage = sd.shape[0]
        if main_draws is not None:
            md, pm = _d(np.ascontiguousarray(self._host_order(np.asarray(main_draws, dtype=np.float64))))
        check(self.lib.odr_advect_set_noise(self.ctx.h, self.h, float(std_normal), float(std_uniform), _abi.RNG_HOST,
                                            pm, ps, nstage, step))

    def advect(self, scheme, t_epoch, dt, factor=1.0):
        s = scheme if isinstance(scheme, int) else _abi.SCHEME.get(scheme, -1)
        if s < 0:
            raise ValueError('Drift scheme not recognised: ' + str(scheme))
        check(self.lib.odr_advect(self.ctx.h, self.h, s, float(t_epoch), float(dt), float(factor)))

    def env_coast_advect(self, variables, t_epoch, scheme, dt, coastline='none', stranded_code=1,
                         seeded_on_land_code=0, store_previous=True, factor=1.0, count=True, seafloor=False,
                         age_dt=0.0, max_age_seconds=0.0, retired_code=0, missing_code=0, main_noise=False, vmix=None):
        """env_sample -> coastline -> store_previous -> advect in one launch (odr_env_coast_advect).
        count=False skips reading back the number of elements on land (no host synchronisation).  seafloor=True and
        age_dt != 0 add interact_with_seafloor ('lift_to_seafloor') and increase_age_and_retire, in the loop's order."""
        s = scheme if isinstance(scheme, int) else _abi.SCHEME.get(scheme, -1)
        if s < 0:
            raise ValueError('Drift scheme not recognised: ' + str(scheme))
        a = coastline if isinstance(coastline, int) else _abi.COAST[coastline]
        ids, pi = _i([_vid(v) for v in variables])
        n = C.c_int64()
        ex = None
        if seafloor or age_dt or missing_code or main_noise or vmix:
            # vmix = dict(dt_mix=, step=, mix_at_surface=False, vertical_advection=None | False (below the surface) | True):
            # OceanDrift.vertical_mixing (+ vertical_advection) of the same step inside the call (device RNG)
            m = vmix or {}
            va = m.get('vertical_advection')
            ex = C.byref(_abi.StepExtras(1 if seafloor else 0, int(retired_code), float(age_dt), float(max_age_seconds),
                                         int(missing_code), 1 if main_noise else 0, 1 if vmix else 0,
                                         int(bool(m.get('mix_at_surface', False))), -1 if va is None else int(bool(va)), 0,
                                         float(m.get('dt_mix', 0.0)), int(m.get('step', 0))))
        check(self.lib.odr_env_coast_advect(self.ctx.h, self.h, len(ids), pi, float(t_epoch), a, stranded_code,
                                            seeded_on_land_code, int(bool(store_previous)), s, float(dt),
                                            float(factor), ex, C.byref(n) if count else None))
        return n.value if count else None

    def set_rank_offset(self, offset):
        """Present elements with smaller IDs held by other particle sets (the lower ranks of a sharded run): added to the
        rank among the present elements that selects the ensemble member (odr_particles_set_rank_offset)."""
        check(self.lib.odr_particles_set_rank_offset(self.ctx.h, self.h, int(offset)))

    def update_positions(self, x_vel, y_vel, dt):
        n = len(self)
        is32 = int(np.asarray(x_vel).dtype == np.float32)
        (a, pa), (b, pb) = _d(x_vel, n), _d(y_vel, n)
        check(self.lib.odr_update_positions(self.ctx.h, self.h, pa, pb, is32, float(dt)))

    def advect_wind(self, dt, wind_drift_depth=0.1, relative_wind=False, factor=1.0):
        check(self.lib.odr_advect_wind(self.ctx.h, self.h, float(dt), float(wind_drift_depth),
                                       int(relative_wind), float(factor)))

    ELEMENT_FACTORS = {None: 0, 'scalar': 0, 'ice_current': 1, 'ice_stokes': 2, 'ice_drift': 3}

    def set_element_factor(self, kind):
        """Per-element factors of the following movers from the sampled sea_ice_area_fraction (OpenOil.advect_oil,
        openoil.py:1179-1216): 'ice_current' (advect, advect_wind), 'ice_stokes' (stokes_drift), 'ice_drift'
        (advect_sea_ice); None: the scalar `factor` arguments again."""
        check(self.lib.odr_set_element_factor(self.ctx.h, self.h, self.ELEMENT_FACTORS[kind]))

    def advect_sea_ice(self, dt, factor=1.0):
        check(self.lib.odr_advect_sea_ice(self.ctx.h, self.h, float(dt), float(factor)))

    def stokes_drift(self, dt, profile=2, hs_mode=0, tp_mode=0, factor=1.0):
        check(self.lib.odr_stokes_drift(self.ctx.h, self.h, float(dt), profile, hs_mode, tp_mode, float(factor)))

    def set_property(self, slot, values, offset=0):
        a, pa = _f(np.atleast_1d(values))
        check(self.lib.odr_particles_set_property(self.ctx.h, self.h, slot, int(offset), a.size, pa))

    def get_property(self, slot):
        out = np.empty(len(self), np.float32)
        check(self.lib.odr_particles_get_property(self.ctx.h, self.h, slot, out.ctypes.data_as(_fp)))
        return out

    def ids(self):
        n = len(self)
        out = np.empty(n, np.int32)
        check(self.lib.odr_particles_download(self.ctx.h, self.h, None, None, None, out.ctypes.data_as(_ip), None, None))
        return out

    def _ref_key(self, ids):
        """Sort key that puts elements in the reference's array order: the release sequence (ID where unknown)."""
        seq = getattr(self, '_release_seq', None)
        if seq is not None and ids.size and ids.max() < seq.size and (seq[ids] >= 0).all():
            return seq[ids]
        return ids

    def _host_order(self, arr):
        """Host-drawn random numbers (RNG_HOST parity mode) arrive in the reference's element order,
        i.e. release order of the active elements (= ascending ID unless the seed times are not monotonic in ID); in-place compaction and spatial sorting permute the
        device arrays, so the numbers are gathered into device order first."""
        n = len(self)
        a = np.asarray(arr)[..., :n]
        if not getattr(self, '_permuted', False):
            return a
        key = self._ref_key(self.ids())
        rank = np.empty(n, np.int64)
        rank[np.argsort(key, kind='stable')] = np.arange(n)
        return np.ascontiguousarray(a[..., rank])

    def leeway_capsize(self, dt, wind_threshold=30.0, wind_threshold_sigma=5.0, step=0, uniforms=None):
        """processes:capsizing (leeway.py:438-455).  uniforms (RNG_HOST parity mode) = np.random.rand(len(can_be_capsized))
        in the reference's order (ascending ID of the elements that can be capsized)."""
        if uniforms is not None:
            n = len(self)
            cap = self.get_property(8)
            can = np.nonzero(cap == (0.0 if dt >= 0 else 1.0))[0]
            order = can[np.argsort(self._ref_key(self.ids()[can]), kind='stable')]      # reference order of those elements
            full = np.zeros(n)
            full[order] = np.asarray(uniforms, dtype=np.float64)[:len(order)]
            a, pa = _d(full, n)
            check(self.lib.odr_leeway_capsize(self.ctx.h, self.h, float(dt), float(wind_threshold),
                                              float(wind_threshold_sigma), _abi.RNG_HOST, pa, step))
        else:
            check(self.lib.odr_leeway_capsize(self.ctx.h, self.h, float(dt), float(wind_threshold),
                                              float(wind_threshold_sigma), _abi.RNG_DEVICE, None, step))

    def leeway(self, dt, capsize_fraction=0.4, step=0, uniforms=None):
        if uniforms is not None:
            u, pu = _d(self._host_order(uniforms), len(self))
            check(self.lib.odr_leeway(self.ctx.h, self.h, float(dt), float(capsize_fraction), _abi.RNG_HOST, pu, step))
        else:
            check(self.lib.odr_leeway(self.ctx.h, self.h, float(dt), float(capsize_fraction), _abi.RNG_DEVICE, None, step))

    def env_coast_leeway(self, variables, t_epoch, dt, capsize_fraction=0.4, coastline='stranding', stranded_code=1,
                         seeded_on_land_code=0, store_previous=False, current_uncertainty=0.0, wind_uncertainty=0.0, step=0,
                         split='finish', missing_code=0, count=True):
        """The Leeway loop body between two compactions in one launch (odr_env_coast_leeway): sample, uncertainties (device
        RNG), coastline, previous state, Leeway.update.  When wind / current / landmask do not come from one gridded reader the
        library has sampled, perturbed and applied the coastline only: compaction and Leeway.update follow here, so that
        env_coast_leeway() + compact() is the same sequence either way.  Returns the number of elements on land.
        split='return': in that case nothing more is done here and (elements on land, True) comes back -- the caller compacts
        (after whatever it does between coastline and update) and calls leeway() itself; (elements on land, False) otherwise."""
        ids, pi = _i([_vid(v) for v in variables])
        nhit = C.c_int64()
        if missing_code:     # report_missing_variables inside the call (status of an element without data)
            check(self.lib.odr_leeway_set_missing_code(self.ctx.h, int(missing_code)))
        rc = self.lib.odr_env_coast_leeway(self.ctx.h, self.h, len(variables), pi, float(t_epoch), _abi.COAST[coastline],
                                           int(stranded_code), int(seeded_on_land_code), int(bool(store_previous)), float(dt),
                                           float(capsize_fraction), float(current_uncertainty), float(wind_uncertainty), int(step),
                                           C.byref(nhit) if count else None)      # count=False: no host synchronisation
        if rc == 1:          # ODR_SPLIT_LANE
            if split == 'return':
                return nhit.value, True
            self.compact()
            self.leeway(dt, capsize_fraction, step=step)
        else:
            check(rc)
        return (nhit.value, False) if split == 'return' else nhit.value

    def hdiffusion(self, dt, step=0, normals=None):
        n = len(self)
        if normals is not None:
            (ax, px), (ay, py) = _d(self._host_order(normals[0]), n), _d(self._host_order(normals[1]), n)
            check(self.lib.odr_hdiffusion(self.ctx.h, self.h, float(dt), _abi.RNG_HOST, px, py, step))
        else:
            check(self.lib.odr_hdiffusion(self.ctx.h, self.h, float(dt), _abi.RNG_DEVICE, None, None, step))

    def movers(self, dt, wind=None, stokes=None, hdiffusion=None):
        """advect_wind -> stokes_drift -> horizontal_diffusion in one launch (odr_movers).  Each argument is None (mover not
        applied) or the keyword arguments of the method of that name: wind=dict(wind_drift_depth=0.1, relative_wind=False,
        factor=1.0), stokes=dict(profile=2, hs_mode=0, tp_mode=0, factor=1.0), hdiffusion=dict(step=0, normals=None)."""
        which = (1 if wind is not None else 0) | (2 if stokes is not None else 0) | (4 if hdiffusion is not None else 0)
        w, s, h = wind or {}, stokes or {}, hdiffusion or {}
        px = py = None
        mode = _abi.RNG_DEVICE
        if h.get('normals') is not None:
            n = len(self)
            (ax, px), (ay, py) = _d(self._host_order(h['normals'][0]), n), _d(self._host_order(h['normals'][1]), n)
            mode = _abi.RNG_HOST
        check(self.lib.odr_movers(self.ctx.h, self.h, float(dt), which, float(w.get('wind_drift_depth', 0.1)),
                                  int(bool(w.get('relative_wind', False))), float(w.get('factor', 1.0)),
                                  int(s.get('profile', 2)), int(s.get('hs_mode', 0)), int(s.get('tp_mode', 0)),
                                  float(s.get('factor', 1.0)), mode, px, py, int(h.get('step', 0))))

    def snapshot_property(self, slot):
        """Device copy of one property as it is now, read by the next History.record(position_from_previous=3)."""
        check(self.lib.odr_particles_snapshot_property(self.ctx.h, self.h, int(slot)))

    def vmix(self, t_epoch, dt, dt_mix, mix_at_surface=False, step=0, uniforms=None, fuse_vertical_advection=None, guarded=False,
             profile_levels=0):
        """guarded=True (between scan_status_begin and scan_status_end): the launch does nothing unless the fold finds that every
        element stays (odr_ctx_guard_next_vmix); returns False when the library could not launch it that way (nothing happened)."""
        if fuse_vertical_advection is not None:   # True: include surface elements, False: z<0 only
            check(self.lib.odr_vmix_fuse_vertical_advection(self.ctx.h, int(bool(fuse_vertical_advection))))
        if profile_levels:     # the K columns end there (a reader that cut its block at the depth asked of it, odr_vmix_set_profile_levels)
            check(self.lib.odr_vmix_set_profile_levels(self.ctx.h, int(profile_levels)))
        if guarded:
            assert uniforms is None
            check(self.lib.odr_ctx_guard_next_vmix(self.ctx.h, 1))
            rc = self.lib.odr_vmix(self.ctx.h, self.h, float(t_epoch), float(dt), float(dt_mix), int(mix_at_surface), _abi.RNG_DEVICE, None, step)
            if rc < 0:
                check(rc)
            return rc == 0
        if uniforms is not None:
            u, pu = _d(np.ascontiguousarray(self._host_order(uniforms)))
            check(self.lib.odr_vmix(self.ctx.h, self.h, float(t_epoch), float(dt), float(dt_mix),
                                    int(mix_at_surface), _abi.RNG_HOST, pu, step))
        else:
            check(self.lib.odr_vmix(self.ctx.h, self.h, float(t_epoch), float(dt), float(dt_mix),
                                    int(mix_at_surface), _abi.RNG_DEVICE, None, step))

    def vmix_analytic(self, model, background_diffusivity, dt, dt_mix, mix_at_surface=False, step=0, uniforms=None,
                      fuse_vertical_advection=None):
        """vertical_mixing with 'windspeed_Large1994' / 'windspeed_Sundby1983' profiles (oceandrift.py:385-395,448-458)."""
        if model not in _abi.DIFFUSIVITY:
            raise ValueError('Unknown diffusivity model: ' + str(model))      # oceandrift.py:395
        if fuse_vertical_advection is not None:
            check(self.lib.odr_vmix_fuse_vertical_advection(self.ctx.h, int(bool(fuse_vertical_advection))))
        pu, mode = None, _abi.RNG_DEVICE
        if uniforms is not None:
            u, pu = _d(np.ascontiguousarray(self._host_order(uniforms)))
            mode = _abi.RNG_HOST
        check(self.lib.odr_vmix_wind_profile(self.ctx.h, self.h, _abi.DIFFUSIVITY[model], float(background_diffusivity),
                                             float(dt), float(dt_mix), int(mix_at_surface), mode, pu, step))

    def oil_prepare_mixing(self, dt, dt_mix, interfacial_tension, distribution, sea_water_density, keep_droplet_diameter=False,
                           hs_mode=1, tp_mode=3, temperature_to_kelvin=True, step=0, uniforms=None):
        """OpenOil.prepare_vertical_mixing on the device (openoil.py:1017-1031); the next vmix / vmix_analytic call runs
        OpenOil's version of the mixing loop.  uniforms (np.random parity): dict(diameter [n], entrain [nt, n],
        intrusion [nt, n])."""
        if distribution not in _abi.DROPLETS:
            raise ValueError('no wave entrainment droplet size distribution specified')      # openoil.py:1070
        pd = pe = pi = None
        mode = _abi.RNG_DEVICE
        if uniforms is not None:
            d, pd = _d(np.ascontiguousarray(self._host_order(uniforms['diameter'])))
            e, pe = _d(np.ascontiguousarray(self._host_order(uniforms['entrain'])))
            i, pi = _d(np.ascontiguousarray(self._host_order(uniforms['intrusion'])))
            mode = _abi.RNG_HOST
        check(self.lib.odr_oil_prepare_mixing(self.ctx.h, self.h, float(dt), float(dt_mix), float(interfacial_tension),
                                              float(sea_water_density), _abi.DROPLETS[distribution],
                                              int(bool(keep_droplet_diameter)), int(hs_mode), int(tp_mode),
                                              int(bool(temperature_to_kelvin)), mode, pd, pe, pi, step))

    def oil_global_stats(self, allreduce_sum, interfacial_tension, distribution, sea_water_density, hs_mode=1):
        """Sharded run: np.mean(dV_50) and np.mean(1.5 Hs) over the elements of ALL ranks, installed for the next
        oil_prepare_mixing (odr_oil_local_sums / odr_oil_set_mixing_stats)."""
        a, b = C.c_double(), C.c_double()
        check(self.lib.odr_oil_local_sums(self.ctx.h, self.h, float(interfacial_tension), float(sea_water_density),
                                          _abi.DROPLETS[distribution], int(hs_mode), C.byref(a), C.byref(b)))
        g = allreduce_sum([a.value, b.value, float(len(self))])
        if g[2] > 0:
            check(self.lib.odr_oil_set_mixing_stats(self.ctx.h, g[1] / g[2], g[0] / g[2]))

    def oil_mixing_stats(self):
        a, b = C.c_double(), C.c_double()
        check(self.lib.odr_oil_mixing_stats(self.ctx.h, C.byref(a), C.byref(b)))
        return dict(mean_zb=a.value, dV_50=b.value)

    def vmix_oil(self, model, background_diffusivity, dt, dt_mix, interfacial_tension, distribution, sea_water_density=None,
                 t_epoch=0.0, uniforms=None, step=0, mix_at_surface=False, profile_levels=0, **kw):
        """prepare_vertical_mixing + vertical_mixing as OpenOil.update runs them (openoil.py:1228-1231).  model: a wind
        parameterisation of the diffusivity or 'environment' (profiles from a reader)."""
        if sea_water_density is None:
            sea_water_density = sea_water_density_default()
        self.oil_prepare_mixing(dt, dt_mix, interfacial_tension, distribution, sea_water_density, step=step,
                                uniforms=uniforms, **kw)
        mix = None if uniforms is None else uniforms['mix']
        if model in ('environment', 'constant'):
            self.vmix(t_epoch, dt, dt_mix, mix_at_surface=mix_at_surface, step=step, uniforms=mix,
                      profile_levels=profile_levels if model == 'environment' else 0)
        else:
            self.vmix_analytic(model, background_diffusivity, dt, dt_mix, mix_at_surface=mix_at_surface, step=step, uniforms=mix)

    def vertical_advection(self, dt, at_surface=False):
        check(self.lib.odr_vertical_advection(self.ctx.h, self.h, float(dt), int(at_surface)))

    def vertical_buoyancy(self, dt):
        check(self.lib.odr_vertical_buoyancy(self.ctx.h, self.h, float(dt)))

    def store_previous(self):
        check(self.lib.odr_store_previous(self.ctx.h, self.h))

    def coastline(self, action, stranded_code=1, seeded_on_land_code=0):
        a = action if isinstance(action, int) else _abi.COAST[action]
        n = C.c_int64()
        check(self.lib.odr_coastline(self.ctx.h, self.h, a, stranded_code, seeded_on_land_code, C.byref(n)))
        return n.value

    def coastline_crossing(self, action, precision, landmask_source, stranded_code=1, seeded_on_land_code=0):
        """interact_with_coastline with general:coastline_approximation_precision (basemodel/__init__.py:694-746)."""
        a = action if isinstance(action, int) else _abi.COAST[action]
        n = C.c_int64()
        check(self.lib.odr_coastline_crossing(self.ctx.h, self.h, a, stranded_code, seeded_on_land_code, float(precision),
                                              int(landmask_source), C.byref(n)))
        return n.value

    def increase_age(self, dt, max_age_seconds=0.0, retired_code=0):
        check(self.lib.odr_increase_age(self.ctx.h, self.h, float(dt), float(max_age_seconds), retired_code))

    def deactivate_missing(self, variables, status_code):
        """report_missing_variables (basemodel/__init__.py:2501-2515) on the last environment sample."""
        ids, pi = _i([_vid(v) for v in variables])
        n = C.c_int64()
        check(self.lib.odr_deactivate_missing(self.ctx.h, self.h, len(ids), pi, int(status_code), C.byref(n)))
        return n.value

    def count_status(self, status_code):
        n = C.c_int64()
        check(self.lib.odr_particles_count_status(self.ctx.h, self.h, int(status_code), C.byref(n)))
        return n.value

    def remap_status(self, from_code, to_code):
        check(self.lib.odr_particles_remap_status(self.ctx.h, self.h, int(from_code), int(to_code)))

    def seafloor(self, action='lift_to_seafloor', status_code=0):
        """interact_with_seafloor (basemodel/__init__.py:748-783): 'lift_to_seafloor' | 'deactivate' | 'previous'."""
        n = C.c_int64()
        a = {'lift_to_seafloor': 1, 'deactivate': 2, 'previous': 3}[action]
        check(self.lib.odr_seafloor_action(self.ctx.h, self.h, a, int(status_code), C.byref(n)))
        return n.value

    def deactivate(self, mask, status_code):
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        check(self.lib.odr_deactivate(self.ctx.h, self.h, m.ctypes.data_as(C.POINTER(C.c_uint8)), status_code))

    def deactivate_outside(self, west, east, south, north, status_code):
        f = lambda v: np.nan if v is None else float(v)
        check(self.lib.odr_deactivate_outside(self.ctx.h, self.h, f(west), f(east), f(south), f(north), int(status_code)))

    def compact(self):
        """Remove the deactivated elements (remove_deactivated_elements).  In place: the survivors are
        permuted (holes filled from the tail); elements are identified by their ID."""
        n0 = len(self)
        n = C.c_int64()
        check(self.lib.odr_compact(self.ctx.h, self.h, C.byref(n)))
        if n.value != n0:
            self._permuted = True
        return n.value

    def scan_status(self):
        """(elements that stay, provisional status numbers present as a bit mask: bit k <-> status 100 + k) in one host
        read (odr_scan_status); follow with compact_apply()."""
        n, f = C.c_int64(), C.c_uint64()
        check(self.lib.odr_scan_status(self.ctx.h, self.h, C.byref(n), C.byref(f)))
        return n.value, f.value

    def scan_status_begin(self):
        """First half of scan_status (odr_scan_status_begin): the fold of the counts the step launch left is enqueued; False
        when that launch left none (then call scan_status())."""
        rc = self.lib.odr_scan_status_begin(self.ctx.h, self.h)
        if rc < 0:
            check(rc)
        return rc == 0

    def scan_status_end(self):
        """Second half: waits for the fold only (not for a guarded mixing launch enqueued behind it)."""
        n, f = C.c_int64(), C.c_uint64()
        check(self.lib.odr_scan_status_end(self.ctx.h, self.h, C.byref(n), C.byref(f)))
        return n.value, f.value

    def compact_apply(self):
        n0 = len(self)
        n = C.c_int64()
        check(self.lib.odr_compact_apply(self.ctx.h, self.h, C.byref(n)))
        if n.value != n0:
            self._permuted = True
        return n.value

    def sort_by_cell(self, source_id, keep_environment=True):
        """Re-order the SoA by grid cell of a gridded source (layout only; IDs are preserved).  keep_environment=False:
        the sampled environment is not carried along (a re-sort right before the next sample)."""
        check(self.lib.odr_sort_particles_ex(self.ctx.h, self.h, int(source_id), int(bool(keep_environment))))
        self._permuted = True

    def truncate_z(self, depth):
        """The sampling calls that follow see max(z, -depth) (drift:truncate_ocean_model_below_m); restore_z() ends it."""
        check(self.lib.odr_particles_truncate_z(self.ctx.h, self.h, float(depth)))

    def restore_z(self):
        check(self.lib.odr_particles_restore_z(self.ctx.h, self.h))

    def tile_stats(self):
        """Diagnostics of the LDS-tile step (odr_particles_tile_stats): launches on that path, elements they handed to the
        HBM path, rectangles cut to the LDS capacity, workgroup ranges of the current table."""
        import ctypes as C
        out = (C.c_uint64 * 4)()
        check(self.lib.odr_particles_tile_stats(self.ctx.h, self.h, out))
        return dict(launches=int(out[0]), handed_over=int(out[1]), rectangles_cut=int(out[2]), ranges=int(out[3]))

    def reduce_global(self, combine, wind_drift_depth=0.1, relative_wind=False):
        """Sharded run: this set's raw reductions -> combine(raw16) over the ranks (counts summed, maxima maximised) ->
        installed for the movers that follow, until reduce_unpin()."""
        raw = np.empty(16)
        check(self.lib.odr_reduce_local(self.ctx.h, self.h, float(wind_drift_depth), int(relative_wind), raw.ctypes.data_as(_dp)))
        g = np.ascontiguousarray(combine(raw), dtype=np.float64)
        check(self.lib.odr_reduce_install(self.ctx.h, self.h, g.ctypes.data_as(_dp)))
        return g

    def reduce_local(self, wind_drift_depth=0.1, relative_wind=False):
        """The 16 raw reduction slots of THIS rank's active elements (status 0; odr_reduce_local): slots 0 and 11 are
        counts, the others maxima (minima negated).  Combined over the ranks by the caller, then reduce_install()."""
        raw = np.empty(16)
        check(self.lib.odr_reduce_local(self.ctx.h, self.h, float(wind_drift_depth), int(relative_wind), raw.ctypes.data_as(_dp)))
        return raw

    def reduce_install(self, combined16):
        g = np.ascontiguousarray(combined16, dtype=np.float64)
        check(self.lib.odr_reduce_install(self.ctx.h, self.h, g.ctypes.data_as(_dp)))

    @staticmethod
    def reduction_dict(raw):
        keys = ['n_active', 'lon_min', 'lon_max', 'lat_min', 'lat_max', 'z_min', 'z_max', 'D_max',
                'stokes_sum_max', 'wind_speed_max', 'wdf_surface_max', 'n_surface', 'hs_max', 'tp_max']
        d = dict(zip(keys, raw))
        for k in ('lon_min', 'lat_min', 'z_min'):
            d[k] = -d[k]
        return d

    def reduce_unpin(self):
        check(self.lib.odr_reduce_unpin(self.ctx.h))

    def reduce_scalars(self, wind_drift_depth=0.1):
        out = np.empty(16)
        check(self.lib.odr_reduce_scalars(self.ctx.h, self.h, float(wind_drift_depth), out.ctypes.data_as(_dp)))
        keys = ['n_active', 'lon_min', 'lon_max', 'lat_min', 'lat_max', 'z_min', 'z_max', 'D_max',
                'stokes_sum_max', 'wind_speed_max', 'wdf_surface_max', 'n_surface', 'hs_max', 'tp_max']
        return dict(zip(keys, out))


_SLOW_MS = float(os.environ.get('ODR_SLOW_CALLS', 0) or 0)


def _touching(fn):
    """A call that changes the device state: what a model wrote into its `o.elements` view goes to the device first
    (oceandrift.ElementsView.flush), and views taken before the call become stale."""
    def wrapper(self, *a, **kw):
        v = self.__dict__.get('_view')
        if v is not None:
            self._view = None
            v.flush()
        self._touch = self.__dict__.get('_touch', 0) + 1
        if _SLOW_MS:     # ODR_SLOW_CALLS=<ms>: calls that keep the host longer than that (stderr)
            import sys
            import time
            t0 = time.perf_counter()
            r = fn(self, *a, **kw)
            ms = 1e3 * (time.perf_counter() - t0)
            if ms > _SLOW_MS:
                print('%s: %.3f ms' % (fn.__name__, ms), file=sys.stderr)
            return r
        return fn(self, *a, **kw)
    wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
    return wrapper


for _name in ('append', 'upload', 'env_sample', 'env_upload', 'env_add_noise', 'advect', 'env_coast_advect',
              'update_positions', 'advect_wind', 'stokes_drift', 'advect_sea_ice', 'set_property', 'leeway_capsize', 'leeway', 'hdiffusion', 'movers',
              'vmix', 'vmix_analytic', 'vmix_oil', 'vertical_advection', 'vertical_buoyancy', 'coastline', 'coastline_crossing',
              'increase_age', 'deactivate_missing', 'remap_status', 'seafloor', 'deactivate', 'deactivate_outside', 'compact',
              'compact_apply', 'sort_by_cell', 'store_previous', 'oil_prepare_mixing', 'env_coast_leeway'):
    setattr(Particles, _name, _touching(getattr(Particles, _name)))


def _reading(fn):
    """A call that only READS the device state (downloads, reductions, scans): pending writes of the `o.elements` view
    reach the device first -- `self.elements.z = ...` inside update() must be seen by the reductions of stokes_drift and
    by the stored previous positions -- but the view stays valid (nothing changed underneath it)."""
    def wrapper(self, *a, **kw):
        v = self.__dict__.get('_view')
        if v is not None and not getattr(v, '_flushing', False):
            touch = self.__dict__.get('_touch', 0)
            v.flush()                    # uploads what differs (a _touching call) ...
            self._touch = touch          # ... which does not make the view stale: it holds what it just wrote
            self._view = v
        return fn(self, *a, **kw)
    wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
    return wrapper


for _name in ('download', 'download_f32', 'env_download', 'get_property', 'reduce_scalars', 'reduce_global', 'reduce_local', 'oil_global_stats',
              'scan_status', 'count_status'):
    setattr(Particles, _name, _reading(getattr(Particles, _name)))


class History:
    """Device-resident float32 result buffer (state_to_buffer, basemodel/__init__.py:2084-2105,2384-2499):
    `variables` = element property names ('lon', 'lat', 'z', 'status', ...), environment variable names, or
    ('property', slot).  record() scatters the current state at (ID, time index); flush() copies time slots to
    pinned host memory asynchronously; array(var) returns the [trajectory, time] float32 view of the last flush."""

    def __init__(self, ctx, n_trajectories, n_times, variables, id_base=0):
        self.ctx, self.lib = ctx, ctx.lib
        self.variables = list(variables)
        self._id_base = int(id_base)
        self.n_trajectories, self.n_times = int(n_trajectories), int(n_times)
        codes = []
        for v in self.variables:
            if isinstance(v, tuple):
                codes.append(_abi.HIST_PROPERTY0 + int(v[1]))
            elif v in _abi.HIST:
                codes.append(_abi.HIST[v])
            else:
                codes.append(_vid(v))
        a, pa = _i(codes)
        self.h = C.c_void_p()
        check(self.lib.odr_history_create(ctx.h, self.n_trajectories, self.n_times, len(codes), pa, C.byref(self.h)))
        if self._id_base:
            check(self.lib.odr_history_set_id_base(ctx.h, self.h, self._id_base))

    def record(self, particles, time_index, only_deactivated=False, position_from_previous=False):
        check(self.lib.odr_history_record(self.ctx.h, particles.h, self.h, int(time_index), int(bool(only_deactivated)),
                                          int(position_from_previous)))      # bit 0: previous position, bit 1: snapshot properties

    def flush(self, t0=0, nt=None):
        check(self.lib.odr_history_flush(self.ctx.h, self.h, int(t0), int(self.n_times - t0 if nt is None else nt)))

    def wait(self):
        check(self.lib.odr_history_wait(self.ctx.h, self.h))

    def array(self, variable):
        """[trajectory, time] float32 view of pinned host memory (valid until the next flush)."""
        k = self.variables.index(variable)
        p, nt = _fp(), C.c_int32()
        check(self.lib.odr_history_host_ptr(self.ctx.h, self.h, k, C.byref(p), C.byref(nt)))
        return np.ctypeslib.as_array(p, shape=(self.n_trajectories, nt.value))

    def minmax(self, variable):
        lo, hi = C.c_double(), C.c_double()
        check(self.lib.odr_history_minmax(self.ctx.h, self.h, self.variables.index(variable), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def reset(self):
        check(self.lib.odr_history_reset(self.ctx.h, self.h))

    def close(self):
        if self.h:
            self.lib.odr_history_destroy(self.ctx.h, self.h)
            self.h = None


class SigmaGrid:
    """ROMS s-coordinate grid on the device (odr_sgrid_*): depths of the rho s-levels (roppy sdepth) and the
    sigma -> z regridding of 3-D variables (roppy multi_zslice) as done by reader_ROMS_native.get_variables."""

    def __init__(self, ctx, H, Hc, Cs_r, zeta=None, S=None, Vtransform=1):
        self.ctx, self.lib = ctx, ctx.lib
        H = np.ascontiguousarray(H, dtype=np.float64)
        self.ny, self.nx = H.shape
        Cs = np.ascontiguousarray(Cs_r, dtype=np.float64)
        self.N = len(Cs)
        ze = None if zeta is None else np.ascontiguousarray(zeta, dtype=np.float64)
        Sa = None if S is None else np.ascontiguousarray(S, dtype=np.float64)
        self.h = C.c_void_p()
        check(self.lib.odr_sgrid_create(ctx.h, self.ny, self.nx, self.N, H.ctypes.data_as(_dp),
                                        None if ze is None else ze.ctypes.data_as(_dp), float(Hc), Cs.ctypes.data_as(_dp),
                                        None if Sa is None else Sa.ctypes.data_as(_dp), int(Vtransform), C.byref(self.h)))

    def z_rho(self):
        out = np.empty((self.N, self.ny, self.nx))
        check(self.lib.odr_sgrid_download_zrho(self.ctx.h, self.h, out.ctypes.data_as(_dp)))
        return out

    def zslice(self, field, Z, want_float64=False, slot=0):
        """field [N, ny, nx] (float32 / float64 host array, or an int device pointer to float32) -> device pointer
        of the float32 [len(Z), ny, nx] result in result slot `slot` (0..7: one per variable of a block, for
        Context.upload_block_device) and, if asked, the float64 values."""
        Z = np.ascontiguousarray(np.atleast_1d(Z), dtype=np.float64)
        out64 = np.empty((len(Z), self.ny, self.nx)) if want_float64 else None
        dev = C.c_void_p()
        if isinstance(field, (int, np.integer)):
            check(self.lib.odr_sgrid_zslice(self.ctx.h, self.h, C.c_void_p(int(field)), 0, 1, len(Z), Z.ctypes.data_as(_dp),
                                            int(slot), C.byref(dev), None if out64 is None else out64.ctypes.data_as(_dp)))
        else:
            f = np.ascontiguousarray(field)
            if f.dtype not in (np.float32, np.float64):
                f = f.astype(np.float64)
            assert f.shape == (self.N, self.ny, self.nx), (f.shape, (self.N, self.ny, self.nx))
            check(self.lib.odr_sgrid_zslice(self.ctx.h, self.h, f.ctypes.data_as(C.c_void_p), int(f.dtype == np.float64), 0,
                                            len(Z), Z.ctypes.data_as(_dp), int(slot), C.byref(dev),
                                            None if out64 is None else out64.ctypes.data_as(_dp)))
        return (dev.value, out64) if want_float64 else dev.value

    def close(self):
        if self.h:
            self.lib.odr_sgrid_destroy(self.ctx.h, self.h)
            self.h = None
