"""Builds the same scenario twice: as an oracle world (CPU checker) and as a device context.

A scenario description is a list of sources + per-variable priority lists + fallbacks:
    sources: [('constant', {name: value}), ('double_gyre', dict(A=..)), ('grid', dict(...))]
"""
import numpy as np

from oracle import oracle as orc

V = orc.VAR


def _orc_proj(proj):
    if proj is None or proj.get('kind', 'latlong') == 'latlong':
        return orc.make_proj()
    rf = proj.get('rf', 0.0)
    f = 0.0 if not rf else 1.0 / rf
    es = proj.get('es', f * (2 - f))
    kind = {'stere_equit_sphere': orc.PROJ_STERE_EQUIT_SPHERE, 'stere_polar': orc.PROJ_STERE_POLAR,
            'merc': orc.PROJ_MERC, 'lcc': orc.PROJ_LCC, 'tmerc': orc.PROJ_TMERC, 'laea': orc.PROJ_LAEA,
            'stere_oblique': orc.PROJ_STERE_OBLIQUE, 'ob_tran': orc.PROJ_OB_TRAN}[proj['kind']]
    return orc.make_proj(kind, a=proj.get('a', 6378137.0), es=es, lat0=proj.get('lat0', 0.0),
                         lon0=proj.get('lon0', 0.0), lat_ts=proj.get('lat_ts', 90.0), k0=proj.get('k0', 1.0),
                         x0=proj.get('x0', 0.0), y0=proj.get('y0', 0.0), lat1=proj.get('lat1', 0.0), lat2=proj.get('lat2'))


def reference_lon_mode(a):
    """modulate_longitude's branch for a grid source (variables.py:259-280): the true longitudes of the four corners of the
    reader's domain -- some negative: np.mod(lon + 180, 360) - 180 (1), else np.mod(lon, 360) (2).  In float64 the two agree
    on [0, 180); in the float32 arithmetic of a run's first get_environment (elements.py:71-88) they do not."""
    if a.get('lon_mode') is not None:
        return a['lon_mode']
    x, y = np.asarray(a['x'], dtype=np.float64), np.asarray(a['y'], dtype=np.float64)
    xx = np.array([x.min(), x.min(), x.max(), x.max()])
    yy = np.array([y.min(), y.max(), y.max(), y.min()])
    proj = a.get('proj')
    if proj is None or proj.get('kind', 'latlong') == 'latlong':
        exlons = xx
    else:
        exlons, _ = orc.proj_inv(_orc_proj(proj), xx, yy)
    return 1 if np.min(exlons) < 0 else 2


class Scenario:
    def __init__(self, sources, fallbacks=None, priority=None):
        self.sources, self.fallbacks, self.priority = sources, fallbacks or {}, priority or {}

    def oracle_world(self):
        wb = orc.WorldBuilder()
        for kind, a in self.sources:
            if kind == 'constant':
                wb.add_constant({V[k]: v for k, v in a.items()})
            elif kind == 'double_gyre':
                wb.add_double_gyre(**a)
            elif kind == 'oscillating':
                wb.add_oscillating(V[a['variable']], a['amplitude'], a['period_s'], a['t0'])
            elif kind == 'grid':
                levels = [(t, {V[k]: arr for k, arr in arrays.items()}) for t, arrays in a['levels']]
                wb.add_grid(_orc_proj(a.get('proj')), a['x'], a['y'], levels, z=a.get('z'),
                            lon_mode=reference_lon_mode(a), mod360_x=a.get('mod360_x', 0),
                            time_coverage=a.get('time_coverage'))
        for k, v in self.fallbacks.items():
            wb.set_fallback(V[k], v)
        for k, ids in self.priority.items():
            wb.set_priority(V[k], ids)
        self._wb = wb
        return wb.finish()

    def device(self, ctx):
        """Registers sources on a device Context with the same priority lists."""
        lists = {}
        for kind, a in self.sources:
            if kind == 'constant':
                sid = ctx.add_constant(a)
                names = list(a)
            elif kind == 'double_gyre':
                sid = ctx.add_double_gyre(**a)
                names = ['x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask']
            elif kind == 'landmask':
                sid = ctx.add_landmask(a['lon0'], a['lat0'], a['dlon'], a['dlat'], a['cells'])
                names = ['land_binary_mask']
                ctx.landmask_sid = sid
            elif kind == 'oscillating':
                sid = ctx.add_oscillating(a['variable'], a['amplitude'], a['period_s'], a['t0'])
                names = [a['variable']]
            else:
                sid = ctx.add_grid(a['x'], a['y'], z=a.get('z'), proj=a.get('proj'),
                                   lon_mode=reference_lon_mode(a), mod360_x=a.get('mod360_x', 0))
                for slot, (t, arrays) in enumerate(a['levels']):
                    ctx.upload_block(sid, slot, t, arrays)
                if a.get('time_coverage') is not None:
                    ctx.set_time_coverage(sid, *a['time_coverage'])
                names = list(a['levels'][0][1])
            for n in names:
                lists.setdefault(n, []).append(sid)
        for k, ids in self.priority.items():
            lists[k] = list(ids)
        for n in set(list(lists) + list(self.fallbacks)):
            ctx.bind(n, lists.get(n, []), self.fallbacks.get(n, np.nan))
        return ctx
