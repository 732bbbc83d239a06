"""OceanDrift model API on the MI355X hot path.

Mirrors the reference's model surface: the constructor, set_config /
get_config, add_reader, seed_elements, run, the live state (`elements`, `environment`, `time`,
`steps_calculation`, num_elements_*), `status_categories`, and the overridable hooks
(`update`, `prepare_run`, `advect_ocean_current`, `advect_wind`, `stokes_drift`,
`vertical_mixing`, `update_positions`, `horizontal_diffusion`, `interact_with_coastline`, ...)
with the reference's names and argument meaning:

    opendrift/models/basemodel/__init__.py  OpenDriftSimulation (run loop :2193-2284)
    opendrift/models/oceandrift.py          OceanDrift.update (:185-211), required_variables (:61-83)
    opendrift/models/physics_methods.py     advect_ocean_current / advect_wind / stokes_drift

Every per-particle operation is a call into libodrift_hip.so; this file is orchestration only.
Out of scope (host-side features of the reference that are not on the path): plotting, netCDF
export, the GSHHG global landmask (`general:use_auto_landmask`), lazy readers.
"""
import logging
import time
from datetime import datetime, timedelta
from types import SimpleNamespace

import os

import numpy as np

from . import _abi
from .config import Configurable, CONFIG_LEVEL_BASIC, CONFIG_LEVEL_ADVANCED, CONFIG_LEVEL_ESSENTIAL
from .device import Context
from .readers import BaseReader, ConstantReader, DeviceReaderBinding, ReaderLevelsError, _epoch

logger = logging.getLogger('opendrift_amd')


class WrongMode(Exception):
    """opendrift/errors.py:1-3"""


class OpenDriftSimulation(Configurable):
    required_variables = {}
    element_properties = {}   # name -> default (in addition to the LagrangianArray core variables)

    # default of seed:ocean_only (the reference's: True, basemodel/__init__.py:424).  The golden vectors of tests/ were
    # written by the reference's loop body without the run() preamble (oracle/refdriver.py): tests/conftest.py sets False.
    SEED_OCEAN_ONLY_DEFAULT = True

    def __init__(self, seed=0, loglevel=None, device=0, rng='device', iomodule=None, logfile=None, stage_math=None):
        super().__init__()
        if loglevel is not None:
            logger.setLevel(loglevel)
        self.mode = 'Config'
        # one process per GPU (torchrun: RANK / LOCAL_RANK / WORLD_SIZE): every rank runs the same script, seeds the same
        # schedule (same np.random seed) and owns a contiguous range of element IDs; rank 0's readers are the ones that
        # read, their blocks are broadcast; the global reductions of the movers are combined over the ranks
        from . import distributed as D
        self._rank, local_rank, self._world = D.env_world()
        if self._world > 1:
            D.init()
            if device == 0:
                if D.backend() == 'rccl':
                    device = local_rank % max(1, D.device_count())
                else:
                    import torch
                    device = local_rank % max(1, torch.cuda.device_count())
            if rng != 'device':
                raise ValueError("a sharded run (WORLD_SIZE > 1) needs rng='device': np.random draws are sized by the "
                                 "elements of one process")
        self._ctx, self._device, self._seed = None, device, seed or 0   # the device context is created on first use
        self.rng = rng                       # 'device' (Philox by ID) | 'numpy' (np.random in reference call order)
        # arithmetic of the Runge-Kutta stage evaluations (odr_ctx_set_stage_math): 'exact' reproduces every float32 rounding
        # point of the reference inside a stage (1e-10 deg per step vs the oracle), 'fast' takes the stage step and the stage
        # sample without them (<= 2e-9 deg per step).  Runs that replay the reference's np.random stream are parity runs: exact.
        self.stage_math = stage_math or ('exact' if rng == 'numpy' else 'fast')
        if self.stage_math not in ('exact', 'fast'):
            raise ValueError("stage_math must be 'exact' or 'fast'")
        if seed is not None:
            np.random.seed(seed)             # basemodel/__init__.py:326
        self.status_categories = ['active']
        self._pending_status = []            # reasons handed to the device with a provisional status number
        self.readers = {}                    # name -> DeviceReaderBinding (created by _finalize_environment)
        self._advected = False
        self._readers_host = {}              # name -> (reader, variables) as given to add_reader
        self.priority_list = {}              # variable -> [reader names]
        self.discarded_readers = {}          # name -> reason (Environment.discarded_readers, environment.py:376-389)
        self.required_variables = {k: dict(v) for k, v in type(self).required_variables.items()}
        self._sched = None                   # scheduled elements (host arrays, seeding order = ID)
        self.P = None
        self.steps_calculation = 0
        self.time = self.start_time = self.time_step = None
        self.newly_seeded = 0
        self._newly_any, self._g_active = False, 0
        self.sort_every = None               # re-sort interval of the device layout in steps (None: chosen in run(); 0: never)
        self._add_config({
            'general:use_auto_landmask': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_ADVANCED,
                                          'description': 'GSHHG landmask (not available on the device path)'},
            'general:coastline_action': {'type': 'enum', 'enum': ['none', 'stranding', 'previous'],
                                         'default': 'stranding', 'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'general:coastline_approximation_precision': {'type': 'float', 'default': None, 'min': 0.0001, 'max': 0.005,
                                                          'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'general:time_step_minutes': {'type': 'float', 'min': .01, 'max': 1440, 'default': 60,
                                          'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'general:time_step_output_minutes': {'type': 'float', 'min': 1, 'max': 1440, 'default': None,
                                                 'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'seed:number': {'type': 'int', 'default': 1, 'min': 1, 'max': 100000000000,
                            'level': CONFIG_LEVEL_ESSENTIAL, 'description': ''},
            'seed:ocean_only': {'type': 'bool', 'default': self.SEED_OCEAN_ONLY_DEFAULT, 'level': CONFIG_LEVEL_ESSENTIAL,
                                'description': 'If True, elements seeded on land will be moved to the closest position in ocean'},
            'drift:profiles_depth': {'type': 'float', 'default': 50, 'min': 0, 'max': None, 'level': CONFIG_LEVEL_ADVANCED,
                                     'description': 'Environment profiles are retrieved from surface and down to this depth'},   # :457
            'drift:max_age_seconds': {'type': 'float', 'default': None, 'min': 0, 'max': 1e12,
                                      'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'drift:deactivate_north_of': {'type': 'float', 'default': None, 'min': -90, 'max': 90,
                                          'level': CONFIG_LEVEL_ADVANCED, 'description': ''},   # :477-516
            'drift:deactivate_south_of': {'type': 'float', 'default': None, 'min': -90, 'max': 90,
                                          'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'drift:deactivate_east_of': {'type': 'float', 'default': None, 'min': -360, 'max': 360,
                                         'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'drift:deactivate_west_of': {'type': 'float', 'default': None, 'min': -360, 'max': 360,
                                         'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'drift:advection_scheme': {'type': 'enum', 'enum': ['euler', 'runge-kutta', 'runge-kutta4'],
                                       'default': 'euler', 'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'drift:current_uncertainty': {'type': 'float', 'default': 0, 'min': 0, 'max': 5,
                                          'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'drift:current_uncertainty_uniform': {'type': 'float', 'default': 0, 'min': 0, 'max': 5,
                                                  'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'drift:wind_uncertainty': {'type': 'float', 'default': 0, 'min': 0, 'max': 5,
                                       'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'drift:relative_wind': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'drift:max_speed': {'type': 'float', 'default': 1, 'min': 0, 'max': 100,
                                'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'readers:max_number_of_fails': {'type': 'int', 'default': 1, 'min': 0, 'max': 1e6,
                                            'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
        })
        for v, spec in self.required_variables.items():   # environment.py:41-76
            self._add_config({
                'environment:constant:%s' % v: {'type': 'float', 'default': None, 'min': -1e12, 'max': 1e12,
                                                'level': CONFIG_LEVEL_BASIC, 'description': ''},
                'environment:fallback:%s' % v: {'type': 'float', 'default': spec.get('fallback'), 'min': -1e12,
                                                'max': 1e12, 'level': CONFIG_LEVEL_BASIC, 'description': ''}})
        for prop, default in self.element_properties.items():
            self._add_config({'seed:%s' % prop: {'type': 'float', 'default': default, 'min': -1e12, 'max': 1e12,
                                                 'level': CONFIG_LEVEL_ESSENTIAL, 'description': ''}})

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = Context(device=self._device, seed=self._seed)
            if not os.environ.get('ODR_STAGE_MATH'):
                self._ctx.set_stage_math(self.stage_math)
        return self._ctx

    # ------------------------------------------------------------------ mode machine (:136-190)
    def _require(self, *modes):
        if self.mode not in modes:
            raise WrongMode('Cannot call this method in mode %s (allowed: %s)' % (self.mode, modes))

    def set_config(self, key, value):
        self._require('Config')
        if key == 'vertical_mixing:TSprofiles' and value and 'sea_water_salinity' in self.required_variables:
            # (oceandrift.py:146,461-477) -- only a model that requires salinity gets T / S profiles handed to its
            # update_terminal_velocity; for OceanDrift itself the option changes nothing (its hook is empty, :285-297) and is
            # accepted.  OpenOil's use of the profiles is not built: in the reference the temperature profiles reach the hook
            # in Celsius and have 273.15 subtracted again.
            raise NotImplementedError('vertical_mixing:TSprofiles is not implemented for %s (DESIGN.md section 7)' % type(self).__name__)
        super().set_config(key, value)

    # ------------------------------------------------------------------ readers (:613-632)
    def add_reader(self, readers, variables=None, first=False):
        self._require('Config')
        if not isinstance(readers, (list, tuple)):
            readers = [readers]
        for r in readers:
            if not (hasattr(r, 'get_variables') and hasattr(r, 'variables')):
                raise TypeError('Please provide Reader object')
            name = r.name
            i = 1
            while name in self._readers_host:
                i += 1
                name = '%s_%d' % (r.name, i)
            vs = [v for v in (variables or r.variables)]
            if getattr(r, 'device_kind', None) == 'double_gyre':
                vs = ['x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask']
            self._readers_host[name] = (r, vs)
            for v in vs:
                lst = self.priority_list.setdefault(v, [])
                if first:
                    lst.insert(0, name)
                else:
                    lst.append(name)

    def _finalize_environment(self, t0, t1):
        """Environment.finalize (environment.py:139-211): constant readers from config, priority lists,
        fallbacks; the first blocks of the gridded readers are uploaded here."""
        consts = {v: self.get_config('environment:constant:%s' % v) for v in self.required_variables}
        consts = {v: c for v, c in consts.items() if c is not None}
        if consts:   # environment.py:172-182: constant reader with highest priority
            self._readers_host['constant_reader_config'] = (ConstantReader(consts), list(consts))
            for v in consts:
                self.priority_list.setdefault(v, []).insert(0, 'constant_reader_config')
        for name, (r, vs) in self._readers_host.items():
            self.readers[name] = DeviceReaderBinding(self.ctx, r, variables=vs)
            self.readers[name].set_extent(getattr(self, '_block_extent', getattr(self, 'simulation_extent', None)))
        landmasks = [n for n, (r, _) in self._readers_host.items() if getattr(r, 'device_kind', None) == 'landmask']
        self._landmask_sid = self.readers[landmasks[0]].sid if landmasks else None
        if self.get_config('general:use_auto_landmask'):
            # environment.py:190-211: with the auto landmask, land_binary_mask comes from the global landmask reader
            # only.  The GSHHG dataset is not part of this repository: the raster is whatever
            # readers.LandmaskRasterReader was added; without one there is nothing to look up.
            if not landmasks and self.get_config('environment:constant:land_binary_mask') is None:
                raise NotImplementedError('general:use_auto_landmask needs a landmask raster (the GSHHG dataset is not '
                                          'shipped): add readers.LandmaskRasterReader, another reader or a constant / '
                                          'fallback for land_binary_mask (DESIGN.md 8e)')
            if landmasks:
                self.priority_list['land_binary_mask'] = [landmasks[0]]
        self._ensure_reader_levels(t0, t1)
        self._bind_variables()

    def _bind_variables(self):
        for v in self.required_variables:
            ids = [self.readers[n].sid for n in self.priority_list.get(v, [])
                   if n in self.readers and self.readers[n].sid is not None and not getattr(self.readers[n], 'host_eval', False)]
            self.ctx.bind(v, ids[:4], self.get_config('environment:fallback:%s' % v))

    def _ensure_reader_levels(self, t0, t1):
        """Make the time levels of every reader resident.  A reader whose get_variables raises is counted and, after
        more than readers:max_number_of_fails failures, discarded (Environment.get_environment, environment.py:640-668,
        discard_reader :376-389; tests/readers/test_readers.py:15-26): its variables fall to the next reader of the
        priority list or to the fallback value."""
        rebind = self._check_reader_windows()
        # the instants a step samples a reader at: every reader at t0 (the main-loop get_environment); the readers that
        # hold the current also at the Runge-Kutta stage times (physics_methods.py:638-670)
        scheme = self.get_config('drift:advection_scheme') if 'drift:advection_scheme' in self._config else 'euler'
        stage_times = {'runge-kutta': [t0, t0 + (t1 - t0) / 2], 'runge-kutta4': [t0, t0 + (t1 - t0) / 2, t1]}.get(scheme, [t0])
        for name, b in list(self.readers.items()):
            try:
                sid_before = b.sid
                holds_current = 'x_sea_water_velocity' in b.variables or 'y_sea_water_velocity' in b.variables
                b.ensure_levels(t0, t1, times=stage_times if holds_current else [t0])
                # a gridded reader gets its device source with its first block: a reader whose time coverage starts
                # inside the run enters the priority lists when the run reaches it (the reference uses any reader
                # that covers the current time, environment.py:597-668)
                rebind |= sid_before is None and b.sid is not None
            except ReaderLevelsError:
                raise
            except Exception as e:   # the reference catches every exception of a reader call
                if getattr(e, 'code', 0) == -3:     # ODR_ERR_CAPACITY: the device library is out of sources / slots -- not a reader failure
                    raise
                rebind |= self._reader_failed(name, b, e)
        if rebind:
            self._bind_variables()

    WINDOW_CHECK_EVERY = 16     # steps between two looks at the elements' lon / lat box (one reduction + host read each)

    def _check_reader_windows(self):
        """Blocks are cut to the box the elements can reach at drift:max_speed (set_extent) -- plus what they could cover
        until the next check.  Elements that are faster than that (OpenOil with strong wind, Stokes drift and current) would
        leave the window and silently take fallback values; the reference's blocks follow the elements instead.  Every
        WINDOW_CHECK_EVERY steps the elements' box is compared with the windows; a reader whose window they approach gets a
        new one around where they are now (its device source is rebuilt: returns True when variables must be rebound)."""
        cut = [b for b in self.readers.values() if getattr(b, 'extent', None) is not None]
        if not cut or self.P is None or self.time_step is None or self.steps_calculation % self.WINDOW_CHECK_EVERY != 0 or \
                self.steps_calculation == 0 or (self._g_active if self._world > 1 else self.num_elements_active()) == 0:
            return False
        r = self._reduce_scalars() if self._world == 1 else self.P.reduction_dict(
            __import__('opendrift_amd.distributed', fromlist=['x']).combine_reductions(self.P.reduce_local()))
        if self._world > 1:
            self._timing_collectives += 2
        reach = self.get_config('drift:max_speed') * self.WINDOW_CHECK_EVERY * abs(self.time_step.total_seconds())
        glat = reach / 111000.
        glon = glat / max(0.05, np.cos(np.radians(0.5 * (r['lat_min'] + r['lat_max']))))
        rebind = False
        for b in cut:
            if b.outside_window(r['lon_min'], r['lon_max'], r['lat_min'], r['lat_max'], glon, glat):
                left = max(0, self.expected_steps_calculation - self.steps_calculation) + self.WINDOW_CHECK_EVERY
                d = self.get_config('drift:max_speed') * left * abs(self.time_step.total_seconds()) / 111000.
                dl = d / max(0.05, np.cos(np.radians(0.5 * (r['lat_min'] + r['lat_max']))))
                box = np.array([max(-360, r['lon_min'] - dl), max(-89, r['lat_min'] - d), min(360, r['lon_max'] + dl),
                                min(89, r['lat_max'] + d)])
                logger.warning('elements approach the edge of the block window of reader %s (faster than drift:max_speed = %s '
                               'm/s?): new window %s', b.reader.name, self.get_config('drift:max_speed'), box)
                b.recut(box)
                rebind = True
        return rebind

    def _reader_failed(self, name, b, e):
        """Count the failure; after more than readers:max_number_of_fails the reader is discarded (discard_reader,
        environment.py:376-389).  Returns True when device priority lists must be rebuilt."""
        r = b.reader
        r.number_of_fails = getattr(r, 'number_of_fails', 0) + 1
        max_fails = self.get_config('readers:max_number_of_fails')
        logger.warning('Reader %s failed (%s), number of fails: %d', name, e, r.number_of_fails)
        if r.number_of_fails > max_fails:
            self.discarded_readers[name] = 'failed more than allowed number of times (%d)' % max_fails
            if hasattr(b, 'close_read_ahead'):
                b.close_read_ahead()
            del self.readers[name]
            for v, lst in self.priority_list.items():
                if name in lst:
                    lst.remove(name)
            return b.sid is not None      # it had delivered blocks before: take it out of the device lists
        return False

    # ------------------------------------------------------------------ seeding (:1033-1330)
    def seed_elements(self, lon, lat, time, radius=0, number=None, number_per_point=None,
                      radius_type='gaussian', **kwargs):
        self._require('Config', 'Ready')
        lon, lat = np.atleast_1d(lon).ravel().astype(float), np.atleast_1d(lat).ravel().astype(float)
        radius = np.atleast_1d(radius).ravel()
        time = list(np.atleast_1d(time))
        if lat.max() > 90 or lat.min() < -90:
            raise ValueError('Latitude must be between -90 and 90 degrees')
        if len(lon) != len(lat):
            raise ValueError('Lon and lat must have same lengths')
        if len(lon) > 1:
            if number_per_point is not None:
                if number is not None:
                    raise ValueError('Both number and number_per_point is provided')
                number = number_per_point * len(lon)
            if number is not None:
                if number % len(lon) != 0:
                    raise ValueError('Lon and lat have length %s, but number is %s, which is not a multiple'
                                     % (len(lon), number))
                npp = int(number / len(lon))
                if npp > 1:
                    lon, lat = np.repeat(lon, npp), np.repeat(lat, npp)
            number = len(lon)
        else:
            if number is None:
                number = len(time) if len(time) > 2 else self.get_config('seed:number')
            lon, lat = lon * np.ones(number), lat * np.ones(number)
        if len(time) != number and len(time) > 1:
            if len(time) == 2:
                td = (time[1] - time[0]) / (number - 1)
                time = [time[0] + i * td for i in range(number)]
            else:
                raise ValueError('Time array has length %s, must be 1, 2 or %s' % (len(time), number))
        single_time = len(time) == 1
        if radius.max() > 0:   # :1151-1170
            if radius_type == 'gaussian':
                x = np.random.randn(number) * radius
                y = np.random.randn(number) * radius
                az, dist = np.degrees(np.arctan2(x, y)), np.sqrt(x * x + y * y)
            elif radius_type == 'uniform':
                az = np.random.rand(number) * 360
                dist = np.sqrt(np.random.uniform(0, 1, number)) * radius
            else:
                raise ValueError('radius_type must be gaussian or uniform')
            lon, lat = self._geod_fwd(lon, lat, az, dist)
        z = kwargs.pop('z', None)
        if z is None and 'seed:seafloor' in self._config and self.get_config('seed:seafloor') is True:
            z = 'seafloor'                     # "Seafloor is selected, neglecting z" (:1169-1173)
        if z is None:
            z = self.get_config('seed:z') if 'seed:z' in self._config else 0.0
        zoff = np.nan                          # finite: metres above the sea floor, depth looked up when the run starts
        if isinstance(z, str):                 # 'seafloor' / 'seafloor+M' (:1174-1210)
            if z[0:8] != 'seafloor':
                raise ValueError('z must be a number, "seafloor" or "seafloor+<metres>": ' + z)
            above = float(z[9:]) if len(z) > 8 and z[8] == '+' else 0.0
            cfg = lambda k: self.get_config(k) if k in self._config else None
            depth_c = cfg('environment:constant:sea_floor_depth_below_sea_level')
            depth_f = cfg('environment:fallback:sea_floor_depth_below_sea_level')
            if depth_c is not None:
                z = -np.float32(depth_c) + above
            elif 'sea_floor_depth_below_sea_level' in self.priority_list:
                # the reference samples its readers here; the readers of this model live on the device once the run is
                # set up, so the lookup (same sampling kernels, at the seeded positions) happens at the start of run()
                z, zoff = np.nan, above
            elif depth_f is not None:
                z = -np.float32(depth_f) + above
            else:
                raise ValueError('A reader providing the variable sea_floor_depth_below_sea_level must be '
                                 'added before seeding elements at seafloor.')
        props = {}
        for prop in self.element_properties:
            val = kwargs.pop(prop, None)
            props[prop] = np.float32(self.get_config('seed:%s' % prop) if val is None else val) * np.ones(number, np.float32)
        if kwargs:
            raise TypeError('Redundant arguments: %s' % list(kwargs))
        # LagrangianArray stores lon/lat/z as float32 at seeding (elements.py:71-88,156-158)
        new = dict(lon=np.float32(lon).astype(np.float64), lat=np.float32(lat).astype(np.float64),
                   z=(np.float32(z) * np.ones(number, np.float32)).astype(np.float64),
                   z_above_seafloor=np.full(number, zoff),
                   time=np.full(number, time[0], dtype=object) if single_time else np.array(time, dtype=object),
                   t_epoch=(np.full(number, _epoch(time[0])) if single_time else
                            np.array(time, dtype='datetime64[us]').astype(np.int64) / 1e6), **props)   # vectorised schedule
        if self._sched is None:
            self._sched = new
        else:
            for k in new:
                self._sched[k] = np.concatenate([self._sched[k], new[k]])
        self._sched['ID'] = np.arange(len(self._sched['lon']), dtype=np.int32)
        st = self._sched['time'][int(np.argmin(self._sched['t_epoch']))]
        self.start_time = st if self.start_time is None else min(self.start_time, st)
        if self.mode == 'Config':
            self.mode = 'Ready'

    def _geod_fwd(self, lon, lat, az, dist):
        """pyproj.Geod.fwd on the device (the only geodesic in the product is the HIP one)."""
        n = len(lon)
        T = self.ctx.particles(n)
        T.append(lon, lat)
        a = np.radians(az)
        T.update_positions(dist * np.sin(a), dist * np.cos(a), 1.0)
        d = T.download()
        T.close()
        return d['lon'], d['lat']

    # ------------------------------------------------------------------ counts (:841-867)
    def num_elements_active(self):
        return len(self.P) if self.P is not None else 0

    def num_elements_deactivated(self):
        return self.P.count()[1] if self.P is not None else 0

    def num_elements_scheduled(self):
        if self._sched is None:
            return 0
        if getattr(self, '_n_unreleased', None) is not None:
            return self._n_unreleased
        return int(self._released_mask().size - self._released_mask().sum())

    def num_elements_total(self):
        if getattr(self, '_n_global', None) is not None:
            return self._n_global       # a sharded run keeps its own ID range of the schedule only
        return 0 if self._sched is None else len(self._sched['lon'])

    def _released_mask(self):
        if not hasattr(self, '_released'):
            self._released = np.zeros(len(self._sched['lon']), bool)
        return self._released

    @property
    def elements(self):
        """The active elements as the reference's `o.elements` presents them: NumPy arrays in ascending-ID order (the
        reference's release order), WRITABLE -- `self.elements.z = ...`, `self.elements.lon[mask] += ...` inside a model's
        update() or any other hook reach the device when the hook returns (run() calls _flush_elements()), or at once
        with o.elements.flush().  The arrays are a snapshot of the device state at the time of access."""
        v = getattr(self, '_elements_view', None)
        if v is None or v.stale():
            v = self._elements_view = ElementsView(self)
        return v

    def _flush_elements(self):
        v = getattr(self, '_elements_view', None)
        if v is not None:
            v.flush()
        self._elements_view = None

    @property
    def elements_deactivated(self):
        return SimpleNamespace(**self.P.download_deactivated())

    @property
    def environment(self):
        return SimpleNamespace(**{v: self.P.env_download(v) for v in self._sampled})

    def _resolve_seafloor_seeds(self, lo_id, hi_id):
        """seed_elements(z='seafloor[+M]') with the depth from a reader (:1185-1210): sea_floor_depth_below_sea_level sampled
        at the seeded positions by the device's own sampling path, z = -float32(depth) + M, stored as float32."""
        s = self._sched
        off = s.get('z_above_seafloor')
        if off is None:
            return
        idx = np.nonzero(np.isfinite(off[lo_id:hi_id]) & np.isnan(s['z'][lo_id:hi_id]))[0] + lo_id
        if idx.size == 0:
            return
        Q = self.ctx.particles(idx.size)
        try:
            Q.append(s['lon'][idx], s['lat'][idx], z=np.zeros(idx.size))
            depth = Q.env_sample(['sea_floor_depth_below_sea_level'], _epoch(self.start_time), download=True)[
                'sea_floor_depth_below_sea_level']
        finally:
            Q.close()
        s['z'][idx] = np.float32(-depth.astype(np.float32) + off[idx]).astype(np.float64)

    def _sample_land(self, lon, lat):
        """land_binary_mask at (lon, lat, z = 0) at the start of the run, by the device's own sampling path"""
        out = np.empty(len(lon), np.float32)
        chunk = 4_000_000
        for a in range(0, len(lon), chunk):
            b = min(len(lon), a + chunk)
            Q = self.ctx.particles(b - a)
            try:
                Q.append(lon[a:b], lat[a:b], z=np.zeros(b - a))
                out[a:b] = Q.env_sample(['land_binary_mask'], _epoch(self.start_time), download=True)['land_binary_mask']
            finally:
                Q.close()
        return out

    def closest_ocean_points(self, lon, lat):
        """seed:ocean_only (basemodel/__init__.py:936-1031): elements seeded on land go to the nearest ocean point of a
        0.01 deg raster around the seeds (at most 1000 x 1000 points), looked up with the reader that provides
        land_binary_mask -- here sampled through the device (the same nearest-node lookup the run uses) at the run's start
        time (the reference asks at the land reader's own start time, :983,:1010: the same mask unless the land mask changes
        in time); nearest neighbour by scipy's cKDTree in (lon, lat) as in the reference; a NaN of the mask counts as land, as
        there (`land != 0`, :990).  Returns (lon, lat, land_indices)."""
        try:
            import scipy.spatial
        except ImportError as e:      # (SciPy is a dependency of the reference as well, pyproject.toml)
            raise ImportError("seed:ocean_only needs scipy.spatial.cKDTree to move elements seeded on land to the nearest ocean "
                              "point (basemodel/__init__.py:1020-1026); install SciPy or set_config('seed:ocean_only', False)") from e
        lon, lat = np.array(lon, dtype=np.float64), np.array(lat, dtype=np.float64)
        live = [n for n in self.priority_list.get('land_binary_mask', []) if n in self.readers and self.readers[n].sid is not None]
        if not live or len(lon) == 0:
            return lon, lat, None      # no land reader on the device (the reference would fetch the GSHHG landmask: not shipped)
        deltalon = deltalat = 0.01
        numbuffer = 10
        # (the reference's scheduled lon / lat are float32 arrays: the raster's corners are float32 sums, NEP 50)
        lonmin, lonmax = np.float32(lon.min()) - deltalon * numbuffer, np.float32(lon.max()) + deltalon * numbuffer
        latmin, latmax = np.float32(lat.min()) - deltalat * numbuffer, np.float32(lat.max()) + deltalat * numbuffer
        land = self._sample_land(lon, lat)
        if not (np.nanmax(land) > 0):
            return lon, lat, None      # 'All points are in ocean'
        land_indices = np.where(land != 0)[0]
        longrid, latgrid = np.arange(lonmin, lonmax, deltalon), np.arange(latmin, latmax, deltalat)
        if len(longrid) > 1000 or len(latgrid) > 1000:
            logger.warning('Particles cover large area - using coarser resolution for closest ocean point')
            longrid, latgrid = np.linspace(lonmin, lonmax, 1000), np.linspace(latmin, latmax, 1000)
        longrid, latgrid = np.meshgrid(longrid, latgrid)
        longrid, latgrid = longrid.ravel(), latgrid.ravel()
        landgrid = self._sample_land(longrid, latgrid)
        covered = np.isfinite(landgrid)            # "Remove grid-points not covered by this reader"
        longrid, latgrid, landgrid = longrid[covered], latgrid[covered], landgrid[covered]
        if landgrid.size == 0:
            logger.warning('Land grid has zero size, cannot move elements.')
            return lon, lat, land_indices
        if landgrid.min() == 1:
            logger.warning('No ocean pixels nearby, cannot move elements.')
            return lon, lat, land_indices
        oceangridlons, oceangridlats = longrid[landgrid == 0], latgrid[landgrid == 0]
        tree = scipy.spatial.cKDTree(np.dstack([oceangridlons, oceangridlats])[0])
        _dist, indices = tree.query(np.dstack([lon[land_indices], lat[land_indices]]))
        indices = indices.ravel()
        lon[land_indices] = np.float32(oceangridlons[indices])      # (stored into the float32 schedule)
        lat[land_indices] = np.float32(oceangridlats[indices])
        logger.info('Moved %i out of %i points from land to water' % (len(land_indices), len(lon)))
        return lon, lat, land_indices

    # ------------------------------------------------------------------ loop pieces
    def release_elements(self):   # :909-934
        s, rel = self._sched, self._released_mask()
        if getattr(self, '_n_unreleased', None) is None:
            self._n_unreleased = int(rel.size - rel.sum())
            self._t_sched = s['t_epoch']
        if self._n_unreleased == 0:      # nothing left to seed: no O(N) work per step
            self.newly_seeded = 0
            return
        t0, t1 = _epoch(self.time), _epoch(self.time + self.time_step)
        if self.time_step.total_seconds() >= 0:
            idx = ~rel if self._all_at_start else (~rel & (self._t_sched >= t0) & (self._t_sched < t1))
        else:
            idx = ~rel & (self._t_sched <= t0) & (self._t_sched > t1)
        n = int(idx.sum())
        self.newly_seeded = n
        if n == 0:
            return
        kw = {p: s[p][idx] for p in self.element_properties if p in ('wind_drift_factor', 'current_drift_factor',
                                                                     'terminal_velocity')}
        n_before = self.num_elements_active()
        self.P.append(s['lon'][idx], s['lat'][idx], z=s['z'][idx], id=s['ID'][idx], **kw)
        for slot, name in enumerate(getattr(self, 'aux_properties', [])):   # model-specific float32 properties
            self.P.set_property(slot, s[name][idx], offset=n_before)
        rel[idx] = True
        self._n_unreleased -= n

    def _can_be_missing(self, names):
        """Variables that can still be NaN after get_environment: those without a fallback (environment.py:781-790)."""
        return [v for v in names if self.get_config('environment:fallback:%s' % v) is None]

    def report_missing_variables(self):   # :2501-2515 on Environment.get_environment's `missing` (environment.py:903-908)
        names = self._can_be_missing(self._sampled)
        if names and (self._world > 1 or self.num_elements_active() > 0):   # sharded: same calls on every rank
            self.P.deactivate_missing(names, self._status_code('missing_data'))

    def deactivate_outside(self):   # :2354-2382, validity domain of :2169-2179
        dom = [self.get_config('drift:deactivate_%s_of' % k) for k in ('west', 'east', 'south', 'north')]
        if dom != [None, None, None, None]:
            self.P.deactivate_outside(*dom, status_code=self._status_code('outside'))

    def deactivate_elements(self, mask, reason='deactivated'):   # :1774-1795
        if not np.any(mask):     # "if sum(indices) == 0: return" -- no status category for an empty mask
            return
        if reason not in self.status_categories:
            self.status_categories.append(reason)
        self.P.deactivate(mask, self.status_categories.index(reason))

    _PROVISIONAL = {'missing_data': 100, 'outside': 101, 'stranded': 102, 'seeded_on_land': 103, 'seafloor': 104, 'retired': 105}

    def _status_code(self, reason):
        """Status number for a reason the device may assign.  The reference appends a category when a reason FIRST
        OCCURS (deactivate_elements, basemodel/__init__.py:1778-1781), so a reason not seen yet gets a provisional
        number; _resolve_status() turns it into the next category index once an element actually carries it."""
        if reason in self.status_categories:
            return self.status_categories.index(reason)
        code = self._PROVISIONAL[reason]
        if reason not in self._pending_status:
            self._pending_status.append(reason)
        return code

    # ---- sharded run: what the control flow and the status categories depend on is combined over the ranks, so that
    # every rank takes the same branches, makes the same collectives and numbers the deactivation reasons alike.
    # ONE collective per step (_step_summary, after the fused launch); the counts at the top of a step need none: every
    # rank knows the release schedule of ALL elements (run() keeps its per-step histogram before it cuts the schedule down
    # to its own ID range), and the number of active elements is what the previous step's collective left plus the releases.
    def _global_counts(self, step=None):
        """(active, scheduled) over all ranks; also whether ANY rank released elements in this step (the reference's
        `newly_seeded_IDs is not None`, which arms the 'seeded_on_land' check)."""
        na, ns = self.num_elements_active(), self.num_elements_scheduled()
        self._newly_any = self.newly_seeded > 0
        self._g_active = na
        if self._world == 1:
            return na, ns
        if step is not None and getattr(self, '_g_release', None) is not None and step < len(self._g_release):
            rel = int(self._g_release[step])
            self._newly_any = rel > 0
            self._g_active = self._g_active_carry + rel
            self._g_active_carry = self._g_active
            # present elements on the lower ranks (all their IDs are smaller): what the last collective left + their releases
            # of this step -- the offset of this rank's elements in the all-rank array whose positions select the ensemble
            # member (readers/interpolation/structured.py:119-135)
            self._below_active += int(self._g_release_below[step])
            self.P.set_rank_offset(self._below_active)
            return self._g_active, int(self._n_global - self._g_release_cum[step + 1])
        from . import distributed as D
        self._timing_collectives += 1
        g = D.allreduce_scalars([na, ns, self.newly_seeded], 'sum')
        self._newly_any = g[2] > 0
        self._g_active = self._g_active_carry = int(g[0])
        return int(g[0]), int(g[1])

    def _step_summary(self, kept, flags, want_reductions, start_only=False, handle=None, from_scan=False):
        """The ONE collective of a sharded step: this rank's kept count, provisional status flags and (when a mover of this
        step needs them) its 16 raw reduction slots, all-gathered; returns the all-rank (kept, flags) and installs the
        all-rank reductions for the movers that follow (released by _step_release at the end of the step).
        start_only: the collective is started and its handle returned; a second call with `handle` finishes it -- between the
        two the loop launches what does not depend on the other ranks (the step's update()), so that the collective's latency
        is spent while the device works (a step whose movers need the all-rank reductions cannot wait that long)."""
        if self._world == 1:
            return kept, flags
        from . import distributed as D
        t0 = time.perf_counter()
        if handle is None:
            wdd, rel = self.get_config('drift:wind_drift_depth', 0.1), bool(self.get_config('drift:relative_wind'))
            raw = self.P.reduce_local(wdd, rel) if want_reductions else np.zeros(16)
            t0 = time.perf_counter()
            if from_scan:
                # (C-ABI collectives) between scan_status_begin and _end: the kept count and the flags of this rank's row are
                # taken on the device from the fold of the status scan -- the collective leaves before the host has read them
                row = np.zeros(25)
                h = D.start_allgather_vector(row, from_scan_ctx=self.ctx)
                self._timing_collective_s += time.perf_counter() - t0
                self._timing_collectives += 1
                return h
            row = np.concatenate([[float(kept)], [float(flags >> k & 1) for k in range(8)], raw])
            if start_only:
                h = D.start_allgather_vector(row)
                self._timing_collective_s += time.perf_counter() - t0
                self._timing_collectives += 1
                return h
            rows = D.allgather_vector(row)
            self._timing_collectives += 1
        else:
            rows = D.finish_allgather_vector(handle)
        self._timing_collective_s += time.perf_counter() - t0
        g_kept = int(round(rows[:, 0].sum()))
        g_flags = sum(1 << k for k in range(8) if rows[:, 1 + k].max() > 0)
        if want_reductions:
            self._step_red = D.combine_rows(rows[:, 9:])
            self.P.reduce_install(self._step_red)
        self._g_active = self._g_active_carry = g_kept
        self._below_active = int(round(rows[:self._rank, 0].sum()))
        return g_kept, g_flags

    def _step_release(self):
        if getattr(self, '_step_red', None) is not None:
            self._step_red = None
            self.P.reduce_unpin()

    def _needs_reductions(self):
        """Does a mover of this step consult global maxima / counts (advect_wind, stokes_drift, horizontal_diffusion, the
        wind-parameterised mixing)?"""
        rv, iz = self.required_variables, self._identically_zero
        sx, sy = 'sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity'
        # a mover whose input no reader delivers and whose fallback is 0 returns early on every rank without looking
        # (advect_wind, _stokes_arguments, horizontal_diffusion); the analytic diffusivity models take the deepest mixed layer
        wind = 'x_wind' in rv and not self._calm_everywhere()
        stokes = sx in rv and self.get_config('drift:stokes_drift') is not False and not (iz(sx) and iz(sy))
        hdiff = 'horizontal_diffusivity' in rv and not iz('horizontal_diffusivity')
        mld = 'ocean_mixed_layer_thickness' in rv and self.get_config('drift:vertical_mixing') is True and \
            self._effective_diffusivity_model() not in ('environment', 'constant')
        return wind or stokes or hdiff or mld

    def _effective_diffusivity_model(self):
        """vertical_mixing:diffusivitymodel as vertical_mixing applies it: 'environment' without a reader that delivers
        ocean_vertical_diffusivity is Large et al. (1994) (oceandrift.py:431-447) -- which consults the deepest mixed layer of
        ALL elements, i.e. the step's collective must carry the reductions."""
        model = self.get_config('vertical_mixing:diffusivitymodel', 'environment') if 'vertical_mixing:diffusivitymodel' in self._config \
            else 'environment'
        if model == 'environment' and not any(self.readers[n].sid is not None
                                              for n in self.priority_list.get('ocean_vertical_diffusivity', []) if n in self.readers):
            model = 'windspeed_Large1994'
        return model

    def _calm_everywhere(self):
        """advect_wind returns at its `wind_speed.max() == 0` test (physics_methods.py:771-780) on every rank whatever the
        elements are: no reader delivers the wind, its fallback is 0 and it is not taken relative to the current."""
        return self._identically_zero('x_wind') and self._identically_zero('y_wind') and \
            not self.get_config('drift:relative_wind')

    def _global_scan(self, kept, flags):
        if self._world == 1:
            return kept, flags
        from . import distributed as D
        bits = [float(flags >> k & 1) for k in range(8)]
        self._timing_collectives = getattr(self, '_timing_collectives', 0) + 1
        g = D.allreduce_scalars(bits, 'max')
        return kept, sum(1 << k for k in range(8) if g[k] > 0)

    def _combine(self):
        """combine(raw16) for Particles.reduce_global: identity in a one-process run"""
        from . import distributed as D
        return D.combine_reductions

    def _resolve_status(self, flags=None, pending=None):
        """Register the pending reasons that occurred (in the order of the calls that could assign them) and renumber
        their elements.  `flags`: the provisional status numbers present (Particles.scan_status); without it one scan is
        made -- one host read however many reasons are pending, none when nothing is pending."""
        if pending is None:      # (`pending`: the list taken when a collective that is finished behind update() was started)
            pending, self._pending_status = self._pending_status, []
        pending = [r for r in pending if r not in self.status_categories]
        if not pending:      # (the pending list is the same on every rank: it follows from the configuration alone)
            return
        if flags is None:
            flags = self._global_scan(0, self.P.scan_status()[1] if len(self.P) else 0)[1]
        for reason in pending:
            code = self._PROVISIONAL[reason]
            if flags >> (code - 100) & 1:
                self.status_categories.append(reason)
                self.P.remap_status(code, self.status_categories.index(reason))

    def interact_with_coastline(self, final=False):   # :670-746
        action = self.get_config('general:coastline_action')
        if action == 'none' or 'land_binary_mask' not in self._sampled or \
                (self._world == 1 and self.num_elements_active() == 0):
            return
        if final:
            self.P.env_sample(['land_binary_mask'], _epoch(self.time))
        precision = self.get_config('general:coastline_approximation_precision')
        if precision:   # :726-746: coastline_crossing between the previous and the current position
            if getattr(self, '_landmask_sid', None) is None:
                raise NotImplementedError('coastline_approximation_precision searches the global landmask: add '
                                          'readers.LandmaskRasterReader (the GSHHG dataset is not shipped, DESIGN.md 8e)')
            self.P.coastline_crossing(action, precision, self._landmask_sid,
                                      stranded_code=self._status_code('stranded') if action == 'stranding' else 1,
                                      seeded_on_land_code=self._status_code('seeded_on_land')
                                      if action == 'previous' and self._newly_any else 0)
            return
        if action == 'stranding':
            self.P.coastline('stranding', stranded_code=self._status_code('stranded'))
        else:
            self.P.coastline('previous', seeded_on_land_code=self._status_code('seeded_on_land')
                             if self._newly_any else 0)

    def interact_with_seafloor(self):   # :748-783, 'lift_to_seafloor'
        if 'sea_floor_depth_below_sea_level' not in self.priority_list or \
                (self._world == 1 and self.num_elements_active() == 0):
            return
        action = self.get_config('general:seafloor_action', 'lift_to_seafloor')
        if action == 'lift_to_seafloor' or action == 'previous':
            self.P.seafloor(action)
        elif action == 'deactivate':
            self.P.seafloor(action, self._status_code('seafloor'))

    def _with_seafloor_action(self, call):
        """The reference calls interact_with_seafloor() again inside update() (vertical_buoyancy oceandrift.py:362-368,
        every sub-step of vertical_mixing :555-559): the device call gets the configured action; the 'seafloor' status
        category appears with the first element it deactivates (deactivate_elements, basemodel/__init__.py:1778)."""
        action = self.get_config('general:seafloor_action', 'lift_to_seafloor')
        if 'sea_floor_depth_below_sea_level' not in self.priority_list:
            action = 'none'     # interact_with_seafloor returns before doing anything (:753-754)
        self.ctx.set_seafloor_action(action, self._status_code('seafloor') if action == 'deactivate' else 0)
        call()
        self._resolve_status()

    def update_positions(self, x_vel, y_vel):   # :4631-4669
        self._require('Run')
        self.P.update_positions(x_vel, y_vel, self.time_step.total_seconds())

    def horizontal_diffusion(self):   # :1746-1772
        if 'horizontal_diffusivity' not in self.required_variables or \
                (self._g_active if self._world > 1 else self.num_elements_active()) == 0:
            return
        if self._world > 1 and self._identically_zero('horizontal_diffusivity'):
            return      # (`horizontal_diffusivity.max() == 0` on every rank, basemodel/__init__.py:1754)
        dt = self.time_step.total_seconds()
        if self.rng == 'numpy':
            if self.P.reduce_scalars()['D_max'] == 0:
                return
            n = self.num_elements_active()
            self.P.hdiffusion(dt, normals=(np.random.normal(scale=1, size=n), np.random.normal(scale=1, size=n)))
        else:
            self._with_global_reduction(lambda: self.P.hdiffusion(dt, step=self.steps_calculation))

    def get_environment(self):
        """Environment.get_environment for all required variables (:2238-2246) + uncertainty (:869-891)."""
        t = _epoch(self.time)
        names = list(self.required_variables)
        self._profile_levels = self._profile_level_cut()
        with self._readers_see_truncated_z():
            self.P.env_sample(names, t)
            self._sampled = names
            self._sample_host_readers(names)
        self._add_uncertainty(names, current=True)

    def _profile_level_cut(self):
        """drift:truncate_ocean_model_below_m with diffusivity profiles from a reader (environment.py:554-566): the reference asks
        the reader for the depth range [0, max(deepest truncated element, min(profiles_depth, truncate))], and a reader that hands
        out the levels asked for -- the reference's file readers: reader_netCDF_CF_generic.py:414-423, reader_ROMS_native.py:
        551-560 -- returns a block that ENDS one level + `verticalbuffer` below that depth; elements further down mix on K and dK/dz
        of its last level.  The device block holds every level: the number of levels of the reference's block, for the step's
        mixing launch (odr_vmix_set_profile_levels); 0 = whole columns (no truncation, or a reader that says
        `always_delivers_all_levels`).  Evaluated per step from the elements' depths (the reference keeps a cached block while it
        covers the request: the same whenever some element is below the truncation depth, the case the option is for)."""
        T = self._config.get('drift:truncate_ocean_model_below_m', {}).get('value')
        if T is None or self.P is None or len(self.P) == 0:
            return 0
        for n in self.priority_list.get('ocean_vertical_diffusivity', []):
            b = self.readers.get(n)
            if b is None or b.sid is None:
                continue
            r = b.reader
            if getattr(r, 'always_delivers_all_levels', False) or getattr(r, 'z', None) is None or np.size(r.z) < 3:
                return 0
            zlev = np.asarray(r.z, dtype=np.float64)
            if not zlev[0] > zlev[-1]:
                raise NotImplementedError('drift:truncate_ocean_model_below_m with diffusivity profiles from reader "%s": ascending '
                                          'z levels' % n)
            deepest = float(self.P.reduce_local()[5])            # max(-z) over the active elements of this rank
            if self._world > 1:
                from . import distributed as D
                self._timing_collectives += 1
                deepest = float(D.allreduce_scalars([deepest], 'max')[0])
            depth = max(min(deepest, float(T)), min(float(self.get_config('drift:profiles_depth')), float(T)))
            cut = int(min(len(zlev), np.searchsorted(-zlev, depth) + 1 + int(getattr(r, 'verticalbuffer', 1))))
            return cut if cut < len(zlev) else 0
        return 0

    def _readers_see_truncated_z(self):
        """drift:truncate_ocean_model_below_m (environment.py:554-566): inside the block the sampling calls see
        max(z, -depth), the elements keep their own z."""
        from contextlib import contextmanager, nullcontext
        d = self._config.get('drift:truncate_ocean_model_below_m', {}).get('value')
        if d is None or len(self.P) == 0:
            return nullcontext()

        @contextmanager
        def clipped():
            self.P.truncate_z(d)
            try:
                yield
            finally:
                self.P.restore_z()
        return clipped()

    def _host_bindings(self):
        return [(n, b) for n, b in self.readers.items() if getattr(b, 'host_eval', False)]

    def _ensemble_current(self):
        """The current comes (also) from a reader that hands it out as a list of ensemble members."""
        return any(
            b.sid is not None and self.ctx._grids.get(b.sid, {}).get('members') and
            ('x_sea_water_velocity' in b.variables or 'y_sea_water_velocity' in b.variables) for b in self.readers.values())

    def _sample_host_readers(self, names, P=None, time=None):
        """Readers evaluated on the host (user-defined ContinuousReaders): where such a reader comes BEFORE the device
        sources of a variable its finite values replace what the device sampled (the priority-list walk of
        environment.py:597-762 with the host reader in first place); elsewhere in the list it only fills what is
        still missing.  P / time: the particle set and time of the sample (default: the elements, now)."""
        hb = self._host_bindings()
        P = self.P if P is None else P
        time = self.time if time is None else time
        if not hb or len(P) == 0:
            return
        d = P.download()
        for name, b in hb:
            vs = [v for v in b.variables if v in names and name in self.priority_list.get(v, [])]
            if not vs:
                continue
            try:
                vals = b.evaluate_on_host(vs, time, d['lon'], d['lat'], d['z'], element_ID=d['ID'])
            except Exception as e:      # the reference catches every exception of a reader call (environment.py:640-668)
                self._reader_failed(name, b, e)
                continue
            for v in vs:
                first = self.priority_list[v][0] == name
                cur = P.env_download(v)
                take = np.isfinite(vals[v]) & (first | ~np.isfinite(cur))
                if take.any():
                    P.env_upload(v, np.where(take, vals[v], cur).astype(np.float32))

    def _advect_stage_split(self, scheme, factor):
        """advect_ocean_current under a Runge-Kutta scheme when a host-evaluated reader is among the sources of the
        current (physics_methods.py:623-680): the stages cannot run inside one kernel -- every stage is a
        get_environment call that has to reach the user's reader -- so the step is split at the stage boundaries.  Per
        stage: the stage positions on the device (geod.fwd along the float32 stage velocity over dt/2: odr_update_positions
        on a scratch particle set holding the elements' positions and IDs), the device sources sampled there
        (odr_env_sample at the stage time), the host readers evaluated at the same positions and merged by priority, the
        stage's uncertainty draws; then the float32 RK combination and update_positions of the elements."""
        U, V = 'x_sea_water_velocity', 'y_sea_water_velocity'
        P, dt = self.P, self.time_step.total_seconds()
        n = len(P)
        if n == 0:
            return
        d = P.download()
        u1, v1 = P.env_download(U), P.env_download(V)
        fac = factor * P.download_f32('current_drift_factor')          # factor*cdf: float32 (physics_methods.py:622)
        std, ustd = self._current_uncertainty()
        S = self.ctx.particles(n)
        try:
            S.append(d['lon'], d['lat'], z=d['z'], id=d['ID'])
            if self._world > 1:      # ensemble members: the elements the lower ranks hold come first
                S.set_rank_offset(self._below_active)

            def stage(k, u, v, t):
                if k:
                    S.upload(lon=d['lon'], lat=d['lat'])
                S.update_positions(u, v, 0.5 * dt)       # geod.fwd(lon, lat, az, speed*dt*.5), all float32 up to the distance
                S.env_sample([U, V], _epoch(t))
                self._sample_host_readers([U, V], P=S, time=t)
                for sd, uniform in ((std, False), (ustd, True)):          # environment.py:869-886, per get_environment call
                    if not sd:
                        continue
                    if self.rng == 'numpy':
                        draw = (lambda: np.random.uniform(-sd, sd, n)) if uniform else (lambda: np.random.normal(0, sd, n))
                        S.env_add_noise(U, V, sd, normals=(draw(), draw()), uniform=uniform)
                    else:
                        S.env_add_noise(U, V, sd, step=self.steps_calculation + ((k + 1) << 24), uniform=uniform)
                return S.env_download(U), S.env_download(V)

            half = self.time + self.time_step / 2
            u2, v2 = stage(0, u1, v1, half)
            if scheme == 'runge-kutta4':
                u3, v3 = stage(1, u2, v2, half)
                u4, v4 = stage(2, u3, v3, self.time + self.time_step)    # (half the distance, the full step later: :662-668)
                ue = (u1 + 2 * u2 + 2 * u3 + u4) / 6.0
                ve = (v1 + 2 * v2 + 2 * v3 + v4) / 6.0
                P.update_positions(ue * fac, ve * fac, dt)
            else:
                P.update_positions(fac * u2, fac * v2, dt)
        finally:
            S.close()

    def _add_uncertainty(self, names, current):
        """environment.py:869-891, in the reference's order of draws: current normal (x, y), current uniform (x, y), wind
        normal (x, y).  current=False: the current's share has been added inside the fused launch."""
        n = self.num_elements_active()
        for (vx, vy, key, uniform) in (
                ('x_sea_water_velocity', 'y_sea_water_velocity', 'drift:current_uncertainty', False),
                ('x_sea_water_velocity', 'y_sea_water_velocity', 'drift:current_uncertainty_uniform', True),
                ('x_wind', 'y_wind', 'drift:wind_uncertainty', False)):
            std = self.get_config(key)
            if not (std and std > 0 and vx in names and vy in names) or (vx == 'x_sea_water_velocity' and not current):
                continue
            if self.rng == 'numpy':
                draw = (lambda: np.random.uniform(-std, std, n)) if uniform else (lambda: np.random.normal(0, std, n))
                self.P.env_add_noise(vx, vy, std, normals=(draw(), draw()), uniform=uniform)
            else:
                self.P.env_add_noise(vx, vy, std, step=self.steps_calculation, uniform=uniform)

    def _current_uncertainty(self):
        return (self.get_config('drift:current_uncertainty') or 0.0, self.get_config('drift:current_uncertainty_uniform') or 0.0)

    # ---- PhysicsMethods (physics_methods.py:611-848)
    def advect_ocean_current(self, factor=1):
        if self._advected:       # already done by the fused launch of this step (run(), fused lane)
            self._advected = False
            return
        with self._readers_see_truncated_z():      # the Runge-Kutta stage calls are get_environment calls
            self._advect_ocean_current(factor)

    def _advect_ocean_current(self, factor=1):
        scheme = self.get_config('drift:advection_scheme')
        std, ustd = self._current_uncertainty()
        nstage = {'runge-kutta': 1, 'runge-kutta4': 3}.get(scheme, 0)
        if nstage and (any('x_sea_water_velocity' in b.variables or 'y_sea_water_velocity' in b.variables
                           for _, b in self._host_bindings()) or self._ensemble_current()):
            # (ensemble data: every stage call numbers the elements its block is handed -- the ones the reader covers at the
            # STAGE positions -- anew, interpolation/structured.py:119-135; a launch that holds all stages cannot)
            return self._advect_stage_split(scheme, factor)
        if nstage and (std > 0 or ustd > 0):
            # every Runge-Kutta stage is a get_environment call of the current: it carries the uncertainty too
            # (environment.py:869-886 inside physics_methods.py:638-670)
            if self.rng == 'numpy':
                n = self.num_elements_active()
                draws = []
                for _ in range(nstage):
                    call = []
                    if std > 0:
                        call += [np.random.normal(0, std, n), np.random.normal(0, std, n)]
                    if ustd > 0:
                        call += [np.random.uniform(-ustd, ustd, n), np.random.uniform(-ustd, ustd, n)]
                    draws.append(call)
                self.P.set_advect_noise(std, ustd, stage_draws=np.array(draws))
            else:
                self.P.set_advect_noise(std, ustd, step=self.steps_calculation)
        self.P.advect(scheme, _epoch(self.time), self.time_step.total_seconds(), factor)

    def _reduce_scalars(self, wdd=0.1):
        """The movers' global scalars on the host; sharded run: over the elements of all ranks, and installed on the
        device for the mover that follows (release with P.reduce_unpin())."""
        if self._world == 1:
            return self.P.reduce_scalars(wdd)
        if getattr(self, '_step_red', None) is not None:      # this step's collective already holds them
            return self.P.reduction_dict(self._step_red)
        self._timing_collectives += 2
        return self.P.reduction_dict(self.P.reduce_global(self._combine(), wdd, False))

    def _with_global_reduction(self, call, wdd=0.1, relwind=False):
        """Sharded run: the movers' global early-outs and maxima over the elements of ALL ranks (odr_reduce_local /
        _install); one process: the device reduces on its own."""
        if self._world == 1 or getattr(self, '_step_red', None) is not None:
            return call()          # one process: the device reduces on its own; sharded: installed by this step's collective
        self._timing_collectives += 2
        self.P.reduce_global(self._combine(), wdd, relwind)
        try:
            return call()
        finally:
            self.P.reduce_unpin()

    def advect_wind(self, factor=1):
        if self._world > 1 and self._calm_everywhere():
            return      # ("No wind drift" / calm on every rank: decided without the all-rank maxima)
        if self._world > 1:
            return self._with_global_reduction(lambda: self.P.advect_wind(
                self.time_step.total_seconds(), self.get_config('drift:wind_drift_depth', 0.1),
                self.get_config('drift:relative_wind'), factor), self.get_config('drift:wind_drift_depth', 0.1),
                self.get_config('drift:relative_wind'))
        self.P.advect_wind(self.time_step.total_seconds(), self.get_config('drift:wind_drift_depth', 0.1),
                           self.get_config('drift:relative_wind'), factor)

    def _identically_zero(self, v):
        """No reader delivers `v` and its fallback is 0: the variable is 0 for every element without looking."""
        live = [n for n in self.priority_list.get(v, []) if n in self.readers and self.readers[n].sid is not None]
        return not live and v in self.required_variables and self.get_config('environment:fallback:%s' % v) == 0

    def _stokes_arguments(self, factor=1):
        """The arguments of the device's Stokes drift for this step, or None when stokes_drift() returns early.  Leaves the
        reduction of a sharded run installed (the caller unpins it)."""
        if self.get_config('drift:stokes_drift') is False:
            return None
        profile = {'monochromatic': 0, 'exponential': 1, 'Phillips': 2, 'windsea_swell': 3}[
            self.get_config('drift:stokes_drift_profile', 'Phillips')]
        if self._identically_zero('sea_surface_wave_stokes_drift_x_velocity') and \
                self._identically_zero('sea_surface_wave_stokes_drift_y_velocity'):
            return None  # "No Stokes drift velocity available" (physics_methods.py:799-804) without a device round trip
        r = self._reduce_scalars(self.get_config('drift:wind_drift_depth', 0.1))
        if r['stokes_sum_max'] == 0:
            return None
        # provenance of Hs / Tp (physics_methods.py:893-943, :809-814)
        hs_mode = 0 if r['hs_max'] > 0 else (1 if r['wind_speed_max'] > 0 else 2)
        tp_mode = 1 if r['wind_speed_max'] >= 0 else 2   # Tp is not an OceanDrift variable: from wind (omega=5 when calm)
        return dict(profile=profile, hs_mode=hs_mode, tp_mode=tp_mode, factor=factor)

    def stokes_drift(self, factor=1):
        try:
            a = self._stokes_arguments(factor)
            if a is not None:
                self.P.stokes_drift(self.time_step.total_seconds(), a['profile'], a['hs_mode'], a['tp_mode'], a['factor'])
        finally:
            if self._world > 1 and getattr(self, '_step_red', None) is None:
                self.P.reduce_unpin()

    def _advect_wind_then_stokes_drift(self):
        """advect_wind() followed by stokes_drift() (update(), oceandrift.py:185-211).  One process and neither method
        overridden: the two movers run as ONE launch (odr_movers: same bits as the two calls)."""
        cls = type(self)
        if self._world > 1 or cls.advect_wind is not OceanDrift.advect_wind or cls.stokes_drift is not OceanDrift.stokes_drift:
            self.advect_wind()
            self.stokes_drift()
            return
        a = self._stokes_arguments()
        self.P.movers(self.time_step.total_seconds(),
                      wind=dict(wind_drift_depth=self.get_config('drift:wind_drift_depth', 0.1),
                                relative_wind=self.get_config('drift:relative_wind'), factor=1), stokes=a)

    def prepare_run(self):
        pass

    def update(self):
        raise NotImplementedError('Any trajectory model implementation must define an update method.')

    # ------------------------------------------------------------------ run (:1828-2340)
    def run(self, time_step=None, steps=None, time_step_output=None, duration=None, end_time=None,
            outfile=None, export_variables=None, export_buffer_length=100, stop_on_error=False):
        self._require('Ready')
        if self._sched is None:
            raise ValueError('Please seed elements before starting a run.')
        if outfile is not None:
            raise NotImplementedError('netCDF export is host-side I/O outside the hot path')
        if sum(x is not None for x in (steps, duration, end_time)) > 1:
            raise ValueError('Only one of "steps", "duration" and "end_time" may be provided simultaneously')
        if time_step is None:
            time_step = timedelta(minutes=self.get_config('general:time_step_minutes'))
        if not isinstance(time_step, timedelta):
            time_step = timedelta(seconds=time_step)
        self.time_step = time_step
        if time_step_output is None:
            m = self.get_config('general:time_step_output_minutes')
            time_step_output = time_step if m is None else timedelta(minutes=m)
        else:
            if not isinstance(time_step_output, timedelta):
                time_step_output = timedelta(seconds=time_step_output)
            if time_step_output.days >= 0 and time_step.days < 0:
                time_step_output = -time_step_output
        self.time_step_output = time_step_output      # (basemodel/__init__.py:1939-1958)
        ratio = time_step_output.total_seconds() / time_step.total_seconds()
        if ratio < 1:
            raise ValueError('Output time step must be equal or larger than calculation time step.')
        if not float(ratio).is_integer():
            raise ValueError('Ratio of calculation and output time steps must be an integer - given ratio is %s' % ratio)
        if time_step.total_seconds() < 0:
            self.start_time = self._sched['time'][int(np.argmax(self._sched['t_epoch']))]
        # simulation duration (:1960-2010): steps | duration | end_time | the end of the first reader; extended to a
        # multiple of the output time step
        if duration is None and end_time is None:
            if steps is not None:
                duration = steps * time_step
            else:
                ends = [r.end_time for r, _ in self._readers_host.values() if getattr(r, 'end_time', None) is not None]
                if not ends:
                    raise ValueError('Exactly one of the keywords steps, duration and end_time must be provided')
                end_time = min(ends)
        if duration is None:
            duration = end_time - self.start_time
        if time_step.days < 0 and duration.days >= 0:
            duration = -duration
        if np.sign(duration.total_seconds()) * np.sign(time_step.total_seconds()) < 0:
            raise ValueError('Time step must be negative if duration is negative.')
        ratio_duration_output = duration / time_step_output
        if not float(ratio_duration_output).is_integer():
            duration = np.ceil(ratio_duration_output) * time_step_output
        steps = int(round(duration.total_seconds() / time_step.total_seconds()))
        self.expected_steps_calculation = steps
        out_every = max(1, int(round(ratio)))
        # skip_if conditionals of required_variables (:1899-1906)
        for vn, var in list(self.required_variables.items()):
            if 'skip_if' in var:
                key, op, val = var['skip_if']
                cur = self.get_config(key)
                if {'is': cur is val, '==': cur == val, '!=': cur != val}[op]:
                    self.required_variables.pop(vn)
        self.time = self.start_time
        # the lon / lat box the elements can reach (:2017-2035): readers are prepared for it
        max_distance = self.get_config('drift:max_speed') * steps * abs(time_step.total_seconds())
        dlat = max_distance / 111000.
        dlon = dlat / np.cos(np.radians(np.mean(self._sched['lat'])))
        ext = np.array([max(-360, self._sched['lon'].min() - dlon), max(-89, self._sched['lat'].min() - dlat),
                        min(360, self._sched['lon'].max() + dlon), min(89, self._sched['lat'].max() + dlat)])
        if ext[2] == 360 and ext[0] < 0:
            ext[0] = 0
        self.simulation_extent = ext
        # the window the reader blocks are cut to (DeviceReaderBinding.set_extent): what can be covered between two looks
        # at the elements' box on top of it (_check_reader_windows)
        mlat = self.get_config('drift:max_speed') * self.WINDOW_CHECK_EVERY * abs(time_step.total_seconds()) / 111000.
        mlon = mlat / np.cos(np.radians(np.mean(self._sched['lat'])))
        self._block_extent = np.array([max(-360, ext[0] - mlon), max(-89, ext[1] - mlat), min(360, ext[2] + mlon), min(89, ext[3] + mlat)])
        self._all_at_start = bool((self._sched['t_epoch'] == _epoch(self.start_time)).all())
        self._finalize_environment(self.start_time, self.start_time + time_step)
        # Move point seeded on land to ocean (:2150-2158)
        if self.get_config('seed:ocean_only') is True and 'land_binary_mask' in self.required_variables:
            self._sched['lon'], self._sched['lat'], _ = self.closest_ocean_points(self._sched['lon'], self._sched['lat'])
        n_total = self.num_elements_total()
        lo_id, hi_id = 0, n_total
        self._n_global = n_total
        self._g_release = self._g_release_cum = None
        self._g_active_carry = self._below_active = 0
        self._timing_collectives, self._timing_collective_s, self._step_red = 0, 0.0, None
        if self._world > 1:     # this rank's contiguous range of element IDs; the others are never released here
            from . import distributed as D
            lo_id, hi_id = D.shard_range(n_total, self._rank, self._world)
            # the release schedule of ALL elements as a histogram over the steps (what _global_counts needs), then the
            # schedule itself cut down to this rank's range: the per-step host work is O(N / ranks) from here on
            te, dts = self._sched['t_epoch'], time_step.total_seconds()
            t_start = _epoch(self.start_time)
            if self._all_at_start:
                k = np.zeros(n_total, np.int64)
            else:
                # the step an element is released in, by the comparisons release_elements() makes on the same epoch values:
                # forward t0 <= t < t1, backward t0 >= t > t1, with t0 / t1 the epochs of start + i time_step
                edges = np.array([_epoch(self.start_time + i * time_step) for i in range(steps + 1)])
                sgn = 1.0 if dts > 0 else -1.0
                k = np.searchsorted(sgn * edges, sgn * te, side='right').astype(np.int64) - 1
            ok = (k >= 0) & (k < steps)
            self._g_release = np.bincount(k[ok], minlength=steps)
            self._g_release_cum = np.concatenate([[0], np.cumsum(self._g_release)])
            self._g_release_below = np.bincount(k[:lo_id][ok[:lo_id]], minlength=steps)
            self._sched = {kk: v[lo_id:hi_id] for kk, v in self._sched.items()}
            for attr in ('_released', '_n_unreleased', '_t_sched'):
                if hasattr(self, attr):
                    delattr(self, attr)
            self._all_at_start = False if n_total == 0 else self._all_at_start
        self._shard = (lo_id, hi_id)
        self._resolve_seafloor_seeds(0, hi_id - lo_id)
        self.P = self.ctx.particles(max(1, hi_id - lo_id))
        self.mode = 'Run'
        self.prepare_run()
        self._flush_elements()
        nout = steps // out_every + 1
        # float32 result buffer on the device (basemodel/__init__.py:2084-2105): element properties and
        # environment variables, [trajectory, time], NaN where an element does not exist
        hvars = self._history_variables(export_variables)
        self._hist = _ResultBuffer(self.ctx, max(1, hi_id - lo_id), nout, min(nout, max(1, int(export_buffer_length))), hvars,
                                   id_base=lo_id)
        times = []
        grid_sid = next((b.sid for b in self.readers.values() if b.is_grid() and b.sid is not None), None)
        # fused lane: the stock loop body and the stock OceanDrift.update order (current advection first), no
        # retirement between coastline and advection.  Uncertainties: with the device RNG (streams keyed by element
        # ID) they are added inside the launch; np.random draws are sized by the elements present at each call, which
        # the fused launch cannot honour -> call-by-call lane
        B = OpenDriftSimulation
        fused_lane = (not os.environ.get('ODR_RUN_UNFUSED') and getattr(type(self), 'update', None) is OceanDrift.update and
                      all(getattr(type(self), m) is getattr(B, m) for m in (
                          'advect_ocean_current', 'get_environment', 'interact_with_coastline', 'interact_with_seafloor',
                          'deactivate_outside', 'deactivate_elements')) and
                      (self.rng == 'device' or not (any(self._current_uncertainty()) or self.get_config('drift:wind_uncertainty'))) and
                      self.get_config('drift:max_age_seconds') is None and
                      not self._config.get('drift:water_column_stretching', {}).get('value') and
                      self._config.get('drift:truncate_ocean_model_below_m', {}).get('value') is None and
                      self.get_config('general:seafloor_action', 'lift_to_seafloor') in ('lift_to_seafloor', 'none') and
                      not self.get_config('general:coastline_approximation_precision') and
                      'x_sea_water_velocity' in self.required_variables and 'y_sea_water_velocity' in self.required_variables and
                      not self._host_bindings() and
                      not (self._ensemble_current() and self.get_config('drift:advection_scheme') != 'euler') and
                      # report_missing_variables comes BEFORE deactivate_outside in the loop (:2251-2253) but sits inside the
                      # launch: an element both outside the domain and without data must end as 'missing_data'
                      not (any(self.get_config('drift:deactivate_%s_of' % k) is not None for k in ('west', 'east', 'south', 'north'))
                           and self._can_be_missing(list(self.required_variables))))
        # the movers' global tests (calm, no Stokes drift, zero diffusivity, nothing at the surface) come out of the fused launch
        # instead of out of a pass over the arrays before the first mover (odr_ctx_set_step_reduce; a sharded run combines
        # its reductions over the ranks in the step's collective instead)
        self.ctx.set_step_reduce(fused_lane and self._world == 1, self.get_config('drift:wind_drift_depth', 0.1) or 0.0,
                                 bool(self.get_config('drift:relative_wind')))
        # Leeway lane: the loop body between two compactions + Leeway.update (leeway, current, jibes) in ONE launch
        # (odr_env_coast_leeway; what bench.py's C5 line times).  Stock methods, device RNG (the jibe draws are keyed by
        # element ID), no capsizing (its draws are sized by the elements that can capsize: call-by-call lane), nothing
        # the launch cannot honour -- as for the OceanDrift lane above.
        lw_update = getattr(type(self), 'leeway_lane_update', None)
        leeway_lane = (not os.environ.get('ODR_RUN_UNFUSED') and lw_update is not None and getattr(type(self), 'update', None) is lw_update and
                       all(getattr(type(self), m) is getattr(B, m) for m in (
                           'get_environment', 'interact_with_coastline', 'interact_with_seafloor', 'deactivate_outside',
                           'deactivate_elements', 'stokes_drift')) and
                       self.rng == 'device' and not self.get_config('processes:capsizing') and
                       not self.get_config('drift:current_uncertainty_uniform') and
                       self.get_config('drift:max_age_seconds') is None and
                       not self.get_config('general:coastline_approximation_precision') and
                       all(v in self.required_variables for v in ('x_wind', 'y_wind', 'x_sea_water_velocity', 'y_sea_water_velocity')) and
                       'sea_floor_depth_below_sea_level' not in self.required_variables and
                       not self._host_bindings() and
                       not (any(self.get_config('drift:deactivate_%s_of' % k) is not None for k in ('west', 'east', 'south', 'north'))
                            and self._can_be_missing(list(self.required_variables))))
        # re-sort interval: what bench.py's bare sequences were tuned to (a re-sort costs ~1.7 step-kernel launches at 10 M
        # elements; the layout decays with the distance travelled: 3-D current + mixing 24 steps, Leeway 48, the rest 16)
        sort_every = self.sort_every
        if sort_every is None:
            sort_every = 48 if leeway_lane else (24 if fused_lane and 'upward_sea_water_velocity' in self.required_variables else 16)
        # Variables nothing reads between this step's sample and the next one are not sampled in the fused lane (the sample
        # of `ocean_vertical_diffusivity` AT the element is only ever exported: the mixing scheme works on the profiles,
        # oceandrift.py:428-449).  They are sampled when the result buffer holds them and in the last step of the run, so
        # that `o.environment` and `o.result` come out as the reference's.
        stock_readers_of_it = isinstance(self, OceanDrift) and all(
            getattr(type(self), m) is getattr(OceanDrift, m) for m in ('vertical_mixing', 'update_terminal_velocity', 'update'))
        unread = [v for v in getattr(self, '_export_only_variables', ()) if v in self.required_variables and
                  v not in self._hist.variables and not self._can_be_missing([v])] if fused_lane and stock_readers_of_it else []
        if any(b.sid is not None and self.ctx._grids.get(b.sid, {}).get('members') for b in self.readers.values()):
            unread = []      # (ensemble data: the member numbering goes with the main-loop call as the reference makes it)
        # The mixing launch of the step enqueued before the host has read the status scan (vertical_mixing(_guarded=True)): when the
        # stock update() makes nothing but that launch between the scan and the end of the step -- no mover that could move an
        # element first (the wind is calm everywhere and there is no Stokes drift: decided on the host), stock methods, reader
        # diffusivity, device RNG.  Between output times only: the result buffer records z as it is BEFORE update().
        cls, OD = type(self), OceanDrift
        sx, sy = 'sea_surface_wave_stokes_drift_x_velocity', 'sea_surface_wave_stokes_drift_y_velocity'
        speculate = bool(
            fused_lane and not os.environ.get('ODR_NO_SPECULATION') and self.rng == 'device' and isinstance(self, OD) and
            all(getattr(cls, m) is getattr(OD, m) for m in ('vertical_mixing', 'vertical_advection', 'update_terminal_velocity',
                                                           '_advect_wind_then_stokes_drift', 'advect_wind', 'stokes_drift',
                                                           '_with_seafloor_action')) and
            self.get_config('drift:vertical_mixing') is True and
            self.get_config('vertical_mixing:diffusivitymodel') == 'environment' and
            ('x_wind' not in self.required_variables or self._calm_everywhere()) and
            (sx not in self.required_variables or self.get_config('drift:stokes_drift') is False or
             (self._identically_zero(sx) and self._identically_zero(sy))))
        # the step's collective may be finished BEHIND update() only when update() cannot register a status category on its own
        # (a subclass's vertical_mixing / update_terminal_velocity calling deactivate_elements would append it before the other
        # ranks' categories on a deferring rank and after them on a blocking one: ADVICE round 5): the stock methods only
        stock_update = isinstance(self, OD) and all(
            getattr(cls, m) is getattr(OD, m) for m in ('update', 'vertical_mixing', 'vertical_advection', 'update_terminal_velocity',
                                                        '_advect_wind_then_stokes_drift', 'advect_wind', 'stokes_drift',
                                                        '_with_seafloor_action', 'vertical_buoyancy'))
        self._vmix_speculated = False
        self.ctx.sync()
        t_loop = [time.perf_counter(), None]      # main-loop wall time (the reference keeps 'main loop' timers, basemodel :2174)
        # increase_age_and_retire comes after state_to_buffer in the loop: inside the fused launch only when the result
        # buffer does not hold age_seconds (it would see the age one step ahead)
        age_in_launch = 'age_seconds' not in self._hist.variables
        # host time of the loop body by phase, steps 1.. (the reference keeps `timers` of its main loop, basemodel :2174):
        # [seconds, longest single step] -- 'status read' is where the host waits for the device
        phases = {}
        pc = time.perf_counter

        def lap(name, t_from):
            t = pc()
            if i > 0:
                e = phases.setdefault(name, [0.0, 0.0])
                e[0] += t - t_from
                e[1] = max(e[1], t - t_from)
            return t
        # Until the first update_positions of a run the reference's elements.lon / lat are float32 ARRAYS (elements.py:71-88):
        # its first get_environment modulates the longitudes in float32 (variables.py:259-280, :914).  The main-loop sample of
        # the steps up to the first one that moves elements does the same (odr_ctx_set_position_class); the Runge-Kutta stage
        # calls inside update() work on float64 positions there and here.
        f32_first = self.steps_calculation == 0
        for i in range(steps):
            try:
                if f32_first:
                    self.ctx.set_position_class(True)
                if i == 1:
                    self.ctx.sync()
                    t_loop[1] = time.perf_counter()   # after the first step: seeding, first uploads and sort are behind
                t_ph = pc()
                self.release_elements()
                g_active, g_sched = self._global_counts(i)
                t_ph = lap('release', t_ph)
                if g_active == 0 and g_sched > 0:
                    self._state_to_buffer(i, out_every, times)   # (:2208)
                    self.steps_calculation += 1
                    self.time = self.time + self.time_step
                    continue
                self._ensure_reader_levels(self.time, self.time + self.time_step)
                t_ph = lap('reader levels', t_ph)
                # device layout maintenance: re-sort by grid cell every sort_every steps and whenever a
                # release added a sizeable share of new (unsorted) elements
                n_act = self.num_elements_active()
                if grid_sid is None:       # (gone after a failed re-cut of a reader's window: it may be back)
                    grid_sid = next((b.sid for b in self.readers.values() if b.is_grid() and b.sid is not None), None)
                if self.rng == 'device' and grid_sid is not None and sort_every and n_act > 65536 and \
                        (i % sort_every == 0 or self.newly_seeded * 20 > n_act):
                    # (the source id of a reader changes when its window is re-cut, and is gone when the re-cut failed)
                    grid_sid = next((b.sid for b in self.readers.values() if b.is_grid() and b.sid is not None), None)
                    if grid_sid is not None:
                        self.P.sort_by_cell(grid_sid, keep_environment=False)   # the step's sample follows
                t_ph = lap('layout', t_ph)
                one_collective, deferred = False, None
                # ensemble data in a sharded run: the stage calls of advect_ocean_current take the member by the rank among
                # the elements that are STILL active after this step's coastline / seafloor deactivations on ALL ranks --
                # known only from the step's collective, which the call-by-call lane makes between the two
                ens_sharded = self._world > 1 and any(
                    b.sid is not None and self.ctx._grids.get(b.sid, {}).get('members') for b in self.readers.values())
                if fused_lane and not ens_sharded:
                    # ONE launch for get_environment + coastline + seafloor + update_previous_state +
                    # advect_ocean_current (odr_env_coast_advect).  deactivate_outside only reads positions and goes
                    # first; the result buffer is written afterwards from the saved pre-advection position.
                    self.deactivate_outside()
                    names = [v for v in self.required_variables if i == steps - 1 or v not in unread]
                    action = self.get_config('general:coastline_action')
                    floor = ('sea_floor_depth_below_sea_level' in self.priority_list and
                             self.get_config('general:seafloor_action', 'lift_to_seafloor') == 'lift_to_seafloor')
                    std, ustd = self._current_uncertainty()
                    noisy = std > 0 or ustd > 0
                    if noisy:
                        self.P.set_advect_noise(std, ustd, step=self.steps_calculation)
                    self.P.env_coast_advect(
                        names, _epoch(self.time), self.get_config('drift:advection_scheme'), self.time_step.total_seconds(),
                        coastline=action if 'land_binary_mask' in names else 'none',
                        stranded_code=self._status_code('stranded') if action == 'stranding' else 1,
                        seeded_on_land_code=(self._status_code('seeded_on_land') if action == 'previous' and self._newly_any
                                             else 0),
                        store_previous=True, count=False, seafloor=floor,
                        missing_code=self._status_code('missing_data') if self._can_be_missing(names) else 0,
                        main_noise=noisy, age_dt=self.time_step.total_seconds() if age_in_launch else 0.0)
                    self._sampled = names
                    self._add_uncertainty(names, current=False)     # the wind's share
                    t_ph = lap('step launch', t_ph)
                    # ONE host read per step: how many elements stay + which new deactivation reasons occurred
                    self._vmix_speculated = False
                    want_red = self._needs_reductions()
                    early = None
                    if speculate and i % out_every != 0 and self.P.scan_status_begin():
                        if self._world > 1 and stock_update and not want_red and not os.environ.get('ODR_SYNC_COLLECTIVE'):
                            from . import distributed as D
                            if D.backend() == 'rccl':     # the step's ONE collective leaves behind the fold, ahead of the host's read
                                early = self._step_summary(None, None, False, start_only=True, from_scan=True)
                        launched = self.vertical_mixing(_guarded=True)      # (does nothing unless every element stays)
                        kept, flags = self.P.scan_status_end()
                        self._vmix_speculated = bool(launched) and kept == len(self.P)
                    else:
                        kept, flags = self.P.scan_status()
                    t_ph = lap('status read', t_ph)
                    all_stay = kept == len(self.P)      # nothing to backfill, nothing to compact on this rank
                    if early is not None and flags == 0:
                        deferred = (early, self._pending_status, kept)
                        self._pending_status = []
                    elif early is not None:      # a reason of this rank waits for its category: the collective is finished now
                        kept, flags = self._step_summary(None, None, False, handle=early)
                        self._resolve_status(flags)
                    elif self._world > 1 and stock_update and not want_red and flags == 0 and not os.environ.get('ODR_SYNC_COLLECTIVE'):
                        # sharded: nothing this rank does before the end of update() depends on the other ranks (no mover
                        # consults all-rank maxima, no element here carries a reason that waits for its category) -- the
                        # step's ONE collective is started here and finished behind the launches of update()
                        deferred = (self._step_summary(kept, flags, False, start_only=True), self._pending_status, kept)
                        self._pending_status = []       # (as _resolve_status leaves it: update() registers its own reasons anew)
                    else:
                        kept, flags = self._step_summary(kept, flags, want_red)   # the step's ONE collective
                        self._resolve_status(flags)
                    self._state_to_buffer(i, out_every, times, from_previous=True, all_stay=all_stay)
                    if not age_in_launch:
                        self.P.increase_age(self.time_step.total_seconds())
                    self.P.compact_apply()
                    self._advected = True
                elif leeway_lane and not ens_sharded:
                    self.deactivate_outside()
                    names = list(self.required_variables)
                    action = self.get_config('general:coastline_action')
                    # the launch jibes before the step's record is taken; the loop records first (basemodel/__init__.py:2276,
                    # :2293): at output steps the record reads the two properties a jibe changes from a copy taken here
                    jibed = [k for k, nm in enumerate(getattr(self, 'aux_properties', []))
                             if nm in ('crosswind_slope', 'orientation') and nm in self._hist.variables]
                    snap = bool(jibed) and i % out_every == 0
                    if snap:
                        for k in jibed:
                            self.P.snapshot_property(k)
                    _, split = self.P.env_coast_leeway(
                        names, _epoch(self.time), self.time_step.total_seconds(), self.get_config('capsizing:leeway_fraction'),
                        coastline=action if 'land_binary_mask' in names else 'none',
                        stranded_code=self._status_code('stranded') if action == 'stranding' else 1,
                        seeded_on_land_code=(self._status_code('seeded_on_land') if action == 'previous' and self._newly_any
                                             else 0),
                        store_previous=True, current_uncertainty=self.get_config('drift:current_uncertainty') or 0.0,
                        wind_uncertainty=self.get_config('drift:wind_uncertainty') or 0.0, step=self.steps_calculation,
                        split='return', count=False,
                        missing_code=self._status_code('missing_data') if self._can_be_missing(names) else 0)
                    self._sampled = names
                    t_ph = lap('step launch', t_ph)
                    kept, flags = self.P.scan_status()
                    t_ph = lap('status read', t_ph)
                    all_stay = kept == len(self.P)
                    kept, flags = self._step_summary(kept, flags, self._needs_reductions())   # the step's ONE collective
                    self._resolve_status(flags)
                    self._state_to_buffer(i, out_every, times, from_previous=3 if snap else 1, all_stay=all_stay)
                    self.P.increase_age(self.time_step.total_seconds())
                    self.P.compact_apply()
                    # wind, current and land mask from more than one reader: the library sampled, perturbed and applied the
                    # coastline; Leeway.update makes its own call on the compacted set
                    self._leeway_in_launch = not split
                else:
                    self.get_environment()
                    self.report_missing_variables()
                    self.deactivate_outside()
                    self.interact_with_coastline()
                    self.interact_with_seafloor()
                    max_age = self.get_config('drift:max_age_seconds')
                    one_collective = self._world > 1 and not max_age
                    if one_collective:      # sharded: the step's ONE collective, as in the fused lane
                        kept, flags = self.P.scan_status()
                        kept, flags = self._step_summary(kept, flags, self._needs_reductions())
                        self.P.set_rank_offset(self._below_active)     # (ensemble members of the stage calls, see above)
                        self._resolve_status(flags)
                    else:
                        self._resolve_status()
                    self._state_to_buffer(i, out_every, times)
                    self.P.increase_age(self.time_step.total_seconds(), max_age or 0.0,
                                        self._status_code('retired') if max_age else 0)
                    if one_collective:
                        self.P.compact_apply()
                    else:
                        self._resolve_status()
                        self.P.compact()
                        if ens_sharded:
                            # (drift:max_age_seconds: the retirements come after the point the step's one collective is made
                            # at) the stage calls of update() number the elements present NOW on all ranks: the elements the
                            # lower ranks kept, from a collective of its own
                            from . import distributed as D
                            rows = D.allgather_vector([float(len(self.P))])
                            self._timing_collectives += 1
                            self._below_active = int(round(rows[:self._rank, 0].sum()))
                            self.P.set_rank_offset(self._below_active)
                    self.P.store_previous()
                    if hasattr(self, '_store_environment_previous'):
                        self._store_environment_previous()
                if deferred is not None:
                    g_active = max(1, deferred[2])      # (decided for good when the collective is finished, below)
                elif self._world > 1 and (((fused_lane or leeway_lane) and not ens_sharded) or one_collective):
                    g_active = self._g_active           # from this step's collective
                elif self._world > 1:
                    newly = self._newly_any
                    g_active = self._global_counts()[0]
                    self._newly_any = newly
                else:
                    g_active = self._g_active = self.num_elements_active()
                t_ph = lap('bookkeeping', t_ph)
                if f32_first:
                    self.ctx.set_position_class(False)
                    f32_first = not g_active > 0
                try:
                    if g_active > 0:
                        self.update()
                        self._flush_elements()      # what a model's update() wrote into self.elements goes to the device
                    elif g_sched == 0:
                        raise ValueError('No more active or scheduled elements, quitting.')
                finally:
                    if deferred is not None:    # an open collective is finished whatever update() did: the ranks stay in step
                        g_kept, g_flags = self._step_summary(None, None, False, handle=deferred[0])
                if deferred is not None:
                    self._resolve_status(g_flags, pending=deferred[1])
                    if g_kept == 0 and g_sched == 0:
                        raise ValueError('No more active or scheduled elements, quitting.')
                self._advected = False
                self.horizontal_diffusion()
                self._step_release()
                self.time = self.time + self.time_step
                self.steps_calculation += 1
                t_ph = lap('update', t_ph)
            except Exception as e:
                self.ctx.set_position_class(False)
                if stop_on_error or self.steps_calculation <= 1:
                    raise
                logger.warning('The simulation stopped before requested end time was reached: %s', e)
                break
        self.ctx.set_position_class(False)
        self.ctx.sync()
        t_end = time.perf_counter()
        self.timing = {'main_loop_s': t_end - t_loop[0], 'steps': self.steps_calculation,
                       'collectives': self._timing_collectives, 'collective_s': self._timing_collective_s,
                       'reader_level_stall_s': sum(getattr(b, 'stall_s', 0.0) for b in self.readers.values()),
                       # rank 0 of a sharded run: levels its worker thread had read ahead / read inline, and the time the worker spent reading
                       'reader_thread': dict(zip(('hits', 'misses', 'worker_s'), map(sum, zip(*([(0, 0, 0.0)] + [
                           tuple(x + getattr(getattr(b, '_ahead', None), k, 0) for x, k in zip(getattr(b, '_ahead_stats', (0, 0, 0.0)),
                                                                                              ('hits', 'misses', 'worker_s')))
                           for b in self.readers.values()]))))),
                       'steady_ms_per_step': (1e3 * (t_end - t_loop[1]) / max(1, self.steps_calculation - 1)) if t_loop[1] else None,
                       'host_phases_ms_per_step': {k: (round(1e3 * v[0] / max(1, self.steps_calculation - 1), 4), round(1e3 * v[1], 3))
                                                   for k, v in phases.items()}}
        for b in self.readers.values():      # no read of the user's Reader is in flight when run() returns
            if hasattr(b, 'close_read_ahead'):
                b.close_read_ahead()
        self.interact_with_coastline(final=True)
        self._resolve_status()
        self._state_to_buffer(self.steps_calculation, out_every, times, final=True)
        self.mode = 'Result'
        self.result = dict(time=times, **self._hist.finish(len(times)))
        self.result_minmax = self._hist.minmax
        return self.result

    def _history_variables(self, export_variables):
        """Variables of the result buffer (:2068-2105): every element property and every required environment
        variable, or `export_variables` + ['lon', 'lat', 'status']."""
        elem = ['lon', 'lat', 'z', 'status', 'moving', 'age_seconds', 'wind_drift_factor', 'current_drift_factor',
                'terminal_velocity']
        aux = [a for a in getattr(self, 'aux_properties', []) if a not in getattr(self, 'internal_properties', ())]
        env = list(self.required_variables)
        if export_variables is not None:
            keep = set(export_variables) | {'lon', 'lat', 'status'}
            elem, aux, env = [v for v in elem if v in keep], [v for v in aux if v in keep], [v for v in env if v in keep]
        self._hist_aux = {name: ('property', list(getattr(self, 'aux_properties', [])).index(name)) for name in aux}
        return elem + aux + env

    def _state_to_buffer(self, step, out_every, times, final=False, from_previous=False, all_stay=False):   # :2384-2403, on the device
        # all_stay: the step's status scan found no deactivated element -- between output times there is nothing to record
        k = step // out_every
        if step % out_every == 0:            # output time: every element present
            if k < self._hist.ntimes:
                if len(self.P) > 0:
                    self._hist.record(self.P, k, False, self._hist_aux, from_previous)
                while len(times) <= k:       # result.time is a regular axis (:2088-2090)
                    times.append(self.start_time + len(times) * out_every * self.time_step)
        elif not final and not all_stay and k + 1 < self._hist.ntimes and len(self.P) > 0:
            self._hist.record(self.P, k + 1, True, self._hist_aux, from_previous)   # deactivated -> next output time (backfill)


class ElementsView:
    """`o.elements` (LagrangianArray, elements/elements.py:22-254) over the device arrays: see OpenDriftSimulation.elements."""
    _F64 = ('lon', 'lat', 'z')
    _F32 = ('wind_drift_factor', 'current_drift_factor', 'terminal_velocity')
    _RO = ('ID', 'status', 'age_seconds')

    def __init__(self, model):
        object.__setattr__(self, '_m', model)
        P = model.P
        d = P.download()
        order = np.argsort(d['ID'], kind='stable')             # device order -> ascending ID
        cur = {k: d[k][order] for k in ('lon', 'lat', 'z', 'ID', 'status', 'moving')}
        for k in self._F32 + ('age_seconds',):
            cur[k] = P.download_f32(k)[order]
        for slot, name in enumerate(getattr(model, 'aux_properties', [])):
            cur[name] = P.get_property(slot)[order]
        object.__setattr__(self, '_order', order)
        object.__setattr__(self, '_cur', cur)
        object.__setattr__(self, '_orig', {k: v.copy() for k, v in cur.items()})
        object.__setattr__(self, '_epoch', (len(P), model.steps_calculation, getattr(P, '_touch', 0)))
        P._view = self      # the next call that changes the device state flushes this view first (device._touching)

    def stale(self):
        P = self._m.P
        return (len(P), self._m.steps_calculation, getattr(P, '_touch', 0)) != self._epoch

    def __len__(self):
        return len(self._cur['ID'])

    def __getattr__(self, k):
        try:
            return self._cur[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if k not in self._cur:
            raise AttributeError('%s is not an element property' % k)
        if k in self._RO:
            raise AttributeError('%s is maintained by the model (use deactivate_elements for status)' % k)
        n = len(self)
        self._cur[k] = np.broadcast_to(np.asarray(v, dtype=self._orig[k].dtype), (n,)).copy()

    def flush(self):
        """Upload every property that differs from what was downloaded (assignment or in-place change)."""
        P, cur, orig = self._m.P, self._cur, self._orig
        if P.__dict__.get('_view') is self:
            P._view = None
        if len(P) != len(self):
            return      # the element set changed underneath: nothing sensible to write
        object.__setattr__(self, '_flushing', True)     # the uploads below read nothing back through this view
        try:
            self._flush_changed(P, cur, orig)
        finally:
            object.__setattr__(self, '_flushing', False)

    def _flush_changed(self, P, cur, orig):
        inv = np.empty(len(self), np.int64)
        inv[self._order] = np.arange(len(self))                # ascending ID -> device order
        changed = [k for k in cur if k not in self._RO and not np.array_equal(cur[k], orig[k], equal_nan=True)]
        core = {k: np.ascontiguousarray(cur[k][inv]) for k in changed if k in self._F64 + self._F32 + ('moving',)}
        if core:
            P.upload(**core)
        for slot, name in enumerate(getattr(self._m, 'aux_properties', [])):
            if name in changed:
                P.set_property(slot, np.ascontiguousarray(cur[name][inv]))
        for k in changed:
            orig[k] = cur[k].copy()


class _ResultBuffer:
    """export_buffer_length output times on the device (device.History); when full it is flushed to host
    chunks asynchronously and reset (:2489-2499)."""

    def __init__(self, ctx, ntraj, ntimes, nbuf, variables, id_base=0):
        self.ctx, self.ntraj, self.ntimes, self.nbuf, self.variables = ctx, ntraj, ntimes, nbuf, list(variables)
        self.id_base = id_base
        self.base, self.used, self.chunks, self.H, self.minmax = 0, 0, [], None, {}

    def _open(self, aux):
        if self.H is None:
            self.H = self.ctx.history(self.ntraj, self.nbuf, [aux.get(v, v) for v in self.variables], id_base=self.id_base)

    def record(self, P, k, only_deactivated, aux, from_previous=False):
        self._open(aux)
        while k - self.base >= self.nbuf:
            self._flush(self.nbuf)
        self.H.record(P, k - self.base, only_deactivated, position_from_previous=from_previous)
        self.used = max(self.used, k - self.base + 1)

    def _flush(self, nt):
        for v, name in zip(self.H.variables, self.variables):
            lo, hi = self.H.minmax(v)
            old = self.minmax.get(name)
            self.minmax[name] = (lo, hi) if old is None else (np.fmin(old[0], lo), np.fmax(old[1], hi))
        self.H.flush(0, nt)
        self.H.wait()
        self.chunks.append({name: self.H.array(v).copy() for v, name in zip(self.H.variables, self.variables)})
        self.H.reset()
        self.base += nt
        self.used = 0

    def finish(self, ntimes_written):
        if self.H is None:
            return {v: np.full((self.ntraj, ntimes_written), np.nan, np.float32) for v in self.variables}
        n_last = max(0, ntimes_written - self.base)
        if n_last > 0:
            self._flush(min(n_last, self.nbuf))
        self.H.close()
        out = {v: np.concatenate([c[v] for c in self.chunks], axis=1)[:, :ntimes_written] for v in self.variables}
        return out


class OceanDrift(OpenDriftSimulation):
    """opendrift/models/oceandrift.py:54-211"""
    element_properties = {'wind_drift_factor': 0.02, 'current_drift_factor': 1.0, 'terminal_velocity': 0.0}
    # environment variables whose value AT the element nothing in run() / OceanDrift.update reads (the mixing scheme takes the
    # diffusivity from the profiles, oceandrift.py:428-449; the element value is only exported): see run(), fused lane
    _export_only_variables = ('ocean_vertical_diffusivity', )
    required_variables = {
        'x_sea_water_velocity': {'fallback': 0},
        'y_sea_water_velocity': {'fallback': 0},
        'sea_surface_height': {'fallback': 0},
        'x_wind': {'fallback': 0},
        'y_wind': {'fallback': 0},
        'upward_sea_water_velocity': {'fallback': 0, 'skip_if': ['drift:vertical_advection', 'is', False]},
        'ocean_vertical_diffusivity': {'fallback': 0, 'skip_if': ['drift:vertical_mixing', 'is', False],
                                       'profiles': True},
        'horizontal_diffusivity': {'fallback': 0},
        'sea_surface_wave_significant_height': {'fallback': 0},
        'sea_surface_wave_stokes_drift_x_velocity': {'fallback': 0, 'skip_if': ['drift:stokes_drift', 'is', False]},
        'sea_surface_wave_stokes_drift_y_velocity': {'fallback': 0, 'skip_if': ['drift:stokes_drift', 'is', False]},
        'ocean_mixed_layer_thickness': {'fallback': 50, 'skip_if': ['drift:vertical_mixing', 'is', False]},
        'sea_floor_depth_below_sea_level': {'fallback': 10000},
        'land_binary_mask': {'fallback': None},
        # the inputs of drift:stokes_drift_profile = 'windsea_swell' (physics_methods.py:831-841).  The reference's stock
        # models do not list them (a run with that profile ends with an AttributeError there); here they are sampled
        # when -- and only when -- the profile is selected.
        'sea_surface_swell_wave_to_direction': {'fallback': 0, 'skip_if': ['drift:stokes_drift_profile', '!=', 'windsea_swell']},
        'sea_surface_swell_wave_peak_period_from_variance_spectral_density': {
            'fallback': 0, 'skip_if': ['drift:stokes_drift_profile', '!=', 'windsea_swell']},
        'sea_surface_swell_wave_significant_height': {'fallback': 0, 'skip_if': ['drift:stokes_drift_profile', '!=', 'windsea_swell']},
        'sea_surface_wind_wave_to_direction': {'fallback': 0, 'skip_if': ['drift:stokes_drift_profile', '!=', 'windsea_swell']},
        'sea_surface_wind_wave_mean_period': {'fallback': 0, 'skip_if': ['drift:stokes_drift_profile', '!=', 'windsea_swell']},
        'sea_surface_wind_wave_significant_height': {'fallback': 0, 'skip_if': ['drift:stokes_drift_profile', '!=', 'windsea_swell']},
    }

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._add_config({   # oceandrift.py:118-183
            'drift:vertical_advection': {'type': 'bool', 'default': True, 'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'drift:water_column_stretching': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_ADVANCED, 'description':
                                              'Elements follow the vertical motion of the water column as the sea surface height changes'},
            'drift:truncate_ocean_model_below_m': {'type': 'float', 'default': None, 'min': 0, 'max': 10000, 'level': CONFIG_LEVEL_ADVANCED,
                                                   'description': 'Ocean model data are sampled at this depth for elements below it'},
            'drift:vertical_advection_at_surface': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_ADVANCED,
                                                    'description': ''},
            'drift:vertical_mixing': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'drift:vertical_mixing_at_surface': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_ADVANCED,
                                                 'description': ''},
            'vertical_mixing:timestep': {'type': 'float', 'min': 0.1, 'max': 3600, 'default': 60,
                                         'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'vertical_mixing:diffusivitymodel': {'type': 'enum', 'default': 'environment',
                                                 'enum': ['environment', 'stepfunction', 'windspeed_Sundby1983',
                                                          'windspeed_Large1994', 'constant'],
                                                 'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'vertical_mixing:TSprofiles': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_ADVANCED, 'description':
                                           'Update T and S profiles within inner loop of vertical mixing.'},
            'vertical_mixing:background_diffusivity': {'type': 'float', 'min': 0, 'max': 1, 'default': 1.2e-5,
                                                       'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'drift:wind_drift_depth': {'type': 'float', 'default': 0.1, 'min': 0, 'max': 10,
                                       'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'drift:stokes_drift': {'type': 'bool', 'default': True, 'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'drift:stokes_drift_profile': {'type': 'enum', 'default': 'Phillips',
                                           'enum': ['monochromatic', 'exponential', 'Phillips', 'windsea_swell'],
                                           'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'general:seafloor_action': {'type': 'enum', 'default': 'lift_to_seafloor',
                                        'enum': ['none', 'lift_to_seafloor', 'deactivate', 'previous'],
                                        'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
            'seed:z': {'type': 'float', 'default': 0, 'min': -10000, 'max': 0, 'level': CONFIG_LEVEL_ESSENTIAL,
                       'description': ''},
            'seed:seafloor': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_ESSENTIAL,
                              'description': 'Elements are seeded at seafloor, and seeding depth (z) is neglected.'},
        })
        self._set_config_default('drift:max_speed', 2)

    def vertical_mixing(self, _guarded=False):   # oceandrift.py:397-571
        """_guarded (run(), fused lane): the launch is enqueued BEHIND the fold of the step's status scan and BEFORE the host has
        read it -- it does nothing unless every element stays (Particles.vmix(guarded=True)); returns whether it was enqueued.
        update() then finds the step's mixing done (`_vmix_speculated`) and returns here at once."""
        if getattr(self, '_vmix_speculated', False) and not _guarded:
            self._vmix_speculated = False
            self._vadv_fused = bool(self.get_config('drift:vertical_advection')) and \
                type(self).vertical_advection is OceanDrift.vertical_advection
            return
        if self.get_config('drift:vertical_mixing') is False:
            return False
        # no reader / constant for ocean_vertical_diffusivity: the profile is the fallback everywhere and the reference switches
        # to Large et al. (1994) (oceandrift.py:431-447).  (A reader that is listed but covers no element at all would do the same
        # there; here its fallback-filled profile is used.)
        model = self._effective_diffusivity_model()
        # (drift:truncate_ocean_model_below_m with reader diffusivity profiles: the columns end where the reference's reader cut
        # its block, _profile_level_cut -- golden c24c; whole columns for a reader that ignores the depth range asked, c24a)
        dt, dt_mix = self.time_step.total_seconds(), self.get_config('vertical_mixing:timestep')
        fuse = None
        if self.get_config('drift:vertical_advection') and type(self).vertical_advection is OceanDrift.vertical_advection:
            fuse = bool(self.get_config('drift:vertical_advection_at_surface'))
            self._vadv_fused = True
        kw = dict(mix_at_surface=self.get_config('drift:vertical_mixing_at_surface'), fuse_vertical_advection=fuse)
        if model == 'environment' and getattr(self, '_profile_levels', 0):
            kw['profile_levels'] = self._profile_levels
        if _guarded:
            if model != 'environment' or self.rng != 'device':
                self._vadv_fused = False
                return False
            action = self.get_config('general:seafloor_action', 'lift_to_seafloor')      # (as _with_seafloor_action; the
            if 'sea_floor_depth_below_sea_level' not in self.priority_list:                 # fused lane knows lift / none only)
                action = 'none'
            self.ctx.set_seafloor_action(action, 0)
            ok = self.P.vmix(_epoch(self.time), dt, dt_mix, step=self.steps_calculation, guarded=True, **kw)
            self._vadv_fused = False        # (set again by the call from update())
            return ok
        if self.rng == 'numpy':
            n, nt = self.num_elements_active(), abs(int(dt / dt_mix))
            kw['uniforms'] = np.stack([np.random.random(n) for _ in range(nt)])
        else:
            kw['step'] = self.steps_calculation
        if model == 'constant':     # oceandrift.py:448-452: the fallback value at every level, whatever the readers say
            self.ctx.bind('ocean_vertical_diffusivity', [], self.get_config('environment:fallback:ocean_vertical_diffusivity'))
            self._with_seafloor_action(lambda: self.P.vmix(_epoch(self.time), dt, dt_mix, **kw))
            self._bind_variables()
        elif model == 'environment':
            self._with_seafloor_action(lambda: self.P.vmix(_epoch(self.time), dt, dt_mix, **kw))
        else:   # get_diffusivity_profile (:385-395): raises ValueError('Unknown diffusivity model') like the reference
            bg = self.get_config('vertical_mixing:background_diffusivity')   # MLD.max() over all elements (oceandrift.py:430)
            self._with_global_reduction(lambda: self._with_seafloor_action(lambda: self.P.vmix_analytic(model, bg, dt, dt_mix, **kw)))

    def vertical_buoyancy(self):   # :352-368
        self._with_seafloor_action(lambda: self.P.vertical_buoyancy(self.time_step.total_seconds()))

    def vertical_advection(self):   # :315-350
        if self.get_config('drift:vertical_advection') is False or getattr(self, '_vadv_fused', False):
            self._vadv_fused = False
            return
        self.P.vertical_advection(self.time_step.total_seconds(), self.get_config('drift:vertical_advection_at_surface'))

    def update_terminal_velocity(self, Tprofiles=None, Sprofiles=None, z_index=None):
        pass

    def water_column_stretching(self):   # oceandrift.py:299-313
        """z + (sea_surface_height - its value of the previous step) * z / sea_floor_depth: the elements follow the water column.
        Rare option, evaluated on the host with NumPy's own dtypes (float32 environment, float64 z) like the reference."""
        if self.get_config('drift:water_column_stretching') is False or len(self.P) == 0:
            return
        prev = getattr(self, '_ssh_previous', None)
        if prev is None or 'sea_surface_height' not in self._sampled:
            logger.warning('water_column_stretching requires storing previous value of sea_surface_height')
            return
        ssh = self.P.env_download('sea_surface_height')
        depth = self.P.env_download('sea_floor_depth_below_sea_level')
        z = self.P.download()['z']
        delta_zeta = ssh - prev                                   # float32
        self.P.upload(z=z + delta_zeta * (z / depth))             # float64

    def _store_environment_previous(self):
        """update_previous_state() for the environment (basemodel/__init__.py:642-656): what update() sees as
        environment_previous.sea_surface_height -- the value of the previous step, the present one for new elements."""
        if self._config.get('drift:water_column_stretching', {}).get('value') is not True or 'sea_surface_height' not in self._sampled:
            return
        if getattr(self, '_ssh_by_id', None) is None:
            self._ssh_by_id = np.full(self._n_global, np.nan, np.float32)
        ids = self.P.ids()
        ssh = self.P.env_download('sea_surface_height')
        prev = self._ssh_by_id[ids]
        self._ssh_previous = np.where(np.isnan(prev), ssh, prev).astype(np.float32)
        self._ssh_by_id[ids] = ssh

    def update(self):   # oceandrift.py:185-211
        self.water_column_stretching()
        self.advect_ocean_current()
        self._advect_wind_then_stokes_drift()
        self.update_terminal_velocity()
        if self.get_config('drift:vertical_mixing') is True:
            self.vertical_mixing()
        else:
            self.vertical_buoyancy()
        self.vertical_advection()
