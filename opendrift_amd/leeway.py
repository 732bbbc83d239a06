"""Leeway (search and rescue) model on the device path.

Mirrors opendrift/models/leeway.py: `LeewayObj` element properties (:50-131), required variables
(:146-173), `seed_elements` with the per-element perturbation of the leeway coefficients
(:290-400) and `update()` (:430-494: leeway from wind + ambient current, Euler by construction, then
random jibing) -- the update is one HIP kernel (odr_leeway).

The object-class table OBJECTPROP.DAT is a data file of the reference (leeway.py:189-219 reads it from the directory of
its own module) and is not shipped here.  It is looked for, in this order: the constructor's first argument `d` (a path, as in
the reference), the environment variable ODR_OBJECTPROP, `models/OBJECTPROP.DAT` of an installed `opendrift` package (found
without importing it).  With a table, `seed_elements(object_type=26)` and `seed:object_type` work as in the reference
(:290-400); without one they RAISE -- a run never silently falls back to other coefficients.  The nine coefficients of a class
can also be handed over directly: `leeway_coefficients=dict(DWSLOPE=..., DWOFFSET=..., DWSTD=..., CWRSLOPE=..., CWROFFSET=...,
CWRSTD=..., CWLSLOPE=..., CWLOFFSET=..., CWLSTD=...)`, or explicit per-element arrays (downwind_slope=..., ...).  Capsizing
(`processes:capsizing`) runs on the device; the ASCII export is host bookkeeping outside the path.
"""
import os

import numpy as np

from .config import CONFIG_LEVEL_ESSENTIAL, CONFIG_LEVEL_BASIC, CONFIG_LEVEL_ADVANCED
from .oceandrift import OpenDriftSimulation

RIGHT, LEFT = 0, 1


def read_objectprop(path):
    """Parse an OBJECTPROP.DAT (leeway.py:186-219): three lines per object class -- key (and number), description,
    the nine coefficients -- up to the first blank line.  Returns {object_type (1-based): coefficients}."""
    lines = open(path).readlines()
    out = {}
    for i in range(len(lines) // 3 + 1):
        if 3 * i >= len(lines) or not lines[i * 3].strip():
            break
        arr = [float(x) for x in lines[i * 3 + 2].split()]
        out[i + 1] = dict(OBJKEY=lines[i * 3].split()[0].strip(), Description=lines[i * 3 + 1].strip(),
                          DWSLOPE=arr[0], DWOFFSET=arr[1], DWSTD=arr[2], CWRSLOPE=arr[3], CWROFFSET=arr[4],
                          CWRSTD=arr[5], CWLSLOPE=arr[6], CWLOFFSET=arr[7], CWLSTD=arr[8])
    return out


def find_objectprop(d=None):
    """Path of the object-class table or None: `d`, $ODR_OBJECTPROP, an installed opendrift's models/OBJECTPROP.DAT."""
    if d is not None:
        return d
    if os.environ.get('ODR_OBJECTPROP'):
        return os.environ['ODR_OBJECTPROP']
    try:
        import importlib.util
        spec = importlib.util.find_spec('opendrift')
    except (ImportError, ValueError):
        spec = None
    for loc in (spec.submodule_search_locations or []) if spec is not None else []:
        cand = os.path.join(loc, 'models', 'OBJECTPROP.DAT')
        if os.path.exists(cand):
            return cand
    return None


class Leeway(OpenDriftSimulation):
    element_properties = {'jibe_probability': 0.04, 'current_drift_factor': 1.0}
    # slot order of odr_particles_set_property (include/odrift.h)
    aux_properties = ['downwind_slope', 'crosswind_slope', 'downwind_offset', 'crosswind_offset', 'downwind_eps',
                      'crosswind_eps', 'jibe_probability', 'orientation', 'capsized']
    required_variables = {
        'x_wind': {'fallback': None},
        'y_wind': {'fallback': None},
        'x_sea_water_velocity': {'fallback': None},
        'y_sea_water_velocity': {'fallback': None},
        'sea_surface_wave_stokes_drift_x_velocity': {'fallback': 0, 'skip_if': ['drift:stokes_drift', 'is', False]},
        'sea_surface_wave_stokes_drift_y_velocity': {'fallback': 0, 'skip_if': ['drift:stokes_drift', 'is', False]},
        'land_binary_mask': {'fallback': None},
    }

    def __init__(self, d=None, *args, **kwargs):
        if d is not None and not os.path.exists(d):
            raise FileNotFoundError(d)                        # the reference's open(d) (leeway.py:193)
        path = find_objectprop(d)
        self.leewayprop = read_objectprop(path) if path is not None else None
        super().__init__(*args, **kwargs)
        if self.leewayprop:
            descriptions = [self.leewayprop[p]['Description'] for p in self.leewayprop]
            self._add_config({'seed:object_type': {'type': 'enum', 'enum': descriptions, 'default': descriptions[0],
                                                   'level': CONFIG_LEVEL_ESSENTIAL,
                                                   'description': 'Leeway object category for this simulation'}})   # :228-234
        self._add_config({
            'processes:capsizing': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'capsizing:leeway_fraction': {'type': 'float', 'default': 0.4, 'min': 0, 'max': 1,
                                          'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'capsizing:wind_threshold': {'type': 'float', 'default': 30, 'min': 0, 'max': 50,
                                         'level': CONFIG_LEVEL_BASIC, 'description': ''},          # leeway.py:262-281
            'capsizing:wind_threshold_sigma': {'type': 'float', 'default': 5, 'min': 0, 'max': 20,
                                               'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'drift:stokes_drift': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_ADVANCED, 'description': ''},
        })
        self._set_config_default('drift:max_speed', 5)

    def _object_class(self, object_type):
        """leeway.py:305-322: the class number given, or looked up from `seed:object_type` (OBJKEY or description)."""
        if self.leewayprop is None:
            raise FileNotFoundError(
                'Leeway object classes need the reference\'s OBJECTPROP.DAT, which is not shipped: pass its path as '
                'Leeway(d=...), set ODR_OBJECTPROP, install opendrift, or hand over leeway_coefficients=dict(DWSLOPE=...)')
        if object_type is None:
            name = self.get_config('seed:object_type')
            for k, row in self.leewayprop.items():
                if row['OBJKEY'] == name or row['Description'] == name:
                    return row
            raise ValueError('Object %s not available' % name)
        if object_type not in self.leewayprop:
            raise KeyError(object_type)
        return self.leewayprop[object_type]

    def seed_elements(self, lon, lat, object_type=None, leeway_coefficients=None, **kwargs):
        """leeway.py:290-400.  The coefficients of the object class -- `object_type` (number in OBJECTPROP.DAT), else
        `leeway_coefficients` (a table row handed over), else `seed:object_type` -- perturbed per element like the reference,
        with its draws made BEFORE the base class draws the seeding radius (:327-346 before :386).  Extension: explicit
        per-element arrays downwind_slope=..., crosswind_slope=..., ... (no draws)."""
        explicit = {k: kwargs.pop(k) for k in list(kwargs) if k in self.aux_properties and k != 'jibe_probability'}
        n_before = 0 if self._sched is None else len(self._sched['lon'])
        jibe = kwargs.pop('jibe_probability', None)
        coefficient_arrays = [k for k in explicit if k != 'capsized']
        if object_type is not None or (leeway_coefficients is None and not coefficient_arrays):
            if leeway_coefficients is not None:
                raise ValueError('object_type and leeway_coefficients are both given')
            leeway_coefficients = self._object_class(object_type)
        props = None
        if leeway_coefficients is not None:
            lon_a = np.atleast_1d(lon).ravel()
            if kwargs.get('number') is not None:              # :297-302
                number = kwargs['number']
            elif len(lon_a) > 1:
                number = len(lon_a)
            else:
                number = self.get_config('seed:number')
            props = self._perturbed_coefficients(leeway_coefficients, number, explicit)
        super().seed_elements(lon, lat, **kwargs)
        number = len(self._sched['lon']) - n_before
        if props is not None and len(props['orientation']) != number:
            raise ValueError('Leeway.seed_elements: %d elements were seeded, the coefficients were drawn for %d'
                             % (number, len(props['orientation'])))
        if props is None:
            defaults = dict(downwind_slope=1, crosswind_slope=1, downwind_offset=0, crosswind_offset=0, downwind_eps=0,
                            crosswind_eps=0, orientation=1, capsized=0)     # LeewayObj defaults (:50-131)
            props = {k: np.asarray(explicit.get(k, v), dtype=np.float64) * np.ones(number) for k, v in defaults.items()}
        if jibe is not None:
            self._sched['jibe_probability'][n_before:] = np.float32(jibe)
        for k, v in props.items():
            v = np.asarray(v, dtype=np.float32)
            self._sched[k] = v if n_before == 0 else np.concatenate([self._sched[k], v])

    @staticmethod
    def _perturbed_coefficients(c, number, explicit):
        """leeway.py:323-374: orientation, slopes / offsets of the class, N(0, std) perturbations of the two slopes."""
        orientation = np.r_[:number] % 2          # odd numbered particles are left-drifting (:318-320)
        ones = np.ones(number)
        downwind_slope, downwind_offset = ones * c['DWSLOPE'], ones * c['DWOFFSET']
        # avoid negative downwind slopes (:331-339): the reference draws randn(1) element by element and draws again
        # while slope + eps / 20 < 0.  The slope is the same for every element, so the elements take, in order, the
        # draws of the stream that pass -- whole batches of the legacy generator (randn(n) continues the stream exactly
        # like n calls of randn(1)), the shortfall drawn again, never past the last draw the loop would have made
        epsdw = np.empty(0)
        while len(epsdw) < number:
            e = np.random.randn(number - len(epsdw)) * c['DWSTD']
            epsdw = np.concatenate([epsdw, e[~(c['DWSLOPE'] + e / 20.0 < 0.0)]])
        rcw = np.random.randn(number)
        crosswind_slope = np.where(orientation == RIGHT, c['CWRSLOPE'], c['CWLSLOPE'])
        crosswind_offset = np.where(orientation == RIGHT, c['CWROFFSET'], c['CWLOFFSET'])
        crosswind_eps = np.where(orientation == RIGHT, rcw * c['CWRSTD'], rcw * c['CWLSTD'])
        props = dict(downwind_slope=downwind_slope, crosswind_slope=crosswind_slope,
                     downwind_offset=downwind_offset, crosswind_offset=crosswind_offset, downwind_eps=epsdw,
                     crosswind_eps=crosswind_eps, orientation=orientation,
                     capsized=np.asarray(explicit.get('capsized', 0), dtype=np.float64) * np.ones(number))   # :373-374
        return props

    def update(self):   # leeway.py:430-494
        dt = self.time_step.total_seconds()
        frac = self.get_config('capsizing:leeway_fraction')
        if self.get_config('processes:capsizing'):   # :438-455
            thr, sig = self.get_config('capsizing:wind_threshold'), self.get_config('capsizing:wind_threshold_sigma')
            if self.rng == 'numpy':
                can = int((self.P.get_property(8) == (0.0 if dt >= 0 else 1.0)).sum())
                if can > 0:
                    self.P.leeway_capsize(dt, thr, sig, uniforms=np.random.rand(can))
            else:
                self.P.leeway_capsize(dt, thr, sig, step=self.steps_calculation)
        if getattr(self, '_leeway_in_launch', False):
            self._leeway_in_launch = False      # run(), Leeway lane: leeway, current and jibes were part of this step's launch
        elif self.rng == 'numpy':
            self.P.leeway(dt, frac, uniforms=np.random.random(self.num_elements_active()))
        else:
            self.P.leeway(dt, frac, step=self.steps_calculation)
        self.stokes_drift()

    leeway_lane_update = update      # run() takes the one-launch lane only while update() is this function
