"""OpenOil on the device path (the oil physics inside the mixing loop).

Mirrors opendrift/models/openoil/openoil.py for what lies on the advection path:

  required_variables (:221-296), element properties on the path (Oil: diameter, density, viscosity,
  oil_film_thickness, :105-208), config defaults (:493-499), seed_elements' droplet sizes for elements seeded below
  the surface and `keep_droplet_diameter` (:1629-1700);
  update() = (weathering) -> update_terminal_velocity + vertical_mixing -> vertical advection -> advect_oil()
  (:1218-1239), the reverse of OceanDrift.update;
  advect_oil() = advect_ocean_current + advect_wind + stokes_drift (:1179-1216, no sea ice: k_ice = 0);
  the mixing loop as OpenOil runs it (oceandrift.py:505-565 with the OpenOil hooks): droplet terminal velocities
  (Tkalich et al. 2002, :922-998) in every sub-step, slick formation (:1056-1061), wave entrainment
  (Li et al. 2017 rate, Johansen et al. 2015 / Li et al. 2017 droplet spectra, :1000-1054, :1072-1172) --
  all of it on the device (odr_oil_prepare_mixing + the oil variants of the mixing kernels).

Outside the path and not here: the ADIOS oil database and the NOAA weathering chemistry (evaporation,
emulsification, dispersion, biodegradation).  What the database contributes to THIS path is three numbers per oil --
density, kinematic viscosity, oil-water interfacial tension -- which `set_oiltype` / `seed_elements(oil_type=...)`
take as a dict ({'density': kg/m3, 'viscosity': m2/s, 'oil_water_interfacial_tension': N/m}; the reference accepts
a json dict for `oil_type` too, :1708-1710).  `oil_weathering()` keeps only what the reference's
oil_weathering_noaa does with all processes off: the sea water temperature is in Kelvin afterwards (:722-724).

np.random parity (rng='numpy'): the reference draws the intrusion depths of the entrained elements compacted
(np.random.uniform(0, mean(zb), entrained.sum()), :1048-1049); which elements are entrained is decided on the
device, so this mirror draws one number per element and sub-step instead -- same distributions, a different
consumption of the stream.  Exact parity with recorded draws is tested through Particles.vmix_oil
(tests/test_gpu_oil.py, golden c9).
"""
import numpy as np

from . import _abi
from .config import CONFIG_LEVEL_ADVANCED, CONFIG_LEVEL_BASIC
from .device import sea_water_density_default
from .oceandrift import OceanDrift, _epoch

_DEFAULT_OIL = {'density': 880.0, 'viscosity': 0.005, 'oil_water_interfacial_tension': 0.03}   # Oil defaults (:117-130)


class OpenOil(OceanDrift):
    element_properties = dict(OceanDrift.element_properties, wind_drift_factor=0.03)   # openoil.py:133-140
    # slot order of odr_particles_set_property (include/odrift.h ODR_OIL_*)
    aux_properties = list(_abi.OIL_PROPERTIES)
    internal_properties = ('diameter_if_entrained',)   # device scratch of prepare_vertical_mixing, not an Oil element variable
    required_variables = {   # openoil.py:221-296 (the second-moment wave period is not a device variable)
        'x_sea_water_velocity': {'fallback': None},
        'y_sea_water_velocity': {'fallback': None},
        'x_wind': {'fallback': None},
        'y_wind': {'fallback': None},
        'sea_surface_height': {'fallback': 0},
        'upward_sea_water_velocity': {'fallback': 0, 'skip_if': ['drift:vertical_advection', 'is', False]},
        'sea_surface_wave_significant_height': {'fallback': 0},
        'sea_surface_wave_stokes_drift_x_velocity': {'fallback': 0, 'skip_if': ['drift:stokes_drift', 'is', False]},
        'sea_surface_wave_stokes_drift_y_velocity': {'fallback': 0, 'skip_if': ['drift:stokes_drift', 'is', False]},
        'sea_surface_wave_period_at_variance_spectral_density_maximum': {'fallback': 0},
        'sea_ice_area_fraction': {'fallback': 0},        # advect_oil in ice (openoil.py:263-275, 1179-1216)
        'sea_ice_x_velocity': {'fallback': 0},
        'sea_ice_y_velocity': {'fallback': 0},
        'sea_water_temperature': {'fallback': 10},
        'sea_water_salinity': {'fallback': 34},
        'sea_floor_depth_below_sea_level': {'fallback': 10000},
        'horizontal_diffusivity': {'fallback': 0},
        'ocean_vertical_diffusivity': {'fallback': 0.02, 'skip_if': ['drift:vertical_mixing', 'is', False], 'profiles': True},
        'land_binary_mask': {'fallback': None},
        'ocean_mixed_layer_thickness': {'fallback': 50, 'skip_if': ['drift:vertical_mixing', 'is', False]},
    }

    def __init__(self, *args, **kwargs):
        kwargs.pop('weathering_model', None)
        super().__init__(*args, **kwargs)
        self._add_config({
            'processes:evaporation': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'processes:emulsification': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'processes:dispersion': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'wave_entrainment:droplet_size_distribution': {
                'type': 'enum', 'enum': ['Johansen et al. (2015)', 'Li et al. (2017)'],
                'default': 'Johansen et al. (2015)', 'level': CONFIG_LEVEL_ADVANCED, 'description': ''},     # :457-467
            'wave_entrainment:entrainment_rate': {'type': 'enum', 'enum': ['Li et al. (2017)'], 'default': 'Li et al. (2017)',
                                                  'level': CONFIG_LEVEL_ADVANCED, 'description': ''},        # :468-478
            'seed:droplet_size_distribution': {'type': 'enum', 'enum': ['uniform', 'normal', 'lognormal'],
                                               'default': 'uniform', 'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'seed:droplet_diameter_mu': {'type': 'float', 'default': 0.001, 'min': 1e-8, 'max': 1,
                                         'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'seed:droplet_diameter_sigma': {'type': 'float', 'default': 0.0005, 'min': 1e-8, 'max': 1,
                                            'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'seed:droplet_diameter_min_subsea': {'type': 'float', 'default': 0.0005, 'min': 1e-8, 'max': 1,
                                                 'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'seed:droplet_diameter_max_subsea': {'type': 'float', 'default': 0.005, 'min': 1e-8, 'max': 1,
                                                 'level': CONFIG_LEVEL_BASIC, 'description': ''},
        })
        self._set_config_default('drift:vertical_advection', False)
        self._set_config_default('drift:vertical_advection_at_surface', False)
        self._set_config_default('drift:vertical_mixing', True)
        self._set_config_default('drift:vertical_mixing_at_surface', False)
        self._set_config_default('drift:current_uncertainty', 0.05)
        self._set_config_default('drift:wind_uncertainty', 0.5)
        self._set_config_default('drift:max_speed', 1.3)
        self.oiltype = None
        self.keep_droplet_diameter = False
        self._temperature_in_kelvin = False

    def set_config(self, key, value):
        if key.startswith('processes:') and value is True:
            raise NotImplementedError('oil weathering is outside the advection hot path (DESIGN.md section 9)')
        super().set_config(key, value)

    # ---- oil properties: the three numbers the ADIOS database contributes to this path
    def set_oiltype(self, oiltype):
        if not isinstance(oiltype, dict):
            raise ValueError('The ADIOS oil database is not part of the device path: pass the oil as a dict with '
                             'density [kg/m3], viscosity [m2/s] and oil_water_interfacial_tension [N/m]')
        unknown = set(oiltype) - set(_DEFAULT_OIL) - {'name'}
        if unknown:
            raise ValueError('Unknown oil properties: %s' % sorted(unknown))
        self.oiltype = dict(_DEFAULT_OIL, **{k: v for k, v in oiltype.items() if k != 'name'})
        self.oil_name = oiltype.get('name', 'user-defined oil')
        self.oil_water_interfacial_tension = float(self.oiltype['oil_water_interfacial_tension'])

    def seed_elements(self, lon, lat, time=None, oil_type=None, diameter=None, oil_film_thickness=0.001, **kwargs):
        """openoil.py:1629-1759.  z < 0 without `diameter`: droplet sizes from seed:droplet_size_distribution, drawn
        with np.random like the reference (and before the positions, as there)."""
        if 'oiltype' in kwargs:
            raise ValueError('Seed argument *oiltype* is deprecated, use *oil_type* instead')      # :1701-1702
        kwargs.pop('m3_per_hour', None)     # oil mass is weathering bookkeeping, not on the path
        if oil_type is not None:
            self.set_oiltype(oil_type)
        elif self.oiltype is None:
            self.set_oiltype(dict(_DEFAULT_OIL))
        lon_a = np.atleast_1d(lon).ravel()
        number = kwargs.get('number')
        if number is None:
            number = len(lon_a) if len(lon_a) > 1 else self.get_config('seed:number')
        self.keep_droplet_diameter = diameter is not None                       # :1638-1645
        z = kwargs.get('z')
        if z is None:                                                           # :1646-1650
            z = 'seafloor' if ('seed:seafloor' in self._config and self.get_config('seed:seafloor') is True) else \
                self.get_config('seed:z')
        if isinstance(z, str) and z[0:8] == 'seafloor':
            zz = -np.ones(number)      # "z = -np.ones(number)" (:1655-1657): at the sea floor every element is a droplet
        else:
            zz = np.atleast_1d(z) * np.ones(number) if np.size(z) in (1, number) else np.atleast_1d(z)
        if np.sum(zz < 0) > 0 and diameter is None:                             # :1659-1700
            dsd = self.get_config('seed:droplet_size_distribution')
            if dsd == 'uniform':
                diameter = np.random.uniform(self.get_config('seed:droplet_diameter_min_subsea'),
                                             self.get_config('seed:droplet_diameter_max_subsea'), number)
            elif dsd == 'normal':
                diameter = np.random.normal(self.get_config('seed:droplet_diameter_mu'),
                                            self.get_config('seed:droplet_diameter_sigma'), number)
            else:
                mu, sigma2 = self.get_config('seed:droplet_diameter_mu'), self.get_config('seed:droplet_diameter_sigma')**2
                s2 = np.log(sigma2 / mu**2 + 1)
                diameter = np.random.lognormal(np.log(mu) - s2 / 2, s2**0.5, number)
        n_before = 0 if self._sched is None else len(self._sched['lon'])
        super().seed_elements(lon, lat, time, **kwargs)
        n_new = len(self._sched['lon']) - n_before
        if diameter is not None and np.size(diameter) not in (1, n_new):
            raise ValueError('diameter has length %s, but %s elements were seeded' % (np.size(diameter), n_new))
        props = dict(diameter=0.0 if diameter is None else diameter, density=self.oiltype['density'],
                     viscosity=self.oiltype['viscosity'], oil_film_thickness=oil_film_thickness, diameter_if_entrained=0.0)
        for k, v in props.items():
            v = np.asarray(v, dtype=np.float32) * np.ones(n_new, np.float32)
            self._sched[k] = v if n_before == 0 else np.concatenate([self._sched[k], v])

    # ---- PhysicsMethods pieces that OpenOil evaluates differently from OceanDrift
    def _wave_modes(self):
        """Provenance of wave height and period (physics_methods.py:893-943): from readers when any value is > 0,
        else from the wind; the wind-derived period has passed through the float32 environment
        (calculate_missing_environment_variables, :876-883) because OpenOil requires the variable."""
        r = self._reduce_scalars(self.get_config('drift:wind_drift_depth', 0.1))
        if self._world > 1:
            self.P.reduce_unpin()
        return r, (0 if r['hs_max'] > 0 else 1), (0 if r['tp_max'] > 0 else 3)

    def stokes_drift(self, factor=1):
        if self.get_config('drift:stokes_drift') is False:
            return
        profile = {'monochromatic': 0, 'exponential': 1, 'Phillips': 2, 'windsea_swell': 3}[
            self.get_config('drift:stokes_drift_profile', 'Phillips')]
        if profile == 3 and 'sea_surface_swell_wave_to_direction' not in self.required_variables:
            # OpenOil does not list the swell / wind-sea variables (the reference stops with an AttributeError here)
            raise AttributeError("'Environment' has no 'sea_surface_swell_wave_to_direction': add the six windsea_swell "
                                 "variables to required_variables in a subclass")
        r, hs_mode, tp_mode = self._wave_modes()
        if r['stokes_sum_max'] == 0:
            return
        self._with_global_reduction(lambda: self.P.stokes_drift(self.time_step.total_seconds(), profile, hs_mode, tp_mode, factor))

    def oil_weathering(self):   # :673-680, :717-724 with every process off
        self._temperature_in_kelvin = self.time_step.days >= 0

    def update_terminal_velocity(self, Tprofiles=None, Sprofiles=None, z_index=None):
        pass    # evaluated inside the mixing kernel in every sub-step (and before the first one, like :1229)

    def vertical_mixing(self):   # oceandrift.py:397-571 with OpenOil's prepare_vertical_mixing / surface hooks
        if self.get_config('drift:vertical_mixing') is False:
            return
        model = self.get_config('vertical_mixing:diffusivitymodel')
        if model == 'environment' and not any(self.readers[n].sid is not None
                                              for n in self.priority_list.get('ocean_vertical_diffusivity', [])):
            model = 'windspeed_Large1994'       # oceandrift.py:431-447
        if model not in ('environment', 'constant') and model not in _abi.DIFFUSIVITY:
            raise ValueError('Unknown diffusivity model: ' + str(model))
        dt, dt_mix = self.time_step.total_seconds(), self.get_config('vertical_mixing:timestep')
        _, hs_mode, tp_mode = self._wave_modes()
        kw = dict(keep_droplet_diameter=self.keep_droplet_diameter, hs_mode=hs_mode, tp_mode=tp_mode,
                  temperature_to_kelvin=self._temperature_in_kelvin)
        if self.rng == 'numpy':
            n, nt = self.num_elements_active(), abs(int(dt / dt_mix))
            uni = dict(diameter=np.random.random(n), mix=np.empty((nt, n)), entrain=np.empty((nt, n)),
                       intrusion=np.empty((nt, n)))
            for it in range(nt):
                uni['mix'][it] = np.random.random(n)
                uni['entrain'][it] = np.random.uniform(0, 1, n)
                uni['intrusion'][it] = np.random.uniform(0, 1, n)
            kw['uniforms'] = uni
        else:
            kw['step'] = self.steps_calculation
        if self._world > 1:   # OpenOil's means over ALL elements: np.mean(dV_50), np.mean(1.5 Hs) (openoil.py:1099-1101,1047)
            from . import distributed as D
            self._timing_collectives = getattr(self, '_timing_collectives', 0) + 1
            self.P.oil_global_stats(lambda v: D.allreduce_scalars(v, 'sum'), self.oil_water_interfacial_tension,
                                    self.get_config('wave_entrainment:droplet_size_distribution'),
                                    sea_water_density_default(), hs_mode=hs_mode)
        mix = lambda: self._with_seafloor_action(lambda: self.P.vmix_oil(
            model, self.get_config('vertical_mixing:background_diffusivity'), dt, dt_mix,
            self.oil_water_interfacial_tension, self.get_config('wave_entrainment:droplet_size_distribution'),
            sea_water_density=sea_water_density_default(), t_epoch=_epoch(self.time),
            mix_at_surface=self.get_config('drift:vertical_mixing_at_surface'),
            profile_levels=getattr(self, '_profile_levels', 0), **kw))       # (OceanDrift._profile_level_cut: truncation + reader profiles)
        # a wind-parameterised diffusivity needs MLD.max() over all elements (oceandrift.py:430)
        self._with_global_reduction(mix)

    def advect_oil(self):   # openoil.py:1179-1216
        if self._identically_zero('sea_ice_area_fraction'):
            # no reader delivers the ice concentration: k_ice = 0 and factor_stokes = 1 for every element, and
            # advect_with_sea_ice(factor=0) is a zero-length move (position unchanged)
            self.advect_ocean_current(factor=1)
            self.advect_wind(factor=1)
            self.stokes_drift(1)
            return
        # Nordam et al. (2019) / Arneborg (2017): per-element factors from the sampled float32 concentration, derived
        # inside the kernels (odr_set_element_factor)
        try:
            self.P.set_element_factor('ice_current')      # 1 - k_ice
            self.advect_ocean_current(factor=1)
            self.advect_wind(factor=1)
            self.P.set_element_factor('ice_stokes')       # (0.7 - A) / 0.7, 0 above 70 %
            self.stokes_drift(1)
            self.P.set_element_factor('ice_drift')        # k_ice
            self.advect_with_sea_ice()
        finally:
            self.P.set_element_factor(None)

    def advect_with_sea_ice(self, factor=1):   # physics_methods.py:693-710
        self.P.advect_sea_ice(self.time_step.total_seconds(), factor)

    def update(self):       # openoil.py:1218-1239
        self.oil_weathering()
        if self.get_config('drift:vertical_mixing') is True:
            self.update_terminal_velocity()
            self.vertical_mixing()
        if self.get_config('drift:vertical_advection') is True:
            self.vertical_advection()
        self.advect_oil()
