"""OpenOil on the device path: the ADVECTION of oil elements.

Mirrors the ordering and defaults of opendrift/models/openoil/openoil.py: `update()` =
(weathering) -> vertical mixing -> vertical advection -> `advect_oil()` (:1218-1239), the reverse of
OceanDrift.update; `advect_oil()` = advect_ocean_current(1-k_ice) + advect_wind(1-k_ice) +
stokes_drift(factor_stokes) (:1179-1216) with no sea ice; config defaults of :493-499.  Oil
weathering (evaporation, emulsification, dispersion, droplet spectra, the ADIOS oil database) is
chemistry outside the hot path: `oil_weathering()` is a host hook that does nothing here, and the
surface slick / wave-entrainment terms inside the mixing loop (`surface_wave_mixing`, :1033-1054)
are not applied.
"""
from .config import CONFIG_LEVEL_BASIC
from .oceandrift import OceanDrift


class OpenOil(OceanDrift):
    element_properties = dict(OceanDrift.element_properties, wind_drift_factor=0.03)   # openoil.py:133-140

    def __init__(self, *args, **kwargs):
        kwargs.pop('weathering_model', None)
        super().__init__(*args, **kwargs)
        self._add_config({
            'processes:evaporation': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'processes:emulsification': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_BASIC, 'description': ''},
            'processes:dispersion': {'type': 'bool', 'default': False, 'level': CONFIG_LEVEL_BASIC, 'description': ''},
        })
        self._set_config_default('drift:vertical_advection', False)
        self._set_config_default('drift:vertical_advection_at_surface', False)
        self._set_config_default('drift:vertical_mixing', True)
        self._set_config_default('drift:vertical_mixing_at_surface', False)
        self._set_config_default('drift:current_uncertainty', 0.05)
        self._set_config_default('drift:wind_uncertainty', 0.5)
        self._set_config_default('drift:max_speed', 1.3)

    def set_config(self, key, value):
        if key.startswith('processes:') and value is True:
            raise NotImplementedError('oil weathering is outside the advection hot path (DESIGN.md section 8)')
        super().set_config(key, value)

    def oil_weathering(self):
        pass

    def advect_oil(self):   # openoil.py:1179-1216, no sea ice: k_ice = 0, factor_stokes = 1
        self.advect_ocean_current(factor=1)
        self.advect_wind(factor=1)
        self.stokes_drift(1)

    def update(self):       # openoil.py:1218-1239
        self.oil_weathering()
        if self.get_config('drift:vertical_mixing') is True:
            self.update_terminal_velocity()
            self.vertical_mixing()
        if self.get_config('drift:vertical_advection') is True:
            self.vertical_advection()
        self.advect_oil()
