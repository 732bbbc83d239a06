"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the wind-parameterised vertical diffusivity profiles.

Restates (NumPy, same operand dtypes and operation order)
  verticaldiffusivity_Sundby1983   opendrift/models/physics_methods.py:203-216
  verticaldiffusivity_Large1994    opendrift/models/physics_methods.py:218-250
  OceanDrift.get_diffusivity_profile / the level construction of vertical_mixing
                                   opendrift/models/oceandrift.py:385-395, 425-458
Pinned by tests/golden/c7_wind_diffusivity.npz: outputs of the reference's own functions on the grid of its
test_vertical_diffusivity (tests/models/test_physics.py:50-60: maxima 0.2017 / 0.0585) and on random float32 wind /
mixed-layer inputs, and two reference runs whose vertical_mixing uses these profiles.
"""
import numpy as np


def sundby1983(windspeed, depth, mixedlayerdepth=50, background_diffusivity=0):
    K = 76.1e-4 + 2.26e-4 * windspeed * windspeed * np.ones(np.atleast_1d(depth.shape))
    K[depth > mixedlayerdepth - 1] = (K[depth > mixedlayerdepth - 1] + background_diffusivity) / 2
    K[depth >= mixedlayerdepth] = background_diffusivity
    return K


def large1994(windspeed, depth, mixedlayerdepth=50, background_diffusivity=0):
    depth = np.abs(depth)
    sigma = depth / mixedlayerdepth
    G = 1. * sigma + -2 * sigma**2 + 1 * sigma**3       # vertical shape function
    G[G >= 1] = G[G >= 1] * 0.
    windstress = windspeed * windspeed * 1.25e-3 * 1.22  # cd (Kara et al. 2007), air density
    K = mixedlayerdepth * 0.2 * 0.4 * G * windstress + sigma * background_diffusivity
    K[depth >= mixedlayerdepth] = background_diffusivity
    return K


def profiles(model, x_wind, y_wind, mld, background_diffusivity):
    """-> (mixing_z [nz], Kprofiles [nz, N] float64) as vertical_mixing builds them for an analytical model:
    1 m levels from the surface to max(MLD) + 1 (oceandrift.py:430), wind speed from the float32 environment
    (physics_methods.py:885-887)."""
    x_wind, y_wind, mld = (np.asarray(a, dtype=np.float32) for a in (x_wind, y_wind, mld))
    mixing_z = -np.arange(0, mld.max() + 2)
    wind, depth = np.meshgrid(np.sqrt(x_wind**2 + y_wind**2), np.abs(mixing_z))
    fn = {'windspeed_Large1994': large1994, 'windspeed_Sundby1983': sundby1983}[model]
    return mixing_z, fn(wind, depth, mld, background_diffusivity)
