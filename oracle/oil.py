"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the oil physics inside OpenOil's vertical-mixing loop.

NumPy restatement (same operand dtypes and operation order as the reference evaluates them under NumPy 2) of
  PhysicsMethods.sea_water_density            opendrift/models/physics_methods.py:574-608
  seawater_dynamic_viscosity_sharqawy         opendrift/models/physics_methods.py:159-178
  wind_speed / significant_wave_height / wave_period / sea_surface_wave_breaking_fraction
                                              opendrift/models/physics_methods.py:885-966
  oil_wave_entrainment_rate_li2017            opendrift/models/physics_methods.py:115-137
  OpenOil.update_terminal_velocity            opendrift/models/openoil/openoil.py:922-998
  OpenOil.prepare_vertical_mixing             opendrift/models/openoil/openoil.py:1017-1031
  OpenOil.get_wave_breaking_droplet_diameter_johansen2015 / _liz2017   openoil.py:1072-1172
  OpenOil.surface_stick / surface_wave_mixing openoil.py:1033-1061
  the loop of OceanDrift.vertical_mixing      opendrift/models/oceandrift.py:505-565
Pinned by tests/golden/c9_openoil_mixing.npz, written by the reference's own OpenOil (oracle/gen_golden_oil.py).
Never imported by the product path.
"""
import numpy as np

G = 9.81
RHO_W_DEFAULT = None   # sea_water_density() with its defaults T=10., S=35. (Python floats), set below


def sea_water_density(T=10., S=35.):
    R4 = 4.8314E-04
    DR350 = 28.106331
    R1 = ((((6.536332E-09 * T - 1.120083E-06) * T + 1.001685E-04) * T - 9.095290E-03) * T + 6.793952E-02) * T - 28.263737
    R2 = (((5.3875E-09 * T - 8.2467E-07) * T + 7.6438E-05) * T - 4.0899E-03) * T + 8.24493E-01
    R3 = (-1.6546E-06 * T + 1.0227E-04) * T - 5.72466E-03
    SIG = R1 + (R4 * S + R3 * np.sqrt(S) + R2) * S
    return SIG + DR350 + 1000.


RHO_W_DEFAULT = sea_water_density()   # np.float64 (np.sqrt of a Python float): NOT a weak scalar, float32 arrays are promoted


def seawater_dynamic_viscosity(T, S):
    mu_w = (4.2844e-5 + 1.0 / (0.157 * (T + 64.993)**2 - 91.296))
    A = 1.541 + 1.998e-2 * T - 9.52e-5 * T**2
    B = 7.974 - 7.561e-2 * T + 4.724e-4 * T**2
    return mu_w * (1 + A * (S / 1000) + B * (S / 1000)**2)


def wind_speed(x_wind, y_wind):
    return np.sqrt(x_wind**2 + y_wind**2)


def significant_wave_height(x_wind, y_wind, hs=None):
    """physics_methods.py:893-907: the reader's Hs when any is > 0, else 0.0246 * wind_speed**2 (float32)."""
    if hs is not None and hs.max() > 0:
        return hs
    return 0.0246 * np.power(wind_speed(x_wind, y_wind), 2)


def wave_period(x_wind, y_wind, tp=None, in_environment=True):
    """physics_methods.py:918-943 without a period from readers: 2 pi / omega.  OpenOil has the wave period among
    its required variables, so calculate_missing_environment_variables (:876-883, called right after
    get_environment) has already stored that float64 result in the float32 environment recarray and every later
    call of the step reads the float32 value back (in_environment=True); a model without the variable
    (OceanDrift) computes the float64 value at every call."""
    if tp is not None and tp.max() > 0:
        T = tp.copy()
    else:
        ws = wind_speed(x_wind, y_wind)
        omega = 5 * np.ones(ws.shape)
        omega[ws > 0] = 0.877 * 9.81 / (1.17 * ws[ws > 0])
        T = (2 * np.pi) / omega
        if in_environment:
            T = T.astype(np.float32)
    if T.min() == 0:
        T[T == 0] = np.mean(T[T > 0])
    return T


def wave_breaking_fraction(x_wind, y_wind, tp=None):
    f = 0.032 * (wind_speed(x_wind, y_wind) - 5) / wave_period(x_wind, y_wind, tp)
    f[f < 0] = 0
    return f


def entrainment_rate_li2017(dynamic_viscosity, oil_density, interfacial_tension, hs, wbf, sea_water_density=RHO_W_DEFAULT):
    delta_rho = sea_water_density - oil_density
    d_o = 4 * np.sqrt(interfacial_tension / (delta_rho * G))
    we = sea_water_density * G * hs * d_o / interfacial_tension
    oh = dynamic_viscosity / np.sqrt(oil_density * interfacial_tension * d_o)
    with np.errstate(divide='ignore'):
        return (4.604e-10 * we**1.805 * oh**-1.023) * wbf


def entrainment_probability(density, viscosity, interfacial_tension, hs, wbf, dt_mix):
    rate = entrainment_rate_li2017(viscosity * density, density, interfacial_tension, hs, wbf)
    return 1 - np.exp(-rate * dt_mix)


def terminal_velocity(diameter, density, T_kelvin, S):
    """openoil.py:922-998 without T/S profiles.  diameter, T, S float32 arrays; density float64 after
    oil_weathering_noaa (:743-746)."""
    r = diameter
    T0 = T_kelvin - 273.15
    rho_water = sea_water_density(T=T0, S=S)
    my_w = seawater_dynamic_viscosity(T0, S)
    ny_w = my_w / rho_water
    rhopr = density / rho_water
    kw = 2 * G * (1 - rhopr) / (9 * ny_w)
    W = kw * (r / 2)**2
    Re = r * W / ny_w
    with np.errstate(invalid='ignore'):
        kw2 = (16 * G * (1 - rhopr) / 3)**0.5
    W2 = kw2 * (r / 2)**0.5
    hi = Re > 50
    W = np.array(W, dtype=np.float64)
    W[hi] = W2[hi]
    return W


def droplet_median_johansen2015(density, viscosity, film, hs, interfacial_tension):
    """-> dV_50, the mean over the elements (openoil.py:1133-1158)"""
    re = (density * film * (G * hs)**0.5) / (viscosity * density)
    we = (density * film * G * hs) / interfacial_tension
    A, Bp = 2.251, 0.027
    B = A * Bp
    dN_50 = (A * film * we**-0.6) + (B * film * re**-0.6)
    Sd = np.log(10) * 0.4
    return np.mean(np.exp(np.log(dN_50) + 3 * Sd**2))


def droplet_median_li2017(density, viscosity, hs, interfacial_tension):
    """openoil.py:1082-1101"""
    delta_rho = RHO_W_DEFAULT - density
    d_o = 4 * (interfacial_tension / (delta_rho * G))**0.5
    we = (RHO_W_DEFAULT * G * hs * d_o) / interfacial_tension
    oh = viscosity * density * (density * interfacial_tension * d_o)**-0.5
    dV_50 = d_o * 1.791 * (1 + 10 * oh)**0.460 * we**-0.518
    return np.mean(dV_50)


def droplet_spectrum_cdf(dV_50):
    """The 1e6-point lognormal spectrum between 1 micron and 3 mm (openoil.py:1081,1103-1108 / :1131,1159-1163) and
    the cumulative sum np.random.choice searches (numpy/random/mtrand.pyx choice: cdf = p.cumsum(); cdf /= cdf[-1])."""
    d = np.linspace(1e-6, 3e-3, 1000000)
    Sd = np.log(10) * 0.4
    spectrum = (np.exp(-(np.log(d) - np.log(dV_50))**2 / (2 * Sd**2))) / (d * Sd * np.sqrt(2 * np.pi))
    pdf = spectrum / np.sum(spectrum)
    cdf = pdf.cumsum()
    cdf /= cdf[-1]
    return d, cdf


def droplet_diameters(dV_50, uniforms):
    d, cdf = droplet_spectrum_cdf(dV_50)
    return d[cdf.searchsorted(uniforms, side='right')]


def vertical_mixing_oil(z, moving, diameter, density, T_kelvin, S, depth, ssh, mixing_z, Kprofiles, dt, dt_mix_cfg,
                        probability, diameter_if_entrained, mean_zb, u_mix, u_entrain, u_intrusion,
                        mix_at_surface=False, keep_droplet_diameter=False):
    """The loop of OceanDrift.vertical_mixing (oceandrift.py:505-565) as OpenOil runs it: terminal velocity of the
    droplets in every sub-step, slick formation, wave entrainment.  z and diameter are updated in place; returns the
    terminal velocities of the last sub-step.  u_intrusion[it, i] is the unit uniform of element i if it is entrained
    in sub-step it (the reference draws them compacted, np.random.uniform(0, mean(zb), entrained.sum()))."""
    dt_mix = dt_mix_cfg * np.sign(dt)
    ntimes = np.abs(int(dt / dt_mix))
    n = len(z)
    cols = np.arange(n)
    Zmin = -1. * (depth + ssh)
    gradK = -np.gradient(Kprofiles, mixing_z, axis=0)
    gradK[np.abs(gradK) < 1e-10] = 0
    nz = mixing_z.shape[0]
    from scipy.interpolate import interp1d
    z_index = (lambda d: np.zeros(len(d))) if nz == 1 else \
        interp1d(-mixing_z, range(nz), bounds_error=False, fill_value=(0, nz - 1))
    w = None
    for it in range(ntimes):
        surface = z == 0
        w = terminal_velocity(diameter, density, T_kelvin, S)
        zi = np.round(z_index(-z)).astype(np.uint16)                # oceandrift.py:485-488,513
        Kz = Kprofiles[zi, cols]
        dKdz = gradK[zi, cols]
        R = 2 * u_mix[it] - 1
        r = 1.0 / 3
        z[:] = z - moving * (dKdz * dt_mix - R * np.sqrt((Kz * np.abs(dt_mix) * 2 / r)))
        refl = z >= 0
        z[refl] = -z[refl]
        bottom = (z < Zmin) & (moving == 1)
        z[bottom] = (2 * Zmin[bottom] - z[bottom])
        z[:] = z + w * dt_mix * moving
        if not mix_at_surface:
            z[surface] = 0.
        z[z >= 0] = 0.                                               # surface_stick
        entrained = (z >= 0) & (u_entrain[it] < probability)         # surface_wave_mixing
        if entrained.sum() > 0:
            z[entrained] = -(0.0 + (float(mean_zb) - 0.0) * u_intrusion[it][entrained])
            if not keep_droplet_diameter:
                diameter[entrained] = diameter_if_entrained[entrained]
        below = z < Zmin
        z[below] = Zmin[below]                                       # lift_to_seafloor
    return w
