"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/*.npz from the REFERENCE ITSELF.

Runs the reference's own Python (OpenDrift v1.14.10 under /root/reference) through
oracle/refshim.py + oracle/refdriver.py on small seeded cases shaped like
BASELINE.json's configs and stores inputs + per-step live float64 lon/lat/z.
The fixtures travel to the GPU box; /root/reference does not.

    python oracle/gen_golden.py            # all scenarios
    python oracle/gen_golden.py c2 c4      # selected
"""
import os
import sys
from datetime import datetime, timedelta

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

assert refshim.install(), 'reference tree not found'
from oracle.refdriver import RefStepper, RecordingRandom  # noqa: E402
from opendrift_amd import synthetic as synth  # noqa: E402
from opendrift.models.oceandrift import OceanDrift  # noqa: E402
from opendrift.readers import reader_constant, reader_double_gyre  # noqa: E402
from opendrift.readers.basereader.structured import StructuredReader  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
T0 = datetime(2020, 1, 1)


class GridReader(StructuredReader):
    """In-memory StructuredReader (Reader plugin surface B2, cf. reader_constant_2d.py:20-49)
    with time levels and optional z levels; returns the whole domain as the block."""

    def __init__(self, proj4, x, y, times, arrays, z=None, name='synthetic_grid'):
        self.proj4 = proj4
        self.x, self.y = x, y
        self.xmin, self.xmax, self.ymin, self.ymax = x.min(), x.max(), y.min(), y.max()
        self.delta_x, self.delta_y = x[1] - x[0], y[1] - y[0]
        self.numx, self.numy = len(x), len(y)
        self.times = list(times)
        self.start_time, self.end_time = times[0], times[-1]
        self.time_step = times[1] - times[0] if len(times) > 1 else None
        self.z = z
        self.arrays = arrays  # {var: [nt, (nz,) ny, nx]}
        self.variables = list(arrays.keys())
        self.name = name
        super().__init__()

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        it = self.times.index(time)
        out = {'x': self.x, 'y': self.y, 'time': time,
               'z': self.z if self.z is not None else 0}
        for v in requested_variables:
            out[v] = np.array(self.arrays[v][it], copy=True)
        return out


def _run(o, dt, steps, record_random=False):
    st = RefStepper(o, dt, steps)
    N = st.n_total
    lon = np.empty((steps + 1, N))
    lat = np.empty((steps + 1, N))
    z = np.empty((steps + 1, N))
    status = np.empty((steps + 1, N), np.int32)
    sch = o.elements_scheduled
    lon[0], lat[0] = sch.lon, sch.lat
    z[0] = np.atleast_1d(sch.z) * np.ones(N)
    status[0] = 0
    draws = []
    for k in range(steps):
        if record_random:
            with RecordingRandom() as rr:
                st.step()
            draws.append(rr.draws)
        else:
            st.step()
        lon[k + 1], lat[k + 1], z[k + 1], status[k + 1] = st.state()
    return dict(lon=lon, lat=lat, z=z, status=status), draws


def _base(scheme):
    o = OceanDrift(loglevel=50)
    o.set_config('general:use_auto_landmask', False)
    o.set_config('drift:advection_scheme', scheme)
    return o


def c1_constant():
    """C1: OceanDrift + reader_constant, Euler, 24 h (BASELINE.json configs[0])."""
    o = _base('euler')
    o.add_reader(reader_constant.Reader({'x_sea_water_velocity': 0.3, 'y_sea_water_velocity': 0.2}))
    o.set_config('environment:constant:land_binary_mask', 0)
    np.random.seed(0)
    o.seed_elements(lon=4.0, lat=60.0, number=200, radius=5000, time=T0)
    res, _ = _run(o, 3600, 24)
    np.savez_compressed(os.path.join(GOLD, 'c1_constant_euler.npz'), u=0.3, v=0.2, dt=3600.0, **res)


def c2_double_gyre():
    """C2: analytic double gyre, Euler/RK2/RK4, dt=0.1 s (examples/example_double_gyre.py:20)."""
    rng = np.random.default_rng(0)
    N = 400
    for scheme, steps in (('euler', 20), ('runge-kutta', 20), ('runge-kutta4', 100)):
        o = _base(scheme)
        r = reader_double_gyre.Reader(initial_time=T0, epsilon=0.25, omega=0.628, A=0.1)
        o.add_reader(r)
        o.set_config('environment:fallback:land_binary_mask', 0)
        x = rng.uniform(0.05, 1.95, N)
        y = rng.uniform(0.05, 0.95, N)
        lon, lat = r.xy2lonlat(x, y)
        o.seed_elements(lon=lon, lat=lat, time=T0)
        res, _ = _run(o, 0.1, steps)
        np.savez_compressed(os.path.join(GOLD, 'c2_double_gyre_%s.npz' % scheme.replace('-', '')),
                            A=0.1, epsilon=0.25, omega=0.628, dt=0.1, **res)


def c3_grid3d():
    """C3-shaped: 3D z-level lon/lat grid (u,v,w,K + depth), RK4 + vertical mixing + vertical advection."""
    g = synth.grid3d(nx=48, ny=40, nz=8, nt=3, seed=1)
    times = [T0 + timedelta(seconds=float(t)) for t in g['t']]
    arrays = {k: g[k] for k in ('x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity',
                                'ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level',
                                'land_binary_mask')}
    o = _base('runge-kutta4')
    o.add_reader(GridReader('+proj=latlong', g['x'], g['y'], times, arrays, z=g['z']))
    o.set_config('drift:vertical_mixing', True)
    o.set_config('vertical_mixing:timestep', 60)
    o.set_config('vertical_mixing:diffusivitymodel', 'environment')
    o.set_config('drift:vertical_advection', True)
    o.set_config('general:coastline_action', 'previous')
    rng = np.random.default_rng(2)
    N = 300
    lon = rng.uniform(g['x'][4], g['x'][-5], N)
    lat = rng.uniform(g['y'][4], g['y'][-5], N)
    zz = rng.uniform(-40, -1, N)
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, z=zz, time=T0)
    res, draws = _run(o, 600, 8, record_random=True)
    uni = np.array([np.stack([d[1] for d in step if d[0] == 'random']) for step in draws])
    np.savez_compressed(os.path.join(GOLD, 'c3_grid3d_rk4_vmix.npz'), dt=600.0, dt_mix=60.0,
                        uniforms=uni, **{('g_' + k): v for k, v in g.items()}, **res)


def c4_stere():
    """C4-shaped: NorKyst-like polar-stereographic 2D grid (current, wind, Stokes, landmask with NaN
    current on land), RK4 + horizontal diffusion + stranding."""
    g = synth.grid_stere(nx=70, ny=50, nt=3, seed=3)
    times = [T0 + timedelta(seconds=float(t)) for t in g['t']]
    arrays = {k: g[k] for k in ('x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind',
                                'sea_surface_wave_stokes_drift_x_velocity',
                                'sea_surface_wave_stokes_drift_y_velocity', 'land_binary_mask')}
    o = _base('runge-kutta4')
    r = GridReader(synth.NORKYST_PROJ4, g['x'], g['y'], times, arrays)
    o.add_reader(r)
    o.set_config('environment:constant:horizontal_diffusivity', 10)
    o.set_config('general:coastline_action', 'stranding')
    o.set_config('general:coastline_approximation_precision', None)
    o.set_config('drift:stokes_drift', True)
    rng = np.random.default_rng(4)
    N = 400
    x = rng.uniform(g['x'][5], g['x'][-6], N)
    y = rng.uniform(g['y'][5], g['y'][-6], N)
    lon, lat = r.xy2lonlat(x, y)
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, time=T0, wind_drift_factor=0.03)
    res, draws = _run(o, 900, 10, record_random=True)
    nrm = [[d[1] for d in step if d[0] == 'normal'] for step in draws]
    nmax = max(len(a) for s in nrm for a in s)
    normals = np.full((len(nrm), 2, nmax), np.nan)
    for k, s in enumerate(nrm):
        for j, a in enumerate(s[:2]):
            normals[k, j, :len(a)] = a
    np.savez_compressed(os.path.join(GOLD, 'c4_stere_rk4_hdiff_strand.npz'), dt=900.0, normals=normals,
                        wdf=0.03, **{('g_' + k): v for k, v in g.items()}, **res)


def c5_leeway(capsizing=False):
    """C5-shaped: Leeway (object class 1, PIW-1) on the C4 stere grid with per-element wind / current
    uncertainty; Euler by construction (leeway.py:472-476); jibing.  capsizing=True: processes:capsizing with a
    wind threshold inside the wind range of the block (leeway.py:438-455) -> c5b_leeway_capsizing.npz."""
    from opendrift.models.leeway import Leeway
    g = synth.grid_stere(nx=70, ny=50, nt=3, seed=5)
    times = [T0 + timedelta(seconds=float(t)) for t in g['t']]
    arrays = {k: g[k] for k in ('x_sea_water_velocity', 'y_sea_water_velocity', 'x_wind', 'y_wind', 'land_binary_mask')}
    o = Leeway(loglevel=50)
    o.set_config('general:use_auto_landmask', False)
    r = GridReader(synth.NORKYST_PROJ4, g['x'], g['y'], times, arrays)
    o.add_reader(r)
    o.set_config('drift:wind_uncertainty', 2.0)
    o.set_config('drift:current_uncertainty', 0.1)
    o.set_config('general:coastline_action', 'stranding')
    o.set_config('seed:jibe_probability', 0.5)      # make jibing visible within a few steps
    if capsizing:
        o.set_config('processes:capsizing', True)
        o.set_config('capsizing:wind_threshold', 8.0)
        o.set_config('capsizing:wind_threshold_sigma', 2.0)
        o.set_config('capsizing:leeway_fraction', 0.4)
    rng = np.random.default_rng(6)
    N = 300
    x = rng.uniform(g['x'][5], g['x'][int(0.8 * len(g['x']))], N)
    y = rng.uniform(g['y'][5], g['y'][-6], N)
    lon, lat = r.xy2lonlat(x, y)
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, time=T0, object_type=1)
    sch = o.elements_scheduled
    props = {k: np.array(getattr(sch, k) * np.ones(N), dtype=np.float32) for k in (
        'downwind_slope', 'crosswind_slope', 'downwind_offset', 'crosswind_offset', 'downwind_eps', 'crosswind_eps',
        'jibe_probability', 'orientation', 'capsized')}
    res, draws = _run(o, 600, 8, record_random=True)
    nrm = [[d[1] for d in step if d[0] == 'normal'] for step in draws]
    uni = [[d[1] for d in step if d[0] == 'random'] for step in draws]
    normals = np.full((len(nrm), 4, N), np.nan)
    uniforms = np.full((len(uni), N), np.nan)
    for k in range(len(nrm)):
        for j, a in enumerate(nrm[k][:4]):
            normals[k, j, :len(a)] = a
        uniforms[k, :len(uni[k][0])] = uni[k][0]
    extra = {}
    if capsizing:   # np.random.rand(len(can_be_capsized)) per step, drawn before the jibing random()
        cap = [[d[1] for d in step if d[0] == 'rand'] for step in draws]
        cu = np.full((len(cap), N), np.nan)
        for k in range(len(cap)):
            if cap[k]:
                cu[k, :len(cap[k][0])] = cap[k][0]
        extra = dict(cap_uniforms=cu, wind_threshold=8.0, wind_threshold_sigma=2.0, capsized_final=np.array(o.elements.capsized),
                     ID_final=np.array(o.elements.ID))
    np.savez_compressed(os.path.join(GOLD, 'c5b_leeway_capsizing.npz' if capsizing else 'c5_leeway_stere.npz'), dt=600.0,
                        normals=normals, uniforms=uniforms, **extra,
                        **{('p_' + k): v for k, v in props.items()}, **{('g_' + k): v for k, v in g.items()}, **res)



class CurvilinearReader(StructuredReader):
    """StructuredReader WITHOUT a projection: 2D lon/lat node arrays, pixel coordinates as x/y (the
    'fakeproj' branch, basereader/structured.py:44-113; reader_ROMS_native.py is the stock example)."""

    def __init__(self, lon2d, lat2d, times, arrays, name='synthetic_curvilinear'):
        self.proj4 = None
        self.lon, self.lat = lon2d, lat2d
        self.times = list(times)
        self.start_time, self.end_time = times[0], times[-1]
        self.time_step = times[1] - times[0] if len(times) > 1 else None
        self.z = None
        self.arrays = arrays
        self.variables = list(arrays.keys())
        self.name = name
        super().__init__()

    def get_variables(self, requested_variables, time=None, x=None, y=None, z=None):
        it = self.times.index(time)
        out = {'x': self.x, 'y': self.y, 'time': time, 'z': 0}
        for v in requested_variables:
            out[v] = np.array(self.arrays[v][it], copy=True)
        return out


def c6_curvilinear():
    """Curvilinear lon/lat grid (the lon/lat image of a polar-stereographic mesh, i.e. what a ROMS-native /
    NorKyst file holds as 2D lon,lat) read WITHOUT a projection: positions go through the Delaunay
    LinearNDInterpolator lon,lat -> pixel lookup (structured.py:74-113,438-472); RK4 + stranding."""
    from opendrift_amd.projection import stere_polar_inverse
    g = synth.grid_stere(nx=60, ny=44, nt=3, seed=6, dx=4000.0)
    X, Y = np.meshgrid(g['x'].astype(np.float64), g['y'].astype(np.float64))
    lon2d, lat2d = stere_polar_inverse(X, Y, **synth.NORKYST_PROJ)
    times = [T0 + timedelta(seconds=float(t)) for t in g['t']]
    arrays = {k: g[k] for k in ('x_sea_water_velocity', 'y_sea_water_velocity', 'land_binary_mask')}
    o = _base('runge-kutta4')
    r = CurvilinearReader(lon2d, lat2d, times, arrays)
    o.add_reader(r)
    o.set_config('general:coastline_action', 'stranding')
    o.set_config('general:coastline_approximation_precision', None)
    o.set_config('drift:stokes_drift', False)
    rng = np.random.default_rng(6)
    N = 400
    x = rng.uniform(4, 54, N)
    y = rng.uniform(4, 39, N)
    lon, lat = r.xy2lonlat(x.copy(), y.copy())
    # the lookup on its own, at the seed positions and at a cloud that also leaves the mesh
    qlon = np.concatenate([lon, rng.uniform(lon2d.min() - 0.2, lon2d.max() + 0.2, 600)])
    qlat = np.concatenate([lat, rng.uniform(lat2d.min() - 0.1, lat2d.max() + 0.1, 600)])
    qx, qy = r.lonlat2xy(qlon, qlat)
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, time=T0, wind_drift_factor=0.0)
    res, _ = _run(o, 900, 8)
    np.savez_compressed(os.path.join(GOLD, 'c6_curvilinear_rk4.npz'), dt=900.0, lon2d=lon2d, lat2d=lat2d,
                        qlon=qlon, qlat=qlat, qx=qx, qy=qy,
                        **{('g_' + k): g[k] for k in ('t',) + tuple(arrays)}, **res)



def c7_diffusivity():
    """Wind-parameterised diffusivity profiles in vertical_mixing (oceandrift.py:425-458, physics_methods.py:203-250):
    (a) the default 'environment' model with NO reader for ocean_vertical_diffusivity -> Large et al. (1994),
    (b) 'windspeed_Sundby1983' with a background diffusivity; wind, mixed layer depth and sea floor from a 2D lon/lat
    reader (per-element wind speed and MLD, MLD.max() sets the 1 m mixing levels), Euler current, np.random draws
    recorded.  Plus the two functions on the grid of the reference's test_vertical_diffusivity
    (tests/models/test_physics.py:50-60) and on random float32 inputs."""
    from opendrift.models.physics_methods import verticaldiffusivity_Large1994, verticaldiffusivity_Sundby1983
    rng = np.random.default_rng(7)
    nx, ny, nt = 40, 32, 3
    x = np.linspace(2, 8, nx).astype(np.float32)
    y = np.linspace(59, 63, ny).astype(np.float32)
    X, Y = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    t = np.arange(nt) * 3600.0
    times = [T0 + timedelta(seconds=float(v)) for v in t]
    g = dict(x=x, y=y, t=t)
    g['x_wind'] = np.stack([(9 + 6 * np.sin(3 * X + k)) for k in range(nt)]).astype(np.float32)
    g['y_wind'] = np.stack([(4 * np.cos(4 * Y - k)) for k in range(nt)]).astype(np.float32)
    g['ocean_mixed_layer_thickness'] = np.stack([(30 + 22.5 * (1 + np.sin(2 * X + 3 * Y)))] * nt).astype(np.float32)
    g['sea_floor_depth_below_sea_level'] = np.stack([(25 + 150 * X)] * nt).astype(np.float32)
    g['x_sea_water_velocity'] = np.stack([0.2 * np.cos(3 * Y + k) for k in range(nt)]).astype(np.float32)
    g['y_sea_water_velocity'] = np.stack([0.2 * np.sin(3 * X - k) for k in range(nt)]).astype(np.float32)
    names = [k for k in g if k not in ('x', 'y', 't')]
    N = 300
    lon = rng.uniform(x[3], x[-4], N)
    lat = rng.uniform(y[3], y[-4], N)
    zz = -rng.uniform(0, 60, N)
    zz[:40] = 0.0
    tv = np.where(np.arange(N) % 3 == 0, 0.002, np.where(np.arange(N) % 3 == 1, -0.001, 0.0))
    out = {}
    for tag, model, bg in (('large', 'environment', 0.0), ('sundby', 'windspeed_Sundby1983', 2e-4)):
        o = _base('euler')
        o.add_reader(GridReader('+proj=latlong', x, y, times, {k: g[k] for k in names}))
        o.set_config('environment:fallback:land_binary_mask', 0)
        o.set_config('drift:vertical_mixing', True)
        o.set_config('vertical_mixing:timestep', 60)
        o.set_config('vertical_mixing:diffusivitymodel', model)
        o.set_config('vertical_mixing:background_diffusivity', bg)
        o.set_config('drift:stokes_drift', False)
        np.random.seed(0)
        o.seed_elements(lon=lon, lat=lat, z=zz, time=T0, terminal_velocity=tv, wind_drift_factor=0.0)
        res, draws = _run(o, 900, 6, record_random=True)
        uni = np.array([np.stack([d[1] for d in step if d[0] == 'random']) for step in draws])
        out.update({tag + '_' + k: v for k, v in res.items()})
        out[tag + '_uniforms'] = uni
        out[tag + '_bg'] = bg
    # the functions themselves
    wind, depth = np.meshgrid(np.arange(0, 20, 5), np.arange(0, 80, 5))
    out['kat_large'] = verticaldiffusivity_Large1994(wind, depth)
    out['kat_sundby'] = verticaldiffusivity_Sundby1983(wind, depth)
    w32 = rng.uniform(0, 25, 64).astype(np.float32)
    mld32 = rng.uniform(8, 90, 64).astype(np.float32)
    depths = np.abs(-np.arange(0, mld32.max() + 2))
    W, D = np.meshgrid(w32, depths)
    out['fn_wind'], out['fn_mld'], out['fn_depths'] = w32, mld32, depths
    out['fn_large'] = verticaldiffusivity_Large1994(W, D, mld32, 1e-4)
    out['fn_sundby'] = verticaldiffusivity_Sundby1983(W, D, mld32, 1e-4)
    np.savez_compressed(os.path.join(GOLD, 'c7_wind_diffusivity.npz'), dt=900.0, dt_mix=60.0, tv=tv,
                        **{('g_' + k): v for k, v in g.items()}, **out)



def c8_seafloor():
    """general:seafloor_action 'deactivate' and 'previous' (basemodel/__init__.py:748-783): elements at depth carried
    by an Euler current towards shoaling water (2D lon/lat reader with the sea floor depth), no mixing."""
    nx, ny, nt = 40, 32, 3
    x = np.linspace(2, 8, nx).astype(np.float32)
    y = np.linspace(59, 63, ny).astype(np.float32)
    X, Y = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    t = np.arange(nt) * 3600.0
    times = [T0 + timedelta(seconds=float(v)) for v in t]
    g = dict(x=x, y=y, t=t)
    g['sea_floor_depth_below_sea_level'] = np.stack([(20 + 300 * X + 40 * np.sin(3 * Y))] * nt).astype(np.float32)
    g['x_sea_water_velocity'] = np.stack([-1.2 - 0.3 * np.cos(3 * Y + k) for k in range(nt)]).astype(np.float32)
    g['y_sea_water_velocity'] = np.stack([0.3 * np.sin(3 * X - k) for k in range(nt)]).astype(np.float32)
    names = [k for k in g if k not in ('x', 'y', 't')]
    rng = np.random.default_rng(8)
    N = 200
    lon = rng.uniform(x[8], x[-6], N)
    lat = rng.uniform(y[3], y[-4], N)
    zz = -rng.uniform(5, 250, N)
    out = {}
    for action in ('deactivate', 'previous'):
        o = _base('euler')
        o.add_reader(GridReader('+proj=latlong', x, y, times, {k: g[k] for k in names}))
        o.set_config('environment:fallback:land_binary_mask', 0)
        o.set_config('general:seafloor_action', action)
        o.set_config('drift:vertical_mixing', False)
        o.set_config('drift:vertical_advection', False)
        o.set_config('drift:stokes_drift', False)
        np.random.seed(0)
        o.seed_elements(lon=lon, lat=lat, z=zz, time=T0, wind_drift_factor=0.0)
        res, _ = _run(o, 900, 8)
        out.update({action + '_' + k: v for k, v in res.items()})
        out[action + '_categories'] = np.array(o.status_categories)
    # 'deactivate' reached INSIDE the vertical-mixing loop (oceandrift.py:555-559): sinking elements, Sundby profile
    tv = np.where(np.arange(N) % 2 == 0, -0.02, -0.004)
    o = _base('euler')
    o.add_reader(GridReader('+proj=latlong', x, y, times, {k: g[k] for k in names}))
    o.add_reader(reader_constant.Reader({'x_wind': 9.0, 'y_wind': -3.0}))
    o.set_config('environment:fallback:land_binary_mask', 0)
    o.set_config('general:seafloor_action', 'deactivate')
    o.set_config('drift:vertical_mixing', True)
    o.set_config('vertical_mixing:timestep', 60)
    o.set_config('vertical_mixing:diffusivitymodel', 'windspeed_Sundby1983')
    o.set_config('drift:vertical_advection', False)
    o.set_config('drift:stokes_drift', False)
    np.random.seed(0)
    o.seed_elements(lon=lon, lat=lat, z=zz * 0.5, time=T0, wind_drift_factor=0.0, terminal_velocity=tv)
    res, draws = _run(o, 900, 6, record_random=True)
    out.update({'deactmix_' + k: v for k, v in res.items()})
    out['deactmix_categories'] = np.array(o.status_categories)
    out['deactmix_tv'] = tv
    mx = max(len(d[1]) for step in draws for d in step if d[0] == 'random')
    uni = np.full((len(draws), 15, mx), np.nan)
    for k, step in enumerate(draws):
        for j, d in enumerate([d for d in step if d[0] == 'random']):
            uni[k, j, :len(d[1])] = d[1]
    out['deactmix_uniforms'] = uni
    np.savez_compressed(os.path.join(GOLD, 'c8_seafloor_actions.npz'), dt=900.0,
                        **{('g_' + k): v for k, v in g.items()}, **out)


def c11_mixing_profiles():
    """The reference's own known-answer test of an isolated mixing time step (tests/models/test_run.py:359-410,
    test_vertical_mixing_profiles): hand-made diffusivity profiles on z = 0, -2, ..., -28, six cases (no mixing,
    sinking, mixing, mixing + rising, mixing + sinking, mixed layer), 100 elements from z = -10, 2 h in 120
    sub-steps, mixing at the surface allowed.  The body of the test is executed verbatim on the reference's
    OceanDrift; stored: the np.random draws and the final depths (the test's own assertions -- min / max / mean
    to one decimal -- are checked here too)."""
    cases = [{'vt': 0, 'K': 0, 'K_below': .01, 'T': 60, 'zmin': -10, 'zmax': -10, 'zmean': -10},
             {'vt': -.005, 'K': 0, 'K_below': .01, 'T': 60, 'zmin': -74.79, 'zmax': -21.6, 'zmean': -49.97},
             {'vt': 0, 'K': .01, 'K_below': .01, 'T': 60, 'zmin': -42.76, 'zmax': -0.02, 'zmean': -14.38},
             {'vt': .005, 'K': .01, 'K_below': .01, 'T': 60, 'zmin': -7.85, 'zmax': -0.01, 'zmean': -2.1},
             {'vt': -0.005, 'K': .01, 'K_below': .01, 'T': 60, 'zmin': -78.76, 'zmax': -19.74, 'zmean': -48.0},
             {'vt': 0, 'K': .02, 'K_below': .001, 'T': 60, 'zmin': -21.3, 'zmax': -0.1, 'zmean': -9.55}]
    N = 100
    z = np.arange(0, -30, -2)
    time = T0
    out = dict(z_levels=z, cases=np.array([[c['vt'], c['K'], c['K_below'], c['T'], c['zmin'], c['zmax'], c['zmean']] for c in cases]))
    for ic, case in enumerate(cases):
        diffusivity = np.ones(z.shape) * case['K']
        diffusivity[z < -15] = case['K_below']
        o = OceanDrift(loglevel=50)
        o.set_config('drift:vertical_mixing', True)
        o.set_config('drift:vertical_mixing_at_surface', True)
        o.set_config('drift:vertical_advection_at_surface', True)
        o.set_config('vertical_mixing:diffusivitymodel', 'environment')
        o.set_config('vertical_mixing:timestep', case['T'])
        o.set_config('environment:fallback:land_binary_mask', 0)
        o.seed_elements(lon=4, lat=60, z=-10, time=time, number=N, terminal_velocity=case['vt'])
        o.time = time
        o.time_step = timedelta(hours=2)
        o.release_elements()
        o.environment = np.array([(100, 0) for _ in range(N)],
                                 dtype=[('sea_floor_depth_below_sea_level', np.float32),
                                        ('sea_surface_height', np.float32)]).view(np.recarray)
        o.environment.ocean_mixed_layer_thickness = np.ones(N) * 50
        o.environment_profiles = {'z': z, 'ocean_vertical_diffusivity': np.tile(diffusivity, (N, 1)).T}
        o.env.finalize()
        with RecordingRandom() as rr:
            o.vertical_mixing()
        zz = np.array(o.elements.z, dtype=np.float64)
        assert abs(zz.min() - case['zmin']) < 0.05 and abs(zz.max() - case['zmax']) < 0.05 and abs(zz.mean() - case['zmean']) < 0.05, \
            (ic, zz.min(), zz.max(), zz.mean())
        out['z_final_%d' % ic] = zz
        out['uniforms_%d' % ic] = np.stack([d[1] for d in rr.draws if d[0] == 'random'])
        out['K_%d' % ic] = diffusivity
    np.savez_compressed(os.path.join(GOLD, 'c11_mixing_profiles.npz'), **out)


SCEN = dict(c11=c11_mixing_profiles, c8=c8_seafloor, c7=c7_diffusivity, c6=c6_curvilinear, c1=c1_constant, c2=c2_double_gyre, c3=c3_grid3d, c4=c4_stere, c5=c5_leeway,
            c5b=lambda: c5_leeway(capsizing=True))

if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    for name in (sys.argv[1:] or list(SCEN)):
        print('generating', name)
        SCEN[name]()
