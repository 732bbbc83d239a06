"""ctypes binding of libodrift_hip.so (include/odrift.h).  This is the stub a maintainer of the
reference would add: plain pointers and sizes, no torch types.

The product path fails loudly when the HIP library is missing -- there is no CPU fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ODR_LIB') or os.path.join(HERE, 'libodrift_hip.so')   # ODR_LIB: A/B builds (tools/gpu_ab.sh)

NVAR = 26
VARIABLES = {
    'x_sea_water_velocity': 0, 'y_sea_water_velocity': 1, 'x_wind': 2, 'y_wind': 3,
    'upward_sea_water_velocity': 4, 'ocean_vertical_diffusivity': 5,
    'sea_surface_wave_stokes_drift_x_velocity': 6, 'sea_surface_wave_stokes_drift_y_velocity': 7,
    'land_binary_mask': 8, 'sea_floor_depth_below_sea_level': 9, 'sea_surface_height': 10,
    'horizontal_diffusivity': 11, 'sea_surface_wave_significant_height': 12,
    'sea_surface_wave_period_at_variance_spectral_density_maximum': 13,
    'ocean_mixed_layer_thickness': 14,
    'sea_water_temperature': 15, 'sea_water_salinity': 16,     # OpenOil.required_variables (openoil.py:271-278)
    'sea_ice_area_fraction': 17, 'sea_ice_x_velocity': 18, 'sea_ice_y_velocity': 19,      # advect_oil in ice (openoil.py:1179-1216)
    # the windsea_swell Stokes profile (physics_methods.py:418-456)
    'sea_surface_swell_wave_to_direction': 20,
    'sea_surface_swell_wave_peak_period_from_variance_spectral_density': 21,
    'sea_surface_swell_wave_significant_height': 22, 'sea_surface_wind_wave_to_direction': 23,
    'sea_surface_wind_wave_mean_period': 24, 'sea_surface_wind_wave_significant_height': 25,
}
VARIABLE_NAMES = {v: k for k, v in VARIABLES.items()}
PROJ_LATLONG, PROJ_STERE_EQUIT_SPHERE, PROJ_STERE_POLAR = 0, 1, 2
PROJ_MERC, PROJ_LCC = 4, 5
PROJ_TMERC, PROJ_LAEA, PROJ_STERE_OBLIQUE, PROJ_OB_TRAN = 6, 7, 8, 9
SCHEME = {'euler': 0, 'runge-kutta': 1, 'runge-kutta4': 2}
RNG_DEVICE, RNG_HOST = 0, 1
STAGE_MATH = {'exact': 0, 'fast': 1}     # odr_ctx_set_stage_math
COAST = {'none': 0, 'stranding': 1, 'previous': 2}
# odr_history variable codes of the element properties (include/odrift.h ODR_HIST_*); environment
# variables use their ODR_VAR_* id
HIST = {'lon': 1000, 'lat': 1001, 'z': 1002, 'status': 1003, 'moving': 1004, 'age_seconds': 1005,
        'wind_drift_factor': 1006, 'current_drift_factor': 1007, 'terminal_velocity': 1008}
HIST_PROPERTY0 = 2000
COMM_ID_BYTES = 256      # ODR_COMM_ID_BYTES


class StepExtras(C.Structure):   # odr_step_extras
    _fields_ = [('seafloor_action', C.c_int32), ('retired_code', C.c_int32), ('age_dt', C.c_double),
                ('max_age_seconds', C.c_double), ('missing_code', C.c_int32), ('main_noise', C.c_int32),
                ('vmix', C.c_int32), ('vmix_at_surface', C.c_int32), ('vmix_vadv', C.c_int32), ('pad', C.c_int32),
                ('vmix_dt_mix', C.c_double), ('vmix_step', C.c_uint64)]
ANALYTIC_DOUBLE_GYRE, ANALYTIC_OSCILLATING = 1, 2


class ProjDesc(C.Structure):
    _fields_ = [('kind', C.c_int32), ('a', C.c_double), ('es', C.c_double), ('lat0_deg', C.c_double),
                ('lon0_deg', C.c_double), ('lat_ts_deg', C.c_double), ('k0', C.c_double),
                ('x0', C.c_double), ('y0', C.c_double), ('lat1_deg', C.c_double), ('lat2_deg', C.c_double)]


class OdrError(RuntimeError):
    code = 0          # the ODR_ERR_* value (include/odrift.h): -3 = a capacity of the device library is exhausted


_P = C.POINTER
_dp, _fp, _ip, _i64p = _P(C.c_double), _P(C.c_float), _P(C.c_int32), _P(C.c_int64)
_vp = C.c_void_p

_SIGNATURES = {
    'odr_ctx_create': [C.c_int, C.c_uint64, _P(_vp)],
    'odr_ctx_destroy': [_vp],
    'odr_sync': [_vp],
    'odr_set_stream': [_vp, _vp],
    'odr_particles_create': [_vp, C.c_int64, _P(_vp)],
    'odr_particles_destroy': [_vp, _vp],
    'odr_particles_append': [_vp, _vp, C.c_int64, _dp, _dp, _dp, _ip, _ip, _fp, _fp, _fp],
    'odr_particles_count': [_vp, _vp, _i64p, _i64p],
    'odr_particles_download': [_vp, _vp, _dp, _dp, _dp, _ip, _ip, _ip],
    'odr_particles_download_deactivated': [_vp, _vp, _dp, _dp, _dp, _ip, _ip],
    'odr_particles_upload': [_vp, _vp, _dp, _dp, _dp, _ip, _fp, _fp, _fp],
    'odr_particles_device_ptr': [_vp, _vp, C.c_char_p, _P(_vp)],
    'odr_particles_download_f32': [_vp, _vp, C.c_char_p, _fp],
    'odr_source_constant': [_vp, C.c_int, _ip, _dp, _ip],
    'odr_source_analytic': [_vp, C.c_int, _dp, C.c_int, _ip],
    'odr_source_grid': [_vp, _P(ProjDesc), _dp, C.c_int, C.c_int, C.c_int, _dp, _ip],
    'odr_source_grid_curvilinear': [_vp, _dp, _dp, C.c_int, C.c_int, _dp, C.c_int, C.c_int, _dp, _ip],
    'odr_source_landmask': [_vp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, _P(C.c_uint8), _ip],
    'odr_source_lonlat2xy': [_vp, C.c_int32, C.c_int64, _dp, _dp, _dp, _dp],
    'odr_block_upload': [_vp, C.c_int32, C.c_int32, C.c_double, C.c_int, _ip, _P(_fp), _ip, C.c_int,
                         C.c_int, _dp],
    'odr_block_upload_device': [_vp, C.c_int32, C.c_int32, C.c_double, C.c_int, _ip, _P(_vp), _ip,
                                C.c_int, C.c_int, _dp],
    'odr_block_upload_async': [_vp, C.c_int32, C.c_int32, C.c_double, C.c_int, _ip, _P(_vp), _ip, C.c_int, C.c_int, _dp],
    'odr_block_commit': [_vp, C.c_int32, C.c_int32],
    'odr_host_register': [_vp, _vp, C.c_uint64],
    'odr_host_unregister': [_vp, _vp],
    'odr_block_drop': [_vp, C.c_int32, C.c_int32],
    'odr_env_bind': [_vp, C.c_int32, C.c_int, _ip, C.c_float],
    'odr_env_sample': [_vp, _vp, C.c_int, _ip, C.c_double, _P(_fp)],
    'odr_env_download': [_vp, _vp, C.c_int32, _fp],
    'odr_env_upload': [_vp, _vp, C.c_int32, _fp],
    'odr_env_add_noise': [_vp, _vp, C.c_int32, C.c_int32, C.c_double, C.c_int, C.c_int, _dp, _dp, C.c_uint64],
    'odr_advect_set_noise': [_vp, _vp, C.c_double, C.c_double, C.c_int, _dp, _dp, C.c_int, C.c_uint64],
    'odr_ctx_set_stage_math': [_vp, C.c_int],
    'odr_advect': [_vp, _vp, C.c_int, C.c_double, C.c_double, C.c_double],
    'odr_env_coast_advect': [_vp, _vp, C.c_int, _ip, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_double, C.c_double, _vp, _P(C.c_int64)],
    'odr_update_positions': [_vp, _vp, _dp, _dp, C.c_int, C.c_double],
    'odr_source_set_members': [_vp, C.c_int32, C.c_int32, C.c_int32],
    'odr_particles_set_rank_offset': [_vp, _vp, C.c_int64],
    'odr_block_set_content_ids': [_vp, C.c_int32, C.c_int32, C.c_int, _P(C.c_int32), _P(C.c_uint64)],
    'odr_set_element_factor': [_vp, _vp, C.c_int],
    'odr_advect_sea_ice': [_vp, _vp, C.c_double, C.c_double],
    'odr_advect_wind': [_vp, _vp, C.c_double, C.c_double, C.c_int, C.c_double],
    'odr_stokes_drift': [_vp, _vp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double],
    'odr_particles_set_property': [_vp, _vp, C.c_int, C.c_int64, C.c_int64, _fp],
    'odr_particles_get_property': [_vp, _vp, C.c_int, _fp],
    'odr_particles_snapshot_property': [_vp, _vp, C.c_int],
    'odr_env_coast_leeway': [_vp, _vp, C.c_int, _ip, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                             C.c_double, C.c_double, C.c_uint64, _i64p],
    'odr_leeway_set_missing_code': [_vp, C.c_int32],
    'odr_leeway_capsize': [_vp, _vp, C.c_double, C.c_double, C.c_double, C.c_int, _dp, C.c_uint64],
    'odr_leeway': [_vp, _vp, C.c_double, C.c_double, C.c_int, _dp, C.c_uint64],
    'odr_hdiffusion': [_vp, _vp, C.c_double, C.c_int, _dp, _dp, C.c_uint64],
    'odr_movers': [_vp, _vp, C.c_double, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _dp, _dp,
                   C.c_uint64],
    'odr_vmix': [_vp, _vp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, _dp, C.c_uint64],
    'odr_vmix_wind_profile': [_vp, _vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, _dp, C.c_uint64],
    'odr_vmix_fuse_vertical_advection': [_vp, C.c_int],
    'odr_oil_prepare_mixing': [_vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_int, _dp, _dp, _dp, C.c_uint64],
    'odr_oil_mixing_stats': [_vp, _dp, _dp],
    'odr_oil_local_sums': [_vp, _vp, C.c_double, C.c_double, C.c_int, C.c_int, _dp, _dp],
    'odr_oil_set_mixing_stats': [_vp, C.c_double, C.c_double],
    'odr_vertical_advection': [_vp, _vp, C.c_double, C.c_int],
    'odr_vertical_buoyancy': [_vp, _vp, C.c_double],
    'odr_store_previous': [_vp, _vp],
    'odr_coastline': [_vp, _vp, C.c_int, C.c_int, C.c_int, _i64p],
    'odr_coastline_crossing': [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int32, _i64p],
    'odr_increase_age': [_vp, _vp, C.c_double, C.c_double, C.c_int],
    'odr_source_time_coverage': [_vp, C.c_int32, C.c_double, C.c_double, C.c_int],
    'odr_source_release': [_vp, C.c_int32],
    'odr_particles_truncate_z': [_vp, _vp, C.c_double],
    'odr_particles_restore_z': [_vp, _vp],
    'odr_seafloor': [_vp, _vp, _i64p],
    'odr_seafloor_action': [_vp, _vp, C.c_int, C.c_int32, _i64p],
    'odr_set_seafloor_action': [_vp, C.c_int, C.c_int32],
    'odr_deactivate_missing': [_vp, _vp, C.c_int, _ip, C.c_int32, _i64p],
    'odr_particles_count_status': [_vp, _vp, C.c_int32, _i64p],
    'odr_particles_remap_status': [_vp, _vp, C.c_int32, C.c_int32],
    'odr_deactivate': [_vp, _vp, _P(C.c_uint8), C.c_int32],
    'odr_deactivate_outside': [_vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32],
    'odr_compact': [_vp, _vp, _i64p],
    # communication (csrc/odr_comm.hip: RCCL, one communicator pair per process)
    'odr_vmix_set_profile_levels': [_vp, C.c_int32],
    'odr_ctx_set_position_class': [_vp, C.c_int],
    'odr_source_set_coordinate_dtype': [_vp, C.c_int32, C.c_int, C.c_int],
    'odr_device_count': [_P(C.c_int32)],
    'odr_comm_unique_id': [_P(C.c_uint8)],
    'odr_comm_init': [_vp, _P(C.c_uint8), C.c_int32, C.c_int32],
    'odr_comm_info': [_P(C.c_int32), _P(C.c_int32), _P(C.c_uint64), _P(C.c_int32), _P(C.c_uint64)],
    'odr_comm_destroy': [],
    'odr_allreduce_scalars': [_vp, _dp, C.c_int32, C.c_int32],
    'odr_comm_allgather_begin': [_vp, _dp, C.c_int32, C.c_int32],
    'odr_comm_allgather_end': [_vp, _dp],
    'odr_comm_broadcast_bytes': [_vp, _vp, C.c_int64, C.c_int32],
    'odr_comm_barrier': [_vp],
    'odr_block_broadcast': [_vp, C.c_int32, C.c_int32, C.c_double, C.c_int, _ip, _P(_vp), _ip, C.c_int, C.c_int, _dp, C.c_int32],
    'odr_scan_status': [_vp, _vp, _i64p, _P(C.c_uint64)],
    'odr_scan_status_begin': [_vp, _vp],
    'odr_scan_status_end': [_vp, _vp, _i64p, _P(C.c_uint64)],
    'odr_ctx_guard_next_vmix': [_vp, C.c_int],
    'odr_compact_apply': [_vp, _vp, _i64p],
    'odr_sort_particles': [_vp, _vp, C.c_int32],
    'odr_sort_particles_ex': [_vp, _vp, C.c_int32, C.c_int],
    'odr_particles_tile_stats': [_vp, _vp, _P(C.c_uint64)],
    'odr_reduce_scalars': [_vp, _vp, C.c_double, _dp],
    'odr_ctx_set_step_reduce': [_vp, C.c_int, C.c_double, C.c_int],
    'odr_reduce_local': [_vp, _vp, C.c_double, C.c_int, _dp],
    'odr_reduce_install': [_vp, _vp, _dp],
    'odr_reduce_unpin': [_vp],
    'odr_timer_begin': [_vp],
    'odr_timer_end': [_vp, _fp],
    'odr_sgrid_create': [_vp, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, C.c_double, _dp, _dp, C.c_int32, _P(_vp)],
    'odr_sgrid_destroy': [_vp, _vp],
    'odr_sgrid_download_zrho': [_vp, _vp, _dp],
    'odr_sgrid_zslice': [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int32, _dp, C.c_int32, _P(_vp), _dp],
    'odr_history_create': [_vp, C.c_int64, C.c_int32, C.c_int32, _ip, _P(_vp)],
    'odr_history_destroy': [_vp, _vp],
    'odr_history_set_id_base': [_vp, _vp, C.c_int64],
    'odr_history_record': [_vp, _vp, _vp, C.c_int32, C.c_int, C.c_int],
    'odr_history_flush': [_vp, _vp, C.c_int32, C.c_int32],
    'odr_history_wait': [_vp, _vp],
    'odr_history_host_ptr': [_vp, _vp, C.c_int32, _P(_fp), _P(C.c_int32)],
    'odr_history_reset': [_vp, _vp],
    'odr_history_minmax': [_vp, _vp, C.c_int32, _dp, _dp],
}
EXPORTS = sorted(list(_SIGNATURES) + ['odr_last_error', 'odr_version'])

_lib = None


def load():
    """Load the shared library (no compute).  Raises OdrError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OdrError('libodrift_hip.so is missing: run `python -c "import __graft_entry__ as g; '
                       'g.build()"` (hipcc --offload-arch=gfx950).  There is no CPU fallback.')
    # The HIP runtime maps a process's streams onto 4 hardware queues by default.  A context uses three streams (compute,
    # block upload, result-buffer flush); with a second context, or PyTorch's and RCCL's streams, in the same process the
    # compute and the upload stream end up sharing a queue and the uploads no longer overlap with the simulation (measured:
    # OceanDrift.run() 2.4 instead of 1.8 ms per step, profiles/r02_ab_variants.txt section 15).  Read by the runtime when
    # it initialises, so it only has an effect if no HIP call has been made yet; an explicit setting is left alone.
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.odr_last_error.restype = C.c_char_p
    lib.odr_last_error.argtypes = []
    lib.odr_version.restype = C.c_char_p
    lib.odr_version.argtypes = []
    _lib = lib
    return lib


NOISE_NORMAL, NOISE_UNIFORM = 0, 1
DIFFUSIVITY = {'windspeed_Large1994': 1, 'windspeed_Sundby1983': 2}
DROPLETS = {'Johansen et al. (2015)': 1, 'Li et al. (2017)': 2}
OIL_PROPERTIES = ['diameter', 'density', 'viscosity', 'oil_film_thickness', 'diameter_if_entrained']


def check(rc):
    if rc != 0:
        msg = load().odr_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        e = OdrError('odrift error %d: %s' % (rc, msg))
        e.code = rc
        raise e
