/* odrift.h -- C ABI of libodrift_hip.so: the MI355X-native particle-advection
 * hot path behind OpenDrift's model / reader API.
 *
 * The reference (OpenDrift v1.14.10) is pure Python and has NO FFI for this
 * path; its boundary is class inheritance + duck typing.  Each entry point below
 * therefore names the reference METHOD it replaces (file:line under
 * /root/reference).  The binding a maintainer adds on the reference side is a
 * ctypes stub; it is shown in INTEGRATION.md and shipped in
 * opendrift_amd/_abi.py.
 *
 * Conventions: every function returns 0 on success, a negative odr_status on
 * error (odr_last_error() gives a thread-local message).  Opaque handles own
 * all device memory; host buffers belong to the caller and may be freed as soon
 * as the call returns.  Calls are asynchronous on the context's HIP stream
 * unless they return data to the host.  One host thread per context.  No torch
 * types, no callbacks.
 */
#ifndef ODRIFT_H
#define ODRIFT_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct odr_ctx odr_ctx;
typedef struct odr_particles odr_particles;

typedef enum {
  ODR_OK = 0,
  ODR_ERR_INVALID = -1,  /* bad argument (ValueError on the Python side) */
  ODR_ERR_HIP = -2,      /* HIP runtime failure */
  ODR_ERR_CAPACITY = -3, /* particle capacity / slot / table exhausted */
  ODR_ERR_STATE = -4     /* call made in the wrong order (WrongMode analogue) */
} odr_status;

/* ---- environment variables on the path (CF standard names, OceanDrift.required_variables,
 *      opendrift/models/oceandrift.py:60-88) ---- */
enum {
  ODR_VAR_X_SEA_WATER_VELOCITY = 0,
  ODR_VAR_Y_SEA_WATER_VELOCITY = 1,
  ODR_VAR_X_WIND = 2,
  ODR_VAR_Y_WIND = 3,
  ODR_VAR_UPWARD_SEA_WATER_VELOCITY = 4,
  ODR_VAR_OCEAN_VERTICAL_DIFFUSIVITY = 5,
  ODR_VAR_STOKES_DRIFT_X_VELOCITY = 6,
  ODR_VAR_STOKES_DRIFT_Y_VELOCITY = 7,
  ODR_VAR_LAND_BINARY_MASK = 8,
  ODR_VAR_SEA_FLOOR_DEPTH = 9,
  ODR_VAR_SEA_SURFACE_HEIGHT = 10,
  ODR_VAR_HORIZONTAL_DIFFUSIVITY = 11,
  ODR_VAR_SIGNIFICANT_WAVE_HEIGHT = 12,
  ODR_VAR_WAVE_PERIOD = 13,
  ODR_VAR_MIXED_LAYER_THICKNESS = 14,
  ODR_VAR_SEA_WATER_TEMPERATURE = 15, /* OpenOil.required_variables (models/openoil/openoil.py:271-278) */
  ODR_VAR_SEA_WATER_SALINITY = 16,
  ODR_VAR_SEA_ICE_AREA_FRACTION = 17, /* OpenOil.advect_oil in ice (models/openoil/openoil.py:1179-1216) */
  ODR_VAR_SEA_ICE_X_VELOCITY = 18,    /* a vector pair: rotated from the reader's projection (basereader/consts.py:27-36) */
  ODR_VAR_SEA_ICE_Y_VELOCITY = 19,
  /* the windsea_swell Stokes profile (models/physics_methods.py:418-456, 831-841) */
  ODR_VAR_SWELL_WAVE_TO_DIRECTION = 20,
  ODR_VAR_SWELL_WAVE_PEAK_PERIOD = 21,
  ODR_VAR_SWELL_WAVE_SIGNIFICANT_HEIGHT = 22,
  ODR_VAR_WIND_WAVE_TO_DIRECTION = 23,
  ODR_VAR_WIND_WAVE_MEAN_PERIOD = 24,
  ODR_VAR_WIND_WAVE_SIGNIFICANT_HEIGHT = 25,
  ODR_NVAR = 26
};

/* ---- projections of a reader (pyproj.Proj(reader.proj4), basereader/__init__.py:119-137) ---- */
enum { ODR_PROJ_LATLONG = 0, ODR_PROJ_STERE_EQUIT_SPHERE = 1, ODR_PROJ_STERE_POLAR = 2,
       ODR_PROJ_CURVILINEAR = 3 /* set by odr_source_grid_curvilinear, not through odr_proj_desc */,
       ODR_PROJ_MERC = 4 /* +proj=merc (+lat_ts or +k_0), sphere or ellipsoid: Snyder ch. 7 */,
       ODR_PROJ_LCC = 5 /* +proj=lcc +lat_1 [+lat_2] +lat_0 +lon_0, sphere or ellipsoid: Snyder ch. 15 */,
       /* round 5 -- pyproj.Proj of variables.py:111-143 for the other projections model files carry: */
       ODR_PROJ_TMERC = 6 /* +proj=tmerc (+proj=utm: the caller resolves the zone into lon0 / k0 / x0 / y0): Krueger's series to
                             order 6 in the third flattening (Karney 2011), sphere or ellipsoid */,
       ODR_PROJ_LAEA = 7 /* +proj=laea, any aspect, sphere or ellipsoid: Snyder ch. 24 */,
       ODR_PROJ_STERE_OBLIQUE = 8 /* +proj=stere with lat_0 not a pole (oblique, equatorial), sphere or ellipsoid, +k_0: Snyder ch. 21 */,
       ODR_PROJ_OB_TRAN = 9 /* +proj=ob_tran +o_proj=longlat: the rotated pole; lat1_deg = o_lat_p, lat2_deg = o_lon_p, lon0_deg = lon_0;
                               reader coordinates are rotated longitude / latitude in DEGREES (variables.py:117-123,136-138); vector
                               pairs are rotated by the WGS84 azimuth of the 0.1-degree line along +y (variables.py:80-97) */ };
typedef struct {
  int32_t kind;
  double a, es;                  /* semi-major axis, eccentricity squared */
  double lat0_deg, lon0_deg, lat_ts_deg, k0, x0, y0;
  double lat1_deg, lat2_deg;     /* standard parallels (ODR_PROJ_LCC; lat2 = lat1 for the tangent cone); o_lat_p, o_lon_p (ODR_PROJ_OB_TRAN) */
} odr_proj_desc;

enum { ODR_SCHEME_EULER = 0, ODR_SCHEME_RK2 = 1, ODR_SCHEME_RK4 = 2 };
enum { ODR_RNG_DEVICE = 0, ODR_RNG_HOST = 1 }; /* Philox on device | numbers drawn by the caller (np.random parity) */
enum { ODR_COAST_NONE = 0, ODR_COAST_STRANDING = 1, ODR_COAST_PREVIOUS = 2 };
enum { ODR_ANALYTIC_DOUBLE_GYRE = 1, ODR_ANALYTIC_OSCILLATING = 2 };

/* ---------------------------------------------------------------- lifecycle */
int odr_ctx_create(int device, uint64_t seed, odr_ctx **out);
int odr_ctx_destroy(odr_ctx *ctx);
int odr_sync(odr_ctx *ctx);
int odr_set_stream(odr_ctx *ctx, void *hip_stream); /* adopt a caller-owned hipStream_t (NULL = own stream) */
const char *odr_last_error(void);
const char *odr_version(void);

/* ------------------------------------------------- particle state (SoA in HBM)
 * replaces LagrangianArray (opendrift/elements/elements.py:22-254): one array per
 * property; lon/lat/z float64, ID/status/moving int32, drift factors float32. */
int odr_particles_create(odr_ctx *ctx, int64_t capacity, odr_particles **out);
int odr_particles_destroy(odr_ctx *ctx, odr_particles *p);
/* release_elements (basemodel/__init__.py:909-934) / LagrangianArray.extend (elements.py:170-195):
 * append n elements to the active set.  NULL optional arrays take the defaults
 * moving=1, wind_drift_factor=0.02, current_drift_factor=1, terminal_velocity=0. */
int odr_particles_append(odr_ctx *ctx, odr_particles *p, int64_t n, const double *lon,
                         const double *lat, const double *z, const int32_t *id,
                         const int32_t *moving, const float *wind_drift_factor,
                         const float *current_drift_factor, const float *terminal_velocity);
int odr_particles_count(odr_ctx *ctx, odr_particles *p, int64_t *n_active, int64_t *n_deactivated);
/* host copy of the live state (state_to_buffer input, basemodel/__init__.py:2384-2403); any pointer may be NULL */
int odr_particles_download(odr_ctx *ctx, odr_particles *p, double *lon, double *lat, double *z,
                           int32_t *id, int32_t *status, int32_t *moving);
int odr_particles_download_deactivated(odr_ctx *ctx, odr_particles *p, double *lon, double *lat,
                                       double *z, int32_t *id, int32_t *status);
/* overwrite properties of the active set from the host (model subclasses that update elements on the host) */
int odr_particles_upload(odr_ctx *ctx, odr_particles *p, const double *lon, const double *lat,
                         const double *z, const int32_t *moving, const float *wind_drift_factor,
                         const float *current_drift_factor, const float *terminal_velocity);
/* float32 element properties of the active set: "wind_drift_factor", "current_drift_factor", "terminal_velocity",
 * "age_seconds" (LagrangianArray variables, elements/elements.py:71-88) */
int odr_particles_download_f32(odr_ctx *ctx, odr_particles *p, const char *name, float *host);
/* raw device pointers (for torch.distributed / zero-copy consumers): name in
 * {"lon","lat","z","id","status","moving","env:<var_id>"} */
int odr_particles_device_ptr(odr_ctx *ctx, odr_particles *p, const char *name, void **dptr);

/* ------------------------------------------------------- field sources (readers)
 * A source is the device image of one Reader.  Priority lists per variable follow
 * Environment.priority_list (environment.py:339-374). */
/* reader_constant.Reader / environment:constant:<var> (reader_constant.py:60-82, environment.py:172-182) */
int odr_source_constant(odr_ctx *ctx, int nvars, const int32_t *var_ids, const double *values,
                        int32_t *source_id);
/* ContinuousReader analytic fields: reader_double_gyre.py:55-79 params {A, epsilon, omega, t0_epoch};
 * reader_oscillating.py:49-59 params {var_id, amplitude, period_seconds, t0_epoch} */
int odr_source_analytic(odr_ctx *ctx, int kind, const double *params, int nparams,
                        int32_t *source_id);
/* reader_global_landmask.Reader (readers/reader_global_landmask.py:201-255) as a device source: a lon/lat raster,
 * cells[iy * nx + ix] != 0 is land, cell (ix, iy) covers lon0 + [ix, ix+1) dlon x lat0 + [iy, iy+1) dlat; ocean outside.
 * Exact at the element position (ContinuousReader), longitudes modulated to [-180, 180).  The GSHHG data the
 * reference ships through roaring_landmask is not part of this repository: any raster can be handed over. */
int odr_source_landmask(odr_ctx *ctx, int32_t nx, int32_t ny, double lon0, double lat0, double dlon, double dlat,
                        const uint8_t *cells, int32_t *source_id);
/* StructuredReader on a regular grid in its own projection (basereader/structured.py).
 * domain = {xmin,xmax,ymin,ymax,zmin,zmax}; lon_mode 1: [-180,180), 2: [0,360) (variables.py:259-280);
 * z = block z levels (NULL / nz<=1 for surface fields). */
int odr_source_grid(odr_ctx *ctx, const odr_proj_desc *proj, const double *domain6,
                    int lon_mode, int mod360_x, int nz, const double *z, int32_t *source_id);
/* StructuredReader WITHOUT a projection (basereader/structured.py:44-113; reader_ROMS_native.py and every reader that
 * only has 2D lon/lat arrays): x/y are pixel indices, and lonlat2xy (structured.py:438-472) is scipy's
 * LinearNDInterpolator over the Delaunay triangulation of the nodes.  lon/lat: [ny, nx] float64 node coordinates;
 * domain = {0, nx-1, 0, ny-1, zmin, zmax}.  The same triangulation is built here from the structured mesh (cell
 * diagonals + edge flips, csrc/odr_mesh.h), the device walks it per particle; positions outside the
 * mesh outline are "not covered".  ODR_ERR_INVALID for folded / non-convex cells or non-finite nodes. */
int odr_source_grid_curvilinear(odr_ctx *ctx, const double *lon, const double *lat, int ny, int nx,
                                const double *domain6, int lon_mode, int nz, const double *z,
                                int32_t *source_id);
/* Variables.lonlat2xy of one source (variables.py:111-143, longitudes modulated as :259-280) for n host positions ->
 * reader coordinates x, y (NaN outside a curvilinear mesh); synchronous.  Host-side callers (seeding, tests); the
 * step kernels do the same transform in registers. */
int odr_source_lonlat2xy(odr_ctx *ctx, int32_t source_id, int64_t n, const double *lon, const double *lat,
                         double *x, double *y);
/* One time level = one ReaderBlock (interpolation/structured.py:15-94).  data[k] is a host
 * float32 array [var_nz[k], ny, nx] (var_nz 1 => 2D).  xy8 = {x0, xspan, y0, yspan, xmin,
 * xrange, ymin, yrange} with the spans formed in the dtype of the reader's x/y arrays
 * (interpolators.py:32-33,110-111).  The upload masks |v|>1e9, fills NaN towards the
 * seafloor and pre-dilates NaN cells 10x (interpolators.py:9-20,127-137). */
int odr_block_upload(odr_ctx *ctx, int32_t source_id, int32_t slot, double t_epoch, int nvars,
                     const int32_t *var_ids, const float *const *data, const int32_t *var_nz,
                     int ny, int nx, const double *xy8);
/* same, the float32 data already live in device memory (RCCL-broadcast blocks, odr_sgrid_zslice results); with
 * unified addressing each data[k] may individually be a host or a device pointer in either call */
int odr_block_upload_device(odr_ctx *ctx, int32_t source_id, int32_t slot, double t_epoch,
                            int nvars, const int32_t *var_ids, const void *const *dev_data,
                            const int32_t *var_nz, int ny, int nx, const double *xy8);
/* Asynchronous upload: odr_block_upload_async only ENQUEUES the copy and the preparation kernels on the context's
 * upload stream and returns; the simulation continues on the compute stream.  The staged block becomes the content
 * of its slot with odr_block_commit (the compute stream then waits for the upload's event; the block it replaces
 * is released once the compute stream has passed).  data[k] must stay valid until the commit and should be
 * page-locked (odr_host_register) or device memory, otherwise the copies are staged synchronously by the runtime.
 * odr_block_upload = upload_async + wait + commit. */
int odr_block_upload_async(odr_ctx *ctx, int32_t source_id, int32_t slot, double t_epoch, int nvars,
                           const int32_t *var_ids, const void *const *data, const int32_t *var_nz,
                           int ny, int nx, const double *xy8);
int odr_block_commit(odr_ctx *ctx, int32_t source_id, int32_t slot);
/* Optional: content ids of variables of the level just uploaded (the staged level of the slot if there is one, else the
 * resident one).  Two resident levels that carry the same non-zero id for a variable hold the same values for it -- the
 * reference re-reads sea_floor_depth / land_binary_mask with every ReaderBlock (structured.py:15-94) -- and the samplers then
 * gather that variable at one level only (the time interpolation keeps its arithmetic: same bits).  0 = unknown. */
int odr_block_set_content_ids(odr_ctx *ctx, int32_t source_id, int32_t slot, int nvars, const int32_t *var_ids,
                              const uint64_t *ids);
int odr_host_register(odr_ctx *ctx, void *ptr, uint64_t bytes);   /* hipHostRegister: reader arrays that are uploaded repeatedly */
int odr_host_unregister(odr_ctx *ctx, void *ptr);
int odr_block_drop(odr_ctx *ctx, int32_t source_id, int32_t slot);
/* drift:truncate_ocean_model_below_m (models/basemodel/environment.py:554-566): every get_environment call samples the readers
 * at max(z, -truncate_depth) while the elements keep their depth.  odr_particles_truncate_z puts the clipped depths in place for
 * the sampling calls that follow (odr_env_sample, odr_advect / odr_env_coast_advect: the Runge-Kutta stage calls are
 * get_environment calls), odr_particles_restore_z brings the elements' own z back (before anything that changes z or the
 * element set). */
int odr_particles_truncate_z(odr_ctx *ctx, odr_particles *p, double truncate_depth);
int odr_particles_restore_z(odr_ctx *ctx, odr_particles *p);
/* A source that is no longer used (e.g. a gridded reader whose blocks are re-cut to another window and re-registered):
 * drops its resident blocks, removes it from every priority list (odr_env_bind) and frees its id for the next
 * odr_source_* call -- the context holds at most 16 sources at a time, not 16 per run. */
int odr_source_release(odr_ctx *ctx, int32_t source_id);
/* reader.start_time / end_time / always_valid (covers_time, variables.py:392-400): outside the
 * interval the reader is skipped and the next reader / the fallback applies */
int odr_source_time_coverage(odr_ctx *ctx, int32_t source_id, double t_start_epoch, double t_end_epoch,
                             int always_valid);
/* Ensemble data: the reader hands `var` out as a LIST of `members` arrays (readers/basereader/structured.py:125-147);
 * ReaderBlock.interpolate gives element number j of a call the member j % members
 * (readers/interpolation/structured.py:119-135).  Declare it before the blocks are uploaded; the block arrays of `var`
 * then hold the members one after the other along the layer axis ([members x nz][ny][nx], var_nz = members x nz).
 * Element number = rank, in ascending ID (the reference's array order for seed times that are monotonic in ID), among the
 * elements the reference HANDS to the block: the active ones the reader's domain covers at the positions of that call
 * (get_variables_interpolated, variables.py:747-765) -- formed per call by odr_env_sample and, for the stage calls of
 * odr_advect / odr_env_coast_advect, once at the elements' positions (a launch that holds all stages cannot renumber at
 * the stage positions: a host that needs that -- elements crossing the reader's edge -- samples the stages one by one, as
 * opendrift_amd's stage-split lane does).  With several ensemble readers, one behind another reader in its priority
 * list, or in a sharded run (odr_particles_set_rank_offset) every active element is numbered.  Sampled by the generic
 * kernels; ocean_vertical_diffusivity profiles cannot be ensemble data. */
int odr_source_set_members(odr_ctx *ctx, int32_t source_id, int32_t var, int32_t members);
/* Ensemble data hands element number j OF THE CALL member j % M (readers/interpolation/structured.py:119-135): j is the rank
 * of the element among the present ones in ascending ID.  A particle set that holds only a contiguous ID range of a larger
 * simulation (one rank of a sharded run) adds the number of present elements with smaller IDs held elsewhere; the host
 * knows it from the step's collective.  Default 0. */
int odr_particles_set_rank_offset(odr_ctx *ctx, odr_particles *p, int64_t offset);
/* priority list + fallback of one variable (environment.py:592-595,782-791); NaN = no fallback */
int odr_env_bind(odr_ctx *ctx, int32_t var_id, int nsources, const int32_t *source_ids,
                 float fallback);

/* ------------------------------------------------------------ hot-path stages */
/* Environment.get_environment (environment.py:499-923): sample var_ids at the particles'
 * (lon,lat,z) and time into the device environment (float32).  out_host[k] (optional)
 * receives a copy.  The sample positions are remembered for the vertical profiles used by
 * odr_vmix (environment_profiles are taken at these positions, oceandrift.py:431-447). */
int odr_env_sample(odr_ctx *ctx, odr_particles *p, int nvars, const int32_t *var_ids,
                   double t_epoch, float *const *out_host);
int odr_env_download(odr_ctx *ctx, odr_particles *p, int32_t var_id, float *out_host);
int odr_env_upload(odr_ctx *ctx, odr_particles *p, int32_t var_id, const float *host);
/* drift:current_uncertainty / wind_uncertainty (environment.py:869-879,887-891): env[x] += N(0,std), env[y] += N(0,std)
 * (ODR_NOISE_NORMAL); drift:current_uncertainty_uniform (:880-886): env[x] += U(-std,std), env[y] += U(-std,std)
 * (ODR_NOISE_UNIFORM).  ODR_RNG_HOST: host_nx / host_ny = the np.random draws (already scaled by std). */
enum { ODR_NOISE_NORMAL = 0, ODR_NOISE_UNIFORM = 1 };
int odr_env_add_noise(odr_ctx *ctx, odr_particles *p, int32_t var_x, int32_t var_y, double std, int distribution,
                      int rng_mode, const double *host_nx, const double *host_ny, uint64_t step);
/* The same uncertainties INSIDE advect_ocean_current: the reference adds them in every get_environment call whose
 * variables hold the current, i.e. also in the one (RK2) / three (RK4) stage calls (physics_methods.py:638-670 ->
 * environment.py:869-886).  Arms the NEXT odr_advect / odr_env_coast_advect on `p` (call it right before).
 * ODR_RNG_DEVICE: one Philox stream per (ID, step, call, distribution).  ODR_RNG_HOST: host_stage = the draws of the
 * stage calls in np.random order, [nstage][ncomp][n] with ncomp = 2 (normal x, y | uniform x, y) or 4 (normal x, y,
 * uniform x, y), n = active elements; host_main = [ncomp][n] draws of the main-loop sample for odr_env_coast_advect
 * with odr_step_extras.main_noise (NULL otherwise). */
int odr_advect_set_noise(odr_ctx *ctx, odr_particles *p, double std_normal, double std_uniform, int rng_mode,
                         const double *host_main, const double *host_stage, int nstage, uint64_t step);
/* Arithmetic of the Runge-Kutta STAGE evaluations inside odr_advect / odr_env_coast_advect (physics_methods.py:629-670: the
 * sub-stage position geod.fwd(lon, lat, azimuth, speed*dt*.5) and the get_environment call of the current there).
 *   ODR_STAGE_EXACT (default): every float32 rounding point of the reference inside a stage is reproduced (float32
 *     azimuth / distance, each (time, z) layer of the ReaderBlock rounded to float32): <= 1e-10 deg per step vs the oracle.
 *   ODR_STAGE_FAST: stage step taken directly along (u, v) dt/2, the stage value of a gridded reader as one float32
 *     weighted sum of its 16 corner values: <= 2e-9 deg per step vs the oracle (the stage value is a few float32 ulp off).
 * The main-loop sample (o.environment), the RK combination and the final update_positions are identical in both modes;
 * Euler is unaffected.  Applies to every particle set of the context from the next call on. */
enum { ODR_STAGE_EXACT = 0, ODR_STAGE_FAST = 1 };
int odr_ctx_set_stage_math(odr_ctx *ctx, int mode);
/* PhysicsMethods.advect_ocean_current (physics_methods.py:611-691) + update_positions
 * (basemodel/__init__.py:4631-4657), all sub-stages fused in one kernel. */
int odr_advect(odr_ctx *ctx, odr_particles *p, int scheme, double t_epoch, double dt, double factor);
/* One iteration of the run() loop up to and including the current advection
 * (basemodel/__init__.py:2136-2248): odr_env_sample(var_ids, t) -> odr_coastline(action, codes) ->
 * [odr_seafloor -> odr_increase_age, if `extras` asks for them] -> odr_store_previous (if store_previous) ->
 * odr_advect(scheme, t, dt, factor), fused into ONE kernel
 * when the group holding x/y_sea_water_velocity comes from one gridded reader (else the four calls are
 * made in that order).  Elements the coastline deactivates ('stranded', 'seeded_on_land') are
 * flagged and left where the coastline put them -- the reference removes them before update() --
 * so  fused call + odr_compact  is bit-identical to  sample, coastline, compact, store_previous,
 * advect. */
typedef struct {              /* optional bookkeeping of the same loop body, between the coastline and update_previous_state */
  int32_t seafloor_action;    /* 1: interact_with_seafloor 'lift_to_seafloor' (:748-783), else none */
  int32_t retired_code;       /* status of elements older than max_age_seconds */
  double age_dt;              /* increase_age_and_retire (:2342-2352): age_seconds += age_dt; 0: not part of this call */
  double max_age_seconds;     /* 0: no retirement */
  int32_t missing_code;       /* report_missing_variables (:2501-2515), first in the loop's order: elements with a NaN in
                               * any sampled variable (no reader covers them and there is no fallback) get this status;
                               * 0: not part of this call */
  int32_t main_noise;        /* 1: add the armed current uncertainty (odr_advect_set_noise) to the sampled current
                              * right after the sample, like get_environment does (environment.py:869-886) */
  /* OceanDrift.vertical_mixing (+ vertical_advection) of the same step as part of this call: what
   * odr_vmix_fuse_vertical_advection + odr_vmix(t_epoch, dt, vmix_dt_mix, vmix_at_surface, ODR_RNG_DEVICE, NULL,
   * vmix_step) would do after it (vertical mixing reads the environment of the step start and changes z only, so it
   * commutes with the horizontal movers).  When the diffusivity comes from the gridded reader of the current the
   * mixing runs inside the same launch (k_step_grid<..., MIXQ>), otherwise the two calls are made in sequence. */
  int32_t vmix;              /* 0: not part of this call */
  int32_t vmix_at_surface;   /* drift:vertical_mixing_at_surface */
  int32_t vmix_vadv;         /* vertical advection after the mixing: -1 none, 0 below the surface, 1 including it */
  int32_t pad;
  double vmix_dt_mix;        /* vertical_mixing:timestep */
  uint64_t vmix_step;        /* Philox offset (the model step number) */
} odr_step_extras;
int odr_env_coast_advect(odr_ctx *ctx, odr_particles *p, int nvars, const int32_t *var_ids, double t_epoch,
                         int coastline_action, int stranded_code, int seeded_on_land_code,
                         int store_previous, int scheme, double dt, double factor,
                         const odr_step_extras *extras /* NULL: none */, int64_t *n_on_land);
/* update_positions with caller-supplied velocities (models that compute them on the host) */
int odr_update_positions(odr_ctx *ctx, odr_particles *p, const double *x_vel, const double *y_vel,
                         int velocities_are_float32, double dt);

/* Per-element factors of the movers that follow, as OpenOil.advect_oil derives them from the float32
 * sea_ice_area_fraction A of the sampled environment (models/openoil/openoil.py:1182-1216): ICE_CURRENT = 1 - k_ice with
 * k_ice = clip((A - 0.3) / 0.5, 0, 1) for odr_advect and odr_advect_wind, ICE_STOKES = (0.7 - A) / 0.7 (0 above 0.7) for
 * odr_stokes_drift, ICE_DRIFT = k_ice for odr_advect_sea_ice.  With a kind other than SCALAR the scalar `factor`
 * argument of those calls is not used.  Stays in force until set again. */
enum { ODR_FACTOR_SCALAR = 0, ODR_FACTOR_ICE_CURRENT = 1, ODR_FACTOR_ICE_STOKES = 2, ODR_FACTOR_ICE_DRIFT = 3 };
int odr_set_element_factor(odr_ctx *ctx, odr_particles *p, int kind);
/* advect_with_sea_ice (models/physics_methods.py:693-710): update_positions(factor * sea_ice_x/y_velocity) with the
 * sampled ice velocity; without it the rule of thumb current + 1.5 % of the wind; nothing when neither is there. */
int odr_advect_sea_ice(odr_ctx *ctx, odr_particles *p, double dt, double factor);
/* advect_wind (physics_methods.py:712-791) */
int odr_advect_wind(odr_ctx *ctx, odr_particles *p, double dt, double wind_drift_depth,
                    int relative_wind, double factor);
/* stokes_drift (physics_methods.py:793-848); profile 0 monochromatic, 1 exponential, 2 Phillips, 3 windsea_swell
 * (stokes_drift_profile_windsea_swell :418-456: needs the six ODR_VAR_SWELL_* / ODR_VAR_WIND_WAVE_* variables sampled,
 * hs_mode / tp_mode not used); hs_mode/tp_mode 0 environment, 1 from wind, 2 scalar default, tp_mode 3 from wind and
 * read back as float32 */
int odr_stokes_drift(odr_ctx *ctx, odr_particles *p, double dt, int profile, int hs_mode,
                     int tp_mode, double factor);
/* model-specific float32 element properties (slots 0..8); for LeewayObj (models/leeway.py:50-131):
 * 0 downwind_slope 1 crosswind_slope 2 downwind_offset 3 crosswind_offset 4 downwind_eps
 * 5 crosswind_eps 6 jibe_probability 7 orientation 8 capsized.  set: elements [offset, offset+count) */
int odr_particles_set_property(odr_ctx *ctx, odr_particles *p, int slot, int64_t offset, int64_t count,
                               const float *host);
int odr_particles_get_property(odr_ctx *ctx, odr_particles *p, int slot, float *host);
/* copy of property `slot` as it is now, on the device, for the next odr_history_record (ODR_HIST_PROPERTIES_FROM_SNAPSHOT);
 * valid until the set is appended to, compacted or sorted */
int odr_particles_snapshot_property(odr_ctx *ctx, odr_particles *p, int slot);
/* The Leeway loop body between two compactions in ONE launch: Environment.get_environment of `var_ids` at t_epoch
 * (environment.py:499-923; the list must hold x/y_wind and x/y_sea_water_velocity) + drift:current_uncertainty /
 * drift:wind_uncertainty (:869-891; device RNG, 0 = none) + interact_with_coastline (basemodel/__init__.py:670-746) +
 * update_previous_state + Leeway.update (models/leeway.py:430-494, capsizing not included).  Elements the coastline
 * deactivates are flagged and not moved: call + odr_compact is bit-identical to odr_env_sample, odr_env_add_noise x 2,
 * odr_coastline, odr_compact, odr_leeway.  Returns ODR_SPLIT_LANE (1, not an error) when wind, current and land mask do not
 * come from one gridded reader: sampling, noise, coastline and previous state are then done, the caller compacts and calls
 * odr_leeway itself.  n_on_land == NULL: the count is not read back (no host synchronisation). */
enum { ODR_SPLIT_LANE = 1 };
/* report_missing_variables (basemodel/__init__.py:2501-2515) inside the NEXT odr_env_coast_leeway, between the sampling and
 * the coastline as in the loop: an element with NaN in a sampled variable whose fallback is None gets status `code` and does
 * not move (0 = no such test, the default).  One-shot: taken and reset by that call, fused launch or split lane alike. */
int odr_leeway_set_missing_code(odr_ctx *ctx, int32_t code);
int odr_env_coast_leeway(odr_ctx *ctx, odr_particles *p, int nvars, const int32_t *var_ids, double t_epoch, int coast_action,
                         int stranded_code, int seeded_on_land_code, int store_previous, double dt, double capsize_fraction,
                         double std_current, double std_wind, uint64_t step, int64_t *n_on_land);
/* processes:capsizing (models/leeway.py:438-455): call before odr_leeway.  host_uniforms (ODR_RNG_HOST): one number
 * per element, used by those that can be capsized */
int odr_leeway_capsize(odr_ctx *ctx, odr_particles *p, double dt, double wind_threshold, double wind_threshold_sigma,
                       int rng_mode, const double *host_uniforms, uint64_t step);
/* Leeway.update (models/leeway.py:430-494, after the capsizing block): leeway from wind, current,
 * jibing; host_uniforms[i] = np.random.random draws in ODR_RNG_HOST mode */
int odr_leeway(odr_ctx *ctx, odr_particles *p, double dt, double capsize_leeway_fraction, int rng_mode,
               const double *host_uniforms, uint64_t step);
/* horizontal_diffusion (basemodel/__init__.py:1746-1772) */
int odr_hdiffusion(odr_ctx *ctx, odr_particles *p, double dt, int rng_mode,
                   const double *host_nx, const double *host_ny, uint64_t step);
/* advect_wind -> stokes_drift -> horizontal_diffusion (physics_methods.py:712-791, :793-848, basemodel/__init__.py:1746-1772)
 * in ONE launch behind one reduction: `which` = 1 wind | 2 Stokes drift | 4 horizontal diffusion, applied in that order (the
 * reference's); every mover keeps its own arguments, checks and global early-out (a mover that returns early does not call
 * update_positions).  Same results as odr_advect_wind, odr_stokes_drift, odr_hdiffusion called one after the other: a single
 * mover bit for bit; in a chain the later moves form their start-point coefficients from the previous move's (positions equal
 * to the last bit or two of float64, tests/test_gpu_movers.py). */
int odr_movers(odr_ctx *ctx, odr_particles *p, double dt, int which, double wind_drift_depth, int relative_wind,
               double wind_factor, int stokes_profile, int hs_mode, int tp_mode, double stokes_factor, int rng_mode,
               const double *host_nx, const double *host_ny, uint64_t step);
/* OceanDrift.vertical_mixing (oceandrift.py:397-571), environment diffusivity profiles sampled at
 * the positions of the last odr_env_sample; host_uniforms[i_sub*n + i] in ODR_RNG_HOST mode */
int odr_vmix(odr_ctx *ctx, odr_particles *p, double t_epoch, double dt, double dt_mix,
             int mix_at_surface, int rng_mode, const double *host_uniforms, uint64_t step);
/* OceanDrift.vertical_mixing with a wind-parameterised diffusivity profile (oceandrift.py:385-395,425-458;
 * verticaldiffusivity_Large1994 / _Sundby1983, physics_methods.py:203-250): K from each element's wind speed and
 * ocean_mixed_layer_thickness on 1 m levels down to max(MLD)+1.  This is also what the default 'environment' model
 * does when no reader provides ocean_vertical_diffusivity (oceandrift.py:431-447).  Needs x_wind, y_wind,
 * ocean_mixed_layer_thickness, sea_floor_depth_below_sea_level (sea_surface_height) in the environment. */
enum { ODR_DIFFUSIVITY_LARGE1994 = 1, ODR_DIFFUSIVITY_SUNDBY1983 = 2 };
/* drift:truncate_ocean_model_below_m together with reader diffusivity profiles (environment.py:554-566): the reference narrows the
 * depth range it asks the reader for, and a reader that hands out the levels asked for (reader_netCDF_CF_generic.py:414-423,
 * reader_ROMS_native.py:551-560: the span of the request, one level more on either side, plus `verticalbuffer`) returns a block
 * CUT there -- elements below mix on K and dK/dz (np.gradient's edge of the cut grid) of the last level held.  n: levels of that
 * block, taken by the NEXT odr_vmix on the profiles of the device block (which always holds every level); 0 = all. */
int odr_vmix_set_profile_levels(odr_ctx *ctx, int32_t n);
int odr_vmix_wind_profile(odr_ctx *ctx, odr_particles *p, int model, double background_diffusivity, double dt,
                          double dt_mix, int mix_at_surface, int rng_mode, const double *host_uniforms,
                          uint64_t step);
/* OpenOil (models/openoil/openoil.py): the oil physics INSIDE the vertical-mixing loop.  Element properties live in
 * the property slots of odr_particles_set_property: */
enum { ODR_OIL_DIAMETER = 0, ODR_OIL_DENSITY = 1, ODR_OIL_VISCOSITY = 2, ODR_OIL_FILM_THICKNESS = 3,
       ODR_OIL_DIAMETER_IF_ENTRAINED = 4 /* written by odr_oil_prepare_mixing */ };
enum { ODR_DROPLETS_JOHANSEN2015 = 1, ODR_DROPLETS_LI2017 = 2 };   /* wave_entrainment:droplet_size_distribution */
/* OpenOil.prepare_vertical_mixing (openoil.py:1017-1031): entrainment probability after Li et al. 2017
 * (physics_methods.py:115-137) and one droplet diameter per element drawn from the Johansen et al. 2015 / Li et al.
 * 2017 spectrum (:1072-1172: 1e6-point log-normal around the MEAN median diameter of all elements, np.random.choice).
 * It also arms the next odr_vmix / odr_vmix_wind_profile call on `p`: that call then runs the loop as OpenOil does --
 * update_terminal_velocity in every sub-step (:922-998; elements.terminal_velocity is overwritten), surface_stick
 * (:1056-1061), surface_wave_mixing (:1033-1054; entrained elements take their new diameter unless
 * keep_droplet_diameter).  hs_mode / tp_mode as in odr_stokes_drift, plus tp_mode 3 = period from the wind as stored
 * in the float32 environment (physics_methods.py:876-883).  temperature_to_kelvin: oil_weathering_noaa's in-place
 * conversion (:722-724) has happened.  sea_water_density: PhysicsMethods.sea_water_density() (T=10, S=35).
 * ODR_RNG_HOST: host_u_diameter[n] (the uniforms behind np.random.choice), host_u_entrain[i_sub*n + i] (np.random.uniform(0,1)),
 * host_u_intrusion[i_sub*n + i] (unit draws of np.random.uniform(0, mean(1.5 Hs)) placed at the element they belong to). */
int odr_oil_prepare_mixing(odr_ctx *ctx, odr_particles *p, double dt, double dt_mix, double interfacial_tension,
                           double sea_water_density, int droplet_distribution, int keep_droplet_diameter, int hs_mode,
                           int tp_mode, int temperature_to_kelvin, int rng_mode, const double *host_u_diameter,
                           const double *host_u_entrain, const double *host_u_intrusion, uint64_t step);
int odr_oil_mixing_stats(odr_ctx *ctx, double *mean_zb, double *dv50);
/* Sharded run (one process per GPU): OpenOil takes two means over ALL elements -- np.mean(dV_50) of the droplet spectrum
 * (openoil.py:1099-1101,1156-1158) and np.mean(1.5 Hs) (:1047).  odr_oil_local_sums returns this rank's sums (the
 * float32 1.5 Hs summed in float64), the caller adds sums and element counts over the ranks and installs the means;
 * the next odr_oil_prepare_mixing uses them instead of reducing over its own elements. */
int odr_oil_local_sums(odr_ctx *ctx, odr_particles *p, double interfacial_tension, double sea_water_density,
                       int droplet_distribution, int hs_mode, double *sum_dv50, double *sum_zb);
int odr_oil_set_mixing_stats(odr_ctx *ctx, double mean_zb, double dv50);
/* performance hint: apply vertical_advection (oceandrift.py:315-350) inside the next odr_vmix
 * kernel (OceanDrift.update() calls them back to back, oceandrift.py:201-208) */
int odr_vmix_fuse_vertical_advection(odr_ctx *ctx, int at_surface);
/* vertical_advection (oceandrift.py:315-350) / vertical_buoyancy (:352-368) */
int odr_vertical_advection(odr_ctx *ctx, odr_particles *p, double dt, int at_surface);
int odr_vertical_buoyancy(odr_ctx *ctx, odr_particles *p, double dt);
/* update_previous_state (basemodel/__init__.py:642-668): store lon/lat for the 'previous'
 * coastline / seafloor actions.  Newly appended elements start with previous = own position
 * (release_elements, :925-929). */
int odr_store_previous(odr_ctx *ctx, odr_particles *p);
/* interact_with_coastline (basemodel/__init__.py:670-746, approximation precision None) and
 * interact_with_seafloor 'lift_to_seafloor' (:748-783) on the current environment */
int odr_coastline(odr_ctx *ctx, odr_particles *p, int action, int stranded_code,
                  int seeded_on_land_code /* 'previous': deactivate age==0 elements on land, 0 = off */,
                  int64_t *n_on_land);
/* interact_with_coastline with general:coastline_approximation_precision (basemodel/__init__.py:694-746): elements on
 * land are moved to the coastline found by coastline_crossing (:81-134) between their previous (odr_store_previous)
 * and current position -- first sample on land ('stranding') or last sample in water ('previous') of the reference's
 * sampling pattern with step `precision_deg`, evaluated on the landmask raster `landmask_source`. */
int odr_coastline_crossing(odr_ctx *ctx, odr_particles *p, int action, int stranded_code, int seeded_on_land_code,
                           double precision_deg, int32_t landmask_source, int64_t *n_on_land);
/* report_missing_variables (basemodel/__init__.py:2501-2515) on the result of the last odr_env_sample: elements for
 * which any of var_ids is NaN in the environment (Environment.get_environment's `missing`, environment.py:903-908)
 * are deactivated with status_code ('missing_data') */
int odr_deactivate_missing(odr_ctx *ctx, odr_particles *p, int nvars, const int32_t *var_ids,
                           int32_t status_code, int64_t *n_missing);
/* increase_age_and_retire (basemodel/__init__.py:2342-2352); max_age_seconds <= 0: no retirement */
int odr_increase_age(odr_ctx *ctx, odr_particles *p, double dt, double max_age_seconds, int retired_code);
int odr_seafloor(odr_ctx *ctx, odr_particles *p, int64_t *n_below);   /* 'lift_to_seafloor' */
/* the other general:seafloor_action values (:768-783): DEACTIVATE flags the element with status_code ('seafloor') and
 * puts it on the sea floor, PREVIOUS moves it back to the lon/lat of odr_store_previous (z unchanged) */
enum { ODR_SEAFLOOR_LIFT = 1, ODR_SEAFLOOR_DEACTIVATE = 2, ODR_SEAFLOOR_PREVIOUS = 3 };
int odr_seafloor_action(odr_ctx *ctx, odr_particles *p, int action, int32_t status_code, int64_t *n_below);
/* The reference calls interact_with_seafloor() again INSIDE update(): from vertical_buoyancy (oceandrift.py:362-368)
 * and from every sub-step of vertical_mixing (:555-559).  This sets what odr_vertical_buoyancy / odr_vmix* do with an
 * element below the sea floor there (default ODR_SEAFLOOR_LIFT; 0 = 'none'). */
int odr_set_seafloor_action(odr_ctx *ctx, int action, int32_t status_code);
/* number of active-set elements currently flagged with status_code (not yet removed by odr_compact) */
int odr_particles_count_status(odr_ctx *ctx, odr_particles *p, int32_t status_code, int64_t *n);
/* status_categories grow in the order in which reasons FIRST OCCUR (deactivate_elements, :1778-1781): a caller hands
 * out a provisional code for a reason not seen yet, counts it, and renumbers it once the category index is known */
int odr_particles_remap_status(odr_ctx *ctx, odr_particles *p, int32_t from_code, int32_t to_code);
/* deactivate elements flagged by the host (deactivate_elements, :1774-1795) */
int odr_deactivate(odr_ctx *ctx, odr_particles *p, const uint8_t *mask_host, int32_t status_code);
/* remove_deactivated_elements (:1797-1826) = LagrangianArray.move_elements (elements.py:197-228):
 * stable compaction of status==0 elements, the rest appended to the deactivated store */
/* deactivate_outside (basemodel/__init__.py:2354-2382): drift:deactivate_west_of/east_of/south_of/north_of;
 * a NaN bound is "not set" */
int odr_deactivate_outside(odr_ctx *ctx, odr_particles *p, double west, double east, double south, double north,
                           int32_t status_code);
int odr_compact(odr_ctx *ctx, odr_particles *p, int64_t *n_active);
/* odr_compact in two halves, for a loop that needs the deactivation state on the host once per step:
 * odr_scan_status counts the elements that stay and reports which PROVISIONAL status numbers are present (bit k of
 * *status_flags: some element carries status 100 + k -- a deactivation reason without a status category yet, which the
 * caller registers and renumbers with odr_particles_remap_status, basemodel/__init__.py:1778-1781) in ONE host read;
 * odr_compact_apply then removes the deactivated elements without another read.  No call that deactivates, adds or
 * re-orders elements may run in between (ODR_ERR_STATE otherwise). */
int odr_scan_status(odr_ctx *ctx, odr_particles *p, int64_t *n_kept, uint64_t *status_flags);
/* The same read in two halves, so that the loop's next launch need not wait for the host (round 5).  After
 * odr_env_coast_advect, odr_scan_status_begin enqueues the fold of the counts that launch left and returns at once (1 = that
 * launch left none -- another path ran --: call odr_scan_status instead); the device also keeps the verdict "every element
 * stays".  odr_ctx_guard_next_vmix(ctx, 1) makes the NEXT odr_vmix a launch that does nothing unless that verdict is "yes":
 * the mixing of the step (oceandrift.py:397-571) is enqueued BEFORE the host knows whether remove_deactivated_elements
 * (basemodel/__init__.py:2284) has anything to remove -- in the common step it has not, and the device goes from the step
 * launch into the mixing launch without waiting for the host; otherwise the host compacts and calls odr_vmix again, unguarded
 * (the guarded launch has not touched anything).  A guarded odr_vmix that cannot honour the guard (another kernel family than
 * the reader-profile fast path) launches nothing and returns 1.  odr_scan_status_end waits for the fold only (not for the
 * mixing launch behind it) and hands out what odr_scan_status would have. */
int odr_scan_status_begin(odr_ctx *ctx, odr_particles *p);
int odr_scan_status_end(odr_ctx *ctx, odr_particles *p, int64_t *n_kept, uint64_t *status_flags);
int odr_ctx_guard_next_vmix(odr_ctx *ctx, int on);
int odr_compact_apply(odr_ctx *ctx, odr_particles *p, int64_t *n_active);
/* Device-side layout operation with no reference counterpart: re-order the particle arrays by
 * the grid cell of gridded source `source_id` so that the lanes of a wavefront gather
 * neighbouring grid nodes.  Particles keep their ID; results per ID are unchanged (the device
 * RNG is counter-based on ID).  Do not use together with ODR_RNG_HOST arrays, which are
 * indexed by position. */
int odr_sort_particles(odr_ctx *ctx, odr_particles *p, int32_t source_id);
/* keep_environment = 0: the sampled environment and the sample position are not carried along (the next odr_env_sample /
 * odr_env_coast_advect rewrites them for every element): a re-sort at the top of a step moves half the bytes */
int odr_sort_particles_ex(odr_ctx *ctx, odr_particles *p, int32_t source_id, int keep_environment);
/* The sort also leaves a table of workgroup ranges (each inside one 8x8-cell sort tile): while it is valid (no element
 * appended since), odr_env_coast_advect with a Runge-Kutta scheme on a lon/lat or polar-stereographic reader runs its
 * launch with the node records of each workgroup's rectangle staged in LDS (csrc/odr_tile.hip.h) -- same bits as the
 * launch that gathers from the blocks in HBM.  Diagnostics of that path since the set was created:
 * out4 = {launches that took the LDS-tile path, elements handed to the HBM path by those launches (footprint outside their
 *         workgroup's rectangle), workgroups whose rectangle was cut to the LDS capacity, ranges in the current table}.
 * Synchronises the context's stream. */
int odr_particles_tile_stats(odr_ctx *ctx, odr_particles *p, uint64_t *out4);
/* counts and min/max used for the per-step log line and early-outs (:2212-2233):
 * out16 = {n_active, lon_min, lon_max, lat_min, lat_max, z_min, z_max, D_max, stokes_sum_max,
 *          wind_speed_max, wdf_surface_max, n_surface, hs_max, tp_max, 0, 0} */
int odr_reduce_scalars(odr_ctx *ctx, odr_particles *p, double wind_drift_depth, double *out16);
/* The global tests the movers open with -- no element at the surface, wind_drift_factor / (relative) wind speed / Stokes
 * drift / horizontal diffusivity identically zero (physics_methods.py:741-747,771-780,799-804, basemodel/__init__.py:1754)
 * -- formed by the launch of odr_env_coast_advect, which holds every value they read in registers, instead of by a pass of
 * its own over the arrays before the first mover (persistent setting; off by default).  The movers that follow
 * (odr_movers, odr_advect_wind with the same wind_drift_depth / relative_wind, odr_stokes_drift, odr_hdiffusion) take the
 * tests from there as long as nothing they depend on changed in between (a compaction does not; vertical mixing does:
 * the pass is then made as before).  Same results either way (tests/test_gpu_movers.py). */
int odr_ctx_set_step_reduce(odr_ctx *ctx, int on, double wind_drift_depth, int relative_wind);
/* The same reductions for a run sharded over several GPUs (one process per GPU): odr_reduce_local returns the RAW slots
 * of this particle set (0 and 11 are counts, every other slot a maximum; minima negated), the caller combines them over
 * the ranks (sum / max: an all-reduce of 16 doubles) and installs the result; until odr_reduce_unpin the movers
 * (odr_advect_wind, odr_stokes_drift, odr_hdiffusion, odr_vmix_wind_profile) use the installed values -- their global
 * early-outs and MLD.max() are then those of the reference's single process, independent of the number of ranks. */
int odr_reduce_local(odr_ctx *ctx, odr_particles *p, double wind_drift_depth, int relative_wind, double *out16);
int odr_reduce_install(odr_ctx *ctx, odr_particles *p, const double *in16);
int odr_reduce_unpin(odr_ctx *ctx);

/* Until the first update_positions of a run the reference's elements.lon / lat are float32 ARRAYS (elements/elements.py:71-88),
 * so modulate_longitude (readers/basereader/variables.py:259-280, called on the elements' longitudes :914) forms
 * np.mod(lon + 180, 360) - 180 in float32 in the FIRST get_environment: the readers are sampled at lon on the float32 grid of
 * lon + 180 (up to 8e-6 deg off).  f32 != 0: the main-loop samples that follow (odr_env_sample, odr_env_coast_advect,
 * odr_env_coast_leeway; not the Runge-Kutta stage calls, whose positions are float64 there) do the same; 0 ends it. */
int odr_ctx_set_position_class(odr_ctx *ctx, int f32);
/* The reader's x / y coordinate arrays are float32: in the float32 position class the index maps of a GEOGRAPHIC reader --
 * (x - xgrid[0]) / (xgrid[-1] - xgrid[0]) * (len - 1), interpolators.py:110-111, and the nearest-node map :32-37 -- are then
 * float32 arithmetic as well (x is the float32 longitude itself); projected readers get float64 x, y from pyproj. */
int odr_source_set_coordinate_dtype(odr_ctx *ctx, int32_t source_id, int x_is_float32, int y_is_float32);

/* ---------------------------------------------------------------- communication of a sharded run
 * One process per GPU; the reference has no multi-process mode (docs/source/performance.rst:22,36 suggests running several
 * simulations side by side), so there is no reference call to mirror: these are the three entry points SURVEY.md 8(b) lists
 * (odr_comm_init, odr_block_broadcast, odr_allreduce_scalars) plus the step's one collective.  They run over RCCL (librccl.so.1,
 * opened on first use; csrc/odr_comm.hip) -- no torch in the process.  ONE communicator pair per process: `step` for the small
 * collectives of the loop, `bulk` for reader levels (operations on one communicator execute in issue order; a 200 MB level
 * must not hold the step's 200-byte summary behind it).  Every rank makes the same calls in the same order. */
#define ODR_COMM_ID_BYTES 256
int odr_device_count(int32_t *n);
/* rank 0: the id of a new job (two ncclUniqueIds); the host hands it to the other ranks (opendrift_amd/distributed.py: a file) */
int odr_comm_unique_id(uint8_t *id /* [ODR_COMM_ID_BYTES] */);
/* ncclCommInitRank x 2 on the context's device; collective over all ranks */
int odr_comm_init(odr_ctx *ctx, const uint8_t *id, int32_t rank, int32_t nranks);
/* nranks = 0: no communicator.  id_hash: the same number on every rank of one job; collectives: calls made so far */
int odr_comm_info(int32_t *rank, int32_t *nranks, uint64_t *id_hash, int32_t *rccl_version, uint64_t *collectives);
int odr_comm_destroy(void);
/* in place, blocking; op 0 sum | 1 min | 2 max (counts of elements, extents; environment.py / basemodel bookkeeping over ALL elements) */
int odr_allreduce_scalars(odr_ctx *ctx, double *values, int32_t n, int32_t op);
/* THE collective of a sharded step, in two halves: every rank's row of n float64, gathered as [nranks][n].  _begin enqueues
 * (copy in, ncclAllGather, copy out into page-locked memory) on the communication stream and returns; _end waits.
 * from_scan != 0 (between odr_scan_status_begin and _end): row[0] = elements that stay and row[1..8] = status flags are
 * taken ON THE DEVICE from the fold of the step's status scan -- the collective starts when the device has the counts, not
 * when the host has read them; the mixing launch of the step runs meanwhile. */
int odr_comm_allgather_begin(odr_ctx *ctx, const double *row, int32_t n, int32_t from_scan);
int odr_comm_allgather_end(odr_ctx *ctx, double *rows /* [nranks][n] */);
/* host bytes (metadata of a reader level, pickled by the host mirror) from root to every rank; blocking; same nbytes everywhere */
int odr_comm_broadcast_bytes(odr_ctx *ctx, void *buf, int64_t nbytes, int32_t root);
int odr_comm_barrier(odr_ctx *ctx);
/* One reader time level (= one ReaderBlock, interpolation/structured.py:15-94) from the rank that runs the host Reader to
 * every rank: odr_block_upload_async with ONE ncclBroadcast of the level's staged float32 arrays in it (in place in the
 * upload pipeline's staging memory, on the upload stream, bulk communicator) -- `data` is read on `root` only (NULL
 * elsewhere), shapes and geometry are given by every rank.  Staged like an asynchronous upload: odr_block_commit makes it
 * current (a period later, when it is due). */
int odr_block_broadcast(odr_ctx *ctx, int32_t source_id, int32_t slot, double t_epoch, int nvars, const int32_t *var_ids,
                        const void *const *data, const int32_t *var_nz, int ny, int nx, const double *xy8, int32_t root);

/* kernel timing hook for bench.py: HIP events recorded on the context stream around the
 * launches issued between begin and end; returns milliseconds */
int odr_timer_begin(odr_ctx *ctx);
int odr_timer_end(odr_ctx *ctx, float *ms);

/* ---------------------------------------------------------------- output history
 * state_to_buffer (basemodel/__init__.py:2384-2499) and the float32 result buffer it fills
 * (:2084-2105): every exported variable is a float32 [trajectory, time] array initialised with NaN;
 * at an output time all elements present are written at (ID, time) -- float64 / int32 properties are
 * cast to float32 by the assignment -- and between output times only the deactivated elements are,
 * into the slot of the next output time (:2390-2397).  The buffer stays in HBM; a flush copies it to
 * pinned host memory on a second stream while the simulation continues.
 * Variable codes: an ODR_VAR_* id (environment variable) or one of ODR_HIST_*. */
enum {
  ODR_HIST_LON = 1000, ODR_HIST_LAT, ODR_HIST_Z, ODR_HIST_STATUS, ODR_HIST_MOVING, ODR_HIST_AGE_SECONDS,
  ODR_HIST_WIND_DRIFT_FACTOR, ODR_HIST_CURRENT_DRIFT_FACTOR, ODR_HIST_TERMINAL_VELOCITY,
  ODR_HIST_PROPERTY0 = 2000 /* + slot of odr_particles_set_property */
};
typedef struct odr_history odr_history;
int odr_history_create(odr_ctx *ctx, int64_t n_trajectories, int32_t n_times, int32_t nvars,
                       const int32_t *var_codes, odr_history **out);
/* trajectory row 0 of the buffer = element `id_base` (default 0): one rank of a particle-sharded run records its own
 * contiguous ID range */
int odr_history_set_id_base(odr_ctx *ctx, odr_history *h, int64_t id_base);
int odr_history_destroy(odr_ctx *ctx, odr_history *h);
/* position_from_previous (bit 0): lon / lat are taken from the state saved by update_previous_state (odr_store_previous,
 * odr_env_coast_advect) -- the position before this step's advection -- so that the record may follow the fused launch.
 * ODR_HIST_PROPERTIES_FROM_SNAPSHOT (bit 1): a property that odr_particles_snapshot_property has copied is read from
 * that copy (Leeway: crosswind_slope / orientation as they were before the jibes of odr_env_coast_leeway). */
enum { ODR_HIST_PROPERTIES_FROM_SNAPSHOT = 2 };
int odr_history_record(odr_ctx *ctx, odr_particles *p, odr_history *h, int32_t time_index, int only_deactivated,
                       int position_from_previous);
/* asynchronous: extract time slots [t0, t0+nt) of every variable into pinned host memory as
 * [trajectory][time] float32 (the reference's dims); _wait blocks; _host_ptr is valid until the next flush */
int odr_history_flush(odr_ctx *ctx, odr_history *h, int32_t t0, int32_t nt);
int odr_history_wait(odr_ctx *ctx, odr_history *h);
int odr_history_host_ptr(odr_ctx *ctx, odr_history *h, int32_t var_index, float **ptr, int32_t *nt);
int odr_history_reset(odr_ctx *ctx, odr_history *h);   /* new buffer: NaN (:2493-2499) */
int odr_history_minmax(odr_ctx *ctx, odr_history *h, int32_t var_index, double *minval, double *maxval); /* :2409-2414 */

/* ---------------------------------------------------------------- ROMS sigma grid
 * The sigma -> z regridding reader_ROMS_native.get_variables applies to every 4-D variable of a block
 * (reader_ROMS_native.py:512-538,617-684) with roppy (readers/roppy/depth.py): sdepth (:31-113, rho points,
 * Vtransform 1 | 2, S = NULL: -1 + (k + 0.5)/N), z_rho -= zeta, positive z_rho -> NaN; multi_zslice (:213-284),
 * values > 1e9 -> NaN.  float64 arithmetic in NumPy's operation order.  H, zeta: [ny][nx] host arrays. */
typedef struct odr_sgrid odr_sgrid;
int odr_sgrid_create(odr_ctx *ctx, int32_t ny, int32_t nx, int32_t N, const double *H, const double *zeta_or_null,
                     double Hc, const double *Cs_r, const double *S_or_null, int32_t Vtransform, odr_sgrid **out);
int odr_sgrid_destroy(odr_ctx *ctx, odr_sgrid *g);
int odr_sgrid_download_zrho(odr_ctx *ctx, odr_sgrid *g, double *out /* [N][ny][nx] */);
/* field [N][ny][nx] float32 (is_f64 = 0) or float64, host or device memory -> *out_dev32: device float32
 * [kmax][ny][nx] in result slot out_slot (0..7; valid until the next call with the same slot -- one slot per
 * variable of a block, then pass the pointers to odr_block_upload_device); out_host64 (optional): the float64 result */
int odr_sgrid_zslice(odr_ctx *ctx, odr_sgrid *g, const void *field, int is_f64, int on_device, int32_t kmax,
                     const double *Z, int32_t out_slot, void **out_dev32, double *out_host64);

#ifdef __cplusplus
}
#endif
#endif /* ODRIFT_H */
