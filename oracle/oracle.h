/* TEST INFRASTRUCTURE ONLY -- CPU oracle, never on the product path.
 *
 * Plain-C restatement of the reference's per-timestep particle advection path
 * (OpenDrift v1.14.10).  Every function cites the reference file:line it
 * follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library.
 */
#ifndef ODR_ORACLE_H
#define ODR_ORACLE_H
#include "geodesic.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- variables on the hot path (same ids as include/odrift.h) ---- */
enum {
  ORC_VAR_U = 0,      /* x_sea_water_velocity */
  ORC_VAR_V = 1,      /* y_sea_water_velocity */
  ORC_VAR_XWIND = 2,  /* x_wind */
  ORC_VAR_YWIND = 3,  /* y_wind */
  ORC_VAR_W = 4,      /* upward_sea_water_velocity */
  ORC_VAR_KZ = 5,     /* ocean_vertical_diffusivity */
  ORC_VAR_STOKES_X = 6,
  ORC_VAR_STOKES_Y = 7,
  ORC_VAR_LAND = 8,   /* land_binary_mask */
  ORC_VAR_DEPTH = 9,  /* sea_floor_depth_below_sea_level */
  ORC_VAR_SSH = 10,   /* sea_surface_height */
  ORC_VAR_HDIFF = 11, /* horizontal_diffusivity */
  ORC_VAR_HS = 12,    /* sea_surface_wave_significant_height */
  ORC_VAR_TP = 13,    /* sea_surface_wave_period_at_variance_spectral_density_maximum */
  ORC_VAR_MLD = 14,   /* ocean_mixed_layer_thickness */
  ORC_VAR_TEMP = 15,  /* sea_water_temperature (OpenOil.required_variables, openoil.py:271-278) */
  ORC_VAR_SALT = 16,  /* sea_water_salinity */
  ORC_VAR_ICE_A = 17, /* sea_ice_area_fraction (OpenOil.advect_oil, openoil.py:1179-1216) */
  ORC_VAR_ICE_U = 18, /* sea_ice_x_velocity */
  ORC_VAR_ICE_V = 19, /* sea_ice_y_velocity */
  ORC_VAR_SWELL_DIR = 20, /* sea_surface_swell_wave_to_direction (windsea_swell Stokes profile, physics_methods.py:418-456) */
  ORC_VAR_SWELL_TP = 21,  /* sea_surface_swell_wave_peak_period_from_variance_spectral_density */
  ORC_VAR_SWELL_HS = 22,  /* sea_surface_swell_wave_significant_height */
  ORC_VAR_WW_DIR = 23,    /* sea_surface_wind_wave_to_direction */
  ORC_VAR_WW_TM = 24,     /* sea_surface_wind_wave_mean_period */
  ORC_VAR_WW_HS = 25,     /* sea_surface_wind_wave_significant_height */
  ORC_NVAR = 26
};

/* ---- projections (proj.c) ---- */
enum { ORC_PROJ_LATLONG = 0, ORC_PROJ_STERE_EQUIT_SPHERE = 1, ORC_PROJ_STERE_POLAR = 2, ORC_PROJ_MERC = 4, ORC_PROJ_LCC = 5,
       ORC_PROJ_TMERC = 6 /* +proj=tmerc / utm */, ORC_PROJ_LAEA = 7, ORC_PROJ_STERE_OBLIQUE = 8 /* +proj=stere, lat_0 not a pole */,
       ORC_PROJ_OB_TRAN = 9 /* +proj=ob_tran +o_proj=longlat: rotated pole, x / y in DEGREES as the reference's lonlat2xy hands them out */ };
typedef struct {
  int kind, south;
  double a, es, e, lon0, lat0, x0, y0, k0, akm1;
  double n, c, rho0;   /* ORC_PROJ_LCC: cone constant, F, rho0 / a (orc_proj_init_conic) */
  int mode, pad;       /* ORC_PROJ_LAEA / STERE_OBLIQUE aspect: 0 north pole, 1 south pole, 2 equatorial, 3 oblique */
  double q[16];        /* set-up constants of the round-5 projections (proj.c: orc_proj_init_ext) */
} orc_proj;
/* tmerc (utm: the caller resolves the zone into lon0 / k0 / x0 / y0), laea, oblique / equatorial stere on sphere or ellipsoid,
 * ob_tran with o_proj=longlat (lat1 = o_lat_p, lat2 = o_lon_p, lon0 = lon_0; a, es, k0, x0, y0 unused) */
void orc_proj_init_ext(orc_proj *p, int kind, double a, double es, double lat0_deg, double lon0_deg, double k0, double x0,
                       double y0, double lat1_deg, double lat2_deg);
/* merc (lat1/lat2 unused; k0 from lat_ts when that is not 0) and lcc (standard parallels lat1, lat2 = lat1 for a tangent cone) */
void orc_proj_init_conic(orc_proj *p, int kind, double a, double es, double lat0_deg, double lon0_deg, double lat_ts_deg,
                         double k0, double x0, double y0, double lat1_deg, double lat2_deg);
void orc_proj_init(orc_proj *p, int kind, double a, double es, double lat0_deg,
                   double lon0_deg, double lat_ts_deg, double k0, double x0, double y0);
void orc_proj_fwd(const orc_proj *p, double lon_deg, double lat_deg, double *x, double *y);
void orc_proj_inv(const orc_proj *p, double x, double y, double *lon_deg, double *lat_deg);

/* ---- gridded blocks (interp.c) ---- */
typedef struct {
  int nz, ny, nx;
  /* Linear2DInterpolator index map  xi=(x-x0)/xspan*(nx-1)  (interpolators.py:110-111) */
  double x0, xspan, y0, yspan;
  /* Nearest2DInterpolator index map xi=round((x-xmin)/xrange*nx) (interpolators.py:32-37) */
  double xmin, xrange, ymin, yrange;
  const double *z; /* nz levels (NULL when nz<=1) */
  double t;        /* epoch seconds of this time level */
  float *data[ORC_NVAR];       /* NULL if absent; mutated by the in-place NaN dilation */
  int var_nz[ORC_NVAR];        /* 1 => [ny,nx], nz => [nz,ny,nx] */
  int members[ORC_NVAR];       /* > 1: the reader hands the variable out as a LIST of ensemble members; data holds them
                                * one after the other, each [var_nz][ny][nx] (readers/interpolation/structured.py:119-135) */
} orc_block;

/* expand_numpy_array (interpolators.py:9-20): one 3x3 grey dilation of the NaN cells. */
void orc_dilate_nan_once(float *a, int ny, int nx);
/* fill_NaN_towards_seafloor (interpolators.py:203-211) */
void orc_fill_nan_towards_seafloor(float *a, int nz, int ny, int nx);
/* scipy.ndimage.map_coordinates(order=1) on one f32 layer; mode 0 = constant(cval=nan), 1 = nearest */
float orc_bilinear_f32(const float *a, int ny, int nx, double yi, double xi, int mode_nearest);
/* Linear2DInterpolator.__call__ incl. the dilate-and-retry loop; MUTATES `a` like the reference. */
void orc_linear2d_call(float *a, int ny, int nx, long n, const double *yi, const double *xi, float *out);

/* ---- field sources / environment (step.c) ---- */
enum { ORC_SRC_CONSTANT = 0, ORC_SRC_DOUBLE_GYRE = 1, ORC_SRC_OSCILLATING = 2, ORC_SRC_GRID = 3 };
#define ORC_MAXLEVELS 4
typedef struct {
  int kind;
  orc_proj proj;
  double xmin, xmax, ymin, ymax, zmin, zmax; /* coverage in reader coordinates */
  int lon_mode;       /* modulate_longitude (variables.py:259-280): 1 => [-180,180), 2 => [0,360) */
  int mod360_x;       /* structured.py:212-214 geographic reader with xmin>0 */
  int has_var[ORC_NVAR];
  double const_val[ORC_NVAR]; /* CONSTANT */
  double params[8];          /* DOUBLE_GYRE: A, epsilon, omega, t0 ; OSCILLATING: var, amplitude, period_s, t0 */
  int nlevels;               /* GRID: resident time levels, ascending t */
  orc_block level[ORC_MAXLEVELS];
  int always_valid;
  double tmin, tmax;  /* covers_time (variables.py:392-400): reader start_time / end_time */
  int xy_f32;         /* GRID: bit 0 / 1 = the reader's x / y coordinate arrays are float32 (index arithmetic of a run's first call) */
} orc_source;

typedef struct {
  int nsrc;
  const orc_source *src;
  int nlist[ORC_NVAR];      /* priority list per variable (environment.py priority_list) */
  int list[ORC_NVAR][4];
  float fallback[ORC_NVAR]; /* environment:fallback:<var>, NaN if none */
} orc_world;

/* Environment.get_environment (environment.py:499-923) minus uncertainty noise:
 * out[v][i] float32, NaN where missing. */
void orc_set_position_class(int f32);   /* step.c: float32 element arrays of the first get_environment of a run */
void orc_get_environment(const orc_world *w, int nv, const int *vars, long n,
                         const double *lon, const double *lat, const double *z,
                         double t, float *const *out);

/* profiles (structured.py:231-241,366-385): out[k*n+i] for k<nz_prof, no z interpolation */
void orc_get_profile(const orc_world *w, int var, long n, const double *lon,
                     const double *lat, double t, int nz_prof, double *out);

/* update_positions (basemodel/__init__.py:4631-4657); f32 velocities */
void orc_update_positions_f32(long n, double *lon, double *lat, const float *u,
                              const float *v, const int *moving, double dt);
/* same, float64 velocities (advect_wind / stokes_drift / horizontal_diffusion callers) */
void orc_update_positions_f64(long n, double *lon, double *lat, const double *u,
                              const double *v, const int *moving, double dt);

/* advect_ocean_current (physics_methods.py:611-691); scheme 0 euler, 1 rk2, 2 rk4.
 * u_env/v_env = main-loop environment (float32). */
void orc_advect_ocean_current(const orc_world *w, int scheme, long n, double *lon,
                              double *lat, const double *z, const int *moving,
                              const float *cdf, const float *u_env,
                              const float *v_env, double t, double dt, double factor);
/* the same with the uncertainty draws of the Runge-Kutta stage calls (environment.py:869-886 inside
 * physics_methods.py:638-670): stage_noise = [nstage][ncomp][n], ncomp = 2 (normal x, y) or 4 (+ uniform x, y) */
void orc_advect_ocean_current_noise(const orc_world *w, int scheme, long n, double *lon,
                                    double *lat, const double *z, const int *moving,
                                    const float *cdf, const float *u_env,
                                    const float *v_env, double t, double dt, double factor,
                                    int ncomp, const double *stage_noise);

/* advect_wind (physics_methods.py:712-791) */
void orc_advect_wind(long n, double *lon, double *lat, const double *z,
                     const int *moving, const float *wdf, const float *xwind,
                     const float *ywind, const float *u_env, const float *v_env,
                     double wind_drift_depth, int relative_wind, double factor, double dt);

void orc_advect_wind_ef(long n, double *lon, double *lat, const double *z,
                        const int *moving, const float *wdf, const float *xwind,
                        const float *ywind, const float *u_env, const float *v_env,
                        double wind_drift_depth, int relative_wind, double factor, const float *efac, double dt);

/* stokes_drift (physics_methods.py:793-848) with profile 0 monochromatic, 1 exponential, 2 Phillips */
void orc_stokes_drift(long n, double *lon, double *lat, const double *z,
                      const int *moving, const float *sx, const float *sy,
                      const float *hs, const float *tp, const float *xwind,
                      const float *ywind, int hs_mode, int tp_mode, int profile,
                      double factor, double dt);

void orc_stokes_drift_ef(long n, double *lon, double *lat, const double *z,
                         const int *moving, const float *sx, const float *sy,
                         const float *hs, const float *tp, const float *xwind,
                         const float *ywind, int hs_mode, int tp_mode, int profile,
                         double factor, const float *efac, double dt);

/* stokes_drift_profile_windsea_swell (physics_methods.py:418-456): (stokes_u, stokes_v) at depth z */
void orc_stokes_windsea_swell(long n, const double *z, const float *sx, const float *sy,
                              const float *swell_dir, const float *swell_tp, const float *swell_hs,
                              const float *ww_dir, const float *ww_tm, const float *ww_hs,
                              double *out_u, double *out_v);
void orc_stokes_drift_windsea_swell(long n, double *lon, double *lat, const double *z, const int *moving,
                                    const float *sx, const float *sy, const float *swell_dir, const float *swell_tp,
                                    const float *swell_hs, const float *ww_dir, const float *ww_tm, const float *ww_hs,
                                    double factor, double dt);

/* horizontal_diffusion (basemodel/__init__.py:1746-1772), normals drawn by the caller */
void orc_horizontal_diffusion(long n, double *lon, double *lat, const int *moving,
                              const float *D, const double *nx, const double *ny, double dt);

/* OceanDrift.vertical_mixing inner loop (oceandrift.py:480-564), environment diffusivity
 * model; uniforms[i_sub*n + i] = np.random.random draws */
void orc_vertical_mixing(long n, double *z, const int *moving, const float *tv,
                         const float *depth, const float *ssh, int nzp,
                         const double *zp, const double *Kprof, double dt,
                         double dt_mix, int mix_at_surface, const double *uniforms);

/* vertical_advection (oceandrift.py:315-350), no elevation correction */
void orc_vertical_advection(long n, double *z, const int *moving, const float *w,
                            double dt, int at_surface);

/* Leeway.update (models/leeway.py:430-494); capsizing when cap_uniforms != NULL; aux[9][n] = LeewayObj properties in the
 * slot order of include/odrift.h; uniforms = np.random.random(n) of the jibing draw */
void orc_leeway(long n, double *lon, double *lat, const int *moving, float *const *aux,
                const float *xwind, const float *ywind, const float *u, const float *v, double dt,
                double capsize_fraction, const double *uniforms, const double *cap_uniforms,
                double wind_threshold, double wind_threshold_sigma);

/* interact_with_coastline 'stranding' / 'previous' (basemodel/__init__.py:670-746), precision None */
void orc_coastline(long n, int action, float *land, double *lon, double *lat,
                   const double *z, const double *prev_lon, const double *prev_lat,
                   int *status, int *moving, int stranded_code, const float *age_seconds,
                   int seeded_on_land_code);

#ifdef __cplusplus
}
#endif
#endif
