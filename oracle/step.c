/* TEST INFRASTRUCTURE ONLY -- CPU oracle, never on the product path.
 *
 * Restatement of the per-timestep hot path of OpenDrift v1.14.10 in plain C:
 * Environment.get_environment -> advect_ocean_current (Euler/RK2/RK4) ->
 * advect_wind / stokes_drift -> vertical_mixing / vertical_advection ->
 * horizontal_diffusion -> interact_with_coastline.  Batch (all-particle)
 * semantics are kept where the reference is stateful (the in-place NaN dilation
 * of cached blocks, interpolators.py:127-137).  float32/float64 rounding points
 * follow the NumPy dtypes of the reference expressions, cited inline.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const double PI = 3.14159265358979323846264338327950288;
static const double DEG = 3.14159265358979323846264338327950288 / 180.0;

static orc_geod g_wgs84;
static int g_ready = 0;
static const orc_geod *wgs84(void) {
  if (!g_ready) { orc_geod_init(&g_wgs84, 6378137.0, 1 / 298.257223563); g_ready = 1; }
  return &g_wgs84;
}

static double np_mod(double x, double m) { /* numpy.mod: sign follows the divisor */
  double r = fmod(x, m);
  if (r != 0 && ((r < 0) != (m < 0))) r += m;
  return r;
}

/* np.degrees(np.arctan2(x_vel, y_vel)) evaluated in float32
 * (physics_methods.py:629, basemodel/__init__.py:4645).  NumPy's float32 loop
 * for degrees multiplies by float32(180)/float32(pi) = 57.295776f (probed,
 * bit-exact); its float32 arctan2 is an AVX512/SVML kernel on this host that is
 * 1 ulp off the correctly rounded value in ~38 % of the cases (and libm's on
 * other hosts), so the canonical form used here is the correctly rounded one:
 * atan2 in float64 rounded to float32. */
static float azimuth_f32(float xv, float yv) {
  float a = (float)atan2((double)xv, (double)yv);
  return a * (180.0f / 3.14159274101257324f);
}

/* np.sqrt(x*x + y*y) in float32 without contraction */
static float speed_f32(float xv, float yv) {
  volatile float xx = xv * xv, yy = yv * yv;
  volatile float s = xx + yy;
  return sqrtf(s);
}

/* ------------------------------------------------------------------ */
/* update_positions, basemodel/__init__.py:4631-4657                   */
/* ------------------------------------------------------------------ */
void orc_update_positions_f32(long n, double *lon, double *lat, const float *u,
                              const float *v, const int *moving, double dt) {
  long i;
  for (i = 0; i < n; ++i) {
    float az = azimuth_f32(u[i], v[i]);          /* float32 */
    float sp = speed_f32(u[i], v[i]);            /* float32 */
    double vel = (double)sp * (double)moving[i]; /* f32 * int32 array -> float64 */
    double la, lo;
    orc_geod_direct(wgs84(), lat[i], lon[i], (double)az, vel * dt, &la, &lo, 0);
    lon[i] = lo;
    lat[i] = la;
  }
}

void orc_update_positions_f64(long n, double *lon, double *lat, const double *u,
                              const double *v, const int *moving, double dt) {
  long i;
  for (i = 0; i < n; ++i) {
    double az = atan2(u[i], v[i]) * (180.0 / PI); /* np.degrees on float64 */
    double vel = sqrt(u[i] * u[i] + v[i] * v[i]) * (double)moving[i];
    double la, lo;
    orc_geod_direct(wgs84(), lat[i], lon[i], az, vel * dt, &la, &lo, 0);
    lon[i] = lo;
    lat[i] = la;
  }
}

/* ------------------------------------------------------------------ */
/* Reader front door + structured/continuous readers                   */
/* ------------------------------------------------------------------ */

/* rotate_vectors, variables.py:59-109, for a projected reader towards latlong:
 * azimuth of the +y axis by a 10 m finite difference, Geod.inv on WGS84. */
static double rotation_angle(const orc_source *s, double x, double y) {
  double lo1, la1, lo2, la2, az, dist;
  /* delta_y: 10 m along y, 0.1 degree for a CRS that is geographic -- the rotated pole (variables.py:80-83) */
  orc_proj_inv(&s->proj, x, y, &lo1, &la1);
  orc_proj_inv(&s->proj, x, y + (s->proj.kind == ORC_PROJ_OB_TRAN ? 0.1 : 10.0), &lo2, &la2);
  orc_geod_inverse(wgs84(), la1, lo1, la2, lo2, &az, &dist);
  return -(az * DEG); /* rot_angle_rad = -rot_angle_vectors_rad */
}

/* Linear1DInterpolator, interpolators.py:174-197 (interp1d linear on level index) */
static void zinterp(const double *zg, int nz, double z, int *ia, int *ib, double *wa) {
  int asc = zg[1] > zg[0], lo, hi, k;
  double zmin = zg[0], zmax = zg[0], xl, xh, yl, yh, slope, zi;
  for (k = 1; k < nz; ++k) { if (zg[k] < zmin) zmin = zg[k]; if (zg[k] > zmax) zmax = zg[k]; }
  if (z < zmin) z = zmin;
  if (z > zmax) z = zmax;
  /* ascending abscissa xa[j] = asc ? zg[j] : zg[nz-1-j]; ordinate ya[j] = asc ? j : nz-1-j */
  hi = 0;
  while (hi < nz && (asc ? zg[hi] : zg[nz - 1 - hi]) < z) ++hi; /* searchsorted left */
  if (hi < 1) hi = 1;
  if (hi > nz - 1) hi = nz - 1;
  lo = hi - 1;
  xl = asc ? zg[lo] : zg[nz - 1 - lo];
  xh = asc ? zg[hi] : zg[nz - 1 - hi];
  yl = asc ? lo : nz - 1 - lo;
  yh = asc ? hi : nz - 1 - hi;
  slope = (yh - yl) / (xh - xl);
  zi = slope * (z - xl) + yl;
  *ia = (int)(signed char)(long)floor(zi); /* .astype(np.int8) */
  if (*ia < 0) *ia = 0;
  *ib = *ia + 1 < nz - 1 ? *ia + 1 : nz - 1;
  *wa = 1 - (zi - *ia);
}

/* float32 positions of a run's first get_environment (orc_set_position_class) on a geographic reader whose coordinate arrays are
 * float32 (orc_source.xy_f32: bit 0 x, bit 1 y): the reader's x IS the float32 longitude, and the index maps of the 2-D
 * interpolators (interpolators.py:32-37,110-111) are float32 arithmetic -- (x - xgrid[0]) / (xgrid[-1] - xgrid[0]) * (len - 1)
 * with float32 arrays and scalars, the Python int weak.  Set per source call (source_call / orc_get_profile). */
static int g_f32idx = 0;
static double index_f32(double v, double v0, double span, int n) {
  float t = (float)v - (float)v0;
  t = t / (float)span;
  t = t * (float)n;
  return (double)t;
}

static int nearest_index_f(double v, double vmin, double vrange, int n, int f32) {
  /* Nearest2DInterpolator, interpolators.py:32-37 */
  double r = f32 ? (double)rintf((float)index_f32(v, vmin, vrange, n)) : rint((v - vmin) / vrange * n);
  if (!(r >= 0) || r >= n) return n - 1; /* uint32 wrap of negatives, then clip */
  return (int)r;
}

static int nearest_index(double v, double vmin, double vrange, int n) {
  /* Nearest2DInterpolator, interpolators.py:32-37 */
  double r = rint((v - vmin) / vrange * n);
  if (!(r >= 0) || r >= n) return n - 1; /* uint32 wrap of negatives, then clip */
  return (int)r;
}

/* ReaderBlock.interpolate for one variable over all covered particles
 * (interpolation/structured.py:107-163).  out64 receives the value in the dtype
 * class the reference produces: is_f32 = 1 => float32-valued (2D layers),
 * 0 => float64 (3D + z interpolation). */
static void block_interp_one(orc_block *b, int var, long n, const double *x,
                             const double *y, const double *z, double *out64,
                             int *is_f32);

/* ReaderBlock.interpolate for one variable.  Ensemble data (a list of member arrays): every member is interpolated for
 * ALL n positions of the call -- the NaN dilation of a member can therefore be triggered by a position that does not use
 * it -- and position j takes member j % M (readers/interpolation/structured.py:119-135). */
static void block_interp_var(orc_block *b, int var, long n, const double *x,
                             const double *y, const double *z, double *out64,
                             int *is_f32) {
  int M = b->members[var], m;
  if (M <= 1) { block_interp_one(b, var, n, x, y, z, out64, is_f32); return; }
  {
    float *base = b->data[var];
    long per = (long)(b->var_nz[var] > 1 ? b->var_nz[var] : 1) * b->ny * b->nx, j;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n);
    for (m = 0; m < M; ++m) {
      b->data[var] = base + (long)m * per;
      block_interp_one(b, var, n, x, y, z, tmp, is_f32);
      for (j = m; j < n; j += M) out64[j] = tmp[j];
    }
    b->data[var] = base;
    free(tmp);
  }
}

static void block_interp_one(orc_block *b, int var, long n, const double *x,
                             const double *y, const double *z, double *out64,
                             int *is_f32) {
  long i;
  int nzv = b->var_nz[var], k;
  float *data = (float *)b->data[var];
  long plane = (long)b->ny * b->nx;
  if (var == ORC_VAR_LAND) {
    *is_f32 = 1;
    for (i = 0; i < n; ++i) {
      int xi = nearest_index_f(x[i], b->xmin, b->xrange, b->nx, g_f32idx & 1);
      int yi = nearest_index_f(y[i], b->ymin, b->yrange, b->ny, g_f32idx & 2);
      out64[i] = data[(long)yi * b->nx + xi];
    }
    return;
  }
  {
    double *xi = (double *)malloc(sizeof(double) * (size_t)n);
    double *yi = (double *)malloc(sizeof(double) * (size_t)n);
    float *lay = (float *)malloc(sizeof(float) * (size_t)n);
    for (i = 0; i < n; ++i) {
      xi[i] = (g_f32idx & 1) ? index_f32(x[i], b->x0, b->xspan, b->nx - 1) : (x[i] - b->x0) / b->xspan * (b->nx - 1);
      yi[i] = (g_f32idx & 2) ? index_f32(y[i], b->y0, b->yspan, b->ny - 1) : (y[i] - b->y0) / b->yspan * (b->ny - 1);
    }
    if (nzv <= 1) {
      *is_f32 = 1;
      orc_linear2d_call(data, b->ny, b->nx, n, yi, xi, lay);
      for (i = 0; i < n; ++i) out64[i] = lay[i];
    } else {
      double *all = (double *)malloc(sizeof(double) * (size_t)n * (size_t)nzv);
      *is_f32 = 0;
      for (k = 0; k < nzv; ++k) {
        orc_linear2d_call(data + k * plane, b->ny, b->nx, n, yi, xi, lay);
        for (i = 0; i < n; ++i) all[(long)k * n + i] = lay[i];
      }
      for (i = 0; i < n; ++i) {
        int ia, ib;
        double wa;
        zinterp(b->z, b->nz, z[i], &ia, &ib, &wa);
        out64[i] = all[(long)ia * n + i] * wa + all[(long)ib * n + i] * (1 - wa);
      }
      free(all);
    }
    free(xi); free(yi); free(lay);
  }
}

/* time bracket: nearest_time (variables.py:402-443) on the resident levels */
static void bracket(const orc_source *s, double t, int *ib, int *ia) {
  int k, b = 0;
  for (k = 0; k < s->nlevels; ++k) if (s->level[k].t <= t) b = k;
  *ib = b;
  *ia = (b + 1 < s->nlevels && s->level[b].t != t) ? b + 1 : -1;
}

/* one reader.get_variables_interpolated call (variables.py:860-920) for the
 * particles idx[0..m); out[v][j] float64 carrier, NaN = masked / not covered */
/* The reference's element positions are float32 ARRAYS until the first update_positions of a run (elements/elements.py:71-88),
 * so modulate_longitude (variables.py:259-280, called with the elements' lon in get_variables_interpolated :914) forms
 * np.mod(lon + 180, 360) - 180 in float32 in that one get_environment call: the sample longitude is lon rounded to the float32
 * grid of lon + 180.  orc_set_position_class(1) makes the calls that follow do the same (the test harness switches it on for
 * the first main-loop sample of a replay and off again before the Runge-Kutta stage calls, whose positions are float64). */
static int g_f32pos = 0;
void orc_set_position_class(int f32) { g_f32pos = f32; }
static double modulate_longitude(int lon_mode, double lo) {
  if (g_f32pos) {
    float l = (float)lo;
    if (lon_mode == 1) { float a = l + 180.0f; a = (float)np_mod((double)a, 360.0); l = a - 180.0f; }   /* np.mod of float32 is exact */
    else if (lon_mode == 2) l = (float)np_mod((double)l, 360.0);
    return (double)l;
  }
  if (lon_mode == 1) return np_mod(lo + 180, 360) - 180;
  if (lon_mode == 2) return np_mod(lo, 360);
  return lo;
}

static void source_call(const orc_source *s, int nv, const int *vars, long m,
                        const long *idx, const double *lon, const double *lat,
                        const double *z, double t, double **out) {
  double *x = (double *)malloc(sizeof(double) * (size_t)m);
  double *y = (double *)malloc(sizeof(double) * (size_t)m);
  double *zc = (double *)malloc(sizeof(double) * (size_t)m);
  long *cov = (long *)malloc(sizeof(long) * (size_t)m);
  long j, nc = 0;
  int v;
  for (v = 0; v < nv; ++v) for (j = 0; j < m; ++j) out[v][j] = NAN;
  if (!s->always_valid && (t < s->tmin || t > s->tmax)) { /* OutsideTemporalCoverageError, variables.py:730-733 */
    free(x); free(y); free(zc); free(cov);
    return;
  }
  g_f32idx = (g_f32pos && s->proj.kind == ORC_PROJ_LATLONG) ? s->xy_f32 : 0;
  for (j = 0; j < m; ++j) {
    double lo = lon[idx[j]], la = lat[idx[j]], xx, yy, xchk;
    lo = modulate_longitude(s->lon_mode, lo);
    orc_proj_fwd(&s->proj, lo, la, &xx, &yy);
    xchk = xx;
    if (s->proj.kind == ORC_PROJ_LATLONG || s->proj.kind == ORC_PROJ_OB_TRAN) { /* covers_positions_xy re-modulates (crs.is_geographic, variables.py:246) */
      if (s->lon_mode == 1) xchk = np_mod(xx + 180, 360) - 180;
      else if (s->lon_mode == 2) xchk = np_mod(xx, 360);
    }
    if (xchk >= s->xmin && xchk <= s->xmax && yy >= s->ymin && yy <= s->ymax &&
        z[idx[j]] >= s->zmin && z[idx[j]] <= s->zmax) {
      x[nc] = xx; y[nc] = yy; zc[nc] = z[idx[j]]; cov[nc] = j; ++nc;
    }
  }
  if (nc > 0) {
    double **val = (double **)malloc(sizeof(double *) * (size_t)nv);
    int *f32 = (int *)calloc((size_t)nv, sizeof(int));
    for (v = 0; v < nv; ++v) val[v] = (double *)malloc(sizeof(double) * (size_t)nc);
    if (s->kind == ORC_SRC_CONSTANT) {
      /* reader_constant.get_variables, reader_constant.py:60-82 */
      for (v = 0; v < nv; ++v) for (j = 0; j < nc; ++j) val[v][j] = s->const_val[vars[v]];
    } else if (s->kind == ORC_SRC_OSCILLATING) {
      /* reader_oscillating.py:49-59 */
      double phase = ((t - s->params[3]) / s->params[2]) * PI;
      double value = s->params[1] * sin(phase);
      for (v = 0; v < nv; ++v) for (j = 0; j < nc; ++j) val[v][j] = value;
    } else if (s->kind == ORC_SRC_DOUBLE_GYRE) {
      /* reader_double_gyre.get_variables, reader_double_gyre.py:55-79 */
      double A = s->params[0], eps = s->params[1], om = s->params[2];
      double tt = t - s->params[3];
      double a = eps * sin(om * tt), b = 1 - 2 * eps * sin(om * tt);
      for (j = 0; j < nc; ++j) {
        double f = a * x[j] * x[j] + b * x[j], dfdx = 2 * a * x[j] + b;
        for (v = 0; v < nv; ++v) {
          if (vars[v] == ORC_VAR_U)
            val[v][j] = -PI * A * sin(PI * f) * cos(PI * y[j]);
          else if (vars[v] == ORC_VAR_V)
            val[v][j] = PI * A * cos(PI * f) * sin(PI * y[j]) * dfdx;
          else
            val[v][j] = 0; /* land_binary_mask = zeros */
        }
      }
    } else { /* GRID: StructuredReader._get_variables_interpolated_, structured.py:202-400 */
      int ib, ia, all_static = 1;
      double *va = (double *)malloc(sizeof(double) * (size_t)nc);
      orc_source *sm = (orc_source *)s; /* blocks are mutated by the NaN dilation */
      if (s->mod360_x) for (j = 0; j < nc; ++j) x[j] = np_mod(x[j], 360); /* :212-214 */
      bracket(s, t, &ib, &ia);
      for (v = 0; v < nv; ++v)
        if (vars[v] != ORC_VAR_LAND && vars[v] != ORC_VAR_DEPTH) all_static = 0;
      if (all_static) ia = -1; /* :224-229 */
      for (v = 0; v < nv; ++v) {
        int fb = 1, fa = 1;
        block_interp_var(&sm->level[ib], vars[v], nc, x, y, zc, val[v], &fb);
        f32[v] = fb;
        if (ia >= 0 && !s->always_valid) {
          double w = (t - s->level[ib].t) / (s->level[ia].t - s->level[ib].t); /* :353-354 */
          block_interp_var(&sm->level[ia], vars[v], nc, x, y, zc, va, &fa);
          for (j = 0; j < nc; ++j) {
            if (fb && fa) { /* float32 arrays * python floats stay float32 (:362-364) */
              volatile float p = (float)val[v][j] * (float)(1 - w);
              volatile float q = (float)va[j] * (float)w;
              val[v][j] = (float)(p + q);
            } else {
              val[v][j] = val[v][j] * (1 - w) + va[j] * w;
            }
          }
        }
      }
      free(va);
    }
    /* rotate vector pairs to lon/lat CRS (variables.py:799-837) */
    if (s->proj.kind != ORC_PROJ_LATLONG) {
      static const int pairs[4][2] = {{ORC_VAR_XWIND, ORC_VAR_YWIND},        /* vector_pairs_xy, basereader/consts.py:27-36 */
                                      {ORC_VAR_ICE_U, ORC_VAR_ICE_V},
                                      {ORC_VAR_U, ORC_VAR_V},
                                      {ORC_VAR_STOKES_X, ORC_VAR_STOKES_Y}};
      int p;
      double *rot = NULL;
      for (p = 0; p < 4; ++p) {
        int iu = -1, iv = -1;
        for (v = 0; v < nv; ++v) { if (vars[v] == pairs[p][0]) iu = v; if (vars[v] == pairs[p][1]) iv = v; }
        if (iu < 0 || iv < 0) continue;
        if (!rot) {
          rot = (double *)malloc(sizeof(double) * (size_t)nc);
          for (j = 0; j < nc; ++j) rot[j] = rotation_angle(s, x[j], y[j]);
        }
        for (j = 0; j < nc; ++j) {
          double uu = val[iu][j], vv = val[iv][j], c = cos(rot[j]), sn = sin(rot[j]);
          val[iu][j] = uu * c - vv * sn;
          val[iv][j] = uu * sn + vv * c;
        }
      }
      free(rot);
    }
    for (v = 0; v < nv; ++v) {
      for (j = 0; j < nc; ++j) out[v][cov[j]] = val[v][j];
      free(val[v]);
    }
    free(val); free(f32);
  }
  free(x); free(y); free(zc); free(cov);
}

/* Environment.get_environment, environment.py:499-923 (no lazy readers, no noise) */
void orc_get_environment(const orc_world *w, int nv, const int *vars, long n,
                         const double *lon, const double *lat, const double *z,
                         double t, float *const *out) {
  int done[ORC_NVAR] = {0}, v, u, k;
  long i;
  for (v = 0; v < nv; ++v) /* fallback pre-fill, :592-595 */
    for (i = 0; i < n; ++i) out[v][i] = w->fallback[vars[v]];
  for (v = 0; v < nv; ++v) {
    int gv[ORC_NVAR], gi[ORC_NVAR], ng = 0;
    long *miss, nm;
    double **tmp;
    if (done[v]) continue;
    /* variable group = variables sharing the same reader list (get_reader_groups :339-374) */
    for (u = v; u < nv; ++u) {
      int same = w->nlist[vars[u]] == w->nlist[vars[v]];
      for (k = 0; same && k < w->nlist[vars[v]]; ++k)
        if (w->list[vars[u]][k] != w->list[vars[v]][k]) same = 0;
      if (same && !done[u]) { gv[ng] = vars[u]; gi[ng] = u; ++ng; done[u] = 1; }
    }
    if (w->nlist[vars[v]] == 0) continue;
    miss = (long *)malloc(sizeof(long) * (size_t)n);
    tmp = (double **)malloc(sizeof(double *) * (size_t)ng);
    for (u = 0; u < ng; ++u) tmp[u] = (double *)malloc(sizeof(double) * (size_t)n);
    nm = n;
    for (i = 0; i < n; ++i) miss[i] = i;
    for (k = 0; k < w->nlist[vars[v]] && nm > 0; ++k) {
      const orc_source *s = &w->src[w->list[vars[v]][k]];
      long j, nm2 = 0;
      source_call(s, ng, gv, nm, miss, lon, lat, z, t, tmp);
      for (j = 0; j < nm; ++j) {
        int bad = 0;
        for (u = 0; u < ng; ++u) {
          out[gi[u]][miss[j]] = (float)tmp[u][j]; /* masked_invalid(...).astype('float32') :695-696 */
          if (!isfinite(tmp[u][j])) bad = 1;      /* combined_mask :727-746 */
        }
        if (bad) miss[nm2++] = miss[j];
      }
      nm = nm2;
    }
    for (u = 0; u < ng; ++u) free(tmp[u]);
    free(tmp); free(miss);
  }
  for (v = 0; v < nv; ++v) /* fallback for masked, :782-791 */
    if (isfinite(w->fallback[vars[v]]))
      for (i = 0; i < n; ++i) if (!isfinite(out[v][i])) out[v][i] = w->fallback[vars[v]];
  /* "Some extra checks of units" (:829-838): temperatures above 100 are Kelvin and become Celsius.  env is a MASKED
   * float32 recarray at that point: numpy.ma wraps the Python float into a 0-d float64 array, which is not a weak
   * scalar, so the difference is formed in float64 and rounded to float32 by the assignment (golden c12) */
  for (v = 0; v < nv; ++v)
    if (vars[v] == ORC_VAR_TEMP)
      for (i = 0; i < n; ++i)
        if (out[v][i] > 100.f) out[v][i] = (float)((double)out[v][i] - 273.15);
}

/* Profiles of one variable from the first GRID source of its priority list:
 * all block layers, horizontal interpolation only, time interpolation in
 * float64 (structured.py:366-385).  Particles the source does not cover get the
 * fallback value.  out[k*n + i]. */
void orc_get_profile(const orc_world *w, int var, long n, const double *lon,
                     const double *lat, double t, int nz_prof, double *out) {
  const orc_source *s = 0;
  int k, ib, ia;
  long i;
  for (k = 0; k < w->nlist[var]; ++k)
    if (w->src[w->list[var][k]].kind == ORC_SRC_GRID) { s = &w->src[w->list[var][k]]; break; }
  for (i = 0; i < (long)nz_prof * n; ++i) out[i] = w->fallback[var];
  if (!s) return;
  bracket(s, t, &ib, &ia);
  {
    orc_block *bb = (orc_block *)&s->level[ib], *ba = ia >= 0 ? (orc_block *)&s->level[ia] : 0;
    long plane = (long)bb->ny * bb->nx;
    double wgt = ba ? (t - bb->t) / (ba->t - bb->t) : 0;
    double *xi = (double *)malloc(sizeof(double) * (size_t)n);
    double *yi = (double *)malloc(sizeof(double) * (size_t)n);
    float *l0 = (float *)malloc(sizeof(float) * (size_t)n);
    float *l1 = (float *)malloc(sizeof(float) * (size_t)n);
    char *cov = (char *)malloc((size_t)n);
    for (i = 0; i < n; ++i) {
      double lo = lon[i], xx, yy;
      lo = modulate_longitude(s->lon_mode, lo);
      orc_proj_fwd(&s->proj, lo, lat[i], &xx, &yy);
      cov[i] = xx >= s->xmin && xx <= s->xmax && yy >= s->ymin && yy <= s->ymax;
      if (s->mod360_x) xx = np_mod(xx, 360);
      {
        const int f32idx = (g_f32pos && s->proj.kind == ORC_PROJ_LATLONG) ? s->xy_f32 : 0;   /* (see index_f32) */
        xi[i] = (f32idx & 1) ? index_f32(xx, bb->x0, bb->xspan, bb->nx - 1) : (xx - bb->x0) / bb->xspan * (bb->nx - 1);
        yi[i] = (f32idx & 2) ? index_f32(yy, bb->y0, bb->yspan, bb->ny - 1) : (yy - bb->y0) / bb->yspan * (bb->ny - 1);
      }
    }
    {
      /* ensemble data: position j of the call takes the COLUMN of member j % M (readers/interpolation/structured.py:119-135:
       * `horizontal[:, elnum] = int_full[:, elnum]`), the numbering of its element values */
      int M = bb->members[var] > 1 ? bb->members[var] : 1, m;
      long per = (long)(bb->var_nz[var] > 1 ? bb->var_nz[var] : 1) * plane;
      for (m = 0; m < M; ++m)
        for (k = 0; k < nz_prof && k < bb->var_nz[var]; ++k) {
          orc_linear2d_call((float *)bb->data[var] + m * per + k * plane, bb->ny, bb->nx, n, yi, xi, l0);
          if (ba) orc_linear2d_call((float *)ba->data[var] + m * per + k * plane, ba->ny, ba->nx, n, yi, xi, l1);
          for (i = m; i < n; i += M) {
            double val = ba ? (double)l0[i] * (1 - wgt) + (double)l1[i] * wgt : (double)l0[i];
            if (cov[i] && isfinite(val)) out[(long)k * n + i] = val;
          }
        }
    }
    free(xi); free(yi); free(l0); free(l1); free(cov);
  }
}

/* ------------------------------------------------------------------ */
/* advect_ocean_current, physics_methods.py:611-691                    */
/* ------------------------------------------------------------------ */
static void stage_positions(long n, const double *lon, const double *lat,
                            const float *u, const float *v, double dt_half,
                            double *lon2, double *lat2) {
  long i;
  for (i = 0; i < n; ++i) {
    float az = azimuth_f32(u[i], v[i]);
    float sp = speed_f32(u[i], v[i]);
    /* dist = speed*dt*.5 stays float32 (f32 array * python floats, :631) */
    volatile float d1 = sp * (float)(dt_half * 2);
    volatile float dist = d1 * 0.5f;
    orc_geod_direct(wgs84(), lat[i], lon[i], (double)az, (double)dist, &lat2[i], &lon2[i], 0);
  }
}

/* drift:current_uncertainty / drift:current_uncertainty_uniform (environment.py:869-886) are part of EVERY
 * get_environment call whose variables hold the current -- also of the Runge-Kutta stage calls
 * (physics_methods.py:638-670): env[var] += draw, a float32 array += float64 array, i.e. float32(float64(u) + draw),
 * first the normal pair (x, y), then the uniform pair.  noise = [ncomp][n] draws of one call in np.random order. */
static void add_uncertainty(long n, float *u, float *v, int ncomp, const double *noise) {
  long i;
  int c;
  for (c = 0; c + 1 < ncomp; c += 2)
    for (i = 0; i < n; ++i) {
      u[i] = (float)((double)u[i] + noise[(size_t)c * (size_t)n + (size_t)i]);
      v[i] = (float)((double)v[i] + noise[(size_t)(c + 1) * (size_t)n + (size_t)i]);
    }
}

void orc_advect_ocean_current(const orc_world *w, int scheme, long n, double *lon,
                              double *lat, const double *z, const int *moving,
                              const float *cdf, const float *u_env,
                              const float *v_env, double t, double dt, double factor) {
  orc_advect_ocean_current_noise(w, scheme, n, lon, lat, z, moving, cdf, u_env, v_env, t, dt, factor, 0, NULL);
}

/* stage_noise: [nstage][ncomp][n] (nstage = 1 for RK2, 3 for RK4; ncomp = 2 or 4), or NULL */
void orc_advect_ocean_current_noise(const orc_world *w, int scheme, long n, double *lon,
                                    double *lat, const double *z, const int *moving,
                                    const float *cdf, const float *u_env,
                                    const float *v_env, double t, double dt, double factor,
                                    int ncomp, const double *stage_noise) {
  static const int uv[2] = {ORC_VAR_U, ORC_VAR_V};
  const size_t per = (size_t)ncomp * (size_t)n;
  float *fu = (float *)malloc(sizeof(float) * (size_t)n);
  float *fv = (float *)malloc(sizeof(float) * (size_t)n);
  long i;
  if (scheme == 0) {
    for (i = 0; i < n; ++i) { /* factor*cdf float32, times float32 env (:686-688) */
      float f = (float)factor * cdf[i];
      fu[i] = f * u_env[i];
      fv[i] = f * v_env[i];
    }
  } else {
    double *lon2 = (double *)malloc(sizeof(double) * (size_t)n);
    double *lat2 = (double *)malloc(sizeof(double) * (size_t)n);
    float *u2 = (float *)malloc(sizeof(float) * (size_t)n), *v2 = (float *)malloc(sizeof(float) * (size_t)n);
    float *o2[2];
    o2[0] = u2; o2[1] = v2;
    stage_positions(n, lon, lat, u_env, v_env, dt * .5, lon2, lat2);
    orc_get_environment(w, 2, uv, n, lon2, lat2, z, t + dt / 2, o2);
    if (stage_noise) add_uncertainty(n, u2, v2, ncomp, stage_noise);
    if (scheme == 1) {
      for (i = 0; i < n; ++i) {
        float f = (float)factor * cdf[i];
        fu[i] = f * u2[i];
        fv[i] = f * v2[i];
      }
    } else {
      float *u3 = (float *)malloc(sizeof(float) * (size_t)n), *v3 = (float *)malloc(sizeof(float) * (size_t)n);
      float *u4 = (float *)malloc(sizeof(float) * (size_t)n), *v4 = (float *)malloc(sizeof(float) * (size_t)n);
      float *o3[2], *o4[2];
      o3[0] = u3; o3[1] = v3; o4[0] = u4; o4[1] = v4;
      stage_positions(n, lon, lat, u2, v2, dt * .5, lon2, lat2);
      orc_get_environment(w, 2, uv, n, lon2, lat2, z, t + dt / 2, o3);
      if (stage_noise) add_uncertainty(n, u3, v3, ncomp, stage_noise + per);
      stage_positions(n, lon, lat, u3, v3, dt * .5, lon2, lat2); /* dt*.5 again: reference quirk :662 */
      orc_get_environment(w, 2, uv, n, lon2, lat2, z, t + dt, o4);
      if (stage_noise) add_uncertainty(n, u4, v4, ncomp, stage_noise + 2 * per);
      for (i = 0; i < n; ++i) { /* (x_vel + 2*x_vel2 + 2*x_vel3 + x_vel4)/6.0 in float32 (:674-675) */
        volatile float a2 = 2 * u2[i], a3 = 2 * u3[i], b2 = 2 * v2[i], b3 = 2 * v3[i];
        volatile float su = u_env[i] + a2, sv = v_env[i] + b2;
        float f = (float)factor * cdf[i];
        su = su + a3; sv = sv + b3;
        su = su + u4[i]; sv = sv + v4[i];
        su = su / 6.0f; sv = sv / 6.0f;
        fu[i] = su * f;
        fv[i] = sv * f;
      }
      free(u3); free(v3); free(u4); free(v4);
    }
    free(lon2); free(lat2); free(u2); free(v2);
  }
  orc_update_positions_f32(n, lon, lat, fu, fv, moving, dt);
  free(fu); free(fv);
}

/* ------------------------------------------------------------------ */
/* advect_wind, physics_methods.py:712-791                             */
/* ------------------------------------------------------------------ */
void orc_advect_wind(long n, double *lon, double *lat, const double *z,
                     const int *moving, const float *wdf_in, const float *xwind,
                     const float *ywind, const float *u_env, const float *v_env,
                     double wind_drift_depth, int relative_wind, double factor, double dt) {
  orc_advect_wind_ef(n, lon, lat, z, moving, wdf_in, xwind, ywind, u_env, v_env, wind_drift_depth, relative_wind,
                     factor, NULL, dt);
}

/* the same with a per-element float32 factor (OpenOil.advect_oil in ice: factor = 1 - k_ice, openoil.py:1210);
 * efac == NULL: the scalar factor */
void orc_advect_wind_ef(long n, double *lon, double *lat, const double *z,
                        const int *moving, const float *wdf_in, const float *xwind,
                        const float *ywind, const float *u_env, const float *v_env,
                        double wind_drift_depth, int relative_wind, double factor, const float *efac, double dt) {
  double *xu = (double *)malloc(sizeof(double) * (size_t)n);
  double *xv = (double *)malloc(sizeof(double) * (size_t)n);
  double wdd = fabs(wind_drift_depth), wmax = 0, smax = 0;
  int surface_only = wind_drift_depth == 0, any = 0;
  long i;
  for (i = 0; i < n; ++i) {
    int surf = z[i] >= -wdd;
    double wdf = wdf_in[i];
    float xw = xwind[i], yw = ywind[i];
    if (!surface_only) {
      wdf = wdf * (wdd + z[i]) / wdd;       /* float64: f32 * f64 array (:756) */
      if (z[i] > 0) wdf = wdf_in[i];        /* elements in air (:758) */
    }
    if (!surf) wdf = 0.0;
    if (relative_wind) { xw = xw - u_env[i]; yw = yw - v_env[i]; } /* float32 (:769-770) */
    if (surf) {
      float sp = speed_f32(xw, yw);
      any = 1;
      if (wdf > wmax) wmax = wdf;
      if (sp > smax) smax = sp;
    }
    xu[i] = (double)xw * wdf * (efac ? (double)efac[i] : factor);   /* x_wind*wdf*factor (:791): f64 * f32 array */
    xv[i] = (double)yw * wdf * (efac ? (double)efac[i] : factor);
  }
  /* early returns (:741-747, :775-780) */
  if (any && wmax != 0 && smax != 0) orc_update_positions_f64(n, lon, lat, xu, xv, moving, dt);
  free(xu); free(xv);
}

/* ------------------------------------------------------------------ */
/* stokes_drift, physics_methods.py:793-848 + profiles :336-416        */
/* ------------------------------------------------------------------ */
/* NumPy dtype classes of an operand: python scalar (weak), float32 array, float64 array */
enum { K_WEAK = 0, K_F32 = 1, K_F64 = 2 };
typedef struct { double v; int k; } tval;
static tval tmul(tval a, tval b) {
  tval r; r.k = a.k > b.k ? a.k : b.k;
  if (r.k == K_F32) { volatile float q = (float)a.v * (float)b.v; r.v = q; } else r.v = a.v * b.v;
  return r;
}
static tval tdiv(tval a, tval b) {
  tval r; r.k = a.k > b.k ? a.k : b.k;
  if (r.k == K_F32) { volatile float q = (float)a.v / (float)b.v; r.v = q; } else r.v = a.v / b.v;
  return r;
}
static tval tv_(double v, int k) { tval r; r.v = v; r.k = k; return r; }

/* hs_mode / tp_mode: 0 = environment array (float32), 1 = parameterised from wind
 * (significant_wave_height :893-906 float32; wave_period :918-933 float64),
 * 2 = python scalar 1 / 8 (:809-814). */
void orc_stokes_drift(long n, double *lon, double *lat, const double *z,
                      const int *moving, const float *sx, const float *sy,
                      const float *hs_in, const float *tp_in, const float *xwind,
                      const float *ywind, int hs_mode, int tp_mode, int profile,
                      double factor, double dt) {
  orc_stokes_drift_ef(n, lon, lat, z, moving, sx, sy, hs_in, tp_in, xwind, ywind, hs_mode, tp_mode, profile, factor, NULL, dt);
}

/* the same with a per-element float32 factor (OpenOil.advect_oil in ice: factor_stokes, openoil.py:1200-1213) and
 * tp_mode 3 = parameterised from the wind, then read back from the float32 environment (a model that has the wave
 * period among its variables: calculate_missing_environment_variables, physics_methods.py:876-883) */
/* one profile function of physics_methods.py:336-416 for one element: surface components (float32), wave height and
 * period with their NumPy dtype classes, depth -> (stokes_u, stokes_v) float64 */
static void stokes_profile(int profile, float sxs, float sys, tval H, tval T, double zz, double *su, double *sv) {
  float speed = speed_f32(sxs, sys); /* float32 */
  tval mwf, pw, transport, num, km;
  double unit, az = fabs(zz);
  mwf = tdiv(tv_(2. * PI, K_WEAK), T);               /* stokes_transport_monochromatic :332-334 */
  pw = tmul(H, H);                                   /* np.power(H, 2) */
  transport = tdiv(tmul(mwf, pw), tv_(16, K_WEAK));
  num = tv_(speed, K_F32);
  if (profile == 2) num = tmul(num, tv_(1 - 2 * 1.0 / 3, K_WEAK)); /* (1-2*beta/3) */
  km = tdiv(num, tmul(tv_(2, K_WEAK), transport));
  if (profile == 0) unit = exp(tmul(tv_(2, K_WEAK), km).v * zz);
  else if (profile == 1) {
    tval ke = tdiv(km, tv_(3, K_WEAK));
    unit = exp(tmul(tv_(2.0, K_WEAK), ke).v * zz) / (1.0 - tmul(tv_(8.0, K_WEAK), ke).v * zz);
  } else {
    double k2 = tmul(tv_(2, K_WEAK), km).v, c2 = tmul(tv_(2 * PI, K_WEAK), km).v;
    unit = exp(k2 * zz) - 1 * sqrt(c2 * az) * erfc(sqrt(k2 * az));
  }
  *su = speed == 0 ? 0 : (double)sxs * unit;
  *sv = speed == 0 ? 0 : (double)sys * unit;
}

/* stokes_drift_profile_windsea_swell (physics_methods.py:418-456; Breivik & Christensen 2020): the surface Stokes drift
 * split into a swell part along the swell direction (monochromatic profile with the swell height / period) and a
 * wind-sea part (the rest; Phillips profile with the wind-sea height / period).  All inputs float32 environment arrays:
 * the unit vectors and the split are float32 arithmetic, the profiles float64. */
void orc_stokes_windsea_swell(long n, const double *z, const float *sx, const float *sy,
                              const float *swell_dir, const float *swell_tp, const float *swell_hs,
                              const float *ww_dir, const float *ww_tm, const float *ww_hs,
                              double *out_u, double *out_v) {
  long i;
  for (i = 0; i < n; ++i) {
    volatile float rws = ww_dir[i] * (float)(PI / 180.), rsw = swell_dir[i] * (float)(PI / 180.);   /* np.radians, float32 */
    /* float32 cos / sin as the rounded float64 functions (correctly rounded float32 but for ~1e-9 of the arguments) */
    float ws_n = (float)cos((double)rws), ws_e = (float)sin((double)rws), sw_n = (float)cos((double)rsw), sw_e = (float)sin((double)rsw);
    volatile float a1 = sx[i] * ws_n, a2 = sy[i] * ws_e, numr = a1 - a2;
    volatile float d1 = sw_e * ws_n, d2 = sw_n * ws_e, den = d1 - d2;
    volatile float sp = numr / den;
    volatile float swu = sp * sw_e, swv = sp * sw_n;
    volatile float wu = sx[i] - swu, wv = sy[i] - swv;
    double u1, v1, u2, v2;
    stokes_profile(0, swu, swv, tv_(swell_hs[i], K_F32), tv_(swell_tp[i], K_F32), z[i], &u1, &v1);
    stokes_profile(2, wu, wv, tv_(ww_hs[i], K_F32), tv_(ww_tm[i], K_F32), z[i], &u2, &v2);
    out_u[i] = u1 + u2;
    out_v[i] = v1 + v2;
  }
}

/* stokes_drift with drift:stokes_drift_profile = 'windsea_swell' (physics_methods.py:793-848) */
void orc_stokes_drift_windsea_swell(long n, double *lon, double *lat, const double *z, const int *moving,
                                    const float *sx, const float *sy, const float *swell_dir, const float *swell_tp,
                                    const float *swell_hs, const float *ww_dir, const float *ww_tm, const float *ww_hs,
                                    double factor, double dt) {
  double *su = (double *)malloc(sizeof(double) * (size_t)n);
  double *sv = (double *)malloc(sizeof(double) * (size_t)n);
  float mx = -INFINITY;
  long i;
  for (i = 0; i < n; ++i) { volatile float s = sx[i] + sy[i]; if (s > mx) mx = s; }
  if (n == 0 || mx == 0) { free(su); free(sv); return; } /* "No Stokes drift velocity available" */
  orc_stokes_windsea_swell(n, z, sx, sy, swell_dir, swell_tp, swell_hs, ww_dir, ww_tm, ww_hs, su, sv);
  for (i = 0; i < n; ++i) { su[i] *= factor; sv[i] *= factor; }
  orc_update_positions_f64(n, lon, lat, su, sv, moving, dt);
  free(su); free(sv);
}

void orc_stokes_drift_ef(long n, double *lon, double *lat, const double *z,
                         const int *moving, const float *sx, const float *sy,
                         const float *hs_in, const float *tp_in, const float *xwind,
                         const float *ywind, int hs_mode, int tp_mode, int profile,
                         double factor, const float *efac, double dt) {
  double *su = (double *)malloc(sizeof(double) * (size_t)n);
  double *sv = (double *)malloc(sizeof(double) * (size_t)n);
  float mx = -INFINITY;
  long i;
  for (i = 0; i < n; ++i) { volatile float s = sx[i] + sy[i]; if (s > mx) mx = s; }
  if (n == 0 || mx == 0) { free(su); free(sv); return; } /* "No Stokes drift velocity available" */
  for (i = 0; i < n; ++i) {
    float ws = (hs_mode == 1 || tp_mode == 1 || tp_mode == 3) ? speed_f32(xwind[i], ywind[i]) : 0.f;
    tval H, T;
    double u, v, f = efac ? (double)efac[i] : factor;
    if (hs_mode == 0) H = tv_(hs_in[i], K_F32);
    else if (hs_mode == 1) { volatile float w2 = ws * ws; volatile float h = (float)0.0246 * w2; H = tv_(h, K_F32); }
    else H = tv_(1, K_WEAK);
    if (tp_mode == 0) T = tv_(tp_in[i], K_F32);
    else if (tp_mode == 1 || tp_mode == 3) {
      double omega = 5;
      if (ws > 0) { volatile float d = (float)1.17 * ws; volatile float o = (float)(0.877 * 9.81) / d; omega = o; }
      T = tv_((2 * PI) / omega, K_F64);
      if (tp_mode == 3) { volatile float tf = (float)T.v; T = tv_(tf, K_F32); }
    } else T = tv_(8, K_WEAK);
    stokes_profile(profile, sx[i], sy[i], H, T, z[i], &u, &v);
    su[i] = u * f;      /* stokes_u*factor (:843) */
    sv[i] = v * f;
  }
  orc_update_positions_f64(n, lon, lat, su, sv, moving, dt);
  free(su); free(sv);
}

/* ------------------------------------------------------------------ */
/* horizontal_diffusion, basemodel/__init__.py:1746-1772               */
/* ------------------------------------------------------------------ */
void orc_horizontal_diffusion(long n, double *lon, double *lat, const int *moving,
                              const float *D, const double *nx, const double *ny, double dt) {
  double *xu = (double *)malloc(sizeof(double) * (size_t)n);
  double *xv = (double *)malloc(sizeof(double) * (size_t)n);
  float dmax = 0;
  long i;
  double adt = fabs(dt);
  for (i = 0; i < n; ++i) if (D[i] > dmax) dmax = D[i];
  if (n == 0 || dmax == 0) { free(xu); free(xv); return; }
  for (i = 0; i < n; ++i) {
    volatile float twoD = 2 * D[i];
    volatile float q = twoD / (float)adt; /* float32 (f32 array / python float) */
    float s = sqrtf(q);
    xu[i] = (double)moving[i] * (double)s * nx[i]; /* int32 * f32 -> f64 */
    xv[i] = (double)moving[i] * (double)s * ny[i];
  }
  orc_update_positions_f64(n, lon, lat, xu, xv, moving, dt);
  free(xu); free(xv);
}

/* ------------------------------------------------------------------ */
/* OceanDrift.vertical_mixing, oceandrift.py:397-571 (environment model) */
/* ------------------------------------------------------------------ */
static void np_gradient_axis0(const double *K, const double *zp, int nz, long n, double *g) {
  /* numpy.gradient(K, zp, axis=0), edge_order=1 */
  int k, uniform = 1;
  long i;
  for (k = 1; k < nz - 1; ++k) if ((zp[k + 1] - zp[k]) != (zp[1] - zp[0])) uniform = 0;
  for (i = 0; i < n; ++i) {
    if (nz < 2) { g[i] = 0; continue; }
    for (k = 1; k < nz - 1; ++k) {
      if (uniform) {
        g[(long)k * n + i] = (K[(long)(k + 1) * n + i] - K[(long)(k - 1) * n + i]) / (2. * (zp[1] - zp[0]));
      } else {
        double dx1 = zp[k] - zp[k - 1], dx2 = zp[k + 1] - zp[k];
        double a = -(dx2) / (dx1 * (dx1 + dx2)), b = (dx2 - dx1) / (dx1 * dx2), c = dx1 / (dx2 * (dx1 + dx2));
        g[(long)k * n + i] = a * K[(long)(k - 1) * n + i] + b * K[(long)k * n + i] + c * K[(long)(k + 1) * n + i];
      }
    }
    g[i] = (K[n + i] - K[i]) / (zp[1] - zp[0]);
    g[(long)(nz - 1) * n + i] = (K[(long)(nz - 1) * n + i] - K[(long)(nz - 2) * n + i]) / (zp[nz - 1] - zp[nz - 2]);
  }
}

void orc_vertical_mixing(long n, double *z, const int *moving, const float *tv,
                         const float *depth, const float *ssh, int nzp,
                         const double *zp, const double *Kprof, double dt,
                         double dt_mix_cfg, int mix_at_surface, const double *uniforms) {
  double dt_mix = dt_mix_cfg * (dt > 0 ? 1 : (dt < 0 ? -1 : 0));
  int ntimes = abs((int)(dt / dt_mix)), it, k;
  double *gradK = (double *)malloc(sizeof(double) * (size_t)n * (size_t)(nzp > 0 ? nzp : 1));
  const double r = 1.0 / 3;
  long i;
  np_gradient_axis0(Kprof, zp, nzp, n, gradK);
  for (i = 0; i < (long)nzp * n; ++i) { gradK[i] = -gradK[i]; if (fabs(gradK[i]) < 1e-10) gradK[i] = 0; }
  for (it = 0; it < ntimes; ++it) {
    for (i = 0; i < n; ++i) {
      int surface = z[i] == 0, zi;
      volatile float zsum = depth[i] + ssh[i];
      float Zmin = -1.f * zsum; /* float32 (:408) */
      double zz = z[i], d = -zz, idx, Kz, dK, R, w;
      /* z_index = interp1d(-mixing_z, range, fill_value=(0, nz-1)) (:485-488) */
      if (nzp == 1) idx = 0;
      else if (d < -zp[0]) idx = 0;
      else if (d > -zp[nzp - 1]) idx = nzp - 1;
      else {
        int hi = 0;
        while (hi < nzp && -zp[hi] < d) ++hi;
        if (hi < 1) hi = 1;
        if (hi > nzp - 1) hi = nzp - 1;
        { double xl = -zp[hi - 1], xh = -zp[hi]; idx = (1.0 / (xh - xl)) * (d - xl) + (hi - 1); }
      }
      zi = (int)(unsigned short)(long)rint(idx); /* np.round(...).astype(np.uint16) */
      Kz = Kprof[(long)zi * n + i];
      dK = gradK[(long)zi * n + i];
      R = 2 * uniforms[(long)it * n + i] - 1;
      zz = zz - moving[i] * (dK * dt_mix - R * sqrt((Kz * fabs(dt_mix) * 2 / r)));
      if (zz >= 0) zz = -zz;                                      /* reflect from surface */
      if (zz < Zmin && moving[i] == 1) zz = 2 * Zmin - zz;        /* reflect from seafloor; 2*Zmin float32 */
      /* w*dt_mix: dt_mix = timestep * np.sign(...) is a NumPy float64 SCALAR (oceandrift.py:416), so under NumPy 2
       * (NEP 50; the golden vectors were written with NumPy 2.2) float32 array * float64 scalar is a float64 product
       * (NumPy 1.x value-based casting would round it to float32: 2e-9 m per sub-step) */
      w = (double)tv[i] * dt_mix * moving[i];
      zz = zz + w;
      if (!mix_at_surface && surface) zz = 0.;
      if (zz > 0) zz = 0;                                         /* surface_stick */
      if (zz < Zmin) zz = Zmin;                                   /* interact_with_seafloor: lift_to_seafloor */
      z[i] = zz;
    }
  }
  (void)k;
  free(gradK);
}

/* vertical_advection, oceandrift.py:315-350 */
void orc_vertical_advection(long n, double *z, const int *moving, const float *w,
                            double dt, int at_surface) {
  long i;
  for (i = 0; i < n; ++i) {
    if (at_surface ? z[i] <= 0 : z[i] < 0) {
      double zz = z[i] + (double)moving[i] * (double)w[i] * dt;
      z[i] = zz < 0 ? zz : 0;
    }
  }
}

/* interact_with_coastline, basemodel/__init__.py:670-746 (precision None);
 * action 1 = stranding, 2 = previous */
void orc_coastline(long n, int action, float *land, double *lon, double *lat,
                   const double *z, const double *prev_lon, const double *prev_lat,
                   int *status, int *moving, int stranded_code, const float *age_seconds,
                   int seeded_on_land_code) {
  long i;
  for (i = 0; i < n; ++i) {
    if (land[i] != 1) continue;
    if (action == 1) {
      if (z[i] <= 0) { /* deactivate_elements(..., reason='stranded') :1774-1795 */
        if (status[i] == 0) status[i] = stranded_code;
        moving[i] = 0;
      }
    } else if (action == 2) {
      if (seeded_on_land_code > 0 && age_seconds && age_seconds[i] == 0) { /* :715-719 */
        if (status[i] == 0) status[i] = seeded_on_land_code;
        moving[i] = 0;
      }
      lon[i] = prev_lon[i]; /* on_land includes the elements just deactivated (:720-730) */
      lat[i] = prev_lat[i];
      land[i] = 0; /* self.environment.land_binary_mask[on_land] = 0 (:746) */
    }
  }
}

/* ------------------------------------------------------------------ */
/* Leeway.update, models/leeway.py:430-494.  processes:capsizing (:438-455): cap_uniforms != NULL     */
/* holds np.random.rand(len(can_be_capsized)) in element order; forward runs capsize (0 -> 1), backward */
/* runs un-capsize (1 -> 0)                                                                            */
/* ------------------------------------------------------------------ */
void orc_leeway(long n, double *lon, double *lat, const int *moving, float *const *aux,
                const float *xwind, const float *ywind, const float *u, const float *v, double dt,
                double capsize_fraction, const double *uniforms, const double *cap_uniforms,
                double wind_threshold, double wind_threshold_sigma) {
  float *xl = (float *)malloc(sizeof(float) * (size_t)n), *yl = (float *)malloc(sizeof(float) * (size_t)n);
  long i;
  if (cap_uniforms) {
    const float from = dt >= 0 ? 0.0f : 1.0f;   /* simulation_direction() (basemodel/__init__.py:4524-4529) */
    const double scale = fabs(dt) / 3600;        /* python floats */
    long j = 0;
    for (i = 0; i < n; ++i) {
      if (aux[8][i] != from) continue;
      {
        float ws = speed_f32(xwind[i], ywind[i]);
        volatile float a = ws - (float)wind_threshold;     /* float32 array with python scalars: float32 */
        volatile float b = a / (float)wind_threshold_sigma;
        volatile float th = (float)tanh((double)b);
        volatile float c = 0.5f * th;
        volatile float pr = 0.5f + c;
        volatile float prob = pr * (float)scale;
        if (cap_uniforms[j] < (double)prob) aux[8][i] = 1.0f - aux[8][i];
        ++j;
      }
    }
  }
  for (i = 0; i < n; ++i) {
    float ws = speed_f32(xwind[i], ywind[i]);
    float wd = (float)atan2((double)xwind[i], (double)ywind[i]);
    volatile float a = aux[4][i] / 20.0f, b = aux[5][i] / 20.0f, ha = aux[4][i] / 2.0f, hb = aux[5][i] / 2.0f;
    volatile float dw = aux[0][i] + a, cw = aux[1][i] + b;
    volatile float sn = (float)sin((double)wd), cs = (float)cos((double)wd), t1, t2, t3, t4;
    dw = dw * ws; dw = dw + aux[2][i]; dw = dw + ha; dw = dw * (float).01;
    cw = cw * ws; cw = cw + aux[3][i]; cw = cw + hb; cw = cw * (float).01;
    t1 = dw * cs; t2 = cw * sn; t3 = -dw * sn; t4 = cw * cs;
    yl[i] = t1 + t2;
    xl[i] = t3 + t4;
    if (aux[8][i] == 1.0f) { xl[i] = xl[i] * (float)capsize_fraction; yl[i] = yl[i] * (float)capsize_fraction; }
    xl[i] = -xl[i];
  }
  orc_update_positions_f32(n, lon, lat, xl, yl, moving, dt);
  orc_update_positions_f32(n, lon, lat, u, v, moving, dt);
  for (i = 0; i < n; ++i) {
    volatile float om = 1.0f - aux[6][i];
    volatile float rate = -(float)log((double)om) / 3600.0f;
    volatile float arg = -rate * (float)fabs(dt);
    volatile float ps = 1.0f - (float)exp((double)arg);
    if ((double)ps > uniforms[i]) { aux[1][i] = -aux[1][i]; aux[7][i] = 1.0f - aux[7][i]; }
  }
  free(xl); free(yl);
}
