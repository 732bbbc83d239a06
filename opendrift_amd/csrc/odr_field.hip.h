// Device-side field sources: projections, analytic readers, gridded ReaderBlock
// sampling (bilinear / nearest / z-lerp / time-lerp), vector rotation and the
// Environment priority-list walk.  One call = what one particle receives from
// Environment.get_environment (opendrift/models/basemodel/environment.py:499-923).
#pragma once
#include <hip/hip_runtime.h>
#include "odr_geodesic.hip.h"

namespace odr {

// -DODR_PHASE_TIMING (developer build, tools/vbuild.sh): per-phase shader cycles of k_step_grid, summed over waves
// (s_memtime stamps where the phase's results are consumed; dumped by odr_i_phase_dump at context destruction)
#ifdef ODR_PHASE_TIMING
__device__ unsigned long long g_phase[32];
#define ODR_PT_DECL unsigned long long pt_[24] = {0}
#define ODR_PT(k) do { if (pt_) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); pt_[k] = __builtin_readcyclecounter(); } } while (0)
#define ODR_PT_USE(x) asm volatile("" :: "v"(x))
#define ODR_PT_ARG , pt_
#define ODR_PT_PARAM , unsigned long long *pt_ = nullptr
#define ODR_PT_NULLARG , nullptr
#else
#define ODR_PT_DECL
#define ODR_PT(k)
#define ODR_PT_USE(x)
#define ODR_PT_ARG
#define ODR_PT_PARAM
#define ODR_PT_NULLARG
#endif

constexpr int NVAR = 26;
constexpr int MAXLEVELS = 6;   // resident time levels per reader: t0, t0 + dt/2 and t0 + dt may each sit between two different levels
constexpr int MAXSRC = 16;
constexpr int MAXNZ = 64;
constexpr int MAXLIST = 4;

enum { VAR_U = 0, VAR_V = 1, VAR_XWIND = 2, VAR_YWIND = 3, VAR_W = 4, VAR_KZ = 5, VAR_SX = 6,
       VAR_SY = 7, VAR_LAND = 8, VAR_DEPTH = 9, VAR_SSH = 10, VAR_HDIFF = 11, VAR_HS = 12,
       VAR_TP = 13, VAR_MLD = 14, VAR_TEMP = 15, VAR_SALT = 16,
       // OpenOil.advect_oil in ice (openoil.py:1179-1216); the windsea_swell Stokes profile (physics_methods.py:418-456)
       VAR_ICE_A = 17, VAR_ICE_U = 18, VAR_ICE_V = 19, VAR_SWELL_DIR = 20, VAR_SWELL_TP = 21, VAR_SWELL_HS = 22,
       VAR_WW_DIR = 23, VAR_WW_TM = 24, VAR_WW_HS = 25 };
enum { SRC_CONSTANT = 0, SRC_DOUBLE_GYRE = 1, SRC_OSCILLATING = 2, SRC_GRID = 3, SRC_LANDMASK = 4 };
enum { PROJ_LATLONG = 0, PROJ_STERE_EQUIT_SPHERE = 1, PROJ_STERE_POLAR = 2, PROJ_CURVILINEAR = 3, PROJ_MERC = 4, PROJ_LCC = 5,
       PROJ_TMERC = 6, PROJ_LAEA = 7, PROJ_STERE_OBLIQUE = 8, PROJ_OB_TRAN = 9,   // round 5 (include/odrift.h)
       PROJ_EXT = 10 };   // (template value only: the kernel instantiation that serves PROJ_TMERC ... PROJ_OB_TRAN, odr_proj_template)
// (kernels templated on the projection: anything but latlong / polar stere / curvilinear takes the PROJ_STERE_EQUIT_SPHERE
// instantiation, whose proj_fwd / proj_inv / rotation_angle switch on DevProj::kind at run time)
// x/y vector pairs are rotated from the reader's CRS to lon/lat (variables.py:799-837); for a reader without a
// projection (fakeproj) the rotation the reference computes is by the azimuth of due north, i.e. exactly zero
#define ODR_PROJ_ROTATES(K) ((K) == PROJ_STERE_EQUIT_SPHERE || (K) == PROJ_STERE_POLAR || (K) == PROJ_MERC || (K) == PROJ_LCC || (K) == PROJ_EXT)

struct D2 { double x, y; };

struct DevProj {
  int kind, south;
  double a, es, e, lon0, lat0, x0, y0, k0, akm1;
  double cchi[4];  // conformal -> geodetic latitude series (Snyder 3-5), used by rotation_angle
  double cn, cc, crho0;  // PROJ_LCC: cone constant n, F = m1 / (n t1^n), rho0 / a (Snyder 15-1..15-8); PROJ_MERC: k0 in k0
  // round 5 -- set-up constants of tmerc / laea / oblique stere / ob_tran (proj_init, odrift.hip):
  //   PROJ_TMERC          q[0] = k0 A / a, q[1] = xi of the origin latitude, q[2..7] = alpha_1..6, q[8..13] = beta_1..6 (Karney 2011)
  //   PROJ_LAEA           q[0] = qp, q[1] = Rq, q[2] = D, q[3] = Rq D, q[4] = Rq / D, q[5] / q[6] = sin / cos beta1, q[7] / q[8] = sin / cos lat0
  //   PROJ_STERE_OBLIQUE  q[0] / q[1] = sin / cos of the conformal latitude of the origin (its geodetic latitude on a sphere)
  //   PROJ_OB_TRAN        q[0] / q[1] = sin / cos o_lat_p, q[2] = o_lon_p
  // mode (laea, oblique stere): 0 north pole, 1 south pole, 2 equatorial, 3 oblique
  int mode, pad2;
  double q[14];
  // PROJ_CURVILINEAR: a reader WITHOUT a projection (2D lon/lat node arrays, pixel indices as x/y;
  // basereader/structured.py:44-113).  cv_nodes[j*cv_nx + i] = (lon, lat) of node (i, j); cv_tri_v / cv_tri_n = the
  // Delaunay triangulation of the nodes (counter-clockwise vertices; neighbour across the edge opposite vertex k,
  // -1 on the outline; csrc/odr_mesh.h); cv_bucket = a triangle near every bucket of a uniform lon/lat raster.
  const D2 *cv_nodes;
  const int *cv_tri_v, *cv_tri_n;
  const int *cv_bucket;
  int cv_nx, cv_maxit, cv_nbx, cv_nby;
  double cv_bx0, cv_by0, cv_ibx, cv_iby;
};

struct DevBlock {
  int ny, nx, valid, pad;
  double x0, xspan, y0, yspan;      // Linear2DInterpolator index map (interpolators.py:110-111)
  double xmin, xrange, ymin, yrange;  // Nearest2DInterpolator index map (interpolators.py:32-37)
  double ixspan, iyspan, ixrange, iyrange;  // correctly rounded reciprocals (host) for div_cr()
  double t;
  // Pre-dilated device data, ONE allocation per time level laid out as node records: `rec`
  // floats per grid node hold every variable of the reader at that node, z innermost.  Element
  // (k, y, x) of variable v lives at data[v][(y*nx + x)*rec + k*es[v]] with data[v] = base + the
  // variable's offset in the record.  x/y_sea_water_velocity (and the other vector pairs) are
  // interleaved (es = 2): one 16-byte load fetches (u,v) at two adjacent z levels.  All variables
  // of a particle's 2x2 footprint share four node offsets, and neighbouring variables share
  // cache lines.
  const float *base;
  int rec, small;  // small: node count < 2^24 and the block < 4 GiB -> 24-bit / 32-bit offset math
  const float *data[NVAR];
  int var_nz[NVAR];
  int es[NVAR];
};

struct DevSource {
  int kind, lon_mode, mod360_x, nlevels, nz, always_valid;
  DevProj proj;
  double xmin, xmax, ymin, ymax, zmin, zmax;
  double tmin, tmax;  // covers_time (variables.py:392-400)
  double const_val[NVAR];
  double params[8];
  double z[MAXNZ];
  // Linear1DInterpolator (interp1d) per ascending interval j: node depth, slope (+-1/dz, IEEE
  // division done once on the host) and the index value at the node
  double zi_x[MAXNZ], zi_slope[MAXNZ], zi_y[MAXNZ];
  double zasc[MAXNZ];  // z levels in ascending order
  // np.gradient(K, mixing_z) constants (oceandrift.py:501): edge / uniform-interior divisors with
  // their reciprocals, and the second-order coefficients of non-uniform interior levels
  double vg_d[3], vg_id[3];
  double vg_a[MAXNZ], vg_b[MAXNZ], vg_c[MAXNZ];
  int vg_uniform;
  int xy_f32;   // bit 0 / 1: the reader's x / y coordinate arrays are float32 (odr_source_set_coordinate_dtype): with DevWorld::f32pos
                // the index maps of a geographic reader are float32 arithmetic (index_f32 below)
  double zmid[MAXNZ];  // mid-depths -(z[k] + z[k+1])/2 formed as d[k] + 0.5*(d[k+1]-d[k]), d = -z (k_vmix level search)
  int level_slot[MAXLEVELS];  // slots sorted by time
  // > 1: the reader hands the variable out as a LIST of ensemble members (readers/interpolation/structured.py:119-135);
  // the block holds them one after the other along the layer axis (var_nz = members x levels); element number j of a
  // call takes member j % members (odr_source_set_members; sampled by the generic kernels only)
  int members[NVAR];
  DevBlock slot[MAXLEVELS];
};

struct DevWorld {
  int nsrc;
  // odr_ctx_set_position_class: the MAIN-LOOP samples treat the element positions as the reference's float32 arrays of the first
  // get_environment of a run (elements/elements.py:71-88: lon / lat are float32 until the first update_positions) --
  // modulate_longitude (variables.py:259-280, :914) then forms np.mod(lon + 180, 360) - 180 in float32: the sample longitude is
  // lon on the float32 grid of lon + 180 (lon_f32class).  The Runge-Kutta stage positions are float64 in the reference.
  int f32pos;   // bit 0: the main-loop samples (odr_ctx_set_position_class); bit 1: set by odr_vmix for its launch when the step's profiles
                // were sampled in that class (k_vmix: float32 index maps of the K column's footprint)
  int nlist[NVAR];
  int list[NVAR][MAXLIST];
  float fallback[NVAR];
  DevSource src[MAXSRC];
};

static constexpr double kPi = 3.14159265358979323846264338327950288;
static constexpr double kHalfPi = 1.57079632679489661923;

// Correctly rounded a / b for a divisor whose correctly rounded reciprocal ib = RN(1/b) is
// known (wave-uniform grid constants, computed once on the host): quotient estimate plus two
// exact-residual corrections (Markstein).  5 instructions instead of the ~25 of the IEEE divide
// expansion, same bits (tests/test_gpu_parity.py compares environment values bit for bit).
__device__ __forceinline__ double div_cr(double a, double b, double ib) {
  double q = a * ib;
  q = fma(fma(-b, q, a), ib, q);
  q = fma(fma(-b, q, a), ib, q);
  return q;
}
__device__ __forceinline__ float div_cr_f32(float a, float b, float ib) {
  float q = a * ib;
  return fmaf(fmaf(-b, q, a), ib, q);
}

__device__ __forceinline__ double np_mod(double x, double m) {  // numpy.mod
  // |x| < m (m > 0): fmod returns x itself -- the usual case for longitudes -- without the
  // library's exponent-ladder loop
  if (m > 0 && fabs(x) < m) return x < 0 ? x + m : x;
  double r = fmod(x, m);
  if (r != 0 && ((r < 0) != (m < 0))) r += m;
  return r;
}

// modulate_longitude on a float32 longitude (DevWorld::f32pos).  The result is a fixed point of the float64 modulation the
// samplers apply afterwards, so a sampler is simply handed this longitude.
__device__ __forceinline__ double lon_f32class(int lon_mode, double lon) {
  float l = (float)lon;
  if (lon_mode == 1) l = __fsub_rn((float)np_mod((double)__fadd_rn(l, 180.0f), 360.0), 180.0f);   // (np.mod of a float32 is exact)
  else if (lon_mode == 2) l = (float)np_mod((double)l, 360.0);
  return (double)l;
}

// ---------------------------------------------------------------- projections
// pyproj.Proj forward / inverse for the reader projections on the path
// (variables.py:111-143); formulas: Snyder PP1395 ch. 21.
__device__ __forceinline__ double wrap_pi(double lam) {
  if (fabs(lam) <= kPi + 1e-12) return lam;
  lam += kPi;
  lam -= 2 * kPi * floor(lam / (2 * kPi));
  lam -= kPi;
  return lam;
}

// Snyder 15-9: t = tan(pi/4 - phi/2) / ((1 - e sin phi)/(1 + e sin phi))^(e/2)
//             = tan(pi/4 - phi/2) * exp(e * atanh(e sin phi))      (|e sin phi| < 0.09: 7-term series)
__device__ __forceinline__ double tsfn(double phi, double sinphi, double e) {
#pragma clang fp contract(fast)
  double es = e * sinphi, q = es * es;
  double ath = es * (1 + q * (1.0 / 3 + q * (1.0 / 5 + q * (1.0 / 7 + q * (1.0 / 9 + q * (1.0 / 11 + q * (1.0 / 13)))))));
  return tan(0.5 * (kHalfPi - phi)) * exp(e * ath);
}

// StructuredReader.lonlat2xy of a reader without projection (structured.py:438-472): the reference evaluates
// scipy's LinearNDInterpolator -- piecewise-linear interpolation of the pixel indices over the Delaunay
// triangulation of the (lon, lat) nodes.  The triangulation is prepared once on the host from the structured
// mesh (odr_mesh.h); here: visibility walk from the bucket raster's start triangle, then the barycentric
// combination of the vertices' pixel indices.  Outside the mesh outline: NaN (not covered).
__device__ __forceinline__ double cross2(double ax, double ay, double bx, double by) {
  return __dsub_rn(__dmul_rn(ax, by), __dmul_rn(ay, bx));
}
__device__ __forceinline__ void curvi_locate(const DevProj &p, double lon, double lat, double &x, double &y) {
  x = y = __builtin_nan("");
  const double fx = (lon - p.cv_bx0) * p.cv_ibx, fy = (lat - p.cv_by0) * p.cv_iby;
  if (!(fx >= 0 && fy >= 0 && fx < (double)p.cv_nbx && fy < (double)p.cv_nby)) return;
  int t = p.cv_bucket[(int)fy * p.cv_nbx + (int)fx];
  for (int it = 0; t >= 0 && it < p.cv_maxit; ++it) {
    const int *v = p.cv_tri_v + 3 * (size_t)t;
    const int v0 = v[0], v1 = v[1], v2 = v[2];
    const D2 p0 = p.cv_nodes[v0], p1 = p.cv_nodes[v1], p2 = p.cv_nodes[v2];
    const double e0 = cross2(p2.x - p1.x, p2.y - p1.y, lon - p1.x, lat - p1.y);  // edge opposite vertex 0
    const double e1 = cross2(p0.x - p2.x, p0.y - p2.y, lon - p2.x, lat - p2.y);
    const double e2 = cross2(p1.x - p0.x, p1.y - p0.y, lon - p0.x, lat - p0.y);
    const double worst = fmin(e0, fmin(e1, e2));
    if (!(worst < 0)) {   // inside (or on the border of) this triangle
      const double det = cross2(p1.x - p0.x, p1.y - p0.y, p2.x - p0.x, p2.y - p0.y);
      const double c0 = __ddiv_rn(e0, det), c1 = __ddiv_rn(e1, det);
      const int nx = p.cv_nx;
      const int j0 = v0 / nx, j1 = v1 / nx, j2 = v2 / nx;
      const int i0 = v0 - j0 * nx, i1 = v1 - j1 * nx, i2 = v2 - j2 * nx;
      x = (double)i2 + __dadd_rn(__dmul_rn(c0, (double)(i0 - i2)), __dmul_rn(c1, (double)(i1 - i2)));
      y = (double)j2 + __dadd_rn(__dmul_rn(c0, (double)(j0 - j2)), __dmul_rn(c1, (double)(j1 - j2)));
      return;
    }
    t = p.cv_tri_n[3 * (size_t)t + (worst == e0 ? 0 : worst == e1 ? 1 : 2)];
  }
}

// covers_positions_xy (variables.py:236-257): a reader whose CRS is geographic -- latlong, and the rotated pole, whose
// `crs.is_geographic` the reference relies on (:117, :800) -- has its x modulated once more for the test of its domain
template <int PROJ>
__device__ __forceinline__ double cover_x(int kind, int lon_mode, double x) {
  if (PROJ == PROJ_LATLONG || (PROJ == PROJ_EXT && kind == PROJ_OB_TRAN)) {
    if (lon_mode == 1) return np_mod(x + 180.0, 360.0) - 180.0;
    if (lon_mode == 2) return np_mod(x, 360.0);
  }
  return x;
}

// ---- round 5: transverse Mercator / UTM, Lambert azimuthal equal-area, oblique stereographic, rotated pole ------------
// (variables.py:111-143 hands any proj4 to pyproj; these are the ones model files carry besides stere / merc / lcc.  Same
// formulas as oracle/proj.c, which is pinned on Snyder's numerical examples; written for the device: no iteration where a
// closed form or a fixed small count does, hyperbolic multiples by recurrence.)
// conformal latitude as its tangent (Karney 2011, eqs. 7-9)
__device__ __forceinline__ double tm_taup(double tau, double e) {
#pragma clang fp contract(fast)
  const double t1 = sqrt(1 + tau * tau), sg = sinh(e * atanh(e * tau / t1));
  return sqrt(1 + sg * sg) * tau - sg * t1;
}
// sum_k c[k] sin(2 k xi) cosh(2 k eta) and sum_k c[k] cos(2 k xi) sinh(2 k eta), k = 1..6: the complex Clenshaw sum of
// c_k sin(2 k (xi + i eta)) -- real and imaginary part
__device__ __forceinline__ void tm_series(const double *c, double xi, double eta, double &sre, double &sim) {
#pragma clang fp contract(fast)
  double s2, c2;
  sincos(2 * xi, &s2, &c2);
  const double ch = cosh(2 * eta), sh = sinh(2 * eta);
  // sin(2z) = s2 ch + i c2 sh,  2 cos(2z) = 2 c2 ch - 2 i s2 sh
  const double ar = 2 * c2 * ch, ai = -2 * s2 * sh;
  double y1r = 0, y1i = 0, y0r = c[5], y0i = 0;
#pragma unroll
  for (int k = 4; k >= 0; --k) {
    const double tr = ar * y0r - ai * y0i - y1r + c[k], ti = ar * y0i + ai * y0r - y1i;
    y1r = y0r; y1i = y0i; y0r = tr; y0i = ti;
  }
  const double zr = s2 * ch, zi = c2 * sh;
  sre = zr * y0r - zi * y0i;
  sim = zr * y0i + zi * y0r;
}
__device__ __forceinline__ double qsfn_dev(double sinphi, double e, double one_es) {   // Snyder 3-12
#pragma clang fp contract(fast)
  if (e < 1e-7) return 2 * sinphi;
  const double con = e * sinphi;
  return one_es * (sinphi / (1 - con * con) + (1.0 / e) * atanh(con));               // -(1/2e) ln((1-x)/(1+x)) = atanh(x) / e
}
__device__ __forceinline__ double conformal_lat(double phi, double sinphi, double e) { // Snyder 3-1
  return 2 * atan(tan(0.5 * (kHalfPi + phi)) * exp(-e * atanh(e * sinphi))) - kHalfPi;
}
// (lam, phi) relative to the central meridian -> projected X, Y in units of a (ob_tran: rotated longitude / latitude, radians)
// Compiled into the PROJ_EXT instantiations of the kernels only (template flag EXT of proj_fwd / proj_inv / rotation_*): inlined
// into every sampler of the kernels that serve Mercator / Lambert / stereographic readers it took k_step_grid<.., generic> from
// 195-243 to 256 registers + 112-272 B of scratch memory (and as out-of-line calls to 420-630 B around the call sites).
__device__ __forceinline__ void proj_fwd_ext(const DevProj &p, double lam, double phi, double &X, double &Y) {
#pragma clang fp contract(fast)
  double sinlam, coslam, sinphi, cosphi;
  sincos(lam, &sinlam, &coslam);
  sincos(phi, &sinphi, &cosphi);
  if (p.kind == PROJ_TMERC) {
    const double taup = tm_taup(sinphi / cosphi, p.e);
    double xip = atan2(taup, coslam), etap = asinh(sinlam / sqrt(taup * taup + coslam * coslam));
    if (fabs(phi) >= kHalfPi) { xip = phi > 0 ? kHalfPi : -kHalfPi; etap = 0; }
    double sr, si;
    tm_series(p.q + 2, xip, etap, sr, si);
    X = p.q[0] * (etap + si);
    Y = p.q[0] * (xip + sr - p.q[1]);
  } else if (p.kind == PROJ_LAEA) {
    if (p.es != 0) {
      double q = qsfn_dev(sinphi, p.e, 1 - p.es);
      const double qp = p.q[0];
      if (p.mode >= 2) {
        const double sinb = q / qp, cosb = sqrt(1 - sinb * sinb);
        double b;
        if (p.mode == 3) {
          b = sqrt(2 / (1 + p.q[5] * sinb + p.q[6] * cosb * coslam));
          Y = p.q[4] * b * (p.q[6] * sinb - p.q[5] * cosb * coslam);
        } else {
          b = sqrt(2 / (1 + cosb * coslam));
          Y = b * sinb * p.q[4];
        }
        X = p.q[3] * b * cosb * sinlam;
      } else {
        q = p.mode == 0 ? qp - q : qp + q;
        const double b = q >= 1e-30 ? sqrt(q) : 0.0;
        X = b * sinlam;
        Y = coslam * (p.mode == 1 ? b : -b);
      }
    } else if (p.mode >= 2) {
      const double k = sqrt(2 / (1 + p.q[7] * sinphi + p.q[8] * cosphi * coslam));
      X = k * cosphi * sinlam;
      Y = k * (p.q[8] * sinphi - p.q[7] * cosphi * coslam);
    } else {
      const double h = 0.25 * kPi - 0.5 * phi, k = 2 * (p.mode == 1 ? cos(h) : sin(h));
      X = k * sinlam;
      Y = k * (p.mode == 0 ? -coslam : coslam);
    }
  } else if (p.kind == PROJ_STERE_OBLIQUE) {
    double sinX = sinphi, cosX = cosphi;
    if (p.es != 0) sincos(conformal_lat(phi, sinphi, p.e), &sinX, &cosX);
    const double A = p.akm1 / ((p.es != 0 ? p.q[1] : 1.0) * (1 + p.q[0] * sinX + p.q[1] * cosX * coslam));
    X = A * cosX * sinlam;
    Y = A * (p.q[1] * sinX - p.q[0] * cosX * coslam);
  } else {   // PROJ_OB_TRAN (PROJ's o_forward)
    X = wrap_pi(atan2(cosphi * sinlam, p.q[0] * cosphi * coslam + p.q[1] * sinphi) + p.q[2]);
    Y = asin(fmin(1.0, fmax(-1.0, p.q[0] * sinphi - p.q[1] * cosphi * coslam)));
  }
}
__device__ __forceinline__ void proj_inv_ext(const DevProj &p, double X, double Y, double &lam, double &phi) {
#pragma clang fp contract(fast)
  if (p.kind == PROJ_TMERC) {
    const double xi = Y / p.q[0] + p.q[1], eta = X / p.q[0];
    double sr, si;
    tm_series(p.q + 8, xi, eta, sr, si);
    const double xip = xi - sr, etap = eta - si;
    const double sh = sinh(etap), c = cos(xip), taup = sin(xip) / sqrt(sh * sh + c * c);
    lam = atan2(sh, c);
    // tau from its conformal counterpart: Newton (Karney 2011, eqs. 19-21), quadratic from a start that is e^2 off
    const double e2m = 1 - p.es;
    double tau = taup / e2m;
#pragma unroll 1
    for (int i = 0; i < 5 && p.es != 0; ++i) {
      const double tp = tm_taup(tau, p.e);
      const double d = (taup - tp) * (1 + e2m * tau * tau) / (e2m * sqrt(1 + tau * tau) * sqrt(1 + tp * tp));
      tau += d;
      if (!(fabs(d) >= 1e-15 * fmax(1.0, fabs(taup)))) break;
    }
    phi = atan(p.es != 0 ? tau : taup);
  } else if (p.kind == PROJ_LAEA) {
    double x = X, y = Y;
    if (p.es != 0) {
      const double qp = p.q[0];
      double ab;
      if (p.mode >= 2) {
        x /= p.q[2]; y *= p.q[2];
        const double rho = sqrt(x * x + y * y);
        if (rho < 1e-10) { lam = 0; phi = p.lat0; return; }
        double sCe, cCe;
        sincos(2 * asin(0.5 * rho / p.q[1]), &sCe, &cCe);
        x *= sCe;
        if (p.mode == 3) { ab = cCe * p.q[5] + y * sCe * p.q[6] / rho; y = rho * p.q[6] * cCe - y * p.q[5] * sCe; }
        else { ab = y * sCe / rho; y = rho * cCe; }
      } else {
        if (p.mode == 0) y = -y;
        const double q = x * x + y * y;
        if (q == 0) { lam = 0; phi = p.lat0; return; }
        ab = 1 - q / qp;
        if (p.mode == 1) ab = -ab;
      }
      lam = atan2(x, y);
      ab = fmin(1.0, fmax(-1.0, ab));
      double ph = asin(ab);        // authalic latitude; geodetic by Newton on q(phi) = qp sin(beta) (Snyder 3-16)
#pragma unroll 1
      for (int i = 0; i < 8; ++i) {
        double sp, cp;
        sincos(ph, &sp, &cp);
        if (!(fabs(cp) > 1e-12)) break;
        const double w = 1 - p.es * sp * sp;
        const double d = w * w / (2 * cp) * (qp * ab / (1 - p.es) - sp / w - (1.0 / p.e) * atanh(p.e * sp));
        ph += d;
        if (fabs(d) < 1e-15) break;
      }
      phi = ph;
    } else {
      const double rh = sqrt(x * x + y * y), c = 2 * asin(fmin(1.0, 0.5 * rh));
      double sinz, cosz;
      sincos(c, &sinz, &cosz);
      if (p.mode >= 2) {
        const double ph = rh <= 1e-10 ? p.lat0 : asin(cosz * p.q[7] + y * sinz * p.q[8] / rh);
        x *= sinz * p.q[8];
        y = (cosz - sin(ph) * p.q[7]) * rh;
        lam = (y == 0 && x == 0) ? 0.0 : atan2(x, y);
        phi = ph;
      } else {
        if (p.mode == 0) { y = -y; phi = kHalfPi - c; } else phi = c - kHalfPi;
        lam = atan2(x, y);
      }
    }
  } else if (p.kind == PROJ_STERE_OBLIQUE) {
    double x = X, y = Y;
    const double rho = sqrt(x * x + y * y);
    if (p.es != 0) {
      double sinphi, cosphi;
      sincos(2 * atan2(rho * p.q[1], p.akm1), &sinphi, &cosphi);
      double phi_l = rho == 0 ? asin(cosphi * p.q[0]) : asin(cosphi * p.q[0] + y * sinphi * p.q[1] / rho);
      const double tp = tan(0.5 * (kHalfPi + phi_l));
      x *= sinphi;
      y = rho * p.q[1] * cosphi - y * p.q[0] * sinphi;
      double ph = phi_l;
#pragma unroll 1
      for (int i = 0; i < 10; ++i) {   // Snyder 3-4 (contraction ~ e^2 per sweep)
        ph = 2 * atan(tp * exp(p.e * atanh(p.e * sin(phi_l)))) - kHalfPi;
        if (fabs(ph - phi_l) < 1e-15) break;
        phi_l = ph;
      }
      phi = ph;
      lam = (x == 0 && y == 0) ? 0.0 : atan2(x, y);
    } else {
      double sinc, cosc;
      sincos(2 * atan(rho / p.akm1), &sinc, &cosc);
      const double ph = rho <= 1e-10 ? p.lat0 : asin(cosc * p.q[0] + y * sinc * p.q[1] / rho);
      const double cc = cosc - p.q[0] * sin(ph);
      lam = (cc != 0 || x != 0) ? atan2(x * sinc * p.q[1], cc * rho) : 0.0;
      phi = ph;
    }
  } else {   // PROJ_OB_TRAN (PROJ's o_inverse)
    const double l = X - p.q[2];
    double sl, cl, sp, cp;
    sincos(l, &sl, &cl);
    sincos(Y, &sp, &cp);
    phi = asin(fmin(1.0, fmax(-1.0, p.q[0] * sp + p.q[1] * cp * cl)));
    lam = atan2(cp * sl, p.q[0] * cp * cl - p.q[1] * sp);
  }
}

// polar stereographic (ellipsoid) from the sines and cosines of (lambda - lambda0, phi): the part of proj_fwd behind the
// trigonometry, shared with the step launch that has them from its start point (ProjStart, env_front_from_start)
__device__ __forceinline__ void stere_polar_from_sc(const DevProj &p, double phi, double sinlam, double coslam, double sinphi,
                                                    double cosphi, double &x, double &y) {
#pragma clang fp contract(fast)
  if (p.south) { phi = -phi; coslam = -coslam; sinphi = -sinphi; }
  double rho = 0.0;
  if (!(fabs(phi - kHalfPi) < 1e-15)) {
    if (sinphi > -0.9 && p.e < 0.1) {
      const double es = p.e * sinphi, q = es * es;
      const double ath = es * (1 + q * (1.0 / 3 + q * (1.0 / 5 + q * (1.0 / 7 + q * (1.0 / 9 + q * (1.0 / 11 + q * (1.0 / 13)))))));
      rho = p.akm1 * (cosphi * fast_rcp(1 + sinphi)) * exp_small(p.e * ath);
    } else rho = p.akm1 * tsfn(phi, sinphi, p.e);
  }
  x = p.a * (rho * sinlam) + p.x0;
  y = p.a * (-rho * coslam) + p.y0;
}

// ELLPOLAR: the caller knows the projection to be the polar stereographic one on an ellipsoid (the kernels instantiated
// for PROJ_STERE_POLAR: the host sends spherical polar readers to the generic instantiation) -- no code, and no registers,
// for the other kinds (k_step_grid<RK4, polar> held 39 700 VALU instructions / 215 VGPRs with them)
template <bool ELLPOLAR = false, bool EXT = true>
__device__ __forceinline__ void proj_fwd(const DevProj &p, double lon_deg, double lat_deg,
                                         double &x, double &y) {
#pragma clang fp contract(fast)
  if (!ELLPOLAR && p.kind == PROJ_LATLONG) { x = lon_deg; y = lat_deg; return; }
  double lam = wrap_pi(lon_deg * kDeg - p.lon0), phi = lat_deg * kDeg;
  double sinlam, coslam, sinphi, cosphi, X, Y;
  if (!ELLPOLAR && EXT && p.kind >= PROJ_TMERC) {
    proj_fwd_ext(p, lam, phi, X, Y);
    if (p.kind == PROJ_OB_TRAN) { x = X * kRad2Deg; y = Y * kRad2Deg; return; }   // np.degrees(self.proj(lon, lat)), variables.py:136-138
    x = p.a * X + p.x0;
    y = p.a * Y + p.y0;
    return;
  }
  if (ELLPOLAR || (p.kind == PROJ_STERE_POLAR && p.es != 0)) {
    // (polar stereographic on an ellipsoid: the particle's own position in every step of C4 / C5; the same arithmetic in the
    // instantiation of the polar readers (ELLPOLAR) and in the kernels that take the projection at run time.  Sines and
    // cosines on their known ranges (sincos_pi), tan(pi/4 - phi/2) = cos phi / (1 + sin phi) away from the opposite pole, the
    // ellipsoidal factor's exponential by its series: ~150 instead of ~290 instructions, equal to rounding)
    sincos_pi(lam, sinlam, coslam);
    sincos_pi(phi, sinphi, cosphi);
    stere_polar_from_sc(p, phi, sinlam, coslam, sinphi, cosphi, x, y);
    return;
  }
  if (p.kind == PROJ_MERC) {        // Snyder 7-6 / 7-7 (sphere: e = 0)
    x = p.a * (p.k0 * lam) + p.x0;
    y = p.a * (-p.k0 * log(tsfn(phi, sin(phi), p.e))) + p.y0;
    return;
  }
  if (p.kind == PROJ_LCC) {         // Snyder 15-7, 14-4, 14-1, 14-2
    const double rho = fabs(fabs(phi) - kHalfPi) < 1e-10 ? 0.0 : p.cc * pow(tsfn(phi, sin(phi), p.e), p.cn);
    sincos(p.cn * lam, &sinlam, &coslam);
    x = p.a * (p.k0 * (rho * sinlam)) + p.x0;
    y = p.a * (p.k0 * (p.crho0 - rho * coslam)) + p.y0;
    return;
  }
  sincos(lam, &sinlam, &coslam);
  sincos(phi, &sinphi, &cosphi);
  if (p.kind == PROJ_STERE_EQUIT_SPHERE) {
    double k = p.akm1 / (1 + cosphi * coslam);
    X = k * cosphi * sinlam;
    Y = k * sinphi;
  } else {
    double rho;
    if (p.south) { phi = -phi; coslam = -coslam; sinphi = -sinphi; }
    if (p.es == 0) rho = p.akm1 * tan(0.5 * (kHalfPi - phi));
    else rho = fabs(phi - kHalfPi) < 1e-15 ? 0.0 : p.akm1 * tsfn(phi, sinphi, p.e);
    X = rho * sinlam;
    Y = -rho * coslam;
  }
  x = p.a * X + p.x0;
  y = p.a * Y + p.y0;
}

// ---- projection of a position NEAR one whose sines and cosines are known (the Runge-Kutta stage positions: a few km from
// the particle).  sin / cos of (lambda, phi) by the angle-addition formulas with 4-term series of the small differences
// (|d| < 8e-3 rad: truncation < 1e-20), tan(pi/4 - phi/2) = cos phi / (1 + sin phi): ~110 instead of ~360 instructions.
// Agrees with proj_fwd to rounding (1e-16 relative); anything farther away, and the spherical variant, take proj_fwd.
struct ProjStart { double lam = 0, phi = 0, sl = 0, cl = 0, sp = 0, cp = 0; int ok = 0, pad = 0; };
__device__ __forceinline__ ProjStart proj_start(const DevProj &p, double lon_deg, double lat_deg) {
  ProjStart o;
  o.ok = p.kind == PROJ_STERE_POLAR && p.es != 0;
  o.pad = 0;
  o.lam = wrap_pi(lon_deg * kDeg - p.lon0);
  o.phi = lat_deg * kDeg;
  sincos_pi(o.lam, o.sl, o.cl);
  sincos_pi(o.phi, o.sp, o.cp);
  return o;
}
// The closed forms of the vector rotation below hold when the map is conformal on the ellipsoid the reference's geodesic runs on
// (pyproj.Geod(ellps='WGS84'); GRS80 differs by 3e-11 in e^2).  A conformal map of ANOTHER figure (a sphere: 1.7e-3 rad) keeps the
// azimuths of that figure, not WGS84's: those readers take rotation_cs / rotation_angle as in rounds 1-4.
__device__ __forceinline__ bool rot_same_ellipsoid(const DevProj &p) { return fabs(p.es - c_geod.e2) <= 1e-7; }
// rot (ODR_STAGE_FAST stage samples): also cos / sin of rot_angle_rad = -(WGS84 azimuth of the 10 m line (x, y) -> (x, y + 10)) of
// rotate_vectors (variables.py:59-109) in closed form -- a polar stereographic map is azimuthal and conformal: its +y axis has the
// azimuth +-(lambda - lambda0) at every point, the chord's azimuth at its start differs from that by half the change of longitude
// along the chord times (1 - |sin phi|) (mid-point direction minus half the meridian convergence of the geodesic): ~25 instructions
// from the sines the projection has formed anyway instead of ~250 (rotation_cs); 4e-10 rad from the geodesic inverse (oracle).
template <bool ELLPOLAR = false, bool EXT = true>
__device__ __forceinline__ void proj_fwd_near(const DevProj &p, const ProjStart &o, double lon_deg, double lat_deg,
                                              double &x, double &y, double *rot = nullptr) {
#pragma clang fp contract(fast)
  const double lam = wrap_pi(lon_deg * kDeg - p.lon0), phi0 = lat_deg * kDeg;
  const double dl = lam - o.lam, dp = phi0 - o.phi;
  if (!(o.ok && fabs(dl) < 8e-3 && fabs(dp) < 8e-3)) {
    proj_fwd<ELLPOLAR, EXT>(p, lon_deg, lat_deg, x, y);
    if (rot) rot[0] = 2.0;     // "not formed here" (a cosine is never 2): the caller takes rotation_cs
    return;
  }
  const double l2 = dl * dl, p2 = dp * dp;
  const double sdl = dl * (1 - l2 * (1.0 / 6) * (1 - l2 * (1.0 / 20) * (1 - l2 * (1.0 / 42))));
  const double cdl = 1 - l2 * 0.5 * (1 - l2 * (1.0 / 12) * (1 - l2 * (1.0 / 30) * (1 - l2 * (1.0 / 56))));
  const double sdp = dp * (1 - p2 * (1.0 / 6) * (1 - p2 * (1.0 / 20) * (1 - p2 * (1.0 / 42))));
  const double cdp = 1 - p2 * 0.5 * (1 - p2 * (1.0 / 12) * (1 - p2 * (1.0 / 30) * (1 - p2 * (1.0 / 56))));
  double sinlam = o.sl * cdl + o.cl * sdl, coslam = o.cl * cdl - o.sl * sdl;
  double sinphi = o.sp * cdp + o.cp * sdp, cosphi = o.cp * cdp - o.sp * sdp;
  double phi = phi0;
  if (p.south) { phi = -phi; coslam = -coslam; sinphi = -sinphi; }
  double rho = 0.0;
  if (!(fabs(phi - kHalfPi) < 1e-15)) {
    const double es = p.e * sinphi, q = es * es;
    const double ath = es * (1 + q * (1.0 / 3 + q * (1.0 / 5 + q * (1.0 / 7 + q * (1.0 / 9 + q * (1.0 / 11 + q * (1.0 / 13)))))));
    rho = p.akm1 * (cosphi / (1 + sinphi)) * (p.e < 0.1 ? exp_small(p.e * ath) : exp(p.e * ath));   // tsfn with tan(pi/4 - phi/2) = cos / (1 + sin)
  }
  x = p.a * (rho * sinlam) + p.x0;
  y = p.a * (-rho * coslam) + p.y0;
  if (rot && !rot_same_ellipsoid(p)) rot[0] = 2.0;     // conformal on another ellipsoid than the geodesic's: rotation_cs
  else if (rot) {
    // (sinlam, +-coslam: sine and cosine of lambda - lambda0; sinphi: of the latitude counted from the projection's own pole)
    const double cl = p.south ? -coslam : coslam;
    const double q = rho > 0 ? 5.0 * sinlam * (1 - sinphi) * fast_rcp(p.a * rho) : 0.0;   // 0.5 * 10 m * d(lambda)/dy * (1 - |sin phi|)
    if (p.south) { rot[0] = cl + q * sinlam; rot[1] = sinlam - q * cl; }
    else { rot[0] = cl - q * sinlam; rot[1] = -sinlam - q * cl; }
  }
}

// reader projection chosen at run time (kernels that serve any reader)
__device__ __forceinline__ void proj_fwd_rt(const DevProj &p, double lon_deg, double lat_deg, double &x, double &y) {
  if (p.kind == PROJ_CURVILINEAR) curvi_locate(p, lon_deg, lat_deg, x, y);
  else proj_fwd(p, lon_deg, lat_deg, x, y);
}

template <bool EXT = true>
__device__ __forceinline__ void proj_inv(const DevProj &p, double x, double y, double &lon_deg,
                                         double &lat_deg) {
#pragma clang fp contract(fast)
  if (p.kind == PROJ_LATLONG) { lon_deg = x; lat_deg = y; return; }
  if (p.kind >= PROJ_TMERC) {
    if constexpr (EXT) {
      const bool ob = p.kind == PROJ_OB_TRAN;      // self.proj(np.radians(x), np.radians(y), inverse=True), :117-123
      double lam_, phi_;
      proj_inv_ext(p, ob ? x * kDeg : (x - p.x0) / p.a, ob ? y * kDeg : (y - p.y0) / p.a, lam_, phi_);
      lon_deg = wrap_pi(lam_ + p.lon0) / kDeg;
      lat_deg = phi_ / kDeg;
    } else lon_deg = lat_deg = __builtin_nan("");
    return;
  }
  double X = (x - p.x0) / p.a, Y = (y - p.y0) / p.a;
  double rh = hypot(X, Y), lam = 0, phi = 0;
  if (p.kind == PROJ_MERC || p.kind == PROJ_LCC) {
    double ts;
    if (p.kind == PROJ_MERC) { ts = exp(-Y / p.k0); lam = X / p.k0; }       // Snyder 7-10, 7-12
    else {                                                                   // Snyder 14-10, 14-11, 15-11, 14-9
      double xx = X / p.k0, yy = p.crho0 - Y / p.k0, rho = hypot(xx, yy);
      if (p.cn < 0) { rho = -rho; xx = -xx; yy = -yy; }
      ts = rho != 0 ? pow(rho / p.cc, 1 / p.cn) : 0.0;
      lam = rho != 0 ? atan2(xx, yy) / p.cn : 0.0;
      if (rho == 0) { lon_deg = wrap_pi(p.lon0) / kDeg; lat_deg = p.cn > 0 ? 90.0 : -90.0; return; }
    }
    double phi_l = kHalfPi - 2 * atan(ts), halfe = 0.5 * p.e;                // Snyder 7-9 to convergence
    phi = phi_l;
#pragma unroll 1
    for (int i = 0; i < 10 && p.e != 0; ++i) {
      double es = p.e * sin(phi_l);
      phi = kHalfPi - 2 * atan(ts * pow((1 - es) / (1 + es), halfe));
      if (fabs(phi - phi_l) < 1e-15) break;
      phi_l = phi;
    }
    lon_deg = wrap_pi(lam + p.lon0) / kDeg;
    lat_deg = phi / kDeg;
    return;
  }
  if (p.kind == PROJ_STERE_EQUIT_SPHERE) {
    double c = 2 * atan(rh / p.akm1), sinc, cosc;
    sincos(c, &sinc, &cosc);
    phi = fabs(rh) <= 1e-10 ? 0.0 : asin(Y * sinc / rh);
    if (cosc != 0 || X != 0) lam = atan2(X * sinc, cosc * rh);
  } else if (p.es == 0) {
    double c = 2 * atan(rh / p.akm1), cosc = cos(c);
    if (!p.south) Y = -Y;
    phi = fabs(rh) <= 1e-10 ? p.lat0 : asin(p.south ? -cosc : cosc);
    lam = (X == 0 && Y == 0) ? 0.0 : atan2(X, Y);
  } else {
    // conformal-latitude inverse, fixed-point iteration (contraction ~ e^2 per sweep)
    double tp = rh / p.akm1, phi_l = kHalfPi - 2 * atan(tp), halfe = 0.5 * p.e;
    if (!p.south) Y = -Y;
#pragma unroll 1
    for (int i = 0; i < 8; ++i) {
      double es = p.e * sin(phi_l);
      phi = kHalfPi - 2 * atan(tp * pow((1 - es) / (1 + es), halfe));
      if (fabs(phi - phi_l) < 1e-15) break;
      phi_l = phi;
    }
    if (p.south) phi = -phi;
    lam = (X == 0 && Y == 0) ? 0.0 : atan2(X, Y);
  }
  lon_deg = wrap_pi(lam + p.lon0) / kDeg;
  lat_deg = phi / kDeg;
}

// rotate_vectors (variables.py:59-109): azimuth of the reader's +y axis from a 10 m
// finite difference, as the forward azimuth of the WGS84 geodesic between the two
// points.  For a 10 m line the Gauss mid-latitude solution (azimuth at the mid point
// minus half the meridian convergence) equals Karney's inverse to O((s/R)^3) ~ 1e-18 rad.
// Only the DIFFERENCES (dphi, dlam) of the two inverse projections enter, so the polar
// ellipsoidal case uses the closed series for the geodetic latitude (Snyder eq. 3-5,
// 2e-12 rad, the error is common to both points) instead of the fixed-point iteration.
template <bool EXT = true>
__device__ __forceinline__ void proj_inv_diff(const DevProj &p, double x, double y, double dy, double &phim,
                                              double &dphi, double &dlam) {
#pragma clang fp contract(fast)
  if (p.kind == PROJ_STERE_POLAR && p.es != 0) {
    double X = (x - p.x0) / p.a, Y1 = (y - p.y0) / p.a, Y2 = (y + dy - p.y0) / p.a;
    if (!p.south) { Y1 = -Y1; Y2 = -Y2; }
    double ph[2], lam[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      double Y = k ? Y2 : Y1;
      double chi = kHalfPi - 2 * atan(sqrt(X * X + Y * Y) / p.akm1);
      double s2, c2;
      sincos(2 * chi, &s2, &c2);
      // sum c_k sin(2k chi), Clenshaw on the multiple angles
      double ar = 2 * c2, y1 = p.cchi[3], y0 = ar * y1 + p.cchi[2];
      y1 = ar * y0 - y1 + p.cchi[1];
      y0 = ar * y1 - y0 + p.cchi[0];
      ph[k] = chi + s2 * y0;
      lam[k] = atan2(X, Y);
    }
    double sgn = p.south ? -1.0 : 1.0;
    phim = 0.5 * sgn * (ph[0] + ph[1]);
    dphi = sgn * (ph[1] - ph[0]);
    dlam = lam[1] - lam[0];
    if (dlam > kPi) dlam -= 2 * kPi;
    if (dlam < -kPi) dlam += 2 * kPi;
    return;
  }
  double lo1, la1, lo2, la2;
  proj_inv<EXT>(p, x, y, lo1, la1);
  proj_inv<EXT>(p, x, y + dy, lo2, la2);
  phim = 0.5 * (la1 + la2) * kDeg;
  dphi = (la2 - la1) * kDeg;
  dlam = ang_normalize(lo2 - lo1) * kDeg;
}

// Forward azimuth (radians) of the WGS84 geodesic from point 1 to a point 2 some kilometres away: what Geod.inv returns
// for the 0.1-degree line rotate_vectors draws for a reader whose CRS is geographic (variables.py:80-97; the rotated pole).
// The Gauss mid-latitude azimuth is off by ~1e-7 rad over 11 km; it starts a shooting solve on the direct problem --
// the miss at point 2, carried back to point 1 through the meridian convergence, corrects the start vector; each sweep
// leaves (s / R)^2 ~ 3e-6 of the error, two sweeps reach round-off.
__device__ __forceinline__ double geod_inverse_azimuth(double lat1, double lon1, double lat2, double lon2) {
#pragma clang fp contract(fast)
  const GeodConst &g = c_geod;
  const double dphi = (lat2 - lat1) * kDeg, dlam = ang_normalize(lon2 - lon1) * kDeg, phim = 0.5 * (lat1 + lat2) * kDeg;
  double sm, cm, s2, c2;
  sincos(phim, &sm, &cm);
  sincos(lat2 * kDeg, &s2, &c2);
  const double wm2 = 1 - g.e2 * sm * sm, wm = sqrt(wm2), w22 = 1 - g.e2 * s2 * s2, w2 = sqrt(w22);
  const double Mm = g.a * (1 - g.e2) / (wm2 * wm), Nm = g.a / wm, M2 = g.a * (1 - g.e2) / (w22 * w2), N2 = g.a / w2;
  const double gam = dlam * sm;                     // meridian convergence between the two points
  double sg, cg;
  sincos(gam, &sg, &cg);
  // start vector (north, east) at point 1: the chord at the mid latitude turned back by half the convergence
  double sh, ch;
  sincos(0.5 * gam, &sh, &ch);
  const double nm = dphi * Mm, em = dlam * Nm * cm;
  double n1 = nm * ch + em * sh, e1 = em * ch - nm * sh;
  const GeodOrigin o = geod_origin(lat1, lon1);
#pragma unroll 1
  for (int it = 0; it < 2; ++it) {
    const double s = sqrt(n1 * n1 + e1 * e1);
    if (!(s > 0)) break;
    double la, lo;
    geod_direct_sc(o, e1 / s, n1 / s, s, la, lo);
    const double dn = (lat2 - la) * kDeg * M2, de = ang_normalize(lon2 - lo) * kDeg * N2 * c2;
    n1 += dn * cg + de * sg;
    e1 += de * cg - dn * sg;
  }
  return atan2(e1, n1);
}

template <bool EXT = true>
__device__ __forceinline__ double rotation_angle(const DevProj &p, double x, double y) {
#pragma clang fp contract(fast)
  if constexpr (EXT) {
    if (p.kind == PROJ_OB_TRAN) {     // delta_y = 0.1 degree northwards in the reader's CRS (variables.py:80-81)
      double lo1, la1, lo2, la2;
      proj_inv<true>(p, x, y, lo1, la1);
      proj_inv<true>(p, x, y + 0.1, lo2, la2);
      return -geod_inverse_azimuth(la1, lo1, la2, lo2);
    }
  }
  double phim, dphi, dlam;
  proj_inv_diff<EXT>(p, x, y, 10.0, phim, dphi, dlam);
  const GeodConst &g = c_geod;
  double sphi, cphi;
  sincos(phim, &sphi, &cphi);
  double w2 = 1 - g.e2 * sphi * sphi, w = sqrt(w2);
  double M = g.a * (1 - g.e2) / (w2 * w), N = g.a / w;
  double az_mid = atan2(dlam * N * cphi, dphi * M);
  double az1 = az_mid - 0.5 * dlam * sphi;
  return -az1;  // rot_angle_rad = -rot_angle_vectors_rad
}

// cos / sin of rot_angle_rad = -(azimuth of the reader's +y axis) without any transcendental call
// for the polar ellipsoidal stereographic projection: with t = tan(pi/4 - chi/2) = rho/akm1 the
// differences of the two inverse projections are
//   dlam = atan2(X (Y1 - Y2), X^2 + Y1 Y2),   dchi = -2 atan((t2 - t1) / (1 + t1 t2)),
//   dphi = dchi (1 + sum 2k c_k cos 2k chi_m)                       (Snyder eq. 3-5 differentiated)
// both atans have arguments ~1e-6 (10 m over the earth radius -> Gregory series), sin/cos of chi_m are
// rational in t_m, sin/cos of the multiples by recurrence, and the geodetic mid latitude is
// chi_m + delta with |delta| < 3.4e-3 (angle-addition with a short Taylor series).
template <bool ELLPOLAR = false, bool EXT = true>
__device__ __forceinline__ void rotation_cs(const DevProj &p, double x, double y, double &cs, double &sn) {
#pragma clang fp contract(fast)
  if (!ELLPOLAR && !(p.kind == PROJ_STERE_POLAR && p.es != 0)) {
    double rot = rotation_angle<EXT>(p, x, y);
    sincos(rot, &sn, &cs);
    return;
  }
#ifndef ODR_ROT_CS_SERIES   // (-DODR_ROT_CS_SERIES: the difference-of-inverses form of rounds 2-4 for every reader, below)
  if (rot_same_ellipsoid(p)) {
    // Round 5: the map is azimuthal and conformal on the geodesic's ellipsoid -- its +y axis has the azimuth +-(lambda - lambda0)
    // = the direction to the pole in the plane, and the 10 m line of rotate_vectors starts half its change of longitude times
    // (1 - |sin phi|) off that (proj_fwd_near); sin phi from the conformal latitude of rho with the first term of Snyder 3-5.
    // 5e-10 rad from the reference's recipe on the oracle (4 000 random points per hemisphere, 40-89.9 deg); ~40 instructions.
    const double inva = fast_rcp(p.a);
    const double X = (x - p.x0) * inva, Y = (y - p.y0) * inva;
    const double h2 = X * X + Y * Y;
    if (!(h2 > 0)) { cs = 1.0; sn = 0.0; return; }
    const double ir = fast_rsqrt(h2), rho = h2 * ir;
    const double sD = X * ir, cD = (p.south ? Y : -Y) * ir;                 // sin, cos of lambda - lambda0
    const double t = rho * fast_rcp(p.akm1), it = fast_rcp(1 + t * t);
    const double sch = (1 - t * t) * it, cch = 2 * t * it;                  // sin, cos of the conformal latitude
    const double sphi = sch + cch * (p.cchi[0] * 2 * sch * cch);
    const double q = 5.0 * sD * (1 - sphi) * ir * inva;
    if (p.south) { cs = cD + q * sD; sn = sD - q * cD; }
    else { cs = cD - q * sD; sn = -sD - q * cD; }
    return;
  }
#endif
  const GeodConst &g = c_geod;
  const double inva = fast_rcp(p.a);
  double X = (x - p.x0) * inva, Y1 = (y - p.y0) * inva, Y2 = (y + 10.0 - p.y0) * inva;
  if (!p.south) { Y1 = -Y1; Y2 = -Y2; }
  double r1 = fast_sqrt(X * X + Y1 * Y1), r2 = fast_sqrt(X * X + Y2 * Y2);
  double dlam = atan_ratio(X * (Y1 - Y2), X * X + Y1 * Y2);
  double ik = fast_rcp(p.akm1);
  double t1 = r1 * ik, t2 = r2 * ik;
  double dt = (Y2 - Y1) * (Y2 + Y1) * fast_rcp(r1 + r2) * ik;
  double dchi = -2 * atan_ratio(dt, 1 + t1 * t2);
  double tm = 0.5 * (t1 + t2), q = fast_rcp(1 + tm * tm);
  double sch = (1 - tm * tm) * q, cch = 2 * tm * q;            // sin, cos of chi_m
  double c2 = cch * cch - sch * sch, s2 = 2 * sch * cch;       // cos, sin of 2 chi_m
  double c4 = 2 * c2 * c2 - 1, s4 = 2 * s2 * c2;
  double c6 = 2 * c2 * c4 - c2, s6 = 2 * c2 * s4 - s2;
  double c8 = 2 * c2 * c6 - c4, s8 = 2 * c2 * s6 - s4;
  double dphi = dchi * (1 + 2 * p.cchi[0] * c2 + 4 * p.cchi[1] * c4 + 6 * p.cchi[2] * c6 + 8 * p.cchi[3] * c8);
  double dl = p.cchi[0] * s2 + p.cchi[1] * s4 + p.cchi[2] * s6 + p.cchi[3] * s8;   // phi_m - chi_m
  double d2 = dl * dl;
  double sd = dl * (1 - d2 * (1.0 / 6 - d2 * (1.0 / 120))), cd = 1 - d2 * (0.5 - d2 * (1.0 / 24));
  double sphi = sch * cd + cch * sd, cphi = cch * cd - sch * sd;
  if (p.south) { sphi = -sphi; dphi = -dphi; }
  double w2 = 1 - g.e2 * sphi * sphi, iw = fast_rsqrt(w2);
  double N = g.a * iw, M = g.a * (1 - g.e2) * iw * iw * iw;
  double se = dlam * N * cphi, snn = dphi * M;
  double ih = fast_rsqrt(se * se + snn * snn);
  double saz = se * ih, caz = snn * ih;                         // azimuth of the 10 m chord at its mid point
  double eps = 0.5 * dlam * sphi;                               // half the meridian convergence
  double ce = 1 - 0.5 * eps * eps;
  cs = caz * ce + saz * eps;                                    // cos(-(az_mid - eps))
  sn = -(saz * ce - caz * eps);                                 // sin(-(az_mid - eps))
}

// Footprint of scipy.ndimage.map_coordinates(order=1) along one axis in the form the reference
// finally delivers: a NaN of the first (mode='constant') pass is always retried with
// mode='nearest' (interpolators.py:122-137), and on finite data both modes agree, so the device
// uses the 'nearest' footprint throughout: start = floor(c), t = c - start, both indices clamped
// to [0, n-1] (the coordinate itself is not clamped -- probed against SciPy 1.15.3).
struct Axis { int i0, i1; double t; };
__device__ __forceinline__ Axis axis_fp(double c, int n) {
  Axis a;
  double f = floor(c);
  a.t = c - f;
  f = fmin(fmax(f, -1.0), (double)n);
  int i = (int)f;
  a.i0 = min(max(i, 0), n - 1);
  a.i1 = min(max(i + 1, 0), n - 1);
  return a;
}

// ------------------------------------------------------ gridded block sampling
// scipy.ndimage.map_coordinates(order=1) on a float32 layer that was pre-dilated 10x at
// upload: coordinates are clamped (the reference's retry pass uses mode='nearest'), the
// 2x2 footprint accumulates (v*wy)*wx in float64 in row-major order and rounds to float32.
__device__ __forceinline__ float bilinear_f32(const float *__restrict__ a, int ny, int nx,
                                              size_t ns, double yi, double xi) {
  Axis ay = axis_fp(yi, ny), ax = axis_fp(xi, nx);
  const float *r0 = a + (size_t)ay.i0 * nx * ns, *r1 = a + (size_t)ay.i1 * nx * ns;
  double v00 = r0[ax.i0 * ns], v01 = r0[ax.i1 * ns], v10 = r1[ax.i0 * ns], v11 = r1[ax.i1 * ns];
  double wy0 = 1 - ay.t, wx0 = 1 - ax.t;
  double t = __dmul_rn(__dmul_rn(v00, wy0), wx0);
  t = __dadd_rn(t, __dmul_rn(__dmul_rn(v01, wy0), ax.t));
  t = __dadd_rn(t, __dmul_rn(__dmul_rn(v10, ay.t), wx0));
  t = __dadd_rn(t, __dmul_rn(__dmul_rn(v11, ay.t), ax.t));
  return (float)t;
}

// The index maps of the 2-D interpolators -- (x - xgrid[0]) / (xgrid[-1] - xgrid[0]) * (len - 1), interpolators.py:110-111, and
// round((x - xgrid.min()) / (xgrid.max() - xgrid.min()) * len), :32-37 -- in FLOAT32 arithmetic: what NumPy evaluates in a run's
// first get_environment, when the elements' lon / lat are float32 arrays (elements.py:71-88), on a geographic reader (x IS the
// longitude) whose coordinate arrays are float32 as well (DevSource::xy_f32).  The Python int is weak: float32 throughout.
__device__ __forceinline__ double index_f32(double v, double v0, double span, int n) {
  return (double)__fmul_rn(__fdiv_rn(__fsub_rn((float)v, (float)v0), (float)span), (float)n);
}
__device__ __forceinline__ int nearest_index_f32(double v, double vmin, double vrange, int n) {
  const float r = rintf((float)index_f32(v, vmin, vrange, n));
  if (!(r >= 0) || r >= (float)n) return n - 1;
  return (int)r;
}

__device__ __forceinline__ int nearest_index(double v, double vmin, double vrange, double ivrange, int n) {
  double r = rint(__dmul_rn(div_cr(v - vmin, vrange, ivrange), (double)n));
  if (!(r >= 0) || r >= n) return n - 1;
  return (int)r;
}

// Linear1DInterpolator (interpolators.py:174-197): scipy interp1d(zgrid -> index), int8 floor.
// The interval is found by counting the (wave-uniform) levels below z; its node, slope and index
// value come from the per-source tables the host filled with the same IEEE operations interp1d
// performs -- no per-particle divide and no dependent table walk.
// ZT: the three interp1d tables of the source come from the workgroup's LDS copy `zt` ([0] zi_x, [MAXNZ] zi_slope,
// [2 MAXNZ] zi_y; zt_stage()) instead of the source in global memory -- the index is per lane, so the global form is a
// vector-memory round trip in front of the gathers that need the bracket.  The level count reads the (wave-uniform) levels
// four per scalar load; zasc is padded with +inf beyond nz (round 3: a scalar load + wait per level and the table gather
// were 3 500 cycles of a wave's life in k_step_grid).
// ZT_STRIDE: levels of the LDS copy.  40 (the ROMS z list has 35): with 3 x 64 entries the step kernel's workgroup took 32 256 B
// of LDS -- 26 allocation granules of 1 280 B, four workgroups per CU where its 96 registers allow five.
constexpr int ZT_STRIDE = 40;
template <bool ZT = false>
__device__ __forceinline__ void zinterp(const DevSource &s, double z, int &ia, int &ib, double &wa, const double *zt = nullptr) {
  const int nz = s.nz;
  const double zmin = s.zasc[0], zmax = s.zasc[nz - 1];  // zgrid.min()/max(): monotone grids only
  z = z < zmin ? zmin : z;
  z = z > zmax ? zmax : z;
  int hi = 0;
  for (int k = 0; k < nz; k += 4) {
    const double a0 = s.zasc[k], a1 = s.zasc[k + 1], a2 = s.zasc[k + 2], a3 = s.zasc[k + 3];   // MAXNZ is a multiple of 4
    hi += (a0 < z ? 1 : 0) + (a1 < z ? 1 : 0) + (a2 < z ? 1 : 0) + (a3 < z ? 1 : 0);
  }
  hi = hi < 1 ? 1 : hi;
  hi = hi > nz - 1 ? nz - 1 : hi;
  const int lo = hi - 1;
  double tx, ts, tyv;
  bool from_lds = false;
  if constexpr (ZT) from_lds = nz <= ZT_STRIDE;      // (wave-uniform; readers with more levels than the LDS copy holds read the source)
  if (from_lds) { tx = zt[lo]; ts = zt[ZT_STRIDE + lo]; tyv = zt[2 * ZT_STRIDE + lo]; }
  else { tx = s.zi_x[lo]; ts = s.zi_slope[lo]; tyv = s.zi_y[lo]; }
  double zi = __dadd_rn(__dmul_rn(ts, z - tx), tyv);
  ia = (int)(signed char)(long long)floor(zi);
  ia = ia < 0 ? 0 : ia;
  ib = ia + 1 < nz - 1 ? ia + 1 : nz - 1;
  wa = 1 - (zi - ia);
}

// value of one variable from one block; f32class = the reference hands back float32 (2D layer)
__device__ __forceinline__ double block_value(const DevBlock &b, const DevSource &s, int var,
                                              double x, double y, double z, bool &f32class, int rank = 0, int f32idx = 0) {
  const float *d = b.data[var];
  int nzv = b.var_nz[var];
  const int es = b.es[var];
  const size_t ns = (size_t)b.rec;
  if (s.members[var] > 1) {   // ensemble data: this element's member, `nzv` levels of it
    nzv /= s.members[var];
    d += (size_t)(rank % s.members[var]) * (size_t)nzv * (size_t)es;
  }
  if (var == VAR_LAND) {
    f32class = true;
    int xi = (f32idx & 1) ? nearest_index_f32(x, b.xmin, b.xrange, b.nx) : nearest_index(x, b.xmin, b.xrange, b.ixrange, b.nx);
    int yi = (f32idx & 2) ? nearest_index_f32(y, b.ymin, b.yrange, b.ny) : nearest_index(y, b.ymin, b.yrange, b.iyrange, b.ny);
    return d[((size_t)yi * b.nx + xi) * ns];
  }
  double xi = (f32idx & 1) ? index_f32(x, b.x0, b.xspan, b.nx - 1) : __dmul_rn(div_cr(x - b.x0, b.xspan, b.ixspan), (double)(b.nx - 1));
  double yi = (f32idx & 2) ? index_f32(y, b.y0, b.yspan, b.ny - 1) : __dmul_rn(div_cr(y - b.y0, b.yspan, b.iyspan), (double)(b.ny - 1));
  if (nzv <= 1) {
    f32class = true;
    return bilinear_f32(d, b.ny, b.nx, ns, yi, xi);
  }
  f32class = false;
  int ia, ib;
  double wa;
  zinterp(s, z, ia, ib, wa);
  double va = bilinear_f32(d + (size_t)ia * es, b.ny, b.nx, ns, yi, xi);
  double vb = bilinear_f32(d + (size_t)ib * es, b.ny, b.nx, ns, yi, xi);
  return __dadd_rn(__dmul_rn(va, wa), __dmul_rn(vb, 1 - wa));
}

// time bracket on the resident levels: nearest_time (variables.py:402-443)
__device__ __forceinline__ void bracket(const DevSource &s, double t, int &ib, int &ia) {
  int b = 0;
  for (int k = 0; k < s.nlevels; ++k)
    if (s.slot[s.level_slot[k]].t <= t) b = k;
  ib = s.level_slot[b];
  ia = (b + 1 < s.nlevels && s.slot[ib].t != t) ? s.level_slot[b + 1] : -1;
}

// Landmask raster (the device image of reader_global_landmask.Reader, readers/reader_global_landmask.py:201-255: a
// ContinuousReader, exact at the element position): bit (iy * nx + ix) of the packed raster, cell
// ix = floor((lon - lon0) / dlon), iy = floor((lat - lat0) / dlat); ocean outside the raster.
// params = {lon0, lat0, dlon, dlat}; slot[0].data[VAR_LAND] = the packed words; slot[0].nx / ny = raster size.
__device__ __forceinline__ bool landmask_contains(const DevSource &s, double lon, double lat) {
  const double fx = floor(__ddiv_rn(__dsub_rn(lon, s.params[0]), s.params[2]));
  const double fy = floor(__ddiv_rn(__dsub_rn(lat, s.params[1]), s.params[3]));
  const int nx = s.slot[0].nx, ny = s.slot[0].ny;
  if (!(fx >= 0 && fx < (double)nx && fy >= 0 && fy < (double)ny)) return false;
  const size_t bit = (size_t)fy * (size_t)nx + (size_t)fx;
  const unsigned *w = (const unsigned *)s.slot[0].data[VAR_LAND];
  return (w[bit >> 5] >> (bit & 31)) & 1u;
}

// One reader.get_variables_interpolated (variables.py:860-920) for one particle and the
// covers_positions_xy (variables.py:170-215, 747): the position in the reader's own coordinates and whether its domain
// holds it -- the elements a reader is HANDED are the covered ones (ind_covered), which is also what the ensemble members
// of a ReaderBlock are numbered over (interpolation/structured.py:119-135; k_rank_mark)
__device__ __forceinline__ bool source_covers_xyz(const DevSource &s, double lon, double lat, double z, double &x, double &y,
                                                  int f32pos = 0) {
  if (f32pos) lon = lon_f32class(s.lon_mode, lon);
  if (s.lon_mode == 1) lon = np_mod(lon + 180.0, 360.0) - 180.0;
  else if (s.lon_mode == 2) lon = np_mod(lon, 360.0);
  proj_fwd_rt(s.proj, lon, lat, x, y);
  double xchk = x;
  if (s.proj.kind == PROJ_LATLONG || s.proj.kind == PROJ_OB_TRAN) {
    if (s.lon_mode == 1) xchk = np_mod(x + 180.0, 360.0) - 180.0;
    else if (s.lon_mode == 2) xchk = np_mod(x, 360.0);
  }
  return xchk >= s.xmin && xchk <= s.xmax && y >= s.ymin && y <= s.ymax && z >= s.zmin && z <= s.zmax;
}

// NV variables of a group.  Returns false when the reader does not cover the position.
template <int NV>
__device__ __forceinline__ bool source_sample(const DevSource &s, const int (&vars)[NV], double lon,
                                              double lat, double z, double t, double (&val)[NV], int rank = 0, int f32pos = 0) {
  if (!s.always_valid && (t < s.tmin || t > s.tmax)) return false;  // OutsideTemporalCoverageError
  double x, y;
  if (!source_covers_xyz(s, lon, lat, z, x, y, f32pos)) return false;
  if (s.kind == SRC_CONSTANT) {
#pragma unroll
    for (int v = 0; v < NV; ++v) val[v] = s.const_val[vars[v]];
  } else if (s.kind == SRC_OSCILLATING) {
    double phase = ((t - s.params[3]) / s.params[2]) * kPi;
    double value = s.params[1] * sin(phase);
#pragma unroll
    for (int v = 0; v < NV; ++v) val[v] = value;
  } else if (s.kind == SRC_LANDMASK) {
    const double land = landmask_contains(s, x, y) ? 1.0 : 0.0;
#pragma unroll
    for (int v = 0; v < NV; ++v) val[v] = vars[v] == VAR_LAND ? land : __builtin_nan("");
  } else if (s.kind == SRC_DOUBLE_GYRE) {
    // reader_double_gyre.get_variables (reader_double_gyre.py:55-79)
    double A = s.params[0], eps = s.params[1], om = s.params[2], tt = t - s.params[3];
    double sn = sin(om * tt);
    double a = eps * sn, b = 1 - 2 * eps * sn;
    double f = a * x * x + b * x, dfdx = 2 * a * x + b;
    double sf, cf, sy, cy;
    sincos(kPi * f, &sf, &cf);
    sincos(kPi * y, &sy, &cy);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (vars[v] == VAR_U) val[v] = -kPi * A * sf * cy;
      else if (vars[v] == VAR_V) val[v] = kPi * A * cf * sy * dfdx;
      else val[v] = 0.0;
    }
  } else {
    // StructuredReader._get_variables_interpolated_ (structured.py:202-400)
    if (s.mod360_x) x = np_mod(x, 360.0);
    const int f32idx = (f32pos && s.proj.kind == PROJ_LATLONG) ? s.xy_f32 : 0;    // (index_f32)
    int ib, ia;
    bracket(s, t, ib, ia);
    bool all_static = true;
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (vars[v] != VAR_LAND && vars[v] != VAR_DEPTH) all_static = false;
    if (all_static || s.always_valid) ia = -1;
    const DevBlock &bb = s.slot[ib];
    if (ia < 0) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        bool f32c;
        val[v] = block_value(bb, s, vars[v], x, y, z, f32c, rank, f32idx);
      }
    } else {
      const DevBlock &ba = s.slot[ia];
      double w = __ddiv_rn(t - bb.t, ba.t - bb.t);  // structured.py:353-354
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        bool fb, fa;
        double vb = block_value(bb, s, vars[v], x, y, z, fb, rank, f32idx);
        double va = block_value(ba, s, vars[v], x, y, z, fa, rank, f32idx);
        if (fb && fa) {  // float32 arrays * python floats stay float32 (:362-364)
          float pq = __fadd_rn(__fmul_rn((float)vb, (float)(1 - w)), __fmul_rn((float)va, (float)w));
          val[v] = pq;
        } else {
          val[v] = __dadd_rn(__dmul_rn(vb, 1 - w), __dmul_rn(va, w));
        }
      }
    }
  }
  // rotate x/y vector pairs to the lon/lat CRS (variables.py:799-837)
  if (ODR_PROJ_ROTATES(s.proj.kind)) {
    bool need = false;
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (vars[v] == VAR_U || vars[v] == VAR_XWIND || vars[v] == VAR_SX || vars[v] == VAR_ICE_U) need = true;
    if (need) {
      double sn, cs;
      rotation_cs(s.proj, x, y, cs, sn);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        int partner = vars[v] == VAR_U ? VAR_V : vars[v] == VAR_XWIND ? VAR_YWIND
                                              : vars[v] == VAR_SX ? VAR_SY : vars[v] == VAR_ICE_U ? VAR_ICE_V : -1;
        if (partner < 0) continue;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
          if (vars[u] != partner) continue;
          double uu = val[v], vv = val[u];
          val[v] = __dsub_rn(__dmul_rn(uu, cs), __dmul_rn(vv, sn));
          val[u] = __dadd_rn(__dmul_rn(uu, sn), __dmul_rn(vv, cs));
        }
      }
    }
  }
  return true;
}

// "Some extra checks of units" at the end of get_environment (environment.py:829-838): a sea_water_temperature above
// 100 is taken as Kelvin.  env is a masked float32 recarray there: the difference is formed in float64 and stored
// as float32 (golden c12).
__device__ __forceinline__ float kelvin_to_celsius(float T) {
  return T > 100.f ? (float)__dsub_rn((double)T, 273.15) : T;
}

// Environment.get_environment for one particle and one variable group (variables that
// share a priority list): walk the readers until every variable is finite
// (environment.py:597-762), then the fallback (:782-791).  out = float32 environment.
template <int NV>
__device__ __forceinline__ void env_group(const DevWorld &W, const int (&vars)[NV], double lon,
                                          double lat, double z, double t, float (&out)[NV], int rank = 0, int f32pos = 0) {
#pragma unroll
  for (int v = 0; v < NV; ++v) out[v] = W.fallback[vars[v]];
  int nl = W.nlist[vars[0]];
  for (int k = 0; k < nl; ++k) {
    const DevSource &s = W.src[W.list[vars[0]][k]];
    double val[NV];
    bool covered = source_sample<NV>(s, vars, lon, lat, z, t, val, rank, f32pos);
    bool bad = !covered;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float f = covered ? (float)val[v] : __builtin_nanf("");
      out[v] = f;  // masked_invalid(...).astype('float32') overwrites every group variable
      if (!isfinite(f)) bad = true;
    }
    if (!bad) break;
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    if (!isfinite(out[v]) && isfinite(W.fallback[vars[v]])) out[v] = W.fallback[vars[v]];
    if (vars[v] == VAR_TEMP) out[v] = kelvin_to_celsius(out[v]);
  }
}


// ------------------------------------------------------- analytic double gyre, fast path
// reader_double_gyre (readers/reader_double_gyre.py:24-79) as the only source of the current: equatorial
// stereographic projection on a sphere.  Same arithmetic as source_sample() to round-off, organised for the
// vector pipe: sin(omega t) is wave-uniform (host), sin / cos of the small lon / lat by the reduced-range
// kernels, sin / cos of pi f and pi y by sincospi, and the rotation of the vector to east / north
// (rotate_vectors, variables.py:59-109) WITHOUT any transcendental call: for the 10 m line (x,y) -> (x,y+10)
// the differences of the two inverse projections are, at the mid point, d(sin phi)/dy and d(tan lambda)/dy in
// closed form (sin c and cos c of the inverse are rational in t = rho / 2k0), central differences being exact
// to O((10 m / R)^3); the WGS84 Gauss mid-latitude azimuth is then a normalised 2-vector plus a first-order
// rotation by the half meridian convergence.
struct GyreTime { double sn; };  // sin(omega (t - t0)) of one evaluation time

__device__ __forceinline__ void rotation_cs_equit(const DevProj &p, double x, double y, double &cs, double &sn) {
#pragma clang fp contract(fast)
  const GeodConst &g = c_geod;
  const double inva = fast_rcp(p.a), ik = fast_rcp(p.akm1);
  const double X = (x - p.x0) * inva, Y = (y + 5.0 - p.y0) * inva, h = 10.0 * inva;
  const double t2 = (X * X + Y * Y) * ik * ik, D = 1 + t2, iD = fast_rcp(D);
  const double S = 2 * Y * ik * iD;                                   // sin(phi_m)
  const double S_Y = 2 * ik * iD * (1 - 2 * Y * Y * ik * ik * iD);    // d sin(phi) / dY
  const double cphi = fast_sqrt(1 - S * S);
  const double dphi = S_Y * h * fast_rcp(cphi);
  const double P = 2 * X * ik, Q = 1 - t2;
  const double dlam = P * 2 * Y * ik * ik * h * fast_rcp(P * P + Q * Q);
  const double w2 = 1 - g.e2 * S * S, iw = fast_rsqrt(w2);
  const double N = g.a * iw, M = g.a * (1 - g.e2) * iw * iw * iw;
  const double east = dlam * N * cphi, north = dphi * M;
  const double ir = fast_rsqrt(east * east + north * north);
  const double sa = east * ir, ca = north * ir;                       // sin, cos of the mid-point azimuth
  const double d = -0.5 * dlam * S, cd = 1 - 0.5 * d * d;             // az1 = az_mid + d
  const double s1 = sa * cd + ca * d, c1 = ca * cd - sa * d;
  cs = c1;    // rot_angle_rad = -az1
  sn = -s1;
}

// (u, v) float32 environment of one particle from the double gyre source `s`
__device__ __forceinline__ bool gyre_sample(const DevSource &s, double lon, double lat, double z, double snw,
                                            float fbu, float fbv, float &uo, float &vo) {
#pragma clang fp contract(fast)
  if (s.lon_mode == 1) lon = np_mod(lon + 180.0, 360.0) - 180.0;
  else if (s.lon_mode == 2) lon = np_mod(lon, 360.0);
  const DevProj &p = s.proj;
  double lam = wrap_pi(lon * kDeg - p.lon0), phi = lat * kDeg;
  double sl, cl, sp, cp;
  sincos_small(lam, sl, cl);
  sincos_small(phi, sp, cp);
  const double k = p.akm1 * fast_rcp(1 + cp * cl);
  const double x = p.a * (k * cp * sl) + p.x0, y = p.a * (k * sp) + p.y0;
  float fu = __builtin_nanf(""), fv = __builtin_nanf("");
  const bool covered = x >= s.xmin && x <= s.xmax && y >= s.ymin && y <= s.ymax && z >= s.zmin && z <= s.zmax;
  if (covered) {
    const double A = s.params[0], eps = s.params[1];
    const double a = eps * snw, b = 1 - 2 * eps * snw;
    const double f = a * x * x + b * x, dfdx = 2 * a * x + b;
    double sf, cf, sy, cy;
    sincospi(f, &sf, &cf);
    sincospi(y, &sy, &cy);
    double u = -kPi * A * sf * cy, v = kPi * A * cf * sy * dfdx;
    double cs, sn;
    rotation_cs_equit(p, x, y, cs, sn);
    fu = (float)__dsub_rn(__dmul_rn(u, cs), __dmul_rn(v, sn));
    fv = (float)__dadd_rn(__dmul_rn(u, sn), __dmul_rn(v, cs));
  }
  uo = isfinite(fu) ? fu : (isfinite(fbu) ? fbu : fu);
  vo = isfinite(fv) ? fv : (isfinite(fbv) ? fbv : fv);
  return covered;
}

// ------------------------------------------------------------------ fast (u,v) path
// The common case of the RK sub-stages: x/y_sea_water_velocity come from ONE gridded reader
// (plus the fallback constant).  The time bracket of each stage is the same for every
// particle, so the host resolves it (UVTime) and the kernel receives the two interleaved
// z-innermost arrays directly; the vertical bracket depends on z only and is computed once
// per particle.  One 16-byte load returns (u,v) at two adjacent z levels of a grid node
// (3D) or at two adjacent x nodes of a row (2D): 8 (3D) / 4 (2D) loads per field evaluation
// instead of 32 / 16 scalar gathers.  Arithmetic and rounding points are those of
// block_value()/source_sample() above.
struct UVTime {
  const float *b, *a;  // interleaved (u,v) arrays of the bracketing time levels (a == nullptr: on time)
  double w;            // weight_after (structured.py:353-354)
};
struct __attribute__((aligned(4))) F4 { float x, y, z, w; };

struct ZBracket { int iz0, same; double wa; };  // levels (ia, ib): ia = iz0 + (same?1:0) ...
// the workgroup's copy of the interp1d tables of source `s` (call from every thread, before any divergence)
__device__ __forceinline__ void zt_stage(const DevSource &s, double *zt) {
  const int t = threadIdx.x;
  if (t < s.nz && t < ZT_STRIDE) { zt[t] = s.zi_x[t]; zt[ZT_STRIDE + t] = s.zi_slope[t]; zt[2 * ZT_STRIDE + t] = s.zi_y[t]; }
  __syncthreads();
}
template <bool ZT = false>
__device__ __forceinline__ ZBracket zbracket(const DevSource &s, double z, const double *zt = nullptr) {
  ZBracket zb;
  int ia, ib;
  zinterp<ZT>(s, z, ia, ib, zb.wa, zt);
  zb.same = ia == ib;                       // clamped at the deepest level
  zb.iz0 = zb.same ? (ia > 0 ? ia - 1 : 0) : ia;
  return zb;
}

__device__ __forceinline__ float bil4(double v00, double v01, double v10, double v11, double wy0,
                                      double ty, double wx0, double tx) {
#ifdef ODR_ABL_BIL4   // what-if build (wrong values): the bilinear layer value without its eleven float64 operations
  return (float)v00 + (float)v11;
#endif
#ifdef ODR_WHATIF_BILW   // what-if build: weights multiplied out once per particle (the products are common subexpressions), fused
  return (float)__builtin_fma(v11, ty * tx, __builtin_fma(v10, ty * wx0, __builtin_fma(v01, wy0 * tx, v00 * (wy0 * wx0))));
#endif
  double t = __dmul_rn(__dmul_rn(v00, wy0), wx0);
  t = __dadd_rn(t, __dmul_rn(__dmul_rn(v01, wy0), tx));
  t = __dadd_rn(t, __dmul_rn(__dmul_rn(v10, ty), wx0));
  t = __dadd_rn(t, __dmul_rn(__dmul_rn(v11, ty), tx));
  return (float)t;
}

// 2x2 footprint of a particle in a block: byte offsets of the four node records and the
// weights -- shared by every variable and both time levels of a reader call.  24-bit multiplies
// (full rate on CDNA; the host guarantees node count < 2^24 and block size < 4 GiB on this path).
struct Foot {
  unsigned o00, o01, o10, o11;
  double wy0, ty, wx0, tx;
  unsigned n00, n11;   // node numbers of the first and the last corner in the block: the identity of the footprint (UVKeep)
};
__device__ __forceinline__ Foot footprint(double yi, double xi, int ny, int nx, unsigned rec_bytes) {
  const Axis ay = axis_fp(yi, ny), ax = axis_fp(xi, nx);
  const unsigned r0 = __umul24((unsigned)ay.i0, (unsigned)nx), r1 = __umul24((unsigned)ay.i1, (unsigned)nx);
  Foot f;
  f.n00 = r0 + (unsigned)ax.i0; f.n11 = r1 + (unsigned)ax.i1;
  f.o00 = __umul24(r0 + (unsigned)ax.i0, rec_bytes); f.o01 = __umul24(r0 + (unsigned)ax.i1, rec_bytes);
  f.o10 = __umul24(r1 + (unsigned)ax.i0, rec_bytes); f.o11 = __umul24(r1 + (unsigned)ax.i1, rec_bytes);
  f.ty = ay.t; f.tx = ax.t; f.wy0 = 1 - ay.t; f.wx0 = 1 - ax.t;
  return f;
}
struct __attribute__((aligned(4))) F2 { float x, y; };
// load at a 32-bit byte offset from a wave-uniform base (scalar base + vector offset addressing)
template <typename T>
__device__ __forceinline__ T ld_off(const float *__restrict__ base, unsigned byte_off) {
#ifdef ODR_ABLATE_LOADS   // what-if build: no field gathers (tools/ab_bench.sh)
  T t;
  float *f = (float *)&t;
  for (unsigned k = 0; k < sizeof(T) / 4; ++k) f[k] = (float)(byte_off & 1023u) * 1e-4f + (float)k;
  return t;
#else
#ifdef ODR_FLAT_GATHERS
  return *(const T *)((const char *)base + byte_off);
#else
  // buffer addressing: descriptor of the wave-uniform base in scalar registers + the 32-bit offset as it is -- no
  // 64-bit address arithmetic per gather (the flat form costs one v_lshl_add_u64 per load: 503 in k_step_grid<RK4>)
  static_assert(sizeof(T) == 4 || sizeof(T) == 8 || sizeof(T) == 16, "gather width");
#ifdef ODR_ABLATE_UNIFORM_GATHER   // what-if build: every lane reads lane 0's address (one L1 access per gather instead of up to 64; wrong values)
  byte_off = (unsigned)__builtin_amdgcn_readfirstlane((int)byte_off);
#endif
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0xffffffff, 0x00020000);
  if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
  else if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
  else return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
#endif
#endif
}

// ------------------------------------------------------------------ where a sample reads its node records
// A sampler reads the records of a particle's 2x2 footprint at (up to) two time levels through a LOADER: LdGlobal -- the
// blocks in HBM through buffer descriptors (wave-uniform base + 32-bit byte offset) -- or LdTile -- the workgroup's copy
// of the node rectangle around its particles in LDS (k_step_tile: the whole records of the rectangle, both time levels,
// fetched once by LDS-DMA; odr_tile.hip.h).  A loader also makes the footprint: the SAME indices and weights, the byte
// offsets relative to its own image; LdTile reports footprints that leave its rectangle (ok = false: the particle is
// redone on the global path, nothing of it has been written by then).  Same values, same arithmetic, same bits.
typedef __attribute__((address_space(3))) const char lds_cbyte;   // explicit LDS pointer: ds_read, not flat loads
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <class T, int ALIGN>
__device__ __forceinline__ T lds_ld(lds_cbyte *q) {
  typedef __attribute__((address_space(3))) const float lds_cf32;
  typedef __attribute__((address_space(3))) const f32x2 lds_cf64;   // 8-byte aligned: ds_read_b64
  if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, *(lds_cf32 *)q);
  else if constexpr (sizeof(T) == 8 && ALIGN >= 8) return __builtin_bit_cast(T, *(lds_cf64 *)q);
  else if constexpr (sizeof(T) == 8) { F2 r; r.x = ((lds_cf32 *)q)[0]; r.y = ((lds_cf32 *)q)[1]; return __builtin_bit_cast(T, r); }
  else if constexpr (ALIGN >= 8) {   // 16 bytes at an 8-byte boundary: two ds_read_b64 (a ds_read_b128 off its alignment is replayed)
    const f32x2 lo = ((lds_cf64 *)q)[0], hi = ((lds_cf64 *)q)[1];
    F4 r; r.x = lo.x; r.y = lo.y; r.z = hi.x; r.w = hi.y;
    return __builtin_bit_cast(T, r);
  } else {
    F4 r; r.x = ((lds_cf32 *)q)[0]; r.y = ((lds_cf32 *)q)[1]; r.z = ((lds_cf32 *)q)[2]; r.w = ((lds_cf32 *)q)[3];
    return __builtin_bit_cast(T, r);
  }
}
struct LdGlobal {
  const float *b, *a;     // records (or one variable of them) at the bracketing time levels; a == b when there is one level
  template <class T, int ALIGN = 4>
  __device__ __forceinline__ T ld(int time, unsigned off) const { return ld_off<T>(time ? a : b, off); }
  __device__ __forceinline__ Foot foot(double yi, double xi, int ny, int nx, unsigned rec_bytes, bool &ok) const {
    ok = true;
    return footprint(yi, xi, ny, nx, rec_bytes);
  }
  __device__ __forceinline__ unsigned node(int iy, int ix, int nx, unsigned rec_bytes, bool &ok) const {
    ok = true;
    return __umul24(__umul24((unsigned)iy, (unsigned)nx) + (unsigned)ix, rec_bytes);
  }
  __device__ __forceinline__ LdGlobal at(unsigned byte_off) const { LdGlobal r; r.b = (const float *)((const char *)b + byte_off); r.a = (const float *)((const char *)a + byte_off); return r; }
};
struct TileRect { int x0, y0, w, h; };   // node rectangle [x0, x0 + w) x [y0, y0 + h) of the workgroup's tile
struct LdTile {
  lds_cbyte *b, *a;       // images of the two time levels: [h][w] node records each
  TileRect R;
  template <class T, int ALIGN = 4>
  __device__ __forceinline__ T ld(int time, unsigned off) const { return lds_ld<T, ALIGN>((time ? a : b) + off); }
  __device__ __forceinline__ Foot foot(double yi, double xi, int ny, int nx, unsigned rec_bytes, bool &ok) const {
    const Axis ay = axis_fp(yi, ny), ax = axis_fp(xi, nx);
    const int lx0 = ax.i0 - R.x0, lx1 = ax.i1 - R.x0, ly0 = ay.i0 - R.y0, ly1 = ay.i1 - R.y0;
    ok = lx0 >= 0 && lx1 < R.w && ly0 >= 0 && ly1 < R.h;
    const unsigned r0 = ok ? (unsigned)(ly0 * R.w) : 0u, r1 = ok ? (unsigned)(ly1 * R.w) : 0u;
    const unsigned c0 = ok ? (unsigned)lx0 : 0u, c1 = ok ? (unsigned)lx1 : 0u;
    Foot f;
    f.n00 = __umul24((unsigned)ay.i0, (unsigned)nx) + (unsigned)ax.i0; f.n11 = __umul24((unsigned)ay.i1, (unsigned)nx) + (unsigned)ax.i1;
    f.o00 = __umul24(r0 + c0, rec_bytes); f.o01 = __umul24(r0 + c1, rec_bytes);
    f.o10 = __umul24(r1 + c0, rec_bytes); f.o11 = __umul24(r1 + c1, rec_bytes);
    f.ty = ay.t; f.tx = ax.t; f.wy0 = 1 - ay.t; f.wx0 = 1 - ax.t;
    return f;
  }
  __device__ __forceinline__ unsigned node(int iy, int ix, int nx, unsigned rec_bytes, bool &ok) const {
    const int lx = ix - R.x0, ly = iy - R.y0;
    ok = lx >= 0 && lx < R.w && ly >= 0 && ly < R.h;
    return ok ? __umul24((unsigned)(ly * R.w + lx), rec_bytes) : 0u;
  }
  __device__ __forceinline__ LdTile at(unsigned byte_off) const { LdTile r = *this; r.b = b + byte_off; r.a = a + byte_off; return r; }
};
// the four corner offsets of a footprint + a loader: what uv_level_ld reads an interleaved (u,v) pair through
template <class LD>
struct PairLd {
  LD ld;
  unsigned o[4], kb;
  __device__ __forceinline__ F4 q4(int time, int c) const { return ld.template ld<F4, 8>(time, o[c] + kb); }
  __device__ __forceinline__ F2 q2(int time, int c) const { return ld.template ld<F2, 8>(time, o[c]); }
};
// ---- the records of a (u,v) footprint, kept between the samples of one particle-step.  The Runge-Kutta stage positions of
// advect_ocean_current (physics_methods.py:638-670) are a fraction of a cell away from the element: most of them have the
// SAME 2x2 footprint, levels and time bracket as the main-loop sample of the step (or as the previous stage) -- the eight
// corner records are then the ones already in registers and only the weights change.  A sample whose footprint differs
// fetches its records (lanes whose footprint is unchanged sit the gathers out: the texture addresser, which bounds
// k_step_grid, works per lane) and keeps them for the next stage.  Same floats, same arithmetic: bit-identical.
template <bool IS3D> struct UVCorner { typedef F4 T; };
template <> struct UVCorner<false> { typedef F2 T; };
template <bool IS3D>
struct UVRec { typename UVCorner<IS3D>::T b[4], a[4]; };   // 3-D: (u,v) at levels iz0, iz0 + 1; 2-D: (u,v)
template <bool IS3D>
struct UVKeep {
  UVRec<IS3D> q;
  unsigned n00, n11, kb;   // identity of the footprint the records belong to (Foot::n00 / n11, level offset)
  bool valid;
};
template <bool IS3D, class LD>
__device__ __forceinline__ void uv_fetch(const LD &ld, bool has_a, const Foot &ft, unsigned kb, UVRec<IS3D> &q) {
  const unsigned o[4] = {ft.o00, ft.o01, ft.o10, ft.o11};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if constexpr (IS3D) q.b[c] = ld.template ld<F4, 8>(0, o[c] + kb); else q.b[c] = ld.template ld<F2, 8>(0, o[c]);
  }
  if (has_a) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if constexpr (IS3D) q.a[c] = ld.template ld<F4, 8>(1, o[c] + kb); else q.a[c] = ld.template ld<F2, 8>(1, o[c]);
    }
  }
}
// the records of footprint `ft` for a stage sample: those kept in K when it is the same footprint (keep: the sample's time
// bracket is K's), fetched into K otherwise
// KEEPS (compile time, ODR_UV_KEEPS(PROJ)): readers without vector rotation.  Measured (profiles/r04_ab_variants.txt): C3
// (lon/lat, 3-D) 0.705 -> 0.645 ms per launch, texture-addresser busy -33 %; C4 (polar stereographic: the launch is bound by
// float64 issue, 142 registers) 0.546 -> 0.599 ms -- the projected readers fetch every sample.
#define ODR_UV_KEEPS(PROJ) (!ODR_PROJ_ROTATES(PROJ))
// Returns the records to sample from.  KEEPS: K.q itself, whatever the case -- a sample that may not keep (`keep` false: the
// full-step stage when its time bracket is not the half-step stages') fetches into K.q all the same and leaves K invalid; it is
// the last sample of the step.  (Round 5: a separate copy of the records next to the kept ones held 32 more registers through the
// stage phase -- the phase that sets the kernel's register count, and with it the waves per SIMD its launch time goes with.)
template <bool IS3D, bool KEEPS, class LD>
__device__ __forceinline__ const UVRec<IS3D> &uv_records(const LD &ld, bool has_a, const Foot &ft, unsigned kb, UVKeep<IS3D> &K, bool keep,
                                                         UVRec<IS3D> &scratch) {
#if defined(ODR_NO_KEEP) || defined(ODR_TU_TILE)   // A/B build; the LDS-tile kernels (their records are two cycles away, the registers are not there)
  keep = false;
#endif
  if constexpr (KEEPS) {
    const bool same = keep && K.valid && K.n00 == ft.n00 && K.n11 == ft.n11 && K.kb == kb;
    if (!same) {
      uv_fetch<IS3D>(ld, has_a, ft, kb, K.q);
      K.n00 = ft.n00; K.n11 = ft.n11; K.kb = kb; K.valid = keep;
    }
    return K.q;
  } else {
    uv_fetch<IS3D>(ld, has_a, ft, kb, scratch);
    return scratch;
  }
}
// ---- ODR_STAGE_FAST, 3-D readers: the kept records COMBINED over the two levels of the particle's vertical bracket.  The
// element's depth does not change between the samples of a step, so the bracket's weights (za, zw) are the same for all of them:
// what a stage sample needs of a corner is m = lo za + hi zw per time level -- (u, v) x 2 time levels = one F4 per corner, 16
// registers for the footprint instead of 32 (K.q.b[c] = {m_before.u, m_before.v, m_after.u, m_after.v}; K.q.a is not used).
// The stage value is then sum_c w_c m_c per time level and the time interpolation: the sum of round 4's FAST sample in another
// association (there: (sum_c w_c lo_c) za + (sum_c w_c hi_c) zw) -- float32 round-off apart, inside the same gate
// (tests/test_gpu_stage_math.py).  With the geodesic coefficients read from LDS where they are used this takes k_step_grid<RK4,
// lat/lon, 3-D, FAST> from 126 to <= 96 registers: 5 waves per SIMD, and its launch time goes with 1 / waves
// (profiles/r05_ab_variants.txt section 4).
__device__ __forceinline__ F4 uv_combine_z(const F4 &b, const F4 &a, bool has_a, float za, float zw) {
  F4 m;
  m.x = __builtin_fmaf(b.z, zw, b.x * za); m.y = __builtin_fmaf(b.w, zw, b.y * za);
  m.z = has_a ? __builtin_fmaf(a.z, zw, a.x * za) : m.x; m.w = has_a ? __builtin_fmaf(a.w, zw, a.y * za) : m.y;
  return m;
}
template <class LD>
__device__ __forceinline__ void uv_fetch_combined(const LD &ld, bool has_a, const Foot &ft, unsigned kb, float za, float zw, UVRec<true> &q) {
  const unsigned o[4] = {ft.o00, ft.o01, ft.o10, ft.o11};
  F4 b[4], a[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) b[c] = ld.template ld<F4, 8>(0, o[c] + kb);
  if (has_a) {
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = ld.template ld<F4, 8>(1, o[c] + kb);
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = b[c];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) q.b[c] = uv_combine_z(b[c], a[c], has_a, za, zw);
}
__device__ __forceinline__ void uv_zweights(const ZBracket &zb, int nz, float &za, float &zw) {
  // levels (iz0, iz0 + 1) = ("above", "below"); clamped at the deepest level both are iz0 + 1
  const bool same = zb.same && nz > 1;
  za = same ? 0.f : (float)zb.wa;
  zw = 1.f - za;
}

// uv_level_ld's view of fetched records
template <bool IS3D>
struct RecLd {
  const UVRec<IS3D> &q;
  __device__ __forceinline__ F4 q4(int time, int c) const { if constexpr (IS3D) return time ? q.a[c] : q.b[c]; else return F4(); }
  __device__ __forceinline__ F2 q2(int time, int c) const { if constexpr (!IS3D) return time ? q.a[c] : q.b[c]; else return F2(); }
};

// one float32 layer value from multiplied-out weights (w00 = wy0 wx0, ...): the same sum as bil4 evaluated with three
// fused multiply-adds -- a float64 round-off apart, which reaches the float32 rounding in ~1e-8 of the values
__device__ __forceinline__ float bilw(float v00, float v01, float v10, float v11, double w00, double w01, double w10,
                                      double w11) {
  return (float)fma((double)v11, w11, fma((double)v10, w10, fma((double)v01, w01, (double)v00 * w00)));
}
struct FootW { double w00, w01, w10, w11; };
__device__ __forceinline__ FootW foot_weights(const Foot &ft) {
  FootW w;
  w.w00 = ft.wy0 * ft.wx0; w.w01 = ft.wy0 * ft.tx; w.w10 = ft.ty * ft.wx0; w.w11 = ft.ty * ft.tx;
  return w;
}

// (u,v) of an interleaved pair at one time level through a loader LD (PairLd<LdGlobal> / PairLd<LdTile>).
// FASTW: layer values from the multiplied-out weights `fw` (Runge-Kutta stage samples) instead of scipy's
// (v*wy)*wx order (the stored environment)
template <bool IS3D, bool FASTW, class LD>
__device__ __forceinline__ void uv_level_ld(const LD &ld, int time, int nz, const Foot &ft, const ZBracket &zb, double &u,
                                            double &v, bool &f32class, const FootW &fw) {
  const double ty = ft.ty, tx = ft.tx, wy0 = ft.wy0, wx0 = ft.wx0;
  if (IS3D) {
    const F4 q00 = ld.q4(time, 0), q01 = ld.q4(time, 1), q10 = ld.q4(time, 2), q11 = ld.q4(time, 3);
    // level "above" (ia) and "below" (ib): (x,y) = level iz0, (z,w) = level iz0+1
    float ua, va, ub, vb;
    if (FASTW) {
      ub = bilw(q00.z, q01.z, q10.z, q11.z, fw.w00, fw.w01, fw.w10, fw.w11);
      vb = bilw(q00.w, q01.w, q10.w, q11.w, fw.w00, fw.w01, fw.w10, fw.w11);
    } else {
      ub = bil4(q00.z, q01.z, q10.z, q11.z, wy0, ty, wx0, tx);
      vb = bil4(q00.w, q01.w, q10.w, q11.w, wy0, ty, wx0, tx);
    }
    if (zb.same && nz > 1) { ua = ub; va = vb; }
    else if (FASTW) {
      ua = bilw(q00.x, q01.x, q10.x, q11.x, fw.w00, fw.w01, fw.w10, fw.w11);
      va = bilw(q00.y, q01.y, q10.y, q11.y, fw.w00, fw.w01, fw.w10, fw.w11);
    } else {
      ua = bil4(q00.x, q01.x, q10.x, q11.x, wy0, ty, wx0, tx);
      va = bil4(q00.y, q01.y, q10.y, q11.y, wy0, ty, wx0, tx);
    }
    u = __dadd_rn(__dmul_rn((double)ua, zb.wa), __dmul_rn((double)ub, 1 - zb.wa));
    v = __dadd_rn(__dmul_rn((double)va, zb.wa), __dmul_rn((double)vb, 1 - zb.wa));
    f32class = false;
  } else {
    const F2 q00 = ld.q2(time, 0), q01 = ld.q2(time, 1), q10 = ld.q2(time, 2), q11 = ld.q2(time, 3);
    if (FASTW) {
      u = bilw(q00.x, q01.x, q10.x, q11.x, fw.w00, fw.w01, fw.w10, fw.w11);
      v = bilw(q00.y, q01.y, q10.y, q11.y, fw.w00, fw.w01, fw.w10, fw.w11);
    } else {
      u = bil4(q00.x, q01.x, q10.x, q11.x, wy0, ty, wx0, tx);
      v = bil4(q00.y, q01.y, q10.y, q11.y, wy0, ty, wx0, tx);
    }
    f32class = true;
  }
}
template <bool IS3D, bool FASTW = false>
__device__ __forceinline__ void uv_level(const float *__restrict__ uv, int nz, const Foot &ft,
                                         const ZBracket &zb, double &u, double &v, bool &f32class,
                                         const FootW &fw = FootW()) {
  PairLd<LdGlobal> g;
  g.ld.b = uv; g.ld.a = uv;
  g.o[0] = ft.o00; g.o[1] = ft.o01; g.o[2] = ft.o10; g.o[3] = ft.o11;
  g.kb = IS3D ? (unsigned)zb.iz0 * 8u : 0u;
  uv_level_ld<IS3D, FASTW>(g, 0, nz, ft, zb, u, v, f32class, fw);
}

// (u,v) float32 environment of one particle from the single grid source `s` at a Runge-Kutta STAGE position: the
// reference's layer-by-layer arithmetic (every (time, z) layer interpolated horizontally and rounded to float32, then
// z and time interpolation) with the horizontal weights multiplied out once for all layers.
// The records come through the loader `ld` (LdGlobal: time 0 = tm.b, 1 = tm.a; LdTile: the workgroup's LDS images of the
// same two levels); returns false when the footprint leaves the loader's image (LdTile only).
template <int PROJ, bool IS3D, class LD = LdGlobal>
__device__ __forceinline__ bool uv_sample_fast(const DevSource &s, const DevBlock &geo, const UVTime &tm, const LD &ld,
                                               double lon, double lat, double z, const ZBracket &zb,
                                               float fbu, float fbv, float &uo, float &vo,
                                               const ProjStart &ps, UVKeep<IS3D> &K, bool keep) {
  if (s.lon_mode == 1) lon = np_mod(lon + 180.0, 360.0) - 180.0;
  else if (s.lon_mode == 2) lon = np_mod(lon, 360.0);
  double x, y;
  if (PROJ == PROJ_LATLONG) { x = lon; y = lat; }
  else if (PROJ == PROJ_CURVILINEAR) curvi_locate(s.proj, lon, lat, x, y);
#ifndef ODR_FULL_STAGE_PROJECTION
  else if (ps.ok) proj_fwd_near<PROJ == PROJ_STERE_POLAR, PROJ == PROJ_EXT>(s.proj, ps, lon, lat, x, y);   // (by value: a pointer would pin the struct to scratch memory)
#endif
  else proj_fwd<PROJ == PROJ_STERE_POLAR, PROJ == PROJ_EXT>(s.proj, lon, lat, x, y);
  const double xchk = cover_x<PROJ>(s.proj.kind, s.lon_mode, x);
  float fu = __builtin_nanf(""), fv = __builtin_nanf("");
  bool ok = true;
  if (xchk >= s.xmin && xchk <= s.xmax && y >= s.ymin && y <= s.ymax && z >= s.zmin && z <= s.zmax) {
    if (s.mod360_x) x = np_mod(x, 360.0);
    double xi = __dmul_rn(div_cr(x - geo.x0, geo.xspan, geo.ixspan), (double)(geo.nx - 1));
    double yi = __dmul_rn(div_cr(y - geo.y0, geo.yspan, geo.iyspan), (double)(geo.ny - 1));
    double u, v;
    bool f32c;
    {
      const Foot ft = ld.foot(yi, xi, geo.ny, geo.nx, (unsigned)geo.rec * 4u, ok);
      if (!ok) return false;     // (LdTile) outside the rectangle: the caller samples from the blocks in HBM
      const FootW fw = foot_weights(ft);
      UVRec<IS3D> rec_;
      const UVRec<IS3D> &rec = uv_records<IS3D, ODR_UV_KEEPS(PROJ)>(ld, tm.a != nullptr, ft, IS3D ? (unsigned)zb.iz0 * 8u : 0u, K, keep, rec_);
      const RecLd<IS3D> L = {rec};
      double ub, vb;
      uv_level_ld<IS3D, true>(L, 0, s.nz, ft, zb, ub, vb, f32c, fw);
      u = ub; v = vb;
      if (tm.a) {
        double ua, va;
        uv_level_ld<IS3D, true>(L, 1, s.nz, ft, zb, ua, va, f32c, fw);
        if (f32c) {
          u = __fadd_rn(__fmul_rn((float)ub, (float)(1 - tm.w)), __fmul_rn((float)ua, (float)tm.w));
          v = __fadd_rn(__fmul_rn((float)vb, (float)(1 - tm.w)), __fmul_rn((float)va, (float)tm.w));
        } else {
          u = __dadd_rn(__dmul_rn(ub, 1 - tm.w), __dmul_rn(ua, tm.w));
          v = __dadd_rn(__dmul_rn(vb, 1 - tm.w), __dmul_rn(va, tm.w));
        }
      }
    }
    if (ODR_PROJ_ROTATES(PROJ)) {
      double sn, cs;
      rotation_cs<PROJ == PROJ_STERE_POLAR, PROJ == PROJ_EXT>(s.proj, x, y, cs, sn);
      double uu = u, vv = v;
      u = __dsub_rn(__dmul_rn(uu, cs), __dmul_rn(vv, sn));
      v = __dadd_rn(__dmul_rn(uu, sn), __dmul_rn(vv, cs));
    }
    fu = (float)u;
    fv = (float)v;
  }
  uo = isfinite(fu) ? fu : (isfinite(fbu) ? fbu : fu);
  vo = isfinite(fv) ? fv : (isfinite(fbv) ? fbv : fv);
  return ok;
}


// cos / sin of rot_angle_rad at (x, y) of a Mercator or Lambert conformal conic reader in closed form (ODR_STAGE_FAST stage
// samples): Mercator's +y axis is north everywhere and the 10 m line runs along a meridian -- no rotation; Lambert's +y axis has
// the azimuth theta = n (lambda - lambda0), the line starts half its change of longitude times (n - sin phi) off that (see
// proj_fwd_near).  3e-10 rad from the geodesic inverse on 2 000 random points per projection (oracle, CPU); rotation_angle (two
// inverse projections + the geodesic inverse: the reference's own recipe) stays for the main-loop sample and for EXACT.
__device__ __forceinline__ bool rot_closed_form(const DevProj &p, double x, double y, double lat_deg, double *rot) {
#pragma clang fp contract(fast)
  if (!rot_same_ellipsoid(p)) return false;
  if (p.kind == PROJ_MERC) { rot[0] = 1.0; rot[1] = 0.0; return true; }
  if (p.kind == PROJ_LCC) {
    const double iak = fast_rcp(p.a * p.k0);
    const double X = (x - p.x0) * iak, Yr = p.crho0 - (y - p.y0) * iak;
    const double h2 = X * X + Yr * Yr;
    if (!(h2 > 0)) return false;
    const double ir = p.cn > 0 ? fast_rsqrt(h2) : -fast_rsqrt(h2);      // 1 / rho, signed like the projection's rho
    const double st = X * ir, ct = Yr * ir;                               // sin, cos of theta
    const double c = 5.0 * iak * st * ir * (1.0 - sin(lat_deg * kDeg) * fast_rcp(p.cn));   // 0.5 * 10 m * d(lambda)/dy * (n - sin phi)
    rot[0] = ct - c * st;
    rot[1] = -(st + c * ct);
    return true;
  }
  return false;
}

// ODR_STAGE_FAST (odr_ctx_set_stage_math): the (u,v) of a Runge-Kutta STAGE position as ONE weighted sum of the
// 4 corners x 2 z levels x 2 time levels in float32 -- the reference's result is a float32 as well (the environment is cast to
// float32, environment.py:543), reached through float32 roundings of every (time, z) layer; here the 16 corner values per
// component go through packed float32 FMAs (u and v of a node are neighbours in the record: one v_pk_fma_f32 serves
// both), i.e. 16 instructions instead of 32 conversions + 72 float64 operations.  The value differs from the reference's
// stage value by a few float32 ulp (like the stage value of ODR_FAST_STAGE_SAMPLE in round 2): <= 2e-9 deg per step in
// the position.  Front door (longitude convention, projection, coverage, fractional indices) as in uv_sample_fast.
template <int PROJ, bool IS3D, class LD = LdGlobal>
__device__ __forceinline__ bool uv_sample_stage_f32(const DevSource &s, const DevBlock &geo, const UVTime &tm, const LD &ld,
                                                    double lon, double lat, double z, const ZBracket &zb,
                                                    float fbu, float fbv, float &uo, float &vo, const ProjStart &ps,
                                                    UVKeep<IS3D> &K, bool keep) {
  if (s.lon_mode == 1) lon = np_mod(lon + 180.0, 360.0) - 180.0;
  else if (s.lon_mode == 2) lon = np_mod(lon, 360.0);
  double x, y;
  double rot[2];             // cos, sin of the vector rotation at the stage position, when the projection hands them out
  bool have_rot = false;
  if (PROJ == PROJ_LATLONG) { x = lon; y = lat; }
  else if (PROJ == PROJ_CURVILINEAR) curvi_locate(s.proj, lon, lat, x, y);
  else if (ps.ok) {
#ifdef ODR_NO_STAGE_ROT_CLOSED_FORM   // A/B build: the rotation of every stage sample from rotation_cs, as in rounds 3-4
    proj_fwd_near<PROJ == PROJ_STERE_POLAR, PROJ == PROJ_EXT>(s.proj, ps, lon, lat, x, y);
#else
    proj_fwd_near<PROJ == PROJ_STERE_POLAR, PROJ == PROJ_EXT>(s.proj, ps, lon, lat, x, y, rot);
    have_rot = rot[0] <= 1.5;
#endif
  } else {
    proj_fwd<PROJ == PROJ_STERE_POLAR, PROJ == PROJ_EXT>(s.proj, lon, lat, x, y);
#ifndef ODR_NO_STAGE_ROT_CLOSED_FORM
    if (ODR_PROJ_ROTATES(PROJ) && PROJ != PROJ_STERE_POLAR && PROJ != PROJ_EXT) have_rot = rot_closed_form(s.proj, x, y, lat, rot);
#endif
  }
  const double xchk = cover_x<PROJ>(s.proj.kind, s.lon_mode, x);
  float fu = __builtin_nanf(""), fv = __builtin_nanf("");
  bool ok = true;
  if (xchk >= s.xmin && xchk <= s.xmax && y >= s.ymin && y <= s.ymax && z >= s.zmin && z <= s.zmax) {
#pragma clang fp contract(fast)
    if (s.mod360_x) x = np_mod(x, 360.0);
    const double xi = (x - geo.x0) * geo.ixspan * (double)(geo.nx - 1);
    const double yi = (y - geo.y0) * geo.iyspan * (double)(geo.ny - 1);
    const Foot ft = ld.foot(yi, xi, geo.ny, geo.nx, (unsigned)geo.rec * 4u, ok);
    if (!ok) return false;     // (LdTile) outside the rectangle: the caller samples from the blocks in HBM
    const float tx = (float)ft.tx, ty = (float)ft.ty, sx = 1.f - tx, sy = 1.f - ty;
    const float w00 = sy * sx, w01 = sy * tx, w10 = ty * sx, w11 = ty * tx;
    const float wt = tm.a ? (float)tm.w : 0.f;
    f32x2 r;
    UVRec<IS3D> rec_;
    if constexpr (IS3D) {
      // the footprint's records combined over the bracket's two levels (uv_fetch_combined): kept in K when the reader keeps
      bool kp = keep;
#if defined(ODR_NO_KEEP) || defined(ODR_TU_TILE)
      kp = false;
#endif
      const unsigned kb = (unsigned)zb.iz0 * 8u;
      float za, zw;
      uv_zweights(zb, s.nz, za, zw);
      UVRec<IS3D> &m = ODR_UV_KEEPS(PROJ) ? K.q : rec_;
      // (the bracket is the same for every sample of the step -- the kept values are dropped when the sea floor moved the element --
      // so the footprint's two node numbers identify the kept values)
      const bool same = ODR_UV_KEEPS(PROJ) && kp && K.valid && K.n00 == ft.n00 && K.n11 == ft.n11;
      if (!same) {
        uv_fetch_combined(ld, tm.a != nullptr, ft, kb, za, zw, m);
        if (ODR_UV_KEEPS(PROJ)) { K.n00 = ft.n00; K.n11 = ft.n11; K.valid = kp; }
      }
      r = (f32x2){m.b[0].x, m.b[0].y} * w00;
      r = __builtin_elementwise_fma((f32x2){m.b[1].x, m.b[1].y}, (f32x2){w01, w01}, r);
      r = __builtin_elementwise_fma((f32x2){m.b[2].x, m.b[2].y}, (f32x2){w10, w10}, r);
      r = __builtin_elementwise_fma((f32x2){m.b[3].x, m.b[3].y}, (f32x2){w11, w11}, r);
      if (tm.a) {
        f32x2 ra = (f32x2){m.b[0].z, m.b[0].w} * w00;
        ra = __builtin_elementwise_fma((f32x2){m.b[1].z, m.b[1].w}, (f32x2){w01, w01}, ra);
        ra = __builtin_elementwise_fma((f32x2){m.b[2].z, m.b[2].w}, (f32x2){w10, w10}, ra);
        ra = __builtin_elementwise_fma((f32x2){m.b[3].z, m.b[3].w}, (f32x2){w11, w11}, ra);
        r = __builtin_elementwise_fma(ra - r, (f32x2){wt, wt}, r);
      }
    } else {
      const UVRec<IS3D> &rec = uv_records<IS3D, ODR_UV_KEEPS(PROJ)>(ld, tm.a != nullptr, ft, 0u, K, keep, rec_);
      auto level = [&](int time) {
        const F2 q00 = time ? rec.a[0] : rec.b[0], q01 = time ? rec.a[1] : rec.b[1];
        const F2 q10 = time ? rec.a[2] : rec.b[2], q11 = time ? rec.a[3] : rec.b[3];
        f32x2 a = (f32x2){q00.x, q00.y} * w00;
        a = __builtin_elementwise_fma((f32x2){q01.x, q01.y}, (f32x2){w01, w01}, a);
        a = __builtin_elementwise_fma((f32x2){q10.x, q10.y}, (f32x2){w10, w10}, a);
        return __builtin_elementwise_fma((f32x2){q11.x, q11.y}, (f32x2){w11, w11}, a);
      };
      r = level(0);
      if (tm.a) { const f32x2 ra = level(1); r = __builtin_elementwise_fma(ra - r, (f32x2){wt, wt}, r); }
    }
    fu = r.x; fv = r.y;
    if (ODR_PROJ_ROTATES(PROJ)) {
      double sn, cs;
      if (have_rot) { cs = rot[0]; sn = rot[1]; }
      else rotation_cs<PROJ == PROJ_STERE_POLAR, PROJ == PROJ_EXT>(s.proj, x, y, cs, sn);
      const float c = (float)cs, n = (float)sn, uu = fu, vv = fv;
      fu = uu * c - vv * n;
      fv = uu * n + vv * c;
    }
  }
  uo = isfinite(fu) ? fu : (isfinite(fbu) ? fbu : fu);
  vo = isfinite(fv) ? fv : (isfinite(fbv) ? fbv : fv);
  return ok;
}

// the stage sample of the chosen stage math (SM: 0 = ODR_STAGE_EXACT, 1 = ODR_STAGE_FAST) through the loader `ld`, whose
// time 0 / 1 are the (u,v) arrays of tm.b / tm.a; false: the footprint left the loader's image (LdTile)
template <int PROJ, bool IS3D, int SM, class LD>
__device__ __forceinline__ bool uv_stage(const DevSource &s, const DevBlock &geo, const UVTime &tm, const LD &ld, double lon, double lat,
                                         double z, const ZBracket &zb, float fbu, float fbv, float &uo, float &vo,
                                         const ProjStart &ps, UVKeep<IS3D> &K, bool keep) {
  if constexpr (SM == 1) return uv_sample_stage_f32<PROJ, IS3D>(s, geo, tm, ld, lon, lat, z, zb, fbu, fbv, uo, vo, ps, K, keep);
  else return uv_sample_fast<PROJ, IS3D>(s, geo, tm, ld, lon, lat, z, zb, fbu, fbv, uo, vo, ps, K, keep);
}
// the global loader of a stage sample: the (u,v) arrays of the bracketing levels
__device__ __forceinline__ LdGlobal uv_global(const UVTime &tm) { LdGlobal g; g.b = tm.b; g.a = tm.a ? tm.a : tm.b; return g; }

// ---------------------------------------------------------- fast environment group
// Main-loop Environment.get_environment for a variable group served by ONE gridded reader:
// projection, coverage, fractional indices, the 2x2 footprint, the vertical bracket and the
// (host-resolved) time bracket are computed once per particle and shared by all variables of
// the group; every variable is then a few loads at (footprint offset + its offset in the node
// record): 16 bytes for an interleaved 3D vector pair (both components, two z levels), 8 bytes
// for a 3D scalar or a 2D pair, 4 bytes for a 2D scalar.
constexpr int MAXG = 8;
struct EnvGroupDesc {
  int nv, sid, geo_slot, all_static;
  const float *bb, *ba;  // node-record bases of the bracketing time levels (ba == nullptr: on a time level)
  double w;
  int var[MAXG];
  int off[MAXG];         // float offset of the variable in the node record
  int nz[MAXG];          // 1 => 2D
  int es[MAXG];          // 2 => interleaved with its vector partner
  int partner[MAXG];     // index in this group of the y-component to rotate with, or -1
  int kind[MAXG];        // 0 scalar; 1 x-component of an interleaved pair whose y-component is slot k+1; 2 that y-component
  float fallback[MAXG];
  int has_land, pad;
  int mode[MAXG];        // ENV_* below: how the slot is gathered and combined (host: odr_i_build_env_group)
  // burst sampler (env_burst): group index of the variable in each physical slot A, B, C, D, L (-1: empty); burst = 0: the
  // group does not fit the slots and takes the serial sampler; rotates: some slot holds a vector pair to rotate
  int bs[5], burst, rotates, pad2;
  // per physical slot (static indices in the kernel: one batched scalar load instead of dependent ones): byte offset of the
  // variable in the node record, ENV_* mode, 1 = vector pair to rotate; temp_mask: bit k = group variable k is sea_water_temperature
  int ps_off[5], ps_mode[5], ps_rot[5], temp_mask;
  int ps_static, pad3;   // bit q: the 2-D variable in physical slot q holds the same values at both time levels (host: content ids)
  float *out_ptr[MAXG];  // where group variable k of the launch's particle set is stored (env_bind_out: p.env[var[k]]): a static
                         // index in the kernels instead of two dependent scalar loads per stored variable
};
// gather / arithmetic class of a group slot.  S = scalar, P = interleaved vector pair (the slot and the one after it);
// 2 / 3 = 2-D / z-interpolated; S3I = one component of an interleaved 3-D pair on its own
enum { ENV_S2 = 0, ENV_S3 = 1, ENV_S3I = 2, ENV_P2 = 3, ENV_P3 = 4, ENV_LAND = 5, ENV_SKIP = 6 };

// one scalar variable at one time level -> value in the reference's dtype class; d = address of
// the variable in node record 0
__device__ __forceinline__ double var_level(const float *__restrict__ d, int var, int nzv, int es, int snz,
                                            unsigned near_off, const Foot &ft, const ZBracket &zb,
                                            bool &f32class) {
  if (var == VAR_LAND) {
    f32class = true;
    return ld_off<float>(d, near_off);
  }
  const double ty = ft.ty, tx = ft.tx, wy0 = ft.wy0, wx0 = ft.wx0;
  if (nzv <= 1) {
    f32class = true;
    return bil4(ld_off<float>(d, ft.o00), ld_off<float>(d, ft.o01), ld_off<float>(d, ft.o10),
                ld_off<float>(d, ft.o11), wy0, ty, wx0, tx);
  }
  f32class = false;
  float a00, a01, a10, a11, b00, b01, b10, b11;  // level iz0 (a) and iz0+1 (b)
  const unsigned kb = (unsigned)zb.iz0 * 4u * (unsigned)es;
  if (es == 1) {
    const F2 q00 = ld_off<F2>(d, ft.o00 + kb), q01 = ld_off<F2>(d, ft.o01 + kb);
    const F2 q10 = ld_off<F2>(d, ft.o10 + kb), q11 = ld_off<F2>(d, ft.o11 + kb);
    a00 = q00.x; b00 = q00.y; a01 = q01.x; b01 = q01.y; a10 = q10.x; b10 = q10.y; a11 = q11.x; b11 = q11.y;
  } else {  // one component of an interleaved pair on its own
    const unsigned sb = 4u * (unsigned)es;
    a00 = ld_off<float>(d, ft.o00 + kb); b00 = ld_off<float>(d, ft.o00 + kb + sb);
    a01 = ld_off<float>(d, ft.o01 + kb); b01 = ld_off<float>(d, ft.o01 + kb + sb);
    a10 = ld_off<float>(d, ft.o10 + kb); b10 = ld_off<float>(d, ft.o10 + kb + sb);
    a11 = ld_off<float>(d, ft.o11 + kb); b11 = ld_off<float>(d, ft.o11 + kb + sb);
  }
  float vb = bil4(b00, b01, b10, b11, wy0, ty, wx0, tx);
  float va = (zb.same && snz > 1) ? vb : bil4(a00, a01, a10, a11, wy0, ty, wx0, tx);
  return __dadd_rn(__dmul_rn((double)va, zb.wa), __dmul_rn((double)vb, 1 - zb.wa));
}

// ---- burst sampler (round 3).  Phase timing of k_step_grid showed the main-loop sample as 45 % of a wave's life: the
// serial sampler below gathers one variable and time level after the other (a memory round trip each, behind uniform
// branches the compiler cannot move loads across).  Here the host assigns the group's variables to five PHYSICAL
// slots of fixed register width -- A: up to 16 bytes per corner (a 3-D vector pair, or anything narrower), B and C: up to
// 8 bytes (a 3-D scalar or a 2-D pair), D: 4 bytes (a 2-D scalar), L: the land mask's nearest node -- and ALL their
// gathers, both time levels, leave in one burst; the arithmetic follows, slot by slot, with the rounding points of the
// serial sampler (same bits: tests/test_gpu_parity.py, test_gpu_generic_paths.py).  Groups that do not fit the slots
// (more variables, pairs that are not interleaved) take the serial sampler.
__device__ __forceinline__ void put_slot(float *out /*[MAXG]*/, int k, float v) {
#pragma unroll
  for (int j = 0; j < MAXG; ++j) out[j] = j == k ? v : out[j];
}
template <int WIDTH, class Get>
__device__ __forceinline__ void burst_math(int m, Get get, const Foot &ft, const ZBracket &zb, int snz, bool tl, double w,
                                           double &v0, double &v1) {
  const double ty = ft.ty, tx = ft.tx, wy0 = ft.wy0, wx0 = ft.wx0;
  auto lerp32 = [&](double b, double a) { return (double)__fadd_rn(__fmul_rn((float)b, (float)(1 - w)), __fmul_rn((float)a, (float)w)); };
  auto lerp64 = [&](double b, double a) { return __dadd_rn(__dmul_rn(b, 1 - w), __dmul_rn(a, w)); };
  auto b4 = [&](int t, int c) { return bil4(get(t, 0, c), get(t, 1, c), get(t, 2, c), get(t, 3, c), wy0, ty, wx0, tx); };
  const bool same = zb.same && snz > 1;
  auto z2 = [&](float a, float b) { return __dadd_rn(__dmul_rn((double)a, zb.wa), __dmul_rn((double)b, 1 - zb.wa)); };
  v0 = v1 = 0;
  if (m == ENV_S2) {
    const float b = b4(0, 0);
    v0 = tl ? lerp32(b, b4(1, 0)) : (double)b;
    return;
  }
  if constexpr (WIDTH >= 2) {
    if (m == ENV_P2) {
      const float ub = b4(0, 0), vb = b4(0, 1);
      if (tl) { v0 = lerp32(ub, b4(1, 0)); v1 = lerp32(vb, b4(1, 1)); }
      else { v0 = ub; v1 = vb; }
      return;
    }
    if (m == ENV_S3) {
      const float lb = b4(0, 1), la = same ? lb : b4(0, 0);
      v0 = z2(la, lb);
      if (tl) { const float mb = b4(1, 1), ma = same ? mb : b4(1, 0); v0 = lerp64(v0, z2(ma, mb)); }
      return;
    }
  }
  if constexpr (WIDTH >= 4) {
    if (m == ENV_S3I) {
      const float lb = b4(0, 2), la = same ? lb : b4(0, 0);
      v0 = z2(la, lb);
      if (tl) { const float mb = b4(1, 2), ma = same ? mb : b4(1, 0); v0 = lerp64(v0, z2(ma, mb)); }
      return;
    }
    if (m == ENV_P3) {
      const float ub = b4(0, 2), vb = b4(0, 3);
      const float ua = same ? ub : b4(0, 0), va = same ? vb : b4(0, 1);
      v0 = z2(ua, ub); v1 = z2(va, vb);
      if (tl) {
        const float pb = b4(1, 2), qb = b4(1, 3);
        const float pa = same ? pb : b4(1, 0), qa = same ? qb : b4(1, 1);
        v0 = lerp64(v0, z2(pa, pb)); v1 = lerp64(v1, z2(qa, qb));
      }
    }
  }
}
// what the main-loop sample hands to the stage samples of the same particle-step (UVKeep): the records of slot A as
// fetched (the step kernels put the current there), the footprint's identity and the level offset
struct EnvExport { float b[16], a[16]; unsigned n00, n11; int iz0; bool valid; bool combine = false; };   // combine: b[4c..4c+3] = uv_combine_z of corner c (ODR_STAGE_FAST, 3-D), a[] unused   // (plain floats: arrays of structs behind a pointer stay in scratch memory)
// L: the loader of the node records (time 0 = the level before, 1 = the level after, or the same level when !tl); ft and
// near_off hold ITS byte offsets (LdGlobal::foot / LdTile::foot)
template <int PROJ, class LD>
__device__ __forceinline__ void env_burst(const DevSource &s, const EnvGroupDesc &G, const LD &L, const Foot &ft, const ZBracket &zb,
                                          unsigned near_off, bool tl, double x, double y, float *out /*[MAXG]*/, EnvExport *X ODR_PT_PARAM) {
  const unsigned o[4] = {ft.o00, ft.o01, ft.o10, ft.o11};
  const unsigned iz0 = (unsigned)zb.iz0;
#ifdef ODR_WHATIF_NO_BURST2   // what-if build (wrong values): slots B, C, D are not sampled -- what a fifth wave per SIMD buys
  const int kA = G.bs[0], kB = -1, kC = -1, kD = -1, kL = G.bs[4];
#elif defined(ODR_WHATIF_C3SPEC)   // what-if build (bench C3 only): the group's layout as compile-time constants
  constexpr int kA = 0, kB = 2, kC = 3, kD = -1, kL = 4;
#else
  const int kA = G.bs[0], kB = G.bs[1], kC = G.bs[2], kD = G.bs[3], kL = G.bs[4];
#endif
#ifdef ODR_WHATIF_C3SPEC
  constexpr int mA = ENV_P3, mB = ENV_S3, mC = ENV_S2;
  double cs = 1, sn = 0;
  constexpr int temp_mask = 0;
#else
  const int mA = G.ps_mode[0], mB = G.ps_mode[1], mC = G.ps_mode[2];
  double cs = 1, sn = 0;
  if (ODR_PROJ_ROTATES(PROJ) && G.rotates) rotation_cs<PROJ == PROJ_STERE_POLAR, PROJ == PROJ_EXT>(s.proj, x, y, cs, sn);
  const int temp_mask = G.temp_mask;
#endif
  auto finish = [&](int k, double v) {      // masked_invalid(...).astype('float32'), fallback, Kelvin -> Celsius
    float f = (float)v;
    if (!isfinite(f)) { const float fb = G.fallback[k]; f = isfinite(fb) ? fb : f; }   // rare: the scalar load stays in here
    put_slot(out, k, (temp_mask >> k & 1) ? kelvin_to_celsius(f) : f);
  };
  auto emit = [&](int k, int m, int rot, double v0, double v1) {
    if (m == ENV_P2 || m == ENV_P3) {
      if (ODR_PROJ_ROTATES(PROJ) && rot) {
        const double uu = v0, vv = v1;
        v0 = __dsub_rn(__dmul_rn(uu, cs), __dmul_rn(vv, sn));
        v1 = __dadd_rn(__dmul_rn(uu, sn), __dmul_rn(vv, cs));
      }
      finish(k, v0); finish(k + 1, v1);
    } else finish(k, v0);
  };
  const double w = G.w;
  // The gathers of a slot are the SAME instructions whatever the slot holds -- always 16 (A), 8 (B, C), 4 (D) bytes per
  // corner, only the byte offset depends on the mode: a load whose destination registers differ between uniform branches
  // ends in a copy at the join, i.e. in a wait right behind it, and the burst would be gone.  A narrower variable in a
  // wider slot over-reads into its neighbours in the node record (blocks end with 64 spare bytes); empty slots read offset 0.
#ifndef ODR_BURST_A_FIRST
  // (round 5: the slots B, C, D go FIRST -- their 40 registers of returns are consumed before slot A's records, which the stage
  // samples keep (UVKeep: 32 registers), are requested; with slot A first the kept records and this burst's returns were in
  // flight together, through the phase that set the kernel's register count.  Same gathers, same arithmetic, same bits.)
  // ---- burst 2: slots B, C, D (40 registers), then their arithmetic
  if (kB >= 0 || kC >= 0 || kD >= 0) {
    const unsigned dB = (unsigned)G.ps_off[1] + (mB == ENV_S3 ? iz0 * 4u : 0u);
    const unsigned dC = (unsigned)G.ps_off[2] + (mC == ENV_S3 ? iz0 * 4u : 0u);
    const unsigned dD = (unsigned)G.ps_off[3];
    F2 Bb[4], Ba[4], Cb[4], Ca[4];
    float Db[4], Da[4];
    // (measured, round 4: these gathers issued right behind slot A's, one memory round trip less per particle at the same 126
    // registers: launch 0.615 -> 0.633 ms, C5 0.776 -> 0.861 -- 74 registers of returns in flight per lane queue behind one another)
    // (empty slots issue their gathers too, at offset 0.  Measured alternatives, C3: guarding a slot's gathers with the
    // condition that also guards its arithmetic lets the compiler merge the two blocks -- gathers, wait, arithmetic, slot by
    // slot -- 1.26 -> 1.45 ms per step; a zero-length buffer descriptor for empty slots 1.26 -> 1.31)
#pragma unroll
    for (int c = 0; c < 4; ++c) { Bb[c] = L.template ld<F2>(0, o[c] + dB); Ba[c] = L.template ld<F2>(1, o[c] + dB); }
    // (a slot whose variable holds the same values at both time levels -- ps_static, from the blocks' content ids -- is
    // gathered once: four 32-cycle gathers less; the copies sit behind the slot's own last load, where the arithmetic
    // waits anyway)
    const int ps_static = G.ps_static;
#pragma unroll
    for (int c = 0; c < 4; ++c) Cb[c] = L.template ld<F2>(0, o[c] + dC);
    if (ps_static & 4) {
#pragma unroll
      for (int c = 0; c < 4; ++c) Ca[c] = Cb[c];
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) Ca[c] = L.template ld<F2>(1, o[c] + dC);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) Db[c] = L.template ld<float>(0, o[c] + dD);
    if (ps_static & 8) {
#pragma unroll
      for (int c = 0; c < 4; ++c) Da[c] = Db[c];
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) Da[c] = L.template ld<float>(1, o[c] + dD);
    }
    if (kB >= 0) {
      double v0, v1;
      burst_math<2>(mB, [&](int t, int c, int q) { const F2 &r = t ? Ba[c] : Bb[c]; return q == 0 ? r.x : r.y; }, ft, zb, s.nz, tl, w, v0, v1);
      emit(kB, mB, G.ps_rot[1], v0, v1);
    }
    if (kC >= 0) {
      double v0, v1;
      burst_math<2>(mC, [&](int t, int c, int q) { const F2 &r = t ? Ca[c] : Cb[c]; return q == 0 ? r.x : r.y; }, ft, zb, s.nz, tl, w, v0, v1);
      emit(kC, mC, G.ps_rot[2], v0, v1);
    }
    if (kD >= 0) {
      double v0, v1;
      burst_math<1>(ENV_S2, [&](int t, int c, int q) { return t ? Da[c] : Db[c]; }, ft, zb, s.nz, tl, w, v0, v1);
      emit(kD, ENV_S2, 0, v0, v1);
    }
  }
#endif
  // ---- burst 1: slot A and the land mask (34 registers in flight), then their arithmetic
  {
    const unsigned dA = (unsigned)G.ps_off[0] + ((mA == ENV_P3 || mA == ENV_S3I) ? iz0 * 8u : mA == ENV_S3 ? iz0 * 4u : 0u);
    const unsigned dL = kL >= 0 ? near_off + (unsigned)G.ps_off[4] : 0u;
    F4 Ab[4], Aa[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { Ab[c] = L.template ld<F4>(0, o[c] + dA); Aa[c] = L.template ld<F4>(1, o[c] + dA); }
    if (X) {
      if (X->combine) {   // the stage samples' kept values right away: 16 registers carried to the stage phase instead of 32
        float za, zw;
        uv_zweights(zb, s.nz, za, zw);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          X->b[4 * c] = __builtin_fmaf(Ab[c].z, zw, Ab[c].x * za); X->b[4 * c + 1] = __builtin_fmaf(Ab[c].w, zw, Ab[c].y * za);
          X->b[4 * c + 2] = __builtin_fmaf(Aa[c].z, zw, Aa[c].x * za); X->b[4 * c + 3] = __builtin_fmaf(Aa[c].w, zw, Aa[c].y * za);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          X->b[4 * c] = Ab[c].x; X->b[4 * c + 1] = Ab[c].y; X->b[4 * c + 2] = Ab[c].z; X->b[4 * c + 3] = Ab[c].w;
          X->a[4 * c] = Aa[c].x; X->a[4 * c + 1] = Aa[c].y; X->a[4 * c + 2] = Aa[c].z; X->a[4 * c + 3] = Aa[c].w;
        }
      }
      X->n00 = ft.n00; X->n11 = ft.n11; X->iz0 = zb.iz0; X->valid = true;
    }
    const int ps_static = G.ps_static;
    const float Lb = L.template ld<float>(0, dL);
    float La;
    if (ps_static & 16) La = Lb; else La = L.template ld<float>(1, dL);   // same values at both levels: one gather less
    if (kA >= 0) {
      double v0, v1;
      burst_math<4>(mA, [&](int t, int c, int q) { const F4 &r = t ? Aa[c] : Ab[c]; return q == 0 ? r.x : q == 1 ? r.y : q == 2 ? r.z : r.w; },
                    ft, zb, s.nz, tl, w, v0, v1);
      emit(kA, mA, G.ps_rot[0], v0, v1);
    }
    if (kL >= 0)
      finish(kL, tl ? (double)__fadd_rn(__fmul_rn(Lb, (float)(1 - w)), __fmul_rn(La, (float)w)) : (double)Lb);
  }
  ODR_PT_USE(out[0]); ODR_PT_USE(out[1]); ODR_PT_USE(out[2]); ODR_PT(15);
#ifdef ODR_BURST_A_FIRST
  // ---- burst 2: slots B, C, D (40 registers), then their arithmetic
  if (kB >= 0 || kC >= 0 || kD >= 0) {
    const unsigned dB = (unsigned)G.ps_off[1] + (mB == ENV_S3 ? iz0 * 4u : 0u);
    const unsigned dC = (unsigned)G.ps_off[2] + (mC == ENV_S3 ? iz0 * 4u : 0u);
    const unsigned dD = (unsigned)G.ps_off[3];
    F2 Bb[4], Ba[4], Cb[4], Ca[4];
    float Db[4], Da[4];
    // (measured, round 4: these gathers issued right behind slot A's, one memory round trip less per particle at the same 126
    // registers: launch 0.615 -> 0.633 ms, C5 0.776 -> 0.861 -- 74 registers of returns in flight per lane queue behind one another)
    // (empty slots issue their gathers too, at offset 0.  Measured alternatives, C3: guarding a slot's gathers with the
    // condition that also guards its arithmetic lets the compiler merge the two blocks -- gathers, wait, arithmetic, slot by
    // slot -- 1.26 -> 1.45 ms per step; a zero-length buffer descriptor for empty slots 1.26 -> 1.31)
#pragma unroll
    for (int c = 0; c < 4; ++c) { Bb[c] = L.template ld<F2>(0, o[c] + dB); Ba[c] = L.template ld<F2>(1, o[c] + dB); }
    // (a slot whose variable holds the same values at both time levels -- ps_static, from the blocks' content ids -- is
    // gathered once: four 32-cycle gathers less; the copies sit behind the slot's own last load, where the arithmetic
    // waits anyway)
    const int ps_static = G.ps_static;
#pragma unroll
    for (int c = 0; c < 4; ++c) Cb[c] = L.template ld<F2>(0, o[c] + dC);
    if (ps_static & 4) {
#pragma unroll
      for (int c = 0; c < 4; ++c) Ca[c] = Cb[c];
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) Ca[c] = L.template ld<F2>(1, o[c] + dC);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) Db[c] = L.template ld<float>(0, o[c] + dD);
    if (ps_static & 8) {
#pragma unroll
      for (int c = 0; c < 4; ++c) Da[c] = Db[c];
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) Da[c] = L.template ld<float>(1, o[c] + dD);
    }
    if (kB >= 0) {
      double v0, v1;
      burst_math<2>(mB, [&](int t, int c, int q) { const F2 &r = t ? Ba[c] : Bb[c]; return q == 0 ? r.x : r.y; }, ft, zb, s.nz, tl, w, v0, v1);
      emit(kB, mB, G.ps_rot[1], v0, v1);
    }
    if (kC >= 0) {
      double v0, v1;
      burst_math<2>(mC, [&](int t, int c, int q) { const F2 &r = t ? Ca[c] : Cb[c]; return q == 0 ? r.x : r.y; }, ft, zb, s.nz, tl, w, v0, v1);
      emit(kC, mC, G.ps_rot[2], v0, v1);
    }
    if (kD >= 0) {
      double v0, v1;
      burst_math<1>(ENV_S2, [&](int t, int c, int q) { return t ? Da[c] : Db[c]; }, ft, zb, s.nz, tl, w, v0, v1);
      emit(kD, ENV_S2, 0, v0, v1);
    }
  }
#endif
  ODR_PT_USE(out[3]); ODR_PT_USE(out[4]); ODR_PT(16);
}

// The reader front door of the main-loop sample (longitude convention, projection, coverage, fractional grid indices):
// computed once per particle; k_step_tile also takes the footprint rectangle of its workgroup from it.
struct EnvFront { double x, y, xi, yi; bool covered; int f32idx; };
// f32idx (k_env_grid in a run's first get_environment: DevWorld::f32pos on a geographic reader with float32 coordinate arrays,
// DevSource::xy_f32): the index maps in float32 arithmetic (index_f32).  Every other caller leaves it at 0, a compile-time constant.
// ps (the step launch of a polar stereographic reader): sines and cosines of the element's position, which the Runge-Kutta
// stages need anyway (proj_start) -- formed once in front of the sample instead of once here and once there; same bits
// (proj_fwd<ELLPOLAR> is sincos_pi + stere_polar_from_sc as well)
template <int PROJ>
__device__ __forceinline__ EnvFront env_front(const DevSource &s, const DevBlock &geo, double lon, double lat, double z, int f32idx = 0,
                                              const ProjStart ps = ProjStart()) {
  EnvFront f;
  f.f32idx = f32idx;
  if (s.lon_mode == 1) lon = np_mod(lon + 180.0, 360.0) - 180.0;
  else if (s.lon_mode == 2) lon = np_mod(lon, 360.0);
  double x, y;
  if (PROJ == PROJ_LATLONG) { x = lon; y = lat; }
  else if (PROJ == PROJ_CURVILINEAR) curvi_locate(s.proj, lon, lat, x, y);
  else if (PROJ == PROJ_STERE_POLAR && ps.ok) stere_polar_from_sc(s.proj, ps.phi, ps.sl, ps.cl, ps.sp, ps.cp, x, y);
  else proj_fwd<PROJ == PROJ_STERE_POLAR, PROJ == PROJ_EXT>(s.proj, lon, lat, x, y);
  const double xchk = cover_x<PROJ>(s.proj.kind, s.lon_mode, x);
  f.covered = xchk >= s.xmin && xchk <= s.xmax && y >= s.ymin && y <= s.ymax && z >= s.zmin && z <= s.zmax;
  if (s.mod360_x) x = np_mod(x, 360.0);
  f.x = x; f.y = y;
  f.xi = (f32idx & 1) ? index_f32(x, geo.x0, geo.xspan, geo.nx - 1) : __dmul_rn(div_cr(x - geo.x0, geo.xspan, geo.ixspan), (double)(geo.nx - 1));
  f.yi = (f32idx & 2) ? index_f32(y, geo.y0, geo.yspan, geo.ny - 1) : __dmul_rn(div_cr(y - geo.y0, geo.yspan, geo.iyspan), (double)(geo.ny - 1));
  return f;
}

// BURST_ONLY: the caller guarantees G.burst (the fused step kernel: groups that do not fit the slots take the separate launches)
// L: the loader of the node records (LdGlobal of G.bb / G.ba, or the workgroup's LDS tile); returns false when the
// footprint or the land mask's nearest node is outside the loader's image (LdTile) -- `out` is then meaningless.
template <int PROJ, bool BURST_ONLY, bool ZT, class LD>
__device__ __forceinline__ bool env_group_sample(const DevWorld &W, const EnvGroupDesc &G, const LD &L, const EnvFront &fr,
                                                 double z, float *out /*[MAXG]*/, const double *zt,
                                                 ZBracket &zb_out, EnvExport *X = nullptr ODR_PT_PARAM) {
  ODR_PT(10);
  const DevSource &s = W.src[G.sid];
  const DevBlock &geo = s.slot[G.geo_slot];
  const double x = fr.x, y = fr.y;
  const bool covered = fr.covered;
  double val[MAXG];
  const bool burst = BURST_ONLY || G.burst;
  if (burst) {
#pragma unroll
    for (int k = 0; k < MAXG; ++k) out[k] = 0.f;
    if (!covered) {     // NaN -> fallback (rare: the scalar loads of the fallbacks stay in here)
#pragma unroll
      for (int k = 0; k < MAXG; ++k) {
        if (k >= G.nv) break;
        const float fb = G.fallback[k];
        out[k] = !isfinite(fb) ? __builtin_nanf("") : ((G.temp_mask >> k & 1) ? kelvin_to_celsius(fb) : fb);
      }
    }
  }
  ODR_PT_USE(covered); ODR_PT(11);
  if (covered) {
    const double xi = fr.xi, yi = fr.yi;
    ODR_PT_USE(xi); ODR_PT_USE(yi); ODR_PT(12);
    ZBracket zb;
    zb.iz0 = 0; zb.same = 0; zb.wa = 1;
    if (s.nz > 1) zb = zbracket<ZT>(s, z, zt);
    zb_out = zb;
    ODR_PT_USE(zb.iz0); ODR_PT_USE(zb.wa); ODR_PT(13);
    // shared by all variables and both time levels: bilinear footprint, nearest node (land mask)
    const unsigned rec_bytes = (unsigned)geo.rec * 4u;
    bool ok, ok_near = true;
    const Foot ft = L.foot(yi, xi, geo.ny, geo.nx, rec_bytes, ok);
    unsigned near_off = 0;
    if (G.has_land)
      near_off = L.node((fr.f32idx & 2) ? nearest_index_f32(y, geo.ymin, geo.yrange, geo.ny) : nearest_index(y, geo.ymin, geo.yrange, geo.iyrange, geo.ny),
                        (fr.f32idx & 1) ? nearest_index_f32(x, geo.xmin, geo.xrange, geo.nx) : nearest_index(x, geo.xmin, geo.xrange, geo.ixrange, geo.nx),
                        geo.nx, rec_bytes, ok_near);
    if (!(ok && ok_near)) return false;
    const bool tl = G.ba != nullptr && !G.all_static;
    ODR_PT_USE(ft.o00); ODR_PT_USE(ft.o11); ODR_PT_USE(near_off); ODR_PT(14);
    if (burst) env_burst<PROJ>(s, G, L, ft, zb, near_off, tl, x, y, out, X ODR_PT_ARG);
    else if constexpr (!BURST_ONLY) {
    static_assert(BURST_ONLY || sizeof(LD) == sizeof(LdGlobal), "the serial sampler reads the blocks in HBM");
#pragma unroll
    for (int k = 0; k < MAXG; ++k) {
      if (k >= G.nv) break;
      if (G.kind[k] == 2) continue;  // done with its x-component
      bool fb, fa;
      if (G.kind[k] == 1 && k + 1 < MAXG) {  // interleaved vector pair: one load serves both components
        double ub, vb2;
        if (G.nz[k] > 1) uv_level<true>(G.bb + G.off[k], s.nz, ft, zb, ub, vb2, fb);
        else uv_level<false>(G.bb + G.off[k], s.nz, ft, zb, ub, vb2, fb);
        if (tl) {
          double ua, va2;
          if (G.nz[k] > 1) uv_level<true>(G.ba + G.off[k], s.nz, ft, zb, ua, va2, fa);
          else uv_level<false>(G.ba + G.off[k], s.nz, ft, zb, ua, va2, fa);
          if (fb && fa) {
            ub = __fadd_rn(__fmul_rn((float)ub, (float)(1 - G.w)), __fmul_rn((float)ua, (float)G.w));
            vb2 = __fadd_rn(__fmul_rn((float)vb2, (float)(1 - G.w)), __fmul_rn((float)va2, (float)G.w));
          } else {
            ub = __dadd_rn(__dmul_rn(ub, 1 - G.w), __dmul_rn(ua, G.w));
            vb2 = __dadd_rn(__dmul_rn(vb2, 1 - G.w), __dmul_rn(va2, G.w));
          }
        }
        val[k] = ub;
        val[k + 1] = vb2;
        continue;
      }
      double vb = var_level(G.bb + G.off[k], G.var[k], G.nz[k], G.es[k], s.nz, near_off, ft, zb, fb);
      if (tl) {
        double va = var_level(G.ba + G.off[k], G.var[k], G.nz[k], G.es[k], s.nz, near_off, ft, zb, fa);
        if (fb && fa) vb = __fadd_rn(__fmul_rn((float)vb, (float)(1 - G.w)), __fmul_rn((float)va, (float)G.w));
        else vb = __dadd_rn(__dmul_rn(vb, 1 - G.w), __dmul_rn(va, G.w));
      }
      val[k] = vb;
    }
    if (ODR_PROJ_ROTATES(PROJ)) {
      bool need = false;
#pragma unroll
      for (int k = 0; k < MAXG; ++k) if (k < G.nv && G.partner[k] >= 0) need = true;
      if (need) {
        double sn, cs;
        rotation_cs<PROJ == PROJ_STERE_POLAR, PROJ == PROJ_EXT>(s.proj, x, y, cs, sn);
#pragma unroll
        for (int k = 0; k + 1 < MAXG; ++k) {  // the host places the y-component right after its x-component
          if (k >= G.nv || G.partner[k] < 0) continue;
          double uu = val[k], vv = val[k + 1];
          val[k] = __dsub_rn(__dmul_rn(uu, cs), __dmul_rn(vv, sn));
          val[k + 1] = __dadd_rn(__dmul_rn(uu, sn), __dmul_rn(vv, cs));
        }
      }
    }
    }
  }
  if (burst) return true;
#pragma unroll
  for (int k = 0; k < MAXG; ++k) {
    if (k >= G.nv) break;
    float f = covered ? (float)val[k] : __builtin_nanf("");
    f = isfinite(f) ? f : (isfinite(G.fallback[k]) ? G.fallback[k] : f);
    out[k] = G.var[k] == VAR_TEMP ? kelvin_to_celsius(f) : f;
  }
  return true;
}

// the global loader of the group's two bracketing levels
__device__ __forceinline__ LdGlobal env_global(const EnvGroupDesc &G) {
  LdGlobal g;
  g.b = G.bb;
  g.a = (G.ba != nullptr && !G.all_static) ? G.ba : G.bb;
  return g;
}
// front door + sample from the blocks in HBM
template <int PROJ, bool BURST_ONLY, bool ZT>
__device__ __forceinline__ void env_group_fast(const DevWorld &W, const EnvGroupDesc &G, double lon,
                                               double lat, double z, float *out /*[MAXG]*/, const double *zt,
                                               ZBracket &zb_out, EnvExport *X = nullptr ODR_PT_PARAM, int f32idx = 0,
                                               const ProjStart ps = ProjStart()) {
  const DevSource &s = W.src[G.sid];
  const EnvFront fr = env_front<PROJ>(s, s.slot[G.geo_slot], lon, lat, z, f32idx, ps);
  env_group_sample<PROJ, BURST_ONLY, ZT>(W, G, env_global(G), fr, z, out, zt, zb_out, X ODR_PT_ARG);
}
// The records the main-loop sample fetched for slot A, as the kept footprint of the stage samples -- valid when slot A holds
// the current (an interleaved pair of the stage samples' dimensionality) and the sample's time bracket is that of the
// half-step stages `th` (wave-uniform tests)
template <bool IS3D>
__device__ __forceinline__ UVKeep<IS3D> uv_keep_from(const EnvGroupDesc &G, const EnvExport &X, const UVTime &th) {
  UVKeep<IS3D> K;
  const bool tl = G.ba != nullptr && !G.all_static;
  const float *mb = (const float *)((const char *)G.bb + G.ps_off[0]);
  const float *ma = tl ? (const float *)((const char *)G.ba + G.ps_off[0]) : nullptr;
  const bool fits = G.bs[0] == 0 && G.var[0] == VAR_U && G.ps_mode[0] == (IS3D ? ENV_P3 : ENV_P2) && mb == th.b && ma == th.a;
  K.valid = fits && X.valid;
  K.n00 = X.n00; K.n11 = X.n11; K.kb = IS3D ? (unsigned)X.iz0 * 8u : 0u;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    K.q.b[c].x = X.b[4 * c]; K.q.b[c].y = X.b[4 * c + 1]; K.q.a[c].x = X.a[4 * c]; K.q.a[c].y = X.a[4 * c + 1];
    if constexpr (IS3D) { K.q.b[c].z = X.b[4 * c + 2]; K.q.b[c].w = X.b[4 * c + 3]; K.q.a[c].z = X.a[4 * c + 2]; K.q.a[c].w = X.a[4 * c + 3]; }
  }
  return K;
}
// ODR_STAGE_FAST, 3-D: the same hand-over with the records combined over the bracket of the main-loop sample (uv_combine_z);
// z_unchanged: the stages sample at the depth of the main-loop sample (false when the sea floor lifted the element in between:
// another bracket, the kept values do not apply)
template <bool IS3D, int SM>
__device__ __forceinline__ UVKeep<IS3D> uv_keep_from_sm(const EnvGroupDesc &G, const EnvExport &X, const UVTime &th, const ZBracket &zb,
                                                        int nz, bool z_unchanged) {
  if constexpr (!(IS3D && SM == 1)) return uv_keep_from<IS3D>(G, X, th);
  else {
    UVKeep<IS3D> K;
    const bool tl = G.ba != nullptr && !G.all_static;
    const float *mb = (const float *)((const char *)G.bb + G.ps_off[0]);
    const float *ma = tl ? (const float *)((const char *)G.ba + G.ps_off[0]) : nullptr;
    const bool fits = G.bs[0] == 0 && G.var[0] == VAR_U && G.ps_mode[0] == ENV_P3 && mb == th.b && ma == th.a;
    K.valid = fits && X.valid && z_unchanged;
    K.n00 = X.n00; K.n11 = X.n11; K.kb = (unsigned)X.iz0 * 8u;
    // X.combine: the burst already combined the corners over the bracket (time level `a` = `b` when the sample is on a level:
    // the loader reads the same records for both)
#pragma unroll
    for (int c = 0; c < 4; ++c) { K.q.b[c].x = X.b[4 * c]; K.q.b[c].y = X.b[4 * c + 1]; K.q.b[c].z = X.b[4 * c + 2]; K.q.b[c].w = X.b[4 * c + 3]; }
    return K;
  }
}
// for the kernels that need neither LDS tables nor the bracket
template <int PROJ>
__device__ __forceinline__ void env_group_fast(const DevWorld &W, const EnvGroupDesc &G, double lon, double lat, double z, float *out,
                                               int f32idx = 0) {
  ZBracket zb;
  env_group_fast<PROJ, false, false>(W, G, lon, lat, z, out, nullptr, zb, nullptr ODR_PT_NULLARG, f32idx);
}

}  // namespace odr
