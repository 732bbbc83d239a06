// Device-side WGS84 direct geodesic (Karney 2013, 6th-order series) for gfx950.
//
// Replaces pyproj.Geod(ellps='WGS84').fwd as used by
//   opendrift/models/basemodel/__init__.py:4643-4657  (update_positions)
//   opendrift/models/physics_methods.py:632-666        (RK2/RK4 sub-stage positions)
// Written from the published algorithm (J. Geodesy 87:43-55).  This is the f64-ALU
// heavy part of a particle-step (4-7 calls per step), so it is organised for the
// CDNA4 vector f64 pipe rather than for a CPU:
//   * the series in eps are Horner forms with reciprocal constants; the remaining
//     float64 divides / square roots use the v_rcp_f64 / v_rsq_f64 seeds + two Newton
//     steps (~8 instructions instead of ~19 / ~31 for the IEEE expansions);
//   * everything that depends only on the start point (reduced latitude) lives in a
//     GeodOrigin that the RK sub-stages of one particle share;
//   * angle reductions are exact fma reductions instead of remquo()/remainder()
//     library loops; hypot() is sqrt(fma) (arguments are O(1));
//   * every sin/cos is taken on a reduced range |x| <= pi/4 (exact quadrant reduction
//     in degrees; B11, tau12, sig12 are < pi/4 for steps < 5000 km) with the classic
//     minimax kernels -- no library range reduction; the atan of the small latitude /
//     longitude increments is a Gregory series when |ratio| < 1/16.
// The result equals the plain algorithm to float64 round-off (tests/test_gpu_parity.py:
// <= 1e-11 deg against the CPU oracle over 1 mm ... 1100 km, poles included; measured
// 1.1e-13 deg).
#pragma once
#include <hip/hip_runtime.h>

namespace odr {

struct GeodConst {
  double a, f, f1, e2, ep2, n, b, ib;  // ib = 1/b
  double A3x[6];
  double C3x[15];
};
constexpr GeodConst geod_wgs84() {
  GeodConst g{};
  // WGS84 (pyproj.Geod(ellps='WGS84')); A3/C3 coefficient polynomials in n, Karney (2013) eqs. 24-25
  g.a = 6378137.0;
  g.f = 1 / 298.257223563;
  g.f1 = 1 - g.f;
  g.e2 = g.f * (2 - g.f);
  g.ep2 = g.e2 / (g.f1 * g.f1);
  g.n = g.f / (2 - g.f);
  g.b = g.a * g.f1;
  g.ib = 1.0 / g.b;
  const double n = g.n;
  g.A3x[0] = -3.0 / 128;
  g.A3x[1] = (-2 * n - 3) / 64;
  g.A3x[2] = ((-n - 3) * n - 1) / 16;
  g.A3x[3] = ((3 * n - 1) * n - 2) / 8;
  g.A3x[4] = (n - 1) / 2;
  g.A3x[5] = 1;
  double *c = g.C3x;
  c[0] = 3.0 / 128;              c[1] = (2 * n + 5) / 128;        c[2] = ((-n + 3) * n + 3) / 64;
  c[3] = ((-n + 0) * n + 1) / 8; c[4] = (-n + 1) / 4;
  c[5] = 5.0 / 256;              c[6] = (n + 3) / 128;            c[7] = ((-3 * n - 2) * n + 3) / 64;
  c[8] = ((n - 3) * n + 2) / 32;
  c[9] = 7.0 / 512;              c[10] = (-10 * n + 9) / 384;     c[11] = ((5 * n - 9) * n + 5) / 192;
  c[12] = 7.0 / 512;             c[13] = (-14 * n + 7) / 512;
  c[14] = 21.0 / 2560;
  return g;
}
// one copy per translation unit (no relocatable device code), initialised at compile time
static __constant__ GeodConst c_geod = geod_wgs84();

static constexpr double kDeg = 3.14159265358979323846264338327950288 / 180.0;
static constexpr double kDegLo = 2.9486522708701687e-19;  // pi/180 - kDeg
static constexpr double kRad2Deg = 180.0 / 3.14159265358979323846264338327950288;
static constexpr double kTiny = 1.4916681462400413e-154;

// a * b + k for a literal / constant-memory k.  (Tried: VOP3 v_fma_f64 with k as a scalar-register addend via
// inline asm, which removes the two v_mov_b32 the compiler spends per 64-bit literal -- 12 % fewer vector
// instructions, but 4 % SLOWER in an A/B run on the same box: the Horner chains are latency-bound at ~4 waves
// per SIMD and the moves were filling their bubbles.  ODR_SGPR_FMA keeps the experiment buildable.)
__device__ __forceinline__ double fma_k(double a, double b, double k) {
#ifdef ODR_SGPR_FMA
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
  return d;
#else
  return fma(a, b, k);
#endif
}

// float64 reciprocal / reciprocal square root from the hardware seeds (v_rcp_f64 / v_rsq_f64)
// plus two Newton steps: ~8 instructions instead of the ~19 (divide) / ~31 (sqrt) of the IEEE
// expansions.  Used only inside the geodesic / projection code, whose results are compared
// to round-off (not bit for bit) -- never at the reference's float32/float64 rounding points.
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
  r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
  return r;
}
__device__ __forceinline__ double fast_sqrt(double x) { return x > 0 ? x * fast_rsqrt(x) : 0.0; }

// sin and cos on the reduced range |x| <= pi/4 (minimax kernels of the classic fdlibm k_sin/k_cos,
// < 1 ulp): no range reduction, ~25 instructions for the pair
__device__ __forceinline__ void sincos_q(double x, double &s, double &c) {
#pragma clang fp contract(fast)
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double rs = fma_k(fma_k(fma_k(fma_k(z, S6, S5), z, S4), z, S3), z, S2);
  s = x + (z * x) * fma_k(z, rs, S1);
  double rc = z * fma_k(fma_k(fma_k(fma_k(fma_k(z, C6, C5), z, C4), z, C3), z, C2), z, C1);
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  c = w + (((1.0 - w) - hz) + z * rc);
}

// sin and cos for |x| <~ 3.3 (longitude differences wrapped to [-pi, pi], latitudes): quadrant k = rint(x 2/pi) in -2 .. 2,
// remainder by a two-term Cody-Waite subtraction (k pi/2: the 53-bit head is exact for |k| <= 2), the pair from the reduced-range
// kernels -- < 1 ulp like the library's sincos, ~45 instead of ~65 instructions (no large-argument path, one call for the pair)
__device__ __forceinline__ void sincos_pi(double x, double &sinx, double &cosx) {
  const double k = rint(x * 0.63661977236758134308);
  double r = fma(-k, 1.57079632679489655800e+00, x);
  r = fma(-k, 6.12323399573676603587e-17, r);
  double s, c;
  sincos_q(r, s, c);
  switch ((unsigned)(int)k & 3U) {
    case 0U: sinx = s; cosx = c; break;
    case 1U: sinx = c; cosx = -s; break;
    case 2U: sinx = -s; cosx = -c; break;
    default: sinx = -c; cosx = s; break;
  }
}
// exp for |x| <= 0.0101 (e atanh(e sin phi) of an earth-like ellipsoid, e < 0.1): Taylor series, x^8 / 8! < 3e-21
__device__ __forceinline__ double exp_small(double x) {
#pragma clang fp contract(fast)
  return fma(x, fma(x, fma(x, fma(x, fma(x, fma(x, fma(x, 1.0 / 5040, 1.0 / 720), 1.0 / 120), 1.0 / 24), 1.0 / 6), 0.5), 1.0), 1.0);
}

// ln x for a positive, finite, normal x: x = m 2^e with m in [sqrt(1/2), sqrt 2), ln m = 2 atanh((m - 1) / (m + 1)) by its series
// (|s| <= 0.1716: s^22 / 23 < 7e-19) -- ~25 instructions, <= 3 ulp (tests/test_geod_host.py; the library's ln: ~50 with its special cases)
__device__ __forceinline__ double log_pos(double x) {
#pragma clang fp contract(fast)
  int e;
  double m = frexp(x, &e);
  if (m < 0.70710678118654752440) { m *= 2; e -= 1; }
  const double sq = (m - 1) * fast_rcp(m + 1), q = sq * sq;
  const double poly = fma(q, fma(q, fma(q, fma(q, fma(q, fma(q, fma(q, fma(q, fma(q, fma(q, 1.0 / 21, 1.0 / 19), 1.0 / 17), 1.0 / 15), 1.0 / 13),
                                                                 1.0 / 11), 1.0 / 9), 1.0 / 7), 1.0 / 5), 1.0 / 3), 1.0);
  return fma((double)e, 0.69314718055994530942, 2 * sq * poly);
}

// AngNormalize: reduce to [-180, 180]; x - 360*rint(x/360) is exact in float64
__device__ __forceinline__ double ang_normalize(double x) {
  double q = rint(x * (1.0 / 360.0));
  double y = fma(-360.0, q, x);
  y = y > 180.0 ? y - 360.0 : y;
  y = y < -180.0 ? y + 360.0 : y;
  return fabs(y) == 180.0 ? copysign(180.0, x) : y;
}

__device__ __forceinline__ double ang_round(double x) {
  const double z = 1 / 16.0;
  double y = fabs(x);
  y = y < z ? __dsub_rn(z, __dsub_rn(z, y)) : y;
  return copysign(y, x);
}

// sin/cos of an angle in degrees (|x| <= 540) with exact quadrant reduction
__device__ __forceinline__ void sincosd(double x, double &sinx, double &cosx) {
  double qd = rint(x * (1.0 / 90.0));
  double r = fma(-90.0, qd, x);  // exact
  int q = (int)qd;
  double s, c;
  sincos_q(r * kDeg, s, c);  // |r| <= 45 deg
  switch ((unsigned)q & 3U) {
    case 0U: sinx = s; cosx = c; break;
    case 1U: sinx = c; cosx = -s; break;
    case 2U: sinx = -s; cosx = -c; break;
    default: sinx = -c; cosx = s; break;
  }
}

__device__ __forceinline__ double atan2d(double y, double x) {
  int q = 0;
  if (fabs(y) > fabs(x)) { double t = x; x = y; y = t; q = 2; }
  if (signbit(x)) { x = -x; ++q; }
  double ang = atan2(y, x) * kRad2Deg;
  switch (q) {
    case 1: ang = copysign(180.0, y) - ang; break;
    case 2: ang = 90 - ang; break;
    case 3: ang = -90 + ang; break;
    default: break;
  }
  return ang;
}

// sin and cos of the small angles tau12, sig12: Taylor forms for |x| <= 1/64 (steps up to ~99 km:
// truncation x^9/9! < 1e-20 relative, x^8/8! < 9e-20), the reduced-range kernel up to pi/4 (steps
// up to ~5000 km), else the library call
__device__ __forceinline__ void sincos_small(double x, double &s, double &c) {
#pragma clang fp contract(fast)
  if (fabs(x) <= 0.015625) {
    double z = x * x;
    s = fma(x * z, fma_k(fma_k(z, -1.0 / 5040, 1.0 / 120), z, -1.0 / 6), x);
    c = fma(z, fma_k(fma_k(z, -1.0 / 720, 1.0 / 24), z, -0.5), 1.0);
  } else if (fabs(x) <= 0.78539816339744830962) sincos_q(x, s, c);
  else sincos(x, &s, &c);
}
// |x| <= 1e-3 (B11 = sum C1k sin 2k sigma, |B11| <= eps/2 < 8.4e-4): truncation < 2e-22
__device__ __forceinline__ void sincos_tiny(double x, double &s, double &c) {
#pragma clang fp contract(fast)
  double z = x * x;
  s = fma(x * z, fma_k(z, 1.0 / 120, -1.0 / 6), x);
  c = fma(z, fma_k(z, 1.0 / 24, -0.5), 1.0);
}

// atan2(y, x) for x > 0 and |y/x| <= 1/16 by the Gregory series (truncation < 1e-20)
__device__ __forceinline__ double atan_ratio(double y, double x) {
#pragma clang fp contract(fast)
  if (x > 0 && fabs(y) <= 0.0625 * x) {
    double r = y * fast_rcp(x), r2 = r * r;
    double g = fma_k(fma_k(fma_k(fma_k(fma_k(fma_k(r2, -1.0 / 15, 1.0 / 13), r2, -1.0 / 11), r2, 1.0 / 9), r2, -1.0 / 7), r2, 1.0 / 5), r2, -1.0 / 3);
    return fma(r * r2, g, r);
  }
  return atan2(y, x);
}

// atan2 for finite arguments that are not both zero, to < 1 ulp: r = min/max by a Newton reciprocal with one
// exact-residual correction, atan(r) = r + r z q(z), z = r^2, q = degree-19 interpolant of (atan(r)/r - 1)/z
// on [0, 1] (max relative error 1.6e-17), octant reconstruction.  The coefficients live in constant memory and
// reach the FMAs as scalar operands: the library routine spends 46 of its 102 vector instructions on
// materialising 64-bit literals.
static __constant__ double c_atanq[20] = {
    -0.33333333333333330252, 0.19999999999997532204, -0.14285714285384131545, 0.11111111093490828096,
    -0.090909085908919342252, 0.07692298971033216917, -0.066665646992891042112, 0.058815068777936560666,
    -0.05257973334284110807, 0.047377495795277789723, -0.042603566326016521328, 0.037494868125352471116,
    -0.03127718906699638479, 0.023696731580048622013, -0.015535152475414176125, 0.0083689311784501621357,
    -0.00349588597391630952, 0.0010496035084968515073, -0.00019996189377901382062, 0.000018061954618612152202};
__device__ __forceinline__ double atan2_fin(double y, double x) {
#pragma clang fp contract(fast)
  const double ay = fabs(y), ax = fabs(x);
  const double mx = fmax(ay, ax), mn = fmin(ay, ax);
  const double rr = fast_rcp(mx);
  double r = mn * rr;
  r = fma(fma(-mx, r, mn), rr, r);
  const double z = r * r;
#ifdef ODR_ATAN_HORNER
  double q = c_atanq[19];
#pragma unroll
  for (int k = 18; k >= 0; --k) q = fma_k(q, z, c_atanq[k]);
#else
  // even / odd halves in w = z^2: two independent Horner chains of depth 9 instead of one of depth 19
  // (the evaluation is latency-bound, not issue-bound)
  const double w = z * z;
  double qe = c_atanq[18], qo = c_atanq[19];
#pragma unroll
  for (int k = 16; k >= 0; k -= 2) { qe = fma_k(qe, w, c_atanq[k]); qo = fma_k(qo, w, c_atanq[k + 1]); }
  double q = fma(qo, z, qe);
#endif
  double a = fma(r * z, q, r);
  a = ay > ax ? 1.57079632679489661923 - a : a;
  a = signbit(x) ? 3.14159265358979323846 - a : a;
  return copysign(a, y);
}

// sum_{k=1..6} c[k] sin(2 k x), Clenshaw
__device__ __forceinline__ double sin_series6(double sinx, double cosx, double c1, double c2,
                                               double c3, double c4, double c5, double c6) {
#pragma clang fp contract(fast)
  double ar = 2 * (cosx - sinx) * (cosx + sinx);
  double y1 = c6;
  double y0 = ar * y1 + c5;
  y1 = ar * y0 - y1 + c4;
  y0 = ar * y1 - y0 + c3;
  y1 = ar * y0 - y1 + c2;
  y0 = ar * y1 - y0 + c1;
  return 2 * sinx * cosx * y0;
}
// sum_{k=1..5}
__device__ __forceinline__ double sin_series5(double sinx, double cosx, double c1, double c2,
                                               double c3, double c4, double c5) {
#pragma clang fp contract(fast)
  double ar = 2 * (cosx - sinx) * (cosx + sinx);
  double y0 = c5;
  double y1 = ar * y0 + c4;
  y0 = ar * y1 - y0 + c3;
  y1 = ar * y0 - y1 + c2;
  y0 = ar * y1 - y0 + c1;
  return 2 * sinx * cosx * y0;
}

// start point of a geodesic: what geod_lineinit derives from (lat1, lon1) alone
struct GeodOrigin {
  double lat1, lon1n, sbet1, cbet1, tanphi1;
};

__device__ __forceinline__ GeodOrigin geod_origin(double lat1, double lon1) {
#pragma clang fp contract(fast)
  GeodOrigin o;
  if (fabs(lat1) > 90) lat1 = __builtin_nan("");
  o.lat1 = lat1;
  o.lon1n = ang_normalize(lon1);
  double sphi, cphi;
  sincosd(ang_round(lat1), sphi, cphi);
  double sb = sphi * c_geod.f1;
  double inv = fast_rsqrt(sb * sb + cphi * cphi);
  o.sbet1 = sb * inv;
  o.cbet1 = fmax(kTiny, cphi * inv);
  o.tanphi1 = sphi / fmax(kTiny, cphi);
  return o;
}

// Direct problem from a prepared origin and the sine / cosine of the azimuth.  s12 in metres.
// lon2 in [-180,180].
__device__ __forceinline__ void geod_direct_sc(const GeodOrigin &o, double salp1, double calp1, double s12,
                                                double &lat2, double &lon2) {
  // the TU is built with -ffp-contract=off so that the float32 rounding points of the
  // reference stay exact; the geodesic series are float64 and may fuse multiply-adds
#pragma clang fp contract(fast)
  const GeodConst &g = c_geod;
#ifdef ODR_ABLATE_GEOD   // what-if build: a flat-earth step instead of the geodesic (tools/ab_bench.sh)
  lat2 = o.lat1 + s12 * calp1 * 8.98e-6;
  lon2 = o.lon1n + s12 * salp1 * 8.98e-6 * o.cbet1;
  return;
#endif
  const double sbet1 = o.sbet1, cbet1 = o.cbet1;

  double salp0 = salp1 * cbet1;
  double t0 = salp1 * sbet1;
  double calp0 = fast_sqrt(calp1 * calp1 + t0 * t0);
  double ssig1 = sbet1, somg1 = salp0 * sbet1;
  double csig1 = (sbet1 != 0 || calp1 != 0) ? cbet1 * calp1 : 1.0;
  double comg1 = csig1;
  { double inv = fast_rsqrt(ssig1 * ssig1 + csig1 * csig1); ssig1 *= inv; csig1 *= inv; }

  // eps = k2 / (2 (1 + sqrt(1 + k2)) + k2) by its Maclaurin series: k2 <= e'^2 = 0.00674, the x^8
  // term is < 1e-19 (no square root, no reciprocal)
  double k2 = calp0 * calp0 * g.ep2;
  double eps = k2 * fma_k(fma_k(fma_k(fma_k(fma_k(fma_k(k2, 429.0 / 16384, -33.0 / 1024), k2, 21.0 / 512), k2, -7.0 / 128), k2, 5.0 / 64), k2, -0.125), k2, 0.25);
  double e2 = eps * eps;
  // 1 / (1 + A1m1) = (1 - eps) / (1 + eps^2/4 + eps^4/64 + eps^6/256)
  //               = (1 - eps) (1 - eps^2/4 + 3 eps^4/64) + O(eps^6 = 2e-17)
  double iA1 = (1 - eps) * fma(e2, fma_k(e2, 3.0 / 64, -0.25), 1.0);
  double d = eps;
  double C11 = d * fma_k(e2, 6 - e2, -16.0) * (1.0 / 32);                 d *= eps;
  double C12 = d * fma_k(e2, fma_k(e2, -9.0, 64.0), -128.0) * (1.0 / 2048);  d *= eps;
  double C13 = d * fma_k(e2, 9.0, -16.0) * (1.0 / 768);                   d *= eps;
  double C14 = d * fma_k(e2, 3.0, -5.0) * (1.0 / 512);                    d *= eps;
  double C15 = d * (-7.0 / 1280);                               d *= eps;
  double C16 = d * (-7.0 / 2048);
  double B11 = sin_series6(ssig1, csig1, C11, C12, C13, C14, C15, C16);
  double sB, cB;
  sincos_tiny(B11, sB, cB);
  double stau1 = ssig1 * cB + csig1 * sB;
  double ctau1 = csig1 * cB - ssig1 * sB;
  d = eps;
  double P1 = d * fma_k(e2, fma_k(e2, 205.0, -432.0), 768.0) * (1.0 / 1536);     d *= eps;
  double P2 = d * fma_k(e2, fma_k(e2, 4005.0, -4736.0), 3840.0) * (1.0 / 12288); d *= eps;
  double P3 = d * fma_k(e2, -225.0, 116.0) * (1.0 / 384);                        d *= eps;
  double P4 = d * fma_k(e2, -7173.0, 2695.0) * (1.0 / 7680);                     d *= eps;
  double P5 = d * (3467.0 / 7680);                                    d *= eps;
  double P6 = d * (38081.0 / 61440);
  const double *x3 = g.C3x;
  double m = eps;
  double C31 = m * fma_k(fma_k(fma_k(fma_k(eps, x3[0], x3[1]), eps, x3[2]), eps, x3[3]), eps, x3[4]);  m *= eps;
  double C32 = m * fma_k(fma_k(fma_k(eps, x3[5], x3[6]), eps, x3[7]), eps, x3[8]);                     m *= eps;
  double C33 = m * fma_k(fma_k(eps, x3[9], x3[10]), eps, x3[11]);                                      m *= eps;
  double C34 = m * fma_k(eps, x3[12], x3[13]);                                                         m *= eps;
  double C35 = m * x3[14];
  const double *a3 = g.A3x;
  double A3 = fma_k(fma_k(fma_k(fma_k(fma_k(eps, a3[0], a3[1]), eps, a3[2]), eps, a3[3]), eps, a3[4]), eps, a3[5]);
  double A3c = -g.f * salp0 * A3;
  double B31 = sin_series5(ssig1, csig1, C31, C32, C33, C34, C35);

  double tau12 = s12 * (g.ib * iA1);
  double st, ct;
  sincos_small(tau12, st, ct);
  double B12 = -sin_series6(stau1 * ct + ctau1 * st, ctau1 * ct - stau1 * st, P1, P2, P3, P4, P5, P6);
  double sig12 = tau12 - (B12 - B11);
  double ssig12, csig12;
  sincos_small(sig12, ssig12, csig12);
  double ssig2 = ssig1 * csig12 + csig1 * ssig12;
  double csig2 = csig1 * csig12 - ssig1 * ssig12;
  double sbet2 = calp0 * ssig2;
  double t1 = calp0 * csig2;
  double cbet2 = fast_sqrt(salp0 * salp0 + t1 * t1);
  if (cbet2 == 0) cbet2 = csig2 = kTiny;
  double somg2 = salp0 * ssig2, comg2 = csig2;
  double omg12 = atan_ratio(somg2 * comg1 - comg2 * somg1, comg2 * comg1 + somg2 * somg1);
  double lam12 = omg12 + A3c * (sig12 + (sin_series5(ssig2, csig2, C31, C32, C33, C34, C35) - B31));
  lon2 = ang_normalize(o.lon1n + ang_normalize(lam12 * kRad2Deg));
  // latitude: atan2d(sbet2, f1*cbet2); for short steps as lat1 + atan of the tangent-difference
  // ratio (same value to round-off, one atan2 fewer)
  double den = g.f1 * cbet2;
  double num = sbet2 - o.tanphi1 * den, dd = den + o.tanphi1 * sbet2;
  if (fabs(o.lat1) < 89.0 && dd > 0 && fabs(num) <= 0.0625 * dd)
    lat2 = o.lat1 + atan_ratio(num, dd) * kRad2Deg;
  else
    lat2 = atan2d(sbet2, den);
}

// ---------------------------------------------------------------- short steps: Legendre series
// A particle-step moves an element by metres to a few kilometres, 1e-6 ... 1e-3 of the earth radius.  For such
// steps the direct problem is the Taylor series of the geodesic equations on the ellipsoid about the start point
//   dphi/ds = cos(alpha) V^3 / c,  dlam/ds = sin(alpha) V / (c cos phi),  dalpha/ds = sin(alpha) tan(phi) V / c
// (V^2 = 1 + eta^2, eta^2 = e'^2 cos^2 phi, c = a / sqrt(1 - e^2) -- Legendre's series, e.g. Rapp, Geometric
// Geodesy I, sec. 6.3; the coefficients below were re-derived symbolically to 4th order, tools/derive_geodesic_series.py):
//   dphi / V^2   = u + a20 u^2 + a02 v^2 + a30 u^3 + a12 u v^2 + a40 u^4 + a22 u^2 v^2 + a04 v^4
//   dlam cos phi = v (1 + b11 u + b21 u^2 + b31 u^3 + (b03 + b13 u) v^2)
// with u = s cos(alpha) / N, v = s sin(alpha) / N (N = a / sqrt(1 - e^2 sin^2 phi)), t = tan phi:
//   a20 = -3/2 eta^2 t, a02 = -t/2, a30 = eta^2 (5 eta^2 t^2 - eta^2 + t^2 - 1) / 2, a12 = (9 eta^2 t^2 - eta^2 - 3 t^2 - 1) / 6,
//   a40 = -eta^2 t (35 eta^4 t^2 - 19 eta^4 + 15 eta^2 t^2 - 23 eta^2 - 4) / 8,
//   a22 = -t (45 eta^4 t^2 - 17 eta^4 - 9 eta^2 t^2 - 13 eta^2 + 6 t^2 + 4) / 12, a04 = -t a12 / 4,
//   b11 = t, b21 = (eta^2 + 3 t^2 + 1) / 3, b03 = -t^2 / 3, b31 = t (-eta^4 + eta^2 + 3 t^2 + 2) / 3, b13 = -t b21.
// The coefficients depend on the start latitude only: they are formed once per particle and step and serve all
// Runge-Kutta stage positions and the final move (~30 instructions each instead of ~275 for the full solution).
// Truncation: O(q^5), q = (s / N) max(1, |t|); measured against the full solution (tests/test_gpu_parity.py,
// tools/derive_geodesic_series.py --check): < 3e-12 deg for q <= kGeodLocalQ, i.e. steps up to 16 km at low
// latitudes, 9 km at 60 deg and 2.8 km at 80 deg.  Longer steps and start points within 1 deg of a pole take the full solution (geod_direct_sc).
static constexpr double kGeodLocalQ = 2.5e-3;
struct GeodLocal {
  double lat1, lon1n;        // start point (longitude normalised)
  double iN, qs;             // 1 / N; max(1, |t|) (validity scale), infinity within 1 deg of a pole
  double kphi, klam;         // V^2 * 180/pi;  180/pi / cos(phi)
  double t, a20, a30, a12, a40, a22, b21, b31;
};

// the coefficients from the sine and cosine of the start latitude
__device__ __forceinline__ GeodLocal geod_local_coeffs(double lat1, double lon1n, double sphi, double cphi) {
#pragma clang fp contract(fast)
  const GeodConst &g = c_geod;
  GeodLocal L;
  L.lat1 = lat1;
  L.lon1n = lon1n;
  const double ic = fast_rcp(fmax(kTiny, cphi));
  const double t = sphi * ic, t2 = t * t;
  const double h = g.ep2 * cphi * cphi, h2 = h * h;
  L.iN = fast_sqrt(1 - g.e2 * sphi * sphi) * (1.0 / g.a);
  L.qs = fabs(lat1) < 89.0 ? fmax(1.0, fabs(t)) : __builtin_inf();
  L.kphi = (1 + h) * kRad2Deg;
  L.klam = ic * kRad2Deg;
  L.t = t;
  L.a20 = -1.5 * h * t;
  L.a30 = 0.5 * h * (t2 * (5 * h + 1) - h - 1);
  L.a12 = (t2 * (9 * h - 3) - h - 1) * (1.0 / 6);
  L.a40 = -0.125 * h * t * (t2 * (35 * h2 + 15 * h) - 19 * h2 - 23 * h - 4);
  L.a22 = (-1.0 / 12) * t * (t2 * (45 * h2 - 9 * h + 6) - 17 * h2 - 13 * h + 4);
  L.b21 = (h + 3 * t2 + 1) * (1.0 / 3);
  L.b31 = (1.0 / 3) * t * (h - h2 + 3 * t2 + 2);
  return L;
}
__device__ __forceinline__ GeodLocal geod_local_origin_sc(double lat1, double lon1, double &sphi, double &cphi) {
  if (fabs(lat1) > 90) lat1 = __builtin_nan("");
  sincosd(ang_round(lat1), sphi, cphi);
  return geod_local_coeffs(lat1, ang_normalize(lon1), sphi, cphi);
}
__device__ __forceinline__ GeodLocal geod_local_origin(double lat1, double lon1) {
  double sphi, cphi;
  return geod_local_origin_sc(lat1, lon1, sphi, cphi);
}
// Start point of the NEXT move of the same element (advect_wind -> stokes_drift -> horizontal_diffusion: each an
// update_positions from where the previous one ended): the latitude a SERIES move just reached is within 2.6e-3 rad of the
// previous start latitude, so its sine and cosine follow from the previous pair by the addition theorem with sin d / cos d - 1
// as polynomials (d^7/5040, d^8/40320 < 2e-22: exact to float64 round-off) -- 12 instead of ~60 instructions for the degree
// reduction and the two minimax kernels of sincosd.  sphi / cphi: in = of the previous start latitude lat0, out = of lat1.
__device__ __forceinline__ GeodLocal geod_local_origin_next(double lat0, double lat1, double lon1, double &sphi, double &cphi) {
#pragma clang fp contract(fast)
  if (fabs(lat1) > 90) lat1 = __builtin_nan("");
  const double d = (lat1 - lat0) * kDeg, d2 = d * d;
  const double sd = fma(d * d2, fma(d2, 1.0 / 120, -1.0 / 6), d);                  // sin d
  const double cm = d2 * fma(d2, fma(d2, -1.0 / 720, 1.0 / 24), -0.5);              // cos d - 1
  const double s1 = sphi + fma(cphi, sd, sphi * cm), c1 = cphi + fma(-sphi, sd, cphi * cm);
  sphi = s1; cphi = c1;
  return geod_local_coeffs(lat1, ang_normalize(lon1), sphi, cphi);
}

// the full solution for the steps the series does not cover (rare: kept out of line)
// (the result comes back by value, in registers: reference outputs of an out-of-line call pin the callers' lon / lat
// to scratch memory for the whole kernel)
struct GeodLL { double lat, lon; };
__device__ __attribute__((noinline)) GeodLL geod_local_far(double lat1, double lon1n, double x, double y) {
  const GeodOrigin o = geod_origin(lat1, lon1n);
  const double s = sqrt(x * x + y * y);
  double salp = 0.0, calp = 1.0;
  if (s > 0) { salp = x / s; calp = y / s; }
  GeodLL r;
  geod_direct_sc(o, salp, calp, s, r.lat, r.lon);
  return r;
}

// end point of the geodesic that starts at the origin of L with azimuth atan2(x, y) and length hypot(x, y):
// x = east, y = north component of the step in metres
// (returns whether the series served the step: the next start point may then be formed by geod_local_origin_next)
__device__ __forceinline__ bool geod_local_move_ok(const GeodLocal &L, double x, double y, double &lat2, double &lon2) {
#pragma clang fp contract(fast)
  const double u = y * L.iN, v = x * L.iN;
  const double u2 = u * u, v2 = v * v;
  if (!((u2 + v2) * (L.qs * L.qs) <= kGeodLocalQ * kGeodLocalQ)) {   // also NaN steps
    if (u2 + v2 == u2 + v2 && L.lat1 == L.lat1) { const GeodLL r = geod_local_far(L.lat1, L.lon1n, x, y); lat2 = r.lat; lon2 = r.lon; return false; }
    lat2 = lon2 = __builtin_nan("");
    return false;
  }
  const double t = L.t;
  const double a02 = -0.5 * t, a04 = -0.25 * t * L.a12, b03 = (-1.0 / 3) * t * t, b13 = -t * L.b21;
  const double pu = fma(fma(L.a40, u, L.a30), u, L.a20);                       // a20 + a30 u + a40 u^2
  const double pv = fma(a04, v2, fma(fma(L.a22, u, L.a12), u, a02));           // a02 + a12 u + a22 u^2 + a04 v^2
  const double p = fma(pv, v2, fma(pu, u2, u));
  const double lu = fma(fma(fma(L.b31, u, L.b21), u, t), u, 1.0);              // 1 + b11 u + b21 u^2 + b31 u^3
  const double l = v * fma(fma(b13, u, b03), v2, lu);
  lat2 = fma(L.kphi, p, L.lat1);
  lon2 = ang_normalize(fma(L.klam, l, L.lon1n));
  return true;
}
__device__ __forceinline__ void geod_local_move(const GeodLocal &L, double x, double y, double &lat2, double &lon2) {
  (void)geod_local_move_ok(L, x, y, lat2, lon2);
}

// azimuth in degrees (float64 callers: advect_wind, stokes_drift, horizontal diffusion)
__device__ __forceinline__ void geod_direct_from(const GeodOrigin &o, double azi1, double s12,
                                                  double &lat2, double &lon2) {
  double salp1, calp1;
  azi1 = ang_normalize(azi1);
  sincosd(ang_round(azi1), salp1, calp1);
  geod_direct_sc(o, salp1, calp1, s12, lat2, lon2);
}

__device__ __forceinline__ void geod_direct(double lat1, double lon1, double azi1, double s12,
                                             double &lat2, double &lon2) {
  GeodOrigin o = geod_origin(lat1, lon1);
  geod_direct_from(o, azi1, s12, lat2, lon2);
}

}  // namespace odr
