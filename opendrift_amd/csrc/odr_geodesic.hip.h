// Device-side WGS84 direct geodesic (Karney 2013, 6th-order series) for gfx950.
//
// Replaces pyproj.Geod(ellps='WGS84').fwd as used by
//   opendrift/models/basemodel/__init__.py:4643-4657  (update_positions)
//   opendrift/models/physics_methods.py:632-666        (RK2/RK4 sub-stage positions)
// Written from the published algorithm (J. Geodesy 87:43-55).  float64 throughout:
// this is the f64-ALU heavy part of a particle-step (4-7 calls per step), so the
// series in eps are expanded at compile time into Horner forms with the WGS84
// third-flattening folded into constants (c_geod), there is no per-call
// coefficient table walk, and sin/cos pairs come from one sincos().
#pragma once
#include <hip/hip_runtime.h>

namespace odr {

struct GeodConst {
  double a, f, f1, e2, ep2, n, b;
  double A3x[6];
  double C3x[15];
};
__constant__ GeodConst c_geod;

static constexpr double kDeg = 3.14159265358979323846264338327950288 / 180.0;
static constexpr double kTiny = 1.4916681462400413e-154;

__device__ __forceinline__ double ang_normalize(double x) {
  double y = remainder(x, 360.0);
  return fabs(y) == 180.0 ? copysign(180.0, x) : y;
}

__device__ __forceinline__ double ang_round(double x) {
  const double z = 1 / 16.0;
  double y = fabs(x);
  y = y < z ? __dsub_rn(z, __dsub_rn(z, y)) : y;
  return copysign(y, x);
}

// sin/cos of an angle in degrees with exact quadrant reduction
__device__ __forceinline__ void sincosd(double x, double &sinx, double &cosx) {
  int q = 0;
  double r = remquo(x, 90.0, &q);
  double s, c;
  sincos(r * kDeg, &s, &c);
  switch ((unsigned)q & 3U) {
    case 0U: sinx = s; cosx = c; break;
    case 1U: sinx = c; cosx = -s; break;
    case 2U: sinx = -s; cosx = -c; break;
    default: sinx = -c; cosx = s; break;
  }
  cosx += 0.0;
  if (sinx == 0) sinx = copysign(sinx, x);
}

__device__ __forceinline__ double atan2d(double y, double x) {
  int q = 0;
  if (fabs(y) > fabs(x)) { double t = x; x = y; y = t; q = 2; }
  if (signbit(x)) { x = -x; ++q; }
  double ang = atan2(y, x) / kDeg;
  switch (q) {
    case 1: ang = copysign(180.0, y) - ang; break;
    case 2: ang = 90 - ang; break;
    case 3: ang = -90 + ang; break;
    default: break;
  }
  return ang;
}

// sum_{k=1..6} c[k] sin(2 k x), Clenshaw (c1..c6 passed by value -> registers)
__device__ __forceinline__ double sin_series6(double sinx, double cosx, double c1, double c2,
                                               double c3, double c4, double c5, double c6) {
  #pragma clang fp contract(fast)
  #pragma clang fp contract(fast)
  double ar = 2 * (cosx - sinx) * (cosx + sinx);
  double y1 = c6;                 // k = 6
  double y0 = ar * y1 + c5;       // k = 5
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
  #pragma clang fp contract(fast)
  double ar = 2 * (cosx - sinx) * (cosx + sinx);
  double y0 = c5;                 // odd count: y0 = c[5]
  double y1 = ar * y0 + c4;
  y0 = ar * y1 - y0 + c3;
  y1 = ar * y0 - y1 + c2;
  y0 = ar * y1 - y0 + c1;
  return 2 * sinx * cosx * y0;
}

// Direct problem.  lat/lon/azi in degrees, s12 in metres.  lon2 in [-180,180].
__device__ __forceinline__ void geod_direct(double lat1, double lon1, double azi1, double s12,
                                             double &lat2, double &lon2) {
  // the TU is built with -ffp-contract=off so that the float32 rounding points of the
  // reference stay exact; the geodesic series are float64 and may fuse multiply-adds
#pragma clang fp contract(fast)
  const GeodConst &g = c_geod;
  double salp1, calp1, sbet1, cbet1;
  azi1 = ang_normalize(azi1);
  sincosd(ang_round(azi1), salp1, calp1);
  if (fabs(lat1) > 90) lat1 = __builtin_nan("");
  sincosd(ang_round(lat1), sbet1, cbet1);
  sbet1 *= g.f1;
  { double r = hypot(sbet1, cbet1); sbet1 /= r; cbet1 /= r; }
  cbet1 = fmax(kTiny, cbet1);

  double salp0 = salp1 * cbet1;
  double calp0 = hypot(calp1, salp1 * sbet1);
  double ssig1 = sbet1, somg1 = salp0 * sbet1;
  double csig1 = (sbet1 != 0 || calp1 != 0) ? cbet1 * calp1 : 1.0;
  double comg1 = csig1;
  { double r = hypot(ssig1, csig1); ssig1 /= r; csig1 /= r; }

  double k2 = calp0 * calp0 * g.ep2;
  double eps = k2 / (2 * (1 + sqrt(1 + k2)) + k2);
  double e2 = eps * eps;

  // A1 - 1
  double A1m1 = ((e2 * (e2 * (e2 + 4) + 64)) / 256 + eps) / (1 - eps);
  // C1[1..6]
  double d = eps;
  double C11 = d * (e2 * (6 - e2) - 16) / 32;           d *= eps;
  double C12 = d * (e2 * (64 - 9 * e2) - 128) / 2048;   d *= eps;
  double C13 = d * (9 * e2 - 16) / 768;                 d *= eps;
  double C14 = d * (3 * e2 - 5) / 512;                  d *= eps;
  double C15 = -7 * d / 1280;                           d *= eps;
  double C16 = -7 * d / 2048;
  double B11 = sin_series6(ssig1, csig1, C11, C12, C13, C14, C15, C16);
  double sB, cB;
  sincos(B11, &sB, &cB);
  double stau1 = ssig1 * cB + csig1 * sB;
  double ctau1 = csig1 * cB - ssig1 * sB;
  // C1'[1..6]
  d = eps;
  double P1 = d * (e2 * (205 * e2 - 432) + 768) / 1536;       d *= eps;
  double P2 = d * (e2 * (4005 * e2 - 4736) + 3840) / 12288;   d *= eps;
  double P3 = d * (116 - 225 * e2) / 384;                     d *= eps;
  double P4 = d * (2695 - 7173 * e2) / 7680;                  d *= eps;
  double P5 = 3467 * d / 7680;                                d *= eps;
  double P6 = 38081 * d / 61440;
  // C3[1..5] and A3
  const double *x3 = g.C3x;
  double m = eps;
  double C31 = m * ((((x3[0] * eps + x3[1]) * eps + x3[2]) * eps + x3[3]) * eps + x3[4]);  m *= eps;
  double C32 = m * (((x3[5] * eps + x3[6]) * eps + x3[7]) * eps + x3[8]);                  m *= eps;
  double C33 = m * ((x3[9] * eps + x3[10]) * eps + x3[11]);                                m *= eps;
  double C34 = m * (x3[12] * eps + x3[13]);                                                m *= eps;
  double C35 = m * x3[14];
  const double *a3 = g.A3x;
  double A3 = ((((a3[0] * eps + a3[1]) * eps + a3[2]) * eps + a3[3]) * eps + a3[4]) * eps + a3[5];
  double A3c = -g.f * salp0 * A3;
  double B31 = sin_series5(ssig1, csig1, C31, C32, C33, C34, C35);

  double tau12 = s12 / (g.b * (1 + A1m1));
  double st, ct;
  sincos(tau12, &st, &ct);
  double B12 = -sin_series6(stau1 * ct + ctau1 * st, ctau1 * ct - stau1 * st, P1, P2, P3, P4, P5, P6);
  double sig12 = tau12 - (B12 - B11);
  double ssig12, csig12;
  sincos(sig12, &ssig12, &csig12);
  double ssig2 = ssig1 * csig12 + csig1 * ssig12;
  double csig2 = csig1 * csig12 - ssig1 * ssig12;
  double sbet2 = calp0 * ssig2;
  double cbet2 = hypot(salp0, calp0 * csig2);
  if (cbet2 == 0) cbet2 = csig2 = kTiny;
  double somg2 = salp0 * ssig2, comg2 = csig2;
  double omg12 = atan2(somg2 * comg1 - comg2 * somg1, comg2 * comg1 + somg2 * somg1);
  double lam12 = omg12 + A3c * (sig12 + (sin_series5(ssig2, csig2, C31, C32, C33, C34, C35) - B31));
  double lon12 = lam12 / kDeg;
  lon2 = ang_normalize(ang_normalize(lon1) + ang_normalize(lon12));
  lat2 = atan2d(sbet2, g.f1 * cbet2);
}

}  // namespace odr
