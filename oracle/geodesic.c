/* TEST INFRASTRUCTURE ONLY -- CPU oracle, never on the product path.
 *
 * Karney (2013) direct geodesic, 6th-order series, restated from the published
 * algorithm (J. Geodesy 87:43-55, eqs. 7-26 for the auxiliary sphere, eqs.
 * 15-25 for the A1/C1/C1'/A3/C3 series).  This is the arithmetic behind
 * pyproj.Geod(ellps='WGS84').fwd as called from
 *   opendrift/models/basemodel/__init__.py:4643-4657 and
 *   opendrift/models/physics_methods.py:632-666.
 * pyproj/PROJ are NOT vendored in the reference tree; parity of this file is
 * pinned in tests/test_oracle_golden.py (test_geodesic_*) against (i) an independent mpmath
 * 40-digit integration of the geodesic ODE on the ellipsoid, (ii) the
 * reference's own coarse known answers (tests/models/test_models.py:61-64,
 * tests/models/test_environment.py:30-51).
 */
#include "geodesic.h"
#include <float.h>
#include <math.h>

#define NA3 6
#define NC1 6
#define NC3 6

static const double degree = 3.14159265358979323846264338327950288 / 180.0;
static const double tiny = 1.4916681462400413e-154; /* sqrt(DBL_MIN) */

static double sq(double x) { return x * x; }

static double polyval(int N, const double *p, double x) {
  double y = N < 0 ? 0 : *p++;
  while (--N >= 0) y = y * x + *p++;
  return y;
}

static void norm2(double *s, double *c) {
  double r = hypot(*s, *c);
  *s /= r;
  *c /= r;
}

static double ang_normalize(double x) {
  double y = remainder(x, 360.0);
  return fabs(y) == 180 ? copysign(180.0, x) : y;
}

static double lat_fix(double x) { return fabs(x) > 90 ? NAN : x; }

static double ang_round(double x) {
  const double z = 1 / 16.0;
  volatile double y = fabs(x);
  volatile double w = z - y;
  y = y < z ? z - w : y;
  return copysign(y, x);
}

static void sincosd(double x, double *sinx, double *cosx) {
  int q = 0;
  double r = remquo(x, 90.0, &q);
  double s, c;
  r *= degree;
  s = sin(r);
  c = cos(r);
  switch ((unsigned)q & 3U) {
    case 0U: *sinx = s; *cosx = c; break;
    case 1U: *sinx = c; *cosx = -s; break;
    case 2U: *sinx = -s; *cosx = -c; break;
    default: *sinx = -c; *cosx = s; break;
  }
  *cosx += 0;
  if (*sinx == 0) *sinx = copysign(*sinx, x);
}

static double atan2d(double y, double x) {
  int q = 0;
  double ang;
  if (fabs(y) > fabs(x)) { double t = x; x = y; y = t; q = 2; }
  if (signbit(x)) { x = -x; ++q; }
  ang = atan2(y, x) / degree;
  switch (q) {
    case 1: ang = copysign(180.0, y) - ang; break;
    case 2: ang = 90 - ang; break;
    case 3: ang = -90 + ang; break;
    default: break;
  }
  return ang;
}

/* sum_{k=1..n} c[k] sin(2 k x) by Clenshaw summation (c[0] unused). */
static double sin_series(double sinx, double cosx, const double *c, int n) {
  double ar = 2 * (cosx - sinx) * (cosx + sinx), y0, y1 = 0;
  c += n + 1;
  y0 = (n & 1) ? *--c : 0;
  n /= 2;
  while (n--) {
    y1 = ar * y0 - y1 + *--c;
    y0 = ar * y1 - y0 + *--c;
  }
  return 2 * sinx * cosx * y0;
}

static double A1m1f(double eps) {
  static const double coeff[] = {1, 4, 64, 0, 256};
  double t = polyval(3, coeff, sq(eps)) / coeff[4];
  return (t + eps) / (1 - eps);
}

static void C1f(double eps, double c[]) {
  static const double coeff[] = {
      -1, 6, -16, 32,       /* C1[1]/eps^1 */
      -9, 64, -128, 2048,   /* C1[2]/eps^2 */
      9, -16, 768,          /* C1[3]/eps^3 */
      3, -5, 512,           /* C1[4]/eps^4 */
      -7, 1280,             /* C1[5]/eps^5 */
      -7, 2048,             /* C1[6]/eps^6 */
  };
  double eps2 = sq(eps), d = eps;
  int o = 0, l;
  for (l = 1; l <= NC1; ++l) {
    int m = (NC1 - l) / 2;
    c[l] = d * polyval(m, coeff + o, eps2) / coeff[o + m + 1];
    o += m + 2;
    d *= eps;
  }
}

static void C1pf(double eps, double c[]) {
  static const double coeff[] = {
      205, -432, 768, 1536,       /* C1p[1]/eps^1 */
      4005, -4736, 3840, 12288,   /* C1p[2]/eps^2 */
      -225, 116, 384,             /* C1p[3]/eps^3 */
      -7173, 2695, 7680,          /* C1p[4]/eps^4 */
      3467, 7680,                 /* C1p[5]/eps^5 */
      38081, 61440,               /* C1p[6]/eps^6 */
  };
  double eps2 = sq(eps), d = eps;
  int o = 0, l;
  for (l = 1; l <= NC1; ++l) {
    int m = (NC1 - l) / 2;
    c[l] = d * polyval(m, coeff + o, eps2) / coeff[o + m + 1];
    o += m + 2;
    d *= eps;
  }
}

static void A3coeff(orc_geod *g) {
  static const double coeff[] = {
      -3, 128,          /* eps^5 */
      -2, -3, 64,       /* eps^4 */
      -1, -3, -1, 16,   /* eps^3 */
      3, -1, -2, 8,     /* eps^2 */
      1, -1, 2,         /* eps^1 */
      1, 1,             /* eps^0 */
  };
  int o = 0, k = 0, j;
  for (j = NA3 - 1; j >= 0; --j) {
    int m = NA3 - j - 1 < j ? NA3 - j - 1 : j;
    g->A3x[k++] = polyval(m, coeff + o, g->n) / coeff[o + m + 1];
    o += m + 2;
  }
}

static void C3coeff(orc_geod *g) {
  static const double coeff[] = {
      3, 128,          /* C3[1] eps^5 */
      2, 5, 128,       /* C3[1] eps^4 */
      -1, 3, 3, 64,    /* C3[1] eps^3 */
      -1, 0, 1, 8,     /* C3[1] eps^2 */
      -1, 1, 4,        /* C3[1] eps^1 */
      5, 256,          /* C3[2] eps^5 */
      1, 3, 128,       /* C3[2] eps^4 */
      -3, -2, 3, 64,   /* C3[2] eps^3 */
      1, -3, 2, 32,    /* C3[2] eps^2 */
      7, 512,          /* C3[3] eps^5 */
      -10, 9, 384,     /* C3[3] eps^4 */
      5, -9, 5, 192,   /* C3[3] eps^3 */
      7, 512,          /* C3[4] eps^5 */
      -14, 7, 512,     /* C3[4] eps^4 */
      21, 2560,        /* C3[5] eps^5 */
  };
  int o = 0, k = 0, l, j;
  for (l = 1; l < NC3; ++l) {
    for (j = NC3 - 1; j >= l; --j) {
      int m = NC3 - j - 1 < j ? NC3 - j - 1 : j;
      g->C3x[k++] = polyval(m, coeff + o, g->n) / coeff[o + m + 1];
      o += m + 2;
    }
  }
}

static double A3f(const orc_geod *g, double eps) {
  return polyval(NA3 - 1, g->A3x, eps);
}

static void C3f(const orc_geod *g, double eps, double c[]) {
  double mult = 1;
  int o = 0, l;
  for (l = 1; l < NC3; ++l) {
    int m = NC3 - l - 1;
    mult *= eps;
    c[l] = mult * polyval(m, g->C3x + o, eps);
    o += m + 1;
  }
}

void orc_geod_init(orc_geod *g, double a, double f) {
  g->a = a;
  g->f = f;
  g->f1 = 1 - f;
  g->e2 = f * (2 - f);
  g->ep2 = g->e2 / sq(g->f1);
  g->n = f / (2 - f);
  g->b = a * g->f1;
  A3coeff(g);
  C3coeff(g);
}

void orc_geod_direct(const orc_geod *g, double lat1, double lon1, double azi1,
                     double s12, double *plat2, double *plon2, double *pazi2) {
  double salp1, calp1, sbet1, cbet1, salp0, calp0, ssig1, csig1, somg1, comg1;
  double k2, eps, A1m1, B11, stau1, ctau1, A3c, B31, s, c;
  double C1a[NC1 + 1], C1pa[NC1 + 1], C3a[NC3];
  double tau12, B12, sig12, ssig12, csig12, ssig2, csig2, sbet2, cbet2;
  double salp2, calp2, somg2, comg2, omg12, lam12, lon12;

  azi1 = ang_normalize(azi1);
  sincosd(ang_round(azi1), &salp1, &calp1);
  lat1 = lat_fix(lat1);
  sincosd(ang_round(lat1), &sbet1, &cbet1);
  sbet1 *= g->f1;
  norm2(&sbet1, &cbet1);
  cbet1 = fmax(tiny, cbet1);

  salp0 = salp1 * cbet1;
  calp0 = hypot(calp1, salp1 * sbet1);
  ssig1 = sbet1;
  somg1 = salp0 * sbet1;
  csig1 = comg1 = (sbet1 != 0 || calp1 != 0) ? cbet1 * calp1 : 1;
  norm2(&ssig1, &csig1);

  k2 = sq(calp0) * g->ep2;
  eps = k2 / (2 * (1 + sqrt(1 + k2)) + k2);

  A1m1 = A1m1f(eps);
  C1f(eps, C1a);
  B11 = sin_series(ssig1, csig1, C1a, NC1);
  s = sin(B11);
  c = cos(B11);
  stau1 = ssig1 * c + csig1 * s;
  ctau1 = csig1 * c - ssig1 * s;
  C1pf(eps, C1pa);

  C3f(g, eps, C3a);
  A3c = -g->f * salp0 * A3f(g, eps);
  B31 = sin_series(ssig1, csig1, C3a, NC3 - 1);

  tau12 = s12 / (g->b * (1 + A1m1));
  s = sin(tau12);
  c = cos(tau12);
  B12 = -sin_series(stau1 * c + ctau1 * s, ctau1 * c - stau1 * s, C1pa, NC1);
  sig12 = tau12 - (B12 - B11);
  ssig12 = sin(sig12);
  csig12 = cos(sig12);
  if (fabs(g->f) > 0.01) {
    /* one Newton step for very eccentric ellipsoids (not WGS84) */
    double ssig2n = ssig1 * csig12 + csig1 * ssig12;
    double csig2n = csig1 * csig12 - ssig1 * ssig12;
    double serr;
    B12 = sin_series(ssig2n, csig2n, C1a, NC1);
    serr = (1 + A1m1) * (sig12 + (B12 - B11)) - s12 / g->b;
    sig12 = sig12 - serr / sqrt(1 + k2 * sq(ssig2n));
    ssig12 = sin(sig12);
    csig12 = cos(sig12);
  }
  ssig2 = ssig1 * csig12 + csig1 * ssig12;
  csig2 = csig1 * csig12 - ssig1 * ssig12;
  sbet2 = calp0 * ssig2;
  cbet2 = hypot(salp0, calp0 * csig2);
  if (cbet2 == 0) cbet2 = csig2 = tiny;
  salp2 = salp0;
  calp2 = calp0 * csig2;

  somg2 = salp0 * ssig2;
  comg2 = csig2;
  omg12 = atan2(somg2 * comg1 - comg2 * somg1, comg2 * comg1 + somg2 * somg1);
  lam12 = omg12 +
          A3c * (sig12 + (sin_series(ssig2, csig2, C3a, NC3 - 1) - B31));
  lon12 = lam12 / degree;
  if (plon2)
    *plon2 = ang_normalize(ang_normalize(lon1) + ang_normalize(lon12));
  if (plat2) *plat2 = atan2d(sbet2, g->f1 * cbet2);
  if (pazi2) *pazi2 = atan2d(salp2, calp2);
}

/* Inverse by shooting: iterate the local (north,east) launch vector until the
 * direct solution lands on point 2.  Converges quadratically-ish for lines
 * short against the ellipsoid radius (rotate_vectors uses 10 m / 0.1 deg). */
void orc_geod_inverse(const orc_geod *g, double lat1, double lon1, double lat2,
                      double lon2, double *pazi1, double *ps12) {
  double sphi, cphi, w2, M, N, sn, se, best = INFINITY;
  double dlon = ang_normalize(lon2 - lon1);
  int it;
  sincosd(0.5 * (lat1 + lat2), &sphi, &cphi);
  w2 = 1 - g->e2 * sphi * sphi;
  M = g->a * (1 - g->e2) / (w2 * sqrt(w2));
  N = g->a / sqrt(w2);
  sn = (lat2 - lat1) * degree * M;
  se = dlon * degree * N * cphi;
  for (it = 0; it < 60; ++it) {
    double azi = atan2d(se, sn), s = hypot(sn, se);
    double la, lo, az2, mn, me, sp2, cp2, w22, M2, N2, sr, cr, miss;
    orc_geod_direct(g, lat1, lon1, azi, s, &la, &lo, &az2);
    sincosd(lat2, &sp2, &cp2);
    w22 = 1 - g->e2 * sp2 * sp2;
    M2 = g->a * (1 - g->e2) / (w22 * sqrt(w22));
    N2 = g->a / sqrt(w22);
    mn = (lat2 - la) * degree * M2;
    me = ang_normalize(lon2 - lo) * degree * N2 * cp2;
    miss = hypot(mn, me);
    if (pazi1) *pazi1 = azi;
    if (ps12) *ps12 = s;
    if (miss == 0 || miss >= best) break;
    best = miss;
    /* rotate the miss vector back by the meridian convergence along the line */
    sincosd(az2 - azi, &sr, &cr);
    sn += cr * mn + sr * me;
    se += -sr * mn + cr * me;
  }
}

static orc_geod wgs84;
static int wgs84_ready = 0;
static const orc_geod *get_wgs84(void) {
  if (!wgs84_ready) {
    orc_geod_init(&wgs84, 6378137.0, 1 / 298.257223563);
    wgs84_ready = 1;
  }
  return &wgs84;
}

void orc_wgs84_direct_n(long n, const double *lon1, const double *lat1,
                        const double *azi1, const double *s12, double *lon2,
                        double *lat2, double *azi2) {
  const orc_geod *g = get_wgs84();
  long i;
  for (i = 0; i < n; ++i) {
    double la, lo, az;
    orc_geod_direct(g, lat1[i], lon1[i], azi1[i], s12[i], &la, &lo, &az);
    lon2[i] = lo;
    lat2[i] = la;
    if (azi2) azi2[i] = az;
  }
}

void orc_wgs84_inverse_n(long n, const double *lon1, const double *lat1,
                         const double *lon2, const double *lat2, double *azi1,
                         double *azi2_unused, double *s12) {
  const orc_geod *g = get_wgs84();
  long i;
  (void)azi2_unused;
  for (i = 0; i < n; ++i) {
    double az, s;
    orc_geod_inverse(g, lat1[i], lon1[i], lat2[i], lon2[i], &az, &s);
    if (azi1) azi1[i] = az;
    if (s12) s12[i] = s;
  }
}
