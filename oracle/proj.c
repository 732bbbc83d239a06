/* TEST INFRASTRUCTURE ONLY -- CPU oracle, never on the product path.
 *
 * Map projections the hot path needs, restated from the published formulas
 * (Snyder, "Map Projections -- A Working Manual", USGS PP 1395, ch. 21
 * Stereographic: eqs. 21-2..21-4 sphere, 21-33..21-40 / 15-9 / 7-9 polar
 * ellipsoid), which is what the un-vendored pyproj.Proj evaluates for the
 * reference at
 *   opendrift/readers/basereader/variables.py:111-143 (xy2lonlat / lonlat2xy)
 *   opendrift/readers/reader_double_gyre.py:28-33      (+proj=stere sphere, equatorial)
 *   tests/readers/test_rotate_proj.py:17-19            (NorKyst polar stere on an ellipsoid)
 */
#include "oracle.h"
#include <math.h>

static const double DEG = 3.14159265358979323846264338327950288 / 180.0;
static const double HALFPI = 1.57079632679489661923;

static double tsfn(double phi, double sinphi, double e) {
  /* Snyder 15-9: t = tan(pi/4 - phi/2) / ((1 - e sin phi)/(1 + e sin phi))^(e/2) */
  double es = e * sinphi;
  return tan(0.5 * (HALFPI - phi)) / pow((1 - es) / (1 + es), 0.5 * e);
}

void orc_proj_init(orc_proj *p, int kind, double a, double es, double lat0_deg,
                   double lon0_deg, double lat_ts_deg, double k0, double x0,
                   double y0) {
  p->kind = kind;
  p->a = a;
  p->es = es;
  p->e = sqrt(es);
  p->lon0 = lon0_deg * DEG;
  p->lat0 = lat0_deg * DEG;
  p->x0 = x0;
  p->y0 = y0;
  p->k0 = k0;
  p->south = lat0_deg < 0;
  p->akm1 = 2 * k0;
  if (kind == ORC_PROJ_STERE_POLAR) {
    double phits = fabs(lat_ts_deg) * DEG;
    if (es == 0) {
      p->akm1 = fabs(phits - HALFPI) >= 1e-10 ? cos(phits) / tan(0.5 * (HALFPI - phits))
                                               : 2 * k0;
    } else if (fabs(phits - HALFPI) < 1e-10) {
      p->akm1 = 2 * k0 / sqrt(pow(1 + p->e, 1 + p->e) * pow(1 - p->e, 1 - p->e));
    } else {
      double t = sin(phits);
      p->akm1 = cos(phits) / tsfn(phits, t, p->e);
      t *= p->e;
      p->akm1 /= sqrt(1 - t * t);
    }
  }
}

static double wrap_pi(double lam) {
  /* PROJ adjlon: reduce to [-pi, pi] */
  if (fabs(lam) <= 3.14159265358979323846 + 1e-12) return lam;
  lam += 3.14159265358979323846;
  lam -= 2 * 3.14159265358979323846 * floor(lam / (2 * 3.14159265358979323846));
  lam -= 3.14159265358979323846;
  return lam;
}

void orc_proj_fwd(const orc_proj *p, double lon_deg, double lat_deg, double *x,
                  double *y) {
  if (p->kind == ORC_PROJ_LATLONG) { *x = lon_deg; *y = lat_deg; return; }
  {
    double lam = wrap_pi(lon_deg * DEG - p->lon0), phi = lat_deg * DEG;
    double sinlam = sin(lam), coslam = cos(lam), sinphi = sin(phi), cosphi = cos(phi);
    double X, Y;
    if (p->kind == ORC_PROJ_STERE_EQUIT_SPHERE) {
      /* Snyder 21-2..21-4 with phi1 = 0 */
      double d = 1 + cosphi * coslam;
      double k = p->akm1 / d;
      X = k * cosphi * sinlam;
      Y = k * sinphi;
    } else { /* polar */
      double rho;
      if (p->south) { phi = -phi; coslam = -coslam; sinphi = -sinphi; }
      if (p->es == 0)
        rho = p->akm1 * tan(0.5 * (HALFPI - phi)); /* pi/4 - phi/2 */
      else
        rho = fabs(phi - HALFPI) < 1e-15 ? 0 : p->akm1 * tsfn(phi, sinphi, p->e);
      X = rho * sinlam;
      Y = -rho * coslam;
    }
    *x = p->a * X + p->x0;
    *y = p->a * Y + p->y0;
  }
}

void orc_proj_inv(const orc_proj *p, double x, double y, double *lon_deg,
                  double *lat_deg) {
  if (p->kind == ORC_PROJ_LATLONG) { *lon_deg = x; *lat_deg = y; return; }
  {
    double X = (x - p->x0) / p->a, Y = (y - p->y0) / p->a;
    double rh = hypot(X, Y), lam = 0, phi = 0;
    if (p->kind == ORC_PROJ_STERE_EQUIT_SPHERE) {
      double c = 2 * atan(rh / p->akm1), sinc = sin(c), cosc = cos(c);
      if (fabs(rh) <= 1e-10) phi = 0; else phi = asin(Y * sinc / rh);
      if (cosc != 0 || X != 0) lam = atan2(X * sinc, cosc * rh);
    } else if (p->es == 0) {
      double c = 2 * atan(rh / p->akm1), cosc = cos(c);
      if (!p->south) Y = -Y;
      phi = fabs(rh) <= 1e-10 ? p->lat0 : asin(p->south ? -cosc : cosc);
      lam = (X == 0 && Y == 0) ? 0 : atan2(X, Y);
    } else {
      /* Snyder 21-39 / 7-9: iterate the conformal latitude inverse to full
       * f64 convergence (PROJ stops at 1e-10 rad; the fixed point is the same). */
      double tp = rh / p->akm1, phi_l = HALFPI - 2 * atan(tp), halfe = 0.5 * p->e;
      int i;
      if (!p->south) Y = -Y;
      for (i = 0; i < 16; ++i) {
        double es = p->e * sin(phi_l);
        phi = HALFPI - 2 * atan(tp * pow((1 - es) / (1 + es), halfe));
        if (fabs(phi - phi_l) < 1e-15) break;
        phi_l = phi;
      }
      if (p->south) phi = -phi;
      lam = (X == 0 && Y == 0) ? 0 : atan2(X, Y);
    }
    *lon_deg = wrap_pi(lam + p->lon0) / DEG;
    *lat_deg = phi / DEG;
  }
}
