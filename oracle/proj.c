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
 * and ch. 7 Mercator (7-6..7-12) and ch. 15 Lambert conformal conic (15-1..15-11, 14-1..14-11 for the sphere) for readers
 * whose proj4 names them (reader_netCDF_CF_generic.py reads any grid mapping; MEPS / NORA3 / AROME are lcc).  Pinned on
 * Snyder's numerical examples (pp. 266-267, 295-297; tests/test_oracle_golden.py): PROJ itself is not in this image.
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

static double msfn(double phi, double es) { double sp = sin(phi); return cos(phi) / sqrt(1 - es * sp * sp); }

void orc_proj_init_conic(orc_proj *p, int kind, double a, double es, double lat0_deg, double lon0_deg, double lat_ts_deg,
                         double k0, double x0, double y0, double lat1_deg, double lat2_deg) {
  orc_proj_init(p, kind, a, es, lat0_deg, lon0_deg, lat_ts_deg, k0, x0, y0);
  p->n = p->c = p->rho0 = 0;
  if (kind == ORC_PROJ_MERC) {
    if (lat_ts_deg != 0) p->k0 = msfn(fabs(lat_ts_deg) * DEG, es);   /* Snyder 7-8 scale at the true-scale latitude */
  } else if (kind == ORC_PROJ_LCC) {
    double phi1 = lat1_deg * DEG, phi2 = lat2_deg * DEG, n = sin(phi1);
    double m1 = msfn(phi1, es), t1 = tsfn(phi1, sin(phi1), p->e);
    if (fabs(phi1 - phi2) >= 1e-10) n = log(m1 / msfn(phi2, es)) / log(t1 / tsfn(phi2, sin(phi2), p->e));  /* 15-8 */
    p->n = n;
    p->c = m1 * pow(t1, -n) / n;                                                                         /* 15-10 */
    p->rho0 = fabs(fabs(p->lat0) - HALFPI) < 1e-10 ? 0 : p->c * pow(tsfn(p->lat0, sin(p->lat0), p->e), n); /* 15-7a */
  }
}

static double phi2(double ts, double e) {
  /* Snyder 7-9 iterated to float64 convergence */
  double phi_l = HALFPI - 2 * atan(ts), phi = phi_l;
  int i;
  for (i = 0; i < 16 && e != 0; ++i) {
    double es = e * sin(phi_l);
    phi = HALFPI - 2 * atan(ts * pow((1 - es) / (1 + es), 0.5 * e));
    if (fabs(phi - phi_l) < 1e-15) break;
    phi_l = phi;
  }
  return phi;
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
    if (p->kind == ORC_PROJ_MERC) {          /* Snyder 7-6, 7-7 */
      *x = p->a * (p->k0 * lam) + p->x0;
      *y = p->a * (-p->k0 * log(tsfn(phi, sinphi, p->e))) + p->y0;
      return;
    }
    if (p->kind == ORC_PROJ_LCC) {           /* Snyder 15-7, 14-4, 14-1, 14-2 */
      double rho = fabs(fabs(phi) - HALFPI) < 1e-10 ? 0 : p->c * pow(tsfn(phi, sinphi, p->e), p->n);
      *x = p->a * (p->k0 * (rho * sin(p->n * lam))) + p->x0;
      *y = p->a * (p->k0 * (p->rho0 - rho * cos(p->n * lam))) + p->y0;
      return;
    }
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
    if (p->kind == ORC_PROJ_MERC) {          /* Snyder 7-10, 7-12 */
      *lon_deg = wrap_pi(X / p->k0 + p->lon0) / DEG;
      *lat_deg = phi2(exp(-Y / p->k0), p->e) / DEG;
      return;
    }
    if (p->kind == ORC_PROJ_LCC) {           /* Snyder 14-10, 14-11, 15-11, 14-9 */
      double xx = X / p->k0, yy = p->rho0 - Y / p->k0, rho = hypot(xx, yy);
      if (p->n < 0) { rho = -rho; xx = -xx; yy = -yy; }
      if (rho == 0) { *lon_deg = wrap_pi(p->lon0) / DEG; *lat_deg = p->n > 0 ? 90 : -90; return; }
      *lon_deg = wrap_pi(atan2(xx, yy) / p->n + p->lon0) / DEG;
      *lat_deg = phi2(pow(rho / p->c, 1 / p->n), p->e) / DEG;
      return;
    }
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
