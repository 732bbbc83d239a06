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

void orc_proj_fwd_ext(const orc_proj *p, double lon_deg, double lat_deg, double *x, double *y);
void orc_proj_inv_ext(const orc_proj *p, double x, double y, double *lon_deg, double *lat_deg);

void orc_proj_fwd(const orc_proj *p, double lon_deg, double lat_deg, double *x,
                  double *y) {
  if (p->kind == ORC_PROJ_LATLONG) { *x = lon_deg; *y = lat_deg; return; }
  if (p->kind >= ORC_PROJ_TMERC) { orc_proj_fwd_ext(p, lon_deg, lat_deg, x, y); return; }
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
  if (p->kind >= ORC_PROJ_TMERC) { orc_proj_inv_ext(p, x, y, lon_deg, lat_deg); return; }
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

/* ==================================================================================================================
 * Round 5: the projections real model files use besides stere / merc / lcc -- what pyproj.Proj evaluates for the
 * reference at opendrift/readers/basereader/variables.py:111-143 when a reader's proj4 names them:
 *   +proj=tmerc / +proj=utm   Krueger's series in the third flattening n to order 6 (Karney, "Transverse Mercator with an
 *                             accuracy of a few nanometers", J. Geodesy 85 (2011), eqs. 7-11, 35, 36 with the conformal
 *                             latitude in closed form, eqs. 7-9, and its Newton inverse, eqs. 19-21); e = 0 reduces it
 *                             to Snyder's spherical formulas 8-1..8-8.  PROJ's default tmerc (Poder / Engsager) is the
 *                             same series to the same order.
 *   +proj=laea                Snyder ch. 24: 24-2..24-4 / 24-13..24-16 (sphere), 24-17..24-26 with the authalic latitude
 *                             3-11, 3-12 (ellipsoid); the inverse authalic latitude by Newton iteration on 3-12 (3-16).
 *   +proj=stere, oblique and equatorial aspects: Snyder 21-2..21-4, 21-14, 21-15 (sphere), 21-24..21-27 and 21-36..21-38 with the
 *                             conformal latitude 3-1 and its fixed-point inverse 7-9 (ellipsoid).
 *   +proj=ob_tran +o_proj=longlat  the rotated pole of HIRLAM / AROME / CMEMS-Arctic files: Snyder 5-7..5-10b
 *                             (PROJ's o_forward / o_inverse with the pole at o_lat_p, o_lon_p), no ellipsoid involved;
 *                             x, y in degrees, as Variables.lonlat2xy returns them for such a reader (:136-138).
 * Pinned on Snyder's numerical examples (Appendix A) and on independent properties (meridian arc, conformality, equal
 * area) in tests/test_oracle_golden.py.  PROJ itself is not in this image.
 * ================================================================================================================== */
static const double PI_ = 3.14159265358979323846;

static double qsfn(double sinphi, double e, double one_es) {   /* Snyder 3-12 */
  if (e < 1e-7) return 2 * sinphi;
  {
    double con = e * sinphi, d1 = 1 - con * con;
    return one_es * (sinphi / d1 - (0.5 / e) * log((1 - con) / (1 + con)));
  }
}

/* conformal latitude as its tangent (Karney 2011, eqs. 7-9): tau' = tau sqrt(1 + sigma^2) - sigma sqrt(1 + tau^2) */
static double taup_of(double tau, double e) {
  double tau1 = hypot(1.0, tau), sig = sinh(e * atanh(e * tau / tau1));
  return hypot(1.0, sig) * tau - sig * tau1;
}
static double tau_of(double taup, double e) {                    /* eqs. 19-21, Newton */
  double e2m = 1 - e * e, tau = taup / e2m, stol = 1e-15 * fmax(1.0, fabs(taup));
  int i;
  if (!(fabs(taup) < 1e300)) return taup;
  for (i = 0; i < 8; ++i) {
    double tp = taup_of(tau, e);
    double dtau = (taup - tp) * (1 + e2m * tau * tau) / (e2m * hypot(1.0, tau) * hypot(1.0, tp));
    tau += dtau;
    if (!(fabs(dtau) >= stol)) break;
  }
  return tau;
}

void orc_proj_init_ext(orc_proj *p, int kind, double a, double es, double lat0_deg, double lon0_deg, double k0, double x0,
                       double y0, double lat1_deg, double lat2_deg) {
  int k;
  orc_proj_init(p, kind, a, es, lat0_deg, lon0_deg, 90.0, k0, x0, y0);
  p->n = p->c = p->rho0 = 0;
  p->mode = 0; p->pad = 0;
  for (k = 0; k < 16; ++k) p->q[k] = 0;
  if (kind == ORC_PROJ_TMERC) {
    double f = 1 - sqrt(1 - es), n = f / (2 - f), n2 = n * n, n3 = n2 * n, n4 = n2 * n2, n5 = n4 * n, n6 = n3 * n3;
    double *al = p->q + 2, *be = p->q + 8, xi0;
    p->q[0] = k0 / (1 + n) * (1 + n2 * (1.0 / 4 + n2 * (1.0 / 64 + n2 / 256)));            /* A / a (eq. 14) times k0 */
    al[0] = n / 2 - 2 * n2 / 3 + 5 * n3 / 16 + 41 * n4 / 180 - 127 * n5 / 288 + 7891 * n6 / 37800;      /* eq. 35 */
    al[1] = 13 * n2 / 48 - 3 * n3 / 5 + 557 * n4 / 1440 + 281 * n5 / 630 - 1983433 * n6 / 1935360;
    al[2] = 61 * n3 / 240 - 103 * n4 / 140 + 15061 * n5 / 26880 + 167603 * n6 / 181440;
    al[3] = 49561 * n4 / 161280 - 179 * n5 / 168 + 6601661 * n6 / 7257600;
    al[4] = 34729 * n5 / 80640 - 3418889 * n6 / 1995840;
    al[5] = 212378941 * n6 / 319334400;
    be[0] = n / 2 - 2 * n2 / 3 + 37 * n3 / 96 - n4 / 360 - 81 * n5 / 512 + 96199 * n6 / 604800;         /* eq. 36 */
    be[1] = n2 / 48 + n3 / 15 - 437 * n4 / 1440 + 46 * n5 / 105 - 1118711 * n6 / 3870720;
    be[2] = 17 * n3 / 480 - 37 * n4 / 840 - 209 * n5 / 4480 + 5569 * n6 / 90720;
    be[3] = 4397 * n4 / 161280 - 11 * n5 / 504 - 830251 * n6 / 7257600;
    be[4] = 4583 * n5 / 161280 - 108847 * n6 / 3991680;
    be[5] = 20648693 * n6 / 638668800;
    /* xi of the origin latitude on the central meridian: the false northing refers to it */
    xi0 = atan(taup_of(tan(p->lat0), p->e));
    { double s = xi0; for (k = 0; k < 6; ++k) s += al[k] * sin(2 * (k + 1) * xi0); xi0 = s; }
    p->q[1] = xi0;
  } else if (kind == ORC_PROJ_LAEA) {
    double t = fabs(p->lat0), one_es = 1 - es;
    p->mode = fabs(t - HALFPI) < 1e-10 ? (p->lat0 < 0 ? 1 : 0) : (t < 1e-10 ? 2 : 3);
    p->q[7] = sin(p->lat0); p->q[8] = cos(p->lat0);
    if (es != 0) {
      double qp = qsfn(1.0, p->e, one_es), rq = sqrt(0.5 * qp);
      p->q[0] = qp; p->q[1] = rq; p->q[2] = 1; p->q[3] = 1; p->q[4] = 1;
      if (p->mode == 2) { p->q[2] = 1 / rq; p->q[3] = 1; p->q[4] = 0.5 * qp; }
      if (p->mode == 3) {
        double sinphi = sin(p->lat0), sinb1 = qsfn(sinphi, p->e, one_es) / qp, cosb1 = sqrt(1 - sinb1 * sinb1);
        double dd = cos(p->lat0) / (sqrt(1 - es * sinphi * sinphi) * rq * cosb1);     /* Snyder 24-20 */
        p->q[5] = sinb1; p->q[6] = cosb1; p->q[2] = dd; p->q[4] = rq / dd; p->q[3] = rq * dd;
      }
    }
  } else if (kind == ORC_PROJ_STERE_OBLIQUE) {
    double t = fabs(p->lat0);
    p->mode = t > 1e-10 ? 3 : 2;
    if (es != 0) {
      double sp = sin(p->lat0), X = 2 * atan(tan(0.5 * (HALFPI + p->lat0)) * pow((1 - sp * p->e) / (1 + sp * p->e), 0.5 * p->e)) - HALFPI;
      p->akm1 = 2 * k0 * cos(p->lat0) / sqrt(1 - es * sp * sp);                        /* Snyder 21-27 with 14-15 */
      p->q[0] = sin(X); p->q[1] = cos(X);
    } else {
      p->akm1 = 2 * k0;
      p->q[0] = sin(p->lat0); p->q[1] = cos(p->lat0);
    }
  } else if (kind == ORC_PROJ_OB_TRAN) {
    p->q[0] = sin(lat1_deg * DEG); p->q[1] = cos(lat1_deg * DEG); p->q[2] = lat2_deg * DEG;
    p->a = 1; p->es = 0; p->e = 0; p->x0 = p->y0 = 0; p->k0 = 1;
  }
}

static void ext_fwd(const orc_proj *p, double lam, double phi, double *X, double *Y) {
  double sinlam = sin(lam), coslam = cos(lam), sinphi = sin(phi), cosphi = cos(phi);
  if (p->kind == ORC_PROJ_TMERC) {
    double tau = sinphi / cosphi, taup = taup_of(tau, p->e);
    double xip = atan2(taup, coslam), etap = asinh(sinlam / hypot(taup, coslam)), xi = xip, eta = etap;
    int k;
    if (fabs(phi) >= HALFPI) { xip = phi > 0 ? HALFPI : -HALFPI; etap = 0; xi = xip; eta = 0; }
    for (k = 0; k < 6; ++k) {
      xi += p->q[2 + k] * sin(2 * (k + 1) * xip) * cosh(2 * (k + 1) * etap);
      eta += p->q[2 + k] * cos(2 * (k + 1) * xip) * sinh(2 * (k + 1) * etap);
    }
    *X = p->q[0] * eta;
    *Y = p->q[0] * (xi - p->q[1]);
  } else if (p->kind == ORC_PROJ_LAEA) {
    if (p->es != 0) {
      double q = qsfn(sinphi, p->e, 1 - p->es), qp = p->q[0], b;
      if (p->mode >= 2) {
        double sinb = q / qp, cosb = sqrt(1 - sinb * sinb);
        if (p->mode == 3) {
          b = sqrt(2 / (1 + p->q[5] * sinb + p->q[6] * cosb * coslam));
          *Y = p->q[4] * b * (p->q[6] * sinb - p->q[5] * cosb * coslam);
        } else {
          b = sqrt(2 / (1 + cosb * coslam));
          *Y = b * sinb * p->q[4];
        }
        *X = p->q[3] * b * cosb * sinlam;
      } else {
        q = p->mode == 0 ? qp - q : qp + q;
        b = q >= 1e-30 ? sqrt(q) : 0;
        *X = b * sinlam;
        *Y = coslam * (p->mode == 1 ? b : -b);
      }
    } else {
      double sp0 = p->q[7], cp0 = p->q[8], k;
      if (p->mode >= 2) {
        k = sqrt(2 / (1 + sp0 * sinphi + cp0 * cosphi * coslam));
        *X = k * cosphi * sinlam;
        *Y = k * (cp0 * sinphi - sp0 * cosphi * coslam);
      } else {
        double h = 0.25 * PI_ - 0.5 * phi;
        k = 2 * (p->mode == 1 ? cos(h) : sin(h));
        *X = k * sinlam;
        *Y = k * (p->mode == 0 ? -coslam : coslam);
      }
    }
  } else if (p->kind == ORC_PROJ_STERE_OBLIQUE) {
    if (p->es != 0) {
      double Xc = 2 * atan(tan(0.5 * (HALFPI + phi)) * pow((1 - sinphi * p->e) / (1 + sinphi * p->e), 0.5 * p->e)) - HALFPI;
      double sinX = sin(Xc), cosX = cos(Xc), A;
      if (p->mode == 3) {
        A = p->akm1 / (p->q[1] * (1 + p->q[0] * sinX + p->q[1] * cosX * coslam));
        *Y = A * (p->q[1] * sinX - p->q[0] * cosX * coslam);
      } else {
        A = p->akm1 / (1 + cosX * coslam);
        *Y = A * sinX;
      }
      *X = A * cosX * sinlam;
    } else {
      double A = p->akm1 / (1 + p->q[0] * sinphi + p->q[1] * cosphi * coslam);
      *X = A * cosphi * sinlam;
      *Y = A * (p->q[1] * sinphi - p->q[0] * cosphi * coslam);
    }
  } else { /* ob_tran, o_proj = longlat: PROJ's o_forward; rotated longitude / latitude in radians */
    double sphip = p->q[0], cphip = p->q[1];
    *X = wrap_pi(atan2(cosphi * sinlam, sphip * cosphi * coslam + cphip * sinphi) + p->q[2]);
    { double s = sphip * sinphi - cphip * cosphi * coslam; *Y = asin(s > 1 ? 1 : (s < -1 ? -1 : s)); }
  }
}

static void ext_inv(const orc_proj *p, double X, double Y, double *lam, double *phi) {
  if (p->kind == ORC_PROJ_TMERC) {
    double xi = Y / p->q[0] + p->q[1], eta = X / p->q[0], xip = xi, etap = eta, s, c, taup;
    int k;
    for (k = 0; k < 6; ++k) {
      xip -= p->q[8 + k] * sin(2 * (k + 1) * xi) * cosh(2 * (k + 1) * eta);
      etap -= p->q[8 + k] * cos(2 * (k + 1) * xi) * sinh(2 * (k + 1) * eta);
    }
    s = sinh(etap); c = cos(xip);
    *lam = atan2(s, c);
    taup = sin(xip) / hypot(s, c);
    *phi = atan(tau_of(taup, p->e));
  } else if (p->kind == ORC_PROJ_LAEA) {
    double x = X, y = Y, ab;
    if (p->es != 0) {
      double qp = p->q[0], rq = p->q[1], beta, ph;
      int i;
      if (p->mode >= 2) {
        double rho, sCe, cCe;
        x /= p->q[2]; y *= p->q[2];
        rho = hypot(x, y);
        if (rho < 1e-10) { *lam = 0; *phi = p->lat0; return; }
        sCe = 2 * asin(0.5 * rho / rq); cCe = cos(sCe); sCe = sin(sCe);
        x *= sCe;
        if (p->mode == 3) { ab = cCe * p->q[5] + y * sCe * p->q[6] / rho; y = rho * p->q[6] * cCe - y * p->q[5] * sCe; }
        else { ab = y * sCe / rho; y = rho * cCe; }
      } else {
        double q;
        if (p->mode == 0) y = -y;
        q = x * x + y * y;
        if (q == 0) { *lam = 0; *phi = p->lat0; return; }
        ab = 1 - q / qp;
        if (p->mode == 1) ab = -ab;
      }
      *lam = atan2(x, y);
      /* geodetic from authalic latitude: Newton on q(phi) = qp sin(beta) (Snyder 3-16) */
      beta = asin(ab > 1 ? 1 : (ab < -1 ? -1 : ab));
      ph = beta;
      for (i = 0; i < 12; ++i) {
        double sp = sin(ph), cp = cos(ph), w = 1 - p->es * sp * sp;
        double d = w * w / (2 * cp) * (qp * ab / (1 - p->es) - sp / w + (0.5 / p->e) * log((1 - p->e * sp) / (1 + p->e * sp)));
        if (!(fabs(cp) > 1e-12)) break;
        ph += d;
        if (fabs(d) < 1e-15) break;
      }
      *phi = ph;
    } else {
      double rh = hypot(x, y), c = 2 * asin(fmin(1.0, 0.5 * rh)), sinz = sin(c), cosz = cos(c), ph;
      if (p->mode >= 2) {
        ph = rh <= 1e-10 ? p->lat0 : asin(cosz * p->q[7] + y * sinz * p->q[8] / rh);
        x *= sinz * p->q[8];
        y = (cosz - sin(ph) * p->q[7]) * rh;
        *lam = (y == 0 && x == 0) ? 0 : atan2(x, y);
      } else {
        if (p->mode == 0) { y = -y; ph = HALFPI - c; } else ph = c - HALFPI;
        *lam = atan2(x, y);
      }
      *phi = ph;
    }
  } else if (p->kind == ORC_PROJ_STERE_OBLIQUE) {
    double x = X, y = Y, rho = hypot(x, y);
    if (p->es != 0) {
      double tp = 2 * atan2(rho * p->q[1], p->akm1), cosphi = cos(tp), sinphi = sin(tp), phi_l, ph = 0;
      int i;
      phi_l = rho == 0 ? asin(cosphi * p->q[0]) : asin(cosphi * p->q[0] + y * sinphi * p->q[1] / rho);
      tp = tan(0.5 * (HALFPI + phi_l));
      x *= sinphi;
      y = rho * p->q[1] * cosphi - y * p->q[0] * sinphi;
      for (i = 0; i < 16; ++i) {     /* Snyder 3-4 / 7-9 to float64 convergence (PROJ stops at 1e-10) */
        double es = p->e * sin(phi_l);
        ph = 2 * atan(tp * pow((1 + es) / (1 - es), 0.5 * p->e)) - HALFPI;
        if (fabs(ph - phi_l) < 1e-15) break;
        phi_l = ph;
      }
      *phi = ph;
      *lam = (x == 0 && y == 0) ? 0 : atan2(x, y);
    } else {
      double c = 2 * atan(rho / p->akm1), sinc = sin(c), cosc = cos(c), ph, cc;
      ph = rho <= 1e-10 ? p->lat0 : asin(cosc * p->q[0] + y * sinc * p->q[1] / rho);
      cc = cosc - p->q[0] * sin(ph);
      *lam = (cc != 0 || x != 0) ? atan2(x * sinc * p->q[1], cc * rho) : 0;
      *phi = ph;
    }
  } else { /* ob_tran: PROJ's o_inverse */
    double sphip = p->q[0], cphip = p->q[1], l = X - p->q[2], coslam = cos(l), sinphi = sin(Y), cosphi = cos(Y);
    double s = sphip * sinphi + cphip * cosphi * coslam;
    *phi = asin(s > 1 ? 1 : (s < -1 ? -1 : s));
    *lam = atan2(cosphi * sin(l), sphip * cosphi * coslam - cphip * sinphi);
  }
}

/* front ends: orc_proj_fwd / orc_proj_inv hand the round-5 kinds over to these */
void orc_proj_fwd_ext(const orc_proj *p, double lon_deg, double lat_deg, double *x, double *y) {
  double X, Y;
  ext_fwd(p, wrap_pi(lon_deg * DEG - p->lon0), lat_deg * DEG, &X, &Y);
  if (p->kind == ORC_PROJ_OB_TRAN) { *x = X / DEG; *y = Y / DEG; return; }   /* np.degrees(self.proj(lon, lat)), variables.py:136-138 */
  *x = p->a * X + p->x0;
  *y = p->a * Y + p->y0;
}
void orc_proj_inv_ext(const orc_proj *p, double x, double y, double *lon_deg, double *lat_deg) {
  double lam, phi;
  if (p->kind == ORC_PROJ_OB_TRAN) ext_inv(p, x * DEG, y * DEG, &lam, &phi);   /* self.proj(np.radians(x), np.radians(y), inverse=True), :117-123 */
  else ext_inv(p, (x - p->x0) / p->a, (y - p->y0) / p->a, &lam, &phi);
  *lon_deg = wrap_pi(lam + p->lon0) / DEG;
  *lat_deg = phi / DEG;
}
