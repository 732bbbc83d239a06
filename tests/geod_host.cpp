// TEST INFRASTRUCTURE: the float64 routines of opendrift_amd/csrc/odr_geodesic.hip.h (series geodesic, its start-point
// coefficients and their chained form, the range-limited sine / cosine / logarithm / exponential, arctan2 on finite arguments)
// compiled for the CPU (g++ -ffp-contract=off, tests/hostshim/hip/hip_runtime.h) and exported for tests/test_geod_host.py.
#include "../opendrift_amd/csrc/odr_geodesic.hip.h"

using namespace odr;

extern "C" {
void gh_sincos_pi(long long n, const double *x, double *s, double *c) { for (long long i = 0; i < n; ++i) sincos_pi(x[i], s[i], c[i]); }
void gh_sincosd(long long n, const double *x, double *s, double *c) { for (long long i = 0; i < n; ++i) sincosd(x[i], s[i], c[i]); }
void gh_log_pos(long long n, const double *x, double *y) { for (long long i = 0; i < n; ++i) y[i] = log_pos(x[i]); }
void gh_exp_small(long long n, const double *x, double *y) { for (long long i = 0; i < n; ++i) y[i] = exp_small(x[i]); }
void gh_atan2_fin(long long n, const double *y, const double *x, double *a) { for (long long i = 0; i < n; ++i) a[i] = atan2_fin(y[i], x[i]); }
// one series move from (lat, lon) by (east, north) metres; full = 1: the complete solution instead (geod_local_far)
void gh_move(long long n, const double *lat, const double *lon, const double *x, const double *y, int full, double *lat2, double *lon2,
             int *series) {
  for (long long i = 0; i < n; ++i) {
    if (full) {
      const GeodLL r = geod_local_far(lat[i], ang_normalize(lon[i]), x[i], y[i]);
      lat2[i] = r.lat; lon2[i] = r.lon; series[i] = 0;
    } else {
      const GeodLocal L = geod_local_origin(lat[i], lon[i]);
      series[i] = geod_local_move_ok(L, x[i], y[i], lat2[i], lon2[i]) ? 1 : 0;
    }
  }
}
// two moves one after the other; chained = 1: the second start point from the first one's sine / cosine (geod_local_origin_next)
void gh_two_moves(long long n, const double *lat, const double *lon, const double *x1, const double *y1, const double *x2,
                  const double *y2, int chained, double *lat3, double *lon3) {
  for (long long i = 0; i < n; ++i) {
    double sp, cp, la, lo;
    const GeodLocal L = geod_local_origin_sc(lat[i], lon[i], sp, cp);
    const bool ok = geod_local_move_ok(L, x1[i], y1[i], la, lo);
    const GeodLocal M = (chained && ok) ? geod_local_origin_next(lat[i], la, lo, sp, cp) : geod_local_origin(la, lo);
    geod_local_move(M, x2[i], y2[i], lat3[i], lon3[i]);
  }
}
}
