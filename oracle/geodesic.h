/* TEST INFRASTRUCTURE ONLY -- CPU oracle, never on the product path.
 *
 * Restatement of the geodesic arithmetic OpenDrift obtains from the un-vendored
 * dependency pyproj (>=2.3, PROJ < 9.8, pyproject.toml:18-19): PROJ's
 * geodesic.c = C. F. F. Karney, "Algorithms for geodesics", J. Geodesy 87
 * (2013) 43-55, series order 6.  Call sites this stands in for:
 *   opendrift/models/basemodel/__init__.py:4643-4657  (update_positions)
 *   opendrift/models/physics_methods.py:632-666        (RK2/RK4 sub-stages)
 *   opendrift/readers/basereader/variables.py:95-97    (rotate_vectors, Geod.inv)
 */
#ifndef ODR_ORACLE_GEODESIC_H
#define ODR_ORACLE_GEODESIC_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  double a, f, f1, e2, ep2, n, b;
  double A3x[6];
  double C3x[15];
} orc_geod;

void orc_geod_init(orc_geod *g, double a, double f);

/* Direct problem, degrees / metres in and out; lon2 normalised to [-180,180]
 * exactly as PROJ geod_direct (which is what pyproj.Geod.fwd returns). */
void orc_geod_direct(const orc_geod *g, double lat1, double lon1, double azi1,
                     double s12, double *lat2, double *lon2, double *azi2);

/* Inverse problem by shooting on the direct solver (forward azimuth at point
 * 1 and distance).  Valid for non-antipodal pairs; used for the short lines
 * of rotate_vectors / y_azimuth / pixel_size. */
void orc_geod_inverse(const orc_geod *g, double lat1, double lon1, double lat2,
                      double lon2, double *azi1, double *s12);

/* Vectorised WGS84 entry points for ctypes. */
void orc_wgs84_direct_n(long n, const double *lon1, const double *lat1,
                        const double *azi1, const double *s12, double *lon2,
                        double *lat2, double *azi2);
void orc_wgs84_inverse_n(long n, const double *lon1, const double *lat1,
                         const double *lon2, const double *lat2, double *azi1,
                         double *azi2_unused, double *s12);

#ifdef __cplusplus
}
#endif
#endif
