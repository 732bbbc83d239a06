/* TEST INFRASTRUCTURE ONLY -- CPU oracle, never on the product path.
 *
 * Restatement of opendrift/readers/interpolation/interpolators.py.  The
 * arithmetic inside scipy.ndimage.map_coordinates(order=1) and grey_dilation
 * (un-vendored SciPy) was probed in this container (SciPy 1.15.3) and is pinned
 * bit-for-bit against SciPy itself in tests/test_oracle_interp.py.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* expand_numpy_array, interpolators.py:9-20: every non-finite cell takes the
 * max of the finite cells in its 3x3 neighbourhood (grey_dilation(size=3),
 * border mode 'reflect' == duplicate the edge cell), or stays NaN. */
void orc_dilate_nan_once(float *a, int ny, int nx) {
  long n = (long)ny * nx, i;
  float *src = (float *)malloc(sizeof(float) * (size_t)n);
  int any = 0, y, x, dy, dx;
  for (i = 0; i < n; ++i) if (isfinite(a[i])) { any = 1; break; }
  if (!any) { free(src); return; } /* "Only NaNs, returning" */
  memcpy(src, a, sizeof(float) * (size_t)n);
  for (y = 0; y < ny; ++y)
    for (x = 0; x < nx; ++x) {
      float best = 0;
      int have = 0;
      if (isfinite(src[(long)y * nx + x])) continue;
      for (dy = -1; dy <= 1; ++dy)
        for (dx = -1; dx <= 1; ++dx) {
          int yy = y + dy, xx = x + dx;
          float v;
          if (yy < 0 || yy >= ny || xx < 0 || xx >= nx) continue;
          v = src[(long)yy * nx + xx];
          if (!isfinite(v)) continue;
          if (!have || v > best) { best = v; have = 1; }
        }
      a[(long)y * nx + x] = have ? best : NAN;
    }
  free(src);
}

/* fill_NaN_towards_seafloor, interpolators.py:203-211 (np.isnan mask, layer i <- layer i-1) */
void orc_fill_nan_towards_seafloor(float *a, int nz, int ny, int nx) {
  long plane = (long)ny * nx, i;
  int k;
  for (k = 1; k < nz; ++k)
    for (i = 0; i < plane; ++i)
      if (isnan(a[k * plane + i])) a[k * plane + i] = a[(k - 1) * plane + i];
}

/* map_coordinates(order=1) footprint along one axis (probed against SciPy 1.15.3).
 * mode constant: coordinate outside [0,n-1] => cval; index n is reached only with weight 0 and
 * is mirrored to n-2.  mode nearest: the coordinate is NOT clamped -- start = floor(c),
 * t = c - start, and both footprint indices are clamped to [0,n-1]. */
static int axis(double c, int n, int mode_nearest, int *i0, int *i1, double *t) {
  double fl;
  if (!mode_nearest && !(c >= 0 && c <= n - 1)) return 0;
  if (c != c) return 0;
  fl = floor(c);
  *t = c - fl;
  if (fl < -1) fl = -1;
  if (fl > n) fl = n;
  *i0 = (int)fl;
  *i1 = *i0 + 1;
  if (mode_nearest) {
    if (*i0 < 0) *i0 = 0;
    if (*i0 > n - 1) *i0 = n - 1;
    if (*i1 < 0) *i1 = 0;
    if (*i1 > n - 1) *i1 = n - 1;
  } else if (*i1 > n - 1) {
    *i1 = n >= 2 ? n - 2 : 0;
  }
  return 1;
}

float orc_bilinear_f32(const float *a, int ny, int nx, double yi, double xi,
                       int mode_nearest) {
  int y0, y1, x0, x1;
  double ty, tx, t;
  if (!axis(yi, ny, mode_nearest, &y0, &y1, &ty)) return NAN;
  if (!axis(xi, nx, mode_nearest, &x0, &x1, &tx)) return NAN;
  /* SciPy accumulates sum_k (value * w_y) * w_x in double, row-major footprint */
  t = ((double)a[(long)y0 * nx + x0] * (1 - ty)) * (1 - tx);
  t += ((double)a[(long)y0 * nx + x1] * (1 - ty)) * tx;
  t += ((double)a[(long)y1 * nx + x0] * ty) * (1 - tx);
  t += ((double)a[(long)y1 * nx + x1] * ty) * tx;
  return (float)t;
}

/* Linear2DInterpolator.__call__, interpolators.py:113-139 */
void orc_linear2d_call(float *a, int ny, int nx, long n, const double *yi,
                       const double *xi, float *out) {
  long i, nmiss = 0;
  int any = 0, it = 0;
  for (i = 0; i < (long)ny * nx; ++i) if (isfinite(a[i])) { any = 1; break; }
  if (!any) { for (i = 0; i < n; ++i) out[i] = NAN; return; }
  for (i = 0; i < n; ++i) {
    out[i] = orc_bilinear_f32(a, ny, nx, yi[i], xi[i], 0);
    if (!isfinite(out[i])) ++nmiss;
  }
  while (nmiss > 0) {
    if (++it > 10) return; /* "Still NaN-values after 10 iterations, exiting!" */
    orc_dilate_nan_once(a, ny, nx);
    nmiss = 0;
    for (i = 0; i < n; ++i) {
      if (isfinite(out[i])) continue;
      out[i] = orc_bilinear_f32(a, ny, nx, yi[i], xi[i], 1);
      if (!isfinite(out[i])) ++nmiss;
    }
  }
}
