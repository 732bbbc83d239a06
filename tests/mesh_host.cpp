// TEST INFRASTRUCTURE: C entry points around opendrift_amd/csrc/odr_mesh.h (the host-side triangulation of a
// curvilinear reader mesh) so that tests/test_curvilinear_mesh.py can compare it with scipy's Delaunay on the CPU.
#include <cstring>

#include "../opendrift_amd/csrc/odr_mesh.h"

extern "C" {
void *mesh_build(const double *lon, const double *lat, int ny, int nx, long long *flips, long long *ntri, char *err, int errlen) {
  auto *m = new odr_mesh::Mesh();
  if (!odr_mesh::build(*m, lon, lat, ny, nx)) {
    strncpy(err, m->error.c_str(), errlen - 1);
    err[errlen - 1] = 0;
    delete m;
    return nullptr;
  }
  *flips = m->flips;
  *ntri = (long long)(m->tri_v.size() / 3);
  return m;
}
void mesh_tris(void *h, int32_t *v, int32_t *n) {
  auto *m = (odr_mesh::Mesh *)h;
  memcpy(v, m->tri_v.data(), m->tri_v.size() * 4);
  memcpy(n, m->tri_n.data(), m->tri_n.size() * 4);
}
void mesh_locate(void *h, long long n, const double *lon, const double *lat, double *x, double *y) {
  auto *m = (odr_mesh::Mesh *)h;
  for (long long k = 0; k < n; ++k) odr_mesh::locate(*m, lon[k], lat[k], x[k], y[k]);
}
void mesh_free(void *h) { delete (odr_mesh::Mesh *)h; }
}
