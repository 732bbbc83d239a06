// Host-side preparation of the lon/lat -> pixel lookup of a reader WITHOUT a projection
// (opendrift/readers/basereader/structured.py:44-113).  Plain C++ (no HIP): included by odrift.hip and compiled on
// its own by tests/ (g++) to compare the triangulation with scipy's qhull Delaunay on the CPU.
//
// The reference interpolates the pixel indices linearly over the Delaunay triangulation of the (lon, lat) nodes
// (scipy LinearNDInterpolator).  The nodes form a structured quad mesh, so the triangulation is built from the
// mesh instead of from a point cloud: every cell is split along its Delaunay diagonal, then illegal edges are
// flipped (Lawson) until every interior edge passes the in-circle test -- a triangulation whose edges are all
// locally Delaunay is the Delaunay triangulation of its vertices (restricted to the mesh outline; qhull also
// fills the concave parts of the outline with slivers between boundary nodes, which are treated as "outside"
// here).  For meshes that are conformal in the lon/lat plane no flip is needed; meshes sheared by the 1/cos(lat)
// stretch of longitude need some.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace odr_mesh {

struct Mesh {
  int nx = 0, ny = 0;
  std::vector<double> nodes;      // [ny*nx][2] lon, lat
  std::vector<int32_t> tri_v;     // [m][3] node indices, counter-clockwise in the lon/lat plane
  std::vector<int32_t> tri_n;     // [m][3] triangle across the edge opposite vertex k, -1 on the outline
  std::vector<int32_t> bucket;    // [nby*nbx] a triangle overlapping the bucket, -1: none
  int nbx = 0, nby = 0;
  double bx0 = 0, by0 = 0, ibx = 0, iby = 0;
  long long flips = 0;
  std::string error;
};

static inline double incircle(const double *a, const double *b, const double *c, const double *d) {
  // > 0: d inside the circumcircle of the counter-clockwise triangle a b c
  const double ax = a[0] - d[0], ay = a[1] - d[1], bx = b[0] - d[0], by = b[1] - d[1], cx = c[0] - d[0], cy = c[1] - d[1];
  return (ax * ax + ay * ay) * (bx * cy - by * cx) - (bx * bx + by * by) * (ax * cy - ay * cx) +
         (cx * cx + cy * cy) * (ax * by - ay * bx);
}
static inline double orient2d(const double *a, const double *b, const double *c) {
  return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]);
}

// false + m.error when the mesh folds / has non-convex cells / contains non-finite nodes
static bool build(Mesh &m, const double *lon, const double *lat, int ny, int nx) {
  m.nx = nx; m.ny = ny;
  if (nx < 2 || ny < 2) { m.error = "a curvilinear grid needs at least 2x2 nodes"; return false; }
  const size_t nn = (size_t)nx * ny;
  m.nodes.resize(2 * nn);
  double lo0 = INFINITY, lo1 = -INFINITY, la0 = INFINITY, la1 = -INFINITY;
  for (size_t k = 0; k < nn; ++k) {
    if (!std::isfinite(lon[k]) || !std::isfinite(lat[k])) { m.error = "lon/lat must be finite at every node"; return false; }
    m.nodes[2 * k] = lon[k]; m.nodes[2 * k + 1] = lat[k];
    lo0 = std::min(lo0, lon[k]); lo1 = std::max(lo1, lon[k]); la0 = std::min(la0, lat[k]); la1 = std::max(la1, lat[k]);
  }
  auto P = [&](int32_t v) { return &m.nodes[2 * (size_t)v]; };
  const int ncx = nx - 1, ncy = ny - 1;
  const size_t nt = (size_t)2 * ncx * ncy;
  m.tri_v.assign(3 * nt, -1);
  m.tri_n.assign(3 * nt, -1);
  double orient = 0, scale4 = 0;
  for (int j = 0; j < ncy; ++j)
    for (int i = 0; i < ncx; ++i) {
      const int32_t a = j * nx + i, b = a + 1, d = a + nx, c = d + 1;
      const double c1 = orient2d(P(a), P(b), P(d)), c2 = orient2d(P(c), P(d), P(b)), c3 = orient2d(P(b), P(c), P(a)),
                   c4 = orient2d(P(d), P(a), P(c));
      const double sg = c1 > 0 ? 1 : -1;
      if (orient == 0) orient = sg;
      if (!(c1 * sg > 0 && c2 * sg > 0 && c3 * sg > 0 && c4 * sg > 0 && sg == orient)) {
        m.error = "cell (" + std::to_string(i) + ", " + std::to_string(j) + ") is degenerate, non-convex or folded";
        return false;
      }
      scale4 = std::max(scale4, c1 * c1);
    }
  const double tol = 1e-9 * scale4;   // ties (rectangles, isosceles trapezoids) are left as they are
  // counter-clockwise vertex order: (a b c d) as they are for a right-handed mesh, mirrored otherwise
  auto set_tri = [&](size_t t, int32_t v0, int32_t v1, int32_t v2) {
    if (orient < 0) std::swap(v1, v2);
    m.tri_v[3 * t] = v0; m.tri_v[3 * t + 1] = v1; m.tri_v[3 * t + 2] = v2;
  };
  for (int j = 0; j < ncy; ++j)
    for (int i = 0; i < ncx; ++i) {
      const int32_t a = j * nx + i, b = a + 1, d = a + nx, c = d + 1;
      const size_t t = (size_t)2 * (j * ncx + i);
      const double ic = orient > 0 ? incircle(P(a), P(b), P(c), P(d)) : incircle(P(a), P(c), P(b), P(d));
      if (ic > 0) { set_tri(t, a, b, d); set_tri(t + 1, b, c, d); }   // diagonal b-d
      else { set_tri(t, a, b, c); set_tri(t + 1, a, c, d); }          // diagonal a-c
    }
  // adjacency through the sorted list of undirected edges
  struct E { uint64_t key; int32_t t, k; };
  std::vector<E> edges(3 * nt);
  for (size_t t = 0; t < nt; ++t)
    for (int k = 0; k < 3; ++k) {
      uint32_t p = (uint32_t)m.tri_v[3 * t + (k + 1) % 3], q = (uint32_t)m.tri_v[3 * t + (k + 2) % 3];
      if (p > q) std::swap(p, q);
      edges[3 * t + k] = {((uint64_t)p << 32) | q, (int32_t)t, k};
    }
  std::sort(edges.begin(), edges.end(), [](const E &x, const E &y) { return x.key < y.key; });
  for (size_t e = 0; e + 1 < edges.size(); ++e)
    if (edges[e].key == edges[e + 1].key) {
      m.tri_n[3 * (size_t)edges[e].t + edges[e].k] = edges[e + 1].t;
      m.tri_n[3 * (size_t)edges[e + 1].t + edges[e + 1].k] = edges[e].t;
      ++e;
    }
  // Lawson flips
  std::vector<std::pair<int32_t, int32_t>> stack;
  stack.reserve(3 * nt);
  for (size_t t = 0; t < nt; ++t)
    for (int k = 0; k < 3; ++k)
      if (m.tri_n[3 * t + k] > (int32_t)t) stack.emplace_back((int32_t)t, k);
  const long long max_flips = 64LL * (long long)nt + 1024;
  while (!stack.empty()) {
    const int32_t t = stack.back().first;
    const int k = stack.back().second;
    stack.pop_back();
    const int32_t n = m.tri_n[3 * (size_t)t + k];
    if (n < 0) continue;
    int32_t *vt = &m.tri_v[3 * (size_t)t], *vn = &m.tri_v[3 * (size_t)n];
    int32_t *nt_ = &m.tri_n[3 * (size_t)t], *nn_ = &m.tri_n[3 * (size_t)n];
    int kk = nn_[0] == t ? 0 : nn_[1] == t ? 1 : 2;
    if (nn_[kk] != t) continue;  // stale entry
    const int k1 = (k + 1) % 3, k2 = (k + 2) % 3, kk1 = (kk + 1) % 3, kk2 = (kk + 2) % 3;
    const int32_t a = vt[k], p = vt[k1], q = vt[k2], d = vn[kk];
    if (vn[kk1] != q || vn[kk2] != p) continue;  // stale entry (the pair was flipped meanwhile)
    if (!(incircle(P(a), P(p), P(q), P(d)) > tol)) continue;
    if (!(orient2d(P(a), P(p), P(d)) > 0 && orient2d(P(a), P(d), P(q)) > 0)) continue;  // not strictly convex
    const int32_t t_qa = nt_[k1], t_ap = nt_[k2], n_pd = nn_[kk1], n_dq = nn_[kk2];
    vt[0] = a; vt[1] = p; vt[2] = d; nt_[0] = n_pd; nt_[1] = n; nt_[2] = t_ap;
    vn[0] = a; vn[1] = d; vn[2] = q; nn_[0] = n_dq; nn_[1] = t_qa; nn_[2] = t;
    if (n_pd >= 0) { int32_t *x = &m.tri_n[3 * (size_t)n_pd]; for (int e = 0; e < 3; ++e) if (x[e] == n) x[e] = t; }
    if (t_qa >= 0) { int32_t *x = &m.tri_n[3 * (size_t)t_qa]; for (int e = 0; e < 3; ++e) if (x[e] == t) x[e] = n; }
    stack.emplace_back(t, 0); stack.emplace_back(t, 2); stack.emplace_back(n, 0); stack.emplace_back(n, 1);
    if (++m.flips > max_flips) { m.error = "edge flipping did not terminate"; return false; }
  }
  // start raster: about one bucket per cell; every bucket overlapped by a triangle's bounding box gets that triangle
  m.nbx = std::max(1, std::min(4096, nx)); m.nby = std::max(1, std::min(4096, ny));
  const double bw = (lo1 - lo0) / m.nbx * (1 + 1e-12) + 1e-300, bh = (la1 - la0) / m.nby * (1 + 1e-12) + 1e-300;
  m.bx0 = lo0; m.by0 = la0; m.ibx = 1.0 / bw; m.iby = 1.0 / bh;
  m.bucket.assign((size_t)m.nbx * m.nby, -1);
  for (size_t t = 0; t < nt; ++t) {
    double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    for (int k = 0; k < 3; ++k) {
      const double *q = P(m.tri_v[3 * t + k]);
      x0 = std::min(x0, q[0]); x1 = std::max(x1, q[0]); y0 = std::min(y0, q[1]); y1 = std::max(y1, q[1]);
    }
    const int bx0 = std::max(0, (int)((x0 - lo0) * m.ibx)), bx1 = std::min(m.nbx - 1, (int)((x1 - lo0) * m.ibx));
    const int by0 = std::max(0, (int)((y0 - la0) * m.iby)), by1 = std::min(m.nby - 1, (int)((y1 - la0) * m.iby));
    for (int by = by0; by <= by1; ++by)
      for (int bx = bx0; bx <= bx1; ++bx) m.bucket[(size_t)by * m.nbx + bx] = (int32_t)t;
  }
  return true;
}

// The walk + interpolation of csrc/odr_field.hip.h::curvi_locate, on the host (tests compare it with scipy; the
// device kernel is compared with both).
static inline void locate(const Mesh &m, double lon, double lat, double &x, double &y) {
  x = y = NAN;
  const double fx = (lon - m.bx0) * m.ibx, fy = (lat - m.by0) * m.iby;
  if (!(fx >= 0 && fy >= 0 && fx < (double)m.nbx && fy < (double)m.nby)) return;
  int32_t t = m.bucket[(size_t)(int)fy * m.nbx + (int)fx];
  if (t < 0) return;
  const int maxit = 4 * (m.nx + m.ny) + 16;
  for (int it = 0;; ++it) {
    if (it >= maxit) return;
    const int32_t *v = &m.tri_v[3 * (size_t)t];
    const double *p0 = &m.nodes[2 * (size_t)v[0]], *p1 = &m.nodes[2 * (size_t)v[1]], *p2 = &m.nodes[2 * (size_t)v[2]];
    const double q[2] = {lon, lat};
    const double e0 = orient2d(p1, p2, q), e1 = orient2d(p2, p0, q), e2 = orient2d(p0, p1, q);
    const double worst = std::min(e0, std::min(e1, e2));
    if (!(worst < 0)) {
      const double det = orient2d(p0, p1, p2);
      const double c0 = e0 / det, c1 = e1 / det;
      const int i0 = v[0] % m.nx, j0 = v[0] / m.nx, i1 = v[1] % m.nx, j1 = v[1] / m.nx, i2 = v[2] % m.nx, j2 = v[2] / m.nx;
      x = i2 + (c0 * (i0 - i2) + c1 * (i1 - i2));
      y = j2 + (c0 * (j0 - j2) + c1 * (j1 - j2));
      return;
    }
    t = m.tri_n[3 * (size_t)t + (worst == e0 ? 0 : worst == e1 ? 1 : 2)];
    if (t < 0) return;
  }
}

}  // namespace odr_mesh
