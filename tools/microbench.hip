// Dynamic instruction counts of the device building blocks of the hot path (developer tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include tools/microbench.hip -o /tmp/microbench
//   rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU -d out -- /tmp/microbench
// Each kernel runs ONE building block per thread on realistic inputs; SQ_INSTS_VALU / SQ_WAVES is
// the dynamic VALU instruction count per wave (= per particle) of that block.
#include "../opendrift_amd/csrc/odrift.hip"
#include <random>

using namespace odr;
constexpr int N = 1 << 20;

__global__ void mb_baseline(const double *in, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = in[i] + in[i + N];
}
__global__ void mb_origin(const double *in, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  GeodOrigin o = geod_origin(in[i + N], in[i]);
  out[i] = o.sbet1 + o.cbet1 + o.tanphi1 + o.lon1n;
}
__global__ void mb_azimuth_sincos(const float *uv, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double s, c;
  azimuth_sincos_f32(uv[i], uv[i + N], s, c);
  out[i] = s + c;
}
__global__ void mb_atan2_lib(const float *uv, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = atan2((double)uv[i], (double)uv[i + N]);
}
__global__ void mb_atan2_fin(const float *uv, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = atan2_fin((double)uv[i], (double)uv[i + N]);
}
// accuracy of atan2_fin against the library: out[0] = max |diff| in ulps, out[1] = count of different float32 roundings
__global__ void mb_atan2_check(const float *uv, unsigned long long *res) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double a = atan2_fin((double)uv[i], (double)uv[i + N]), b = atan2((double)uv[i], (double)uv[i + N]);
  double ulp = fabs(b) * 2.220446049250313e-16;
  unsigned long long d = (unsigned long long)(fabs(a - b) / ulp * 16.0);   // 1/16 ulp units
  atomicMax(&res[0], d);
  if ((float)a != (float)b) atomicAdd(&res[1], 1ull);
}
__global__ void mb_speed(const float *uv, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = speed_f32(uv[i], uv[i + N]);
}
__global__ void mb_direct_sc(const double *in, const float *uv, double *out) {  // origin + direct: subtract mb_origin
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  GeodOrigin o = geod_origin(in[i + N], in[i]);
  double la, lo;
  double h = 1.0 / sqrt((double)uv[i] * uv[i] + (double)uv[i + N] * uv[i + N]);
  geod_direct_sc(o, uv[i] * h, uv[i + N] * h, 300.0 + in[i], la, lo);  // + ~30 for the normalisation
  out[i] = la + lo;
}
__global__ void mb_zbracket(const DevWorld *W, const double *in, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  ZBracket zb = zbracket(W->src[0], in[i + 2 * N]);
  out[i] = zb.iz0 + zb.same + zb.wa;
}
__global__ void mb_uv3d(const DevWorld *W, UVTime tm, const double *in, double *out) {  // includes zbracket
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const DevSource &s = W->src[0];
  ZBracket zb = zbracket(s, in[i + 2 * N]);
  float u, v;
  uv_sample_fast<PROJ_LATLONG, true>(s, s.slot[0], tm, in[i], in[i + N], in[i + 2 * N], zb, 0.f, 0.f, u, v);
  out[i] = u + v;
}
__global__ void mb_env(const DevWorld *W, EnvGroupDesc G, const double *in, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float o[MAXG];
  env_group_fast<PROJ_LATLONG>(*W, G, in[i], in[i + N], in[i + 2 * N], o);
  double a = 0;
#pragma unroll
  for (int k = 0; k < MAXG; ++k) if (k < G.nv) a += o[k];
  out[i] = a;
}
__global__ void mb_uv2d_polar(const DevWorld *W, UVTime tm, const double *in, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const DevSource &s = W->src[1];
  ZBracket zb; zb.iz0 = 0; zb.same = 0; zb.wa = 1;
  float u, v;
  uv_sample_fast<PROJ_STERE_POLAR, false>(s, s.slot[0], tm, in[i + 3 * N], in[i + 4 * N], 0.0, zb, 0.f, 0.f, u, v);
  out[i] = u + v;
}
__global__ void mb_projfwd_polar(const DevWorld *W, const double *in, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double x, y;
  proj_fwd(W->src[1].proj, in[i + 3 * N], in[i + 4 * N], x, y);
  out[i] = x + y;
}
__global__ void mb_rotcs_polar(const DevWorld *W, const double *in, double *out) {  // includes proj_fwd
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double x, y, c, s;
  proj_fwd(W->src[1].proj, in[i + 3 * N], in[i + 4 * N], x, y);
  rotation_cs(W->src[1].proj, x, y, c, s);
  out[i] = c + s;
}
__global__ void mb_philox2(double *out, unsigned long long seed) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  rocrand_state_philox4x32_10 st;
  rng_init(st, seed, i, 7, RNG_OFF_VMIX);
  double2 u = rocrand_uniform_double2(&st);
  out[i] = u.x + u.y;
}

#define CK(x) do { int rc_ = (x); if (rc_) { printf("error %d at %s:%d: %s\n", rc_, __FILE__, __LINE__, odr_last_error()); return 1; } } while (0)

int main() {
  odr_ctx *c;
  CK(odr_ctx_create(0, 1, &c));
  std::mt19937_64 rng(1);
  std::uniform_real_distribution<double> U01(0, 1);
  // source 0: lat/lon 3-D grid 256 x 256 x 12, two time levels; source 1: polar-stereographic 2-D
  const int nx = 256, ny = 256, nz = 12;
  double z[nz];
  for (int k = 0; k < nz; ++k) z[k] = -(k * k * 1.5 + k);
  double dom[6] = {0, 10, 60, 66, -1e9, 1e9};
  odr_proj_desc pd; memset(&pd, 0, sizeof pd);
  pd.kind = 0;
  int32_t sid0, sid1;
  CK(odr_source_grid(c, &pd, dom, 1, 0, nz, z, &sid0));
  std::vector<float> a3((size_t)nx * ny * nz), a2((size_t)nx * ny);
  for (auto &v : a3) v = (float)(U01(rng) - 0.5);
  for (auto &v : a2) v = (float)(U01(rng) * 100 + 50);
  std::vector<float> land((size_t)nx * ny, 0.f);
  int32_t ids[6] = {VAR_U, VAR_V, VAR_W, VAR_KZ, VAR_DEPTH, VAR_LAND};
  const float *dat[6] = {a3.data(), a3.data(), a3.data(), a3.data(), a2.data(), land.data()};
  int32_t vnz[6] = {nz, nz, nz, nz, 1, 1};
  double xy8[8] = {0, 10, 60, 6, 0, 10.0 + 10.0 / (nx - 1), 60, 6.0 + 6.0 / (ny - 1)};
  for (int slot = 0; slot < 2; ++slot) CK(odr_block_upload(c, sid0, slot, slot * 3600.0, 6, ids, dat, vnz, ny, nx, xy8));
  pd.kind = PROJ_STERE_POLAR; pd.a = 6371000.0; pd.es = 0.0066943799901413165; pd.lat0_deg = 90; pd.lon0_deg = 70; pd.lat_ts_deg = 60; pd.k0 = 1;
  double dom1[6] = {0, 800.0 * 255, 0, 800.0 * 255, -1e9, 1e9};
  dom1[0] = -1.0e6; dom1[1] = dom1[0] + 800.0 * 255; dom1[2] = -2.5e6; dom1[3] = dom1[2] + 800.0 * 255;
  CK(odr_source_grid(c, &pd, dom1, 1, 0, 1, nullptr, &sid1));
  int32_t ids2[2] = {VAR_U, VAR_V};
  const float *dat2[2] = {a2.data(), a2.data()};
  int32_t vnz2[2] = {1, 1};
  double xy82[8] = {dom1[0], 800.0 * 255, dom1[2], 800.0 * 255, dom1[0], 800.0 * 256, dom1[2], 800.0 * 256};
  for (int slot = 0; slot < 2; ++slot) CK(odr_block_upload(c, sid1, slot, slot * 3600.0, 2, ids2, dat2, vnz2, ny, nx, xy82));
  int32_t l0[1] = {sid0};
  for (int v : {VAR_U, VAR_V, VAR_W, VAR_KZ, VAR_DEPTH, VAR_LAND}) CK(odr_env_bind(c, v, 1, l0, 0.f));
  CK(flush_world(c));

  std::vector<double> in((size_t)5 * N);
  std::vector<float> uv((size_t)2 * N);
  // positions clustered like a sorted particle set: consecutive threads are neighbours
  for (int i = 0; i < N; ++i) {
    double fx = (i % 1024) / 1024.0, fy = (i / 1024) / 1024.0;
    in[i] = 0.5 + 9 * fx + 1e-3 * U01(rng);
    in[i + N] = 60.3 + 5.4 * fy + 1e-3 * U01(rng);
    in[i + 2 * N] = -60 * U01(rng);
    uv[i] = (float)(U01(rng) - 0.5);
    uv[i + N] = (float)(U01(rng) - 0.5);
  }
  {  // lon/lat of points inside the polar grid
    odr_particles *dummy = nullptr; (void)dummy;
    for (int i = 0; i < N; ++i) {
      double x = dom1[0] + 800.0 * (5 + 245.0 * ((i % 1024) / 1024.0)), y = dom1[2] + 800.0 * (5 + 245.0 * ((i / 1024) / 1024.0));
      double rho = sqrt(x * x + y * y), lam = atan2(x, -y);     // sphere-ish inverse is enough for placing points
      double k0 = (1 + sin(60 * kDeg)) * 6371000.0;
      double phi = kHalfPi - 2 * atan(rho / k0);
      in[i + 3 * N] = 70 + lam / kDeg;
      in[i + 4 * N] = phi / kDeg;
    }
  }
  double *din, *dout; float *duv;
  hipMalloc(&din, in.size() * 8); hipMalloc(&dout, (size_t)N * 8); hipMalloc(&duv, uv.size() * 4);
  hipMemcpy(din, in.data(), in.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(duv, uv.data(), uv.size() * 4, hipMemcpyHostToDevice);
  const DevSource &s0 = c->hw.src[sid0], &s1 = c->hw.src[sid1];
  UVTime t0 = uv_time(s0, 1234.5), t1 = uv_time(s1, 1234.5);
  int grp[5] = {VAR_U, VAR_V, VAR_LAND, VAR_W, VAR_DEPTH};
  EnvGroupDesc G;
  if (!build_env_group(c, grp, 5, 1234.5, G)) { printf("no env group\n"); return 1; }
  dim3 g(N / 256), b(256);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(mb_baseline, g, b, 0, 0, din, dout);
    hipLaunchKernelGGL(mb_origin, g, b, 0, 0, din, dout);
    hipLaunchKernelGGL(mb_azimuth_sincos, g, b, 0, 0, duv, dout);
    hipLaunchKernelGGL(mb_speed, g, b, 0, 0, duv, dout);
    hipLaunchKernelGGL(mb_atan2_lib, g, b, 0, 0, duv, dout);
    hipLaunchKernelGGL(mb_atan2_fin, g, b, 0, 0, duv, dout);
    hipLaunchKernelGGL(mb_direct_sc, g, b, 0, 0, din, duv, dout);
    hipLaunchKernelGGL(mb_zbracket, g, b, 0, 0, c->dw, din, dout);
    hipLaunchKernelGGL(mb_uv3d, g, b, 0, 0, c->dw, t0, din, dout);
    hipLaunchKernelGGL(mb_env, g, b, 0, 0, c->dw, G, din, dout);
    hipLaunchKernelGGL(mb_uv2d_polar, g, b, 0, 0, c->dw, t1, din, dout);
    hipLaunchKernelGGL(mb_projfwd_polar, g, b, 0, 0, c->dw, din, dout);
    hipLaunchKernelGGL(mb_rotcs_polar, g, b, 0, 0, c->dw, din, dout);
    hipLaunchKernelGGL(mb_philox2, g, b, 0, 0, dout, 5ull);
  }
  {
    unsigned long long *dres, hres[2] = {0, 0};
    hipMalloc(&dres, 16); hipMemcpy(dres, hres, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mb_atan2_check, g, b, 0, 0, duv, dres);
    hipMemcpy(hres, dres, 16, hipMemcpyDeviceToHost);
    printf("atan2_fin vs library over %d random float32 pairs: max diff %.3f ulp, %llu different float32 roundings\n", N, hres[0] / 16.0, hres[1]);
  }
  hipDeviceSynchronize();
  std::vector<double> o(N);
  hipMemcpy(o.data(), dout, (size_t)N * 8, hipMemcpyDeviceToHost);
  printf("done %g %s\n", o[12345], hipGetErrorString(hipGetLastError()));
  return 0;
}
