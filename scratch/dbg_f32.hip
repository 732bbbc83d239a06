#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const float* x, const float* y, float* az, float* sp, float* at, double* at64, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  double a64 = atan2((double)x[i], (double)y[i]);
  float a = (float)a64;
  at[i] = a; at64[i] = a64;
  az[i] = __fmul_rn(a, 180.0f / 3.14159274101257324f);
  sp[i] = __fsqrt_rn(__fadd_rn(__fmul_rn(x[i], x[i]), __fmul_rn(y[i], y[i])));
}
int main() {
  int n = 1 << 16; std::mt19937 g(1); std::normal_distribution<float> d(0, 10);
  std::vector<float> x(n), y(n), az(n), sp(n), at(n); std::vector<double> at64(n);
  for (int i = 0; i < n; ++i) { x[i] = d(g); y[i] = d(g); }
  float *dx, *dy, *daz, *dsp, *dat; double* dat64;
  hipMalloc(&dx, 4 * n); hipMalloc(&dy, 4 * n); hipMalloc(&daz, 4 * n); hipMalloc(&dsp, 4 * n); hipMalloc(&dat, 4 * n); hipMalloc(&dat64, 8 * n);
  hipMemcpy(dx, x.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), 4 * n, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dy, daz, dsp, dat, dat64, n);
  hipMemcpy(az.data(), daz, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(sp.data(), dsp, 4 * n, hipMemcpyDeviceToHost);
  hipMemcpy(at.data(), dat, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(at64.data(), dat64, 8 * n, hipMemcpyDeviceToHost);
  int m_at = 0, m_az = 0, m_sp = 0, m64 = 0; double worst64 = 0;
  for (int i = 0; i < n; ++i) {
    double h64 = atan2((double)x[i], (double)y[i]);
    float ha = (float)h64;
    volatile float xx = x[i] * x[i], yy = y[i] * y[i]; volatile float s = xx + yy;
    float hs = sqrtf(s);
    float haz = ha * (180.0f / 3.14159274101257324f);
    m_at += ha != at[i]; m_az += haz != az[i]; m_sp += hs != sp[i]; m64 += h64 != at64[i];
    double e = fabs(h64 - at64[i]) / fabs(h64); if (e > worst64) worst64 = e;
  }
  printf("mismatch atan2f32 %d az %d speed %d of %d ; atan2 f64 mismatches %d worst rel %.3e\n", m_at, m_az, m_sp, n, m64, worst64);
}
