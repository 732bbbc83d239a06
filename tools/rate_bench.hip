// Issue rate of the VALU instruction classes of the hot path on gfx950 (MI355X): cycles per wave64 instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rate_bench.hip -o /tmp/rate_bench && /tmp/rate_bench
// Every kernel runs NCH independent dependency chains per lane (register-resident, no memory traffic) for ITER
// iterations; the grid fills every SIMD with WPS waves.  cycles / instruction / SIMD = elapsed shader cycles
// (s_memtime, first start to last end over all waves) x SIMDs / (waves x instructions per wave).
// The figure is what `roofline_issue` in bench.py prices a VALU instruction at (profiles/r03_rate_bench.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int NCH = 8, ITER = 2048, BLOCK = 256;

#define CHAIN8(OP, C)                                                                                            \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                                    \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])   \
               : "v"(C), "v"(C##2))

#define OP_FMA_F64(k) "v_fma_f64 %" #k ", %" #k ", %8, %9\n"
#define OP_MUL_F64(k) "v_mul_f64 %" #k ", %" #k ", %8\n"
#define OP_ADD_F64(k) "v_add_f64 %" #k ", %" #k ", %9\n"
#define OP_FMA_F32(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define OP_MUL_F32(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define OP_ADD_F32(k) "v_add_f32 %" #k ", %" #k ", %9\n"
#define OP_ADD_U32(k) "v_add_u32 %" #k ", %" #k ", %9\n"
#define OP_MULLO_U32(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define OP_PKFMA_F32(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define OP_PKMUL_F32(k) "v_pk_mul_f32 %" #k ", %" #k ", %8\n"
#define OP_RCP_F64(k) "v_rcp_f64 %" #k ", %" #k "\n"
#define OP_RSQ_F64(k) "v_rsq_f64 %" #k ", %" #k "\n"
#define OP_SQRT_F32(k) "v_sqrt_f32 %" #k ", %" #k "\n"

template <class T> struct Stamp { unsigned long long t0, t1; };

#define KERNEL(NAME, T, OP, INIT, CV, CV2)                                                       \
  __global__ __launch_bounds__(BLOCK) void NAME(unsigned long long *stamps, T *sink) {           \
    T a[NCH];                                                                                    \
    for (int k = 0; k < NCH; ++k) a[k] = (T)(INIT) + (T)threadIdx.x * (T)1e-3 + (T)k;            \
    T c = (T)(CV), c2 = (T)(CV2);                                                                \
    asm volatile("" : "+v"(c), "+v"(c2));                                                        \
    unsigned long long t0 = __builtin_readcyclecounter();                                        \
    for (int it = 0; it < ITER; ++it) { CHAIN8(OP, c); CHAIN8(OP, c); CHAIN8(OP, c); CHAIN8(OP, c); } \
    unsigned long long t1 = __builtin_readcyclecounter();                                        \
    T s = a[0];                                                                                  \
    for (int k = 1; k < NCH; ++k) s += a[k];                                                     \
    if ((threadIdx.x & 63) == 0) {                                                               \
      size_t w = ((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;                                \
      stamps[2 * w] = t0; stamps[2 * w + 1] = t1;                                                \
    }                                                                                            \
    if (s == (T)123456789) sink[0] = s;                                                          \
  }

typedef float __attribute__((ext_vector_type(2))) float2v;
KERNEL(k_fma_f64, double, OP_FMA_F64, 1.0, 0.999999, 1e-9)
KERNEL(k_mul_f64, double, OP_MUL_F64, 1.0, 0.999999, 1e-9)
KERNEL(k_add_f64, double, OP_ADD_F64, 1.0, 0.999999, 1e-9)
KERNEL(k_fma_f32, float, OP_FMA_F32, 1.0f, 0.9999f, 1e-6f)
KERNEL(k_mul_f32, float, OP_MUL_F32, 1.0f, 0.9999f, 1e-6f)
KERNEL(k_add_f32, float, OP_ADD_F32, 1.0f, 0.9999f, 1e-6f)
KERNEL(k_add_u32, unsigned, OP_ADD_U32, 1u, 3u, 7u)
KERNEL(k_mullo_u32, unsigned, OP_MULLO_U32, 1u, 3u, 7u)
KERNEL(k_pkfma_f32, double, OP_PKFMA_F32, 1.0, 0.999999, 1e-9)   // two float32 lanes in a 64-bit register pair
KERNEL(k_pkmul_f32, double, OP_PKMUL_F32, 1.0, 0.999999, 1e-9)
KERNEL(k_rcp_f64, double, OP_RCP_F64, 1.5, 0.999999, 1e-9)
KERNEL(k_rsq_f64, double, OP_RSQ_F64, 1.5, 0.999999, 1e-9)
KERNEL(k_sqrt_f32, float, OP_SQRT_F32, 1.5f, 0.9999f, 1e-6f)
// v_cmp_gt_f64 + v_addc_co_u32: the level search of the mixing loop (2 instructions per step)
__global__ __launch_bounds__(BLOCK) void k_cmp_addc(unsigned long long *stamps, double *sink) {
  unsigned n[NCH];
  double d[NCH];
  for (int k = 0; k < NCH; ++k) { d[k] = 1.0 + threadIdx.x * 1e-3 + k; n[k] = k; }
  double c = 3.5;
  asm volatile("" : "+v"(c));
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int k = 0; k < NCH; ++k)
        asm volatile("v_cmp_gt_f64 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(n[k]) : "v"(d[k]), "v"(c) : "vcc");
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned s = n[0];
  for (int k = 1; k < NCH; ++k) s += n[k];
  if ((threadIdx.x & 63) == 0) {
    size_t w = ((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    stamps[2 * w] = t0; stamps[2 * w + 1] = t1;
  }
  if (s == 123456789u) sink[0] = s;
}

// float32 -> float64 -> float32 conversions: separate register classes, own kernels
__global__ __launch_bounds__(BLOCK) void k_cvt(unsigned long long *stamps, float *sink) {
  float a[NCH];
  double d[NCH];
  for (int k = 0; k < NCH; ++k) a[k] = 1.0f + threadIdx.x * 1e-3f + k;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(a[k]));
#pragma unroll
      for (int k = 0; k < NCH; ++k) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[k]) : "v"(d[k]));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = a[0];
  for (int k = 1; k < NCH; ++k) s += a[k];
  if ((threadIdx.x & 63) == 0) {
    size_t w = ((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    stamps[2 * w] = t0; stamps[2 * w + 1] = t1;
  }
  if (s == 123456789.f) sink[0] = s;
}

template <class K, class T>
static void run(const char *name, K kern, T *sink, unsigned long long *dst, int wps, int cus, double ops_per_iter) {
  const int blocks = cus * wps;   // BLOCK = 256 = 4 waves, one per SIMD: wps blocks per CU = wps waves per SIMD
  const size_t waves = (size_t)blocks * (BLOCK / 64);
  std::vector<unsigned long long> h(2 * waves);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(BLOCK), 0, 0, dst, sink);
    hipDeviceSynchronize();
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(BLOCK), 0, 0, dst, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost);
  // per-wave cycles (its own start to its own end): with wps waves resident per SIMD for the whole run, a wave gets
  // 1/wps of the SIMD's issue slots
  std::vector<double> per(waves);
  for (size_t w = 0; w < waves; ++w) per[w] = (double)(h[2 * w + 1] - h[2 * w]);
  std::sort(per.begin(), per.end());
  const double med = per[waves / 2];
  const double instr = ops_per_iter * ITER;
  const double clk_ghz = med / (ms * 1e6);   // cycle-counter ticks per ns of the launch (ticks: see the note printed below)
  printf("%-14s waves/SIMD %d  ticks/wave %.0f  ticks per instruction per SIMD %.3f   launch %.3f ms  (ticks/ns %.3f)\n", name, wps,
         med, med / (instr * wps), ms, clk_ghz);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s  CUs %d  clockRate %d kHz  (s_memtime / readcyclecounter ticks; clock64 rate may differ from the shader clock:\n"
         " the ratio ticks/ns is printed per run -- divide 'ticks per instruction' by it and multiply by the shader GHz)\n",
         p.gcnArchName, cus, p.clockRate);
  unsigned long long *dst;
  hipMalloc(&dst, (size_t)cus * 8 * 4 * 2 * 8);
  double *sd; float *sf; unsigned *su;
  hipMalloc(&sd, 8); hipMalloc(&sf, 4); hipMalloc(&su, 4);
  for (int wps : {1, 2, 4, 8}) {
    run("v_fma_f64", k_fma_f64, sd, dst, wps, cus, 32);
    run("v_mul_f64", k_mul_f64, sd, dst, wps, cus, 32);
    run("v_add_f64", k_add_f64, sd, dst, wps, cus, 32);
    run("v_fma_f32", k_fma_f32, sf, dst, wps, cus, 32);
    run("v_mul_f32", k_mul_f32, sf, dst, wps, cus, 32);
    run("v_add_f32", k_add_f32, sf, dst, wps, cus, 32);
    run("v_add_u32", k_add_u32, su, dst, wps, cus, 32);
    run("v_mul_lo_u32", k_mullo_u32, su, dst, wps, cus, 32);
    run("v_pk_fma_f32", k_pkfma_f32, sd, dst, wps, cus, 32);
    run("v_pk_mul_f32", k_pkmul_f32, sd, dst, wps, cus, 32);
    run("v_rcp_f64", k_rcp_f64, sd, dst, wps, cus, 32);
    run("v_rsq_f64", k_rsq_f64, sd, dst, wps, cus, 32);
    run("v_sqrt_f32", k_sqrt_f32, sf, dst, wps, cus, 32);
    run("cmp_f64+addc", k_cmp_addc, sd, dst, wps, cus, 64);
    run("cvt f32<->f64", k_cvt, sf, dst, wps, cus, 32);
  }
  return 0;
}
