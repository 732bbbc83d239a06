// Micro-benchmark (MI355X): cost of one 128-bit block of counter-based generators and of a few VALU instruction kinds,
// in SIMD issue cycles per wave -- to decide what the mixing kernel's random numbers should cost.
// hipcc --offload-arch=gfx950 -O3 -o rng_cost rng_cost.hip && ./rng_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint4 philox(uint4 c, unsigned k0, unsigned k1, int rounds) {
  for (int r = 0; r < rounds; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x, p1 = (unsigned long long)0xCD9E8D57u * c.z;
    c = make_uint4((unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ unsigned rotl(unsigned x, int r) { return __builtin_rotateleft32(x, r); }
// Threefry4x32 (Salmon et al. 2011): rotation constants of the 4x32 variant
template <int ROUNDS>
__device__ __forceinline__ uint4 threefry(uint4 c, uint4 k) {
  const unsigned ks[5] = {k.x, k.y, k.z, k.w, 0x1BD11BDAu ^ k.x ^ k.y ^ k.z ^ k.w};
  const int R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
  unsigned x0 = c.x + ks[0], x1 = c.y + ks[1], x2 = c.z + ks[2], x3 = c.w + ks[3];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    if (r % 2 == 0) {
      x0 += x1; x1 = rotl(x1, R[r % 8][0]) ^ x0;
      x2 += x3; x3 = rotl(x3, R[r % 8][1]) ^ x2;
    } else {
      x0 += x3; x3 = rotl(x3, R[r % 8][0]) ^ x0;
      x2 += x1; x1 = rotl(x1, R[r % 8][1]) ^ x2;
    }
    if (r % 4 == 3) {
      const int s = r / 4 + 1;
      x0 += ks[s % 5]; x1 += ks[(s + 1) % 5]; x2 += ks[(s + 2) % 5]; x3 += ks[(s + 3) % 5] + s;
    }
  }
  return make_uint4(x0, x1, x2, x3);
}

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, int n, unsigned seed) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  uint4 acc = make_uint4(t, seed, 0, 0);
  double f = (double)t * 1e-9 + 1.0;
  for (int i = 0; i < n; ++i) {
    if (KIND == 0) acc = philox(make_uint4(i, acc.x, t, acc.y), seed, 7u, 10);
    else if (KIND == 1) acc = philox(make_uint4(i, acc.x, t, acc.y), seed, 7u, 7);
    else if (KIND == 2) acc = threefry<20>(make_uint4(i, acc.x, t, acc.y), make_uint4(seed, 7u, 1u, 2u));
    else if (KIND == 3) acc = threefry<12>(make_uint4(i, acc.x, t, acc.y), make_uint4(seed, 7u, 1u, 2u));
    else if (KIND == 4) {   // 64 dependent 32-bit adds + xors (full-rate reference)
#pragma unroll
      for (int j = 0; j < 32; ++j) { acc.x += acc.y ^ (unsigned)j; acc.y ^= acc.x + 0x9E3779B9u; }
    } else if (KIND == 5) {   // 32 dependent 32x32->64 multiplies
#pragma unroll
      for (int j = 0; j < 32; ++j) { const unsigned long long q = (unsigned long long)acc.x * (0xD2511F53u + j); acc.x = (unsigned)(q >> 32) ^ (unsigned)q; }
    } else if (KIND == 6) {   // 32 dependent double fma
#pragma unroll
      for (int j = 0; j < 32; ++j) f = fma(f, 1.0000001, 1e-9);
    } else if (KIND == 7) {   // 32 dependent double sqrt (correctly rounded)
#pragma unroll
      for (int j = 0; j < 32; ++j) f = sqrt(f + 1.5);
    } else if (KIND == 8) {   // 32 u32 -> f64 conversions + add
#pragma unroll
      for (int j = 0; j < 32; ++j) { f += (double)(acc.x + j); acc.x ^= (unsigned)j * 77u; }
    }
  }
  out[t] = acc.x ^ acc.y ^ acc.z ^ acc.w ^ (unsigned)__double_as_longlong(f);
}

template <int KIND>
static void run(const char *name, double units_per_iter, const char *unit) {
  const int waves_per_simd = 4, nblk = 256 * waves_per_simd, n = 2000;   // 256 CUs x 4 SIMDs, one 256-thread block = one wave per SIMD of a CU
  unsigned *out;
  hipMalloc(&out, sizeof(unsigned) * nblk * 256);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<KIND><<<nblk, 256>>>(out, 10, 1u);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<KIND><<<nblk, 256>>>(out, n, 1u);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  // every SIMD runs waves_per_simd waves; per wave n iterations
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)n * waves_per_simd);
  printf("%-34s %8.3f ms   %8.1f SIMD cycles per iteration per wave (2.4 GHz)   = %.1f per %s\n", name, ms, cyc, cyc / units_per_iter, unit);
  hipFree(out);
}

int main() {
  run<0>("philox4x32-10 block", 1, "block");
  run<1>("philox4x32-7 block", 1, "block");
  run<2>("threefry4x32-20 block", 1, "block");
  run<3>("threefry4x32-12 block", 1, "block");
  run<4>("64 add/xor", 64, "instruction");
  run<5>("32 mul 32x32->64 (+xor)", 32, "multiply(+xor)");
  run<6>("32 fma f64", 32, "fma");
  run<7>("32 sqrt f64 (+add)", 32, "sqrt(+add)");
  run<8>("32 cvt u32->f64 + add (+2 int)", 32, "cvt+add(+2)");
  return 0;
}
