// Rate of the vector-memory gather path (TA / TCP) of one CU on gfx950 for the access shapes of the field samplers:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_bench.hip -o /tmp/gather_bench && /tmp/gather_bench
// Every lane issues NLD independent loads of WIDTH bytes per iteration from an L1-resident table of 208-byte node records
// (the C3 record), the lanes of a wave addressing records in one of these shapes:
//   same   all 64 lanes the same record            pairs   lanes 2j, 2j+1 the same record     quads  4 lanes per record
//   nodes  lane i -> record i (a sorted wave: one particle per cell)                            dense  lane i -> base + i*WIDTH
// Printed: shader cycles (at 2.4 GHz) per wave-instruction per CU with 8 waves per SIMD resident.  With rocprofv3 --pmc
// TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum the same binary gives the L1 accesses per wave-instruction of each shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int BLOCK = 256, ITER = 512, NLD = 8, REC = 208, NREC = 128;
typedef float f1;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <class T> __device__ float fold(T v);
template <> __device__ float fold<f1>(f1 v) { return v; }
template <> __device__ float fold<f2>(f2 v) { return v.x + v.y; }
template <> __device__ float fold<f4>(f4 v) { return v.x + v.y + v.z + v.w; }

template <class T, int SHAPE>
__global__ __launch_bounds__(BLOCK) void k_gather(const char *__restrict__ tab, float *sink) {
  const unsigned lane = threadIdx.x & 63;
  unsigned r = SHAPE == 0 ? 0u : SHAPE == 1 ? lane >> 1 : SHAPE == 2 ? lane >> 2 : lane;
  float acc = 0;
  for (int it = 0; it < ITER; ++it) {
    T v[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      unsigned off = SHAPE == 4 ? (lane * (unsigned)sizeof(T) + (unsigned)((it + k) & 7) * 1024u)
                                : ((r + (unsigned)(it & 31) + (unsigned)k) & (NREC - 1)) * REC + (unsigned)(k & 3) * 16u;
      asm volatile("" : "+v"(off));
      v[k] = *(const T *)(tab + off);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) acc += fold<T>(v[k]);
  }
  if (acc == 123456.789f) sink[0] = acc;
}

template <class T, int SHAPE>
void run(const char *name, const char *tab, float *sink, int cus) {
  const int blocks = cus * 8;   // 8 blocks x 4 waves = 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_gather<T, SHAPE>), dim3(blocks), dim3(BLOCK), 0, 0, tab, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_gather<T, SHAPE>), dim3(blocks), dim3(BLOCK), 0, 0, tab, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = 8.0 * 4 * ITER * NLD;
  printf("%-6s %2zu B/lane  %7.3f ms  %6.1f cycles per wave-instruction per CU  (%5.1f B/clk/CU)\n", name, sizeof(T), ms,
         ms * 1e-3 * 2.4e9 / instr_per_cu, 64.0 * sizeof(T) / (ms * 1e-3 * 2.4e9 / instr_per_cu));
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s CUs %d\n", p.gcnArchName, cus);
  char *tab; float *sink;
  hipMalloc(&tab, NREC * REC + 65536);
  hipMemset(tab, 0, NREC * REC + 65536);
  hipMalloc(&sink, 4);
#define ALL(T)                                                                                              \
  run<T, 0>("same", tab, sink, cus); run<T, 1>("pairs", tab, sink, cus); run<T, 2>("quads", tab, sink, cus); \
  run<T, 3>("nodes", tab, sink, cus); run<T, 4>("dense", tab, sink, cus);
  ALL(f1) ALL(f2) ALL(f4)
  return 0;
}
