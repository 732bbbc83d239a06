// TEST INFRASTRUCTURE: the per-element arithmetic of opendrift_amd/csrc/odr_oil.hip.h (the device code of the oil
// physics inside OpenOil's mixing loop) compiled for the CPU with g++ -ffp-contract=off, so that it can be compared
// with the NumPy oracle without a GPU (tests/test_oil_device_arithmetic.py).  The HIP rounding intrinsics are IEEE
// single operations; the kernels themselves (reductions, scan, launch code) are excluded by ODR_OIL_HOST.
#include <cmath>
#include <cstddef>
#include <cstdint>

#define ODR_OIL_HOST 1
#define __device__
#define __forceinline__ inline
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }
static inline double __dadd_rn(double a, double b) { volatile double r = a + b; return r; }
static inline double __dsub_rn(double a, double b) { volatile double r = a - b; return r; }
static inline double __ddiv_rn(double a, double b) { volatile double r = a / b; return r; }
struct double2 { double x, y; };
struct uint4 { unsigned x, y, z, w; };

namespace odr {
constexpr int BLOCK = 256;
constexpr int NVAR = 18;
constexpr double kPi = 3.14159265358979323846;
enum { VAR_U = 0, VAR_V = 1, VAR_XWIND = 2, VAR_YWIND = 3, VAR_W = 4, VAR_KZ = 5, VAR_SX = 6, VAR_SY = 7, VAR_LAND = 8,
       VAR_DEPTH = 9, VAR_SSH = 10, VAR_HDIFF = 11, VAR_HS = 12, VAR_TP = 13, VAR_MLD = 14, VAR_TEMP = 15, VAR_SALT = 16 };
struct PView {
  long long n;
  int *id;
  float *aux[9];
  float *env[NVAR];
};
static inline float speed_f32(float xv, float yv) { return sqrtf(__fadd_rn(__fmul_rn(xv, xv), __fmul_rn(yv, yv))); }
// (the device generator is not part of this shim: it runs with host-drawn numbers, rng_mode = 1)
static inline uint4 rng_block(unsigned long long, int, unsigned long long, unsigned long long) { return uint4{0, 0, 0, 0}; }
static inline double rng_u53(unsigned, unsigned) { return 0.5; }
static inline double2 rng_uniform2(uint4) { return double2{0.5, 0.5}; }
}  // namespace odr
#include "../opendrift_amd/csrc/odr_oil.hip.h"

using namespace odr;

extern "C" {
// per element: entrainment probability, terminal velocity of the current droplet and of the droplet it would get,
// kinematic water viscosity, median droplet diameter of the spectrum, zb = 1.5 Hs
void oilh_elements(long long n, const float *xw, const float *yw, const float *temp, const float *salt, const float *hs,
                   const float *tp, const float *diameter, const float *density, const float *viscosity, const float *film,
                   const float *d_if, double sigma_ow, double rho_w, double dt_mix, int droplets, int hs_mode, int tp_mode,
                   int to_kelvin, double *prob, double *w_now, double *w_if, float *nyw, double *dv50, float *zb) {
  PView p{};
  p.n = n;
  p.env[VAR_XWIND] = (float *)xw; p.env[VAR_YWIND] = (float *)yw; p.env[VAR_TEMP] = (float *)temp; p.env[VAR_SALT] = (float *)salt;
  p.env[VAR_HS] = (float *)hs; p.env[VAR_TP] = (float *)tp;
  p.aux[OIL_DIAMETER] = (float *)diameter; p.aux[OIL_DENSITY] = (float *)density; p.aux[OIL_VISCOSITY] = (float *)viscosity;
  p.aux[OIL_FILM] = (float *)film; p.aux[OIL_DIAMETER_IF_ENTRAINED] = (float *)d_if;
  double stat[OIL_STAT_N] = {1.0, 1e-4, 0, 0};
  OilArgs a{};
  a.hs_mode = hs_mode; a.tp_mode = tp_mode; a.to_kelvin = to_kelvin; a.droplets = droplets; a.rng_mode = 1;
  a.sigma_ow = sigma_ow; a.rho_w = rho_w; a.dt_mix_cfg = dt_mix; a.stat = stat;
  for (long long i = 0; i < n; ++i) {
    OilLane L;
    L.init(p, i, a);
    prob[i] = L.prob;
    nyw[i] = L.nyw;
    w_now[i] = L.terminal_velocity();
    L.d = L.d_if; L.dirty = true;
    w_if[i] = L.terminal_velocity();
    const float H = oil_hs(p, i, hs_mode, speed_f32(xw[i], yw[i]));
    zb[i] = __fmul_rn(1.5f, H);
    dv50[i] = oil_dv50_element(p, i, a, H);
  }
}

// the cumulative spectrum as the device builds it (chunks of OIL_SPEC_CHUNK points, fixed summation order) and the
// np.random.choice lookup of k_oil_choice
void oilh_choice(double dv50, long long n, const double *u, double *diameter, long long *index) {
  static double cdf[OIL_NSPEC];
  const double ldv = log(dv50);
  double run = 0;
  for (int c = 0; c < OIL_SPEC_BLOCKS; ++c) {            // k_oil_spectrum_sums / _offsets / _scan
    double tsum[BLOCK];
    for (int t = 0; t < BLOCK; ++t) {
      double s = 0;
      for (int j = 0; j < OIL_SPEC_PER_THREAD; ++j) {
        const int k = c * OIL_SPEC_CHUNK + t * OIL_SPEC_PER_THREAD + j;
        if (k < OIL_NSPEC) s += oil_spectrum_at(k, ldv);
      }
      tsum[t] = s;
    }
    double csum = 0;
    for (int t = 0; t < BLOCK; ++t) {
      double off = run;
      for (int k = 0; k < t; ++k) off += tsum[k];
      for (int j = 0; j < OIL_SPEC_PER_THREAD; ++j) {
        const int k = c * OIL_SPEC_CHUNK + t * OIL_SPEC_PER_THREAD + j;
        if (k < OIL_NSPEC) { off += oil_spectrum_at(k, ldv); cdf[k] = off; }
      }
      csum += tsum[t];
    }
    run += csum;
  }
  const double total = cdf[OIL_NSPEC - 1];
  static int guide[OIL_GUIDE + 1];                       // k_oil_guide
  for (int j = 0; j <= OIL_GUIDE; ++j) guide[j] = oil_search_right(cdf, total, (double)j / (double)OIL_GUIDE, 0, OIL_NSPEC);
  for (long long i = 0; i < n; ++i) {                    // k_oil_choice
    int lo = 0, hi = OIL_NSPEC;
    if (u[i] >= 0.0 && u[i] < 1.0) {
      const int j = (int)(u[i] * (double)OIL_GUIDE);
      lo = guide[j];
      hi = guide[j + 1];
    }
    lo = oil_search_right(cdf, total, u[i], lo, hi);
    const int plain = oil_search_right(cdf, total, u[i], 0, OIL_NSPEC);
    if (plain != lo) lo = -1000000;                      // the bracketed search must equal the plain one
    if (lo > OIL_NSPEC - 1) lo = OIL_NSPEC - 1;
    index[i] = lo;
    diameter[i] = (double)(float)oil_diameter_of(lo);
  }
}
}
