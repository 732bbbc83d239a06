// OpenOil: the per-particle oil physics that sits INSIDE the vertical-mixing loop.
//
//   OpenOil.update_terminal_velocity            models/openoil/openoil.py:922-998      OilLane::terminal_velocity
//   PhysicsMethods.sea_water_density            models/physics_methods.py:574-608      oil_sea_water_density_f32
//   seawater_dynamic_viscosity_sharqawy         models/physics_methods.py:159-178      oil_water_viscosity_f32
//   OpenOil.prepare_vertical_mixing             models/openoil/openoil.py:1017-1031    OilLane::init (probability), k_oil_*
//   oil_wave_entrainment_rate_li2017            models/physics_methods.py:115-137      oil_entrainment_probability
//   get_wave_breaking_droplet_diameter_*        models/openoil/openoil.py:1072-1172    oil_dv50_element, k_oil_stats*, k_oil_spectrum_*, k_oil_guide, k_oil_choice
//   surface_stick / surface_wave_mixing         models/openoil/openoil.py:1033-1061    OilLane::surface_wave_mixing
//
// Everything is evaluated with the operand dtypes NumPy 2 gives the reference: the environment (wind, wave height,
// temperature, salinity) and the element properties diameter / oil_film_thickness are float32 arrays, Python float
// constants multiply them in float32, density and viscosity are float64 after oil_weathering_noaa (:743-760; stored
// as float32 on the device like the Oil element type declares them), sea_water_density() with its default arguments
// is a NumPy float64 scalar (it promotes), and the wave period derived from the wind has passed through the float32
// environment recarray (calculate_missing_environment_variables, physics_methods.py:876-883).
//
// Included by odr_kernels.hip.h (needs PView, speed_f32, rng_init).
#pragma once

namespace odr {

enum { OIL_DIAMETER = 0, OIL_DENSITY = 1, OIL_VISCOSITY = 2, OIL_FILM = 3, OIL_DIAMETER_IF_ENTRAINED = 4 };  // property slots
enum { OIL_STAT_MEAN_ZB = 0, OIL_STAT_DV50 = 1, OIL_STAT_CDF_TOTAL = 2, OIL_STAT_N = 4 };
constexpr unsigned long long RNG_OFF_OIL_ENTRAIN = 5120, RNG_OFF_OIL_DIAMETER = 8000;   // Philox offsets within a step (32-bit units)
constexpr int OIL_MAX_SUBSTEPS_DEVICE_RNG = (8000 - 5120) / 4;
constexpr int OIL_NSPEC = 1000000;              // np.linspace(1e-6, 3e-3, 1000000) (openoil.py:1081,1131)
constexpr int OIL_SPEC_PER_THREAD = 16;
constexpr int OIL_SPEC_CHUNK = BLOCK * OIL_SPEC_PER_THREAD;
constexpr int OIL_SPEC_BLOCKS = (OIL_NSPEC + OIL_SPEC_CHUNK - 1) / OIL_SPEC_CHUNK;

struct OilArgs {
  int keep_diameter;       // seed_elements(diameter=...): entrained elements keep their droplet size (:1051)
  int hs_mode, tp_mode;    // provenance of wave height / period: 0 environment variable, 1 from the wind (float64),
                           // 2 none, 3 (period) from the wind via the float32 environment
  int to_kelvin;           // oil_weathering_noaa converted the temperature in place (:722-724)
  int droplets;            // 1 Johansen et al. (2015), 2 Li et al. (2017)
  int rng_mode;
  double sigma_ow;         // oil_water_interfacial_tension
  double rho_w;            // PhysicsMethods.sea_water_density() (T=10, S=35)
  double dt_mix_cfg;       // vertical_mixing:timestep
  const double *u_ent, *u_int;   // ODR_RNG_HOST: [ntimes][n] entrainment uniforms / unit intrusion depths
  const double *stat;      // OIL_STAT_*
};

// ---- float32 chains (Python float constants are cast to float32, no contraction)
#define OF(x) ((float)(x))
__device__ __forceinline__ float oil_sea_water_density_f32(float T, float S) {
  float R1 = __fsub_rn(__fmul_rn(OF(6.536332E-09), T), OF(1.120083E-06));
  R1 = __fadd_rn(__fmul_rn(R1, T), OF(1.001685E-04));
  R1 = __fsub_rn(__fmul_rn(R1, T), OF(9.095290E-03));
  R1 = __fadd_rn(__fmul_rn(R1, T), OF(6.793952E-02));
  R1 = __fsub_rn(__fmul_rn(R1, T), OF(28.263737));
  float R2 = __fsub_rn(__fmul_rn(OF(5.3875E-09), T), OF(8.2467E-07));
  R2 = __fadd_rn(__fmul_rn(R2, T), OF(7.6438E-05));
  R2 = __fsub_rn(__fmul_rn(R2, T), OF(4.0899E-03));
  R2 = __fadd_rn(__fmul_rn(R2, T), OF(8.24493E-01));
  float R3 = __fadd_rn(__fmul_rn(OF(-1.6546E-06), T), OF(1.0227E-04));
  R3 = __fsub_rn(__fmul_rn(R3, T), OF(5.72466E-03));
  const float in = __fadd_rn(__fadd_rn(__fmul_rn(OF(4.8314E-04), S), __fmul_rn(R3, sqrtf(S))), R2);
  const float SIG = __fadd_rn(R1, __fmul_rn(in, S));
  return __fadd_rn(__fadd_rn(SIG, OF(28.106331)), 1000.f);
}

__device__ __forceinline__ float oil_water_viscosity_f32(float T, float S) {
  const float t1 = __fadd_rn(T, OF(64.993));
  const float mu_w = __fadd_rn(OF(4.2844e-5), __fdiv_rn(1.0f, __fsub_rn(__fmul_rn(OF(0.157), __fmul_rn(t1, t1)), OF(91.296))));
  const float T2 = __fmul_rn(T, T);
  const float A = __fsub_rn(__fadd_rn(OF(1.541), __fmul_rn(OF(1.998e-2), T)), __fmul_rn(OF(9.52e-5), T2));
  const float B = __fadd_rn(__fsub_rn(OF(7.974), __fmul_rn(OF(7.561e-2), T)), __fmul_rn(OF(4.724e-4), T2));
  const float s = __fdiv_rn(S, 1000.f);
  return __fmul_rn(mu_w, __fadd_rn(__fadd_rn(1.f, __fmul_rn(A, s)), __fmul_rn(B, __fmul_rn(s, s))));
}

// significant_wave_height() (physics_methods.py:893-907), float32
__device__ __forceinline__ float oil_hs(const PView &p, long long i, int hs_mode, float ws) {
  if (hs_mode == 0) return p.env[VAR_HS][i];
  if (hs_mode == 1) return __fmul_rn(OF(0.0246), __fmul_rn(ws, ws));
  return 0.f;
}

// sea_surface_wave_breaking_fraction() (physics_methods.py:961-966): 0.032*(wind_speed - 5)/wave_period, < 0 -> 0
__device__ __forceinline__ double oil_breaking_fraction(const PView &p, long long i, int tp_mode, float ws) {
  const float num = __fmul_rn(OF(0.032), __fsub_rn(ws, 5.f));
  double f;
  if (tp_mode == 0) f = (double)__fdiv_rn(num, p.env[VAR_TP][i]);
  else if (tp_mode == 2) f = 0.0;
  else {
    double omega = 5;    // _wave_frequency (:909-916)
    if (ws > 0) omega = (double)__fdiv_rn((float)(0.877 * 9.81), __fmul_rn(OF(1.17), ws));
    const double T = __ddiv_rn(2 * kPi, omega);
    f = tp_mode == 3 ? (double)__fdiv_rn(num, (float)T) : __ddiv_rn((double)num, T);
  }
  return f < 0 ? 0.0 : f;
}

// 1 - exp(-oil_wave_entrainment_rate_li2017 * vertical_mixing:timestep)
__device__ __forceinline__ double oil_entrainment_probability(double rho, double visc, float hs, double wbf,
                                                              const OilArgs &a) {
  const double g = 9.81;
  const double delta_rho = __dsub_rn(a.rho_w, rho);
  const double d_o = __dmul_rn(4.0, sqrt(__ddiv_rn(a.sigma_ow, __dmul_rn(delta_rho, g))));
  const double we = __ddiv_rn(__dmul_rn(__dmul_rn(__dmul_rn(a.rho_w, g), (double)hs), d_o), a.sigma_ow);
  const double oh = __ddiv_rn(__dmul_rn(visc, rho), sqrt(__dmul_rn(__dmul_rn(rho, a.sigma_ow), d_o)));
  const double rate = __dmul_rn(__dmul_rn(__dmul_rn(4.604e-10, pow(we, 1.805)), pow(oh, -1.023)), wbf);
  return __dsub_rn(1.0, exp(__dmul_rn(-rate, a.dt_mix_cfg)));
}

// per-particle state of the oil physics during the sub-steps of one vertical_mixing call
struct OilLane {
  float d, d_if, nyw;
  double kw, kw2, prob, W, mean_zb;
  bool dirty;

  __device__ __forceinline__ void init(const PView &p, long long i, const OilArgs &a) {
    d = p.aux[OIL_DIAMETER][i];
    d_if = p.aux[OIL_DIAMETER_IF_ENTRAINED][i];
    const double rho = (double)p.aux[OIL_DENSITY][i], visc = (double)p.aux[OIL_VISCOSITY][i];
    float Tk = p.env[VAR_TEMP][i];
    if (a.to_kelvin && Tk < 100.f) Tk = __fadd_rn(Tk, OF(273.15));
    const float T0 = __fsub_rn(Tk, OF(273.15)), S0 = p.env[VAR_SALT][i];
    const float rho_water = oil_sea_water_density_f32(T0, S0);
    nyw = __fdiv_rn(oil_water_viscosity_f32(T0, S0), rho_water);
    const double one_m = __dsub_rn(1.0, __ddiv_rn(rho, (double)rho_water));
    kw = __ddiv_rn(__dmul_rn(2 * 9.81, one_m), (double)__fmul_rn(9.f, nyw));
    kw2 = sqrt(__ddiv_rn(__dmul_rn(16 * 9.81, one_m), 3.0));
    const float ws = speed_f32(p.env[VAR_XWIND][i], p.env[VAR_YWIND][i]);
    prob = oil_entrainment_probability(rho, visc, oil_hs(p, i, a.hs_mode, ws), oil_breaking_fraction(p, i, a.tp_mode, ws), a);
    mean_zb = a.stat[OIL_STAT_MEAN_ZB];
    dirty = true;
    W = 0;
  }
  // update_terminal_velocity: Stokes law, the high-Reynolds form above Re = 50 (Tkalich et al. 2002)
  __device__ __forceinline__ double terminal_velocity() {
    if (dirty) {
      dirty = false;
      const float h = __fmul_rn(d, 0.5f);
      W = __dmul_rn(kw, (double)__fmul_rn(h, h));
      const double Re = __ddiv_rn(__dmul_rn((double)d, W), (double)nyw);
      if (Re > 50) W = __dmul_rn(kw2, (double)sqrtf(h));
    }
    return W;
  }
  // surface_wave_mixing for an element at the surface (z >= 0 after surface_stick); returns the new z
  __device__ __forceinline__ double surface_wave_mixing(double z, const PView &p, long long i, int it, const OilArgs &a,
                                                        unsigned long long seed, unsigned long long step) {
    double ue, ui;
    if (a.rng_mode == 1) {
      ue = a.u_ent[(size_t)it * p.n + i];
      ui = a.u_int[(size_t)it * p.n + i];
    } else {
      if (!(prob > 0)) return z;   // calm (wind <= 5 m/s: no breaking waves): nothing can be entrained, no number is needed
      const double2 u = rng_uniform2(rng_block(seed, p.id[i], step, RNG_OFF_OIL_ENTRAIN + 4ull * (unsigned long long)it));   // one Philox block (4 x 32 bit) per sub-step
      ue = u.x; ui = u.y;
    }
    if (ue < prob) {   // np.random.uniform(0, np.mean(zb)) = 0 + (mean - 0) * u  (Delvigne and Sweeney 1988)
      z = -__dmul_rn(mean_zb, ui);
      if (!a.keep_diameter && d != d_if) { d = d_if; dirty = true; }
    }
    return z;
  }
};

// median droplet diameter (volume distribution) of one element for the two spectra
__device__ __forceinline__ double oil_dv50_element(const PView &p, long long i, const OilArgs &a, float H) {
  const double g = 9.81;
  const double rho = (double)p.aux[OIL_DENSITY][i], visc = (double)p.aux[OIL_VISCOSITY][i];
  if (a.droplets == 1) {   // Johansen et al. (2015), eqs. 7a, 7b (:1136-1154)
    const float film = p.aux[OIL_FILM][i];
    const double rf = __dmul_rn(rho, (double)film);
    const double re = __ddiv_rn(__dmul_rn(rf, (double)sqrtf(__fmul_rn(OF(g), H))), __dmul_rn(visc, rho));
    const double we = __ddiv_rn(__dmul_rn(__dmul_rn(rf, g), (double)H), a.sigma_ow);
    const double A = 2.251, B = 2.251 * 0.027;
    const double dN = __dadd_rn(__dmul_rn((double)__fmul_rn(OF(A), film), pow(we, -0.6)),
                                __dmul_rn((double)__fmul_rn(OF(B), film), pow(re, -0.6)));
    const double Sd = 2.302585092994046 * 0.4;   // np.log(10) * 0.4
    return exp(__dadd_rn(log(dN), __dmul_rn(3.0, __dmul_rn(Sd, Sd))));
  }
  // Li et al. (2017) (:1083-1097)
  const double delta_rho = __dsub_rn(a.rho_w, rho);
  const double d_o = __dmul_rn(4.0, sqrt(__ddiv_rn(a.sigma_ow, __dmul_rn(delta_rho, g))));
  const double we = __ddiv_rn(__dmul_rn(__dmul_rn(__dmul_rn(a.rho_w, g), (double)H), d_o), a.sigma_ow);
  const double oh = __dmul_rn(__dmul_rn(visc, rho), pow(__dmul_rn(__dmul_rn(rho, a.sigma_ow), d_o), -0.5));
  return __dmul_rn(__dmul_rn(__dmul_rn(d_o, 1.791), pow(__dadd_rn(1.0, __dmul_rn(10.0, oh)), 0.460)), pow(we, -0.518));
}

__device__ __forceinline__ double oil_diameter_of(int k) {   // np.linspace: start + k*step, last point = stop
  const double step = (3e-3 - 1e-6) / (double)(OIL_NSPEC - 1);
  return k == OIL_NSPEC - 1 ? 3e-3 : __dadd_rn(__dmul_rn((double)k, step), 1e-6);
}
__device__ __forceinline__ double oil_spectrum_at(int k, double log_dv50) {
  const double Sd = 2.302585092994046 * 0.4;
  const double d = oil_diameter_of(k);
  const double q = __dsub_rn(log(d), log_dv50);
  return __ddiv_rn(exp(__ddiv_rn(-__dmul_rn(q, q), __dmul_rn(2.0, __dmul_rn(Sd, Sd)))),
                   __dmul_rn(__dmul_rn(d, Sd), sqrt(2 * kPi)));
}

constexpr int OIL_GUIDE = 65536;
__device__ __forceinline__ int oil_search_right(const double *__restrict__ cdf, double total, double u, int lo, int hi) {
  while (lo < hi) {   // number of entries with cdf/total <= u in [lo, hi)
    const int mid = (lo + hi) >> 1;
    if (__ddiv_rn(cdf[mid], total) <= u) lo = mid + 1; else hi = mid;
  }
  return lo;
}

#ifdef ODR_TU_MIX
// ---------------------------------------------------------------- prepare_vertical_mixing
#ifndef ODR_OIL_HOST   // tests/oil_host.cpp compiles the per-element arithmetic above for the CPU
// Per-element median droplet diameter dV_50 of the spectrum (its MEAN over the elements parameterises the one
// spectrum all elements draw from, :1099-1101 / :1156-1158) and zb = 1.5 Hs (:1047): block sums -> part[2][nblocks],
// summed in a fixed order by k_oil_stats_final (deterministic, unlike floating-point atomics).
__global__ __launch_bounds__(BLOCK) void k_oil_stats(PView p, OilArgs a, double *__restrict__ part) {
  __shared__ double sh[2][BLOCK / 64];
  const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  double dv = 0, zb = 0;
  if (i < p.n) {
    const float H = oil_hs(p, i, a.hs_mode, speed_f32(p.env[VAR_XWIND][i], p.env[VAR_YWIND][i]));
    zb = (double)__fmul_rn(1.5f, H);
    dv = oil_dv50_element(p, i, a, H);
  }
  for (int o = 32; o > 0; o >>= 1) { dv += __shfl_down(dv, o); zb += __shfl_down(zb, o); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = dv; sh[1][w] = zb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0, s1 = 0;
    for (int k = 0; k < BLOCK / 64; ++k) { s0 += sh[0][k]; s1 += sh[1][k]; }
    part[blockIdx.x] = s0;
    part[gridDim.x + blockIdx.x] = s1;
  }
}

__global__ __launch_bounds__(BLOCK) void k_oil_stats_final(const double *__restrict__ part, int nb, long long n,
                                                           double *__restrict__ stat) {
  __shared__ double sh[2][BLOCK];
  double s0 = 0, s1 = 0;
  for (int k = threadIdx.x; k < nb; k += BLOCK) { s0 += part[k]; s1 += part[nb + k]; }
  sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1;
  __syncthreads();
  for (int o = BLOCK / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    stat[OIL_STAT_DV50] = sh[0][0] / (double)n;                      // np.mean(dV_50), float64
    stat[OIL_STAT_MEAN_ZB] = (double)(float)(sh[1][0] / (double)n);  // np.mean of a float32 array is float32
  }
}

// The droplet spectrum: a log-normal number density around dV_50 (sd 0.4 in log10 units) on 1e6 diameters between
// 1 micron and 3 mm (:1103-1108), and the cumulative sum np.random.choice searches (numpy/random/mtrand.pyx:
// cdf = p.cumsum(); cdf /= cdf[-1]; idx = cdf.searchsorted(uniform, side='right')).  Three deterministic passes:
// chunk sums, scan of the 245 chunk sums, chunk-local scan + offset.  The normalisation by the total is applied
// by the reader (k_oil_choice), so that no fourth pass is needed.
__global__ __launch_bounds__(BLOCK) void k_oil_spectrum_sums(const double *__restrict__ stat, double *__restrict__ chunk) {
  __shared__ double sh[BLOCK];
  const double ldv = log(stat[OIL_STAT_DV50]);
  const int base = blockIdx.x * OIL_SPEC_CHUNK + threadIdx.x * OIL_SPEC_PER_THREAD;
  double s = 0;
  for (int j = 0; j < OIL_SPEC_PER_THREAD; ++j)
    if (base + j < OIL_NSPEC) s += oil_spectrum_at(base + j, ldv);
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {   // fixed order
    double t = 0;
    for (int k = 0; k < BLOCK; ++k) t += sh[k];
    chunk[blockIdx.x] = t;
  }
}

__global__ void k_oil_spectrum_offsets(double *__restrict__ chunk, double *__restrict__ stat) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {   // exclusive scan of the chunk sums (245 values)
    double run = 0;
    for (int k = 0; k < OIL_SPEC_BLOCKS; ++k) { const double c = chunk[k]; chunk[k] = run; run += c; }
    stat[OIL_STAT_CDF_TOTAL] = run;
  }
}

__global__ __launch_bounds__(BLOCK) void k_oil_spectrum_scan(const double *__restrict__ stat, const double *__restrict__ chunk,
                                                             double *__restrict__ cdf) {
  __shared__ double sh[BLOCK];
  const double ldv = log(stat[OIL_STAT_DV50]);
  const int base = blockIdx.x * OIL_SPEC_CHUNK + threadIdx.x * OIL_SPEC_PER_THREAD;
  double v[OIL_SPEC_PER_THREAD], s = 0;
#pragma unroll
  for (int j = 0; j < OIL_SPEC_PER_THREAD; ++j) {
    v[j] = base + j < OIL_NSPEC ? oil_spectrum_at(base + j, ldv) : 0.0;
    s += v[j];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  double off = chunk[blockIdx.x];
  for (int k = 0; k < (int)threadIdx.x; ++k) off += sh[k];   // fixed order, 255 additions at most
#pragma unroll
  for (int j = 0; j < OIL_SPEC_PER_THREAD; ++j) {
    off += v[j];
    if (base + j < OIL_NSPEC) cdf[base + j] = off;
  }
}

// droplet_diameter_if_entrained = np.random.choice(diameters, n, p=pdf): one uniform per element, first grid point
// whose normalised cumulative sum exceeds it; stored as the float32 the element property would hold.
// A plain binary search is 20 dependent probes into the 8 MB table (0.61 ms for 10 M elements); the guide table
// G[j] = searchsorted(cdf, j / OIL_GUIDE, 'right') brackets the answer for every uniform of bucket j
// (G[j] <= index(u) <= G[j+1] for j/OIL_GUIDE <= u < (j+1)/OIL_GUIDE, both products exact), so that the search
// covers ~15 neighbouring entries instead of 1e6 -- the same predicate, hence the same index.
__global__ __launch_bounds__(BLOCK) void k_oil_guide(const double *__restrict__ cdf, int *__restrict__ guide) {
  const int j = blockIdx.x * BLOCK + threadIdx.x;
  if (j > OIL_GUIDE) return;
  guide[j] = oil_search_right(cdf, cdf[OIL_NSPEC - 1], (double)j / (double)OIL_GUIDE, 0, OIL_NSPEC);
}
__global__ __launch_bounds__(BLOCK) void k_oil_choice(PView p, const double *__restrict__ cdf, const int *__restrict__ guide,
                                                      int rng_mode, const double *__restrict__ huni,
                                                      unsigned long long seed, unsigned long long step) {
  const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  double u;
  if (rng_mode == 1) u = huni[i];
  else {
    const uint4 b = rng_block(seed, p.id[i], step, RNG_OFF_OIL_DIAMETER);
    u = rng_u53(b.x, b.y);
  }
  int lo = 0, hi = OIL_NSPEC;
  if (u >= 0.0 && u < 1.0) {
    const int j = (int)(u * (double)OIL_GUIDE);
    lo = guide[j];
    hi = guide[j + 1];
  }
  lo = oil_search_right(cdf, cdf[OIL_NSPEC - 1], u, lo, hi);
  if (lo > OIL_NSPEC - 1) lo = OIL_NSPEC - 1;
  p.aux[OIL_DIAMETER_IF_ENTRAINED][i] = (float)oil_diameter_of(lo);
}

#endif  // ODR_OIL_HOST
#undef OF
#endif  // ODR_TU_MIX
}  // namespace odr
