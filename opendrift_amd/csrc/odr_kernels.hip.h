// HIP kernels of the particle-advection hot path (gfx950 / MI355X).
//
// Layout: structure-of-arrays particle state resident in HBM, one thread per
// particle, 256-thread workgroups (4 waves of 64), every state array read and
// written fully coalesced (8 B or 4 B per lane).  Field blocks are read through
// the vector L1 / per-XCD L2 / Infinity Cache; the 2x2(x2z x2t) footprints of a
// wave's particles land on few cache lines when the particles are spatially
// coherent.  No MFMA: the path is gather + f64 scalar math.
#pragma once
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include "odr_field.hip.h"

namespace odr {

#ifndef ODR_BLOCK
#define ODR_BLOCK 256
#endif
constexpr int BLOCK = ODR_BLOCK;  // threads per workgroup (A/B builds may override)
#ifndef ODR_POLAR_WAVES
#define ODR_POLAR_WAVES 3   // the environment-only kernel of a projected reader (latency-bound gathers: occupancy first)
#endif
#ifndef ODR_POLAR_STEP_WAVES
// the advection kernels of a projected reader.  C4, round 1: 2 waves 1.31 ms, 3 waves (168 VGPRs, 52 B scratch) 1.06 ms,
// 4 waves (280 B scratch) 2.06 ms; round 2 (stage noise, series geodesic: 128-160 B scratch at 3 waves): 3 waves 1.045 ms,
// 2 waves (no scratch) 0.998 ms  (profiles/r02_ab_variants.txt)
#define ODR_POLAR_STEP_WAVES 2
#endif
// (PROJ_EXT -- tmerc / laea / oblique stere / rotated pole inlined -- needs more than 256 registers: one wave per SIMD rather than scratch memory)
#define ODR_STEP_WAVES(PROJ) ((PROJ) == PROJ_LATLONG ? ODR_LATLONG_WAVES : ((PROJ) == PROJ_EXT ? 1 : ODR_POLAR_STEP_WAVES))
// minimum waves per SIMD requested for the projected-reader instantiations (their stereographic forward /
// rotation code otherwise takes ~185 VGPRs = 2 waves per SIMD)
#ifndef ODR_LATLONG_WAVES
#define ODR_LATLONG_WAVES 1
#endif
#define ODR_WAVES(PROJ) ((PROJ) == PROJ_LATLONG ? ODR_LATLONG_WAVES : ((PROJ) == PROJ_EXT ? 1 : ODR_POLAR_WAVES))
#ifndef ODR_MIX_WAVES
#define ODR_MIX_WAVES 3   // the step kernel with the mixing inside: 167 VGPRs with the kept (u,v) records (3 waves per SIMD); held at 128 it spills
#endif

// XCD-aware block order.  Workgroups are dispatched round-robin over the 8 XCDs (each with its own L2), so with
// the natural order the eight L2s all stream the whole (spatially sorted) particle range and every field tile is
// fetched by all of them.  pid() gives XCD x the x-th contiguous eighth of the range instead: the particles of one
// region -- and the field records they gather -- stay in one L2.
__device__ __forceinline__ long long pid() {
#ifdef ODR_NO_XCD_REMAP
  return (long long)blockIdx.x * BLOCK + threadIdx.x;
#else
  const unsigned nb = gridDim.x, b = blockIdx.x, per = nb >> 3, rem = nb & 7u, x = b & 7u, j = b >> 3;
  const unsigned lb = x * per + (x < rem ? x : rem) + j;
  return (long long)lb * BLOCK + threadIdx.x;
#endif
}

struct PView {  // device pointers of the active set
  long long n;
  double *lon, *lat, *z, *plon, *plat;  // plon/plat: update_previous_state (basemodel/__init__.py:642-668)
  double *slon, *slat;                  // position of the last environment sample (profiles)
  int *id, *status, *moving;
  float *wdf, *cdf, *tv, *age;
  float *aux[9];  // model-specific float32 element properties (Leeway: LeewayObj, leeway.py:50-131)
  float *env[NVAR];
  int ice;        // odr_set_element_factor: 0 scalar factors; 1 (1 - k_ice), 2 factor_stokes, 3 k_ice per element
  int pad;
  const int *rank;  // position among the present elements in ascending ID (ensemble member = rank % members), or null
};

// OpenOil.advect_oil in sea ice (openoil.py:1182-1201; Nordam et al. 2019, Arneborg 2017), float32 like NumPy on the
// float32 environment array: k_ice = (A - 0.3) / (0.8 - 0.3), 0 below 30 %, 1 above 80 %;
// factor_stokes = (0.7 - A) / 0.7, 0 above 70 %
__device__ __forceinline__ float ice_k(float A) {
  float k = __fdiv_rn(__fsub_rn(A, 0.3f), 0.5f);
  if (A < 0.3f) k = 0.f;
  if (A > 0.8f) k = 1.f;
  return k;
}
__device__ __forceinline__ float ice_stokes_factor(float A) {
  float f = __fdiv_rn(__fsub_rn(0.7f, A), 0.7f);
  if (A > 0.7f) f = 0.f;
  return f;
}
// factor of advect_ocean_current / advect_wind for element i: the caller's scalar, or 1 - k_ice (float32)
__device__ __forceinline__ float current_factor(const PView &p, long long i, float factor) {
  return p.ice == 1 ? __fsub_rn(1.0f, ice_k(p.env[VAR_ICE_A][i])) : factor;
}

// ------------------------------------------------------------ float32 rounding points
// np.degrees(np.arctan2(x_vel, y_vel)) in float32 (physics_methods.py:629,
// basemodel/__init__.py:4645): correctly rounded float32 arctan2,
// times float32(180)/float32(pi) exactly as NumPy's float32 degrees loop does.
__device__ __forceinline__ float azimuth_f32(float xv, float yv) {
  float a = (float)atan2((double)xv, (double)yv);
  return __fmul_rn(a, 180.0f / 3.14159274101257324f);
}
__device__ __forceinline__ float speed_f32(float xv, float yv) {
  return sqrtf(__fadd_rn(__fmul_rn(xv, xv), __fmul_rn(yv, yv)));  // IEEE sqrt (the __fsqrt_rn alias is native_sqrt)
}

// Sine and cosine of the float32 azimuth the reference hands to geod.fwd.  With theta = atan2(x, y)
// in float64, sin / cos of theta are x/h and y/h (no trigonometric call), and the two float32
// roundings (arctan2, degrees) move the angle by delta = az_f32 * pi/180 - theta, |delta| < 4e-7:
//   sin(theta + delta) = sin theta (1 - delta^2/2) + cos theta delta     (delta^3/6 < 1e-20).
// Same value as sincosd(az_f32) to float64 round-off, ~25 instructions instead of ~75.
__device__ __forceinline__ void azimuth_sincos_f32(float xv, float yv, double &salp, double &calp) {
#pragma clang fp contract(fast)
  const double x = (double)xv, y = (double)yv;
  const double h2 = x * x + y * y;
  if (!(h2 > 0) || h2 > 1.7e308) {  // calm (azimuth 0 or 180 exactly), NaN or infinite velocities: library semantics
    const double th = atan2(x, y);
    const double azd = (double)__fmul_rn((float)th, 180.0f / 3.14159274101257324f);
    if (th != th) { salp = calp = th; return; }
    if (h2 > 0) { sincosd(azd, salp, calp); return; }
    salp = 0.0;
    calp = azd == 0.0 ? 1.0 : -1.0;
    return;
  }
  const double theta = atan2_fin(x, y);
  const float azf = __fmul_rn((float)theta, 180.0f / 3.14159274101257324f);  // == azimuth_f32(xv, yv)
  const double az = (double)azf;
  const double delta = fma(az, kDeg, -theta) + az * kDegLo;
  const double rh = fast_rsqrt(h2);
  const double st = x * rh, ct = y * rh, c2 = 1 - 0.5 * delta * delta;
  salp = fma(ct, delta, st * c2);
  calp = fma(-st, delta, ct * c2);
}

// Start point of the (up to five) geodesics of one particle-step and one step along it.  Default: Legendre series
// about the start point (odr_geodesic.hip.h), the full Karney solution only for steps it does not cover.
// -DODR_FULL_GEODESIC: every step through the full solution (A/B measurements, strict build).
#ifdef ODR_FULL_GEODESIC
typedef GeodOrigin GeodStart;
__device__ __forceinline__ GeodStart geod_start(double lat, double lon) { return geod_origin(lat, lon); }
__device__ __forceinline__ void geod_step(const GeodStart &o, double salp, double calp, double s12, double &lat2,
                                          double &lon2) {
  geod_direct_sc(o, salp, calp, s12, lat2, lon2);
}
#else
typedef GeodLocal GeodStart;
__device__ __forceinline__ GeodStart geod_start(double lat, double lon) { return geod_local_origin(lat, lon); }
__device__ __forceinline__ void geod_step(const GeodStart &o, double salp, double calp, double s12, double &lat2,
                                          double &lon2) {
  geod_local_move(o, s12 * salp, s12 * calp, lat2, lon2);
}
#endif

// The start-point coefficients of the series geodesic (12 doubles = 24 registers) live from the first stage position to the
// final move, across the three stage samples -- the register peak of k_step_grid.  PARKED: they wait in the workgroup's LDS
// (one column per thread, [coefficient][thread]: conflict-free 8-byte accesses) and are read back where a move needs them;
// the kernel then fits 96 registers = 5 waves per SIMD without scratch memory (round 3; 111 registers = 4 waves before,
// 80 B of scratch when merely capped at 96).
constexpr int GEOD_PARK = 15;   // (round 5: + the start point itself, slots 12 / 13, and the drift factor of the final move, slot 14)
#ifndef ODR_FULL_GEODESIC
typedef volatile __attribute__((address_space(3))) double lds_f64;   // explicit LDS pointer: ds_read / ds_write, not flat accesses
__device__ __forceinline__ void geod_park(const GeodLocal &o, lds_f64 *slot) {
  slot[0 * BLOCK] = o.iN; slot[1 * BLOCK] = o.qs; slot[2 * BLOCK] = o.kphi; slot[3 * BLOCK] = o.klam;
  slot[4 * BLOCK] = o.t; slot[5 * BLOCK] = o.a20; slot[6 * BLOCK] = o.a30; slot[7 * BLOCK] = o.a12;
  slot[8 * BLOCK] = o.a40; slot[9 * BLOCK] = o.a22; slot[10 * BLOCK] = o.b21; slot[11 * BLOCK] = o.b31;
  slot[12 * BLOCK] = o.lat1; slot[13 * BLOCK] = o.lon1n;
}
__device__ __forceinline__ GeodLocal geod_unpark(double lat1, double lon1n, lds_f64 *slot) {
  GeodLocal o;
  o.lat1 = lat1; o.lon1n = lon1n;
  o.iN = slot[0 * BLOCK]; o.qs = slot[1 * BLOCK]; o.kphi = slot[2 * BLOCK]; o.klam = slot[3 * BLOCK];
  o.t = slot[4 * BLOCK]; o.a20 = slot[5 * BLOCK]; o.a30 = slot[6 * BLOCK]; o.a12 = slot[7 * BLOCK];
  o.a40 = slot[8 * BLOCK]; o.a22 = slot[9 * BLOCK]; o.b21 = slot[10 * BLOCK]; o.b31 = slot[11 * BLOCK];
  return o;
}
// geod_local_move straight from the parked coefficients (odr_geodesic.hip.h for the series): they are read in three groups,
// each right in front of the terms that use it, so that at most five of the twelve are in registers at a time (geod_unpark
// reads all twelve = 24 registers in front of the move -- the transient that stood between k_step_grid and 96 registers).
// Same operations in the same order as geod_local_move: same bits.
__device__ __forceinline__ void geod_local_move_parked(lds_f64 *slot, double x, double y, double &lat2, double &lon2) {
#pragma clang fp contract(fast)
  const double iN = slot[0 * BLOCK], qs = slot[1 * BLOCK];
  const double u = y * iN, v = x * iN;
  const double u2 = u * u, v2 = v * v;
  if (!((u2 + v2) * (qs * qs) <= kGeodLocalQ * kGeodLocalQ)) {   // also NaN steps
    const double lat1 = slot[12 * BLOCK], lon1n = slot[13 * BLOCK];
    if (u2 + v2 == u2 + v2 && lat1 == lat1) { const GeodLL r = geod_local_far(lat1, lon1n, x, y); lat2 = r.lat; lon2 = r.lon; return; }
    lat2 = lon2 = __builtin_nan("");
    return;
  }
  const double a40 = slot[8 * BLOCK], a30 = slot[6 * BLOCK], a20 = slot[5 * BLOCK];
  const double pu = fma(fma(a40, u, a30), u, a20);                               // a20 + a30 u + a40 u^2
  __builtin_amdgcn_sched_barrier(0);
  const double t = slot[4 * BLOCK], a12 = slot[7 * BLOCK], a22 = slot[9 * BLOCK];
  const double a02 = -0.5 * t, a04 = -0.25 * t * a12, b03 = (-1.0 / 3) * t * t;
  const double pv = fma(a04, v2, fma(fma(a22, u, a12), u, a02));                 // a02 + a12 u + a22 u^2 + a04 v^2
  const double p = fma(pv, v2, fma(pu, u2, u));
  __builtin_amdgcn_sched_barrier(0);
  const double b21 = slot[10 * BLOCK], b31 = slot[11 * BLOCK];
  const double b13 = -t * b21;
  const double lu = fma(fma(fma(b31, u, b21), u, t), u, 1.0);                    // 1 + b11 u + b21 u^2 + b31 u^3
  const double l = v * fma(fma(b13, u, b03), v2, lu);
  __builtin_amdgcn_sched_barrier(0);
  const double kphi = slot[2 * BLOCK], klam = slot[3 * BLOCK], lat1 = slot[12 * BLOCK], lon1n = slot[13 * BLOCK];
  lat2 = fma(kphi, p, lat1);
  lon2 = ang_normalize(fma(klam, l, lon1n));
}
#endif

// update_positions (basemodel/__init__.py:4631-4657), float32 velocities
__device__ __forceinline__ void move_f32_from(const GeodStart &o, double &lon, double &lat, float u,
                                              float v, int moving, double dt) {
  double salp, calp;
  azimuth_sincos_f32(u, v, salp, calp);
  double vel = (double)speed_f32(u, v) * (double)moving;  // f32 * int32 array -> float64
  geod_step(o, salp, calp, vel * dt, lat, lon);
}
__device__ __forceinline__ void move_f32(double &lon, double &lat, float u, float v, int moving,
                                         double dt) {
  GeodStart o = geod_start(lat, lon);
  move_f32_from(o, lon, lat, u, v, moving, dt);
}
// float64 velocities (advect_wind / stokes_drift / horizontal_diffusion callers):
//   azimuth = degrees(arctan2(u, v)), distance = sqrt(u^2 + v^2) * moving * dt, geod.fwd(lon, lat, azimuth, distance)  (float64).
// The series geodesic takes the step by its east / north components distance * sin / cos(azimuth) = u dt, v dt (moving is 0 or
// 1): formed directly -- no arctan2, no square root, no division.  The reference's own azimuth carries a float64 rounding of up
// to 180 deg (3e-16 rad of direction = 3e-16 of the step across it); the direct components are inside that.
// ODR_FULL_GEODESIC builds and non-finite velocities keep azimuth and distance (library semantics).
__device__ __forceinline__ void move_f64_polar(double u, double v, int moving, double dt, double &salp, double &calp, double &s12) {
  const double h2 = fma(u, u, v * v);
  salp = 0.0; calp = 1.0;
  if (h2 > 0 && h2 < 1.7e308) {
    const double rh = fast_rsqrt(h2);
    salp = u * rh;
    calp = v * rh;
  } else if (!(h2 == 0)) {  // NaN / infinite velocities
    double az = atan2(u, v) * (180.0 / kPi);
    az = ang_normalize(az);
    sincosd(ang_round(az), salp, calp);
  }
  s12 = sqrt(__dadd_rn(__dmul_rn(u, u), __dmul_rn(v, v))) * (double)moving * dt;
}
__device__ __forceinline__ void move_f64(double &lon, double &lat, double u, double v, int moving,
                                         double dt) {
  GeodStart o = geod_start(lat, lon);
  double lo, la;
#ifdef ODR_FULL_GEODESIC
  double salp, calp, s12;
  move_f64_polar(u, v, moving, dt, salp, calp, s12);
  geod_step(o, salp, calp, s12, la, lo);
#else
  const double h2 = fma(u, u, v * v);
  if (h2 < 1.7e308) {
    const double hd = (double)moving * dt;
    geod_local_move(o, u * hd, v * hd, la, lo);
  } else {
    double salp, calp, s12;
    move_f64_polar(u, v, moving, dt, salp, calp, s12);
    geod_step(o, salp, calp, s12, la, lo);
  }
#endif
  lon = lo;
  lat = la;
}
// The moves of one element that follow one another in one launch (k_movers): the start point of a move is the end point of
// the previous one; after a series move its coefficients come from the previous start latitude (geod_local_origin_next).
struct MoveChain {
  double lat0, sphi, cphi;
  bool next;        // the previous move was a series move from lat0 (sphi / cphi = its sine / cosine)
};
__device__ __forceinline__ void move_f64_chain(MoveChain &mc, double &lon, double &lat, double u, double v, int moving, double dt) {
#ifdef ODR_FULL_GEODESIC
  move_f64(lon, lat, u, v, moving, dt);
#else
  const GeodLocal o = mc.next ? geod_local_origin_next(mc.lat0, lat, lon, mc.sphi, mc.cphi) : geod_local_origin_sc(lat, lon, mc.sphi, mc.cphi);
  mc.lat0 = lat;
  double lo, la;
  const double h2 = fma(u, u, v * v);
  if (h2 < 1.7e308) {
    const double hd = (double)moving * dt;
    mc.next = geod_local_move_ok(o, u * hd, v * hd, la, lo);
  } else {
    double salp, calp, s12;
    move_f64_polar(u, v, moving, dt, salp, calp, s12);
    mc.next = geod_local_move_ok(o, s12 * salp, s12 * calp, la, lo);
  }
  lon = lo;
  lat = la;
#endif
}

// the same for float32 velocities (Leeway.update: the leeway move, then the current's)
__device__ __forceinline__ void move_f32_chain(MoveChain &mc, double &lon, double &lat, float u, float v, int moving, double dt) {
#ifdef ODR_FULL_GEODESIC
  move_f32(lon, lat, u, v, moving, dt);
#else
  const GeodLocal o = mc.next ? geod_local_origin_next(mc.lat0, lat, lon, mc.sphi, mc.cphi) : geod_local_origin_sc(lat, lon, mc.sphi, mc.cphi);
  mc.lat0 = lat;
  double salp, calp;
  azimuth_sincos_f32(u, v, salp, calp);
  const double s12 = (double)speed_f32(u, v) * (double)moving * dt;
  double lo, la;
  mc.next = geod_local_move_ok(o, s12 * salp, s12 * calp, la, lo);
  lon = lo;
  lat = la;
#endif
}

// RK sub-stage position: geod.fwd(lon, lat, az, speed*dt*.5) with az and dist in float32 (physics_methods.py:629-635);
// the start point is shared by all stages of a particle.
// Stage math (odr_ctx_set_stage_math):
//   ODR_STAGE_EXACT (SM = 0): the reference's float32 rounding points inside a stage are reproduced -- azimuth = float32
//     arctan2 in degrees, distance = float32 speed * dt * .5 -- trajectories stay within 1e-10 deg per step of the oracle.
//   ODR_STAGE_FAST  (SM = 1): the stage step taken directly along (u, v) dt/2 -- no arctan2, no square root.  A stage
//     position only matters through the velocity sampled there; the two skipped roundings move it by < 1e-4 m, the sampled
//     float32 velocity then differs in its last bits now and then: <= 2e-9 deg per step against the oracle, 500 times inside
//     the 1e-6 deg of the task and below the 1e-8 deg the oracle itself differs from the reference by.
//     The main-loop sample, the RK combination and the final move keep every rounding point in both modes.
template <int SM>
__device__ __forceinline__ void stage_pos(const GeodStart &o, float u, float v, float dtf,
                                          double &lon2, double &lat2) {
#if defined(ODR_FULL_GEODESIC)
  constexpr bool direct = false;
#else
  constexpr bool direct = SM == 1;
#endif
  if constexpr (!direct) {
    double salp, calp;
    azimuth_sincos_f32(u, v, salp, calp);
    float dist = __fmul_rn(__fmul_rn(speed_f32(u, v), dtf), 0.5f);
    geod_step(o, salp, calp, (double)dist, lat2, lon2);
  } else {
#ifndef ODR_FULL_GEODESIC
    const double hd = 0.5 * (double)dtf;
    geod_local_move(o, (double)u * hd, (double)v * hd, lat2, lon2);
#endif
  }
}
#ifndef ODR_FULL_GEODESIC
// ODR_STAGE_FAST with the coefficients parked in LDS
__device__ __forceinline__ void stage_pos_parked(lds_f64 *slot, float u, float v, float dtf, double &lon2, double &lat2) {
  const double hd = 0.5 * (double)dtf;
  geod_local_move_parked(slot, (double)u * hd, (double)v * hd, lat2, lon2);
}
#endif
// the same with the mode as a (wave-uniform) run-time value: kernels that serve any reader mix
__device__ __forceinline__ void stage_pos_rt(int sm, const GeodStart &o, float u, float v, float dtf, double &lon2, double &lat2) {
  if (sm == 1) stage_pos<1>(o, u, v, dtf, lon2, lat2);
  else stage_pos<0>(o, u, v, dtf, lon2, lat2);
}

// --------------------------------------------------------------- random numbers
// Philox4x32-10 (rocRAND device API), counter = (particle ID, step, stream): results do
// not depend on how particles are sharded over GPUs or ordered in memory.
constexpr unsigned long long RNG_STEP_STRIDE = 8192;
constexpr unsigned long long RNG_OFF_VMIX = 0, RNG_OFF_HDIFF = 4096, RNG_OFF_NOISE = 4352;

// The streams are rocRAND's Philox4x32-10 streams rocrand_init(seed, subsequence = ID, offset = step * RNG_STEP_STRIDE + off)
// (rounds 1-5 drew through its state object): the counter holds offset / 4 in words 0-1 and the subsequence in words 2-3, the
// key is the seed.  Every consumer needs ONE 128-bit block of its stream (off is a multiple of 4), which rng_block evaluates
// directly -- the state object evaluates a block in rocrand_init and another one ahead in every rocrand4() -- and turns into
// rocRAND's own double-precision distributions: rng_uniform2 = rocrand_uniform_double2 bit for bit (53-bit fractions),
// rng_normal2 = rocrand_normal_double2 (Box-Muller on the same two 53-bit fractions) to float64 round-off, with the logarithm,
// the square root and sincospi taken on the ranges Box-Muller feeds them (~75 instead of ~140 instructions; C5: noise and jibing
// of the Leeway launch 0.111 + 0.053 ms of 0.73, profiles/r06_ab_variants.txt).  -DODR_RNG_ROCRAND: the library's calls (A/B).
template <int ROUNDS>
__device__ __forceinline__ uint4 philox4x32(uint4 c, unsigned k0, unsigned k1);
__device__ __forceinline__ uint4 rng_block(unsigned long long seed, int id, unsigned long long step, unsigned long long off) {
#ifdef ODR_RNG_ROCRAND
  rocrand_state_philox4x32_10 st;
  rocrand_init(seed, (unsigned long long)(unsigned)id, step * RNG_STEP_STRIDE + off, &st);
  return rocrand4(&st);
#else
  const unsigned long long c = (step * RNG_STEP_STRIDE + off) >> 2;
  return philox4x32<10>(make_uint4((unsigned)c, (unsigned)(c >> 32), (unsigned)id, 0u), (unsigned)seed, (unsigned)(seed >> 32));
#endif
}
__device__ __forceinline__ double rng_u53(unsigned lo, unsigned hi) {   // rocrand: uniform_distribution_double(v1, v2), in (0, 1]
  const unsigned long long v = (unsigned long long)lo | ((unsigned long long)(hi >> 11) << 32);
  return __dadd_rn(0x1p-53, __dmul_rn((double)v, 0x1p-53));
}
__device__ __forceinline__ double2 rng_uniform2(uint4 b) {
  return make_double2(rng_u53(b.x, b.y), rng_u53(b.z, b.w));
}
__device__ __forceinline__ double2 rng_normal2(uint4 b) {
#ifdef ODR_RNG_ROCRAND
  return rocrand_device::detail::box_muller_double(b);
#else
  // rocrand: box_muller_double(uint4): u in (0, 1], w in (0, 2]; (sin, cos)(pi w) sqrt(-2 ln u)
  const unsigned long long v1 = (unsigned long long)b.x ^ ((unsigned long long)b.y << 21);
  const unsigned long long v2 = (unsigned long long)b.z ^ ((unsigned long long)b.w << 21);
  const double u = __dadd_rn(0x1p-53, __dmul_rn((double)v1, 0x1p-53));
  const double w = __dadd_rn(0x1p-52, __dmul_rn((double)v2, 0x1p-52));
  double2 r;
  {
#pragma clang fp contract(fast)
    const double lnu = log_pos(u);
    const double rad = fast_sqrt(-2.0 * lnu);
    // sincospi(w): quadrant k = rint(2 w) in 0 .. 4, remainder |w - k / 2| <= 1/4 exactly
    const double k = rint(2 * w), t = fma(-0.5, k, w);
    double sn, cs;
    sincos_q(t * kPi, sn, cs);
    const unsigned kq = (unsigned)(int)k & 3u;
    const double s1 = (kq & 1u) ? cs : sn, c1 = (kq & 1u) ? sn : cs;
    r.x = rad * ((kq == 2u || kq == 3u) ? -s1 : s1);
    r.y = rad * ((kq == 1u || kq == 2u) ? -c1 : c1);
  }
  return r;
#endif
}

// drift:current_uncertainty / drift:current_uncertainty_uniform of ONE Environment.get_environment call
// (environment.py:869-886).  The reference adds them in every call whose variables hold the current: the main-loop
// sample (call 0) and the one / three Runge-Kutta stage calls of advect_ocean_current (calls 1..3,
// physics_methods.py:638-670).  env[var] += draws is a float32 array += float64 array: float32(float64(u) + draw), first
// the normal pair, then the uniform pair.  ODR_RNG_HOST: the caller's np.random draws in the reference's call order
// (main: [ncomp][n], stage: [nstage][ncomp][n]); ODR_RNG_DEVICE: one Philox stream per (ID, step, call, distribution).
constexpr unsigned long long RNG_OFF_NOISE_UNIFORM = 64, RNG_OFF_NOISE_STAGE = 128;
struct StageNoise {
  int on, rng_mode, ncomp;
  int sm;   // stage math of the launch (ODR_STAGE_EXACT / ODR_STAGE_FAST): read by the kernels that take it at run time
  double std_n, std_u;
  const double *main, *stage;
  unsigned long long seed, step;
};
__device__ __forceinline__ void add_f32_f64(float &u, float &v, double nx, double ny) {
  u = (float)__dadd_rn((double)u, nx);
  v = (float)__dadd_rn((double)v, ny);
}
__device__ __forceinline__ void add_current_noise(const StageNoise &N, int call, long long i, long long n, int id,
                                                  float &u, float &v) {
  if (N.rng_mode == 1) {
    const double *a = call == 0 ? N.main : N.stage + (size_t)(call - 1) * (size_t)N.ncomp * (size_t)n;
    int c = 0;
    if (N.std_n > 0) { add_f32_f64(u, v, a[i], a[(size_t)n + i]); c = 2; }
    if (N.std_u > 0) add_f32_f64(u, v, a[(size_t)c * (size_t)n + i], a[(size_t)(c + 1) * (size_t)n + i]);
  } else {
    const unsigned long long off = RNG_OFF_NOISE + (call == 0 ? 4ull * (unsigned)VAR_U : RNG_OFF_NOISE_STAGE + 8ull * (unsigned)call);
    if (N.std_n > 0) {
      const double2 g = rng_normal2(rng_block(N.seed, id, N.step, off));
      add_f32_f64(u, v, g.x * N.std_n, g.y * N.std_n);
    }
    if (N.std_u > 0) {   // np.random.uniform(-std, std): low + (high - low) * u01
      const double2 q = rng_uniform2(rng_block(N.seed, id, N.step, off + (call == 0 ? RNG_OFF_NOISE_UNIFORM : 4ull)));
      add_f32_f64(u, v, fma(2.0 * N.std_u, q.x, -N.std_u), fma(2.0 * N.std_u, q.y, -N.std_u));
    }
  }
}

// ---- vertical mixing on a K column in LDS: device functions shared by k_vmix_col (odr_mix.hip) and the fused
// step + mixing kernel k_step_grid<..., MIXQ> (odr_step_mix.hip)
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), one 128-bit block
// from a 128-bit counter and a 64-bit key, evaluated directly: the mixing loop names the block it needs (element, step,
// block number) instead of advancing a generator state.  (Rounds 1-3 drew the same words through rocrand's state object,
// whose init and every rocrand4() each evaluate a block ahead: three blocks for the two that ten sub-steps consume.)
// Known answers: tests/test_philox.py (oracle/philox.py against the Random123 vectors), tests/test_gpu_parity.py.
// ODR_MIX_ROUNDS: rounds of the mixing loop's blocks.  Philox4x32-7 is the smallest member of the family the paper reports as
// passing BigCrush (its Table 2; 10 rounds is the library default with a safety margin); a block costs 154 instead of 305 issue
// cycles per wave on MI355X (profiles/r04_rng_cost.txt).  odr_version() names the count; oracle/philox.py restates it and is
// pinned on the Random123 known answers of BOTH counts (tests/test_philox.py).
#ifndef ODR_MIX_ROUNDS
#define ODR_MIX_ROUNDS 10
#endif
template <int ROUNDS>
__device__ __forceinline__ uint4 philox4x32(uint4 c, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x, p1 = (unsigned long long)0xCD9E8D57u * c.z;
    c = make_uint4((unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c;
}
// The uniforms of OceanDrift.vertical_mixing in ODR_RNG_DEVICE mode.  Block b of element `id` in step `step`:
//   counter = {b, step (low word), id, 0x4D495856 ("MIXV") ^ step (high word)},  key = seed.
// (The rocrand streams of the other consumers count in words 0-1 and hold the element ID in word 2 with word 3 = 0.)
// One block serves FIVE sub-steps -- 24 bits each: the upper 24 bits of the four words, then the four low bytes' worth
// taken from words 0..2 -- as (x + 1/2) 2^-24 in (0, 1), symmetric about 1/2: the random-walk displacement R = 2u - 1 is
// resolved to 1.2e-7 of its range (0.2 micrometres for a 1.9 m sub-step; the diffusivity itself is a float32).  Ten
// sub-steps (600 s / 60 s) cost two blocks of ~100 instructions, 20 of them 32 x 32 -> 64-bit multiplies.
struct MixKey { unsigned k0, k1, step_lo, tag, id; };
__device__ __forceinline__ MixKey mix_key(unsigned long long seed, unsigned long long step, int id) {
  MixKey K;
  K.k0 = (unsigned)seed; K.k1 = (unsigned)(seed >> 32);
  K.step_lo = (unsigned)step; K.tag = 0x4D495856u ^ (unsigned)(step >> 32); K.id = (unsigned)id;
  return K;
}
__device__ __forceinline__ uint4 mix_block(const MixKey &K, unsigned b) {
  return philox4x32<ODR_MIX_ROUNDS>(make_uint4(b, K.step_lo, K.id, K.tag), K.k0, K.k1);
}
__device__ __forceinline__ unsigned mix_word(const uint4 &q, unsigned k) {
  const unsigned lo = ((q.x & 255u) << 16) | ((q.y & 255u) << 8) | (q.z & 255u);
  return k == 0 ? q.x >> 8 : (k == 1 ? q.y >> 8 : (k == 2 ? q.z >> 8 : (k == 3 ? q.w >> 8 : lo)));
}
// the 24-bit draw of sub-step `it`.  primed: the caller drew block 0 itself (ahead of its column gathers, whose latency the
// block's arithmetic then overlaps)
__device__ __forceinline__ unsigned mix_draw(const MixKey &K, uint4 &q, int it, bool primed = false) {
  const unsigned b = (unsigned)it / 5u, k = (unsigned)it - 5u * b;
  if (k == 0 && !(primed && it == 0)) q = mix_block(K, b);
  return mix_word(q, k);
}
__device__ __forceinline__ double mix_uniform(const MixKey &K, uint4 &q, int it, bool primed = false) {
  return ((double)mix_draw(K, q, it, primed) + 0.5) * 5.9604644775390625e-08;
}
// R = 2 u - 1 of a 24-bit draw x, u = (x + 1/2) 2^-24: (2 x + 1 - 2^24) 2^-24.  Every step of 2 * u - 1 is exact for such a
// u (25 significant bits), so this is the value the reference's expression gives for it -- in two instructions instead of five.
__device__ __forceinline__ double mix_R_of(unsigned x) {
  return (double)((int)(2u * x + 1u) - (1 << 24)) * 5.9604644775390625e-08;
}
// Fast version for the common case -- the diffusivity comes from one gridded reader with a
// plain (not interleaved) z-innermost K array.  The host resolves source, time bracket and
// weight (VMixDesc); NQ = number of 4-level quads per column is a compile-time constant, so the
// 8 column gathers are straight-line 16-byte loads that are all in flight together (the generic
// kernel's run-time loops serialise them: one memory round trip per load), the level
// boundaries live in scalar registers and the np.gradient divisors are host constants.
constexpr int AUX_KMEMBER = 8;   // property slot that parks the member of an ensemble diffusivity (k_kmember)
struct VMixDesc {
  int sid, nzp, geo_slot, pad;
  const float *kb, *ka;  // K arrays of the bracketing time levels (ka == nullptr: on a time level)
  double wgt;            // weight_after (structured.py:353-354)
  float Kfb, pad2;
  const unsigned long long *guard;   // nullptr, or: the launch does nothing unless *guard != 0 (odr_ctx_guard_next_vmix)
};

// K column of one particle at (lon, lat) -> Kp[level][tid] (LDS), time-interpolated like the ReaderBlock's profiles
// quads: the 4-level quads of the column to gather (bit q; wave-uniform in k_vmix_col, see vmix_col_particle: a particle only
// ever reads the levels around its own, and a quad nobody of the wave needs is not fetched)
// hook: arithmetic of the caller that does not depend on the column (the first Philox block of the particle's stream), run
// right behind the gathers of the first quad so that it overlaps their flight instead of standing in front of their issue
struct VMixNoHook { __device__ __forceinline__ void operator()() const {} };
template <int NQ, bool TL, class HOOK = VMixNoHook>
__device__ __forceinline__ void vmix_col_fill(const DevSource &s, const VMixDesc &D, double lon, double lat, double *Kp,
                                              int tid, unsigned quads = ~0u, HOOK &&hook = HOOK()) {
  const double Kfb = (double)D.Kfb;
  double x, y;
  if (s.lon_mode == 1) lon = np_mod(lon + 180.0, 360.0) - 180.0;
  else if (s.lon_mode == 2) lon = np_mod(lon, 360.0);
  proj_fwd_rt(s.proj, lon, lat, x, y);
  const bool cov = x >= s.xmin && x <= s.xmax && y >= s.ymin && y <= s.ymax;
  if (s.mod360_x) x = np_mod(x, 360.0);
  const DevBlock &bb = s.slot[D.geo_slot];
  const double xi = __dmul_rn(div_cr(x - bb.x0, bb.xspan, bb.ixspan), (double)(bb.nx - 1));
  const double yi = __dmul_rn(div_cr(y - bb.y0, bb.yspan, bb.iyspan), (double)(bb.ny - 1));
  const int ny = bb.ny, nx = bb.nx;
  const Axis ay = axis_fp(yi, ny), ax = axis_fp(xi, nx);
  const double ty = ay.t, tx = ax.t, wy0 = 1 - ty, wx0 = 1 - tx, wgt = D.wgt;
  // uncovered particles gather node (0,0) and discard it: keeps the loads unconditional
  // byte offsets of the four node records (blocks of the fast path are `small`: < 2^24 nodes, < 4 GiB; 24-bit multiplies)
  const unsigned recb = (unsigned)bb.rec * 4u;
  const unsigned r0 = __umul24((unsigned)ay.i0, (unsigned)nx), r1 = __umul24((unsigned)ay.i1, (unsigned)nx);
#ifdef ODR_ABL_VMIX_UNIFORM   // what-if build (wrong values): every lane gathers node 0 -- what the spread of the gathers costs
  const unsigned o00 = 0u * r0, o01 = 0u * r1, o10 = 0u * recb, o11 = 0u;
#else
  const unsigned o00 = cov ? __umul24(r0 + (unsigned)ax.i0, recb) : 0u, o01 = cov ? __umul24(r0 + (unsigned)ax.i1, recb) : 0u;
  const unsigned o10 = cov ? __umul24(r1 + (unsigned)ax.i0, recb) : 0u, o11 = cov ? __umul24(r1 + (unsigned)ax.i1, recb) : 0u;
#endif
  const float *kb = D.kb, *ka = TL ? D.ka : D.kb;
  // horizontal weights multiplied out once for the whole column (float64; the layer value is rounded to float32 like
  // the ReaderBlock's: same bits as (v*wy)*wx summed, but for a float64 round-off that reaches the float32 rounding
  // in ~1e-8 of the values)
  const double w00 = wy0 * wx0, w01 = wy0 * tx, w10 = ty * wx0, w11 = ty * tx;
  if (!(quads & 1u)) hook();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (!((quads >> q) & 1u)) continue;
    // both time levels requested before anything is consumed
    const F4 b00 = ld_off<F4>(kb, o00 + 16u * q), b01 = ld_off<F4>(kb, o01 + 16u * q);
    const F4 b10 = ld_off<F4>(kb, o10 + 16u * q), b11 = ld_off<F4>(kb, o11 + 16u * q);
    F4 a00, a01, a10, a11;
    if (TL) {
      a00 = ld_off<F4>(ka, o00 + 16u * q); a01 = ld_off<F4>(ka, o01 + 16u * q);
      a10 = ld_off<F4>(ka, o10 + 16u * q); a11 = ld_off<F4>(ka, o11 + 16u * q);
    }
    if (q == 0) hook();
    double v[4];
#ifdef ODR_ABL_VMIX_FILLMATH   // what-if build (wrong values): the column's layer values without their float64 arithmetic
    v[0] = (double)(b00.x + b11.x); v[1] = (double)(b00.y + b11.y); v[2] = (double)(b00.z + b11.z); v[3] = (double)(b00.w + b11.w);
    if (TL) { v[0] += (double)a00.x; v[1] += (double)a01.y; v[2] += (double)a10.z; v[3] += (double)a11.w; }
#else
    v[0] = (double)bilw(b00.x, b01.x, b10.x, b11.x, w00, w01, w10, w11);
    v[1] = (double)bilw(b00.y, b01.y, b10.y, b11.y, w00, w01, w10, w11);
    v[2] = (double)bilw(b00.z, b01.z, b10.z, b11.z, w00, w01, w10, w11);
    v[3] = (double)bilw(b00.w, b01.w, b10.w, b11.w, w00, w01, w10, w11);
    if (TL) {
      double w[4];
      w[0] = (double)bilw(a00.x, a01.x, a10.x, a11.x, w00, w01, w10, w11);
      w[1] = (double)bilw(a00.y, a01.y, a10.y, a11.y, w00, w01, w10, w11);
      w[2] = (double)bilw(a00.z, a01.z, a10.z, a11.z, w00, w01, w10, w11);
      w[3] = (double)bilw(a00.w, a01.w, a10.w, a11.w, w00, w01, w10, w11);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = __dadd_rn(__dmul_rn(v[j], 1 - wgt), __dmul_rn(w[j], wgt));
    }
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) Kp[(4 * q + j) * BLOCK + tid] = (cov && isfinite(v[j])) ? v[j] : Kfb;
  }
}

// The ntimes_mix random-walk sub-steps of one particle on its K column in LDS (oceandrift.py:505-565).  Returns the
// new z; sf_flags: 1 deactivated on the sea floor, 2 moved back horizontally (general:seafloor_action).
struct VMixArgs {
  double dt, dt_mix_cfg;
  int mix_at_surface, rng_mode, sfl, pad;
  const double *huni;
  unsigned long long seed, step;
};
struct MixRng { MixKey key; uint4 q; bool primed; };
// the particle's stream of this step with its first block drawn (ODR_RNG_DEVICE)
__device__ __forceinline__ MixRng mix_rng_begin(const VMixArgs &A, int id) {
  MixRng R;
  R.key = mix_key(A.seed, A.step, id);
  R.q = make_uint4(0u, 0u, 0u, 0u);
  R.primed = false;
#ifndef ODR_ABL_VMIX_NORNG
  if (A.rng_mode == 0) {
    R.q = mix_block(R.key, 0u);
    R.primed = true;
  }
#endif
  return R;
}
// Quads on demand (k_vmix_col): `lz` names the quads vmix_col_fill has put into Kp (wave-uniform).  A sub-step that needs the
// level terms of a level whose neighbours lie in a quad that was not gathered sets `missed`; the caller then gathers the whole
// column and walks the particle again from its start (same draws: the stream is a function of element, step and sub-step) --
// rare: the levels a particle reaches within a step are the ones around its own (C3: 0.07 % of the particles per step leave
// their three-level window at all, none reaches the quad below 100 m).  Level terms are a pure function of the column: same bits.
struct VMixLazy { unsigned loaded; bool missed; };
__device__ __forceinline__ unsigned vmix_quads_of(int lo, int hi, int nzp) {   // quads holding levels lo..hi (clamped to the column)
  lo = lo < 0 ? 0 : lo; hi = hi > nzp - 1 ? nzp - 1 : hi;
  const unsigned a = (unsigned)lo >> 2, b = (unsigned)hi >> 2;
  return ((2u << b) - 1u) & ~((1u << a) - 1u);
}
template <int NQ>
__device__ __forceinline__ double vmix_col_walk(const DevSource &s, int nzp, const double *Kp, const double *gsh, int tid,
                                                const VMixArgs &A, long long i, long long n, int id, double z, int &moving,
                                                float Zmin, float tv, int &sf_flags, const MixRng *pre = nullptr,
                                                VMixLazy *lz = nullptr) {
  constexpr int NL = 4 * NQ;
  const double dt = A.dt;
  const int mix_at_surface = A.mix_at_surface, rng_mode = A.rng_mode, sfl = A.sfl;
  // level boundaries (see k_vmix) in scalar registers; levels past the profile never match
  double zm[NL - 1];
#pragma unroll
  for (int k = 0; k < NL - 1; ++k) zm[k] = k < nzp - 1 ? s.zmid[k] : __builtin_inf();
  const bool uniform_z = s.vg_uniform != 0;
  const double gd0 = s.vg_d[0], gi0 = s.vg_id[0], gd1 = s.vg_d[1], gi1 = s.vg_id[1], gd2 = s.vg_d[2], gi2 = s.vg_id[2];
  const double sgn = dt > 0 ? 1.0 : (dt < 0 ? -1.0 : 0.0);
  const double dt_mix = A.dt_mix_cfg * sgn;
#ifdef ODR_ABL_VMIX_NOLOOP   // what-if build: no sub-steps, the window's level terms still formed (they flow into z below)
  const int ntimes = 0;
#elif defined(ODR_ABL_VMIX_NT)   // what-if build: fewer sub-steps
  const int ntimes = ODR_ABL_VMIX_NT;
#else
  const int ntimes = abs((int)(dt / dt_mix));
#endif
  const double r = 1.0 / 3, ir = 1.0 / r;
  // w*dt_mix*moving: dt_mix is a NumPy float64 scalar (np.sign, oceandrift.py:416) -> float64 product under NumPy 2
  double wstep = __dmul_rn(__dmul_rn((double)tv, dt_mix), (double)moving);
  const MixKey st = mix_key(A.seed, A.step, id);
  uint4 u4 = make_uint4(0u, 0u, 0u, 0u);
  bool primed = false;
  if (pre) { u4 = pre->q; primed = pre->primed; }
  // -dK/dz * dt_mix and sqrt(K |dt_mix| 2 / r) of one level (oceandrift.py:501-502,527-528)
  auto level_terms = [&](int zl, double &dk_dt, double &sg) {
    if (lz && (vmix_quads_of(zl - 1, zl + 1, nzp) & ~lz->loaded)) lz->missed = true;   // (the values formed below are dropped)
    const double Kz = Kp[zl * BLOCK + tid];
    double gK;  // np.gradient(Kprofiles, mixing_z, axis=0)[zl]
    if (zl == 0) gK = div_cr(Kp[BLOCK + tid] - Kz, gd0, gi0);
    else if (zl == nzp - 1) gK = div_cr(Kz - Kp[(nzp - 2) * BLOCK + tid], gd1, gi1);
    else if (uniform_z) gK = div_cr(Kp[(zl + 1) * BLOCK + tid] - Kp[(zl - 1) * BLOCK + tid], gd2, gi2);
    else
      gK = __dadd_rn(__dadd_rn(__dmul_rn(gsh[zl], Kp[(zl - 1) * BLOCK + tid]), __dmul_rn(gsh[NL + zl], Kz)),
                     __dmul_rn(gsh[2 * NL + zl], Kp[(zl + 1) * BLOCK + tid]));
    double dK = -gK;
    if (fabs(dK) < 1e-10) dK = 0;  // gradK[np.abs(gradK)<1e-10] = 0 (:502)
    dk_dt = __dmul_rn(dK, dt_mix);
    sg = sqrt(div_cr(__dmul_rn(__dmul_rn(Kz, fabs(dt_mix)), 2.0), r, ir));
  };
  // A particle rarely leaves the three levels around its starting one within a step: their terms are derived once
  // (branch-free selection in the loop); anything else is derived on demand.  With a per-iteration "derive when the
  // level changes" scheme the 64 lanes of a wave make that branch fire in practically every sub-step.
  int lv0;
  {
    const double d0 = -z;
    int zs = 0;
#pragma unroll
    for (int k = 0; k < NL - 1; ++k) zs += ((k & 1) ? d0 >= zm[k] : d0 > zm[k]) ? 1 : 0;
    lv0 = zs < 1 ? 1 : (zs > nzp - 2 ? nzp - 2 : zs);   // centre of the cached window [lv0-1, lv0+1]
    if (nzp < 3) lv0 = 1;
  }
  double c_dk[3], c_sg[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int zl = lv0 - 1 + q;
    c_dk[q] = 0; c_sg[q] = 0;
    if (zl >= 0 && zl < nzp) level_terms(zl, c_dk[q], c_sg[q]);
  }
#ifdef ODR_ABL_VMIX_NOLOOP
  z += 1e-30 * (c_dk[0] + c_dk[1] + c_dk[2] + c_sg[0] + c_sg[1] + c_sg[2]);
#endif
  // The level of a sub-step is the number of boundaries below d (odd ones count when d >= zm, even ones when d > zm).
  // Inside the cached window only the window's own four boundaries decide -- lv0 - 2 ... lv0 + 1, per lane, with the
  // ">=" of the odd ones folded into the value (d >= b  <=>  d > the double just below b; b > 0) -- four compares
  // instead of NL - 1; a sub-step that leaves the window counts all of them (round 3: 22 -> 8 instructions per sub-step).
  double wb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = lv0 - 2 + j;
    double b = k < 0 ? -1.0 : __builtin_inf();
    if (k >= 0 && k < nzp - 1) {
      b = gsh[3 * NL + k];
      if ((k & 1) && b > 0) b = __longlong_as_double(__double_as_longlong(b) - 1);
    }
    wb[j] = b;
  }
  // the sub-steps in groups of five = one Philox block: inside the unrolled group the word a sub-step takes from the block is
  // known at compile time (a run-time `it % 5` costs a chain of selects and the fifth word's assembly in every sub-step)
  for (int it0 = 0; it0 < ntimes; it0 += 5) {
#ifdef ODR_ABL_VMIX_NORNG   // what-if build: no generator (wrong values)
  u4 = make_uint4(st.id * 2654435761u + (unsigned)it0, st.id ^ 0x9E3779B9u, st.id * 40503u, st.id + (unsigned)it0);
#else
  if (rng_mode == 0 && !(primed && it0 == 0)) u4 = mix_block(st, (unsigned)it0 / 5u);
#endif
#ifdef ODR_VMIX_NO_UNROLL   // A/B build: the word selected at run time, as in rounds 1-3
#pragma unroll 1
#else
#pragma unroll
#endif
  for (int k5 = 0; k5 < 5; ++k5) {
    const int it = it0 + k5;
    if (it >= ntimes) break;
    const bool surface = z == 0;
    const double d = -z;
    int q = (d > wb[1] ? 1 : 0) + (d > wb[2] ? 1 : 0);
    int zi = lv0 - 1 + q;
    if (!(d > wb[0]) || d > wb[3]) {     // outside the window (rare; also NaN)
      zi = 0;
#pragma unroll
      for (int k = 0; k < NL - 1; ++k) zi += ((k & 1) ? d >= zm[k] : d > zm[k]) ? 1 : 0;
      q = zi - lv0 + 1;
    }
    double dKdt = q == 0 ? c_dk[0] : (q == 1 ? c_dk[1] : c_dk[2]);
    double sig = q == 0 ? c_sg[0] : (q == 1 ? c_sg[1] : c_sg[2]);
    if (q < 0 || q > 2) level_terms(zi, dKdt, sig);
    double R;
    if (rng_mode == 1) R = __dsub_rn(__dmul_rn(2.0, A.huni[(size_t)it * n + i]), 1.0);
    else R = mix_R_of(mix_word(u4, (unsigned)k5));
    z = __dsub_rn(z, __dmul_rn((double)moving, __dsub_rn(dKdt, __dmul_rn(R, sig))));
    if (z >= 0) z = -z;
    if (z < (double)Zmin && moving == 1) z = __dsub_rn((double)__fmul_rn(2.f, Zmin), z);
    z = __dadd_rn(z, wstep);
    if (!mix_at_surface && surface) z = 0.0;
    if (z > 0) z = 0.0;
    if (z < (double)Zmin) {   // "let particles stick to bottom": interact_with_seafloor() inside the loop (oceandrift.py:555-559)
      const int act = sfl & 255;
      if (act == 3) sf_flags |= 2;                       // previous: lon/lat go back, z stays
      else if (act) {
        z = (double)Zmin;                                // lift_to_seafloor / deactivate
        if (act == 2) { sf_flags |= 1; moving = 0; wstep = 0.0; }
      }
    }
  }
  }
  return z;
}

// out[j] with a run-time j, as a chain of selects: a dynamically indexed register array lives in scratch memory
// (k_step_grid<RK4>: 80 B per lane, 0.5 GB of HBM writes per launch at 10 M particles)
__device__ __forceinline__ float pick_slot(const float (&a)[MAXG], int j) {
  float r = a[0];
#pragma unroll
  for (int k = 1; k < MAXG; ++k) {
    float v = a[k];
    asm volatile("" : "+v"(v));   // opaque: keeps the optimiser from folding the selects back into one indexed load
    r = j == k ? v : r;
  }
  return r;
}

#ifdef ODR_TU_MISC
// ------------------------------------------------------------------ environment
// Environment.get_environment for one variable group of NV variables
template <int NV>
struct GroupVars { int v[NV]; };

template <int NV>
__global__ __launch_bounds__(BLOCK) void k_env_group(const DevWorld *__restrict__ W, PView p,
                                                     GroupVars<NV> gv, double t, int record_prev) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  int vars[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) vars[k] = gv.v[k];
  double lon = p.lon[i], lat = p.lat[i], z = p.z[i];
  float out[NV];
  env_group<NV>(*W, vars, lon, lat, z, t, out, p.rank ? p.rank[i] : 0, W->f32pos & 1);   // (the main-loop call: DevWorld::f32pos)
#pragma unroll
  for (int k = 0; k < NV; ++k) p.env[vars[k]][i] = out[k];
  if (record_prev) { p.slon[i] = lon; p.slat[i] = lat; }
}

// fast version: the whole group comes from one gridded reader (odr_field.hip.h)
template <int PROJ>
__global__ __launch_bounds__(BLOCK, ODR_WAVES(PROJ)) void k_env_grid(const DevWorld *__restrict__ W, PView p, EnvGroupDesc G,
                                                    int record_prev) {
  long long i = pid();
  if (i >= p.n) return;
  double lon = p.lon[i], lat = p.lat[i], z = p.z[i];
  int f32idx = 0;
  if (W->f32pos & 1) {   // first get_environment of a run (DevWorld::f32pos): float32 longitude modulation, float32 index maps
    lon = lon_f32class(W->src[G.sid].lon_mode, lon);
    if (PROJ == PROJ_LATLONG) f32idx = W->src[G.sid].xy_f32;
  }
  float out[MAXG];
  env_group_fast<PROJ>(*W, G, lon, lat, z, out, f32idx);
#pragma unroll
  for (int k = 0; k < MAXG; ++k)
    if (k < G.nv) G.out_ptr[k][i] = out[k];
  if (record_prev) { p.slon[i] = lon; p.slat[i] = lat; }
}

__global__ __launch_bounds__(BLOCK) void k_fill_f32(float *a, long long n, float v) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) a[i] = v;
}

__global__ __launch_bounds__(BLOCK) void k_record_prev(PView p, int which) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  if (which == 0) { p.slon[i] = p.lon[i]; p.slat[i] = p.lat[i]; }
  else { p.plon[i] = p.lon[i]; p.plat[i] = p.lat[i]; }
}

#endif  // ODR_TU_MISC
#ifdef ODR_TU_STEP
// --------------------------------------------------------------------- advection
// PhysicsMethods.advect_ocean_current (physics_methods.py:611-691) with every RK
// sub-stage -- geodesic to the stage position, reader front door, block gathers,
// time/z interpolation, vector rotation, float32 environment cast -- fused in one
// kernel; the particle never leaves registers between stages.
// RK4 combination (x_vel + 2*x_vel2 + 2*x_vel3 + x_vel4)/6.0 in float32, left to right
// (physics_methods.py:674-675)
__device__ __forceinline__ float rk4_mix(float k1, float k2, float k3, float k4) {
  float s = __fadd_rn(k1, __fmul_rn(2.0f, k2));
  s = __fadd_rn(s, __fmul_rn(2.0f, k3));
  s = __fadd_rn(s, k4);
  return div_cr_f32(s, 6.0f, 1.0f / 6.0f);  // == s / 6.0f, correctly rounded
}

// generic version: any mix of readers behind the (u,v) priority list
template <int SCHEME, bool NOISE>
__global__ __launch_bounds__(BLOCK) void k_advect(const DevWorld *__restrict__ W, PView p, double t,
                                                  double dt, float factor, StageNoise N) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  const int uv[2] = {VAR_U, VAR_V};
  double lon = p.lon[i], lat = p.lat[i], z = p.z[i];
  float u1 = p.env[VAR_U][i], v1 = p.env[VAR_V][i];
  float f = __fmul_rn(current_factor(p, i, factor), p.cdf[i]);  // factor*cdf, float32
  int moving = p.moving[i];
  const int rk = p.rank ? p.rank[i] : 0;
  float fu, fv;
  GeodStart o = geod_start(lat, lon);
  if (SCHEME == 0) {
    fu = __fmul_rn(f, u1);
    fv = __fmul_rn(f, v1);
  } else {
    float dtf = (float)dt;
    double lon2, lat2;
    float k2[2];
    stage_pos_rt(N.sm, o, u1, v1, dtf, lon2, lat2);
    env_group<2>(*W, uv, lon2, lat2, z, t + dt / 2, k2, rk);
    const int id = NOISE ? p.id[i] : 0;
    if (NOISE) add_current_noise(N, 1, i, p.n, id, k2[0], k2[1]);
    if (SCHEME == 1) {
      fu = __fmul_rn(f, k2[0]);
      fv = __fmul_rn(f, k2[1]);
    } else {
      float k3[2], k4[2];
      stage_pos_rt(N.sm, o, k2[0], k2[1], dtf, lon2, lat2);
      env_group<2>(*W, uv, lon2, lat2, z, t + dt / 2, k3, rk);
      if (NOISE) add_current_noise(N, 2, i, p.n, id, k3[0], k3[1]);
      stage_pos_rt(N.sm, o, k3[0], k3[1], dtf, lon2, lat2);  // dt*.5 again: reference quirk (:662)
      env_group<2>(*W, uv, lon2, lat2, z, t + dt, k4, rk);
      if (NOISE) add_current_noise(N, 3, i, p.n, id, k4[0], k4[1]);
      fu = __fmul_rn(rk4_mix(u1, k2[0], k3[0], k4[0]), f);
      fv = __fmul_rn(rk4_mix(v1, k2[1], k3[1], k4[1]), f);
    }
  }
  move_f32_from(o, lon, lat, fu, fv, moving, dt);
  p.lon[i] = lon;
  p.lat[i] = lat;
}

// fast version: (u,v) from one gridded reader, interleaved z-innermost blocks, host-resolved
// time brackets (odr_field.hip.h "fast (u,v) path")
// lh / lf: the loaders of the half-step and full-step stage samples (uv_global(th / tf), or the workgroup's LDS tile: a
// stage footprint that leaves the tile's rectangle is sampled from the blocks in HBM instead -- same values)
template <int PROJ, bool IS3D, int SM, class LD>
__device__ __forceinline__ void uv_stage_or_global(const DevSource &s, const DevBlock &geo, const UVTime &tm, const LD &ld, double lon,
                                                   double lat, double z, const ZBracket &zb, float fbu, float fbv, float &uo,
                                                   float &vo, const ProjStart &ps, UVKeep<IS3D> &K, bool keep) {
  if (!uv_stage<PROJ, IS3D, SM>(s, geo, tm, ld, lon, lat, z, zb, fbu, fbv, uo, vo, ps, K, keep)) {
    if constexpr (sizeof(LD) != sizeof(LdGlobal)) uv_stage<PROJ, IS3D, SM>(s, geo, tm, uv_global(tm), lon, lat, z, zb, fbu, fbv, uo, vo, ps, K, keep);
  }
}
template <int SCHEME, int PROJ, bool IS3D, bool NOISE, int SM = 0, bool PARK = false, class LD = LdGlobal>
__device__ __forceinline__ void advect_grid_body(const DevSource &s, const DevBlock &geo, double &lon, double &lat,
                                                 double z, float u1, float v1, float f, int moving, double dt,
                                                 const UVTime &th, const UVTime &tf, const LD &lh, const LD &lf, float fbu, float fbv,
                                                 const StageNoise &N, long long i, long long n, int id, UVKeep<IS3D> &K,
                                                 ZBracket zb_pre = ZBracket(), bool have_pre = false,
                                                 double *park = nullptr ODR_PT_PARAM, const ProjStart ps_pre = ProjStart()) {
  float fu, fv;
  // the full-step stage may use (and refresh) the kept records when its time bracket is that of the half-step stages
  const bool keep_f = tf.b == th.b && tf.a == th.a;
  GeodStart o0 = geod_start(lat, lon);
#ifndef ODR_FULL_GEODESIC
  // PARK: the series coefficients wait in LDS between the moves (geod_park); the start point is rebuilt where it is used
  lds_f64 *slot = (lds_f64 *)park;
  if constexpr (PARK) { geod_park(o0, slot); if (SM == 1 && SCHEME == 2) slot[14 * BLOCK] = (double)f; }
  auto O = [&]() { if constexpr (PARK) return geod_unpark(slot[12 * BLOCK], slot[13 * BLOCK], slot); else return o0; };
#else
  auto O = [&]() { return o0; };
#endif
#define ODR_O O()
#ifndef ODR_FULL_GEODESIC
#define ODR_STAGE_POS(U_, V_) do { if constexpr (PARK && SM == 1) stage_pos_parked(slot, U_, V_, dtf, lon2, lat2); \
                                   else stage_pos<SM>(ODR_O, U_, V_, dtf, lon2, lat2); } while (0)
#else
#define ODR_STAGE_POS(U_, V_) stage_pos<SM>(ODR_O, U_, V_, dtf, lon2, lat2)
#endif
  if (SCHEME == 0) {
    fu = __fmul_rn(f, u1);
    fv = __fmul_rn(f, v1);
  } else {
    ZBracket zb;
    zb.iz0 = 0; zb.same = 0; zb.wa = 1;
    // the bracket of the main-loop sample serves the stages as well (same z unless the sea floor lifted the element)
    if (IS3D) { if (have_pre) zb = zb_pre; else zb = zbracket(s, z); }
    float dtf = (float)dt;
    double lon2, lat2;
    float u2, v2;
    // projected readers: sines / cosines of the particle's own position once, the stage positions relative to it
    ProjStart ps;
    if (PROJ == PROJ_STERE_POLAR || (ODR_PROJ_ROTATES(PROJ) && s.proj.kind == PROJ_STERE_POLAR && s.proj.es != 0)) {
      if (ps_pre.ok) ps = ps_pre;     // (formed in front of the main-loop sample, k_step_grid; the element has not been moved since)
      else {
        double lw = lon;
        if (s.lon_mode == 1) lw = np_mod(lw + 180.0, 360.0) - 180.0;
        else if (s.lon_mode == 2) lw = np_mod(lw, 360.0);
        ps = proj_start(s.proj, lw, lat);
      }
    }
    ODR_STAGE_POS(u1, v1);
    ODR_PT_USE(lon2); ODR_PT_USE(lat2); ODR_PT(4);
    uv_stage_or_global<PROJ, IS3D, SM>(s, geo, th, lh, lon2, lat2, z, zb, fbu, fbv, u2, v2, ps, K, true);
    if (NOISE) add_current_noise(N, 1, i, n, id, u2, v2);
    ODR_PT_USE(u2); ODR_PT_USE(v2); ODR_PT(5);
    if (SCHEME == 1) {
      fu = __fmul_rn(f, u2);
      fv = __fmul_rn(f, v2);
    } else {
      float u3, v3, u4, v4;
      ODR_STAGE_POS(u2, v2);
      uv_stage_or_global<PROJ, IS3D, SM>(s, geo, th, lh, lon2, lat2, z, zb, fbu, fbv, u3, v3, ps, K, true);
      if (NOISE) add_current_noise(N, 2, i, n, id, u3, v3);
      ODR_PT_USE(u3); ODR_PT_USE(v3); ODR_PT(6);
      ODR_STAGE_POS(u3, v3);
      uv_stage_or_global<PROJ, IS3D, SM>(s, geo, tf, lf, lon2, lat2, z, zb, fbu, fbv, u4, v4, ps, K, keep_f);
      if (NOISE) add_current_noise(N, 3, i, n, id, u4, v4);
      ODR_PT_USE(u4); ODR_PT_USE(v4); ODR_PT(7);
      float fl = f;
#ifndef ODR_FULL_GEODESIC
      if constexpr (PARK && SM == 1) fl = (float)slot[14 * BLOCK];   // (parked with the coefficients: one register less through the stages)
#endif
      fu = __fmul_rn(rk4_mix(u1, u2, u3, u4), fl);
      fv = __fmul_rn(rk4_mix(v1, v2, v3, v4), fl);
    }
  }
  move_f32_from(ODR_O, lon, lat, fu, fv, moving, dt);
#undef ODR_O
#undef ODR_STAGE_POS
}

template <int SCHEME, int PROJ, bool IS3D, bool NOISE, int SM = 0>
__global__ __launch_bounds__(BLOCK, ODR_STEP_WAVES(PROJ)) void k_advect_grid(const DevWorld *__restrict__ W, int sid, int geo_slot,
                                                       PView p, double dt, float factor, UVTime th,
                                                       UVTime tf, StageNoise N) {
  long long i = pid();
  if (i >= p.n) return;
  const DevSource &s = W->src[sid];
  const DevBlock &geo = s.slot[geo_slot];
  double lon = p.lon[i], lat = p.lat[i];
  UVKeep<IS3D> K;
  K.valid = false; K.n00 = K.n11 = K.kb = 0;
  advect_grid_body<SCHEME, PROJ, IS3D, NOISE, SM>(s, geo, lon, lat, p.z[i], p.env[VAR_U][i], p.env[VAR_V][i],
                                              __fmul_rn(current_factor(p, i, factor), p.cdf[i]), p.moving[i], dt, th, tf, uv_global(th), uv_global(tf),
                                              W->fallback[VAR_U], W->fallback[VAR_V], N, i, p.n, NOISE ? p.id[i] : 0, K);
  p.lon[i] = lon;
  p.lat[i] = lat;
}

#endif  // ODR_TU_STEP
// -------------------------------------------------------------------- reductions
// red[] slots
enum { R_NACT = 0, R_LONMIN, R_LONMAX, R_LATMIN, R_LATMAX, R_ZMIN, R_ZMAX, R_DMAX, R_STOKESMAX,
       R_WSPEEDMAX, R_WDFMAX, R_NSURF, R_HSMAX, R_TPMAX, R_RELWSPEEDMAX, R_MLDMAX, R_N };

__device__ __forceinline__ void atomic_max_d(double *addr, double v) {
  unsigned long long *a = (unsigned long long *)addr, old = *a, assumed;
  do {
    assumed = old;
    if (__longlong_as_double((long long)assumed) >= v) break;
    old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
  } while (assumed != old);
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#ifdef ODR_TU_STEP
// One launch for  get_environment -> interact_with_coastline -> update_previous_state ->
// advect_ocean_current  (the run() loop order, basemodel/__init__.py:2136-2248) when the group that
// holds the current comes from one gridded reader: same arithmetic as the four separate kernels
// (k_env_grid, k_coast, k_record_prev, k_advect_grid), but the particle stays in registers, and the
// latency-bound block gathers of the environment sample overlap with the float64 geodesics of other
// waves.  The host orders the group so that slot 0 = x_sea_water_velocity, 1 = y_sea_water_velocity
// and, if `land_slot` == 2, 2 = land_binary_mask.
struct StepDesc {
  int coast_action, stranded_code, seeded_code, land_slot;  // land_slot: 2 or -1 (read p.env[LAND])
  int store_previous, geo_slot_uv;
  int seafloor, depth_slot;   // interact_with_seafloor 'lift_to_seafloor' (:748-783); depth_slot 2 | 3 | -1 (read p.env[DEPTH])
  float age_dt, max_age;      // increase_age_and_retire (:2342-2352); age_dt == 0: not part of this call
  int retired_code, missing_code;   // report_missing_variables: NaN in a sampled variable whose fallback is None
  int nmiss_grp, nmiss_rest;
  int miss_grp[4], miss_rest[4];    // group slots / variable ids (sampled by the preceding launch) to test for NaN
  int main_noise;                   // uncertainty of the main-loop sample of the current (StageNoise call 0)
  int ssh_slot;                     // group slot of sea_surface_height, or -1: sampled by the preceding launch (p.env[SSH])
  // The global tests the movers of this step open with -- no element at the surface, wind_drift_factor / wind speed /
  // Stokes drift / diffusivity identically zero (physics_methods.py:741-747,771-780,799-804, basemodel/__init__.py:1754) --
  // formed by THIS launch (odr_ctx_set_step_reduce), which holds every value they need in registers; the separate pass
  // over the arrays (k_reduce, 0.105 ms at 6.25 M elements) is then not made.  Slots: group slot, -1 = p.env[] (sampled by
  // the preceding launch), -2 = the variable is not there.
  int red_on, red_rel;
  int red_hd, red_sx, red_xw, red_pad;   // horizontal_diffusivity; Stokes x (y = next slot / array); x_wind (y_wind likewise)
  double red_wdd, red_iwdd;         // wind_drift_depth as given (sign and zero matter, :754-757); 1 / |wind_drift_depth|
  double *red;                      // per-wave records [waves of the launch][6] (see the tail of k_step_grid)
  // The status scan the run() loop makes right behind this launch (odr_scan_status: how many elements stay, which provisional
  // status numbers occur) formed by the launch itself: wcount[i / 64] = elements of that wave with status 0, bit (status - 100) of
  // *sflags.  k_cmp_total folds the wave counts; the pass over the status array (k_cmp_count, 32 us at 10 M elements) is not made.
  unsigned *wcount;                 // nullptr: not asked for
  unsigned long long *sflags;
};

// (The LDS field tile is a kernel of its own since round 4: k_step_tile, odr_tile.hip.h.)
// MIXQ > 0: OceanDrift.vertical_mixing (+ vertical_advection) of the same step runs in this launch as well (the body of
// k_vmix_col<MIXQ, MIXTL>): the K column is gathered at the sample position while the particle is in registers, the
// random walk follows the horizontal move; z, moving, depth, ssh and the sample position are not written and read
// again, one launch and one pass over the particle state less.  Same arithmetic, same bits as the two launches.
struct StepMix {
  VMixDesc D;
  VMixArgs A;
  int vadv, w_slot;   // vertical advection: -1 none | 0 below the surface | 1 including it; slot of W in the group or -1
};
// lat / lon and curvilinear readers, Runge-Kutta schemes: geodesic coefficients parked in LDS, 5 waves per SIMD (geod_park)
#ifndef ODR_PARK_WAVES
#ifdef ODR_NO_KEEP
#define ODR_PARK_WAVES 5
#else
#define ODR_PARK_WAVES 4   // with the kept (u,v) records of the footprint (UVKeep: 35 registers): 126 registers; 4 and 5 waves per SIMD ran alike before (profiles/r03_ab_variants.txt)
#endif
#endif
// Round 5: the lat / lon 3-D instantiations with the FAST stage arithmetic fit 96 registers without scratch memory -- kept values
// combined over the vertical bracket (16 registers instead of 32), slots B / C / D sampled before slot A, the start point and the
// drift factor parked with the geodesic coefficients, bookkeeping state requested behind the sample -- and run at FIVE waves per
// SIMD; the launch time of this kernel goes with 1 / waves (profiles/r05_ab_variants.txt section 4: 3 waves +44 %, 2 waves +146 %).
// The 2-D, curvilinear and EXACT instantiations need more and stay at ODR_PARK_WAVES.
#ifndef ODR_PARK_WAVES_FAST3D
#define ODR_PARK_WAVES_FAST3D 5
#endif
#if defined(ODR_NO_KEEP)
#define ODR_PARK_WAVES_OF(PROJ, IS3D, SM) ODR_PARK_WAVES
#else
#define ODR_PARK_WAVES_OF(PROJ, IS3D, SM) (((PROJ) == PROJ_LATLONG && (IS3D) && (SM) == 1) ? ODR_PARK_WAVES_FAST3D : ODR_PARK_WAVES)
#endif
#if defined(ODR_FULL_GEODESIC) || defined(ODR_NO_PARK)
#define ODR_STEP_PARKS(SCHEME, PROJ, MIXQ) false
#else
#define ODR_STEP_PARKS(SCHEME, PROJ, MIXQ) ((SCHEME) > 0 && (MIXQ) == 0 && ((PROJ) == PROJ_LATLONG || (PROJ) == PROJ_CURVILINEAR))
#endif
template <int SCHEME, int PROJ, bool IS3D, bool NOISE, int MIXQ = 0, bool MIXTL = false, int SM = 0>
__global__ __launch_bounds__(BLOCK, ODR_STEP_PARKS(SCHEME, PROJ, MIXQ) ? ODR_PARK_WAVES_OF(PROJ, IS3D, SM) : ((MIXQ > 0 && ODR_STEP_WAVES(PROJ) < ODR_MIX_WAVES) ? ODR_MIX_WAVES : ODR_STEP_WAVES(PROJ))) void k_step_grid(const DevWorld *__restrict__ W, PView p, EnvGroupDesc G,
                                                     StepDesc S, double dt, float factor, UVTime th, UVTime tf,
                                                     unsigned long long *n_hit, StageNoise N,
                                                     StepMix M = StepMix()) {
  long long i = pid();
  bool hit = false;
  ODR_PT_DECL;
  ODR_PT(0);
  constexpr bool PARK = ODR_STEP_PARKS(SCHEME, PROJ, MIXQ);
  __shared__ double s_park[PARK ? GEOD_PARK * BLOCK : 1];
  __shared__ double s_zt[IS3D ? 3 * ZT_STRIDE : 1];   // interp1d tables of the reader's z grid (zinterp)
  const double *zt = nullptr;
  // the particle's position requested BEFORE the table staging and its barrier (round 6: phase stamps put 9.6 % of a wave's
  // life between its entry and the arrival of lon / lat / z -- kernel arguments -> source fields -> table loads -> barrier ->
  // state loads, five dependent round trips; the state loads need the kernel arguments only)
  double lon_e = 0, lat_e = 0, z_e = 0;
#ifndef ODR_NO_EARLY_STATE
  if (i < p.n) { lon_e = p.lon[i]; lat_e = p.lat[i]; z_e = p.z[i]; }
#endif
  if (IS3D) { zt_stage(W->src[G.sid], s_zt); zt = s_zt; }
  double *Kp = nullptr, *gsh = nullptr;
  if (MIXQ > 0) {
    extern __shared__ __attribute__((aligned(16))) char mix_mem[];
    constexpr int NL = 4 * (MIXQ > 0 ? MIXQ : 1);
    Kp = (double *)mix_mem;                  // [NL][BLOCK]
    gsh = Kp + (size_t)NL * BLOCK;           // [4][NL]
    const DevSource &sk = W->src[M.D.sid];
    if ((int)threadIdx.x < M.D.nzp) {
      const int t_ = threadIdx.x;
      gsh[t_] = sk.vg_a[t_]; gsh[NL + t_] = sk.vg_b[t_]; gsh[2 * NL + t_] = sk.vg_c[t_]; gsh[3 * NL + t_] = sk.zmid[t_];
    }
    __syncthreads();
  }
  // this lane's share of the movers' tests (S.red_on): neutral unless it holds an element that stays active
  unsigned r_bits = 0;   // 1 D > 0, 2 D == 0, 4 Stokes sum > 0, 8 == 0, 16 at the surface, 32 wind drift factor > 0, 64 == 0, 128 wind speed > 0, 256 == 0
  if (i < p.n) {
#ifndef ODR_NO_EARLY_STATE
    double lon = lon_e, lat = lat_e;
    const double z = z_e;
#else
    double lon = p.lon[i], lat = p.lat[i];
    const double z = p.z[i];
#endif
    // everything the bookkeeping below reads of this particle, requested together with the position: each of these
    // loads after the environment stores is a memory round trip of its own (float stores may alias float loads)
    // ODR_STATE_LATE (the 3-D lat / lon instantiations of round 5): moving, status, age, drift factor and ssh are requested BEHIND
    // the sample instead of with the position -- one more dependent round trip, five registers less through the main-loop sample,
    // which with the rest of round 5's register work leaves the kernel at <= 96 registers = 5 waves per SIMD
#ifdef ODR_STATE_EARLY
    constexpr bool STATE_LATE = false;
#else
    constexpr bool STATE_LATE = ODR_STEP_PARKS(SCHEME, PROJ, MIXQ) && IS3D;
#endif
    int moving = 0, st = 0;
    float age0 = 0.f, cdf0 = 0.f, ssh0 = 0.f;
    auto load_state = [&]() {
      moving = p.moving[i];
      st = p.status[i];
      age0 = p.age[i]; cdf0 = p.cdf[i];
      ssh0 = (S.seafloor || MIXQ > 0) && p.env[VAR_SSH] ? p.env[VAR_SSH][i] : 0.f;
    };
    if constexpr (!STATE_LATE) load_state();
    constexpr bool RED = !IS3D && MIXQ == 0;      // (the movers' tests, below)
    const float wdf0 = (RED && S.red_on && S.red_xw > -2) ? p.wdf[i] : 0.f;
    ODR_PT_USE(lon); ODR_PT_USE(lat); ODR_PT_USE(z); ODR_PT(1);
    float out[MAXG];
    ZBracket zb_env;
    zb_env.iz0 = 0; zb_env.same = 0; zb_env.wa = 1;
    EnvExport X;
    X.valid = false; X.n00 = X.n11 = 0; X.iz0 = 0;
    X.combine = IS3D && SM == 1;
    // (DevWorld::f32pos -- the float32 element arrays of a run's first get_environment -- never reaches this launch: the host
    // takes the separate launches then, odr_step.hip; three more values live across the sample cost this kernel 0.61 -> 0.79 ms)
    // polar stereographic readers under a Runge-Kutta scheme: sines / cosines of the element's position once for the sample's
    // projection AND the stage positions (ODR_NO_SHARED_START: formed twice, as before round 6)
    ProjStart ps0;
#ifndef ODR_NO_SHARED_START
    if constexpr (PROJ == PROJ_STERE_POLAR && SCHEME != 0) {
      const DevSource &s0 = W->src[G.sid];
      double lw = lon;
      if (s0.lon_mode == 1) lw = np_mod(lw + 180.0, 360.0) - 180.0;
      else if (s0.lon_mode == 2) lw = np_mod(lw, 360.0);
      ps0 = proj_start(s0.proj, lw, lat);
    }
#endif
    env_group_fast<PROJ, true, IS3D>(*W, G, lon, lat, z, out, zt, zb_env, &X ODR_PT_ARG, 0, ps0);
    UVKeep<IS3D> K = uv_keep_from_sm<IS3D, SM>(G, X, th, zb_env, W->src[G.sid].nz, true);   // (FAST, 3-D: combined over the bracket's levels)
    if constexpr (STATE_LATE) load_state();
    ODR_PT_USE(out[0]); ODR_PT_USE(out[1]); ODR_PT_USE(out[2]); ODR_PT_USE(out[3]); ODR_PT_USE(out[4]); ODR_PT(2);
    const int id = (NOISE || MIXQ > 0) ? p.id[i] : 0;
    if (MIXQ > 0) vmix_col_fill<(MIXQ > 0 ? MIXQ : 1), MIXTL>(W->src[M.D.sid], M.D, lon, lat, Kp, threadIdx.x);
    if (NOISE && S.main_noise) add_current_noise(N, 0, i, p.n, id, out[0], out[1]);
#ifndef ODR_ABLATE_STORES   // what-if build (tools/ab_bench.sh)
#ifdef ODR_WHATIF_C3SPEC
#pragma unroll
    for (int k = 0; k < 5; ++k) G.out_ptr[k][i] = out[k];
#else
#pragma unroll
    for (int k = 0; k < MAXG; ++k)
      if (k < G.nv) G.out_ptr[k][i] = out[k];
#endif
    if (MIXQ == 0) {   // the sample position is what odr_vmix gathers its profiles at: not needed when the mixing is in here
      p.slon[i] = lon;
      p.slat[i] = lat;
    }
#endif
    // the movers' tests: the five values they need, read back from the arrays (just stored above, or sampled by the launch
    // before this one) -- requested here, used at the very end of the kernel.  Picking them out of the group's registers
    // by run-time slot cost 70 instructions of select chains in a kernel that is bound by instruction issue (C4: +258
    // instructions per wave with the first version, 0.59 -> 0.65 ms).
    float l_hd = 0.f, l_sx = 0.f, l_sy = 0.f, l_xw = 0.f, l_yw = 0.f;
    if (RED && S.red_on) {
      if (S.red_hd > -2) l_hd = p.env[VAR_HDIFF][i];
      if (S.red_sx > -2) { l_sx = p.env[VAR_SX][i]; l_sy = p.env[VAR_SY][i]; }
      if (S.red_xw > -2) { l_xw = p.env[VAR_XWIND][i]; l_yw = p.env[VAR_YWIND][i]; }
    }
    double zz = z;
    if (S.missing_code) {  // k_deactivate_missing
      bool miss = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < S.nmiss_grp) { const float e = pick_slot(out, S.miss_grp[k]); miss |= e != e; }
        if (k < S.nmiss_rest) { const float e = p.env[S.miss_rest[k]][i]; miss |= e != e; }
      }
      if (miss) {
        if (st == 0) p.status[i] = st = S.missing_code;
        p.moving[i] = moving = 0;
      }
    }
    if (S.coast_action) {  // k_coast
#ifdef ODR_WHATIF_C3SPEC
      const float land = out[4];
#else
      const float land = S.land_slot == 2 ? out[2] : p.env[VAR_LAND][i];
#endif
      if (land == 1.0f) {
        hit = true;
        if (S.coast_action == 1) {
          if (z <= 0) {
            if (st == 0) p.status[i] = st = S.stranded_code;
            p.moving[i] = moving = 0;
          }
        } else {
          if (S.seeded_code > 0 && age0 == 0.0f) {
            if (st == 0) p.status[i] = st = S.seeded_code;
            p.moving[i] = moving = 0;
          }
          lon = p.plon[i];
          lat = p.plat[i];
          ps0.ok = 0;                  // (another start point than the sample's)
          p.env[VAR_LAND][i] = 0.0f;   // self.environment.land_binary_mask[on_land] = 0 (:746)
        }
      }
    }
    if (S.seafloor) {  // k_seafloor
#ifdef ODR_WHATIF_C3SPEC
      const float dep = out[3];
      const float floorz = -__fadd_rn(dep, ssh0);
#else
      const float dep = S.depth_slot == 2 ? out[2] : (S.depth_slot == 3 ? out[3] : p.env[VAR_DEPTH][i]);
      const float floorz = -__fadd_rn(dep, S.ssh_slot >= 0 ? pick_slot(out, S.ssh_slot) : ssh0);
#endif
      if (zz < (double)floorz) { zz = (double)floorz; if (MIXQ == 0) p.z[i] = zz; }
    }
    if (S.age_dt != 0.0f) {  // k_age
      const float a = __fadd_rn(age0, S.age_dt);
      p.age[i] = a;
      if (S.max_age > 0 && a >= S.max_age) {
        if (st == 0) p.status[i] = st = S.retired_code;
        p.moving[i] = moving = 0;
      }
    }
    // deactivated (now or earlier, not yet compacted): the reference removes it before update() -- it does not move
    const bool skip = st != 0;
    if constexpr (IS3D && SM == 1) { if (zz != z) K.valid = false; }   // the sea floor lifted the element: another bracket than the kept values'
#ifndef ODR_ABLATE_STORES
    if (S.store_previous) { p.plon[i] = lon; p.plat[i] = lat; }
#endif
    ODR_PT(3);
    if (!skip) {
      const DevSource &s = W->src[G.sid];
      advect_grid_body<SCHEME, PROJ, IS3D, NOISE, SM, PARK>(s, s.slot[S.geo_slot_uv], lon, lat, zz, out[0], out[1],
                                                        __fmul_rn(current_factor(p, i, factor), cdf0), moving, dt, th, tf, uv_global(th), uv_global(tf),
                                                        W->fallback[VAR_U], W->fallback[VAR_V], N, i, p.n, id, K, zb_env, IS3D && zz == z, PARK ? s_park + threadIdx.x : nullptr ODR_PT_ARG, ps0);
    }
    ODR_PT_USE(lon); ODR_PT_USE(lat); ODR_PT(8);
    if (MIXQ > 0) {   // vertical_mixing + vertical_advection (oceandrift.py:397-571, :315-350) after the horizontal move
      double zn = zz;
      if (!skip) {
        const float dep = S.depth_slot == 2 ? out[2] : (S.depth_slot == 3 ? out[3] : p.env[VAR_DEPTH][i]);
        const float Zmin = __fmul_rn(-1.f, __fadd_rn(dep, S.ssh_slot >= 0 ? pick_slot(out, S.ssh_slot) : ssh0));  // float32 (:408)
        int sf_flags = 0;
        zn = vmix_col_walk<(MIXQ > 0 ? MIXQ : 1)>(W->src[M.D.sid], M.D.nzp, Kp, gsh, threadIdx.x, M.A, i, p.n, id, zz, moving, Zmin,
                                                  p.tv[i], sf_flags);
        if (sf_flags & 1) {   // deactivate_elements(reason='seafloor')
          if (st == 0) p.status[i] = M.A.sfl >> 8;
          p.moving[i] = 0;
        }
        if (sf_flags & 2) { lon = p.plon[i]; lat = p.plat[i]; }
        if (M.vadv >= 0 && (M.vadv ? zn <= 0 : zn < 0)) {
          const float wv = M.w_slot >= 0 ? pick_slot(out, M.w_slot) : p.env[VAR_W][i];
          const double zq = __dadd_rn(zn, __dmul_rn(__dmul_rn((double)moving, (double)wv), dt));
          zn = zq < 0 ? zq : 0.0;
        }
      }
      p.z[i] = zn;
    }
    // (2-D readers only: a 3-D run mixes vertically between this launch and the movers, which changes z and voids the tests;
    // the 3-D instantiations carry no code for it -- k_step_grid<RK4, lat/lon, 3-D> stays at 126 registers)
    if (RED && S.red_on && !skip) {   // k_reduce<false> for this element, as signs: every test is `maximum == 0`
      if (S.red_hd > -2) r_bits |= (l_hd > 0.f ? 1u : 0u) | (l_hd == 0.f ? 2u : 0u);
      if (S.red_sx > -2) { const float st_sum = __fadd_rn(l_sx, l_sy); r_bits |= (st_sum > 0.f ? 4u : 0u) | (st_sum == 0.f ? 8u : 0u); }
      if (S.red_xw > -2) {
        const double wdd = fabs(S.red_wdd);
        if (zz >= -wdd) {
          double wdf = wdf0;
          if (S.red_wdd != 0) {
            wdf = div_cr(wdf * (wdd + zz), wdd, S.red_iwdd);   // == wdf * (wdd + z) / wdd, correctly rounded (k_reduce)
            if (zz > 0) wdf = wdf0;
          }
          float xa = l_xw, ya = l_yw;
          if (S.red_rel) { xa = __fsub_rn(xa, out[0]); ya = __fsub_rn(ya, out[1]); }
          const float rws = speed_f32(xa, ya);
          r_bits |= 16u | (wdf > 0 ? 32u : 0u) | (wdf == 0 ? 64u : 0u) | (rws > 0.f ? 128u : 0u) | (rws == 0.f ? 256u : 0u);
        }
      }
    }
    {
      // (an opaque copy of the index: the addresses of lon / lat formed for the loads at the top would otherwise be carried
      // through the whole kernel for these two stores -- four registers of the 96 a fifth wave per SIMD leaves)
      long long i2 = i;
      asm volatile("" : "+v"(i2));
      p.lon[i2] = lon;
      p.lat[i2] = lat;
    }
#ifdef ODR_PHASE_TIMING
    pt_[9] = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && (blockIdx.x & 127) == 5) {   // a sample of the waves: the atomics must not load the memory system
      for (int k = 0; k < 9; ++k) atomicAdd(&g_phase[k], pt_[k + 1] > pt_[k] && pt_[k] ? pt_[k + 1] - pt_[k] : 0ull);
      for (int k = 10; k < 17; ++k) atomicAdd(&g_phase[k], pt_[k + 1] > pt_[k] && pt_[k] ? pt_[k + 1] - pt_[k] : 0ull);
      atomicAdd(&g_phase[31], 1ull);
    }
#endif
  }
  if (S.coast_action) {
    unsigned long long b = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_hit, (unsigned long long)__popcll(b));
  }
  if (!IS3D && MIXQ == 0 && S.red_on) {
    // one record of six doubles per WAVE, written by its lanes 0..5 in one store; k_red_finish folds the records into
    // red[].  A slot holds +1 (some element > 0), 0 (none > 0, some == 0) or -inf: what `maximum == 0` needs, from nine
    // wave votes -- not the maximum itself (the host marks the reduction `partial`; the wind speed slot nobody tests stays
    // -inf, the surface slot counts the elements).  (Atomics on red[] from here -- one per wave and slot, made only when they
    // would raise the value -- took this launch from 0.59 to 0.79-0.87 ms at 6.25 M elements: every wave ends on dependent,
    // coherent reads of the same lines.  True maxima by DPP reductions: ~80 instructions more per wave.)
    const unsigned long long b1 = __ballot(r_bits & 1u), b2 = __ballot(r_bits & 2u), b4 = __ballot(r_bits & 4u), b8 = __ballot(r_bits & 8u);
    const unsigned long long bs = __ballot(r_bits & 16u), b32 = __ballot(r_bits & 32u), b64 = __ballot(r_bits & 64u);
    const unsigned long long b128 = __ballot(r_bits & 128u), b256 = __ballot(r_bits & 256u);
    const unsigned lane = threadIdx.x & 63u;
    if (lane < 6u) {
      const double ninf = -__builtin_inf();
      auto sgn = [&](unsigned long long pos, unsigned long long zero) { return pos ? 1.0 : (zero ? 0.0 : ninf); };
      const double v = lane == 0 ? sgn(b1, b2) : lane == 1 ? sgn(b4, b8) : lane == 2 ? ninf
                     : lane == 3 ? sgn(b128, b256) : lane == 4 ? sgn(b32, b64) : (double)__popcll(bs);
      S.red[((size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 6 + lane] = v;
    }
  }
  if (MIXQ == 0 && S.wcount) {
    // (the status read back at the very end: carried through the kernel it would be one more register at its peak)
    int s2 = 1;
    long long i3 = i;
    asm volatile("" : "+v"(i3));
    if (i3 < p.n) s2 = p.status[i3];
    const unsigned long long kb = __ballot(s2 == 0);
    if (__ballot(s2 >= 100 && s2 < 164)) {   // rare: only while a reason waits for its first occurrence
      if (s2 >= 100 && s2 < 164) atomicOr(S.sflags, 1ull << (s2 - 100));
    }
    if ((threadIdx.x & 63) == 0 && i3 < p.n) S.wcount[i3 >> 6] = (unsigned)__popcll(kb);
  }
}

#endif  // ODR_TU_STEP
#ifdef ODR_TU_MISC
// analytic double gyre as the only source of the current (odr_field.hip.h "analytic double gyre, fast path")
__global__ __launch_bounds__(BLOCK) void k_env_gyre(const DevWorld *__restrict__ W, int sid, PView p, double snw,
                                                    int with_land, int record_prev) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  const DevSource &s = W->src[sid];
  const double lon = (W->f32pos & 1) ? lon_f32class(s.lon_mode, p.lon[i]) : p.lon[i], lat = p.lat[i];   // (DevWorld::f32pos)
  float u, v;
  const bool covered = gyre_sample(s, lon, lat, p.z[i], snw, W->fallback[VAR_U], W->fallback[VAR_V], u, v);
  p.env[VAR_U][i] = u;
  p.env[VAR_V][i] = v;
  // the reader reports land_binary_mask = 0 inside its domain (reader_double_gyre.py:78)
  if (with_land) p.env[VAR_LAND][i] = covered ? 0.0f : W->fallback[VAR_LAND];
  if (record_prev) { p.slon[i] = lon; p.slat[i] = lat; }
}

#endif  // ODR_TU_MISC
#ifdef ODR_TU_STEP
template <int SCHEME, bool NOISE>
__global__ __launch_bounds__(BLOCK) void k_advect_gyre(const DevWorld *__restrict__ W, int sid, PView p, double dt,
                                                       float factor, double snw_half, double snw_full, StageNoise N) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  const DevSource &s = W->src[sid];
  const float fbu = W->fallback[VAR_U], fbv = W->fallback[VAR_V];
  double lon = p.lon[i], lat = p.lat[i];
  const double z = p.z[i];
  const float u1 = p.env[VAR_U][i], v1 = p.env[VAR_V][i];
  const float f = __fmul_rn(current_factor(p, i, factor), p.cdf[i]);
  float fu, fv;
  GeodStart o = geod_start(lat, lon);
  if (SCHEME == 0) {
    fu = __fmul_rn(f, u1);
    fv = __fmul_rn(f, v1);
  } else {
    const float dtf = (float)dt;
    double lon2, lat2;
    float u2, v2;
    stage_pos_rt(N.sm, o, u1, v1, dtf, lon2, lat2);
    gyre_sample(s, lon2, lat2, z, snw_half, fbu, fbv, u2, v2);
    const int id = NOISE ? p.id[i] : 0;
    if (NOISE) add_current_noise(N, 1, i, p.n, id, u2, v2);
    if (SCHEME == 1) {
      fu = __fmul_rn(f, u2);
      fv = __fmul_rn(f, v2);
    } else {
      float u3, v3, u4, v4;
      stage_pos_rt(N.sm, o, u2, v2, dtf, lon2, lat2);
      gyre_sample(s, lon2, lat2, z, snw_half, fbu, fbv, u3, v3);
      if (NOISE) add_current_noise(N, 2, i, p.n, id, u3, v3);
      stage_pos_rt(N.sm, o, u3, v3, dtf, lon2, lat2);
      gyre_sample(s, lon2, lat2, z, snw_full, fbu, fbv, u4, v4);
      if (NOISE) add_current_noise(N, 3, i, p.n, id, u4, v4);
      fu = __fmul_rn(rk4_mix(u1, u2, u3, u4), f);
      fv = __fmul_rn(rk4_mix(v1, v2, v3, v4), f);
    }
  }
  move_f32_from(o, lon, lat, fu, fv, p.moving[i], dt);
  p.lon[i] = lon;
  p.lat[i] = lat;
}

#endif  // ODR_TU_STEP
#ifdef ODR_TU_MISC
// update_positions with velocities supplied by the caller
__global__ __launch_bounds__(BLOCK) void k_update_positions(PView p, const double *u, const double *v,
                                                            int is_f32, double dt) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  double lon = p.lon[i], lat = p.lat[i];
  if (is_f32) move_f32(lon, lat, (float)u[i], (float)v[i], p.moving[i], dt);
  else move_f64(lon, lat, u[i], v[i], p.moving[i], dt);
  p.lon[i] = lon;
  p.lat[i] = lat;
}

#endif  // ODR_TU_MISC
#ifdef ODR_TU_MISC
// min is stored as max of the negated value; red must be pre-filled with -inf (sums with 0).
// Grid-stride over a bounded grid, wave shuffle + LDS block reduction, ONE atomic per slot per
// workgroup (the first version issued one per wave: 8 ms of atomic contention for 6 M particles).
// EXT = false: without the lon / lat / z extremes (only odr_reduce_scalars / odr_reduce_local hand those out; the movers'
// early-outs do not read them): 24 bytes per particle and six maxima less
template <bool EXT>
__global__ __launch_bounds__(BLOCK) void k_reduce(PView p, double wind_drift_depth, int relative_wind,
                                                  double *red) {
  const double ninf = -__builtin_inf();
  double v[R_N];
#pragma unroll
  for (int k = 0; k < R_N; ++k) v[k] = ninf;
  v[R_NACT] = 0;
  v[R_NSURF] = 0;
  const double wdd = fabs(wind_drift_depth);
  // Every array the reduction reads is requested up front, for two elements per trip (the arrays present are the same for
  // every element: uniform branches): with the loads behind the status test and behind each other a trip was ~6 dependent
  // memory round trips and the launch 0.14 ms for 6.25 M elements (C4: 13 % of the step).
  const bool has_hd = p.env[VAR_HDIFF] != nullptr, has_st = p.env[VAR_SX] && p.env[VAR_SY], has_hs = p.env[VAR_HS] != nullptr;
  const bool has_tp = p.env[VAR_TP] != nullptr, has_mld = p.env[VAR_MLD] != nullptr, has_w = p.env[VAR_XWIND] && p.env[VAR_YWIND];
  const bool rel = relative_wind && p.env[VAR_U] && p.env[VAR_V];
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i0 = (long long)blockIdx.x * BLOCK + threadIdx.x; i0 < p.n; i0 += 2 * stride) {
    int st[2]; double z[2], lo[2], la[2]; float hd[2], sx[2], sy[2], hs[2], tp[2], mld[2], xw[2], yw[2], wdf0[2], cu[2], cv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long long i = i0 + u * stride;
      const bool in = i < p.n;
      const long long j = in ? i : i0;
      st[u] = in ? p.status[j] : 1;
      z[u] = p.z[j];
      if (EXT) { lo[u] = p.lon[j]; la[u] = p.lat[j]; }
      hd[u] = has_hd ? p.env[VAR_HDIFF][j] : 0.f;
      sx[u] = has_st ? p.env[VAR_SX][j] : 0.f; sy[u] = has_st ? p.env[VAR_SY][j] : 0.f;
      hs[u] = has_hs ? p.env[VAR_HS][j] : 0.f; tp[u] = has_tp ? p.env[VAR_TP][j] : 0.f; mld[u] = has_mld ? p.env[VAR_MLD][j] : 0.f;
      xw[u] = has_w ? p.env[VAR_XWIND][j] : 0.f; yw[u] = has_w ? p.env[VAR_YWIND][j] : 0.f; wdf0[u] = has_w ? p.wdf[j] : 0.f;
      cu[u] = rel ? p.env[VAR_U][j] : 0.f; cv[u] = rel ? p.env[VAR_V][j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // elements deactivated in this step and not yet compacted do not count: the reference has removed them by the time
      // its movers reduce (a sharded run reduces once per step, before the compaction: odr_reduce_local)
      if (st[u] != 0) continue;
      v[R_NACT] += 1;
      if (EXT) {
        v[R_LONMIN] = fmax(v[R_LONMIN], -lo[u]); v[R_LONMAX] = fmax(v[R_LONMAX], lo[u]);
        v[R_LATMIN] = fmax(v[R_LATMIN], -la[u]); v[R_LATMAX] = fmax(v[R_LATMAX], la[u]);
        v[R_ZMIN] = fmax(v[R_ZMIN], -z[u]); v[R_ZMAX] = fmax(v[R_ZMAX], z[u]);
      }
      if (has_hd) v[R_DMAX] = fmax(v[R_DMAX], (double)hd[u]);
      if (has_st) v[R_STOKESMAX] = fmax(v[R_STOKESMAX], (double)__fadd_rn(sx[u], sy[u]));
      if (has_hs) v[R_HSMAX] = fmax(v[R_HSMAX], (double)hs[u]);
      if (has_tp) v[R_TPMAX] = fmax(v[R_TPMAX], (double)tp[u]);
      if (has_mld) v[R_MLDMAX] = fmax(v[R_MLDMAX], (double)mld[u]);
      if (has_w) {
        // advect_wind bookkeeping (physics_methods.py:738-775)
        const bool surf = z[u] >= -wdd;
        if (surf) {
          float xa = xw[u], ya = yw[u];
          double wdf = wdf0[u];
          if (wind_drift_depth != 0) {
            wdf = wdf * (wdd + z[u]) / wdd;
            if (z[u] > 0) wdf = wdf0[u];
          }
          v[R_NSURF] += 1;
          v[R_WDFMAX] = fmax(v[R_WDFMAX], wdf);
          v[R_WSPEEDMAX] = fmax(v[R_WSPEEDMAX], (double)speed_f32(xa, ya));
          if (rel) { xa = __fsub_rn(xa, cu[u]); ya = __fsub_rn(ya, cv[u]); }
          v[R_RELWSPEEDMAX] = fmax(v[R_RELWSPEEDMAX], (double)speed_f32(xa, ya));
        }
      }
    }
  }
  __shared__ double sh[BLOCK / 64][R_N];
#pragma unroll
  for (int k = 0; k < R_N; ++k) {
    bool is_sum = (k == R_NACT || k == R_NSURF);
    double r = is_sum ? wave_sum(v[k]) : wave_max(v[k]);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = r;
  }
  __syncthreads();
  if (threadIdx.x < R_N) {
    int k = threadIdx.x;
    bool is_sum = (k == R_NACT || k == R_NSURF);
    double r = sh[0][k];
    for (int w = 1; w < BLOCK / 64; ++w) r = is_sum ? r + sh[w][k] : fmax(r, sh[w][k]);
    if (is_sum) { if (r != 0) atomicAdd(&red[k], r); }
    else if (r > ninf) atomic_max_d(&red[k], r);
  }
}

// the per-wave records of a step launch (StepDesc.red_on: {D, Stokes sum, wind speed, relative wind speed, wind drift factor
// maxima; elements at the surface}) folded into red[] (initialised by k_red_init): grid-stride, one atomic per slot and workgroup
__global__ __launch_bounds__(BLOCK) void k_red_finish(const double *__restrict__ rec, long long nrec, double *red) {
  const double ninf = -__builtin_inf();
  double m[5] = {ninf, ninf, ninf, ninf, ninf}, ns = 0;
  for (long long r = (long long)blockIdx.x * BLOCK + threadIdx.x; r < nrec; r += (long long)gridDim.x * BLOCK) {
    const double *q = rec + 6 * r;
#pragma unroll
    for (int k = 0; k < 5; ++k) m[k] = fmax(m[k], q[k]);
    ns += q[5];
  }
  __shared__ double sh[BLOCK / 64][6];
#pragma unroll
  for (int k = 0; k < 5; ++k) { const double w = wave_max(m[k]); if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = w; }
  { const double w = wave_sum(ns); if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][5] = w; }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    double r = sh[0][k];
    for (int w = 1; w < BLOCK / 64; ++w) r = k == 5 ? r + sh[w][k] : fmax(r, sh[w][k]);
    const int slot = k == 0 ? R_DMAX : k == 1 ? R_STOKESMAX : k == 2 ? R_WSPEEDMAX : k == 3 ? R_RELWSPEEDMAX : k == 4 ? R_WDFMAX : R_NSURF;
    if (k == 5) { if (r != 0) atomicAdd(&red[slot], r); }
    else if (r > ninf) atomic_max_d(&red[slot], r);
  }
}

__global__ void k_red_init(double *red) {
  int k = threadIdx.x;
  if (k < R_N) red[k] = (k == R_NACT || k == R_NSURF) ? 0.0 : -__builtin_inf();
}

#endif  // ODR_TU_MISC
#ifdef ODR_TU_MISC
// ------------------------------------------------------------------- wind / Stokes
// advect_wind (physics_methods.py:712-791).  red[] carries the global early-out tests.
// wind drift velocity of element i (physics_methods.py:749-791), float64 like the reference's products
__device__ __forceinline__ void wind_velocity(const PView &p, long long i, double z, double wind_drift_depth, int relative_wind,
                                              double factor, double &xu, double &xv) {
  double wdd = fabs(wind_drift_depth);
  bool surf = z >= -wdd;
  double wdf = p.wdf[i];
  if (wind_drift_depth != 0) {
    wdf = wdf * (wdd + z) / wdd;  // float64 (:756)
    if (z > 0) wdf = p.wdf[i];
  }
  if (!surf) wdf = 0.0;
  float xw = p.env[VAR_XWIND][i], yw = p.env[VAR_YWIND][i];
  if (relative_wind) {
    xw = __fsub_rn(xw, p.env[VAR_U][i]);
    yw = __fsub_rn(yw, p.env[VAR_V][i]);
  }
  if (p.ice == 1) factor = (double)__fsub_rn(1.0f, ice_k(p.env[VAR_ICE_A][i]));   // x_wind*wdf*factor: float64 * float32 array
  xu = __dmul_rn(__dmul_rn((double)xw, wdf), factor);
  xv = __dmul_rn(__dmul_rn((double)yw, wdf), factor);
}
__global__ __launch_bounds__(BLOCK) void k_advect_wind(PView p, double dt, double wind_drift_depth,
                                                       int relative_wind, double factor,
                                                       const double *__restrict__ red) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  if (red[R_NSURF] == 0 || red[R_WDFMAX] == 0 || red[R_RELWSPEEDMAX] == 0) return;  // :741-747, :775-780
  double lon = p.lon[i], lat = p.lat[i], z = p.z[i];
  double xu, xv;
  wind_velocity(p, i, z, wind_drift_depth, relative_wind, factor, xu, xv);
  move_f64(lon, lat, xu, xv, p.moving[i], dt);
  p.lon[i] = lon;
  p.lat[i] = lat;
}

// NumPy dtype classes of a stokes-profile operand: python scalar (weak), float32, float64
struct TVal { double v; int k; };
__device__ __forceinline__ TVal tv_(double v, int k) { TVal r; r.v = v; r.k = k; return r; }
__device__ __forceinline__ TVal tmul(TVal a, TVal b) {
  TVal r; r.k = a.k > b.k ? a.k : b.k;
  r.v = r.k == 1 ? (double)__fmul_rn((float)a.v, (float)b.v) : __dmul_rn(a.v, b.v);
  return r;
}
__device__ __forceinline__ TVal tdiv(TVal a, TVal b) {
  TVal r; r.k = a.k > b.k ? a.k : b.k;
  r.v = r.k == 1 ? (double)__fdiv_rn((float)a.v, (float)b.v) : __ddiv_rn(a.v, b.v);
  return r;
}

// one profile function of physics_methods.py:336-416 for one element: float32 surface components, wave height and period
// with their NumPy dtype classes, depth -> (stokes_u, stokes_v) in float64
__device__ __forceinline__ void stokes_profile(int profile, float sx, float sy, TVal H, TVal T, double z, double &su, double &sv) {
  const float speed = speed_f32(sx, sy);
  TVal mwf = tdiv(tv_(2. * kPi, 0), T);
  TVal transport = tdiv(tmul(mwf, tmul(H, H)), tv_(16, 0));
  TVal num = tv_(speed, 1);
  if (profile == 2) num = tmul(num, tv_(1 - 2 * 1.0 / 3, 0));
  TVal km = tdiv(num, tmul(tv_(2, 0), transport));
  // Elements at the surface (z = 0: every 2-D run): all arguments of the profile functions are (signed) zeros and the unit
  // profile is exp(0) [/ (1 - 0)] [- sqrt(0) erfc(sqrt(0))] = 1 exactly -- taken without the calls (same bits; a NaN or an
  // infinite wavenumber makes the products NaN, fails the tests and goes through the functions).
  double az = fabs(z), unit;
  if (profile == 0) {
    const double a = __dmul_rn(tmul(tv_(2, 0), km).v, z);
    unit = a == 0 ? 1.0 : exp(a);
  } else if (profile == 1) {
    TVal ke = tdiv(km, tv_(3, 0));
    const double a = __dmul_rn(tmul(tv_(2.0, 0), ke).v, z), b = __dmul_rn(tmul(tv_(8.0, 0), ke).v, z);
    unit = (a == 0 && b == 0) ? 1.0 : exp(a) / (1.0 - b);
  } else {
    double k2 = tmul(tv_(2, 0), km).v, c2 = tmul(tv_(2 * kPi, 0), km).v;
    const double a = __dmul_rn(k2, z), b = __dmul_rn(c2, az), d = __dmul_rn(k2, az);
    unit = (a == 0 && b == 0 && d == 0) ? 1.0 : __dsub_rn(exp(a), __dmul_rn(sqrt(b), erfc(sqrt(d))));
  }
  su = speed == 0 ? 0.0 : __dmul_rn((double)sx, unit);
  sv = speed == 0 ? 0.0 : __dmul_rn((double)sy, unit);
}

// stokes_drift (physics_methods.py:793-848) with the Breivik profiles (:336-416); profile 3 = 'windsea_swell'
// (stokes_drift_profile_windsea_swell :418-456, Breivik & Christensen 2020): the surface drift split into a swell part along
// the swell direction (monochromatic profile, swell height / period) and the wind-sea rest (Phillips profile, wind-sea
// height / period); unit vectors and split in float32 like NumPy on the float32 environment, profiles in float64
__device__ __forceinline__ void stokes_velocity(const PView &p, long long i, double z, int profile, int hs_mode, int tp_mode,
                                                double factor, double &su, double &sv) {
  float sx = p.env[VAR_SX][i], sy = p.env[VAR_SY][i];
  if (profile == 3) {
    const float rws = __fmul_rn(p.env[VAR_WW_DIR][i], (float)(kPi / 180.)), rsw = __fmul_rn(p.env[VAR_SWELL_DIR][i], (float)(kPi / 180.));
    // np.cos / np.sin of a float32 array: float32 results.  Rounded from the float64 functions (= correctly rounded float32
    // in all but ~1e-9 of the cases) so that device and CPU oracle hold the same unit vectors; NumPy's own float32 loops
    // differ from that by one ulp in 17 % of the values (tests/test_oracle_golden.py, c18)
    double sd, cd;
    sincos((double)rws, &sd, &cd);
    const float ws_n = (float)cd, ws_e = (float)sd;
    sincos((double)rsw, &sd, &cd);
    const float sw_n = (float)cd, sw_e = (float)sd;
    const float numr = __fsub_rn(__fmul_rn(sx, ws_n), __fmul_rn(sy, ws_e));
    const float den = __fsub_rn(__fmul_rn(sw_e, ws_n), __fmul_rn(sw_n, ws_e));
    const float sp = __fdiv_rn(numr, den);
    const float swu = __fmul_rn(sp, sw_e), swv = __fmul_rn(sp, sw_n);
    const float wu = __fsub_rn(sx, swu), wv = __fsub_rn(sy, swv);
    double u1, v1, u2, v2;
    stokes_profile(0, swu, swv, tv_(p.env[VAR_SWELL_HS][i], 1), tv_(p.env[VAR_SWELL_TP][i], 1), z, u1, v1);
    stokes_profile(2, wu, wv, tv_(p.env[VAR_WW_HS][i], 1), tv_(p.env[VAR_WW_TM][i], 1), z, u2, v2);
    su = __dadd_rn(u1, u2);
    sv = __dadd_rn(v1, v2);
  } else {
    float ws = 0.f;
    if (hs_mode == 1 || tp_mode == 1 || tp_mode == 3) ws = speed_f32(p.env[VAR_XWIND][i], p.env[VAR_YWIND][i]);
    TVal H, T;
    if (hs_mode == 0) H = tv_(p.env[VAR_HS][i], 1);
    else if (hs_mode == 1) H = tv_(__fmul_rn((float)0.0246, __fmul_rn(ws, ws)), 1);
    else H = tv_(1, 0);
    if (tp_mode == 0) T = tv_(p.env[VAR_TP][i], 1);
    else if (tp_mode == 1 || tp_mode == 3) {
      double omega = 5;
      if (ws > 0) omega = __fdiv_rn((float)(0.877 * 9.81), __fmul_rn((float)1.17, ws));
      T = tv_(__ddiv_rn(2 * kPi, omega), 2);
      // a model that has the wave period among its variables (OpenOil) reads it back from the float32 environment
      // (calculate_missing_environment_variables, physics_methods.py:876-883)
      if (tp_mode == 3) T = tv_((float)T.v, 1);
    } else T = tv_(8, 0);
    stokes_profile(profile, sx, sy, H, T, z, su, sv);
  }
  if (p.ice == 2) factor = (double)ice_stokes_factor(p.env[VAR_ICE_A][i]);   // stokes_u*factor: float64 * float32 array
  su = __dmul_rn(su, factor);
  sv = __dmul_rn(sv, factor);
}
__global__ __launch_bounds__(BLOCK) void k_stokes(PView p, double dt, int profile, int hs_mode,
                                                  int tp_mode, double factor,
                                                  const double *__restrict__ red) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  if (red[R_STOKESMAX] == 0) return;  // "No Stokes drift velocity available" (:799-804)
  double lon = p.lon[i], lat = p.lat[i];
  double su, sv;
  stokes_velocity(p, i, p.z[i], profile, hs_mode, tp_mode, factor, su, sv);
  move_f64(lon, lat, su, sv, p.moving[i], dt);
  p.lon[i] = lon;
  p.lat[i] = lat;
}

// advect_with_sea_ice (physics_methods.py:693-710): update_positions(factor*ice_u, factor*ice_v), float32 products;
// without sea_ice_x/y_velocity the rule of thumb current + 1.5 % of the wind (Nordam et al. 2019)
__global__ __launch_bounds__(BLOCK) void k_advect_ice(PView p, double dt, float factor, int have_ice_velocity) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  const float f = p.ice == 3 ? ice_k(p.env[VAR_ICE_A][i]) : factor;
  float iu, iv;
  if (have_ice_velocity) { iu = p.env[VAR_ICE_U][i]; iv = p.env[VAR_ICE_V][i]; }
  else {   // x_sea_water_velocity + 0.015*x_wind: float32
    iu = __fadd_rn(p.env[VAR_U][i], __fmul_rn(0.015f, p.env[VAR_XWIND][i]));
    iv = __fadd_rn(p.env[VAR_V][i], __fmul_rn(0.015f, p.env[VAR_YWIND][i]));
  }
  double lon = p.lon[i], lat = p.lat[i];
  move_f32(lon, lat, __fmul_rn(f, iu), __fmul_rn(f, iv), p.moving[i], dt);
  p.lon[i] = lon;
  p.lat[i] = lat;
}

#endif  // ODR_TU_MISC
#ifdef ODR_TU_MISC
// horizontal_diffusion (basemodel/__init__.py:1746-1772)
__device__ __forceinline__ void hdiff_velocity(const PView &p, long long i, int moving, double dt, int rng_mode,
                                               const double *__restrict__ hnx, const double *__restrict__ hny,
                                               unsigned long long seed, unsigned long long step, double &xu, double &xv) {
  double nx, ny;
  if (rng_mode == 1) { nx = hnx[i]; ny = hny[i]; }
  else {
    const double2 g = rng_normal2(rng_block(seed, p.id[i], step, RNG_OFF_HDIFF));
    nx = g.x; ny = g.y;
  }
  float s = sqrtf(__fdiv_rn(__fmul_rn(2.0f, p.env[VAR_HDIFF][i]), (float)fabs(dt)));
  xu = __dmul_rn(__dmul_rn((double)moving, (double)s), nx);
  xv = __dmul_rn(__dmul_rn((double)moving, (double)s), ny);
}
__global__ __launch_bounds__(BLOCK) void k_hdiff(PView p, double dt, int rng_mode,
                                                 const double *__restrict__ hnx,
                                                 const double *__restrict__ hny,
                                                 unsigned long long seed, unsigned long long step,
                                                 const double *__restrict__ red) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  if (red[R_DMAX] == 0) return;  // "Horizontal diffusivity is 0, no random walk." (:1754)
  double lon = p.lon[i], lat = p.lat[i];
  const int moving = p.moving[i];
  double xu, xv;
  hdiff_velocity(p, i, moving, dt, rng_mode, hnx, hny, seed, step, xu, xv);
  move_f64(lon, lat, xu, xv, moving, dt);
  p.lon[i] = lon;
  p.lat[i] = lat;
}

// advect_wind -> stokes_drift -> horizontal_diffusion of one step in ONE launch (physics_methods.py:712-848,
// basemodel/__init__.py:1746-1772), in the reference's order: each is an update_positions from where the previous one
// left the element, the element stays in registers in between (three kernels re-read and re-write lon / lat / z / moving
// and the environment: C4 0.30 ms of three launches).  `which`: 1 wind, 2 Stokes drift, 4 diffusion; a mover whose global
// early-out holds (red[]: no element at the surface, wind / Stokes drift / diffusivity identically zero) is skipped as a
// whole -- also its update_positions, which would renormalise the longitude.  Same arithmetic as the three kernels except for the
// start-point coefficients of the second and third move (move_f64_chain: equal to float64 round-off, <= 1 ulp of the position).
struct MoversDesc {
  int which, relative_wind, profile, hs_mode, tp_mode, rng_mode;
  double dt, wind_drift_depth, wind_factor, stokes_factor;
  const double *hnx, *hny;
  unsigned long long seed, step;
};
__global__ __launch_bounds__(BLOCK) void k_movers(PView p, MoversDesc M, const double *__restrict__ red) {
  const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  const bool wind = (M.which & 1) && !(red[R_NSURF] == 0 || red[R_WDFMAX] == 0 || red[R_RELWSPEEDMAX] == 0);
  const bool stokes = (M.which & 2) && !(red[R_STOKESMAX] == 0);
  const bool hdiff = (M.which & 4) && !(red[R_DMAX] == 0);
  if (!(wind || stokes || hdiff)) return;
  double lon = p.lon[i], lat = p.lat[i];
  const double z = p.z[i];
  const int moving = p.moving[i];
  double xu, xv;
  MoveChain mc;
  mc.next = false;
  if (wind) {
    wind_velocity(p, i, z, M.wind_drift_depth, M.relative_wind, M.wind_factor, xu, xv);
    move_f64_chain(mc, lon, lat, xu, xv, moving, M.dt);
  }
  if (stokes) {
    stokes_velocity(p, i, z, M.profile, M.hs_mode, M.tp_mode, M.stokes_factor, xu, xv);
    move_f64_chain(mc, lon, lat, xu, xv, moving, M.dt);
  }
  if (hdiff) {
    hdiff_velocity(p, i, moving, M.dt, M.rng_mode, M.hnx, M.hny, M.seed, M.step, xu, xv);
    move_f64_chain(mc, lon, lat, xu, xv, moving, M.dt);
  }
  p.lon[i] = lon;
  p.lat[i] = lat;
}

// drift:current_uncertainty / wind_uncertainty (environment.py:869-891)
__global__ __launch_bounds__(BLOCK) void k_env_noise(PView p, int vx, int vy, double std, int dist, int rng_mode,
                                                     const double *__restrict__ hnx,
                                                     const double *__restrict__ hny,
                                                     unsigned long long seed, unsigned long long step) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  double nx, ny;
  if (rng_mode == 1) { nx = hnx[i]; ny = hny[i]; }  // np.random.normal(0, std, N) / uniform(-std, std, N): already scaled
  else if (dist == 0) {
    const double2 g = rng_normal2(rng_block(seed, p.id[i], step, RNG_OFF_NOISE + 4ull * (unsigned)vx));
    nx = g.x * std; ny = g.y * std;
  } else {   // drift:current_uncertainty_uniform (environment.py:880-886)
    const double2 q = rng_uniform2(rng_block(seed, p.id[i], step, RNG_OFF_NOISE + RNG_OFF_NOISE_UNIFORM + 4ull * (unsigned)vx));
    nx = fma(2.0 * std, q.x, -std); ny = fma(2.0 * std, q.y, -std);
  }
  // float32 array += float64 array: computed in float64, cast back to float32
  p.env[vx][i] = (float)__dadd_rn((double)p.env[vx][i], nx);
  p.env[vy][i] = (float)__dadd_rn((double)p.env[vy][i], ny);
}

#endif  // ODR_TU_MISC
}  // namespace odr
#include "odr_oil.hip.h"
namespace odr {

#ifdef ODR_TU_MIX
// ---------------------------------------------------------------- vertical mixing
// OceanDrift.vertical_mixing (oceandrift.py:397-571), diffusivity model 'environment'.
// The diffusivity profile of each particle (all block levels at the position of the last
// environment sample, time-interpolated in float64: structured.py:366-385) is gathered once
// -- z-innermost columns, 16-byte loads -- into LDS ([level][thread]: bank = thread, conflict
// free for per-thread dynamic level indices) together with -dK/dz per level, and the whole
// ntimes_mix random walk runs out of registers + LDS.  vertical_advection (:315-350) is fused
// at the end when `vadv` >= 0 (same particle, same z).
__device__ __forceinline__ void kcolumn(const float *__restrict__ col, int nz, float *out /*[MAXNZ]*/) {
  int k = 0;
  for (; k + 4 <= nz; k += 4) {
    F4 q = *(const F4 *)(col + k);
    out[k] = q.x; out[k + 1] = q.y; out[k + 2] = q.z; out[k + 3] = q.w;
  }
  for (; k < nz; ++k) out[k] = col[k];
}

template <int NZMAX, bool OIL = false>
__global__ __launch_bounds__(BLOCK) void k_vmix(const DevWorld *__restrict__ W, PView p, double t,
                                                double dt, double dt_mix_cfg, int mix_at_surface,
                                                int rng_mode, const double *__restrict__ huni,
                                                unsigned long long seed, unsigned long long step,
                                                int vadv /* -1 none, 0 below surface, 1 incl. surface */, int sfl_in,
                                                OilArgs oa = OilArgs()) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  const int tid = threadIdx.x;
  // K source: first GRID reader of the priority list, else the fallback constant
  const DevSource *src = nullptr;
  for (int k = 0; k < W->nlist[VAR_KZ]; ++k) {
    const DevSource &s = W->src[W->list[VAR_KZ][k]];
    if (s.kind == SRC_GRID) { src = &s; break; }
  }
  // (bits 16 .. 23 of sfl_in: odr_vmix_set_profile_levels -- the columns end there, the level search, the gradient's edge and
  // the clamp of deeper elements work on the cut column; the member stride below stays the reader's level count)
  const int sfl = sfl_in & 0xffff, cut = (sfl_in >> 16) & 0xff;
  const int nz_full = src ? (src->nz > 1 ? src->nz : 1) : 1;
  const int nzp = (cut > 0 && cut < nz_full) ? cut : nz_full;
  // ensemble diffusivity: the levels of member m follow those of member m - 1 along the layer axis (odr_source_set_members)
  const int kmembers = src ? src->members[VAR_KZ] : 0;
  double *Kp = (double *)smem;           // [nzp][BLOCK]
  double *gsh = Kp + (size_t)nzp * BLOCK;  // [3][nzp] second-order np.gradient coefficients per level
  const float Kfb = W->fallback[VAR_KZ];
  if (tid < nzp && src) {
    gsh[tid] = src->vg_a[tid]; gsh[nzp + tid] = src->vg_b[tid]; gsh[2 * nzp + tid] = src->vg_c[tid];
  }
  __syncthreads();
  if (i >= p.n) return;  // no barrier below: every thread touches only its own LDS column
  {
    bool cov = false;
    double xi = 0, yi = 0, wgt = 0;
    int ib = 0, ia = -1;
    const size_t moff = (kmembers > 1 && p.aux[AUX_KMEMBER]) ? (size_t)((int)p.aux[AUX_KMEMBER][i]) * (size_t)nz_full : 0;
    if (src) {
      double lon = p.slon[i], lat = p.slat[i], x, y;
      if (src->lon_mode == 1) lon = np_mod(lon + 180.0, 360.0) - 180.0;
      else if (src->lon_mode == 2) lon = np_mod(lon, 360.0);
      proj_fwd_rt(src->proj, lon, lat, x, y);
      cov = x >= src->xmin && x <= src->xmax && y >= src->ymin && y <= src->ymax;
      if (src->mod360_x) x = np_mod(x, 360.0);
      bracket(*src, t, ib, ia);
      const DevBlock &bb = src->slot[ib];
      // (bit 1 of DevWorld::f32pos: the profiles of this step were sampled in the float32 position class -- the first
      // get_environment of a run -- on a geographic reader with float32 coordinate arrays: float32 index maps, index_f32)
      const int f32idx = ((W->f32pos & 2) && src->proj.kind == PROJ_LATLONG) ? src->xy_f32 : 0;
      xi = (f32idx & 1) ? index_f32(x, bb.x0, bb.xspan, bb.nx - 1) : __dmul_rn(div_cr(x - bb.x0, bb.xspan, bb.ixspan), (double)(bb.nx - 1));
      yi = (f32idx & 2) ? index_f32(y, bb.y0, bb.yspan, bb.ny - 1) : __dmul_rn(div_cr(y - bb.y0, bb.yspan, bb.iyspan), (double)(bb.ny - 1));
      if (ia >= 0) wgt = __ddiv_rn(t - bb.t, src->slot[ia].t - bb.t);
    }
    if (src && cov && src->slot[ib].es[VAR_KZ] == 1 && NZMAX > 1) {
      const DevBlock &bb = src->slot[ib];
      const int ny = bb.ny, nx = bb.nx;
      const Axis ay = axis_fp(yi, ny), ax = axis_fp(xi, nx);
      const int y0 = ay.i0, y1 = ay.i1, x0 = ax.i0, x1 = ax.i1;
      const double ty = ay.t, tx = ax.t, wy0 = 1 - ty, wx0 = 1 - tx;
      const size_t rec = (size_t)bb.rec;
      size_t o00 = ((size_t)y0 * nx + x0) * rec, o01 = ((size_t)y0 * nx + x1) * rec;
      size_t o10 = ((size_t)y1 * nx + x0) * rec, o11 = ((size_t)y1 * nx + x1) * rec;
      float c00[NZMAX], c01[NZMAX], c10[NZMAX], c11[NZMAX];
      const float *d = bb.data[VAR_KZ] + moff;
      kcolumn(d + o00, nzp, c00); kcolumn(d + o01, nzp, c01);
      kcolumn(d + o10, nzp, c10); kcolumn(d + o11, nzp, c11);
#pragma unroll
      for (int k = 0; k < NZMAX; ++k)
        if (k < nzp) Kp[k * BLOCK + tid] = (double)bil4(c00[k], c01[k], c10[k], c11[k], wy0, ty, wx0, tx);
      if (ia >= 0) {
        const float *da = src->slot[ia].data[VAR_KZ] + moff;
        kcolumn(da + o00, nzp, c00); kcolumn(da + o01, nzp, c01);
        kcolumn(da + o10, nzp, c10); kcolumn(da + o11, nzp, c11);
#pragma unroll
        for (int k = 0; k < NZMAX; ++k)
          if (k < nzp) {
            double v1 = (double)bil4(c00[k], c01[k], c10[k], c11[k], wy0, ty, wx0, tx);
            Kp[k * BLOCK + tid] = __dadd_rn(__dmul_rn(Kp[k * BLOCK + tid], 1 - wgt), __dmul_rn(v1, wgt));
          }
      }
      for (int k = 0; k < nzp; ++k)
        if (!isfinite(Kp[k * BLOCK + tid])) Kp[k * BLOCK + tid] = Kfb;
    } else {
      for (int k = 0; k < nzp; ++k) {
        double val = Kfb;
        if (src && cov) {
          const DevBlock &bb = src->slot[ib];
          const size_t ns = (size_t)bb.rec;
          double v0 = bilinear_f32(bb.data[VAR_KZ] + (moff + (size_t)k) * bb.es[VAR_KZ], bb.ny, bb.nx, ns, yi, xi), vv;
          if (ia >= 0) {
            const DevBlock &ba = src->slot[ia];
            double v1 = bilinear_f32(ba.data[VAR_KZ] + (moff + (size_t)k) * ba.es[VAR_KZ], ba.ny, ba.nx, ns, yi, xi);
            vv = __dadd_rn(__dmul_rn(v0, 1 - wgt), __dmul_rn(v1, wgt));
          } else vv = v0;
          if (isfinite(vv)) val = vv;
        }
        Kp[k * BLOCK + tid] = val;
      }
    }
  }
  // Level of a particle: zi = round(interp1d(-mixing_z, range)(-z)) (oceandrift.py:485-488,513).
  // The rounded linear index only changes at the mid-depths between levels, so zi is the
  // number of mid-depths passed (ties: np.round is half-to-even) -- a compare chain on
  // wave-uniform constants instead of a dependent table walk.  -dK/dz*dt_mix and
  // sqrt(K*|dt_mix|*2/r) of the current level are re-derived only when zi changes.
  const bool uniform_z = src ? src->vg_uniform != 0 : true;
  const double sgn = dt > 0 ? 1.0 : (dt < 0 ? -1.0 : 0.0);
  const double dt_mix = dt_mix_cfg * sgn;
  const int ntimes = abs((int)(dt / dt_mix));
  const double r = 1.0 / 3, ir = 1.0 / r;
  double z = p.z[i];
  int moving = p.moving[i];
  int sf_flags = 0;   // 1: deactivated on the sea floor, 2: moved back horizontally (general:seafloor_action)
  const float Zmin = __fmul_rn(-1.f, __fadd_rn(p.env[VAR_DEPTH][i], p.env[VAR_SSH][i]));  // float32 (:408)
  // w*dt_mix*moving: dt_mix is a NumPy float64 scalar (np.sign, oceandrift.py:416) -> float64 product under NumPy 2
  double wstep = __dmul_rn(__dmul_rn((double)p.tv[i], dt_mix), (double)moving);
  const MixKey st = mix_key(seed, step, rng_mode == 0 ? p.id[i] : 0);
  uint4 u4 = make_uint4(0u, 0u, 0u, 0u);
  OilLane oil;
  if (OIL) oil.init(p, i, oa);
  int zi_cur = -1;
  double sig = 0, dKdt = 0;
  for (int it = 0; it < ntimes; ++it) {
    const bool surface = z == 0;
    if (OIL)   // update_terminal_velocity at the top of every sub-step (oceandrift.py:509), w*dt_mix*moving (:548)
      wstep = __dmul_rn(__dmul_rn(oil.terminal_velocity(), dt_mix), (double)moving);
    const double d = -z;
    int zi = 0;
#pragma unroll
    for (int k = 0; k < (NZMAX > 1 ? NZMAX - 1 : MAXNZ - 1); ++k)
      if (k < nzp - 1) {
        const double mid = src->zmid[k];   // wave-uniform (scalar load)
        zi += ((k & 1) ? d >= mid : d > mid) ? 1 : 0;
      }
    if (zi != zi_cur) {
      zi_cur = zi;
      const double Kz = Kp[zi * BLOCK + tid];
      double gK = 0;  // np.gradient(Kprofiles, mixing_z, axis=0)[zi] (oceandrift.py:501)
      if (nzp >= 2) {
        // divisors are level constants: host reciprocals + exact-residual correction (div_cr)
        if (zi == 0) gK = div_cr(Kp[BLOCK + tid] - Kz, src->vg_d[0], src->vg_id[0]);
        else if (zi == nzp - 1) {
          if (nzp == nz_full) gK = div_cr(Kz - Kp[(nzp - 2) * BLOCK + tid], src->vg_d[1], src->vg_id[1]);
          else gK = __ddiv_rn(Kz - Kp[(nzp - 2) * BLOCK + tid], src->z[nzp - 1] - src->z[nzp - 2]);   // np.gradient's edge of the CUT grid
        } else if (uniform_z)
          gK = div_cr(Kp[(zi + 1) * BLOCK + tid] - Kp[(zi - 1) * BLOCK + tid], src->vg_d[2], src->vg_id[2]);
        else
          gK = __dadd_rn(__dadd_rn(__dmul_rn(gsh[zi], Kp[(zi - 1) * BLOCK + tid]), __dmul_rn(gsh[nzp + zi], Kz)),
                         __dmul_rn(gsh[2 * nzp + zi], Kp[(zi + 1) * BLOCK + tid]));
      }
      double dK = -gK;
      if (fabs(dK) < 1e-10) dK = 0;  // gradK[np.abs(gradK)<1e-10] = 0 (:502)
      dKdt = __dmul_rn(dK, dt_mix);
      sig = sqrt(div_cr(__dmul_rn(__dmul_rn(Kz, fabs(dt_mix)), 2.0), r, ir));
    }
    double u01;
    if (rng_mode == 1) u01 = huni[(size_t)it * p.n + i];
    else u01 = mix_uniform(st, u4, it);
    double R = __dsub_rn(__dmul_rn(2.0, u01), 1.0);
    // z - moving*(dKdz*dt_mix - R*sqrt(Kz*|dt_mix|*2/r)) (:527-528)
    z = __dsub_rn(z, __dmul_rn((double)moving, __dsub_rn(dKdt, __dmul_rn(R, sig))));
    if (z >= 0) z = -z;                                                       // reflect from surface
    if (z < (double)Zmin && moving == 1) z = __dsub_rn((double)__fmul_rn(2.f, Zmin), z);  // reflect from seafloor
    z = __dadd_rn(z, wstep);                                                  // buoyancy
    if (!mix_at_surface && surface) z = 0.0;
    if (z > 0) z = 0.0;                                                       // surface_stick
    if (OIL && z >= 0) z = oil.surface_wave_mixing(z, p, i, it, oa, seed, step);   // OpenOil.surface_wave_mixing (:553-554)
    if (z < (double)Zmin) {   // "let particles stick to bottom": interact_with_seafloor() inside the loop (oceandrift.py:555-559)
      const int act = sfl & 255;
      if (act == 3) sf_flags |= 2;                       // previous: lon/lat go back, z stays
      else if (act) {
        z = (double)Zmin;                                // lift_to_seafloor / deactivate
        if (act == 2) { sf_flags |= 1; moving = 0; wstep = 0.0; }
      }
    }
  }
  if (sf_flags & 1) {   // deactivate_elements(reason='seafloor') (basemodel/__init__.py:1774-1795)
    if (p.status[i] == 0) p.status[i] = sfl >> 8;
    p.moving[i] = 0;
  }
  if (sf_flags & 2) { p.lon[i] = p.plon[i]; p.lat[i] = p.plat[i]; }
  if (vadv >= 0 && (vadv ? z <= 0 : z < 0)) {  // vertical_advection (oceandrift.py:315-350)
    double zz = __dadd_rn(z, __dmul_rn(__dmul_rn((double)moving, (double)p.env[VAR_W][i]), dt));
    z = zz < 0 ? zz : 0.0;
  }
  if (OIL) {   // elements.terminal_velocity = W of the last sub-step; entrained elements carry their new droplet size
    p.tv[i] = (float)oil.W;
    p.aux[OIL_DIAMETER][i] = oil.d;
  }
  p.z[i] = z;
}

#ifndef ODR_VMIX_WAVES
#define ODR_VMIX_WAVES 6   // 80 registers, no scratch: six workgroups (24.6 KB of LDS each) per CU; unconstrained the allocator took 102 (4 waves)
#endif
// one particle of k_vmix_col: column into the thread's LDS slots, sub-steps, stores.
// LAZY: gather only the quads of the column this WAVE needs -- the union of what its particles' three-level windows (the levels
// lv0 - 2 .. lv0 + 2 their terms are formed from, vmix_col_walk) touch.  The texture addresser's time goes with the number of
// gather instructions (profiles/r05_ab_variants.txt section 1: 0.34 instead of 0.41 ms when every lane gathers the same node,
// 0.26 ms for the launch without its sub-steps), and a particle at 30 m never reads K at 300 m.  Returns true -- nothing stored
// -- when a sub-step needed a quad that was not gathered: the caller runs the particle again with the whole column (a second
// copy of this function behind a wave-uniform branch: everything is read again from memory, so nothing of the first pass stays
// in registers for it; a retry loop around fill + walk spilled 548 B per lane).
// what the mixing of one particle reads of it -- requested by the kernel in ONE round trip together with its LDS tables
// (round 5: the tables' loads stood in front of a barrier in front of these, and the ID's load behind a branch in front of the
// first Philox block in front of the column gathers: three dependent round trips before the first gather was issued)
struct VMixState { double slon, slat, z0; int moving, id; float dep, ssh, tv; };
__device__ __forceinline__ VMixState vmix_load_state(const PView &p, long long i) {
  VMixState S;
  S.slon = p.slon[i]; S.slat = p.slat[i]; S.z0 = p.z[i];
  S.moving = p.moving[i]; S.id = p.id[i];
  S.dep = p.env[VAR_DEPTH][i]; S.ssh = p.env[VAR_SSH][i]; S.tv = p.tv[i];
  return S;
}
template <int NQ, bool TL, bool LAZY>
__device__ __forceinline__ bool vmix_col_particle_pass(const DevSource &s, const PView &p, const VMixDesc &D, const VMixArgs &A,
                                                       int vadv, long long i, double *Kp, const double *gsh, int tid, const VMixState &S) {
  const int nzp = D.nzp, sfl = A.sfl;
  const double dt = A.dt;
  const double slon = S.slon, slat = S.slat, z0 = S.z0;
  int moving = S.moving;
  const float dep0 = S.dep, ssh0 = S.ssh, tv0 = S.tv;
  const int id0 = S.id;
  VMixLazy lz;
  lz.loaded = ~0u; lz.missed = false;
  if (LAZY) {
    int zs = 0;
    const double d0 = -z0;
#pragma unroll
    for (int k = 0; k < 4 * NQ - 1; ++k) zs += (k < nzp - 1 && ((k & 1) ? d0 >= s.zmid[k] : d0 > s.zmid[k])) ? 1 : 0;
    int lv0 = zs < 1 ? 1 : (zs > nzp - 2 ? nzp - 2 : zs);
    if (nzp < 3) lv0 = 1;
    const unsigned mine = vmix_quads_of(lv0 - 2, lv0 + 2, nzp);
    unsigned w = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) w |= __ballot((mine >> q) & 1u) ? (1u << q) : 0u;
    lz.loaded = w;
  }
  MixRng R0;
  R0.key = mix_key(A.seed, A.step, id0); R0.q = make_uint4(0u, 0u, 0u, 0u); R0.primed = false;
#ifdef ODR_VMIX_LATE_RNG
  vmix_col_fill<NQ, TL>(s, D, slon, slat, Kp, tid, lz.loaded);
  const MixRng *pre = nullptr;
#else
  // the stream's first block behind the first gathers of the column: its arithmetic runs while they are in flight
  vmix_col_fill<NQ, TL>(s, D, slon, slat, Kp, tid, lz.loaded, [&]() { R0 = mix_rng_begin(A, id0); });
  const MixRng *pre = &R0;
#endif
  int sf_flags = 0;
  const float Zmin = __fmul_rn(-1.f, __fadd_rn(dep0, ssh0));  // float32 (:408)
  double z = vmix_col_walk<NQ>(s, nzp, Kp, gsh, tid, A, i, p.n, id0, z0, moving, Zmin, tv0, sf_flags, pre, LAZY ? &lz : nullptr);
  if (LAZY && lz.missed) return true;
  if (sf_flags & 1) {   // deactivate_elements(reason='seafloor') (basemodel/__init__.py:1774-1795)
    if (p.status[i] == 0) p.status[i] = sfl >> 8;
    p.moving[i] = 0;
  }
  if (sf_flags & 2) { p.lon[i] = p.plon[i]; p.lat[i] = p.plat[i]; }
  if (vadv >= 0 && (vadv ? z <= 0 : z < 0)) {  // vertical_advection (oceandrift.py:315-350)
    // (w is read here, not with the rest of the state: held through the walk it was the one value the allocator put into
    // scratch memory at 80 registers, behind a wait for every load in flight)
    const float w0 = p.env[VAR_W][i];
    double zz = __dadd_rn(z, __dmul_rn(__dmul_rn((double)moving, (double)w0), dt));
    z = zz < 0 ? zz : 0.0;
  }
  p.z[i] = z;
  return false;
}
template <int NQ, bool TL>
__device__ __forceinline__ void vmix_col_particle(const DevSource &s, const PView &p, const VMixDesc &D, const VMixArgs &A,
                                                  int vadv, long long i, double *Kp, const double *gsh, int tid, const VMixState &S) {
#if !defined(ODR_VMIX_LAZY_QUADS)   // the whole column for every particle (the quads on demand measured slower, profiles/r05_ab_variants.txt)
  vmix_col_particle_pass<NQ, TL, false>(s, p, D, A, vadv, i, Kp, gsh, tid, S);
#else
  if constexpr (NQ == 1) vmix_col_particle_pass<NQ, TL, false>(s, p, D, A, vadv, i, Kp, gsh, tid, S);
  else {
    const bool again = vmix_col_particle_pass<NQ, TL, true>(s, p, D, A, vadv, i, Kp, gsh, tid, S);
    if (__ballot(again)) {
      // (an opaque copy of the index: addresses formed from `i` for the first pass would otherwise be kept alive -- in scratch
      // memory, 36 B per lane -- for this one)
      long long i2 = i;
      asm volatile("" : "+v"(i2));
      if (again) vmix_col_particle_pass<NQ, TL, false>(s, p, D, A, vadv, i2, Kp, gsh, tid, vmix_load_state(p, i2));
    }
  }
#endif
}
template <int NQ, bool TL>
__global__ __launch_bounds__(BLOCK, ODR_VMIX_WAVES) void k_vmix_col(const DevWorld *__restrict__ W, PView p, VMixDesc D,
                                                    double dt, double dt_mix_cfg, int mix_at_surface,
                                                    int rng_mode, const double *__restrict__ huni,
                                                    unsigned long long seed, unsigned long long step,
                                                    int vadv, int sfl) {
  constexpr int NL = 4 * NQ;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const DevSource &s = W->src[D.sid];
  const int nzp = D.nzp;
  double *Kp = (double *)smem;              // [NL][BLOCK]
  double *gsh = Kp + (size_t)NL * BLOCK;    // [4][NL]
  // the particle's state is requested first: its loads and the tables' are one round trip (threads past the end read element 0)
  if (D.guard && *D.guard == 0ull) return;   // (wave-uniform: the launch was enqueued before the host knew whether it may run)
  const long long i = pid();
  const bool valid = i < p.n;
  const VMixState S = vmix_load_state(p, valid ? i : 0);
  if (tid < nzp) {
    gsh[tid] = s.vg_a[tid]; gsh[NL + tid] = s.vg_b[tid]; gsh[2 * NL + tid] = s.vg_c[tid];
    gsh[3 * NL + tid] = s.zmid[tid];     // level boundaries (entries >= nzp - 1 are not read)
  }
  __syncthreads();
  if (!valid) return;  // no barrier below: every thread touches only its own LDS column
  VMixArgs A;
  A.dt = dt; A.dt_mix_cfg = dt_mix_cfg; A.mix_at_surface = mix_at_surface; A.rng_mode = rng_mode; A.sfl = sfl; A.pad = 0;
  A.huni = huni; A.seed = seed; A.step = step;
  vmix_col_particle<NQ, TL>(s, p, D, A, vadv, i, Kp, gsh, tid, S);
}

// ---- k_vmix_win: the same mixing with the diffusivity of FIVE levels per particle instead of the whole column.
// A particle rarely leaves the three levels around its starting one (lv0 - 1 .. lv0 + 1) within a step, and their terms
// need K on lv0 - 2 .. lv0 + 2: per corner and time level one 16-byte gather at level lv0 - 1 (dword-aligned) and one
// 4-byte gather of level lv0 - 2 replace the NQ quads of the column -- for the 12-level field 16 gathers, 8 of them
// narrow, instead of 24 wide ones, five float64 layer values instead of twelve -- and no column in LDS: the cost no
// longer grows with the number of levels of the reader.  A particle whose sub-step does leave the window (0.07 % of them
// per step in the C3 field, i.e. some lane of ~4 % of the waves) finishes its sub-steps in a second loop that fetches the
// three levels around its current one in every sub-step (footprint offsets and weights parked in LDS by the first part).
// Level terms are a pure function of the column, so where they came from does not show in the result: same floats, same
// operations in the same order as vmix_col_fill / vmix_col_walk -- bit-identical (tests/test_gpu_vmix_window.py).
// (Earlier attempts of round 4: refill INSIDE the sub-step loop, 258 registers; leavers appended to a list and a second
// launch of k_vmix_col over it, 33 us for 7 000 particles -- the latency of one pass, more than the window saved; an outer
// loop that re-centres the window: the reader's scalars hoisted out of it overflow the scalar registers, 450 B of scratch.)
#ifndef ODR_VWIN_WAVES
#define ODR_VWIN_WAVES 5
#endif
// one layer value of the column as the ReaderBlock forms it: bilinear in float32 out, time interpolation in float64,
// fallback where the position is not covered or the value is not finite
template <bool TL>
__device__ __forceinline__ double vmix_layer(float b00, float b01, float b10, float b11, float a00, float a01, float a10, float a11,
                                             double w00, double w01, double w10, double w11, double wgt, bool cov, double Kfb) {
  double v = (double)bilw(b00, b01, b10, b11, w00, w01, w10, w11);
  if (TL) {
    const double w = (double)bilw(a00, a01, a10, a11, w00, w01, w10, w11);
    v = __dadd_rn(__dmul_rn(v, 1 - wgt), __dmul_rn(w, wgt));
  }
  return (cov && isfinite(v)) ? v : Kfb;
}
struct VMixGrad { double gd0, gi0, gd1, gi1, gd2, gi2, dt_mix; bool uniform_z; };
// -dK/dz * dt_mix and sqrt(K |dt_mix| 2 / r) of level zl from K below (Kl), on (Kz) and above (Kh) it (vmix_col_walk's level_terms)
__device__ __forceinline__ void vmix_terms(const VMixGrad &G, const double *gsh, int nzp, int zl, double Kl, double Kz, double Kh,
                                           double &dk_dt, double &sg) {
  const double r = 1.0 / 3, ir = 1.0 / r;
  double gK;
  if (zl == 0) gK = div_cr(Kh - Kz, G.gd0, G.gi0);
  else if (zl == nzp - 1) gK = div_cr(Kz - Kl, G.gd1, G.gi1);
  else if (G.uniform_z) gK = div_cr(Kh - Kl, G.gd2, G.gi2);
  else gK = __dadd_rn(__dadd_rn(__dmul_rn(gsh[zl], Kl), __dmul_rn(gsh[nzp + zl], Kz)), __dmul_rn(gsh[2 * nzp + zl], Kh));
  double dK = -gK;
  if (fabs(dK) < 1e-10) dK = 0;
  dk_dt = __dmul_rn(dK, G.dt_mix);
  sg = sqrt(div_cr(__dmul_rn(__dmul_rn(Kz, fabs(G.dt_mix)), 2.0), r, ir));
}
template <bool TL>
__global__ __launch_bounds__(BLOCK, ODR_VWIN_WAVES) void k_vmix_win(const DevWorld *__restrict__ W, PView p, VMixDesc D,
                                                    double dt, double dt_mix_cfg, int mix_at_surface,
                                                    int rng_mode, const double *__restrict__ huni,
                                                    unsigned long long seed, unsigned long long step,
                                                    int vadv, int sfl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (D.guard && *D.guard == 0ull) return;   // (see k_vmix_col)
  const int tid = threadIdx.x;
  const DevSource &s = W->src[D.sid];
  const int nzp = D.nzp;
  double *fw = (double *)smem;                        // [4][BLOCK]: the footprint's weights, for the second loop
  unsigned *fo = (unsigned *)(fw + 4 * BLOCK);        // [4][BLOCK]: byte offsets of its node records
  double *gsh = (double *)(fo + 4 * BLOCK);           // [4][nzp]: np.gradient coefficients a, b, c and the level boundaries
  if (tid < nzp) {
    gsh[tid] = s.vg_a[tid]; gsh[nzp + tid] = s.vg_b[tid]; gsh[2 * nzp + tid] = s.vg_c[tid];
    gsh[3 * nzp + tid] = tid < nzp - 1 ? s.zmid[tid] : __builtin_inf();
  }
  __syncthreads();
  const long long i = pid();
  if (i >= p.n) return;   // no barrier below
  const double slon = p.slon[i], slat = p.slat[i], z0 = p.z[i];
  int moving = p.moving[i];
  const float dep0 = p.env[VAR_DEPTH][i], ssh0 = p.env[VAR_SSH][i], tv0 = p.tv[i];
  const int id0 = rng_mode == 0 ? p.id[i] : 0;
  const float w0 = vadv >= 0 ? p.env[VAR_W][i] : 0.f;
  const MixKey key = mix_key(seed, step, id0);
  uint4 u4 = make_uint4(0u, 0u, 0u, 0u);
  const bool primed = rng_mode == 0;
  VMixGrad G;
  G.uniform_z = s.vg_uniform != 0;
  G.gd0 = s.vg_d[0]; G.gi0 = s.vg_id[0]; G.gd1 = s.vg_d[1]; G.gi1 = s.vg_id[1]; G.gd2 = s.vg_d[2]; G.gi2 = s.vg_id[2];
  const double sgn = dt > 0 ? 1.0 : (dt < 0 ? -1.0 : 0.0);
  const double dt_mix = dt_mix_cfg * sgn;
  G.dt_mix = dt_mix;
  const int ntimes = abs((int)(dt / dt_mix));
  double wstep = __dmul_rn(__dmul_rn((double)tv0, dt_mix), (double)moving);
  const float Zmin = __fmul_rn(-1.f, __fadd_rn(dep0, ssh0));  // float32 (:408)
  const double Kfb = (double)D.Kfb, wgt = D.wgt;
  const float *kb = D.kb, *ka = TL ? D.ka : D.kb;
  // level of a depth as vmix_col_walk counts it: the number of boundaries below d (odd ones count when d >= zm, even ones
  // when d > zm; NaN counts none)
  auto level_of = [&](double d) {
    int zs = 0;
    for (int k = 0; k < nzp - 1; ++k) {
      const double b = gsh[3 * nzp + k];
      zs += ((k & 1) ? d >= b : d > b) ? 1 : 0;
    }
    return zs;
  };
  auto centre_of = [&](int zs) { return zs < 1 ? 1 : (zs > nzp - 2 ? nzp - 2 : zs); };   // window centre: 1 .. nzp - 2
  // the window around level lv0: K on lv0 - 2 .. lv0 + 2 from the footprint (o.., w..) -> the terms of lv0 - 1 .. lv0 + 1
  // and the boundaries lv0 - 2 .. lv0 + 1, the ">=" of the odd ones folded into the value (vmix_col_walk).
  // first: also draws the stream's first Philox block while the gathers are in flight
  double c_dk[3], c_sg[3], wb[4];
  auto load_window = [&](int lv0, unsigned o00, unsigned o01, unsigned o10, unsigned o11, double w00, double w01, double w10,
                         double w11, bool cov, bool first) {
    // level offsets inside the node record; the wide gather reaches level lv0 + 2 <= nzp: at most one float past the K
    // array, inside the record or the 64 spare bytes a block ends with
    const unsigned lq = (unsigned)(lv0 - 1) * 4u, l1 = (unsigned)(lv0 >= 2 ? lv0 - 2 : 0) * 4u;
    const F4 b00 = ld_off<F4>(kb, o00 + lq), b01 = ld_off<F4>(kb, o01 + lq), b10 = ld_off<F4>(kb, o10 + lq), b11 = ld_off<F4>(kb, o11 + lq);
    const float c00 = ld_off<float>(kb, o00 + l1), c01 = ld_off<float>(kb, o01 + l1), c10 = ld_off<float>(kb, o10 + l1), c11 = ld_off<float>(kb, o11 + l1);
    // (on a time level the second set repeats the first: the compiler drops it)
    const F4 a00 = ld_off<F4>(ka, o00 + lq), a01 = ld_off<F4>(ka, o01 + lq), a10 = ld_off<F4>(ka, o10 + lq), a11 = ld_off<F4>(ka, o11 + lq);
    const float e00 = ld_off<float>(ka, o00 + l1), e01 = ld_off<float>(ka, o01 + l1), e10 = ld_off<float>(ka, o10 + l1), e11 = ld_off<float>(ka, o11 + l1);
    if (first && primed) u4 = mix_block(key, 0u);
    double Kw[5];   // K on levels lv0 - 2 .. lv0 + 2 (entries of levels outside 0 .. nzp - 1 are never used)
    Kw[0] = vmix_layer<TL>(c00, c01, c10, c11, e00, e01, e10, e11, w00, w01, w10, w11, wgt, cov, Kfb);
    Kw[1] = vmix_layer<TL>(b00.x, b01.x, b10.x, b11.x, a00.x, a01.x, a10.x, a11.x, w00, w01, w10, w11, wgt, cov, Kfb);
    Kw[2] = vmix_layer<TL>(b00.y, b01.y, b10.y, b11.y, a00.y, a01.y, a10.y, a11.y, w00, w01, w10, w11, wgt, cov, Kfb);
    Kw[3] = vmix_layer<TL>(b00.z, b01.z, b10.z, b11.z, a00.z, a01.z, a10.z, a11.z, w00, w01, w10, w11, wgt, cov, Kfb);
    Kw[4] = vmix_layer<TL>(b00.w, b01.w, b10.w, b11.w, a00.w, a01.w, a10.w, a11.w, w00, w01, w10, w11, wgt, cov, Kfb);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int zl = lv0 - 1 + q;
      c_dk[q] = 0; c_sg[q] = 0;
      if (zl >= 0 && zl < nzp) vmix_terms(G, gsh, nzp, zl, Kw[q], Kw[q + 1], Kw[q + 2], c_dk[q], c_sg[q]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = lv0 - 2 + j;
      double b = k < 0 ? -1.0 : __builtin_inf();
      if (k >= 0 && k < nzp - 1) {
        b = gsh[3 * nzp + k];
        if ((k & 1) && b > 0) b = __longlong_as_double(__double_as_longlong(b) - 1);
      }
      wb[j] = b;
    }
  };
  double z = z0;
  int sf_flags = 0, it = 0;
  // one random-walk sub-step on the window (vmix_col_walk's loop body; oceandrift.py:531-559); false: the particle is on a
  // level outside the window and nothing was done.  x: the sub-step's 24-bit draw (ODR_RNG_DEVICE)
  auto substep = [&](int lv0, unsigned x) -> bool {
    const bool surface = z == 0;
    const double d = -z;
    int q = (d > wb[1] ? 1 : 0) + (d > wb[2] ? 1 : 0);
    if (!(d > wb[0]) || d > wb[3]) {
      // below boundary lv0 - 2 or above lv0 + 1: a level outside the window.  (NaN counts no boundary: level 0, inside
      // the window exactly when lv0 == 1 -- where level_of / centre_of put the window of a NaN)
      if (d != d && lv0 == 1) q = 0;
      else return false;
    }
    const double dKdt = q == 0 ? c_dk[0] : (q == 1 ? c_dk[1] : c_dk[2]);
    const double sig = q == 0 ? c_sg[0] : (q == 1 ? c_sg[1] : c_sg[2]);
    double R;
    if (rng_mode == 1) R = __dsub_rn(__dmul_rn(2.0, huni[(size_t)it * p.n + i]), 1.0);
    else R = mix_R_of(x);
    z = __dsub_rn(z, __dmul_rn((double)moving, __dsub_rn(dKdt, __dmul_rn(R, sig))));
    if (z >= 0) z = -z;
    if (z < (double)Zmin && moving == 1) z = __dsub_rn((double)__fmul_rn(2.f, Zmin), z);
    z = __dadd_rn(z, wstep);
    if (!mix_at_surface && surface) z = 0.0;
    if (z > 0) z = 0.0;
    if (z < (double)Zmin) {   // "let particles stick to bottom": interact_with_seafloor() inside the loop (oceandrift.py:555-559)
      const int act = sfl & 255;
      if (act == 3) sf_flags |= 2;                       // previous: lon/lat go back, z stays
      else if (act) {
        z = (double)Zmin;                                // lift_to_seafloor / deactivate
        if (act == 2) { sf_flags |= 1; moving = 0; wstep = 0.0; }
      }
    }
    return true;
  };
  bool cov;
  {
    // ---- vmix_col_fill's footprint
    double lon = slon, x, y;
    if (s.lon_mode == 1) lon = np_mod(lon + 180.0, 360.0) - 180.0;
    else if (s.lon_mode == 2) lon = np_mod(lon, 360.0);
    proj_fwd_rt(s.proj, lon, slat, x, y);
    cov = x >= s.xmin && x <= s.xmax && y >= s.ymin && y <= s.ymax;
    if (s.mod360_x) x = np_mod(x, 360.0);
    const DevBlock &bb = s.slot[D.geo_slot];
    const double xi = __dmul_rn(div_cr(x - bb.x0, bb.xspan, bb.ixspan), (double)(bb.nx - 1));
    const double yi = __dmul_rn(div_cr(y - bb.y0, bb.yspan, bb.iyspan), (double)(bb.ny - 1));
    const int ny = bb.ny, nx = bb.nx;
    const Axis ay = axis_fp(yi, ny), ax = axis_fp(xi, nx);
    const double ty = ay.t, tx = ax.t, wy0 = 1 - ty, wx0 = 1 - tx;
    // uncovered particles gather node (0,0) and discard it: keeps the loads unconditional
    const unsigned recb = (unsigned)bb.rec * 4u;
    const unsigned r0 = __umul24((unsigned)ay.i0, (unsigned)nx), r1 = __umul24((unsigned)ay.i1, (unsigned)nx);
    const unsigned o00 = cov ? __umul24(r0 + (unsigned)ax.i0, recb) : 0u, o01 = cov ? __umul24(r0 + (unsigned)ax.i1, recb) : 0u;
    const unsigned o10 = cov ? __umul24(r1 + (unsigned)ax.i0, recb) : 0u, o11 = cov ? __umul24(r1 + (unsigned)ax.i1, recb) : 0u;
    const double w00 = wy0 * wx0, w01 = wy0 * tx, w10 = ty * wx0, w11 = ty * tx;
    fw[tid] = w00; fw[BLOCK + tid] = w01; fw[2 * BLOCK + tid] = w10; fw[3 * BLOCK + tid] = w11;
    fo[tid] = o00; fo[BLOCK + tid] = o01; fo[2 * BLOCK + tid] = o10; fo[3 * BLOCK + tid] = o11;
    const int lv0 = centre_of(level_of(-z0));
    load_window(lv0, o00, o01, o10, o11, w00, w01, w10, w11, cov, true);
    // first pass: every lane is at the same sub-step, so a Philox block is consumed in an unrolled group of five -- which of
    // its words a sub-step takes is known at compile time
    bool left = false;
    for (unsigned b5 = 0; !left && it < ntimes; ++b5) {
      if (rng_mode == 0 && !(primed && b5 == 0)) u4 = mix_block(key, b5);
#pragma unroll
      for (unsigned k = 0; k < 5; ++k) {
        if (it >= ntimes) break;
        if (!substep(lv0, mix_word(u4, k))) { left = true; break; }
        ++it;
      }
    }
  }
  while (it < ntimes) {
    // ---- a particle that left its window: another window around its current level, and on from the same sub-step (the
    // lanes are at different sub-steps now; the block of the current one is in u4 unless the sub-step opens a new block)
    const int lv0 = centre_of(level_of(-z));
    load_window(lv0, fo[tid], fo[BLOCK + tid], fo[2 * BLOCK + tid], fo[3 * BLOCK + tid], fw[tid], fw[BLOCK + tid], fw[2 * BLOCK + tid],
                fw[3 * BLOCK + tid], cov, false);
    for (; it < ntimes; ++it) {
      const unsigned b = (unsigned)it / 5u, k = (unsigned)it - 5u * b;
      if (rng_mode == 0 && k == 0 && !(primed && it == 0)) u4 = mix_block(key, b);
      if (!substep(lv0, mix_word(u4, k))) break;
    }
  }
  if (sf_flags & 1) {   // deactivate_elements(reason='seafloor') (basemodel/__init__.py:1774-1795)
    if (p.status[i] == 0) p.status[i] = sfl >> 8;
    p.moving[i] = 0;
  }
  if (sf_flags & 2) { p.lon[i] = p.plon[i]; p.lat[i] = p.plat[i]; }
  if (vadv >= 0 && (vadv ? z <= 0 : z < 0)) {  // vertical_advection (oceandrift.py:315-350)
    double zz = __dadd_rn(z, __dmul_rn(__dmul_rn((double)moving, (double)w0), dt));
    z = zz < 0 ? zz : 0.0;
  }
  p.z[i] = z;
}

// ---- vertical mixing with a wind-parameterised diffusivity profile (oceandrift.py:385-395,425-458) ----
// K at the 1 m level l (depth l metres) for one element: verticaldiffusivity_Large1994 / _Sundby1983
// (physics_methods.py:203-250) with NumPy's dtypes: wind speed and mixed-layer depth are float32 environment
// arrays, Python float constants multiply them in float32, depth / MLD and everything after are float64.
enum { DIFF_LARGE1994 = 1, DIFF_SUNDBY1983 = 2 };
template <int MODEL>
__device__ __forceinline__ double k_windprofile(int l, float w, float mld, double bg) {
  const double d = (double)l;
  double K;
  if (MODEL == DIFF_LARGE1994) {
    const float ws = __fmul_rn(__fmul_rn(__fmul_rn(w, w), 1.25e-3f), 1.22f);  // windspeed*windspeed * cd * rhoa
    const double sig = __ddiv_rn(d, (double)mld);
    const double s2 = __dmul_rn(sig, sig);                                    // sigma**2 (NumPy: square)
    // sigma**3 is libm pow(sigma, 3.0) in NumPy (< 1 ulp): the cube rounded once from the exact square
    const double s2lo = fma(sig, sig, -s2);
    const double ph = __dmul_rn(s2, sig);
    const double s3 = ph + (fma(s2, sig, -ph) + s2lo * sig);
    double G = __dadd_rn(__dadd_rn(sig, __dmul_rn(-2.0, s2)), s3);            // 1.*s + -2*s**2 + 1*s**3
    if (G >= 1) G = 0.0;
    const float m1 = __fmul_rn(__fmul_rn(mld, 0.2f), 0.4f);                   // MLD * stabilityfunction * 0.4
    K = __dadd_rn(__dmul_rn(__dmul_rn((double)m1, G), (double)ws), __dmul_rn(sig, bg));
  } else {
    const float t = __fmul_rn(__fmul_rn(2.26e-4f, w), w);
    K = __dadd_rn(76.1e-4, (double)t);
    if (d > (double)__fsub_rn(mld, 1.0f)) K = __dmul_rn(__dadd_rn(K, bg), 0.5);  // transition
  }
  if (d >= (double)mld) K = bg;                                               // cutoff below the mixed layer
  return K;
}

// Levels: mixing_z = -arange(0, MLD.max() + 2) (1 m spacing, oceandrift.py:430; MLD.max() from the reduction slot),
// level index = round-half-even of the clamped depth (interp1d over arange is the identity), np.gradient with the
// uniform spacing -1.  No profile gather, no LDS: K is a closed form of (level, wind speed, MLD) per element.
template <int MODEL, bool OIL = false>
__global__ __launch_bounds__(BLOCK) void k_vmix_wind(PView p, const double *__restrict__ red, double bg, double dt,
                                                     double dt_mix_cfg, int mix_at_surface, int rng_mode,
                                                     const double *__restrict__ huni, unsigned long long seed,
                                                     unsigned long long step, int vadv, int sfl, OilArgs oa = OilArgs()) {
  const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  const int nlev = (int)ceil((double)__fadd_rn((float)red[R_MLDMAX], 2.0f));
  const float w = speed_f32(p.env[VAR_XWIND][i], p.env[VAR_YWIND][i]);        // wind_speed(), physics_methods.py:885
  const float mld = p.env[VAR_MLD][i];
  const double sgn = dt > 0 ? 1.0 : (dt < 0 ? -1.0 : 0.0);
  const double dt_mix = dt_mix_cfg * sgn;
  const int ntimes = abs((int)(dt / dt_mix));
  const double r = 1.0 / 3, ir = 1.0 / r;
  double z = p.z[i];
  int moving = p.moving[i];
  int sf_flags = 0;   // 1: deactivated on the sea floor, 2: moved back horizontally (general:seafloor_action)
  const float Zmin = __fmul_rn(-1.f, __fadd_rn(p.env[VAR_DEPTH][i], p.env[VAR_SSH][i]));
  double wstep = __dmul_rn(__dmul_rn((double)p.tv[i], dt_mix), (double)moving);
  const MixKey st = mix_key(seed, step, rng_mode == 0 ? p.id[i] : 0);
  uint4 u4 = make_uint4(0u, 0u, 0u, 0u);
  OilLane oil;
  if (OIL) oil.init(p, i, oa);
  int zc = -1;
  double dKdt = 0, sig = 0;
  for (int it = 0; it < ntimes; ++it) {
    const bool surface = z == 0;
    if (OIL)   // update_terminal_velocity at the top of every sub-step (oceandrift.py:509), w*dt_mix*moving (:548)
      wstep = __dmul_rn(__dmul_rn(oil.terminal_velocity(), dt_mix), (double)moving);
    double idx = -z;
    idx = idx < 0 ? 0.0 : (idx > (double)(nlev - 1) ? (double)(nlev - 1) : idx);
    const int zi = (int)rint(idx);
    if (zi != zc) {   // the level changes every few sub-steps at most
      zc = zi;
      const double Kz = k_windprofile<MODEL>(zi, w, mld, bg);
      double g;
      if (zi == 0) g = __ddiv_rn(__dsub_rn(k_windprofile<MODEL>(1, w, mld, bg), Kz), -1.0);
      else if (zi == nlev - 1) g = __ddiv_rn(__dsub_rn(Kz, k_windprofile<MODEL>(zi - 1, w, mld, bg)), -1.0);
      else g = __ddiv_rn(__dsub_rn(k_windprofile<MODEL>(zi + 1, w, mld, bg), k_windprofile<MODEL>(zi - 1, w, mld, bg)), -2.0);
      double dK = -g;
      if (fabs(dK) < 1e-10) dK = 0;
      dKdt = __dmul_rn(dK, dt_mix);
      sig = sqrt(div_cr(__dmul_rn(__dmul_rn(Kz, fabs(dt_mix)), 2.0), r, ir));
    }
    double u01;
    if (rng_mode == 1) u01 = huni[(size_t)it * p.n + i];
    else u01 = mix_uniform(st, u4, it);
    const double R = __dsub_rn(__dmul_rn(2.0, u01), 1.0);
    z = __dsub_rn(z, __dmul_rn((double)moving, __dsub_rn(dKdt, __dmul_rn(R, sig))));
    if (z >= 0) z = -z;
    if (z < (double)Zmin && moving == 1) z = __dsub_rn((double)__fmul_rn(2.f, Zmin), z);
    z = __dadd_rn(z, wstep);
    if (!mix_at_surface && surface) z = 0.0;
    if (z > 0) z = 0.0;
    if (OIL && z >= 0) z = oil.surface_wave_mixing(z, p, i, it, oa, seed, step);   // OpenOil.surface_wave_mixing (:553-554)
    if (z < (double)Zmin) {   // "let particles stick to bottom": interact_with_seafloor() inside the loop (oceandrift.py:555-559)
      const int act = sfl & 255;
      if (act == 3) sf_flags |= 2;                       // previous: lon/lat go back, z stays
      else if (act) {
        z = (double)Zmin;                                // lift_to_seafloor / deactivate
        if (act == 2) { sf_flags |= 1; moving = 0; wstep = 0.0; }
      }
    }
  }
  if (sf_flags & 1) {   // deactivate_elements(reason='seafloor') (basemodel/__init__.py:1774-1795)
    if (p.status[i] == 0) p.status[i] = sfl >> 8;
    p.moving[i] = 0;
  }
  if (sf_flags & 2) { p.lon[i] = p.plon[i]; p.lat[i] = p.plat[i]; }
  if (vadv >= 0 && (vadv ? z <= 0 : z < 0)) {  // vertical_advection (oceandrift.py:315-350)
    double zz = __dadd_rn(z, __dmul_rn(__dmul_rn((double)moving, (double)p.env[VAR_W][i]), dt));
    z = zz < 0 ? zz : 0.0;
  }
  if (OIL) {   // elements.terminal_velocity = W of the last sub-step; entrained elements carry their new droplet size
    p.tv[i] = (float)oil.W;
    p.aux[OIL_DIAMETER][i] = oil.d;
  }
  p.z[i] = z;
}

#endif  // ODR_TU_MIX
#ifdef ODR_TU_MISC
// vertical_advection (oceandrift.py:315-350)
__global__ __launch_bounds__(BLOCK) void k_vadvect(PView p, double dt, int at_surface) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  double z = p.z[i];
  if (at_surface ? z <= 0 : z < 0) {
    double zz = __dadd_rn(z, __dmul_rn(__dmul_rn((double)p.moving[i], (double)p.env[VAR_W][i]), dt));
    p.z[i] = zz < 0 ? zz : 0.0;
  }
}

// vertical_buoyancy (oceandrift.py:352-368); elements below the sea floor "interact_with_seafloor" again, here
// inside update(): sfl = action | status_code << 8 (0 none, 1 lift_to_seafloor, 2 deactivate, 3 previous)
__global__ __launch_bounds__(BLOCK) void k_vbuoy(PView p, double dt, int sfl) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  double z = p.z[i];
  if (z < 0) {
    double zz = __dadd_rn(z, (double)__fmul_rn(p.tv[i], (float)dt));
    z = zz < 0 ? zz : 0.0;
  }
  float Zmin = __fmul_rn(-1.f, __fadd_rn(p.env[VAR_DEPTH][i], p.env[VAR_SSH][i]));
  if (z < (double)Zmin) {
    const int act = sfl & 255;
    if (act == 3) { p.lon[i] = p.plon[i]; p.lat[i] = p.plat[i]; }
    else if (act) {
      z = (double)Zmin;
      if (act == 2) {
        if (p.status[i] == 0) p.status[i] = sfl >> 8;
        p.moving[i] = 0;
      }
    }
  }
  p.z[i] = z;
}

// ------------------------------------------------------------ coastline / seafloor
// interact_with_coastline (basemodel/__init__.py:670-746), precision None
__global__ __launch_bounds__(BLOCK) void k_coast(PView p, int action, int code, int seeded_code,
                                                 unsigned long long *n_hit) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  bool hit = false;
  if (i < p.n && p.env[VAR_LAND][i] == 1.0f) {
    hit = true;
    if (action == 1) {
      if (p.z[i] <= 0) {  // deactivate_elements(reason='stranded') (:1774-1795)
        if (p.status[i] == 0) p.status[i] = code;
        p.moving[i] = 0;
      }
    } else if (action == 2) {
      if (seeded_code > 0 && p.age[i] == 0.0f) {  // reason='seeded_on_land' (:715-719)
        if (p.status[i] == 0) p.status[i] = seeded_code;
        p.moving[i] = 0;
      }
      p.lon[i] = p.plon[i];
      p.lat[i] = p.plat[i];
      p.env[VAR_LAND][i] = 0.0f;   // self.environment.land_binary_mask[on_land] = 0 (:746)
    }
  }
  unsigned long long b = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_hit, (unsigned long long)__popcll(b));
}

// interact_with_coastline with general:coastline_approximation_precision (basemodel/__init__.py:694-746) and
// coastline_crossing (:81-134): every element on land is moved to the first land sample ('stranding', land_side) or
// the last water sample before it ('previous') of the reference's search pattern between its previous and its
// current position -- np.meshgrid(np.linspace(lon1, lon2, xs), np.linspace(lat1, lat2, ys)) raveled, i.e. the
// RECTANGLE spanned by the two positions scanned row by row (x fastest), xs = floor(|dlon| / step) samples (1 when
// |dlon| <= step: only lon1), likewise ys; the first sample on land wins.  In the reference this is a Python loop over
// the stranded elements with one landmask call each; here one thread per element walks its own rectangle against the
// bit-packed raster (`mask`, the landmask source -- the reference uses its global landmask here, whatever reader
// provided land_binary_mask).
__global__ __launch_bounds__(BLOCK) void k_coast_crossing(const DevWorld *__restrict__ W, int mask_sid, PView p, int action,
                                                          int code, int seeded_code, double step_deg,
                                                          unsigned long long *n_hit) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  bool hit = false;
  if (i < p.n && p.env[VAR_LAND][i] == 1.0f) {
    hit = true;
    const DevSource &mask = W->src[mask_sid];
    if (action == 1) {
      if (p.z[i] <= 0) {  // deactivate_elements(reason='stranded') (:1774-1795)
        if (p.status[i] == 0) p.status[i] = code;
        p.moving[i] = 0;
      }
    } else if (seeded_code > 0 && p.age[i] == 0.0f) {  // reason='seeded_on_land' (:715-719)
      if (p.status[i] == 0) p.status[i] = seeded_code;
      p.moving[i] = 0;
    }
    const bool land_side = action == 1;
    const double lon1 = p.plon[i], lat1 = p.plat[i];
    double lon2 = p.lon[i];
    const double lat2 = p.lat[i];
    double lon_c = land_side ? lon2 : lon1, lat_c = land_side ? lat2 : lat1;
    double xd = fabs(__dsub_rn(lon2, lon1));
    const double yd = fabs(__dsub_rn(lat2, lat1));
    if (!(xd == 0 && yd == 0)) {
      if (xd > 180 && lon1 < 0) {   // crossing the dateline
        lon2 = __dsub_rn(lon2, 360.0);
        xd = fabs(__dsub_rn(lon2, lon1));
      }
      const long long xs = xd > step_deg ? (long long)floor(__ddiv_rn(xd, step_deg)) : 1;
      const long long ys = yd > step_deg ? (long long)floor(__ddiv_rn(yd, step_deg)) : 1;
      // np.linspace(a, b, n): k * ((b - a) / (n - 1)) + a, the last sample is b itself, n == 1 gives [a]
      const double sx = xs > 1 ? __ddiv_rn(__dsub_rn(lon2, lon1), (double)(xs - 1)) : 0.0;
      const double sy = ys > 1 ? __ddiv_rn(__dsub_rn(lat2, lat1), (double)(ys - 1)) : 0.0;
      double px = 0, py = 0;   // the sample before the current one in raveled order
      bool found = false, first = true;
      // The reference evaluates all xs * ys samples at once (and runs out of memory when an element jumps across the
      // dateline eastwards: |dlon| ~ 360 is only repaired for lon1 < 0); one thread must not spin on such a rectangle
      // for seconds: after 4 M samples without land the search ends like a search that found none.
      long long budget = 4000000;
      for (long long iy = 0; iy < ys && !found; ++iy) {
        const double yy = (ys > 1 && iy == ys - 1) ? lat2 : __dadd_rn(__dmul_rn((double)iy, sy), lat1);
        for (long long ix = 0; ix < xs; ++ix) {
          if (--budget < 0) { found = true; break; }
          const double xx = (xs > 1 && ix == xs - 1) ? lon2 : __dadd_rn(__dmul_rn((double)ix, sx), lon1);
          if (landmask_contains(mask, xx, yy)) {
            // land_side False: index = max(0, index - 1) -- the very first sample stays itself
            if (land_side || first) { lon_c = xx; lat_c = yy; }
            else { lon_c = px; lat_c = py; }
            found = true;
            break;
          }
          px = xx; py = yy; first = false;
        }
      }
    }
    p.lon[i] = lon_c;
    p.lat[i] = lat_c;
    p.env[VAR_LAND][i] = 0.0f;   // self.environment.land_binary_mask[on_land] = 0 (:729, :746)
  }
  unsigned long long b = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_hit, (unsigned long long)__popcll(b));
}

// increase_age_and_retire (basemodel/__init__.py:2342-2352): age_seconds += dt (float32);
// elements older than drift:max_age_seconds are deactivated with reason 'retired'
__global__ __launch_bounds__(BLOCK) void k_age(PView p, float dt, float max_age, int retired_code) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  float a = __fadd_rn(p.age[i], dt);
  p.age[i] = a;
  if (max_age > 0 && a >= max_age) {
    if (p.status[i] == 0) p.status[i] = retired_code;
    p.moving[i] = 0;
  }
}

// interact_with_seafloor (basemodel/__init__.py:748-783): 1 lift_to_seafloor, 2 deactivate (reason 'seafloor', z set
// to the sea floor), 3 previous (back to the stored lon/lat; z stays)
__global__ __launch_bounds__(BLOCK) void k_seafloor(PView p, int action, int code, unsigned long long *n_hit) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  bool hit = false;
  if (i < p.n) {
    float floorz = -__fadd_rn(p.env[VAR_DEPTH][i], p.env[VAR_SSH] ? p.env[VAR_SSH][i] : 0.f);
    if (p.z[i] < (double)floorz) {
      hit = true;
      if (action == 3) {
        p.lon[i] = p.plon[i];
        p.lat[i] = p.plat[i];
      } else {
        if (action == 2) {
          if (p.status[i] == 0) p.status[i] = code;
          p.moving[i] = 0;
        }
        p.z[i] = (double)floorz;
      }
    }
  }
  unsigned long long b = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_hit, (unsigned long long)__popcll(b));
}

struct VarList { int n; int var[NVAR]; };
// report_missing_variables (basemodel/__init__.py:2501-2515)
__global__ __launch_bounds__(BLOCK) void k_deactivate_missing(PView p, VarList L, int code, unsigned long long *n_hit) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  bool miss = false;
  if (i < p.n) {
    for (int k = 0; k < L.n; ++k) { const float e = p.env[L.var[k]][i]; miss |= e != e; }
    if (miss) {
      if (p.status[i] == 0) p.status[i] = code;
      p.moving[i] = 0;
    }
  }
  unsigned long long b = __ballot(miss);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_hit, (unsigned long long)__popcll(b));
}

__global__ __launch_bounds__(BLOCK) void k_status_remap(PView p, int from, int to) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i < p.n && p.status[i] == from) p.status[i] = to;
}

__global__ __launch_bounds__(BLOCK) void k_count_status(PView p, int code, unsigned long long *n_hit) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  unsigned long long b = __ballot(i < p.n && p.status[i] == code);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_hit, (unsigned long long)__popcll(b));
}

__global__ __launch_bounds__(BLOCK) void k_deactivate(PView p, const unsigned char *mask, int code) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n || !mask[i]) return;
  if (p.status[i] == 0) p.status[i] = code;
  p.moving[i] = 0;
}

// ------------------------------------------------------- stable stream compaction
// remove_deactivated_elements = LagrangianArray.move_elements (elements.py:197-228):
// order-preserving partition.  Pass 1 counts actives per 256-block, pass 2 scans the
// block counts (single workgroup), pass 3 scatters every property.
// flags (may be NULL): bit k is set when an element carries the provisional status number 100 + k (a deactivation reason
// that has no status category yet, opendrift_amd/oceandrift.py:_status_code) -- read back with the count in one transfer
// ---- rank of every present element in ascending ID (what the reference's arrays are ordered by): a bitmap of the IDs,
// popcounts per word, their exclusive scan, then rank = words before + bits before
// (elements already deactivated in this step are not counted: the reference has removed them from its arrays by the time
// the next call is made, remove_deactivated_elements comes before update())
// sid >= 0: only the elements the ensemble reader `sid` covers at their position are numbered (the elements the reference
// hands to its ReaderBlock, variables.py:747-765); sid < 0: every active element
__global__ __launch_bounds__(BLOCK) void k_rank_mark(const DevWorld *__restrict__ W, int sid, PView p, unsigned *words) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n || p.status[i] != 0) return;
  if (sid >= 0) {
    double x, y;
    if (!source_covers_xyz(W->src[sid], p.lon[i], p.lat[i], p.z[i], x, y)) return;
  }
  const unsigned v = (unsigned)p.id[i];
  atomicOr(&words[v >> 5], 1u << (v & 31u));
}
__global__ __launch_bounds__(BLOCK) void k_rank_count(const unsigned *words, long long nw, unsigned *cnt) {
  long long w = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (w < nw) cnt[w] = (unsigned)__popc(words[w]);
}
__global__ __launch_bounds__(BLOCK) void k_rank_assign(const int *id, long long n, const unsigned *words, const unsigned *before,
                                                       int *rank, int offset) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const unsigned v = (unsigned)id[i], w = v >> 5, bit = v & 31u;
  rank[i] = offset + (int)(before[w] + (unsigned)__popc(words[w] & ((1u << bit) - 1u)));
}

// The member of an ensemble diffusivity whose COLUMN an element mixes on: the member of its element values in the main-loop
// get_environment call, rank % M (readers/interpolation/structured.py:119-135: `horizontal[:, elnum] = int_full[:, elnum]`;
// the environment_profiles of the survivors go with them through remove_deactivated_elements).  Parked in property slot 8
// until odr_vmix: the ranks themselves are renumbered by the Runge-Kutta stage calls of the same step.
__global__ __launch_bounds__(BLOCK) void k_kmember(const int *rank, long long n, int members, float *out) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) out[i] = (float)(rank[i] % members);
}

// Grid-stride over the 256-element chunks: the per-chunk counts feed the scan; the grand total is ONE atomic per
// workgroup (one per chunk = 39 063 serialised atomics on one address for 10 M elements cost 0.47 ms).
__global__ __launch_bounds__(BLOCK) void k_cmp_count(const int *status, long long n, unsigned *bcount,
                                                     unsigned long long *flags, unsigned long long *total = nullptr) {
  const long long nchunks = (n + BLOCK - 1) / BLOCK;
  constexpr int U = 4;           // chunks per pass: U independent loads per thread, one barrier pair per pass
  __shared__ unsigned wc[U][BLOCK / 64];
  unsigned long long mine = 0;   // threads 0 .. U-1: elements that stay, over this workgroup's chunks
  for (long long c0 = (long long)blockIdx.x * U; c0 < nchunks; c0 += (long long)gridDim.x * U) {
    int st[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = (c0 + u) * BLOCK + threadIdx.x;
      st[u] = i < n ? status[i] : -1;      // -1: past the end (neither kept nor a pending reason)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (flags && __ballot(st[u] >= 100 && st[u] < 164)) {   // rare: only while a reason waits for its first occurrence
        if (st[u] >= 100 && st[u] < 164) atomicOr(flags, 1ull << (st[u] - 100));
      }
      const unsigned long long b = __ballot(st[u] == 0);
      if ((threadIdx.x & 63) == 0) wc[u][threadIdx.x >> 6] = (unsigned)__popcll(b);
    }
    __syncthreads();
    if (threadIdx.x < U && c0 + threadIdx.x < nchunks) {
      unsigned cnt = 0;
#pragma unroll
      for (int w = 0; w < BLOCK / 64; ++w) cnt += wc[threadIdx.x][w];
      bcount[c0 + threadIdx.x] = cnt;
      mine += cnt;
    }
    __syncthreads();
  }
  if (threadIdx.x < 64) {        // threads 0 .. U-1 hold partial sums: one atomic per workgroup
    mine += __shfl_down(mine, 2, 64);
    mine += __shfl_down(mine, 1, 64);
    if (threadIdx.x == 0 && total && mine) atomicAdd(total, mine);   // the number of elements that stay
  }
}

// The counts of k_cmp_count from the per-wave counts a step launch left (StepDesc.wcount): bcount[c] = the four waves of chunk c.
// The workgroup that finishes last (ticket counter) writes the number of elements that stay and the status flags the step launch
// collected straight into page-locked host memory (host_out[0], [1]): the host waits for this kernel, not for a copy behind it.
// work[0] = running total, work[2] = tickets (both zeroed by the step call; work[1] = the flags).
__global__ __launch_bounds__(1024) void k_cmp_total(const unsigned *__restrict__ wcount, long long nw, unsigned *bcount,
                                                    unsigned long long *work, unsigned long long *host_out, long long n) {
  constexpr int WPC = BLOCK / 64;   // waves per chunk
  static_assert(WPC == 4, "one 16-byte load per chunk");
  const long long nchunks = (nw + WPC - 1) / WPC;
  unsigned long long mine = 0;
  constexpr int U = 4;              // chunks per thread and pass: independent loads in flight
  const long long stride = (long long)gridDim.x * 1024;
  for (long long c0 = (long long)blockIdx.x * 1024 + threadIdx.x; c0 < nchunks; c0 += stride * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long c = c0 + stride * u;
      v[u] = c < nchunks ? ((const uint4 *)wcount)[c] : make_uint4(0u, 0u, 0u, 0u);   // (the array is padded to whole chunks)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long c = c0 + stride * u;
      if (c >= nchunks) continue;
      const long long k = c * WPC;
      const unsigned cnt = v[u].x + (k + 1 < nw ? v[u].y : 0u) + (k + 2 < nw ? v[u].z : 0u) + (k + 3 < nw ? v[u].w : 0u);
      bcount[c] = cnt;
      mine += cnt;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  __shared__ unsigned long long sh[16];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < 16; ++w) t += sh[w];
    if (t) atomicAdd(&work[0], t);
    __threadfence();
    if (atomicAdd(&work[2], 1ull) == (unsigned long long)gridDim.x - 1) {   // every other workgroup's sum is in
      const unsigned long long tot = atomicAdd(&work[0], 0ull);
      work[3] = tot == (unsigned long long)n ? 1ull : 0ull;   // "every element stays": what a guarded mixing launch reads
      host_out[0] = tot;
      host_out[1] = work[1];
      __threadfence_system();
    }
  }
}

// Exclusive scan of up to CMP_SCAN_ROWS x 1024 counts by ONE workgroup (more: the sequential sweep of rounds 1-2): thread t
// owns elements t, t + 1024, ... (coalesced, eight loads in flight at once); each row of 1024 is scanned wave by wave with
// shuffles (the exclusive values go back to memory), the 16 x rows wave totals by the first wave, then the offsets are added:
// two barriers in all.  (The per-thread contiguous version made ~40
// dependent round trips each way: 61 us for the 39 063 chunks of 10 M elements, one compaction per step in C5.)
constexpr int CMP_SCAN_ROWS = 256;   // x 1024 chunks x 256 elements = 67 M elements
__global__ __launch_bounds__(1024) void k_cmp_scan(unsigned *bcount, long long nblocks,
                                                   unsigned long long *total) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (nblocks <= (long long)CMP_SCAN_ROWS * 1024) {
    __shared__ unsigned wtot[CMP_SCAN_ROWS * 16];
    const int rows = (int)((nblocks + 1023) / 1024);
    constexpr int G = 8;                       // rows per batch: G independent loads in flight per thread
    for (int j0 = 0; j0 < rows; j0 += G) {
      unsigned v[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const long long k = (long long)(j0 + u) * 1024 + tid;
        v[u] = j0 + u < rows && k < nblocks ? bcount[k] : 0u;
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int j = j0 + u;
        const long long k = (long long)j * 1024 + tid;
        unsigned x = v[u];                     // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (j < rows) {
          if (lane == 63) wtot[j * 16 + wv] = x;
          if (k < nblocks) bcount[k] = x - v[u];   // exclusive inside the wave; the wave's offset is added below
        }
      }
    }
    __syncthreads();
    const int nw = rows * 16;                  // wave totals in element order: row-major (row j, wave w)
    if (wv == 0) {                             // exclusive scan of the wave totals by the first wave, 64 at a time
      unsigned carry = 0;
      for (int b0 = 0; b0 < nw; b0 += 64) {
        const unsigned t0 = b0 + lane < nw ? wtot[b0 + lane] : 0u;
        unsigned x = t0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (b0 + lane < nw) wtot[b0 + lane] = carry + x - t0;
        carry += __shfl(x, 63, 64);
      }
      if (lane == 0) *total = carry;
    }
    __syncthreads();
    for (int j0 = 0; j0 < rows; j0 += G) {
      unsigned v[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const long long k = (long long)(j0 + u) * 1024 + tid;
        v[u] = j0 + u < rows && k < nblocks ? bcount[k] : 0u;
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int j = j0 + u;
        const long long k = (long long)j * 1024 + tid;
        if (j < rows && k < nblocks) bcount[k] = v[u] + wtot[j * 16 + wv];
      }
    }
    return;
  }
  __shared__ unsigned long long part[1024];
  long long per = (nblocks + 1023) / 1024, lo = tid * per, hi = lo + per < nblocks ? lo + per : nblocks;
  unsigned long long s = 0;
  for (long long k = lo; k < hi; ++k) s += bcount[k];
  part[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
    unsigned long long v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  unsigned long long run = tid ? part[tid - 1] : 0;
  for (long long k = lo; k < hi; ++k) { unsigned c = bcount[k]; bcount[k] = (unsigned)run; run += c; }
  if (tid == 1023) *total = part[1023];
}

// two-level exclusive scan for large bin counts (sort histogram): per-1024 block sums,
// scan of the sums by k_cmp_scan, then add back
__global__ __launch_bounds__(1024) void k_scan_local(unsigned *a, long long n, unsigned *bsum) {
  __shared__ unsigned sh[1024];
  long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
  unsigned v = i < n ? a[i] : 0;
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    unsigned t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  if (i < n) a[i] = sh[threadIdx.x] - v;  // exclusive within the block
  if (threadIdx.x == 1023) bsum[blockIdx.x] = sh[1023];
}
__global__ __launch_bounds__(1024) void k_scan_add(unsigned *a, long long n, const unsigned *boff) {
  long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
  if (i < n) a[i] += boff[blockIdx.x];
}

struct CmpArrays {
  int n64, n32;
  const double *src64[8]; double *dst64[8]; double *dead64[8];  // lon lat z plon plat slon slat
  const int *src32[40];   int *dst32[40];   int *dead32[40];
};

__global__ __launch_bounds__(BLOCK) void k_cmp_scatter(const int *status, long long n,
                                                       const unsigned *boff, CmpArrays A,
                                                       long long dead_base) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  bool valid = i < n, keep = valid && status[i] == 0;
  unsigned long long b = __ballot(keep);
  __shared__ unsigned wc[BLOCK / 64];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wc[w] = (unsigned)__popcll(b);
  __syncthreads();
  unsigned before = 0;
  for (int k = 0; k < w; ++k) before += wc[k];
  unsigned rank_keep = boff[blockIdx.x] + before + (unsigned)__popcll(b & ((1ull << lane) - 1));
  if (!valid) return;
  // rank among the removed = i - (number kept before i)
  long long kept_before = (long long)boff[blockIdx.x] + before + __popcll(b & ((1ull << lane) - 1));
  long long dst = keep ? (long long)rank_keep : dead_base + (i - kept_before);
  for (int k = 0; k < A.n64; ++k) {
    double v = A.src64[k][i];
    if (keep) A.dst64[k][dst] = v; else if (A.dead64[k]) A.dead64[k][dst] = v;
  }
  for (int k = 0; k < A.n32; ++k) {
    int v = A.src32[k][i];
    if (keep) A.dst32[k][dst] = v; else if (A.dead32[k]) A.dead32[k][dst] = v;
  }
}

// deactivate_outside (basemodel/__init__.py:2354-2382): elements beyond the user's validity domain
// (drift:deactivate_west_of / east_of / south_of / north_of) get the status 'outside'
// Variables.lonlat2xy of one reader for n positions (variables.py:111-143 after modulate_longitude :259-280;
// StructuredReader.lonlat2xy structured.py:438-472 for readers without a projection)
__global__ __launch_bounds__(BLOCK) void k_lonlat2xy(const DevWorld *W, int sid, long long n,
                                                      const double *__restrict__ lon, const double *__restrict__ lat,
                                                      double *__restrict__ xo, double *__restrict__ yo) {
  const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const DevSource &s = W->src[sid];
  double lo = lon[i], x, y;
  if (s.lon_mode == 1) lo = np_mod(lo + 180.0, 360.0) - 180.0;
  else if (s.lon_mode == 2) lo = np_mod(lo, 360.0);
  proj_fwd_rt(s.proj, lo, lat[i], x, y);
  xo[i] = x;
  yo[i] = y;
}

__global__ __launch_bounds__(BLOCK) void k_deactivate_outside(PView p, double W, double E, double S, double N,
                                                              int useW, int useE, int useS, int useN, int wrap360,
                                                              int code) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  double lon = p.lon[i], lat = p.lat[i];
  if (wrap360 && lon < 0) lon += 360.0;  // E given in the 0-360 convention and elements wrapped to -180..180 (:2359-2370)
  bool out = (useW && lon < W) || (useE && lon > E) || (useS && lat < S) || (useN && lat > N);
  if (out) {
    if (p.status[i] == 0) p.status[i] = code;
    p.moving[i] = 0;
  }
}

// In-place compaction (the default): the deactivated elements are copied to the deactivated store
// and the holes they leave among the first `kept` slots are filled with the active elements of the
// tail -- O(#removed) data movement instead of rewriting every array.  Deterministic pairing: the
// k-th hole (by index) takes the k-th active tail element counted from the end.  The relative order
// of the survivors changes (IDs identify elements; the reference's move_elements order is not part
// of any result).
__global__ __launch_bounds__(BLOCK) void k_cmp_lists(const int *status, long long n, const unsigned *boff,
                                                     long long kept, CmpArrays A, long long dead_base,
                                                     unsigned *holes, unsigned *fills) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  bool valid = i < n, keep = valid && status[i] == 0;
  unsigned long long b = __ballot(keep);
  __shared__ unsigned wc[BLOCK / 64];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wc[w] = (unsigned)__popcll(b);
  __syncthreads();
  unsigned before = 0;
  for (int k = 0; k < w; ++k) before += wc[k];
  if (!valid) return;
  long long kept_before = (long long)boff[blockIdx.x] + before + __popcll(b & ((1ull << lane) - 1));
  if (!keep) {
    long long r = i - kept_before;  // rank among the removed
    for (int k = 0; k < A.n64; ++k) if (A.dead64[k]) A.dead64[k][dead_base + r] = A.src64[k][i];
    for (int k = 0; k < A.n32; ++k) if (A.dead32[k]) A.dead32[k][dead_base + r] = A.src32[k][i];
    if (i < kept) holes[r] = (unsigned)i;
  } else if (i >= kept) {
    fills[kept - kept_before - 1] = (unsigned)i;  // active elements after i
  }
}

__global__ __launch_bounds__(BLOCK) void k_cmp_move(const unsigned *__restrict__ holes, const unsigned *__restrict__ fills,
                                                    long long cnt, CmpArrays A) {
  long long k = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (k >= cnt) return;
  unsigned d = holes[k], s = fills[k];
  if (d == 0xFFFFFFFFu || s == 0xFFFFFFFFu) return;
  for (int a = 0; a < A.n64; ++a) const_cast<double *>(A.src64[a])[d] = A.src64[a][s];
  for (int a = 0; a < A.n32; ++a) const_cast<int *>(A.src32[a])[d] = A.src32[a][s];
}

// ---------------------------------------------------------------------- Leeway
// Leeway.update (models/leeway.py:430-494) without capsizing: downwind / crosswind leeway from
// the wind (float32 arithmetic of the LeewayObj properties), update_positions(-x_leeway,
// y_leeway), update_positions(current), then random jibing (sign flip of crosswind_slope).
enum { AUX_DW_SLOPE = 0, AUX_CW_SLOPE, AUX_DW_OFFSET, AUX_CW_OFFSET, AUX_DW_EPS, AUX_CW_EPS, AUX_JIBE_P,
       AUX_ORIENTATION, AUX_CAPSIZED };
constexpr unsigned long long RNG_OFF_JIBE = 4608;

// one element of Leeway.update: leeway + current moves of (lon, lat), jibing; xw, yw, u, v = its sampled environment
__device__ __forceinline__ void leeway_body(const PView &p, long long i, double &lon, double &lat, int moving, float xw, float yw,
                                            float u, float v, double dt, float capsize_fraction, int rng_mode,
                                            const double *__restrict__ huni, unsigned long long seed, unsigned long long step) {
  float windspeed = speed_f32(xw, yw);
  float dwe = p.aux[AUX_DW_EPS][i], cwe = p.aux[AUX_CW_EPS][i];
  float cws = p.aux[AUX_CW_SLOPE][i];
  // ((slope + eps/20.0)*windspeed + offset + eps/2.0)*.01, float32 left to right (:458-466)
  float dw = __fmul_rn(__fadd_rn(__fadd_rn(__fmul_rn(__fadd_rn(p.aux[AUX_DW_SLOPE][i], __fdiv_rn(dwe, 20.0f)), windspeed),
                                           p.aux[AUX_DW_OFFSET][i]), __fdiv_rn(dwe, 2.0f)), (float).01);
  float cw = __fmul_rn(__fadd_rn(__fadd_rn(__fmul_rn(__fadd_rn(cws, __fdiv_rn(cwe, 20.0f)), windspeed),
                                           p.aux[AUX_CW_OFFSET][i]), __fdiv_rn(cwe, 2.0f)), (float).01);
#ifdef ODR_ABL_LW_NOTRIG
  float sinth = xw / windspeed, costh = yw / windspeed;
#else
  // np.sin / np.cos of the float32 winddir = float32(theta), theta = arctan2(x_wind, y_wind): sin / cos theta are x/h, y/h and
  // the float32 rounding moves the angle by delta = winddir - theta, |delta| < 2e-7:
  //   sin(theta + delta) = sin theta (1 - delta^2/2) + cos theta delta   (as azimuth_sincos_f32; delta^3/6 < 2e-21)
  // -- the float64 sine and cosine of winddir to round-off, then rounded to float32 like the library results were: one
  // arctan2 on its finite path instead of arctan2 + sin + cos (0.03 of the 0.73 ms of C5's launch).  Calm / non-finite wind: library calls.
  float sinth, costh;
  {
    const double x = (double)xw, y = (double)yw, h2 = x * x + y * y;
    if (h2 > 0 && h2 < 1.7e308) {
#pragma clang fp contract(fast)
      const double theta = atan2_fin(x, y);
      const double delta = (double)(float)theta - theta;
      const double rh = fast_rsqrt(h2), st = x * rh, ct = y * rh, c2 = 1 - 0.5 * delta * delta;
      sinth = (float)fma(ct, delta, st * c2);
      costh = (float)fma(-st, delta, ct * c2);
    } else {
      const float winddir = (float)atan2(x, y);          // np.arctan2 on float32
      sinth = (float)sin((double)winddir);
      costh = (float)cos((double)winddir);
    }
  }
#endif
  float yl = __fadd_rn(__fmul_rn(dw, costh), __fmul_rn(cw, sinth));
  float xl = __fadd_rn(__fmul_rn(-dw, sinth), __fmul_rn(cw, costh));
  if (p.aux[AUX_CAPSIZED][i] == 1.0f) { xl = __fmul_rn(xl, capsize_fraction); yl = __fmul_rn(yl, capsize_fraction); }
  MoveChain mc;
  mc.next = false;
  move_f32_chain(mc, lon, lat, -xl, yl, moving, dt);              // :472
#ifndef ODR_ABL_LW_NOMOVE2
  move_f32_chain(mc, lon, lat, u, v, moving, dt);                 // :475-476 (start-point coefficients from the first move's)
#endif
#ifdef ODR_ABL_LW_NOJIBE
  return;
#endif
  // jibing (:478-487): rate = -log(1-p)/3600, probability per step 1-exp(-rate*|dt|), float32
  float jp = p.aux[AUX_JIBE_P][i];
  const double q1 = (double)__fsub_rn(1.0f, jp);                 // np.log of a float32: the float64 logarithm rounded to float32
  float rate = __fdiv_rn(-(float)((q1 > 0 && q1 < 1.7e308) ? log_pos(q1) : log(q1)), 3600.0f);
  const double ea = (double)__fmul_rn(-rate, (float)fabs(dt));   // (a few 1e-3 for the jibing rates of OBJECTPROP.DAT: the series)
  float pstep = __fsub_rn(1.0f, (float)(fabs(ea) <= 0.0101 ? exp_small(ea) : exp(ea)));
  double u01;
  if (rng_mode == 1) u01 = huni[i];
  else {
    const uint4 b = rng_block(seed, p.id[i], step, RNG_OFF_JIBE);
    u01 = rng_u53(b.x, b.y);
  }
  if ((double)pstep > u01) {
    p.aux[AUX_CW_SLOPE][i] = -cws;
    p.aux[AUX_ORIENTATION][i] = 1.0f - p.aux[AUX_ORIENTATION][i];
  }
}

__global__ __launch_bounds__(BLOCK) void k_leeway(PView p, double dt, float capsize_fraction, int rng_mode,
                                                  const double *__restrict__ huni, unsigned long long seed,
                                                  unsigned long long step) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  double lon = p.lon[i], lat = p.lat[i];
  leeway_body(p, i, lon, lat, p.moving[i], p.env[VAR_XWIND][i], p.env[VAR_YWIND][i], p.env[VAR_U][i], p.env[VAR_V][i], dt,
              capsize_fraction, rng_mode, huni, seed, step);
  p.lon[i] = lon;
  p.lat[i] = lat;
}

// The Leeway loop body between two compactions in ONE launch (round 3; C5 was five launches, each streaming the particle
// state again): get_environment of the group that holds wind and current from one gridded reader (burst sampler) ->
// drift:current_uncertainty / drift:wind_uncertainty (device RNG: the Philox streams of k_env_noise) -> interact_with_coastline
// -> Leeway.update.  Elements the coastline deactivates are flagged and left where they are (the reference removes them
// before update()): fused call + compact is bit-identical to sample, noise, noise, coastline, compact, leeway
// (tests/test_gpu_fused_step.py).  The group is ordered (x_wind, y_wind, current x, current y[, land, ...]) by the host.
struct LeewayStep {
  int coast_action, stranded_code, seeded_code, land_slot;   // land_slot: group slot of land_binary_mask or -1 (p.env[LAND])
  int wind_slot, uv_slot, store_previous, pad;
  double std_current, std_wind;
  float capsize_fraction, pad2;
  unsigned long long seed, step;
  // report_missing_variables (odr_leeway_set_missing_code): NaN in a sampled variable whose fallback is None
  int missing_code, nmiss_grp, nmiss_rest, pad3;
  int miss_grp[MAXG], miss_rest[4];    // group slots / variable ids (sampled by the preceding launch) to test for NaN
  unsigned *wcount;                    // the loop's status scan formed by this launch (StepDesc.wcount / sflags)
  unsigned long long *sflags;
};
template <int PROJ>
__global__ __launch_bounds__(BLOCK, ODR_WAVES(PROJ)) void k_step_leeway(const DevWorld *__restrict__ W, PView p, EnvGroupDesc G,
                                                                        LeewayStep S, double dt, unsigned long long *n_hit) {
  long long i = pid();
  bool hit = false;
  if (i < p.n) {
    double lon = p.lon[i], lat = p.lat[i];
    const double z = p.z[i];
    int moving = p.moving[i];
    int st = p.status[i];
    const float age0 = p.age[i];
    const int id = p.id[i];
    float out[MAXG];
    ZBracket zb;
    env_group_fast<PROJ, true, false>(*W, G, lon, lat, z, out, nullptr, zb);   // (DevWorld::f32pos: the host takes the separate launches)
    float xw = pick_slot(out, S.wind_slot), yw = pick_slot(out, S.wind_slot + 1);
    float u = pick_slot(out, S.uv_slot), v = pick_slot(out, S.uv_slot + 1);
    // environment.py:869-891: env[x] += N(0, std) (float32 array += float64 draws), first the current, then the wind
#ifndef ODR_ABL_LW_NONOISE   // (what-if builds: tools/vbuild_many.py, profiles/r06_ab_variants.txt)
    if (S.std_current > 0) {
      const double2 g = rng_normal2(rng_block(S.seed, id, S.step, RNG_OFF_NOISE + 4ull * (unsigned)VAR_U));
      add_f32_f64(u, v, g.x * S.std_current, g.y * S.std_current);
    }
    if (S.std_wind > 0) {
      const double2 g = rng_normal2(rng_block(S.seed, id, S.step, RNG_OFF_NOISE + 4ull * (unsigned)VAR_XWIND));
      add_f32_f64(xw, yw, g.x * S.std_wind, g.y * S.std_wind);
    }
#endif
#pragma unroll
    for (int k = 0; k < MAXG; ++k) {
      if (k >= G.nv) break;
      float val = out[k];
      if (k == S.wind_slot) val = xw; else if (k == S.wind_slot + 1) val = yw;
      else if (k == S.uv_slot) val = u; else if (k == S.uv_slot + 1) val = v;
      G.out_ptr[k][i] = val;
    }
    p.slon[i] = lon;
    p.slat[i] = lat;
    if (S.missing_code) {  // k_deactivate_missing (a NaN stays one under the uncertainty draws)
      bool miss = false;
#pragma unroll
      for (int k = 0; k < MAXG; ++k)
        if (k < S.nmiss_grp) { const float e = pick_slot(out, S.miss_grp[k]); miss |= e != e; }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < S.nmiss_rest) { const float e = p.env[S.miss_rest[k]][i]; miss |= e != e; }
      if (miss) {
        if (st == 0) p.status[i] = st = S.missing_code;
        p.moving[i] = moving = 0;
      }
    }
    if (S.coast_action) {  // k_coast
      const float land = S.land_slot >= 0 ? pick_slot(out, S.land_slot) : p.env[VAR_LAND][i];
      if (land == 1.0f) {
        hit = true;
        if (S.coast_action == 1) {
          if (z <= 0) {
            if (st == 0) p.status[i] = st = S.stranded_code;
            p.moving[i] = moving = 0;
          }
        } else {
          if (S.seeded_code > 0 && age0 == 0.0f) {
            if (st == 0) p.status[i] = st = S.seeded_code;
            p.moving[i] = moving = 0;
          }
          lon = p.plon[i];
          lat = p.plat[i];
          p.env[VAR_LAND][i] = 0.0f;
        }
      }
    }
    if (S.store_previous) { p.plon[i] = lon; p.plat[i] = lat; }
    if (st == 0) leeway_body(p, i, lon, lat, moving, xw, yw, u, v, dt, S.capsize_fraction, 0, nullptr, S.seed, S.step);
    p.lon[i] = lon;
    p.lat[i] = lat;
  }
  if (S.coast_action) {
    unsigned long long b = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_hit, (unsigned long long)__popcll(b));
  }
  if (S.wcount) {   // (as at the end of k_step_grid)
    const int s2 = i < p.n ? p.status[i] : 1;
    const unsigned long long kb = __ballot(s2 == 0);
    if (__ballot(s2 >= 100 && s2 < 164)) {
      if (s2 >= 100 && s2 < 164) atomicOr(S.sflags, 1ull << (s2 - 100));
    }
    if ((threadIdx.x & 63) == 0 && i < p.n) S.wcount[i >> 6] = (unsigned)__popcll(kb);
  }
}

// processes:capsizing (models/leeway.py:438-455): before the leeway is formed, an element that can be capsized
// (capsized == 0 in a forward run; == 1, i.e. un-capsizing, in a backward run) flips with probability
// (0.5 + 0.5 tanh((windspeed - threshold) / sigma)) * |dt| / 3600, float32 like the reference's array expression
constexpr unsigned long long RNG_OFF_CAPSIZE = 4864;
__global__ __launch_bounds__(BLOCK) void k_capsize(PView p, double dt, float threshold, float sigma, int rng_mode,
                                                   const double *__restrict__ huni, unsigned long long seed,
                                                   unsigned long long step) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  const float from = dt >= 0 ? 0.0f : 1.0f;
  const float cap = p.aux[AUX_CAPSIZED][i];
  if (cap != from) return;
  const float ws = speed_f32(p.env[VAR_XWIND][i], p.env[VAR_YWIND][i]);
  const float th = (float)tanh((double)__fdiv_rn(__fsub_rn(ws, threshold), sigma));
  const float prob = __fmul_rn(__fadd_rn(0.5f, __fmul_rn(0.5f, th)), (float)(fabs(dt) / 3600));
  double u01;
  if (rng_mode == 1) u01 = huni[i];
  else {
    const uint4 b = rng_block(seed, p.id[i], step, RNG_OFF_CAPSIZE);
    u01 = rng_u53(b.x, b.y);
  }
  if (u01 < (double)prob) p.aux[AUX_CAPSIZED][i] = 1.0f - cap;
}

// ------------------------------------------------------------ spatial re-ordering
// Memory order of the particle SoA is a device-side layout choice (particles are identified
// by ID, the RNG is counter-based on ID): binning the particles by the grid cell of one
// gridded reader makes the 64 lanes of a wave touch neighbouring grid nodes, so the block
// gathers of a wave fall into a handful of cache lines (k_advect_grid: 7.4 ms -> 1.2 ms for
// 10 M particles on a 1024x1024x12 block).  Bins = 8x8-cell tiles, cells row-major inside.
// zb > 1: every cell is split into zb depth bands (lpb reader levels each): elements of a cell that sit in the same band
// become neighbours in memory -- the lanes of a wave then read the same 64-byte sectors of the node records (the texture
// addresser merges lanes that share a sector: a 16-byte gather costs 16 instead of 32 cycles, profiles/r03_gather_bench.txt)
// and, moving with the same current, they stay neighbours for longer under vertical shear.
__device__ __forceinline__ unsigned sort_key(const DevSource &s, const DevBlock &b, double lon, double lat, double z,
                                             int ntx, unsigned nbins, int zb = 1, int lpb = 1) {
  if (s.lon_mode == 1) lon = np_mod(lon + 180.0, 360.0) - 180.0;
  else if (s.lon_mode == 2) lon = np_mod(lon, 360.0);
  double x, y;
  proj_fwd_rt(s.proj, lon, lat, x, y);
  double xi = (x - b.x0) / b.xspan * (b.nx - 1), yi = (y - b.y0) / b.yspan * (b.ny - 1);
  if (!(xi >= 0 && xi <= b.nx - 1 && yi >= 0 && yi <= b.ny - 1)) return nbins - 1;
  int ix = (int)xi, iy = (int)yi;
  const unsigned cell = (unsigned)(((iy >> 3) * ntx + (ix >> 3)) * 64 + (iy & 7) * 8 + (ix & 7));
  if (zb <= 1) return cell;
  int below = 0;   // reader levels below the element (zasc: ascending, +inf beyond nz)
  for (int k = 0; k < s.nz; ++k) below += s.zasc[k] < z ? 1 : 0;
  int band = (s.nz - below) / lpb;   // 0 = the band at the surface
  band = band < 0 ? 0 : (band > zb - 1 ? zb - 1 : band);
  return cell * (unsigned)zb + (unsigned)band;
}

// Runs of equal keys inside a wave (the particles arrive nearly sorted: neighbouring lanes mostly share their tile) make ONE
// atomic per run instead of one per particle: `head` lanes start a run, run_len = distance to the next head.  With the
// per-particle atomics of rounds 1-2 the two counting kernels took 0.30 + 0.50 ms for 10 M particles (contended same-address
// atomics are serialised); unsorted input (runs of length 1) costs the same number of atomics as before.
struct KeyRun { bool head; unsigned len, start; };
__device__ __forceinline__ KeyRun key_run(unsigned k, bool live) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned prev = __shfl_up(k, 1, 64);
  const bool plive = __shfl_up(live ? 1 : 0, 1, 64) != 0;
  KeyRun r;
  r.head = live && (lane == 0 || !plive || k != prev);
  const unsigned long long heads = __ballot(r.head), alive = __ballot(live);
  // start of this lane's run: the highest head at or below the lane; its end: the next head above the start (or the end of
  // the live lanes)
  const unsigned long long below = heads & (lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull));
  r.start = below ? 63u - (unsigned)__builtin_clzll(below) : 0u;
  const unsigned long long above = heads & ~((r.start == 63 ? ~0ull : ((1ull << (r.start + 1)) - 1ull)));
  const unsigned nlive = (unsigned)__popcll(alive);      // live lanes are the low ones (only the last wave is ragged)
  const unsigned end = above ? (unsigned)__builtin_ctzll(above) : nlive;
  r.len = end - r.start;
  return r;
}

__global__ __launch_bounds__(BLOCK) void k_sort_hist(const DevWorld *__restrict__ W, int sid, int slot, PView p,
                                                     int ntx, unsigned nbins, unsigned *keys, unsigned *hist, int zb, int lpb) {
  const long long i = pid();        // XCD-contiguous: the histogram bins of a region are touched by one L2
  const bool live = i < p.n;
  unsigned k = 0;
  if (live) {
    k = sort_key(W->src[sid], W->src[sid].slot[slot], p.lon[i], p.lat[i], zb > 1 ? p.z[i] : 0.0, ntx, nbins, zb, lpb);
    keys[i] = k;
  }
  const KeyRun r = key_run(k, live);
  if (r.head) atomicAdd(&hist[k], r.len);
}

__global__ __launch_bounds__(BLOCK) void k_sort_perm(const unsigned *__restrict__ keys, long long n,
                                                     unsigned *cursor, unsigned *perm) {
  const long long i = pid();
  const bool live = i < n;
  const unsigned k = live ? keys[i] : 0u;
  const KeyRun r = key_run(k, live);
  unsigned base = 0;
  if (r.head) base = atomicAdd(&cursor[k], r.len);
  base = __shfl(base, (int)r.start, 64);
  if (live) perm[base + ((threadIdx.x & 63u) - r.start)] = (unsigned)i;
}

__global__ __launch_bounds__(BLOCK) void k_gather_perm(const unsigned *__restrict__ perm, long long n, CmpArrays A) {
  // XCD-contiguous destination order: the sources of neighbouring destinations are neighbours too (the particles moved a few
  // cells since the last sort), so the 64-byte sectors a workgroup only partly consumes are finished by workgroups on the SAME
  // L2 (round 2: 4.84 x the useful bytes fetched with the round-robin order)
  long long j = pid();
  if (j >= n) return;
  unsigned src = perm[j];
  // eight gathers in flight, then eight stores (a plain load-store loop pays one memory round trip per array)
  double v64[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) if (k < A.n64) v64[k] = A.src64[k][src];
#pragma unroll
  for (int k = 0; k < 8; ++k) if (k < A.n64) A.dst64[k][j] = v64[k];
  for (int k0 = 0; k0 < A.n32; k0 += 8) {
    int v32[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k0 + k < A.n32) v32[k] = A.src32[k0 + k][src];
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k0 + k < A.n32) A.dst32[k0 + k][j] = v32[k];
  }
}

// ---------------------------------------------------------------- block preparation
// ReaderBlock.__init__ (interpolation/structured.py:50-63): mask non-finite and |v|>1e9
__global__ __launch_bounds__(BLOCK) void k_blk_mask(float *a, size_t n) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  float v = a[i];
  if (!isfinite(v) || v < -1e9f || v > 1e9f) a[i] = __builtin_nanf("");
}
// fill_NaN_towards_seafloor (interpolators.py:203-211): layer k <- layer k-1, sequential in k
__global__ __launch_bounds__(BLOCK) void k_blk_fill_seafloor(float *a, int nz, size_t plane) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= plane) return;
  float prev = a[i];
  for (int k = 1; k < nz; ++k) {
    float v = a[k * plane + i];
    if (isnan(v)) { v = prev; a[k * plane + i] = v; }
    prev = v;
  }
}
// expand_numpy_array (interpolators.py:9-20): one grey_dilation(size=3) of the NaN cells,
// src -> dst, all layers of a variable in one launch.  any_finite[layer] guards the
// reference's "Only NaNs, returning".
__global__ __launch_bounds__(BLOCK) void k_blk_dilate(const float *__restrict__ src, float *__restrict__ dst,
                                                      int nz, int ny, int nx) {
  size_t plane = (size_t)ny * nx, i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= plane * nz) return;
  size_t rem = i % plane;
  int y = (int)(rem / nx), x = (int)(rem % nx);
  const float *s = src + (i - rem);
  float v = src[i];
  if (!isfinite(v)) {
    float best = 0;
    bool have = false;
    for (int dy = -1; dy <= 1; ++dy) {
      int yy = y + dy;
      if (yy < 0 || yy >= ny) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        int xx = x + dx;
        if (xx < 0 || xx >= nx) continue;
        float c = s[(size_t)yy * nx + xx];
        if (isfinite(c) && (!have || c > best)) { best = c; have = true; }
      }
    }
    v = have ? best : __builtin_nanf("");
  }
  dst[i] = v;
}

// The same sweep, W cells of a row per thread (nx % W == 0), one workgroup per row, and -- for the sweeps after the
// first (FIRST = false) -- aware of what the ping-pong target already holds: `dst` is the state TWO sweeps back, a cell
// that is finite there never changes again, so only the cells that are NaN in `dst` are looked at (their value: the
// source cell if it has become finite, its dilation otherwise) and a vector is written only when a cell actually got a
// value.  For a field with few NaN cells a sweep is one read of the target (no neighbour reads, no writes).
// `changed[0]` (written by the previous sweep when it gave any cell a value) = 0: the field has reached its fixed point,
// both ping-pong buffers hold it, this and every later sweep is a no-op.  `changed[1]` is this sweep's own flag.
template <int W, bool FIRST>
__global__ __launch_bounds__(BLOCK) void k_blk_dilate_row(const float *__restrict__ src, float *__restrict__ dst, int ny, int nx,
                                                          int *__restrict__ changed) {
  if (!FIRST && changed[0] == 0) return;
  bool any = false;
  const unsigned row = blockIdx.x;                 // layer * ny + y
  const int y = (int)(row % (unsigned)ny);
  const size_t r0 = (size_t)row * (size_t)nx;
  const float *s = src + (r0 - (size_t)y * (size_t)nx);   // the layer
  struct alignas(4 * W) Vec { float v[W]; };
  for (int xv = threadIdx.x; xv * W < nx; xv += BLOCK) {
    const int x0 = xv * W;
    Vec d;
    if (!FIRST) {
      d = *(const Vec *)(dst + r0 + x0);
      bool all = true;
#pragma unroll
      for (int q = 0; q < W; ++q) all &= isfinite(d.v[q]);
      if (all) continue;
    }
    const Vec c = *(const Vec *)(src + r0 + x0);
    Vec o;
    bool wr = FIRST;
#pragma unroll
    for (int q = 0; q < W; ++q) {
      float v = c.v[q];
      if (!FIRST && isfinite(d.v[q])) { o.v[q] = d.v[q]; continue; }
      if (!isfinite(v)) {
        const int x = x0 + q;
        float best = 0;
        bool have = false;
        for (int dy = -1; dy <= 1; ++dy) {
          const int yy = y + dy;
          if (yy < 0 || yy >= ny) continue;
          for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= nx) continue;
            const float cc = s[(size_t)yy * nx + xx];
            if (isfinite(cc) && (!have || cc > best)) { best = cc; have = true; }
          }
        }
        v = have ? best : __builtin_nanf("");
      }
      o.v[q] = v;
      wr |= isfinite(v);
    }
    if (wr) *(Vec *)(dst + r0 + x0) = o;
    any |= FIRST ? false : wr;
    if (FIRST) {
#pragma unroll
      for (int q = 0; q < W; ++q) any |= !isfinite(c.v[q]) && isfinite(o.v[q]);
    }
  }
  if (any) changed[1] = 1;
}

// ---- block preparation in three passes (round 3; the ten row sweeps above read a variable ~10 times: 2.4 ms per C3 time
// level).  (1) k_blk_mask_fill: mask + fill_NaN_towards_seafloor in place, one thread per water column, and a flag per
// (layer, 44x44-cell tile) whose interior still holds a NaN.  (2) k_blk_dilate_tile: only flagged tiles -- the tile and a
// 10-cell halo (what ten 3x3 sweeps can reach) on chip, the ten sweeps there (stopping at the fixed point), interior written
// to `dst`; cells outside the grid are NaN there, which the dilation ignores like the plain kernel's bounds checks.
// A halo cell's value is exact only for as many sweeps as its distance to the region's border -- exactly what the interior
// needs.  (3) k_blk_to_record(src, fix): a cell that is NaN in src takes fix's value.  Same bits as the plain sweeps
// (tests/test_gpu_async_upload.py).
constexpr int DIL_T = 44, DIL_H = 10, DIL_R = DIL_T + 2 * DIL_H;   // interior, halo, LDS region edge (64)
struct BlkPrep {   // the variables of one time level, staged side by side ([nz][ny][nx] each)
  int nvars, ny, nx, rec, tiles_x, ntiles, pad0, pad1;
  float *src[NVAR], *fix[NVAR];            // staged array; dilated values of its NaN cells (k_blk_dilate_tile)
  int nz[NVAR], cum[NVAR + 1];             // layers; layers of the variables before it (flag rows, flat layer index)
  int off[NVAR], es[NVAR], eo[NVAR];       // record layout (DevBlock)
  unsigned char fill[NVAR], dil[NVAR];     // fill towards the sea floor; dilate (everything but the land mask)
};
__global__ __launch_bounds__(BLOCK) void k_blk_mask_fill(BlkPrep Q, int *__restrict__ tile_flags) {
  const int x = blockIdx.x * BLOCK + threadIdx.x, y = blockIdx.y, kv = blockIdx.z;
  if (x >= Q.nx) return;
  float *a = Q.src[kv];
  const int nz = Q.nz[kv];
  const bool fill = Q.fill[kv];
  int *flags = Q.dil[kv] ? tile_flags + (size_t)Q.cum[kv] * Q.ntiles : nullptr;
  const size_t plane = (size_t)Q.ny * Q.nx, i = (size_t)y * Q.nx + x;
  const int tile = (y / DIL_T) * Q.tiles_x + x / DIL_T;
  float prev = 0;
  for (int k0 = 0; k0 < nz; k0 += 4) {     // four layers in flight: the stores below would otherwise fence every load
    float v4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v4[j] = k0 + j < nz ? a[(k0 + j) * plane + i] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + j;
      if (k >= nz) break;
      const float v0 = v4[j];
      float v = v0;
      bool wr = false;
      if (!isfinite(v) || v < -1e9f || v > 1e9f) { v = __builtin_nanf(""); wr = !isnan(v0); }
      if (fill && k > 0 && isnan(v) && !isnan(prev)) { v = prev; wr = true; }
      if (wr) a[k * plane + i] = v;
      if (flags && isnan(v)) flags[k * Q.ntiles + tile] = 1;
      prev = v;
    }
  }
}
// The 64x64 region lives in REGISTERS: lane = column, each of the 4 waves holds 16 rows of its column (NaN kept as -inf so
// that "largest finite neighbour" is a plain compare chain in the plain kernel's scan order -- row-major, first maximum
// wins).  Left / right neighbours come over DPP wave shifts, the rows above / below a wave's strip through a small LDS
// exchange, one barrier per sweep.  (First version, two 16 KB LDS planes and 9 LDS reads per cell: 256 us per C3
// variable -- the ~3 flagged tiles per CU could not hide the LDS latency -- against ~20 us for this one.)
__device__ __forceinline__ float dpp_from_left(float v, float absent) {    // lane i <- lane i-1; lane 0 <- absent
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(absent), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_right(float v, float absent) {   // lane i <- lane i+1; lane 63 <- absent
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(absent), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__global__ __launch_bounds__(BLOCK) void k_blk_dilate_tile(BlkPrep Q, const int *__restrict__ tile_flags) {
  const int tile = blockIdx.x, flat = blockIdx.y, ny = Q.ny, nx = Q.nx, tiles_x = Q.tiles_x;
  if (!tile_flags[(size_t)flat * Q.ntiles + tile]) return;   // (never set for the land mask)
  int kv = 0;
  while (kv + 1 < Q.nvars && flat >= Q.cum[kv + 1]) ++kv;
  const int layer = flat - Q.cum[kv];
  const float *__restrict__ src = Q.src[kv];
  float *__restrict__ dst = Q.fix[kv];
  constexpr int ROWS = DIL_R / (BLOCK / 64);   // 16 rows per wave
  __shared__ float edge[2][2][BLOCK / 64][64];   // [sweep parity][top / bottom row of the strip][wave][column]
  const int ty0 = (tile / tiles_x) * DIL_T - DIL_H, tx0 = (tile % tiles_x) * DIL_T - DIL_H;
  const float *s = src + (size_t)layer * ny * nx;
  const int c = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float NINF = -__builtin_inff();
  float w[ROWS];
  const int x = tx0 + c;
#pragma unroll
  for (int j = 0; j < ROWS; ++j) {
    const int y = ty0 + ROWS * wv + j;
    float v = (y >= 0 && y < ny && x >= 0 && x < nx) ? s[(size_t)y * nx + x] : NINF;
    w[j] = isnan(v) ? NINF : v;
  }
  auto row3 = [&](float m) {   // first maximum of (left, centre, right)
    const float l = dpp_from_left(m, NINF), r = dpp_from_right(m, NINF);
    float b = l;
    if (m > b) b = m;
    if (r > b) b = r;
    return b;
  };
  for (int it = 0; it < 10; ++it) {
    const int par = it & 1;
    edge[par][0][wv][c] = w[0];
    edge[par][1][wv][c] = w[ROWS - 1];
    __syncthreads();
    const float above = wv > 0 ? edge[par][1][wv - 1][c] : NINF, below = wv < BLOCK / 64 - 1 ? edge[par][0][wv + 1][c] : NINF;
    float h[ROWS + 2];
    h[0] = row3(above);
#pragma unroll
    for (int j = 0; j < ROWS; ++j) h[j + 1] = row3(w[j]);
    h[ROWS + 1] = row3(below);
    bool any = false;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      float b = h[j];
      if (h[j + 1] > b) b = h[j + 1];
      if (h[j + 2] > b) b = h[j + 2];
      if (w[j] == NINF && b > NINF) { w[j] = b; any = true; }
    }
    if (!__syncthreads_or(any)) break;   // fixed point
  }
  if (c >= DIL_H && c < DIL_H + DIL_T && x < nx) {
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const int r = ROWS * wv + j, y = ty0 + r;
      if (r >= DIL_H && r < DIL_H + DIL_T && y < ny)
        dst[(size_t)layer * ny * nx + (size_t)y * nx + x] = w[j] == NINF ? __builtin_nanf("") : w[j];
    }
  }
}

// ------------------------------------------------------------------ output history
// state_to_buffer (basemodel/__init__.py:2384-2403) on the device: the float32 result buffer
// (:2084-2105: every exported variable is a float32 [trajectory, time] array initialised with NaN)
// stays resident in HBM as [time][trajectory][stride] records, one record per element and output
// time holding all exported variables -- the scatter by ID is one contiguous record per element.
// At an output step every element present is written (active ones and those deactivated during the
// step, which the reference removes only afterwards); between output steps only deactivated
// elements are written, into the slot of the next output time (:2390-2397, method='backfill').
enum { HK_F64 = 0, HK_I32 = 1, HK_F32 = 2 };
constexpr int HIST_MAXV = 40;   // OpenOil with every default variable exports 30 (9 element properties + 4 oil properties + 17 environment variables)
struct HistVars {
  int nvars, stride;
  const void *src[HIST_MAXV];
  int kind[HIST_MAXV];
};

template <int NQ>  // record length in 16-byte quads: the record is assembled in registers and stored with NQ 16-byte writes
__global__ __launch_bounds__(BLOCK) void k_hist_record(long long n, const int *__restrict__ id,
                                                       const int *__restrict__ status, HistVars H,
                                                       float *__restrict__ slab, long long ntraj,
                                                       int only_deactivated, long long id_base) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  if (only_deactivated && status[i] == 0) return;
  const long long tr = (long long)id[i] - id_base;   // trajectory row of this buffer (a shard of a sharded run starts at id_base)
  if (tr < 0 || tr >= ntraj) return;
  float rec[4 * NQ];
#pragma unroll
  for (int v = 0; v < 4 * NQ; ++v) {
    float x = __builtin_nanf("");
    if (v < H.nvars) {
      if (H.kind[v] == HK_F64) x = (float)((const double *)H.src[v])[i];   // float64 -> float32 array assignment
      else if (H.kind[v] == HK_I32) x = (float)((const int *)H.src[v])[i];
      else x = ((const float *)H.src[v])[i];
    }
    rec[v] = x;
  }
  float4 *dst = (float4 *)(slab + tr * (4 * NQ));
#pragma unroll
  for (int q = 0; q < NQ; ++q) dst[q] = make_float4(rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]);
}

// one variable of the buffer in the reference's (trajectory, time) layout: out[(tr - tr0) * nt + t]
__global__ __launch_bounds__(BLOCK) void k_hist_extract(const float *__restrict__ buf, long long ntraj, int stride,
                                                        int v, int t0, int nt, long long tr0, long long ntr,
                                                        float *__restrict__ out) {
  long long k = (long long)blockIdx.x * BLOCK + threadIdx.x;  // over ntr * nt, time fastest
  if (k >= ntr * nt) return;
  long long tr = tr0 + k / nt;
  int t = t0 + (int)(k % nt);
  out[k] = buf[((long long)t * ntraj + tr) * stride + v];
}

// var.min(skipna=True) / var.max(skipna=True) over the buffer (:2412-2414); red = {max(-x), max(x)}
__global__ __launch_bounds__(BLOCK) void k_hist_minmax(const float *__restrict__ buf, long long nrec, int stride, int v,
                                                       double *red) {
  double mn = -__builtin_inf(), mx = -__builtin_inf();
  for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < nrec; k += (long long)gridDim.x * BLOCK) {
    float x = buf[k * stride + v];
    if (x == x) { mn = fmax(mn, -(double)x); mx = fmax(mx, (double)x); }
  }
  mn = wave_max(mn); mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) {
    if (mn > -__builtin_inf()) atomic_max_d(&red[0], mn);
    if (mx > -__builtin_inf()) atomic_max_d(&red[1], mx);
  }
}

__global__ __launch_bounds__(BLOCK) void k_fill_u32(unsigned *a, size_t n, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) a[i] = v;
}

// ------------------------------------------------------------- ROMS sigma -> z regridding
// What reader_ROMS_native.get_variables does to every 4-D variable of a block before handing it to
// the ReaderBlock, one thread per water column, float64 in NumPy's
// operation order:
//   k_roms_zrho : roppy sdepth (depth.py:31-113, stagger 'rho', Vtransform 1 / 2) minus zeta, positive
//                 depths -> NaN (reader_ROMS_native.py:512-538);
//   k_roms_zslice: roppy multi_zslice (depth.py:213-284) + "R > 1e9 -> NaN" (:683-684); the float32
//                 result goes straight into the block upload.
__global__ __launch_bounds__(BLOCK) void k_roms_zrho(const double *__restrict__ H, const double *__restrict__ zeta,
                                                     double Hc, const double *__restrict__ C,
                                                     const double *__restrict__ S, int N, long long M,
                                                     int vtransform, double *__restrict__ zr) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= M) return;
  const double h = H[i], ze = zeta ? zeta[i] : 0.0;
  const double denom = __dadd_rn(1.0, __ddiv_rn(Hc, h));
  for (int k = 0; k < N; ++k) {
    double zo;
    if (vtransform == 1) zo = __dadd_rn(__dmul_rn(Hc, __dsub_rn(S[k], C[k])), __dmul_rn(C[k], h));
    else zo = __ddiv_rn(__dadd_rn(__dmul_rn(Hc, S[k]), __dmul_rn(C[k], h)), denom);
    double z = __dsub_rn(__dadd_rn(zo, __dmul_rn(ze, __dadd_rn(1.0, __ddiv_rn(zo, h)))), ze);
    zr[(long long)k * M + i] = z;
  }
}
// np.nanmax(z_rho) > 0 -> z_rho[z_rho > 0] = NaN (two passes: the flag is global)
__global__ __launch_bounds__(BLOCK) void k_roms_zrho_positive(double *__restrict__ zr, long long n, int *flag, int apply) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  if (zr[i] > 0) {
    if (apply) zr[i] = __builtin_nan("");
    else *flag = 1;
  }
}

// KMAX = compile-time bound of the number of z levels: the counts C[j] = #(S < Z[j]) of all levels are
// accumulated in registers during ONE sweep over the column's s-level depths (each read once, coalesced
// across the wave); the two bracketing levels of every z are then re-read (cache hits).
template <typename TF, int KMAX>
__global__ __launch_bounds__(BLOCK) void k_roms_zslice(const TF *__restrict__ F, const double *__restrict__ zr,
                                                       const double *__restrict__ Z, int N, int kmax, long long M,
                                                       float *__restrict__ out32, double *__restrict__ out64) {
  long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= M) return;
  double zl[KMAX];
  int cnt[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) { zl[j] = j < kmax ? Z[j] : -__builtin_inf(); cnt[j] = 0; }
  for (int k = 0; k < N; ++k) {
    const double sv = zr[(long long)k * M + i];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) cnt[j] += sv < zl[j] ? 1 : 0;   // NaN compares false, as in NumPy
  }
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j < kmax) {
      const double z = zl[j];
      int c = cnt[j];
      c = c < 1 ? 1 : (c > N - 1 ? N - 1 : c);
      const double s0 = zr[(long long)(c - 1) * M + i], s1 = zr[(long long)c * M + i];
      double a = __ddiv_rn(__dsub_rn(z, s0), __dsub_rn(s1, s0));
      a = a < 0.0 ? 0.0 : (a > 1.0 ? 1.0 : a);                               // np.clip keeps NaN
      const double f0 = (double)F[(long long)(c - 1) * M + i], f1 = (double)F[(long long)c * M + i];
      double r = __dadd_rn(__dmul_rn(__dsub_rn(1.0, a), f0), __dmul_rn(a, f1));
      if (r > 1e9) r = __builtin_nan("");
      if (out32) out32[(long long)j * M + i] = (float)r;
      if (out64) out64[(long long)j * M + i] = r;
    }
  }
}

// final device layout of a block: one record per grid node holding every variable of the reader
// at that node, z innermost -- element (k, node) of a variable at record offset `off` lives at
// dst[node * rec + off + k * es + eo] (es = 2, eo = 0/1 for the two components of a vector pair).
// src is the reader's [nz][ny][nx] array.
__global__ __launch_bounds__(BLOCK) void k_blk_to_record(const float *__restrict__ src, float *__restrict__ dst,
                                                        int nz, size_t plane, int rec, int off, int es, int eo,
                                                        const float *__restrict__ fix = nullptr) {
  // 64 nodes per workgroup through LDS: reads are coalesced along the nodes of one level, writes run along the
  // levels of one node (the record's contiguous direction) instead of 64 lanes hitting 64 different records
  __shared__ float t[64][MAXNZ + 1];
  const size_t n0 = (size_t)blockIdx.x * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k = w; k < nz; k += BLOCK / 64)
    if (n0 + lane < plane) {
      float v = src[(size_t)k * plane + n0 + lane];
      if (fix && isnan(v)) v = fix[(size_t)k * plane + n0 + lane];   // dilated value of a cell that had none (k_blk_dilate_tile)
      t[lane][k] = v;
    }
  __syncthreads();
  const int total = 64 * nz;
  for (int q = threadIdx.x; q < total; q += BLOCK) {
    const int node = q / nz, k = q - node * nz;
    if (n0 + node < plane) dst[(n0 + node) * rec + off + k * es + eo] = t[node][k];
  }
}

// The record writer of the three-pass preparation: 64 nodes per workgroup, the COMPLETE records of those nodes assembled
// in LDS (every layer of every variable read along the nodes: 256-byte coalesced reads; NaN cells take the dilated value)
// and written as one contiguous run of 64 x rec floats -- instead of one strided pass over the block per variable.
// Records longer than REC_CH floats go in chunks.  Padding floats are written as 0.
constexpr int REC_CH = 128;
__global__ __launch_bounds__(BLOCK) void k_blk_records(BlkPrep Q, float *__restrict__ dst, size_t plane) {
  extern __shared__ float t[];   // [64][cs], cs = min(rec, REC_CH) | 1: lane-strided writes and row reads without bank conflicts
  // record position -> plane of that (variable, layer) in the staging area, and the plane of its dilated values (or null):
  // looked up per read instead of walking the descriptor (chains of dependent scalar loads: 0.36 ms per C3 level)
  __shared__ const float *tab_src[REC_CH], *tab_fix[REC_CH];
  const size_t n0 = (size_t)blockIdx.x * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, rec = Q.rec;
  const int cs = (rec < REC_CH ? rec : REC_CH) | 1;
  const bool live = n0 + lane < plane;
  const int total = Q.cum[Q.nvars];
  constexpr int NW = BLOCK / 64, UN = 8;
  for (int r0 = 0; r0 < rec; r0 += REC_CH) {
    const int cw = rec - r0 < REC_CH ? rec - r0 : REC_CH;
    if (r0) __syncthreads();
    for (int pp = threadIdx.x; pp < cw; pp += BLOCK) { tab_src[pp] = nullptr; tab_fix[pp] = nullptr; }
    for (int node = w; node < 64; node += NW)
      for (int pp = lane; pp < cw; pp += 64) t[node * cs + pp] = 0.f;
    __syncthreads();
    for (int f = threadIdx.x; f < total; f += BLOCK) {
      int kv = 0;
      while (f >= Q.cum[kv + 1]) ++kv;
      const int k = f - Q.cum[kv], pr = Q.off[kv] + k * Q.es[kv] + Q.eo[kv] - r0;
      if (pr >= 0 && pr < cw) {
        tab_src[pr] = Q.src[kv] + (size_t)k * plane;
        tab_fix[pr] = Q.dil[kv] ? Q.fix[kv] + (size_t)k * plane : nullptr;
      }
    }
    __syncthreads();
    for (int p0 = w; p0 < cw; p0 += NW * UN) {   // UN independent plane reads in flight per lane
      float v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int pr = p0 + NW * u;
        const float *sp = pr < cw ? tab_src[pr] : nullptr;
        v[u] = sp && live ? sp[n0 + lane] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int pr = p0 + NW * u;
        if (pr >= cw) break;
        float x = v[u];
        if (isnan(x) && tab_fix[pr]) x = tab_fix[pr][n0 + lane];   // dilated value of a cell that had none
        t[lane * cs + pr] = x;
      }
    }
    __syncthreads();
    for (int node = w; node < 64; node += NW)
      if (n0 + node < plane)
        for (int pp = lane; pp < cw; pp += 64) dst[(n0 + node) * rec + r0 + pp] = t[node * cs + pp];
  }
}

#endif  // ODR_TU_MISC
}  // namespace odr

#include "odr_tile.hip.h"   // the workgroup table of a sorted set (ODR_TU_MISC) and the LDS-tile step kernel (ODR_TU_TILE)
