// Host-side internals shared by the translation units of libodrift_hip.so (odrift.hip: context, particle sets, field
// blocks, environment sample, bookkeeping; odr_step.hip: current advection and the fused step; odr_mix.hip: vertical
// mixing and the OpenOil mixing physics).  The library is built from several TUs so that hipcc compiles them in
// parallel; no device code is shared across TUs (every kernel is defined where it is launched).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/odrift.h"
#include "odr_kernels.hip.h"

using namespace odr;

// ODR_SLOW_CALLS=<ms>: host-side spans (runtime calls that may block: allocations, frees, event waits, stream synchronisations)
// that take longer than that are reported on stderr with their label -- where a run() loop's host stalls come from
#include <chrono>
struct SlowSpan {
  const char *what;
  std::chrono::steady_clock::time_point t0;
  static double limit() { static const double v = getenv("ODR_SLOW_CALLS") ? atof(getenv("ODR_SLOW_CALLS")) : 0.0; return v; }
  explicit SlowSpan(const char *w) : what(w) { if (limit() > 0) t0 = std::chrono::steady_clock::now(); }
  ~SlowSpan() {
    if (limit() > 0) {
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ms > limit()) fprintf(stderr, "[odr slow] %s: %.3f ms\n", what, ms);
    }
  }
};

int odr_i_fail(int code, const char *fmt, ...);   // records the message for odr_last_error(), returns code
#define fail odr_i_fail
#define HIPCHK(x)                                                                          \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) return fail(ODR_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_));   \
  } while (0)
#define REQUIRE(c, ...) do { if (!(c)) return fail(ODR_ERR_INVALID, __VA_ARGS__); } while (0)

struct Staged { DevBlock blk; float *base; size_t bytes; unsigned long long cid[NVAR]; };   // uploaded, not yet committed (cid: odr_block_set_content_ids)
struct Retired { void *ptr; size_t bytes; hipEvent_t ev; };       // replaced block, freed once the compute stream passed

constexpr int ODR_MAX_LANES = 8;
struct odr_ctx {
  int device;
  unsigned long long seed;
  hipStream_t stream, own_stream;
  DevWorld hw;      // host image
  DevWorld *dw;     // device image
  unsigned src_gen[MAXSRC] = {0};   // counts the releases of a source id: what was derived from the source's geometry (the tile step's workgroup table) dies with it
  hipEvent_t scan_ev = nullptr;     // behind the fold of odr_scan_status_begin
  bool scan_open = false;
  int guard_next_vmix = 0;          // odr_ctx_guard_next_vmix
  int vmix_levels = 0;              // odr_vmix_set_profile_levels: one-shot, taken by the next odr_vmix (0 = every level of the reader)
  unsigned long long *scan_host = nullptr;   // page-locked: what odr_scan_status reads (written by k_cmp_total itself)
  // page-locked copies of `hw` the device image is refreshed from (flush_world): three in turn, each guarded by an event
  DevWorld *hw_pin[3] = {nullptr, nullptr, nullptr};
  hipEvent_t hw_ev[3] = {nullptr, nullptr, nullptr};
  int hw_turn = 0;
  bool dirty;
  std::vector<void *> block_bufs[MAXSRC][MAXLEVELS];  // owned device arrays per slot
  size_t block_bytes[MAXSRC][MAXLEVELS];
  // content id of each variable of a resident block (odr_block_set_content_ids; 0 = unknown): two time levels holding the
  // same id for a variable hold the same values, and the samplers read it at one of them
  unsigned long long block_cid[MAXSRC][MAXLEVELS][NVAR];
  // upload pipeline (stage_block / odr_block_commit)
  hipStream_t up_stream;
  hipEvent_t up_done, up_dep;
  float *prep[2];
  size_t prep_floats;
  int *dilate_flags;   // [NVAR][16]: "sweep k gave a cell a value" (k_blk_dilate_row stops at the fixed point)
  int *tile_flags = nullptr;   // [layer][44x44-cell tile]: the tile still holds a NaN after mask + sea-floor fill (k_blk_mask_fill)
  size_t tile_flags_n = 0;
  // page-locked bounce buffers for copies from / to caller memory: [0] compute stream, [1] upload stream (odr_i_h2d)
  void *bounce[2];
  Staged staged[MAXSRC][MAXLEVELS];
  std::vector<Retired> graveyard;
  std::vector<void *> source_bufs;  // device arrays owned by sources (curvilinear node tables)
  std::vector<void *> registered;   // host ranges page-locked by odr_host_register (released with the context)
  double *red;      // device reduction slots
  // current uncertainty of the next odr_advect / odr_env_coast_advect on `noise_owner` (odr_advect_set_noise)
  const odr_particles *noise_owner;
  StageNoise noise;
  double *noise_buf;
  size_t noise_buf_n;
  // OpenOil mixing-loop physics (odr_oil_prepare_mixing): armed for the next odr_vmix* call on `oil_owner`
  const odr_particles *oil_owner;
  OilArgs oil;
  double *oil_stat, *oil_cdf, *oil_chunk, *oil_part, *oil_u;
  int *oil_guide;
  int oil_override;            // odr_oil_set_mixing_stats: means combined over the ranks of a sharded run
  double oil_stat_host[2];
  size_t oil_part_n, oil_u_n;
  unsigned long long *counter;
  hipEvent_t ev0, ev1;
  int nsrc;
  bool src_free[MAXSRC];   // source ids released by odr_source_release, reused by the next new source
  int stage_math;   // odr_ctx_set_stage_math: ODR_STAGE_EXACT | ODR_STAGE_FAST (Runge-Kutta stage evaluations)
  int fuse_vadv;
  int leeway_missing_code;   // odr_leeway_set_missing_code: one-shot, taken by the next odr_env_coast_leeway
  int seafloor;     // general:seafloor_action for the in-update() sea floor checks: action | status_code << 8
  // reductions cached between the horizontal movers of one step (advect_wind -> stokes_drift -> horizontal
  // diffusion read the same maxima: environment, z and properties do not change in between)
  const odr_particles *red_owner;
  unsigned long long red_epoch;
  double red_wdd;
  int red_rel;
  bool red_extents = true;   // the cached reduction holds the lon / lat / z extremes too (the movers' reductions leave them out)
  int red_pinned;   // odr_reduce_install: red[] holds values combined over the ranks of a sharded run
  double *red_rec = nullptr; long long red_rec_cap = 0;   // per-wave records of the step launch
  bool red_partial = false;  // red[] was formed by the step launch (odr_ctx_set_step_reduce): only the slots the movers' == 0 tests read, R_NSURF as a flag
  int step_reduce_on = 0, step_reduce_rel = 0;
  double step_reduce_wdd = 0.1;
  // lanes of the fused step (odr_step.hip, step_in_lanes): contiguous particle ranges on streams of their own
  hipStream_t lane_stream[ODR_MAX_LANES];
  hipEvent_t lane_step[ODR_MAX_LANES], lane_done[ODR_MAX_LANES], lane_fork;
  int lanes_ready;
};

struct odr_particles {
  long long cap, n, ndead, dead_cap;
  long long win;        // first element of the window that view() exposes (0 except inside step_in_lanes)
  int ice_kind;         // odr_set_element_factor
  // ranks of the present elements in ascending ID (ensemble members, odr_i_ensure_ranks)
  int *rank;
  unsigned *rank_words, *rank_before, *rank_bsum;
  long long rank_words_n, id_max;
  int rank_on;
  int rank_sharded = 0;        // odr_particles_set_rank_offset was called: part of a sharded run (members numbered over the active elements of all ranks)
  long long rank_offset = 0;   // odr_particles_set_rank_offset: present elements with smaller IDs held by other particle sets (sharded run)
  double *d64[7];       // lon lat z plon plat slon slat
  double *alt64[7];
  int *i32[3];          // id status moving
  int *alti32[3];
  float *f32[4];        // wdf cdf tv age_seconds
  float *altf32[4];
  float *env[NVAR];
  float *altenv[NVAR];
  float *aux[9];
  float *altaux[9];
  bool profiles_f32 = false; // the sample positions of the last main-loop sample were taken in the float32 position class (odr_vmix)
  unsigned aux_user = 0;     // bit k: property slot k was written by the caller (odr_particles_set_property): the model owns it
  bool kmember_on = false;   // slot AUX_KMEMBER parks the member of an ensemble diffusivity (k_kmember): the library owns it
  float *aux_snap[9];       // odr_particles_snapshot_property: copies the result buffer may read instead of aux[] (one record)
  long long aux_snap_cap[9];
  double *dead64[3];    // lon lat z of the deactivated store
  int *deadi32[2];      // id status
  unsigned *bcount;
  unsigned *wcount;                 // per-wave counts of the last step launch (StepDesc.wcount), valid while wcount_epoch == status_epoch
  unsigned long long wcount_epoch;
  long long wcount_n;
  void *scratch;
  size_t scratch_bytes;
  unsigned long long epoch;  // bumped by every call that changes z, the environment, properties or the element set
  // odr_scan_status -> odr_compact_apply: the count of the scan stays valid while no call that can deactivate elements,
  // add elements or re-order them has run (those bump status_epoch)
  unsigned long long status_epoch, scan_epoch;
  long long scan_kept;
  // environment variables that hold ONE value for every element (no reader: the fallback; a global constant reader):
  // env[v][0 .. env_cn[v]) == env_cval[v] while env_cok[v].  A repeated sample writes only new elements; movers whose
  // global early-out is decided by such a value (wind, Stokes drift, horizontal diffusivity identically 0) return at once.
  float env_cval[NVAR];
  long long env_cn[NVAR];
  bool env_cok[NVAR];
  bool external;
  // workgroup table of the last odr_sort_particles (k_wg_count / k_wg_fill; read by k_step_tile): (first, count) ranges that
  // partition [0, wg_n), each inside one sort tile of source wg_sid.  Valid while no element has been appended (n <= wg_n);
  // compaction only shortens the set (ranges are cut at n; elements moved into holes are found outside their range's
  // rectangle and take the global path).
  unsigned *wg_tab, *wg_list;                // wg_list: the particles the last k_step_tile handed to k_step_list
  unsigned long long *wg_total, *wg_stats;   // device: wg_total[0] table length, [1] length of wg_list; wg_stats = wg_total + 2: [0] sum of the list lengths, [1] rectangles cut
  long long wg_cap, wg_grid, wg_n, wg_list_cap;
  unsigned long long wg_launches;            // host: launches that took the LDS-tile path
  double *z_keep;       // odr_particles_truncate_z: the elements' own z while the sampling calls see the clipped one
  long long z_keep_n;
  bool z_truncated;
  int wg_sid;
  bool wg_valid;
  unsigned wg_src_gen;              // generation of the source the table was built on (odr_ctx::src_gen)
};

static inline unsigned nblk(long long n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

static inline PView view(const odr_particles *p) {
  PView v;
  v.n = p->n;
  const long long w = p->win;
  v.lon = p->d64[0] + w; v.lat = p->d64[1] + w; v.z = p->d64[2] + w; v.plon = p->d64[3] + w; v.plat = p->d64[4] + w;
  v.slon = p->d64[5] + w; v.slat = p->d64[6] + w;
  v.id = p->i32[0] + w; v.status = p->i32[1] + w; v.moving = p->i32[2] + w;
  v.wdf = p->f32[0] + w; v.cdf = p->f32[1] + w; v.tv = p->f32[2] + w; v.age = p->f32[3] + w;
  for (int k = 0; k < NVAR; ++k) v.env[k] = p->env[k] ? p->env[k] + w : nullptr;
  for (int k = 0; k < 9; ++k) v.aux[k] = p->aux[k] ? p->aux[k] + w : nullptr;
  v.ice = p->ice_kind; v.pad = 0;
  v.rank = p->rank_on && p->rank ? p->rank + w : nullptr;
  return v;
}

// the store targets of a variable group for the particle set behind `v` (EnvGroupDesc::out_ptr)
static inline EnvGroupDesc env_bind_out(const EnvGroupDesc &G0, const PView &v) {
  EnvGroupDesc G = G0;
  for (int k = 0; k < MAXG; ++k) G.out_ptr[k] = k < G.nv ? v.env[G.var[k]] : nullptr;
  return G;
}

// The device image of the world follows the host image without a host synchronisation and without the copy engine
// (odrift.hip: flush_world)
int flush_world(odr_ctx *c);
int flush_world_init(odr_ctx *c);

// env[v] is the constant `val` for every element (see odr_particles::env_cok)
static inline bool env_is_const(const odr_particles *p, int v, float val) {
  return p->env[v] && p->env_cok[v] && p->env_cn[v] >= p->n && p->env_cval[v] == val;
}
static inline int ensure_env(odr_ctx *c, odr_particles *p, int var) {
  if (!p->env[var]) {
    HIPCHK(hipMalloc((void **)&p->env[var], sizeof(float) * (size_t)p->cap));
    HIPCHK(hipMemsetAsync(p->env[var], 0, sizeof(float) * (size_t)p->cap, c->stream));
  }
  return 0;
}

static inline int scratch(odr_ctx *c, odr_particles *p, size_t bytes, void **out) {
  if (p->scratch_bytes < bytes) {
    SlowSpan sp("scratch: synchronize + hipFree + hipMalloc");
    HIPCHK(hipStreamSynchronize(c->stream));
    if (p->scratch) HIPCHK(hipFree(p->scratch));
    HIPCHK(hipMalloc(&p->scratch, bytes));
    p->scratch_bytes = bytes;
  }
  *out = p->scratch;
  return 0;
}

// host copy of nearest_time (variables.py:402-443) on the resident levels of one source
static inline void host_bracket(const DevSource &s, double t, int &ib, int &ia) {
  int b = 0;
  for (int k = 0; k < s.nlevels; ++k) if (s.slot[s.level_slot[k]].t <= t) b = k;
  ib = s.level_slot[b];
  ia = (b + 1 < s.nlevels && s.slot[ib].t != t) ? s.level_slot[b + 1] : -1;
}

static inline UVTime uv_time(const DevSource &s, double t) {
  int ib, ia;
  host_bracket(s, t, ib, ia);
  UVTime tm;
  tm.b = s.slot[ib].data[VAR_U];
  tm.a = (ia >= 0 && !s.always_valid) ? s.slot[ia].data[VAR_U] : nullptr;
  tm.w = ia >= 0 ? (t - s.slot[ib].t) / (s.slot[ia].t - s.slot[ib].t) : 0.0;
  return tm;
}


// The fast column path of vertical mixing (k_vmix_col, and the mixing fused into k_step_grid): K from one gridded
// reader with a plain z-innermost array on every resident level of the same geometry.  Fills D for time t.
static inline bool build_vmix_desc(const odr_ctx *c, double t, VMixDesc &D) {
  memset(&D, 0, sizeof D);
  int ksid = -1, nzp = 1;
  for (int k = 0; k < c->hw.nlist[VAR_KZ]; ++k)
    if (c->hw.src[c->hw.list[VAR_KZ][k]].kind == SRC_GRID) { ksid = c->hw.list[VAR_KZ][k]; break; }
  if (ksid < 0) return false;
  const DevSource &s = c->hw.src[ksid];
  nzp = s.nz > 1 ? s.nz : 1;
  if (nzp <= 1 || s.nlevels < 1) return false;
  const DevBlock &g0 = s.slot[s.level_slot[0]];
  for (int k = 0; k < s.nlevels; ++k) {
    const DevBlock &bk = s.slot[s.level_slot[k]];
    if (s.members[VAR_KZ] > 1 || !bk.small || !bk.data[VAR_KZ] || bk.es[VAR_KZ] != 1 || bk.var_nz[VAR_KZ] != nzp || bk.rec != g0.rec || bk.ny != g0.ny || bk.nx != g0.nx ||
        bk.x0 != g0.x0 || bk.xspan != g0.xspan || bk.y0 != g0.y0 || bk.yspan != g0.yspan)
      return false;
  }
  int ib, ia;
  host_bracket(s, t, ib, ia);
  D.sid = ksid; D.nzp = nzp; D.geo_slot = ib;
  D.kb = s.slot[ib].data[VAR_KZ];
  D.ka = ia >= 0 ? s.slot[ia].data[VAR_KZ] : nullptr;
  D.wgt = ia >= 0 ? (t - s.slot[ib].t) / (s.slot[ia].t - s.slot[ib].t) : 0.0;
  D.Kfb = c->hw.fallback[VAR_KZ];
  return true;
}

// ---- copies between CALLER memory and the device go through a page-locked bounce buffer of the context, chunk by chunk.
// Handing pageable caller memory to hipMemcpyAsync lets the runtime pin the caller's pages on the fly for transfers above
// ~1 MB and read them later; with NumPy arrays that live in the malloc heap this produced a sporadic
// "Memory access fault by GPU ... on address <heap address>" (a 1.6 MB array at an address an earlier, freed array had
// been pinned at), aborting the process.  The bounce buffer is the only host memory the GPU ever touches on these paths.
// Synchronous by construction (one buffer): every call site synchronised right after the copy anyway.
constexpr size_t ODR_BOUNCE_BYTES = 8u << 20;
static inline int odr_i_bounce(odr_ctx *c, int which, void **buf) {
  if (!c->bounce[which]) HIPCHK(hipHostMalloc(&c->bounce[which], ODR_BOUNCE_BYTES, hipHostMallocDefault));
  *buf = c->bounce[which];
  return 0;
}
static inline int odr_i_h2d(odr_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t st, int which = 0) {
  void *b;
  int rc = odr_i_bounce(c, which, &b);
  if (rc) return rc;
  for (size_t o = 0; o < bytes; o += ODR_BOUNCE_BYTES) {
    const size_t m = std::min(ODR_BOUNCE_BYTES, bytes - o);
    memcpy(b, (const char *)src + o, m);
    HIPCHK(hipMemcpyAsync((char *)dst + o, b, m, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  return 0;
}
static inline int odr_i_d2h(odr_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t st, int which = 0) {
  void *b;
  int rc = odr_i_bounce(c, which, &b);
  if (rc) return rc;
  for (size_t o = 0; o < bytes; o += ODR_BOUNCE_BYTES) {
    const size_t m = std::min(ODR_BOUNCE_BYTES, bytes - o);
    HIPCHK(hipMemcpyAsync(b, (const char *)src + o, m, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    memcpy((char *)dst + o, b, m);
  }
  return 0;
}
#define H2D(dst, src, bytes) do { int rc_ = odr_i_h2d(c, (dst), (src), (bytes), c->stream); if (rc_) return rc_; } while (0)
#define D2H(dst, src, bytes) do { int rc_ = odr_i_d2h(c, (dst), (src), (bytes), c->stream); if (rc_) return rc_; } while (0)

static inline bool source_has_members(const DevSource &s) {
  for (int v = 0; v < NVAR; ++v) if (s.members[v] > 1) return true;
  return false;
}
static inline bool any_members(const odr_ctx *c) {
  for (int k = 0; k < c->nsrc; ++k) if (c->hw.src[k].kind == SRC_GRID && source_has_members(c->hw.src[k])) return true;
  return false;
}
// which instantiation of the kernels templated on the projection serves a reader: the PROJ_STERE_POLAR one assumes an
// ellipsoid (proj_fwd<true>, rotation_cs<true>: closed forms only); a spherical polar reader, Mercator, Lambert ... take the
// PROJ_STERE_EQUIT_SPHERE instantiation, whose projection functions switch on DevProj::kind at run time
static inline int odr_proj_template(const odr::DevProj &p) {
  if (p.kind == odr::PROJ_STERE_POLAR) return p.es != 0 ? odr::PROJ_STERE_POLAR : odr::PROJ_STERE_EQUIT_SPHERE;
  if (p.kind >= odr::PROJ_TMERC) return odr::PROJ_EXT;   // tmerc / laea / oblique stere / rotated pole: an instantiation of their own (round 5)
  return p.kind;
}
int odr_i_ensure_ranks(odr_ctx *c, odr_particles *p);

// defined in odrift.hip
bool odr_i_build_env_group(const odr_ctx *c, const int *grp, int ng, double t, EnvGroupDesc &G);
// odr_comm.hip: the process's RCCL communicators
bool odr_i_comm_on();
int odr_i_comm_rank();
int odr_i_comm_bcast_floats(float *dev, size_t count, int root, hipStream_t st);
void odr_i_phase_dump();   // odr_step.hip
bool odr_i_uv_fast_source(const odr_ctx *c, int &sid, double t_lo, double t_hi);

bool odr_i_gyre_source(const odr_ctx *c, int var, int &sid);
int odr_i_env_sample(odr_ctx *c, odr_particles *p, int nvars, const int32_t *var_ids, double t, float *const *out_host,
                     bool record_positions);
int odr_i_red_finish(odr_ctx *c, odr_particles *p);   // red[] <- the per-wave records of the step launch (k_red_init + k_red_finish)
int odr_i_red_records(odr_ctx *c, odr_particles *p, double **rec);   // device array for those records
int odr_i_reduce(odr_ctx *c, odr_particles *p, double wdd, int relwind, bool wind_args_matter = true, bool extents = true, bool partial_ok = false);
int odr_i_read_counter(odr_ctx *c, int64_t *out);
// k_env_noise with DEVICE arrays of draws (ODR_RNG_HOST) or none (ODR_RNG_DEVICE)
int odr_i_env_noise(odr_ctx *c, odr_particles *p, int vx, int vy, double std, int distribution, int rng_mode,
                    const double *dev_nx, const double *dev_ny, unsigned long long step);
#define build_env_group odr_i_build_env_group
#define uv_fast_source odr_i_uv_fast_source
#define gyre_source odr_i_gyre_source
#define env_sample_impl odr_i_env_sample
#define reduce odr_i_reduce
#define read_counter odr_i_read_counter
