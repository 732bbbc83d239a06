// libodrift_hip.so, translation unit 2: advect_ocean_current (Euler / RK2 / RK4) and the fused step
// (get_environment + coastline + previous state + advection in one launch).  See odrift.hip for the rest.
#define ODR_TU_STEP 1
#include <chrono>
#include "odr_step_launch.h"

// drift:current_uncertainty / drift:current_uncertainty_uniform are part of every get_environment call that holds the
// current (environment.py:869-886) -- also of the Runge-Kutta stage calls inside advect_ocean_current
// (physics_methods.py:638-670).  odr_advect_set_noise arms them for the NEXT odr_advect / odr_env_coast_advect on `p`
// (call it right before: host arrays are indexed by the element order of that moment).
int odr_advect_set_noise(odr_ctx *c, odr_particles *p, double std_normal, double std_uniform, int rng_mode,
                         const double *host_main, const double *host_stage, int nstage, uint64_t step) {
  c->noise_owner = nullptr;
  REQUIRE(std_normal >= 0 && std_uniform >= 0, "uncertainties must not be negative");
  REQUIRE(rng_mode == ODR_RNG_DEVICE || rng_mode == ODR_RNG_HOST, "bad rng mode");
  if (!(std_normal > 0) && !(std_uniform > 0)) return 0;
  StageNoise &N = c->noise;
  memset(&N, 0, sizeof N);
  N.on = 1; N.rng_mode = rng_mode; N.std_n = std_normal; N.std_u = std_uniform;
  N.ncomp = 2 * ((std_normal > 0) + (std_uniform > 0));
  N.seed = c->seed; N.step = (unsigned long long)step;
  if (rng_mode == ODR_RNG_HOST && p->n > 0) {
    REQUIRE(nstage >= 0 && nstage <= 3 && (nstage == 0 || host_stage), "host draws of the stage calls required in ODR_RNG_HOST mode");
    const size_t per = (size_t)N.ncomp * (size_t)p->n, need = per * (size_t)(nstage + (host_main ? 1 : 0));
    if (c->noise_buf_n < need) {
      HIPCHK(hipStreamSynchronize(c->stream));
      if (c->noise_buf) HIPCHK(hipFree(c->noise_buf));
      HIPCHK(hipMalloc((void **)&c->noise_buf, sizeof(double) * need));
      c->noise_buf_n = need;
    }
    double *d = c->noise_buf;
    if (host_main) {
      H2D(d, host_main, sizeof(double) * per);
      N.main = d;
      d += per;
    }
    if (nstage) {
      H2D(d, host_stage, sizeof(double) * per * (size_t)nstage);
      N.stage = d;
    }
    HIPCHK(hipStreamSynchronize(c->stream));   // the host arrays are pageable and may change right after
  }
  c->noise_owner = p;
  return 0;
}

ODR_DEFINE_PHASE_DUMP(odr_i_phase_dump_exact, "exact stage math")
void odr_i_phase_dump() { odr_i_phase_dump_exact(); odr_i_phase_dump_fast(); }

int odr_ctx_set_stage_math(odr_ctx *c, int mode) {
  REQUIRE(c, "null context");
  REQUIRE(mode == ODR_STAGE_EXACT || mode == ODR_STAGE_FAST, "stage math must be ODR_STAGE_EXACT (0) or ODR_STAGE_FAST (1)");
  c->stage_math = mode;
  return 0;
}

// the armed uncertainty of `p`, disarmed
static StageNoise take_noise(odr_ctx *c, const odr_particles *p) {
  StageNoise N;
  memset(&N, 0, sizeof N);
  if (c->noise_owner == p) N = c->noise;
  c->noise_owner = nullptr;
  N.sm = c->stage_math;
  return N;
}

static int advect_impl(odr_ctx *c, odr_particles *p, int scheme, double t, double dt, double factor, const StageNoise &N) {
  REQUIRE(scheme >= 0 && scheme <= 2, "Drift scheme not recognised: %d", scheme);
  if (!p->env[VAR_U] || !p->env[VAR_V]) return fail(ODR_ERR_STATE, "odr_env_sample of the current must precede odr_advect");
  HIPCHK(hipSetDevice(c->device));
  int rc = flush_world(c);
  if (rc) return rc;
  if (p->n == 0) return 0;
  if (scheme > 0 && (rc = odr_i_ensure_ranks(c, p))) return rc;   // stage calls on ensemble data: the present elements' ranks
  if (N.on && scheme > 0) {
    if (N.rng_mode == ODR_RNG_HOST && !N.stage) return fail(ODR_ERR_INVALID, "no host draws for the Runge-Kutta stage calls");
    if (!(N.sm == ODR_STAGE_FAST && odr_i_advect_fast_noise(c, p, scheme, t, dt, factor, N))) odr_i_advect_noise(c, p, scheme, t, dt, factor, N);
  } else if (!(N.sm == ODR_STAGE_FAST && scheme > 0 && odr_i_advect_fast(c, p, scheme, t, dt, factor, N)))
    advect_dispatch<false>(c, p, scheme, t, dt, factor, N);
  HIPCHK(hipGetLastError());
  return 0;
}

int odr_advect(odr_ctx *c, odr_particles *p, int scheme, double t, double dt, double factor) {
  const StageNoise N = take_noise(c, p);
  return advect_impl(c, p, scheme, t, dt, factor, N);
}

// get_environment -> interact_with_coastline -> update_previous_state -> advect_ocean_current in
// one launch (k_step_grid) when the current comes from one gridded reader; otherwise exactly the
// four separate entry points, in that order.  Results are bit-identical either way
// (tests/test_gpu_parity.py::test_fused_step_equals_separate_calls).
// extras->vmix when the mixing cannot run inside the step launch: the two calls in sequence
static int mix_after(odr_ctx *c, odr_particles *p, double t, double dt, const odr_step_extras *ex) {
  c->fuse_vadv = ex->vmix_vadv;
  return odr_vmix(c, p, t, dt, ex->vmix_dt_mix, ex->vmix_at_surface, ODR_RNG_DEVICE, nullptr, ex->vmix_step);
}
static bool s_is_latlong_3d(const odr_ctx *c, int sid) {
  const DevSource &s = c->hw.src[sid];
  return s.proj.kind == PROJ_LATLONG && s.nz > 1 && s.slot[s.level_slot[0]].var_nz[VAR_U] > 1;
}

// ---- lanes: the step kernel (gathers: bound by memory latency, VALU about half busy) and the mixing kernel (bound by
// VALU issue) of DIFFERENT particles have nothing to wait for in each other.  The particle range is cut into `lanes`
// contiguous windows, each on a stream of its own: step(l) -> mix(l) in order on lane l, and step(l + 1) starts when
// step(l) has drained, so that mix(l) and step(l + 1) are resident together and fill each other's idle issue slots.
// Results are those of the single launch (kernels are per-particle; the RNG is keyed by element ID).
static int lane_count(odr_ctx *c, const odr_particles *p, bool want_mix, bool noise_on) {
  if (!want_mix || noise_on || c->oil_owner == p) return 1;
  const char *e = getenv("ODR_LANES");
  int lanes = e ? atoi(e) : 1;
  if (lanes > ODR_MAX_LANES) lanes = ODR_MAX_LANES;
  const char *m = getenv("ODR_LANES_MIN_N");
  const long long min_n = m ? atoll(m) : 2000000;
  if (p->n < min_n || p->n / (lanes > 0 ? lanes : 1) < 4 * BLOCK) return 1;
  return lanes;
}

static int step_in_lanes(odr_ctx *c, odr_particles *p, int lanes, const EnvGroupDesc &G, const StepDesc &S, int scheme,
                         double t, double dt, double factor, const StageNoise &N, const odr_step_extras *ex) {
  if (!c->lanes_ready) {
    for (int l = 0; l < ODR_MAX_LANES; ++l) {
      HIPCHK(hipStreamCreateWithFlags(&c->lane_stream[l], hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&c->lane_step[l], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&c->lane_done[l], hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&c->lane_fork, hipEventDisableTiming));
    c->lanes_ready = 1;
  }
  const long long n = p->n;
  const hipStream_t main_stream = c->stream;
  long long per = (n / lanes + BLOCK - 1) / BLOCK * BLOCK;
  HIPCHK(hipEventRecord(c->lane_fork, main_stream));
  int rc = 0, used = 0;
  for (int l = 0; l < lanes && !rc; ++l) {
    const long long first = (long long)l * per, count = l == lanes - 1 ? n - first : std::min(per, n - first);
    if (count <= 0) break;
    hipStream_t s = c->lane_stream[l];
    HIPCHK(hipStreamWaitEvent(s, c->lane_fork, 0));
    if (l > 0) HIPCHK(hipStreamWaitEvent(s, c->lane_step[l - 1], 0));
    p->win = first; p->n = count; c->stream = s;
    step_dispatch<false>(c, p, G, S, scheme, t, dt, factor, N);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(c->lane_step[l], s);
    if (e == hipSuccess) rc = mix_after(c, p, t, dt, ex);
    if (e == hipSuccess && !rc) e = hipEventRecord(c->lane_done[l], s);
    p->win = 0; p->n = n; c->stream = main_stream;
    if (e != hipSuccess) return fail(ODR_ERR_HIP, "lane launch: %s", hipGetErrorString(e));
    used = l + 1;
  }
  for (int l = 0; l < used; ++l) HIPCHK(hipStreamWaitEvent(main_stream, c->lane_done[l], 0));
  return rc;
}

int odr_env_coast_advect(odr_ctx *c, odr_particles *p, int nvars, const int32_t *var_ids, double t,
                         int coast_action, int stranded_code, int seeded_on_land_code, int store_previous,
                         int scheme, double dt, double factor, const odr_step_extras *extras, int64_t *n_on_land) {
  p->status_epoch++;
  p->epoch++;  // invalidates the cached reductions (reduce())
  const StageNoise N = take_noise(c, p);
  const bool main_noise = N.on && extras && extras->main_noise;
  REQUIRE(nvars > 0 && nvars <= NVAR && var_ids, "bad variable list");
  REQUIRE(scheme >= 0 && scheme <= 2, "Drift scheme not recognised: %d", scheme);
  REQUIRE(coast_action >= 0 && coast_action <= 2, "bad coastline action");
  HIPCHK(hipSetDevice(c->device));
  if (n_on_land) *n_on_land = 0;
  int rc;
  bool has_u = false, has_v = false, has_land = false;
  for (int k = 0; k < nvars; ++k) {
    REQUIRE(var_ids[k] >= 0 && var_ids[k] < NVAR, "bad variable id %d", var_ids[k]);
    has_u |= var_ids[k] == VAR_U; has_v |= var_ids[k] == VAR_V; has_land |= var_ids[k] == VAR_LAND;
  }
  REQUIRE(has_u && has_v, "the variable list must hold x/y_sea_water_velocity");
  if (coast_action && !has_land && !p->env[VAR_LAND]) return fail(ODR_ERR_STATE, "land_binary_mask has not been sampled");
  // ODR_SLOW_CALLS=<ms>: a call that keeps the host longer than that says where (stderr)
  static const double slow_ms = getenv("ODR_SLOW_CALLS") ? atof(getenv("ODR_SLOW_CALLS")) : 0.0;
  const auto t_call = std::chrono::steady_clock::now();
  double t_mark[4] = {0, 0, 0, 0};
  auto mark = [&](int k) { if (slow_ms > 0) t_mark[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };
  struct SlowReport {
    const double &lim; const double *m;
    ~SlowReport() { if (lim > 0 && (m[3] > lim || m[2] > lim)) fprintf(stderr, "odr_env_coast_advect: world flushed at %.3f ms, group built at %.3f, other variables sampled at %.3f, launched at %.3f\n", m[0], m[1], m[2], m[3]); }
  } slow_report{slow_ms, t_mark};
  if ((rc = flush_world(c))) return rc;
  mark(0);
  // the group of the current: variables of the list that share its priority list
  int grp[NVAR], ng = 0, rest[NVAR], nrest = 0;
  auto same_list = [&](int a, int b) {
    if (c->hw.nlist[a] != c->hw.nlist[b]) return false;
    for (int k = 0; k < c->hw.nlist[a]; ++k) if (c->hw.list[a][k] != c->hw.list[b][k]) return false;
    return true;
  };
  bool land_in_group = has_land && same_list(VAR_LAND, VAR_U);
  bool has_depth = false;
  for (int k = 0; k < nvars; ++k) has_depth |= var_ids[k] == VAR_DEPTH;
  const bool want_floor = extras && extras->seafloor_action == 1;
  const bool want_mix = extras && extras->vmix;
  if (want_mix) REQUIRE(extras->vmix_dt_mix > 0 && dt != 0, "bad mixing time step");
  if (want_floor && !has_depth && !p->env[VAR_DEPTH]) return fail(ODR_ERR_STATE, "sea_floor_depth_below_sea_level has not been sampled");
  const bool depth_in_group = (want_floor || want_mix) && has_depth && same_list(VAR_DEPTH, VAR_U);
  grp[ng++] = VAR_U; grp[ng++] = VAR_V;
  if (land_in_group) grp[ng++] = VAR_LAND;
  const int depth_slot = depth_in_group ? ng : -1;
  if (depth_in_group) grp[ng++] = VAR_DEPTH;
  bool seen[NVAR] = {false};
  seen[VAR_U] = seen[VAR_V] = true;
  if (land_in_group) seen[VAR_LAND] = true;
  if (depth_in_group) seen[VAR_DEPTH] = true;
  for (int k = 0; k < nvars; ++k) {
    int v = var_ids[k];
    if (seen[v]) continue;
    seen[v] = true;
    if (same_list(v, VAR_U)) grp[ng++] = v; else rest[nrest++] = v;
  }
  // report_missing_variables inside the fused kernel: the variables that can still be NaN (no fallback)
  int miss_grp[NVAR], nmg = 0, miss_rest[NVAR], nmr = 0;
  if (extras && extras->missing_code) {
    for (int k = 0; k < ng; ++k) if (std::isnan(c->hw.fallback[grp[k]])) miss_grp[nmg++] = k;
    for (int k = 0; k < nrest; ++k) if (std::isnan(c->hw.fallback[rest[k]])) miss_rest[nmr++] = rest[k];
  }
  EnvGroupDesc G;
  int sid = -1;
  // (odr_ctx_set_position_class: the float32 element arrays of a run's first get_environment take the separate launches -- the
  // sample kernels know the float32 longitude modulation, the step launch is kept free of it)
  bool fuse = p->n > 0 && !c->hw.f32pos && !getenv("ODR_NO_FAST_PATH") && same_list(VAR_V, VAR_U) && ng <= MAXG && nmg <= 4 && nmr <= 4 &&
              uv_fast_source(c, sid, t < t + dt ? t : t + dt, t < t + dt ? t + dt : t) &&
              build_env_group(c, grp, ng, t, G) && G.sid == sid && G.burst;   // k_step_grid carries the burst sampler only
  mark(1);
  if (main_noise && N.rng_mode == ODR_RNG_HOST && !N.main) return fail(ODR_ERR_INVALID, "no host draws for the main sample");
  if (!fuse) {
    if ((rc = odr_env_sample(c, p, nvars, var_ids, t, nullptr))) return rc;
    if (main_noise) {   // environment.py:869-886 of the main-loop call: normal pair, then uniform pair
      const size_t n = (size_t)p->n;
      if (N.std_n > 0 && (rc = odr_i_env_noise(c, p, VAR_U, VAR_V, N.std_n, ODR_NOISE_NORMAL, N.rng_mode, N.main, N.main ? N.main + n : nullptr, N.step)))
        return rc;
      const double *um = N.main ? N.main + (N.std_n > 0 ? 2 * n : 0) : nullptr;
      if (N.std_u > 0 && (rc = odr_i_env_noise(c, p, VAR_U, VAR_V, N.std_u, ODR_NOISE_UNIFORM, N.rng_mode, um, um ? um + n : nullptr, N.step)))
        return rc;
    }
    if (extras && extras->missing_code && (rc = odr_deactivate_missing(c, p, nvars, var_ids, extras->missing_code, nullptr)))
      return rc;
    if ((rc = odr_coastline(c, p, coast_action, stranded_code, seeded_on_land_code, n_on_land))) return rc;
    if (want_floor && (rc = odr_seafloor(c, p, nullptr))) return rc;
    if (extras && extras->age_dt != 0 &&
        (rc = odr_increase_age(c, p, extras->age_dt, extras->max_age_seconds, extras->retired_code)))
      return rc;
    if (store_previous && (rc = odr_store_previous(c, p))) return rc;
    if ((rc = advect_impl(c, p, scheme, t, dt, factor, N))) return rc;
    return want_mix ? mix_after(c, p, t, dt, extras) : 0;
  }
  p->profiles_f32 = false;     // (this launch records the sample positions, and never runs in the float32 position class)
  for (int k = 0; k < ng; ++k) { if ((rc = ensure_env(c, p, grp[k]))) return rc; p->env_cok[grp[k]] = false; }
  if (coast_action == 2) p->env_cok[VAR_LAND] = false;   // elements on land get land_binary_mask = 0
  if (nrest && (rc = env_sample_impl(c, p, nrest, rest, t, nullptr, false))) return rc;  // k_step_grid records the positions
  mark(2);
  StepDesc S;
  memset(&S, 0, sizeof S);
  S.coast_action = coast_action; S.stranded_code = stranded_code; S.seeded_code = seeded_on_land_code;
  S.land_slot = land_in_group ? 2 : -1;
  S.store_previous = store_previous;
  S.seafloor = want_floor ? 1 : 0;
  S.depth_slot = depth_slot;
  S.age_dt = extras ? (float)extras->age_dt : 0.0f;
  S.max_age = extras ? (float)extras->max_age_seconds : 0.0f;
  S.retired_code = extras ? extras->retired_code : 0;
  S.missing_code = extras ? extras->missing_code : 0;
  S.nmiss_grp = nmg; S.nmiss_rest = nmr;
  for (int k = 0; k < nmg && k < 4; ++k) S.miss_grp[k] = miss_grp[k];
  for (int k = 0; k < nmr && k < 4; ++k) S.miss_rest[k] = miss_rest[k];
  if (want_floor && (rc = ensure_env(c, p, VAR_SSH))) return rc;
  // counter[0]: elements on land; [1], [2]: the status scan formed by the launch (StepDesc.wcount, below) -- one fill for the three
  const bool count_in_launch = p->wcount && !p->external && !getenv("ODR_NO_STEP_COUNT");
  if (coast_action || count_in_launch)
    HIPCHK(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long) * (count_in_launch ? 4 : 1), c->stream));   // (32 bytes: one fill; 24 were two)
  S.main_noise = main_noise ? 1 : 0;
  S.ssh_slot = -1;
  for (int k = 0; k < G.nv; ++k) if (G.var[k] == VAR_SSH) S.ssh_slot = k;
  // vertical mixing inside the launch: K from the reader of the current (lon/lat, 3-D, <= 16 levels), device RNG
  StepMix M;
  memset(&M, 0, sizeof M);
  bool mix_fused = false;
  // Measured on C3 (profiles/r02_ab_variants.txt): the merged kernel needs 129 VGPRs (3 waves per SIMD instead of 4) and is
  // SLOWER than the two launches (1.88 vs 1.75 ms per step; held at 128 VGPRs: 2.48 ms) -- both kernels are bound by
  // instruction issue, there is no idle time for the merger to fill.  It stays behind ODR_FUSED_MIX=1 with its
  // bit-identity test (tests/test_gpu_fused_step.py); by default extras->vmix makes the two launches back to back.
  if (want_mix && !N.on && getenv("ODR_FUSED_MIX") && !getenv("ODR_NO_FUSED_MIX") && s_is_latlong_3d(c, G.sid) && c->oil_owner != p &&
      build_vmix_desc(c, t, M.D) && M.D.sid == G.sid && M.D.nzp <= 16) {
    if ((rc = ensure_env(c, p, VAR_SSH))) return rc;
    M.A.dt = dt; M.A.dt_mix_cfg = extras->vmix_dt_mix; M.A.mix_at_surface = extras->vmix_at_surface;
    M.A.rng_mode = ODR_RNG_DEVICE; M.A.sfl = c->seafloor; M.A.huni = nullptr; M.A.seed = c->seed;
    M.A.step = (unsigned long long)extras->vmix_step;
    M.vadv = extras->vmix_vadv;
    M.w_slot = -1;
    for (int k = 0; k < G.nv; ++k) if (G.var[k] == VAR_W) M.w_slot = k;
    if (M.vadv >= 0 && M.w_slot < 0 && !p->env[VAR_W]) return fail(ODR_ERR_STATE, "upward_sea_water_velocity has not been sampled");
    mix_fused = true;
  }
  if (mix_fused) {
    odr_i_step_mix(c, p, G, S, scheme, t, dt, factor, M);
    HIPCHK(hipGetLastError());
    return coast_action ? read_counter(c, n_on_land) : 0;
  }
  const int lanes = lane_count(c, p, want_mix, N.on);
  if (lanes > 1) {
    if ((rc = ensure_env(c, p, VAR_SSH))) return rc;
    if ((rc = step_in_lanes(c, p, lanes, G, S, scheme, t, dt, factor, N, extras))) return rc;
    return coast_action ? read_counter(c, n_on_land) : 0;
  }
  if (odr_i_step_tile(c, p, G, S, scheme, t, dt, factor, N)) {   // the records of the workgroup's node rectangle in LDS (odr_tile.hip.h)
    HIPCHK(hipGetLastError());
    if ((rc = coast_action ? read_counter(c, n_on_land) : 0)) return rc;
    return want_mix ? mix_after(c, p, t, dt, extras) : 0;
  }
  // the movers' tests (no element at the surface, wind / Stokes drift / diffusivity identically zero) formed by this launch
  // (2-D readers: the 3-D instantiations of the kernel carry no code for it -- a 3-D run mixes vertically before its movers)
  bool red_in_launch = c->step_reduce_on && !p->external && !(c->red_pinned && c->red_owner == p) && !getenv("ODR_NO_STEP_REDUCE") &&
                       !(c->hw.src[G.sid].slot[c->hw.src[G.sid].level_slot[0]].var_nz[VAR_U] > 1);
  if (red_in_launch) {
    auto slot_of = [&](int var) { for (int k = 0; k < G.nv; ++k) if (G.var[k] == var) return k; return p->env[var] ? -1 : -2; };
    auto pair_of = [&](int vx, int vy, int &out) {   // both in the group next to each other, both from arrays, or both absent
      const int a = slot_of(vx), b = slot_of(vy);
      if (a >= 0 && b == a + 1) { out = a; return true; }
      if (a == -1 && b == -1) { out = -1; return true; }
      if (a == -2 || b == -2) { out = -2; return a < 0 && b < 0; }
      return false;
    };
    S.red_hd = slot_of(VAR_HDIFF);
    red_in_launch = pair_of(VAR_SX, VAR_SY, S.red_sx) && pair_of(VAR_XWIND, VAR_YWIND, S.red_xw) && (S.red_xw == -2 || p->f32[0]);
    if (red_in_launch) {
      S.red_on = 1; S.red_rel = c->step_reduce_rel; S.red_wdd = c->step_reduce_wdd;
      S.red_iwdd = c->step_reduce_wdd != 0 ? 1.0 / fabs(c->step_reduce_wdd) : 0.0;
      if ((rc = odr_i_red_records(c, p, &S.red))) return rc;
    }
  }
  if (count_in_launch) { S.wcount = p->wcount; S.sflags = c->counter + 2; }   // (k_step_grid only: the tile / lane / merged paths left above)
  if (N.on && (scheme > 0 || main_noise)) {
    if (scheme > 0 && N.rng_mode == ODR_RNG_HOST && !N.stage) return fail(ODR_ERR_INVALID, "no host draws for the Runge-Kutta stage calls");
    if (N.sm == ODR_STAGE_FAST && scheme > 0) odr_i_step_fast_noise(c, p, G, S, scheme, t, dt, factor, N);
    else odr_i_step_noise(c, p, G, S, scheme, t, dt, factor, N);
  } else if (N.sm == ODR_STAGE_FAST && scheme > 0) odr_i_step_fast(c, p, G, S, scheme, t, dt, factor, N);
  else step_dispatch<false>(c, p, G, S, scheme, t, dt, factor, N);
  HIPCHK(hipGetLastError());
  if (red_in_launch) {   // the cache describes what the launch left in red[] (a mixing launch that follows invalidates it: z changes)
    if ((rc = odr_i_red_finish(c, p))) return rc;
    c->red_owner = p; c->red_epoch = p->epoch; c->red_wdd = c->step_reduce_wdd; c->red_rel = c->step_reduce_rel;
    c->red_extents = false; c->red_partial = true;
  }
  mark(3);
  if (count_in_launch) { p->wcount_epoch = p->status_epoch; p->wcount_n = p->n; }   // (a mixing launch that follows voids it: it can deactivate)
  if ((rc = coast_action ? read_counter(c, n_on_land) : 0)) return rc;
  return want_mix ? mix_after(c, p, t, dt, extras) : 0;
}
