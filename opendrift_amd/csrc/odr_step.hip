// libodrift_hip.so, translation unit 2: advect_ocean_current (Euler / RK2 / RK4) and the fused step
// (get_environment + coastline + previous state + advection in one launch).  See odrift.hip for the rest.
#define ODR_TU_STEP 1
#include "odr_host.h"

template <int SCHEME>
static void launch_advect_grid(odr_ctx *c, odr_particles *p, int sid, double t, double dt, double factor) {
  const DevSource &s = c->hw.src[sid];
  UVTime th = uv_time(s, t + dt / 2), tf = uv_time(s, t + dt);
  int geo = s.level_slot[0];
  bool is3d = s.slot[geo].var_nz[VAR_U] > 1;
  dim3 g(nblk(p->n)), b(BLOCK);
  PView v = view(p);
  float f = (float)factor;
#define ODR_LAUNCH(PROJ, D3) hipLaunchKernelGGL((k_advect_grid<SCHEME, PROJ, D3>), g, b, 0, c->stream, c->dw, sid, geo, v, dt, f, th, tf)
  switch (s.proj.kind) {
    case PROJ_LATLONG: if (is3d) ODR_LAUNCH(PROJ_LATLONG, true); else ODR_LAUNCH(PROJ_LATLONG, false); break;
    case PROJ_STERE_POLAR: if (is3d) ODR_LAUNCH(PROJ_STERE_POLAR, true); else ODR_LAUNCH(PROJ_STERE_POLAR, false); break;
    case PROJ_CURVILINEAR: if (is3d) ODR_LAUNCH(PROJ_CURVILINEAR, true); else ODR_LAUNCH(PROJ_CURVILINEAR, false); break;
    default: if (is3d) ODR_LAUNCH(PROJ_STERE_EQUIT_SPHERE, true); else ODR_LAUNCH(PROJ_STERE_EQUIT_SPHERE, false); break;
  }
#undef ODR_LAUNCH
}

int odr_advect(odr_ctx *c, odr_particles *p, int scheme, double t, double dt, double factor) {
  REQUIRE(scheme >= 0 && scheme <= 2, "Drift scheme not recognised: %d", scheme);
  if (!p->env[VAR_U] || !p->env[VAR_V]) return fail(ODR_ERR_STATE, "odr_env_sample of the current must precede odr_advect");
  HIPCHK(hipSetDevice(c->device));
  int rc = flush_world(c);
  if (rc) return rc;
  if (p->n == 0) return 0;
  dim3 g(nblk(p->n)), b(BLOCK);
  PView v = view(p);
  int sid = -1, gsid = -1;
  if (scheme == 0) hipLaunchKernelGGL(k_advect<0>, g, b, 0, c->stream, c->dw, v, t, dt, (float)factor);
  else if (!getenv("ODR_NO_FAST_PATH") && gyre_source(c, VAR_U, gsid) && c->hw.nlist[VAR_V] == 1 &&
           c->hw.list[VAR_V][0] == gsid &&
           (c->hw.src[gsid].always_valid || (fmin(t, t + dt) >= c->hw.src[gsid].tmin && fmax(t, t + dt) <= c->hw.src[gsid].tmax))) {
    const DevSource &gs = c->hw.src[gsid];
    const double sh = sin(gs.params[2] * (t + dt / 2 - gs.params[3])), sf = sin(gs.params[2] * (t + dt - gs.params[3]));
    if (scheme == 1) hipLaunchKernelGGL(k_advect_gyre<1>, g, b, 0, c->stream, c->dw, gsid, v, dt, (float)factor, sh, sf);
    else hipLaunchKernelGGL(k_advect_gyre<2>, g, b, 0, c->stream, c->dw, gsid, v, dt, (float)factor, sh, sf);
  }
  else if (uv_fast_source(c, sid, t < t + dt ? t : t + dt, t < t + dt ? t + dt : t) && !getenv("ODR_NO_FAST_PATH")) {
    if (scheme == 1) launch_advect_grid<1>(c, p, sid, t, dt, factor);
    else launch_advect_grid<2>(c, p, sid, t, dt, factor);
  } else if (scheme == 1) hipLaunchKernelGGL(k_advect<1>, g, b, 0, c->stream, c->dw, v, t, dt, (float)factor);
  else hipLaunchKernelGGL(k_advect<2>, g, b, 0, c->stream, c->dw, v, t, dt, (float)factor);
  HIPCHK(hipGetLastError());
  return 0;
}

// get_environment -> interact_with_coastline -> update_previous_state -> advect_ocean_current in
// one launch (k_step_grid) when the current comes from one gridded reader; otherwise exactly the
// four separate entry points, in that order.  Results are bit-identical either way
// (tests/test_gpu_parity.py::test_fused_step_equals_separate_calls).
template <int SCHEME>
static void launch_step_grid(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G, StepDesc S, double t, double dt,
                             double factor) {
  const DevSource &s = c->hw.src[G.sid];
  UVTime th = uv_time(s, t + dt / 2), tf = uv_time(s, t + dt);
  S.geo_slot_uv = s.level_slot[0];
  bool is3d = s.slot[S.geo_slot_uv].var_nz[VAR_U] > 1;
  dim3 g(nblk(p->n)), b(BLOCK);
  PView v = view(p);
  float f = (float)factor;
#define ODR_LAUNCH(PROJ, D3) hipLaunchKernelGGL((k_step_grid<SCHEME, PROJ, D3>), g, b, 0, c->stream, c->dw, v, G, S, dt, f, th, tf, c->counter)
  switch (s.proj.kind) {
    case PROJ_LATLONG: if (is3d) ODR_LAUNCH(PROJ_LATLONG, true); else ODR_LAUNCH(PROJ_LATLONG, false); break;
    case PROJ_STERE_POLAR: if (is3d) ODR_LAUNCH(PROJ_STERE_POLAR, true); else ODR_LAUNCH(PROJ_STERE_POLAR, false); break;
    case PROJ_CURVILINEAR: if (is3d) ODR_LAUNCH(PROJ_CURVILINEAR, true); else ODR_LAUNCH(PROJ_CURVILINEAR, false); break;
    default: if (is3d) ODR_LAUNCH(PROJ_STERE_EQUIT_SPHERE, true); else ODR_LAUNCH(PROJ_STERE_EQUIT_SPHERE, false); break;
  }
#undef ODR_LAUNCH
}

int odr_env_coast_advect(odr_ctx *c, odr_particles *p, int nvars, const int32_t *var_ids, double t,
                         int coast_action, int stranded_code, int seeded_on_land_code, int store_previous,
                         int scheme, double dt, double factor, const odr_step_extras *extras, int64_t *n_on_land) {
  p->epoch++;  // invalidates the cached reductions (reduce())
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
  if ((rc = flush_world(c))) return rc;
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
  if (want_floor && !has_depth && !p->env[VAR_DEPTH]) return fail(ODR_ERR_STATE, "sea_floor_depth_below_sea_level has not been sampled");
  const bool depth_in_group = want_floor && has_depth && same_list(VAR_DEPTH, VAR_U);
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
  bool fuse = p->n > 0 && !getenv("ODR_NO_FAST_PATH") && same_list(VAR_V, VAR_U) && ng <= MAXG && nmg <= 4 && nmr <= 4 &&
              uv_fast_source(c, sid, t < t + dt ? t : t + dt, t < t + dt ? t + dt : t) &&
              build_env_group(c, grp, ng, t, G) && G.sid == sid;
  if (!fuse) {
    if ((rc = odr_env_sample(c, p, nvars, var_ids, t, nullptr))) return rc;
    if (extras && extras->missing_code && (rc = odr_deactivate_missing(c, p, nvars, var_ids, extras->missing_code, nullptr)))
      return rc;
    if ((rc = odr_coastline(c, p, coast_action, stranded_code, seeded_on_land_code, n_on_land))) return rc;
    if (want_floor && (rc = odr_seafloor(c, p, nullptr))) return rc;
    if (extras && extras->age_dt != 0 &&
        (rc = odr_increase_age(c, p, extras->age_dt, extras->max_age_seconds, extras->retired_code)))
      return rc;
    if (store_previous && (rc = odr_store_previous(c, p))) return rc;
    return odr_advect(c, p, scheme, t, dt, factor);
  }
  for (int k = 0; k < ng; ++k) if ((rc = ensure_env(c, p, grp[k]))) return rc;
  if (nrest && (rc = env_sample_impl(c, p, nrest, rest, t, nullptr, false))) return rc;  // k_step_grid records the positions
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
  if (coast_action) HIPCHK(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream));
  if (scheme == 0) launch_step_grid<0>(c, p, G, S, t, dt, factor);
  else if (scheme == 1) launch_step_grid<1>(c, p, G, S, t, dt, factor);
  else launch_step_grid<2>(c, p, G, S, t, dt, factor);
  HIPCHK(hipGetLastError());
  return coast_action ? read_counter(c, n_on_land) : 0;
}
