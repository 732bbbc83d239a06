// Launchers of the current-advection kernels, templated on NOISE (drift:current_uncertainty inside the Runge-Kutta
// stage calls).  Included by odr_step.hip (NOISE = false) and odr_step_noise.hip (NOISE = true): the two sets of
// instantiations are separate translation units so that they compile in parallel.
#pragma once
#include "odr_host.h"

template <int SCHEME, bool NOISE, int SM = 0>
static void launch_advect_grid(odr_ctx *c, odr_particles *p, int sid, double t, double dt, double factor, const StageNoise &N) {
  const DevSource &s = c->hw.src[sid];
  UVTime th = uv_time(s, t + dt / 2), tf = uv_time(s, t + dt);
  int geo = s.level_slot[0];
  bool is3d = s.slot[geo].var_nz[VAR_U] > 1;
  dim3 g(nblk(p->n)), b(BLOCK);
  PView v = view(p);
  float f = (float)factor;
#define ODR_LAUNCH(PROJ, D3) hipLaunchKernelGGL((k_advect_grid<SCHEME, PROJ, D3, NOISE, SM>), g, b, 0, c->stream, c->dw, sid, geo, v, dt, f, th, tf, N)
  switch (odr_proj_template(s.proj)) {
    case PROJ_LATLONG: if (is3d) ODR_LAUNCH(PROJ_LATLONG, true); else ODR_LAUNCH(PROJ_LATLONG, false); break;
    case PROJ_STERE_POLAR: if (is3d) ODR_LAUNCH(PROJ_STERE_POLAR, true); else ODR_LAUNCH(PROJ_STERE_POLAR, false); break;
    case PROJ_CURVILINEAR: if (is3d) ODR_LAUNCH(PROJ_CURVILINEAR, true); else ODR_LAUNCH(PROJ_CURVILINEAR, false); break;
    case PROJ_EXT: if (is3d) ODR_LAUNCH(PROJ_EXT, true); else ODR_LAUNCH(PROJ_EXT, false); break;
    default: if (is3d) ODR_LAUNCH(PROJ_STERE_EQUIT_SPHERE, true); else ODR_LAUNCH(PROJ_STERE_EQUIT_SPHERE, false); break;
  }
#undef ODR_LAUNCH
}

// the kernel choice of odr_advect: Euler | analytic double gyre | one gridded reader | any reader mix.
// SM = 1 (ODR_STAGE_FAST) instantiates the gridded-reader kernels only and returns false for everything else: the caller
// then takes the SM = 0 dispatch, whose other kernels read the mode from StageNoise::sm.
template <bool NOISE, int SM = 0>
static bool advect_dispatch(odr_ctx *c, odr_particles *p, int scheme, double t, double dt, double factor, const StageNoise &N) {
  dim3 g(nblk(p->n)), b(BLOCK);
  PView v = view(p);
  int sid = -1, gsid = -1;
  const bool gyre = !getenv("ODR_NO_FAST_PATH") && scheme > 0 && gyre_source(c, VAR_U, gsid) && c->hw.nlist[VAR_V] == 1 &&
                    c->hw.list[VAR_V][0] == gsid &&
                    (c->hw.src[gsid].always_valid || (fmin(t, t + dt) >= c->hw.src[gsid].tmin && fmax(t, t + dt) <= c->hw.src[gsid].tmax));
  const bool grid = scheme > 0 && !gyre && uv_fast_source(c, sid, t < t + dt ? t : t + dt, t < t + dt ? t + dt : t) && !getenv("ODR_NO_FAST_PATH");
  if constexpr (SM != 0) {
    if (!grid) return false;
    if (scheme == 1) launch_advect_grid<1, NOISE, SM>(c, p, sid, t, dt, factor, N);
    else launch_advect_grid<2, NOISE, SM>(c, p, sid, t, dt, factor, N);
    return true;
  } else {
    if (scheme == 0) hipLaunchKernelGGL((k_advect<0, false>), g, b, 0, c->stream, c->dw, v, t, dt, (float)factor, N);
    else if (gyre) {
      const DevSource &gs = c->hw.src[gsid];
      const double sh = sin(gs.params[2] * (t + dt / 2 - gs.params[3])), sf = sin(gs.params[2] * (t + dt - gs.params[3]));
      if (scheme == 1) hipLaunchKernelGGL((k_advect_gyre<1, NOISE>), g, b, 0, c->stream, c->dw, gsid, v, dt, (float)factor, sh, sf, N);
      else hipLaunchKernelGGL((k_advect_gyre<2, NOISE>), g, b, 0, c->stream, c->dw, gsid, v, dt, (float)factor, sh, sf, N);
    } else if (grid) {
      if (scheme == 1) launch_advect_grid<1, NOISE, 0>(c, p, sid, t, dt, factor, N);
      else launch_advect_grid<2, NOISE, 0>(c, p, sid, t, dt, factor, N);
    } else if (scheme == 1) hipLaunchKernelGGL((k_advect<1, NOISE>), g, b, 0, c->stream, c->dw, v, t, dt, (float)factor, N);
    else hipLaunchKernelGGL((k_advect<2, NOISE>), g, b, 0, c->stream, c->dw, v, t, dt, (float)factor, N);
    return true;
  }
}

// get_environment -> interact_with_coastline -> update_previous_state -> advect_ocean_current in
// one launch (k_step_grid) when the current comes from one gridded reader; otherwise exactly the
// four separate entry points, in that order.  Results are bit-identical either way
// (tests/test_gpu_parity.py::test_fused_step_equals_separate_calls).
template <int SCHEME, bool NOISE, int SM = 0>
static void launch_step_grid(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G0, StepDesc S, double t, double dt,
                             double factor, const StageNoise &N) {
  const EnvGroupDesc G = env_bind_out(G0, view(p));
  const DevSource &s = c->hw.src[G.sid];
  UVTime th = uv_time(s, t + dt / 2), tf = uv_time(s, t + dt);
  S.geo_slot_uv = s.level_slot[0];
  bool is3d = s.slot[S.geo_slot_uv].var_nz[VAR_U] > 1;
  dim3 g(nblk(p->n)), b(BLOCK);
  PView v = view(p);
  float f = (float)factor;
  // what-if runs: ODR_OCC_LDS=<bytes> of (unused) dynamic LDS per workgroup caps the workgroups per CU (160 KiB / bytes)
  static const size_t occ_lds = getenv("ODR_OCC_LDS") ? (size_t)atoll(getenv("ODR_OCC_LDS")) : 0;
#define ODR_LAUNCH(PROJ, D3) hipLaunchKernelGGL((k_step_grid<SCHEME, PROJ, D3, NOISE, 0, false, SM>), g, b, occ_lds, c->stream, c->dw, v, G, S, dt, f, th, tf, c->counter, N)
  switch (odr_proj_template(s.proj)) {
    case PROJ_LATLONG: if (is3d) ODR_LAUNCH(PROJ_LATLONG, true); else ODR_LAUNCH(PROJ_LATLONG, false); break;
    case PROJ_STERE_POLAR: if (is3d) ODR_LAUNCH(PROJ_STERE_POLAR, true); else ODR_LAUNCH(PROJ_STERE_POLAR, false); break;
    case PROJ_CURVILINEAR: if (is3d) ODR_LAUNCH(PROJ_CURVILINEAR, true); else ODR_LAUNCH(PROJ_CURVILINEAR, false); break;
    case PROJ_EXT: if (is3d) ODR_LAUNCH(PROJ_EXT, true); else ODR_LAUNCH(PROJ_EXT, false); break;
    default: if (is3d) ODR_LAUNCH(PROJ_STERE_EQUIT_SPHERE, true); else ODR_LAUNCH(PROJ_STERE_EQUIT_SPHERE, false); break;
  }
#undef ODR_LAUNCH
}

template <bool NOISE, int SM = 0>
static void step_dispatch(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G, const StepDesc &S, int scheme, double t, double dt,
                          double factor, const StageNoise &N) {
  if constexpr (SM == 0) { if (scheme == 0) { launch_step_grid<0, NOISE, 0>(c, p, G, S, t, dt, factor, N); return; } }
  if (scheme == 1) launch_step_grid<1, NOISE, SM>(c, p, G, S, t, dt, factor, N);
  else launch_step_grid<2, NOISE, SM>(c, p, G, S, t, dt, factor, N);
}

// odr_step_tile.hip: the fused step on the workgroup's LDS tile; false when it does not apply (the caller launches k_step_grid)
bool odr_i_step_tile(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G, StepDesc S, int scheme, double t, double dt,
                     double factor, const StageNoise &N);
// defined in odr_step_mix.hip: the step with OceanDrift.vertical_mixing inside the launch
void odr_i_step_mix(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G, const StepDesc &S, int scheme, double t, double dt,
                    double factor, const StepMix &M);
// developer build (-DODR_PHASE_TIMING): per-phase cycles of the k_step_grid instantiations of ONE translation unit (g_phase is
// a per-unit device variable), averaged per sampled wave
#ifdef ODR_PHASE_TIMING
#define ODR_DEFINE_PHASE_DUMP(NAME, LABEL)                                                                                      \
  void NAME() {                                                                                                                 \
    unsigned long long h[32];                                                                                                   \
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof h) != hipSuccess || !h[31]) return;                                  \
    static const char *nm[9] = {"entry->state loaded", "env sample (gathers+math)", "stores+bookkeeping", "geod origin+stage1 pos", \
                                "stage1 sample", "stage2 pos+sample", "stage3 pos+sample", "rk4 mix+final move", "final stores issue"}; \
    static const char *sub[6] = {"env: front door + coverage", "env: xi, yi", "env: zbracket", "env: footprint + nearest",       \
                                 "env: burst 1 (A, land)", "env: burst 2 (B, C, D)"};                                            \
    double tot = 0;                                                                                                             \
    for (int k = 0; k < 9; ++k) tot += (double)h[k] / (double)h[31];                                                             \
    fprintf(stderr, "k_step_grid phases, %s (cycles per wave, %llu waves, total %.0f):\n", LABEL, h[31], tot);                   \
    for (int k = 0; k < 9; ++k) fprintf(stderr, "  %-28s %9.0f  %5.1f %%\n", nm[k], (double)h[k] / (double)h[31], 100.0 * (double)h[k] / (double)h[31] / tot); \
    for (int k = 0; k < 6; ++k) fprintf(stderr, "      %-28s %9.0f\n", sub[k], (double)h[10 + k] / (double)h[31]);               \
  }
#else
#define ODR_DEFINE_PHASE_DUMP(NAME, LABEL) void NAME() {}
#endif
void odr_i_phase_dump_fast();
// defined in odr_step_fast.hip / odr_step_fast_noise.hip: the ODR_STAGE_FAST instantiations (Runge-Kutta schemes only; the
// kernels that serve any reader mix and the analytic double gyre take the mode at run time, StageNoise::sm)
bool odr_i_advect_fast(odr_ctx *c, odr_particles *p, int scheme, double t, double dt, double factor, const StageNoise &N);
void odr_i_step_fast(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G, const StepDesc &S, int scheme, double t, double dt,
                     double factor, const StageNoise &N);
bool odr_i_advect_fast_noise(odr_ctx *c, odr_particles *p, int scheme, double t, double dt, double factor, const StageNoise &N);
void odr_i_step_fast_noise(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G, const StepDesc &S, int scheme, double t, double dt,
                           double factor, const StageNoise &N);
// defined in odr_step_noise.hip
void odr_i_advect_noise(odr_ctx *c, odr_particles *p, int scheme, double t, double dt, double factor, const StageNoise &N);
void odr_i_step_noise(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G, const StepDesc &S, int scheme, double t, double dt,
                      double factor, const StageNoise &N);
