// libodrift_hip.so, translation unit 5: the fused step with OceanDrift.vertical_mixing (+ vertical_advection) inside the
// launch -- k_step_grid<SCHEME, PROJ_LATLONG, 3-D, ..., MIXQ, MIXTL> for K columns of up to 16 levels.
#define ODR_TU_STEP 1
#include "odr_step_launch.h"

template <int SCHEME, int NQ, bool TL, int SM>
static void launch_mix_sm(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G0, StepDesc S, double t, double dt, double factor,
                          const StepMix &M) {
  const EnvGroupDesc G = env_bind_out(G0, view(p));
  const DevSource &s = c->hw.src[G.sid];
  UVTime th = uv_time(s, t + dt / 2), tf = uv_time(s, t + dt);
  S.geo_slot_uv = s.level_slot[0];
  StageNoise N;
  memset(&N, 0, sizeof N);
  const size_t lds = sizeof(double) * ((size_t)(4 * NQ) * BLOCK + 4 * (size_t)(4 * NQ));
  N.sm = SM;
  hipLaunchKernelGGL((k_step_grid<SCHEME, PROJ_LATLONG, true, false, NQ, TL, SM>), dim3(nblk(p->n)), dim3(BLOCK), lds, c->stream,
                     c->dw, view(p), G, S, dt, (float)factor, th, tf, c->counter, N, M);
}
template <int SCHEME, int NQ, bool TL>
static void launch_mix(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G0, const StepDesc &S, double t, double dt, double factor,
                       const StepMix &M) {
  if constexpr (SCHEME > 0) {
    if (c->stage_math == ODR_STAGE_FAST) { launch_mix_sm<SCHEME, NQ, TL, 1>(c, p, G0, S, t, dt, factor, M); return; }
  }
  launch_mix_sm<SCHEME, NQ, TL, 0>(c, p, G0, S, t, dt, factor, M);
}

template <int SCHEME>
static void launch_mix_scheme(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G, const StepDesc &S, double t, double dt,
                              double factor, const StepMix &M) {
  const int nq = (M.D.nzp + 3) / 4;
  const bool tl = M.D.ka != nullptr;
#define ODR_MIX(NQ) do { if (tl) launch_mix<SCHEME, NQ, true>(c, p, G, S, t, dt, factor, M); \
                         else launch_mix<SCHEME, NQ, false>(c, p, G, S, t, dt, factor, M); } while (0)
  if (nq <= 2) ODR_MIX(2);      // the smallest instantiated quad count >= nq (over-read stays inside the record padding)
  else if (nq == 3) ODR_MIX(3);
  else ODR_MIX(4);
#undef ODR_MIX
}

void odr_i_step_mix(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G, const StepDesc &S, int scheme, double t, double dt,
                    double factor, const StepMix &M) {
  if (scheme == 0) launch_mix_scheme<0>(c, p, G, S, t, dt, factor, M);
  else if (scheme == 1) launch_mix_scheme<1>(c, p, G, S, t, dt, factor, M);
  else launch_mix_scheme<2>(c, p, G, S, t, dt, factor, M);
}
