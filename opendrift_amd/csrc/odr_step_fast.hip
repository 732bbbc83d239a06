// libodrift_hip.so, translation unit 6: the Runge-Kutta advection kernels with ODR_STAGE_FAST stage math
// (odr_ctx_set_stage_math; stage_pos<1> / uv_sample_stage_f32), without current uncertainty.
#define ODR_TU_STEP 1
#include "odr_step_launch.h"

bool odr_i_advect_fast(odr_ctx *c, odr_particles *p, int scheme, double t, double dt, double factor, const StageNoise &N) {
  return advect_dispatch<false, 1>(c, p, scheme, t, dt, factor, N);
}
void odr_i_step_fast(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G, const StepDesc &S, int scheme, double t, double dt,
                     double factor, const StageNoise &N) {
  step_dispatch<false, 1>(c, p, G, S, scheme, t, dt, factor, N);
}
ODR_DEFINE_PHASE_DUMP(odr_i_phase_dump_fast, "fast stage math")
