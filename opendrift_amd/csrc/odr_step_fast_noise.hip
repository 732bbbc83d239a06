// libodrift_hip.so, translation unit 7: ODR_STAGE_FAST stage math with drift:current_uncertainty(_uniform) inside the
// Runge-Kutta stage calls (the NOISE = true, SM = 1 instantiations).
#define ODR_TU_STEP 1
#include "odr_step_launch.h"

bool odr_i_advect_fast_noise(odr_ctx *c, odr_particles *p, int scheme, double t, double dt, double factor, const StageNoise &N) {
  return advect_dispatch<true, 1>(c, p, scheme, t, dt, factor, N);
}
void odr_i_step_fast_noise(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G, const StepDesc &S, int scheme, double t, double dt,
                           double factor, const StageNoise &N) {
  step_dispatch<true, 1>(c, p, G, S, scheme, t, dt, factor, N);
}
