// libodrift_hip.so, translation unit 4: the advection kernels with drift:current_uncertainty(_uniform) inside the
// Runge-Kutta stage calls (environment.py:869-886 within physics_methods.py:638-670) -- the NOISE = true instantiations.
#define ODR_TU_STEP 1
#include "odr_step_launch.h"

void odr_i_advect_noise(odr_ctx *c, odr_particles *p, int scheme, double t, double dt, double factor, const StageNoise &N) {
  advect_dispatch<true>(c, p, scheme, t, dt, factor, N);
}
void odr_i_step_noise(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G, const StepDesc &S, int scheme, double t, double dt,
                      double factor, const StageNoise &N) {
  step_dispatch<true>(c, p, G, S, scheme, t, dt, factor, N);
}
