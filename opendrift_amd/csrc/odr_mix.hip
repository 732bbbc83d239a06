// libodrift_hip.so, translation unit 3: vertical mixing (reader profiles / wind-parameterised profiles) and the
// OpenOil physics inside the mixing loop.  See odrift.hip for the rest.
#define ODR_TU_MIX 1
#include "odr_host.h"

int odr_vmix(odr_ctx *c, odr_particles *p, double t, double dt, double dt_mix, int mix_at_surface, int rng_mode,
             const double *huni, uint64_t step) {
  const bool guarded = c->guard_next_vmix != 0;     // odr_ctx_guard_next_vmix: THIS call only, however it ends
  c->guard_next_vmix = 0;
  // a guarded call that cannot honour the guard (host-drawn numbers, OpenOil's loop: other kernel families) launches nothing and
  // consumes nothing: the caller calls again, unguarded
  if (guarded && (rng_mode == ODR_RNG_HOST || c->oil_owner == p)) return 1;
  p->status_epoch++;
  p->epoch++;  // invalidates the cached reductions (reduce())
  REQUIRE(dt_mix > 0 && dt != 0, "bad time steps");
  if (!p->env[VAR_DEPTH]) return fail(ODR_ERR_STATE, "sea_floor_depth_below_sea_level has not been sampled");
  int rc = ensure_env(c, p, VAR_SSH);
  if (rc) return rc;
  if ((rc = flush_world(c))) return rc;
  if (p->n == 0) return 0;
  int nzp = 1;
  for (int k = 0; k < c->hw.nlist[VAR_KZ]; ++k) {
    const DevSource &s = c->hw.src[c->hw.list[VAR_KZ][k]];
    if (s.kind == SRC_GRID) { nzp = s.nz > 1 ? s.nz : 1; break; }
  }
  double *du = nullptr;
  if (rng_mode == ODR_RNG_HOST) {
    REQUIRE(huni, "host uniforms required in ODR_RNG_HOST mode");
    int ntimes = abs((int)(dt / (dt_mix * (dt > 0 ? 1 : -1))));
    void *s;
    size_t n = (size_t)ntimes * (size_t)p->n;
    if ((rc = scratch(c, p, sizeof(double) * n, &s))) return rc;
    du = (double *)s;
    H2D(du, huni, sizeof(double) * n);
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  size_t lds = sizeof(double) * ((size_t)nzp * BLOCK + 3 * (size_t)nzp);
  dim3 g(nblk(p->n)), b(BLOCK);
  PView v = view(p);
  unsigned long long st = (unsigned long long)step;
  int vadv = c->fuse_vadv;
  c->fuse_vadv = -1;
  if (vadv >= 0 && !p->env[VAR_W]) return fail(ODR_ERR_STATE, "upward_sea_water_velocity has not been sampled");
  // fast column kernel: K from one gridded reader, plain z-innermost array on every resident level
  const bool oil = c->oil_owner == p;   // OpenOil: terminal velocities, slick and wave entrainment inside the loop
  c->oil_owner = nullptr;
  VMixDesc D;
  // the step's profiles were sampled in the float32 position class (a run's first get_environment, odr_ctx_set_position_class):
  // the generic kernel forms the column's footprint with float32 index maps (DevWorld::f32pos bit 1, set for this launch)
  // odr_vmix_set_profile_levels: the columns end at that level (a reader that cut its block at the depth asked of it): generic kernel
  const int cut = c->vmix_levels;
  c->vmix_levels = 0;
  if (cut > 0 && guarded) return 1;
  const bool f32prof = p->profiles_f32;
  if (f32prof) { c->hw.f32pos |= 2; c->dirty = true; if ((rc = flush_world(c))) return rc; }
  const bool fast = !oil && !f32prof && cut <= 0 && !getenv("ODR_NO_FAST_PATH") && build_vmix_desc(c, t, D);
  if (cut > 0) {
    REQUIRE(cut >= 2 && cut <= 255, "odr_vmix_set_profile_levels: 2 .. 255 levels");
    if (cut < nzp) nzp = cut;
  }
  const int sfl_cut = c->seafloor | ((cut > 0 ? nzp : 0) << 16);     // (the generic kernels read the cut from bits 16 .. 23)
  if (guarded && !fast) { c->fuse_vadv = vadv; return 1; }   // (the generic kernel carries no guard) nothing launched
  if (fast) {
    if (guarded) D.guard = c->counter + 4;
    const int nq = (nzp + 3) / 4;
    const bool tl = D.ka != nullptr;
    // k_vmix_win (five levels per particle in registers: cost independent of the number of reader levels) from 13 levels
    // on, the whole-column kernel below that -- measured on MI355X, 4 M particles (tools/vmix_levels.py): 8 / 12 / 16 / 24 /
    // 40 levels: column 0.155 / 0.176 / 0.218 / 0.327 / 1.268 ms, window 0.167 / 0.186 / 0.207 / 0.237 / 0.274 ms.
    // ODR_VMIX_WINDOW=1 / 0 forces one of them (the parity tests compare the two).
    static const size_t vmix_pad = getenv("ODR_VMIX_LDS_PAD") ? (size_t)atoll(getenv("ODR_VMIX_LDS_PAD")) : 0;   // what-if runs: fewer workgroups per CU
    const char *wenv = getenv("ODR_VMIX_WINDOW");
    const bool win = nzp >= 3 && nzp <= BLOCK && (wenv ? atoi(wenv) != 0 : nzp > 12);
#define VMIX_COL(NQ)                                                                                              \
  do {                                                                                                            \
    size_t l2 = sizeof(double) * ((size_t)(4 * NQ) * BLOCK + 4 * (size_t)(4 * NQ)) + vmix_pad;                    \
    if (tl) hipLaunchKernelGGL((k_vmix_col<NQ, true>), g, b, l2, c->stream, c->dw, v, D, dt, dt_mix,              \
                               mix_at_surface, rng_mode, du, c->seed, st, vadv, c->seafloor);                                  \
    else hipLaunchKernelGGL((k_vmix_col<NQ, false>), g, b, l2, c->stream, c->dw, v, D, dt, dt_mix,                \
                            mix_at_surface, rng_mode, du, c->seed, st, vadv, c->seafloor);                                     \
  } while (0)
    if (win) {
      const size_t lw = (sizeof(double) + sizeof(unsigned)) * 4 * BLOCK + sizeof(double) * 4 * (size_t)nzp;
      if (tl) hipLaunchKernelGGL((k_vmix_win<true>), g, b, lw, c->stream, c->dw, v, D, dt, dt_mix, mix_at_surface, rng_mode, du,
                                 c->seed, st, vadv, c->seafloor);
      else hipLaunchKernelGGL((k_vmix_win<false>), g, b, lw, c->stream, c->dw, v, D, dt, dt_mix, mix_at_surface, rng_mode, du,
                              c->seed, st, vadv, c->seafloor);
    }
    // the smallest instantiated quad count >= nq; over-read stays inside the 64-byte array padding
    else if (nq <= 1) VMIX_COL(1);
    else if (nq == 2) VMIX_COL(2);
    else if (nq == 3) VMIX_COL(3);
    else if (nq == 4) VMIX_COL(4);
    else if (nq <= 6) VMIX_COL(6);
    else if (nq <= 8) VMIX_COL(8);
    else if (nq <= 12) VMIX_COL(12);
    else VMIX_COL(16);
#undef VMIX_COL
  } else if (oil) {
    if (nzp <= 16) hipLaunchKernelGGL((k_vmix<16, true>), g, b, lds, c->stream, c->dw, v, t, dt, dt_mix, mix_at_surface, rng_mode, du, c->seed, st, vadv, sfl_cut, c->oil);
    else if (nzp <= 32) hipLaunchKernelGGL((k_vmix<32, true>), g, b, lds, c->stream, c->dw, v, t, dt, dt_mix, mix_at_surface, rng_mode, du, c->seed, st, vadv, sfl_cut, c->oil);
    else hipLaunchKernelGGL((k_vmix<1, true>), g, b, lds, c->stream, c->dw, v, t, dt, dt_mix, mix_at_surface, rng_mode, du, c->seed, st, vadv, sfl_cut, c->oil);
  } else if (nzp <= 16) hipLaunchKernelGGL(k_vmix<16>, g, b, lds, c->stream, c->dw, v, t, dt, dt_mix, mix_at_surface, rng_mode, du, c->seed, st, vadv, sfl_cut);
  else if (nzp <= 32) hipLaunchKernelGGL(k_vmix<32>, g, b, lds, c->stream, c->dw, v, t, dt, dt_mix, mix_at_surface, rng_mode, du, c->seed, st, vadv, sfl_cut);
  else hipLaunchKernelGGL(k_vmix<1>, g, b, lds, c->stream, c->dw, v, t, dt, dt_mix, mix_at_surface, rng_mode, du, c->seed, st, vadv, sfl_cut);
  HIPCHK(hipGetLastError());
  if (f32prof) { c->hw.f32pos &= ~2; c->dirty = true; }
  return 0;
}

// vertical_mixing with an analytical diffusivity model (oceandrift.py:385-395,448-458): also what the default
// 'environment' model does when no reader provides ocean_vertical_diffusivity (:431-447 -> Large et al. 1994)
// The K columns of the next odr_vmix end at level n - 1 (a reader that hands out the levels asked for cuts its block at the depth
// of the request -- drift:truncate_ocean_model_below_m, environment.py:554-566 -> reader_netCDF_CF_generic.py:414-423 -- and
// elements below mix on K and dK/dz of the last level held); n = 0: every level.
int odr_vmix_set_profile_levels(odr_ctx *c, int32_t n) {
  REQUIRE(n == 0 || (n >= 2 && n <= 255), "bad level count %d", n);
  c->vmix_levels = n;
  return 0;
}

int odr_vmix_wind_profile(odr_ctx *c, odr_particles *p, int model, double background_diffusivity, double dt,
                          double dt_mix, int mix_at_surface, int rng_mode, const double *huni, uint64_t step) {
  p->status_epoch++;
  p->epoch++;
  REQUIRE(model == ODR_DIFFUSIVITY_LARGE1994 || model == ODR_DIFFUSIVITY_SUNDBY1983, "Unknown diffusivity model: %d", model);
  REQUIRE(dt_mix > 0 && dt != 0, "bad time steps");
  if (!p->env[VAR_DEPTH]) return fail(ODR_ERR_STATE, "sea_floor_depth_below_sea_level has not been sampled");
  int rc;
  for (int v : {VAR_SSH, VAR_XWIND, VAR_YWIND, VAR_MLD})
    if ((rc = ensure_env(c, p, v))) return rc;
  if ((rc = flush_world(c))) return rc;
  if (p->n == 0) return 0;
  p->epoch++;  // the reduction must see this step's mixed-layer depths
  if ((rc = reduce(c, p, 0.0, 0, false, false))) return rc;
  double *du = nullptr;
  if (rng_mode == ODR_RNG_HOST) {
    REQUIRE(huni, "host uniforms required in ODR_RNG_HOST mode");
    int ntimes = abs((int)(dt / (dt_mix * (dt > 0 ? 1 : -1))));
    void *s;
    size_t n = (size_t)ntimes * (size_t)p->n;
    if ((rc = scratch(c, p, sizeof(double) * n, &s))) return rc;
    du = (double *)s;
    H2D(du, huni, sizeof(double) * n);
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  int vadv = c->fuse_vadv;
  c->fuse_vadv = -1;
  if (vadv >= 0 && !p->env[VAR_W]) return fail(ODR_ERR_STATE, "upward_sea_water_velocity has not been sampled");
  dim3 g(nblk(p->n)), b(BLOCK);
  const bool oil = c->oil_owner == p;
  c->oil_owner = nullptr;
  if (oil && model == ODR_DIFFUSIVITY_LARGE1994)
    hipLaunchKernelGGL((k_vmix_wind<DIFF_LARGE1994, true>), g, b, 0, c->stream, view(p), c->red, background_diffusivity, dt, dt_mix,
                       mix_at_surface, rng_mode, du, c->seed, (unsigned long long)step, vadv, c->seafloor, c->oil);
  else if (oil)
    hipLaunchKernelGGL((k_vmix_wind<DIFF_SUNDBY1983, true>), g, b, 0, c->stream, view(p), c->red, background_diffusivity, dt, dt_mix,
                       mix_at_surface, rng_mode, du, c->seed, (unsigned long long)step, vadv, c->seafloor, c->oil);
  else if (model == ODR_DIFFUSIVITY_LARGE1994)
    hipLaunchKernelGGL(k_vmix_wind<DIFF_LARGE1994>, g, b, 0, c->stream, view(p), c->red, background_diffusivity, dt, dt_mix,
                       mix_at_surface, rng_mode, du, c->seed, (unsigned long long)step, vadv, c->seafloor);
  else
    hipLaunchKernelGGL(k_vmix_wind<DIFF_SUNDBY1983>, g, b, 0, c->stream, view(p), c->red, background_diffusivity, dt, dt_mix,
                       mix_at_surface, rng_mode, du, c->seed, (unsigned long long)step, vadv, c->seafloor);
  HIPCHK(hipGetLastError());
  p->epoch++;
  return 0;
}

// OpenOil.prepare_vertical_mixing (models/openoil/openoil.py:1017-1031) on the device, and the switch that makes the
// next odr_vmix / odr_vmix_wind_profile call on these particles run OpenOil's version of the loop: terminal velocity
// of the droplets in every sub-step (:922-998), slick formation (:1056-1061), wave entrainment (:1033-1054).
int odr_oil_prepare_mixing(odr_ctx *c, odr_particles *p, double dt, double dt_mix, double interfacial_tension,
                           double sea_water_density, int droplet_distribution, int keep_droplet_diameter, int hs_mode,
                           int tp_mode, int temperature_to_kelvin, int rng_mode, const double *host_u_diameter,
                           const double *host_u_entrain, const double *host_u_intrusion, uint64_t step) {
  c->oil_owner = nullptr;
  REQUIRE(dt_mix > 0 && dt != 0, "bad time steps");
  REQUIRE(droplet_distribution == ODR_DROPLETS_JOHANSEN2015 || droplet_distribution == ODR_DROPLETS_LI2017,
          "no wave entrainment droplet size distribution specified");      // openoil.py:1070
  REQUIRE(interfacial_tension > 0 && sea_water_density > 0, "bad oil / water constants");
  REQUIRE(hs_mode >= 0 && hs_mode <= 2 && tp_mode >= 0 && tp_mode <= 3, "bad wave options");
  for (int k : {OIL_DIAMETER, OIL_DENSITY, OIL_VISCOSITY, OIL_FILM})
    if (!p->aux[k]) return fail(ODR_ERR_STATE, "oil property slot %d has not been set", k);
  for (int v : {VAR_XWIND, VAR_YWIND, VAR_TEMP, VAR_SALT})
    if (!p->env[v]) return fail(ODR_ERR_STATE, "wind, sea_water_temperature and sea_water_salinity must have been sampled");
  if ((hs_mode == 0 && !p->env[VAR_HS]) || (tp_mode == 0 && !p->env[VAR_TP])) return fail(ODR_ERR_STATE, "Hs/Tp not sampled");
  if (!p->aux[OIL_DIAMETER_IF_ENTRAINED]) {
    HIPCHK(hipMalloc((void **)&p->aux[OIL_DIAMETER_IF_ENTRAINED], sizeof(float) * (size_t)p->cap));
    HIPCHK(hipMemsetAsync(p->aux[OIL_DIAMETER_IF_ENTRAINED], 0, sizeof(float) * (size_t)p->cap, c->stream));
  }
  if (!c->oil_stat) {
    HIPCHK(hipMalloc((void **)&c->oil_stat, sizeof(double) * OIL_STAT_N));
    HIPCHK(hipMalloc((void **)&c->oil_cdf, sizeof(double) * OIL_NSPEC));
    HIPCHK(hipMalloc((void **)&c->oil_chunk, sizeof(double) * OIL_SPEC_BLOCKS));
    HIPCHK(hipMalloc((void **)&c->oil_guide, sizeof(int) * (OIL_GUIDE + 1)));
  }
  if (p->n == 0) return 0;
  const int ntimes = abs((int)(dt / (dt_mix * (dt > 0 ? 1 : -1))));
  REQUIRE(rng_mode == ODR_RNG_HOST || ntimes <= OIL_MAX_SUBSTEPS_DEVICE_RNG, "more than %d mixing sub-steps per step", OIL_MAX_SUBSTEPS_DEVICE_RNG);
  const unsigned nb = nblk(p->n);
  if (c->oil_part_n < 2 * (size_t)nb) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->oil_part) HIPCHK(hipFree(c->oil_part));
    HIPCHK(hipMalloc((void **)&c->oil_part, sizeof(double) * 2 * (size_t)nb));
    c->oil_part_n = 2 * (size_t)nb;
  }
  OilArgs &a = c->oil;
  memset(&a, 0, sizeof a);
  a.keep_diameter = keep_droplet_diameter ? 1 : 0;
  a.hs_mode = hs_mode; a.tp_mode = tp_mode; a.to_kelvin = temperature_to_kelvin ? 1 : 0;
  a.droplets = droplet_distribution; a.rng_mode = rng_mode;
  a.sigma_ow = interfacial_tension; a.rho_w = sea_water_density; a.dt_mix_cfg = dt_mix;
  a.stat = c->oil_stat;
  const double *du_d = nullptr;
  if (rng_mode == ODR_RNG_HOST) {
    REQUIRE(host_u_diameter && host_u_entrain && host_u_intrusion, "host uniforms required in ODR_RNG_HOST mode");
    const size_t per = (size_t)ntimes * (size_t)p->n, need = 2 * per + (size_t)p->n;
    if (c->oil_u_n < need) {
      HIPCHK(hipStreamSynchronize(c->stream));
      if (c->oil_u) HIPCHK(hipFree(c->oil_u));
      HIPCHK(hipMalloc((void **)&c->oil_u, sizeof(double) * need));
      c->oil_u_n = need;
    }
    H2D(c->oil_u, host_u_entrain, sizeof(double) * per);
    H2D(c->oil_u + per, host_u_intrusion, sizeof(double) * per);
    H2D(c->oil_u + 2 * per, host_u_diameter, sizeof(double) * (size_t)p->n);
    HIPCHK(hipStreamSynchronize(c->stream));   // the host arrays are pageable and may change right after
    a.u_ent = c->oil_u; a.u_int = c->oil_u + per; du_d = c->oil_u + 2 * per;
  }
  const PView v = view(p);
  const dim3 g(nb), b(BLOCK);
  if (c->oil_override) {   // odr_oil_set_mixing_stats: the means over the elements of ALL ranks of a sharded run
    c->oil_override = 0;
    double h[OIL_STAT_N] = {0};
    h[OIL_STAT_MEAN_ZB] = c->oil_stat_host[0];
    h[OIL_STAT_DV50] = c->oil_stat_host[1];
    H2D(c->oil_stat, h, sizeof h);
    HIPCHK(hipStreamSynchronize(c->stream));
  } else {
    hipLaunchKernelGGL(k_oil_stats, g, b, 0, c->stream, v, a, c->oil_part);
    hipLaunchKernelGGL(k_oil_stats_final, dim3(1), b, 0, c->stream, c->oil_part, (int)nb, (long long)p->n, c->oil_stat);
  }
  hipLaunchKernelGGL(k_oil_spectrum_sums, dim3(OIL_SPEC_BLOCKS), b, 0, c->stream, c->oil_stat, c->oil_chunk);
  hipLaunchKernelGGL(k_oil_spectrum_offsets, dim3(1), dim3(64), 0, c->stream, c->oil_chunk, c->oil_stat);
  hipLaunchKernelGGL(k_oil_spectrum_scan, dim3(OIL_SPEC_BLOCKS), b, 0, c->stream, c->oil_stat, c->oil_chunk, c->oil_cdf);
  hipLaunchKernelGGL(k_oil_guide, dim3((OIL_GUIDE + 1 + BLOCK - 1) / BLOCK), b, 0, c->stream, c->oil_cdf, c->oil_guide);
  hipLaunchKernelGGL(k_oil_choice, g, b, 0, c->stream, v, c->oil_cdf, c->oil_guide, rng_mode, du_d, c->seed,
                     (unsigned long long)step);
  HIPCHK(hipGetLastError());
  c->oil_owner = p;
  return 0;
}

// Sharded run: the two means OpenOil takes over ALL elements -- np.mean(dV_50) that parameterises the droplet spectrum
// (openoil.py:1099-1101,1156-1158) and np.mean(1.5 Hs), the intrusion depth scale (:1047) -- as sums over this rank's
// elements; the caller adds them over the ranks and installs the means with odr_oil_set_mixing_stats before
// odr_oil_prepare_mixing, which then skips its own reduction.
int odr_oil_local_sums(odr_ctx *c, odr_particles *p, double interfacial_tension, double sea_water_density,
                       int droplet_distribution, int hs_mode, double *sum_dv50, double *sum_zb) {
  REQUIRE(sum_dv50 && sum_zb, "NULL output");
  *sum_dv50 = *sum_zb = 0;
  REQUIRE(droplet_distribution == ODR_DROPLETS_JOHANSEN2015 || droplet_distribution == ODR_DROPLETS_LI2017,
          "no wave entrainment droplet size distribution specified");
  if (p->n == 0) return 0;
  for (int k : {OIL_DENSITY, OIL_VISCOSITY, OIL_FILM})
    if (!p->aux[k]) return fail(ODR_ERR_STATE, "oil property slot %d has not been set", k);
  if (!p->env[VAR_XWIND] || !p->env[VAR_YWIND] || (hs_mode == 0 && !p->env[VAR_HS])) return fail(ODR_ERR_STATE, "wind / Hs not sampled");
  const unsigned nb = nblk(p->n);
  double *part = nullptr, *stat = nullptr;
  HIPCHK(hipMalloc((void **)&part, sizeof(double) * (2 * (size_t)nb + OIL_STAT_N)));
  stat = part + 2 * (size_t)nb;
  OilArgs a;
  memset(&a, 0, sizeof a);
  a.hs_mode = hs_mode; a.droplets = droplet_distribution; a.sigma_ow = interfacial_tension; a.rho_w = sea_water_density;
  hipLaunchKernelGGL(k_oil_stats, dim3(nb), dim3(BLOCK), 0, c->stream, view(p), a, part);
  hipLaunchKernelGGL(k_oil_stats_final, dim3(1), dim3(BLOCK), 0, c->stream, part, (int)nb, 1LL, stat);   // n = 1: the sums
  double h[OIL_STAT_N];
  D2H(h, stat, sizeof h);
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipFree(part));
  *sum_dv50 = h[OIL_STAT_DV50];
  *sum_zb = h[OIL_STAT_MEAN_ZB];
  return 0;
}
int odr_oil_set_mixing_stats(odr_ctx *c, double mean_zb, double dv50) {
  c->oil_stat_host[0] = (double)(float)mean_zb;   // np.mean of a float32 array is float32
  c->oil_stat_host[1] = dv50;
  c->oil_override = 1;
  return 0;
}

// mean intrusion depth scale np.mean(1.5 Hs) and the spectrum median dV_50 of the last odr_oil_prepare_mixing
int odr_oil_mixing_stats(odr_ctx *c, double *mean_zb, double *dv50) {
  if (!c->oil_stat) return fail(ODR_ERR_STATE, "odr_oil_prepare_mixing has not run");
  double h[OIL_STAT_N];
  D2H(h, c->oil_stat, sizeof h);
  HIPCHK(hipStreamSynchronize(c->stream));
  if (mean_zb) *mean_zb = h[OIL_STAT_MEAN_ZB];
  if (dv50) *dv50 = h[OIL_STAT_DV50];
  return 0;
}
