// libodrift_hip.so, translation unit 8: the fused step with the field records of the workgroup's node rectangle in LDS
// (k_step_tile, odr_tile.hip.h) -- Runge-Kutta schemes, lon/lat and polar-stereographic readers, both stage arithmetics.
#define ODR_TU_STEP 1
#define ODR_TU_TILE 1
#ifndef ODR_TILE_WAVES
#define ODR_TILE_WAVES 4   // waves per SIMD the register allocation is held to (128 VGPRs)
#endif
#include "odr_step_launch.h"
#include "odr_tile.hip.h"

template <int SCHEME, int SM>
static void launch_tile(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G0, const StepDesc &S, double dt, double factor,
                        const UVTime &th, const UVTime &tf, const StageNoise &N, const TileArgs &T, size_t lds) {
  const EnvGroupDesc G = env_bind_out(G0, view(p));
  const DevSource &s = c->hw.src[G.sid];
  const bool is3d = s.slot[S.geo_slot_uv].var_nz[VAR_U] > 1;
  const dim3 g((unsigned)p->wg_grid), b(BLOCK);
  const PView v = view(p);
  const float f = (float)factor;
  const dim3 gl(nblk(p->n));
#define ODR_LAUNCH(PROJ, D3) do { \
    hipLaunchKernelGGL((k_step_tile<SCHEME, PROJ, D3, false, SM>), g, b, lds, c->stream, c->dw, v, G, S, dt, f, th, tf, c->counter, N, T); \
    hipLaunchKernelGGL((k_step_list<SCHEME, PROJ, D3, false, SM>), gl, b, 0, c->stream, c->dw, v, G, S, dt, f, th, tf, c->counter, N, T.list, T.list_n, T.stats); } while (0)
  if (odr_proj_template(s.proj) == PROJ_LATLONG) { if (is3d) ODR_LAUNCH(PROJ_LATLONG, true); else ODR_LAUNCH(PROJ_LATLONG, false); }
  else { if (is3d) ODR_LAUNCH(PROJ_STERE_POLAR, true); else ODR_LAUNCH(PROJ_STERE_POLAR, false); }
#undef ODR_LAUNCH
}

// Takes the step when the particle set carries a workgroup table of its last sort by the reader of the current
// (odr_sort_particles), the step reads at most two resident time levels and a useful rectangle fits the LDS budget;
// false: the caller launches k_step_grid.  ODR_TILE=1 switches it on, ODR_TILE_LDS=<bytes> sets the dynamic LDS per
// workgroup (default 38 KiB: four workgroups per CU), ODR_TILE_MIN_N the smallest particle count it is used for.
bool odr_i_step_tile(odr_ctx *c, odr_particles *p, const EnvGroupDesc &G, StepDesc S, int scheme, double t, double dt,
                     double factor, const StageNoise &N) {
  // Measured on C3 (profiles/r04_ab_variants.txt): bit-identical, but SLOWER than k_step_grid (0.89 vs 0.69 ms per launch at its
  // best LDS budget): between two sorts vertical shear spreads a workgroup's particles over ~16 x 11 nodes of 416 bytes -- more
  // than four workgroups per CU can hold -- and ranges cut along the sort tiles fill 70 % of the lanes.  It is opt-in.
  const bool off = !getenv("ODR_TILE") || atoi(getenv("ODR_TILE")) == 0;
  const long long min_n = getenv("ODR_TILE_MIN_N") ? atoll(getenv("ODR_TILE_MIN_N")) : 262144;
  const size_t lds_cfg = getenv("ODR_TILE_LDS") ? (size_t)atoll(getenv("ODR_TILE_LDS")) : 38 * 1024;
  if (off || N.on || scheme < 1 || scheme > 2 || p->win != 0 || p->n < min_n) return false;
  if (!p->wg_valid || p->wg_sid != G.sid || p->wg_src_gen != c->src_gen[G.sid] || p->n > p->wg_n || !p->wg_tab) return false;
  const DevSource &s = c->hw.src[G.sid];
  const int pt = odr_proj_template(s.proj);
  if (pt != PROJ_LATLONG && pt != PROJ_STERE_POLAR) return false;
  S.geo_slot_uv = s.level_slot[0];
  // the resident levels the step reads: bracket of t (main sample), t + dt/2 and t + dt (stage samples)
  int slots[6];
  host_bracket(s, t, slots[0], slots[1]);
  host_bracket(s, t + dt / 2, slots[2], slots[3]);
  host_bracket(s, t + dt, slots[4], slots[5]);
  if (s.always_valid) slots[1] = slots[3] = slots[5] = -1;
  TileArgs T;
  memset(&T, 0, sizeof T);
  int idx[6];
  int lev_slot[2] = {-1, -1};
  for (int k = 0; k < 6; ++k) {
    idx[k] = -1;
    if (slots[k] < 0) continue;
    for (int j = 0; j < T.nlev; ++j) if (lev_slot[j] == slots[k]) idx[k] = j;
    if (idx[k] < 0) {
      if (T.nlev == 2) return false;   // three levels in one step (a time step that straddles a level): the global path
      lev_slot[T.nlev] = slots[k];
      T.lev[T.nlev] = s.slot[slots[k]].base;
      idx[k] = T.nlev++;
    }
  }
  T.mb = idx[0]; T.ma = idx[1]; T.hb = idx[2]; T.ha = idx[3]; T.fb = idx[4]; T.fa = idx[5];
  // the group was built on the same bracket: its bases must be the levels found here
  if (T.lev[T.mb] != G.bb || (G.ba && (T.ma < 0 || T.lev[T.ma] != G.ba))) return false;
  const DevBlock &g0 = s.slot[S.geo_slot_uv];
  const size_t recb = (size_t)g0.rec * 4;
  for (int j = 0; j < T.nlev; ++j) {
    const DevBlock &bk = s.slot[lev_slot[j]];
    if (!bk.small || (size_t)bk.rec * 4 != recb || !bk.data[VAR_U] || (bk.data[VAR_U] - bk.base) != (g0.data[VAR_U] - g0.base)) return false;
  }
  if (recb % 16 != 0) return false;
  T.uv_off = (unsigned)((g0.data[VAR_U] - g0.base) * 4);
  if (T.uv_off % 8 != 0) return false;
  const size_t lds = lds_cfg & ~(size_t)15;
  T.cap_nodes = (int)(lds / (recb * (size_t)T.nlev));
  if (T.cap_nodes < 48) return false;    // records too long for a useful rectangle
  T.tab = p->wg_tab;
  T.total = p->wg_total;
  T.stats = p->wg_stats;
  if (p->wg_list_cap < p->n) {
    if (p->wg_list) { (void)hipStreamSynchronize(c->stream); (void)hipFree(p->wg_list); p->wg_list = nullptr; }
    if (hipMalloc((void **)&p->wg_list, sizeof(unsigned) * (size_t)p->cap) != hipSuccess) { (void)hipGetLastError(); p->wg_list_cap = 0; return false; }
    p->wg_list_cap = p->cap;
  }
  T.list = p->wg_list;
  T.list_n = p->wg_total + 1;
  if (hipMemsetAsync(T.list_n, 0, sizeof(unsigned long long), c->stream) != hipSuccess) return false;
  const UVTime th = uv_time(s, t + dt / 2), tf = uv_time(s, t + dt);
  if (N.sm == ODR_STAGE_FAST) {
    if (scheme == 1) launch_tile<1, 1>(c, p, G, S, dt, factor, th, tf, N, T, lds);
    else launch_tile<2, 1>(c, p, G, S, dt, factor, th, tf, N, T, lds);
  } else {
    if (scheme == 1) launch_tile<1, 0>(c, p, G, S, dt, factor, th, tf, N, T, lds);
    else launch_tile<2, 0>(c, p, G, S, dt, factor, th, tf, N, T, lds);
  }
  p->wg_launches++;
  return true;
}

int odr_particles_tile_stats(odr_ctx *c, odr_particles *p, uint64_t *out4) {
  REQUIRE(c && p && out4, "NULL argument");
  out4[0] = p->wg_launches; out4[1] = out4[2] = out4[3] = 0;
  if (!p->wg_total) return 0;
  unsigned long long h[4];
  D2H(h, p->wg_total, sizeof h);
  HIPCHK(hipStreamSynchronize(c->stream));
  out4[1] = h[2]; out4[2] = h[3]; out4[3] = p->wg_valid ? h[0] : 0;
  return 0;
}
