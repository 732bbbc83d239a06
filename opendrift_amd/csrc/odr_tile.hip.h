// The fused step with the field records in LDS (round 4): k_step_tile.
//
// After the spatial sort (k_sort_hist: 8x8-cell tiles) the particles of one workgroup sit in a few neighbouring grid
// cells, and every field access of the step -- the main-loop sample of the group (Environment.get_environment,
// basemodel/environment.py:499-923 -> ReaderBlock.interpolate, readers/interpolation/structured.py:107-163 ->
// Linear2DInterpolator, interpolators.py:105-139) and the one / three Runge-Kutta stage samples of advect_ocean_current
// (models/physics_methods.py:638-670) -- lands in the node rectangle around them.  k_step_grid makes ~50 gathers per
// particle for them, each keeping the texture addresser of the CU busy ~32 cycles per wave (64 lanes, 64 records): that
// unit, not HBM and not the VALUs, bounded the kernel (0.88 of the launch, profiles/r03_c3_pmc.json).  Here the workgroup
//   1. takes the node rectangle of its particles' footprints from the reader front door (+1 node of margin: a stage
//      position is within a cell of the particle), by a min / max reduction;
//   2. fetches the WHOLE node records of that rectangle at the (up to two) time levels of the step straight into LDS with
//      LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, contiguous along a row of the rectangle, no VGPRs, no
//      ds_write pass);
//   3. runs the same per-particle arithmetic as k_step_grid with every record read coming from LDS (LdTile instead of
//      LdGlobal: ds_read_b64 / b32 at 2 cycles per wave-instruction instead of 32).
// A particle whose main-sample footprint is not inside the rectangle (stragglers after in-place compaction) takes, in
// the same launch, the path through the blocks in HBM (nothing of it has been written by then); a stage position that
// leaves the rectangle (elements that outrun their neighbours) takes that one sample from HBM.  Results are bit-identical
// to k_step_grid whatever the rectangle is (tests/test_gpu_tile.py).
// Workgroups are cut along the sort: odr_sort_particles leaves a table of (first, count) ranges, each inside one 8x8-cell
// sort tile and at most BLOCK long (k_wg_count / k_wg_fill), so a workgroup's rectangle is ~11 x 7 nodes.
#pragma once

namespace odr {

#ifdef ODR_TU_MISC
// ---- the workgroup table of a sorted particle set.  ends[k] = end offset of sort key k (the cursor array of
// k_sort_perm after the scatter); keys are (tile * 64 + cell) [* bands + band], the last key collects the particles
// outside the grid.  Tile t holds [ends[spt t - 1], ends[spt t + spt - 1]); it is cut into ceil(count / BLOCK) ranges of equal length.
// (spt = keys per sort tile: 64 cells x depth bands of the sort)
__device__ __forceinline__ void wg_tile_range(const unsigned *__restrict__ ends, int t, int ntiles, unsigned spt, unsigned &start, unsigned &cnt) {
  start = t ? ends[(size_t)t * spt - 1] : 0u;
  const unsigned end = t < ntiles ? ends[(size_t)t * spt + spt - 1] : ends[(size_t)ntiles * spt];
  cnt = end - start;
}
__global__ __launch_bounds__(BLOCK) void k_wg_count(const unsigned *__restrict__ ends, int ntiles, unsigned spt, unsigned *__restrict__ nw) {
  const int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t > ntiles) return;
  unsigned start, cnt;
  wg_tile_range(ends, t, ntiles, spt, start, cnt);
  nw[t] = (cnt + BLOCK - 1) / BLOCK;
}
__global__ __launch_bounds__(BLOCK) void k_wg_fill(const unsigned *__restrict__ ends, int ntiles, unsigned spt, const unsigned *__restrict__ off,
                                                   unsigned *__restrict__ tab, unsigned cap) {
  const int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t > ntiles) return;
  unsigned start, cnt;
  wg_tile_range(ends, t, ntiles, spt, start, cnt);
  const unsigned nw = (cnt + BLOCK - 1) / BLOCK, o = off[t];
  for (unsigned j = 0; j < nw && o + j < cap; ++j) {
    const unsigned a = (unsigned)(((unsigned long long)cnt * j) / nw), b = (unsigned)(((unsigned long long)cnt * (j + 1)) / nw);
    tab[2 * (size_t)(o + j)] = start + a;
    tab[2 * (size_t)(o + j) + 1] = b - a;
  }
}
#endif  // ODR_TU_MISC

#ifdef ODR_TU_TILE
struct TileArgs {
  const unsigned *tab;                 // (first, count) per workgroup
  const unsigned long long *total;     // number of table entries (device: no host read between sort and launch)
  const float *lev[2];                 // node-record bases of the resident time levels the step reads
  int nlev;
  int mb, ma, hb, ha, fb, fa;          // index into lev[] of the level before / after of the main sample, the half-step stages, the full-step stage (a: -1 none)
  unsigned uv_off;                     // byte offset of the interleaved (u,v) pair in the node record
  int cap_nodes;                       // capacity of the dynamic LDS allocation in nodes per level
  unsigned *list;                      // particles whose main-sample footprint is outside their workgroup's rectangle: stepped by k_step_list
  unsigned long long *list_n;
  unsigned long long *stats;           // [1] workgroups whose rectangle was cut to the capacity
};

struct PState { double lon, lat, z; int moving, st; float age0, cdf0, ssh0; };

// One particle of k_step_grid's body -- sample, missing data, coastline, sea floor, age, previous state, advection -- with
// the record reads going through the loaders (Lm: main sample, Lh / Lf: half- and full-step stage samples).  Returns false,
// having written nothing, when the footprint of the main sample is outside Lm's image; a stage sample outside Lh / Lf is
// taken from the blocks in HBM on its own (advect_grid_body).
template <int SCHEME, int PROJ, bool IS3D, bool NOISE, int SM, class LD>
__device__ __forceinline__ bool step_particle(const DevWorld *__restrict__ W, const PView &p, const EnvGroupDesc &G, const StepDesc &S,
                                              double dt, float factor, const UVTime &th, const UVTime &tf, const StageNoise &N,
                                              const double *zt, const LD &Lm, const LD &Lh, const LD &Lf, long long i,
                                              const PState &q, const EnvFront &fr, bool &hit) {
  double lon = q.lon, lat = q.lat;
  const double z = q.z;
  int moving = q.moving, st = q.st;
  float out[MAXG];
  ZBracket zb_env;
  zb_env.iz0 = 0; zb_env.same = 0; zb_env.wa = 1;
  EnvExport X;
  X.valid = false; X.n00 = X.n11 = 0; X.iz0 = 0;
  if (!env_group_sample<PROJ, true, IS3D>(*W, G, Lm, fr, z, out, zt, zb_env, &X)) return false;
  UVKeep<IS3D> K = uv_keep_from<IS3D>(G, X, th);
  const int id = NOISE ? p.id[i] : 0;
  if (NOISE && S.main_noise) add_current_noise(N, 0, i, p.n, id, out[0], out[1]);
#pragma unroll
  for (int k = 0; k < MAXG; ++k)
    if (k < G.nv) G.out_ptr[k][i] = out[k];
  p.slon[i] = lon;
  p.slat[i] = lat;
  double zz = z;
  if (S.missing_code) {  // k_deactivate_missing
    bool miss = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < S.nmiss_grp) { const float e = pick_slot(out, S.miss_grp[k]); miss |= e != e; }
      if (k < S.nmiss_rest) { const float e = p.env[S.miss_rest[k]][i]; miss |= e != e; }
    }
    if (miss) {
      if (st == 0) p.status[i] = st = S.missing_code;
      p.moving[i] = moving = 0;
    }
  }
  if (S.coast_action) {  // k_coast
    const float land = S.land_slot == 2 ? out[2] : p.env[VAR_LAND][i];
    if (land == 1.0f) {
      hit = true;
      if (S.coast_action == 1) {
        if (z <= 0) {
          if (st == 0) p.status[i] = st = S.stranded_code;
          p.moving[i] = moving = 0;
        }
      } else {
        if (S.seeded_code > 0 && q.age0 == 0.0f) {
          if (st == 0) p.status[i] = st = S.seeded_code;
          p.moving[i] = moving = 0;
        }
        lon = p.plon[i];
        lat = p.plat[i];
        p.env[VAR_LAND][i] = 0.0f;   // self.environment.land_binary_mask[on_land] = 0 (:746)
      }
    }
  }
  if (S.seafloor) {  // k_seafloor
    const float dep = S.depth_slot == 2 ? out[2] : (S.depth_slot == 3 ? out[3] : p.env[VAR_DEPTH][i]);
    const float floorz = -__fadd_rn(dep, S.ssh_slot >= 0 ? pick_slot(out, S.ssh_slot) : q.ssh0);
    if (zz < (double)floorz) { zz = (double)floorz; p.z[i] = zz; }
  }
  if (S.age_dt != 0.0f) {  // k_age
    const float a = __fadd_rn(q.age0, S.age_dt);
    p.age[i] = a;
    if (S.max_age > 0 && a >= S.max_age) {
      if (st == 0) p.status[i] = st = S.retired_code;
      p.moving[i] = moving = 0;
    }
  }
  // deactivated (now or earlier, not yet compacted): the reference removes it before update() -- it does not move
  const bool skip = st != 0;
  if (S.store_previous) { p.plon[i] = lon; p.plat[i] = lat; }
  if (!skip) {
    const DevSource &s = W->src[G.sid];
    advect_grid_body<SCHEME, PROJ, IS3D, NOISE, SM, false>(s, s.slot[S.geo_slot_uv], lon, lat, zz, out[0], out[1],
                                                           __fmul_rn(current_factor(p, i, factor), q.cdf0), moving, dt, th, tf, Lh, Lf,
                                                           W->fallback[VAR_U], W->fallback[VAR_V], N, i, p.n, id, K, zb_env, IS3D && zz == z);
  }
  p.lon[i] = lon;
  p.lat[i] = lat;
  return true;
}

// LDS-DMA of one 16-byte piece per lane: LDS destination = `dst` (wave-uniform) + lane * 16
__device__ __forceinline__ void glds16(const char *src, __attribute__((address_space(3))) char *dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}

template <int SCHEME, int PROJ, bool IS3D, bool NOISE, int SM>
__global__ __launch_bounds__(BLOCK, ODR_TILE_WAVES) void k_step_tile(const DevWorld *__restrict__ W, PView p, EnvGroupDesc G, StepDesc S,
                                                                      double dt, float factor, UVTime th, UVTime tf,
                                                                      unsigned long long *n_hit, StageNoise N, TileArgs T) {
  extern __shared__ __attribute__((aligned(16))) char tile_mem[];
  __shared__ double s_zt[IS3D ? 3 * ZT_STRIDE : 1];
  __shared__ int s_red[BLOCK / 64][4];
  __shared__ int s_anchor[2];
  // The grid is the host's upper bound of the table length; the first `total` workgroups take the entries in the
  // XCD-contiguous order of pid(): the ranges of one region -- and the rectangles they fetch -- stay in one L2
  const unsigned nb = (unsigned)*T.total, b0 = blockIdx.x;
  if (b0 >= nb) return;
  const unsigned per = nb >> 3, rem = nb & 7u, xcd = b0 & 7u;
  const unsigned lb = xcd * per + (xcd < rem ? xcd : rem) + (b0 >> 3);
  const unsigned first = T.tab[2 * (size_t)lb], cnt = T.tab[2 * (size_t)lb + 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long i = (long long)first + tid;
  const bool live = (unsigned)tid < cnt && i < p.n;
  const DevSource &s = W->src[G.sid];
  const DevBlock &geo = s.slot[G.geo_slot];
  const double *zt = nullptr;
  if (IS3D) { zt_stage(s, s_zt); zt = s_zt; }
  auto load_state = [&]() {
    PState q;
    q.lon = q.lat = q.z = 0; q.moving = 0; q.st = 0; q.age0 = q.cdf0 = q.ssh0 = 0.f;
    if (live) {
      q.lon = p.lon[i]; q.lat = p.lat[i]; q.z = p.z[i];
      q.moving = p.moving[i]; q.st = p.status[i];
      q.age0 = p.age[i]; q.cdf0 = p.cdf[i];
      q.ssh0 = S.seafloor && p.env[VAR_SSH] ? p.env[VAR_SSH][i] : 0.f;
    }
    return q;
  };
  const PState q = load_state();
  const EnvFront fr = env_front<PROJ>(s, geo, q.lon, q.lat, q.z);
  // ---- 1. node rectangle of the workgroup's footprints
  constexpr int BIG = 0x3fffffff;
  int mnx = BIG, mxx = -BIG, mny = BIG, mxy = -BIG;
  int cx = BIG, cy = BIG;
  if (live && fr.covered) {
    const Axis ax = axis_fp(fr.xi, geo.nx), ay = axis_fp(fr.yi, geo.ny);
    mnx = ax.i0; mxx = ax.i1; mny = ay.i0; mxy = ay.i1;
    cx = ax.i0; cy = ay.i0;
    if (G.has_land) {
      const int jy = nearest_index(fr.y, geo.ymin, geo.yrange, geo.iyrange, geo.ny), jx = nearest_index(fr.x, geo.xmin, geo.xrange, geo.ixrange, geo.nx);
      mnx = min(mnx, jx); mxx = max(mxx, jx); mny = min(mny, jy); mxy = max(mxy, jy);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mnx = min(mnx, __shfl_xor(mnx, o, 64)); mxx = max(mxx, __shfl_xor(mxx, o, 64));
    mny = min(mny, __shfl_xor(mny, o, 64)); mxy = max(mxy, __shfl_xor(mxy, o, 64));
  }
  if (lane == 0) { s_red[wv][0] = mnx; s_red[wv][1] = mxx; s_red[wv][2] = mny; s_red[wv][3] = mxy; }
  if ((unsigned)tid == cnt / 2) { s_anchor[0] = cx; s_anchor[1] = cy; }
  __syncthreads();
  mnx = s_red[0][0]; mxx = s_red[0][1]; mny = s_red[0][2]; mxy = s_red[0][3];
#pragma unroll
  for (int k = 1; k < BLOCK / 64; ++k) {
    mnx = min(mnx, s_red[k][0]); mxx = max(mxx, s_red[k][1]); mny = min(mny, s_red[k][2]); mxy = max(mxy, s_red[k][3]);
  }
  TileRect R;
  R.x0 = R.y0 = R.w = R.h = 0;
  if (mxx >= mnx) {
    int x0 = max(mnx - 1, 0), x1 = min(mxx + 1, geo.nx - 1), y0 = max(mny - 1, 0), y1 = min(mxy + 1, geo.ny - 1);
    int w = x1 - x0 + 1, h = y1 - y0 + 1;
    if (w * h > T.cap_nodes) {   // stragglers stretch the rectangle: keep a window around the middle particle
      const int ax = s_anchor[0], ay = s_anchor[1];
      const int wn = min(w, 12), hn = min(h, T.cap_nodes / wn);
      if (ax != BIG) {
        x0 = min(max(ax - wn / 2 + 1, x0), x1 - wn + 1);
        y0 = min(max(ay - hn / 2 + 1, y0), y1 - hn + 1);
      }
      w = wn; h = hn;
      if (tid == 0 && T.stats) atomicAdd(&T.stats[1], 1ull);
    }
    R.x0 = x0; R.y0 = y0; R.w = w; R.h = h;
  }
  R.x0 = __builtin_amdgcn_readfirstlane(R.x0); R.y0 = __builtin_amdgcn_readfirstlane(R.y0);
  R.w = __builtin_amdgcn_readfirstlane(R.w); R.h = __builtin_amdgcn_readfirstlane(R.h);
  // ---- 2. the records of the rectangle, level by level and row by row: a row is one contiguous run of w records in the
  // block and in the image; lane j of a wave-instruction moves piece j of the run's next 1 KiB
  const unsigned recb = (unsigned)geo.rec * 4u;
  const unsigned lev_bytes = (unsigned)(R.w * R.h) * recb;
  __attribute__((address_space(3))) char *tile = (__attribute__((address_space(3))) char *)tile_mem;
  {
    const unsigned row_bytes = (unsigned)R.w * recb, npieces = row_bytes >> 4;
    for (int lv = 0; lv < T.nlev; ++lv) {
      const char *src_lev = (const char *)T.lev[lv];
      for (int row = wv; row < R.h; row += BLOCK / 64) {
        const char *src = src_lev + ((size_t)(R.y0 + row) * (size_t)geo.nx + (size_t)R.x0) * recb;
        __attribute__((address_space(3))) char *dst = tile + (unsigned)lv * lev_bytes + (unsigned)row * row_bytes;
        for (unsigned c0 = 0; c0 < npieces; c0 += 64) {
          const unsigned c = c0 + (unsigned)lane;
          if (c < npieces) glds16(src + ((size_t)c << 4), dst + (c0 << 4));
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // ---- 3. the particles
  bool hit = false;
  if (live) {
    auto image = [&](int k) { return (lds_cbyte *)(tile + (unsigned)(k < 0 ? 0 : k) * lev_bytes); };
    LdTile Lm, Lh, Lf;
    Lm.R = Lh.R = Lf.R = R;
    const bool tl = G.ba != nullptr && !G.all_static;
    Lm.b = image(T.mb); Lm.a = tl ? image(T.ma) : Lm.b;
    Lh.b = image(T.hb) + T.uv_off; Lh.a = image(T.ha >= 0 ? T.ha : T.hb) + T.uv_off;
    Lf.b = image(T.fb) + T.uv_off; Lf.a = image(T.fa >= 0 ? T.fa : T.fb) + T.uv_off;
    const bool done = step_particle<SCHEME, PROJ, IS3D, NOISE, SM>(W, p, G, S, dt, factor, th, tf, N, zt, Lm, Lh, Lf, i, q, fr, hit);
    if (!done) {   // outside the rectangle, nothing written: onto the list of k_step_list (one atomic per wave)
      const unsigned long long m = __ballot(1);
      const int leader = __builtin_ctzll(m);
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(T.list_n, (unsigned long long)__popcll(m));
      base = __shfl(base, leader, 64);
      T.list[base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull))] = (unsigned)i;
    }
  }
  if (S.coast_action) {
    unsigned long long bm = __ballot(hit);
    if (lane == 0 && bm) atomicAdd(n_hit, (unsigned long long)__popcll(bm));
  }
}
// The particles k_step_tile left out (T.list), stepped through the blocks in HBM: the same step_particle with the global
// loaders, i.e. the arithmetic of k_step_grid.  The list's length is only known on the device: the grid covers the whole
// set and the workgroups beyond the list return at once (no grid-stride loop: a loop keeps the ~60 array pointers of the
// launch alive across iterations -- 496 B of scratch per lane at 128 registers).
template <int SCHEME, int PROJ, bool IS3D, bool NOISE, int SM>
__global__ __launch_bounds__(BLOCK, (PROJ) == PROJ_LATLONG ? 4 : ODR_POLAR_STEP_WAVES) void k_step_list(const DevWorld *__restrict__ W, PView p, EnvGroupDesc G, StepDesc S,
                                                                           double dt, float factor, UVTime th, UVTime tf,
                                                                           unsigned long long *n_hit, StageNoise N, const unsigned *__restrict__ list,
                                                                           const unsigned long long *__restrict__ list_n, unsigned long long *stats) {
  __shared__ double s_zt[IS3D ? 3 * ZT_STRIDE : 1];
  const unsigned long long cnt = *list_n;
  if (blockIdx.x == 0 && threadIdx.x == 0 && stats && cnt) atomicAdd(&stats[0], cnt);
  if ((unsigned long long)blockIdx.x * BLOCK >= cnt) return;
  const DevSource &s = W->src[G.sid];
  const DevBlock &geo = s.slot[G.geo_slot];
  const double *zt = nullptr;
  if (IS3D) { zt_stage(s, s_zt); zt = s_zt; }
  const unsigned long long j = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  bool hit = false;
  if (j < cnt) {
    const long long i = (long long)list[j];
    PState q;
    q.lon = p.lon[i]; q.lat = p.lat[i]; q.z = p.z[i];
    q.moving = p.moving[i]; q.st = p.status[i];
    q.age0 = p.age[i]; q.cdf0 = p.cdf[i];
    q.ssh0 = S.seafloor && p.env[VAR_SSH] ? p.env[VAR_SSH][i] : 0.f;
    const EnvFront fr = env_front<PROJ>(s, geo, q.lon, q.lat, q.z);
    step_particle<SCHEME, PROJ, IS3D, NOISE, SM>(W, p, G, S, dt, factor, th, tf, N, zt, env_global(G), uv_global(th), uv_global(tf), i, q, fr, hit);
  }
  if (S.coast_action) {
    const unsigned long long bm = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && bm) atomicAdd(n_hit, (unsigned long long)__popcll(bm));
  }
}
#endif  // ODR_TU_TILE

}  // namespace odr
