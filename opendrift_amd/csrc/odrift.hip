// libodrift_hip.so -- host side of the C ABI declared in include/odrift.h.
// One context = one HIP stream + the device image of the Environment (field sources,
// priority lists) ; one particle set = SoA arrays in HBM.  Everything is launched on the
// context stream; only calls that hand data back to the host synchronise.
// Translation unit 1 of 3 (odr_step.hip, odr_mix.hip): context, particle sets, field blocks, environment sample,
// Euler movers, bookkeeping, compaction / sort, result buffer, sigma grids.
#define ODR_TU_MISC 1
#include "odr_host.h"
#include "odr_mesh.h"

static thread_local std::string g_err;
int odr_i_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

const char *odr_last_error(void) { return g_err.c_str(); }
#define ODR_STR2(x) #x
#define ODR_STR(x) ODR_STR2(x)
// (tests read the round count of the mixing stream from here: oracle/philox.py restates the generator with that count)
const char *odr_version(void) { return "odrift-hip 0.2 (gfx950; mixing stream Philox4x32-" ODR_STR(ODR_MIX_ROUNDS) ")"; }

int odr_ctx_create(int device, uint64_t seed, odr_ctx **out) {
  REQUIRE(out, "out is NULL");
  HIPCHK(hipSetDevice(device));
  odr_ctx *c = new odr_ctx();
  c->device = device;
  c->seed = seed;
  HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  memset(&c->hw, 0, sizeof(DevWorld));
  for (int v = 0; v < NVAR; ++v) c->hw.fallback[v] = NAN;
  HIPCHK(hipMalloc((void **)&c->dw, sizeof(DevWorld)));
  HIPCHK(hipMalloc((void **)&c->red, sizeof(double) * R_N));
  HIPCHK(hipMalloc((void **)&c->counter, sizeof(unsigned long long) * 8));   // [0] on land, [1] kept, [2] status flags, [3] tickets, [4] "every element stays"
  HIPCHK(hipMemset(c->counter, 0, sizeof(unsigned long long) * 8));
  HIPCHK(hipEventCreateWithFlags(&c->scan_ev, hipEventDisableTiming));
  HIPCHK(hipHostMalloc((void **)&c->scan_host, sizeof(unsigned long long) * 2, hipHostMallocDefault));
  HIPCHK(hipMalloc((void **)&c->dilate_flags, sizeof(int) * 16 * NVAR));
  HIPCHK(hipEventCreate(&c->ev0));
  HIPCHK(hipEventCreate(&c->ev1));
  HIPCHK(hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&c->up_done, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->up_dep, hipEventDisableTiming));
  c->dirty = true;
  c->nsrc = 0;
  c->fuse_vadv = -1;
  c->seafloor = ODR_SEAFLOOR_LIFT;
  int rc = flush_world_init(c);
  if (rc) return rc;
  *out = c;
  return 0;
}

int odr_ctx_destroy(odr_ctx *c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamSynchronize(c->up_stream);
  odr_i_phase_dump();   // -DODR_PHASE_TIMING builds only
  for (int s = 0; s < MAXSRC; ++s)
    for (int l = 0; l < MAXLEVELS; ++l) {
      for (void *b : c->block_bufs[s][l]) (void)hipFree(b);
      if (c->staged[s][l].base) (void)hipFree(c->staged[s][l].base);
    }
  for (void *q : c->source_bufs) (void)hipFree(q);
  for (Retired &r : c->graveyard) { (void)hipFree(r.ptr); (void)hipEventDestroy(r.ev); }
  for (void *q : c->registered) if (hipHostUnregister(q) != hipSuccess) (void)hipGetLastError();
  if (c->prep[0]) (void)hipFree(c->prep[0]);
  if (c->prep[1]) (void)hipFree(c->prep[1]);
  (void)hipStreamDestroy(c->up_stream);
  (void)hipEventDestroy(c->up_done);
  (void)hipEventDestroy(c->up_dep);
  (void)hipFree(c->dw);
  for (int k = 0; k < 3; ++k) if (c->hw_pin[k]) { (void)hipHostFree(c->hw_pin[k]); (void)hipEventDestroy(c->hw_ev[k]); }
  (void)hipFree(c->red);
  if (c->red_rec) (void)hipFree(c->red_rec);
  for (double *q : {c->oil_stat, c->oil_cdf, c->oil_chunk, c->oil_part, c->oil_u}) if (q) (void)hipFree(q);
  if (c->oil_guide) (void)hipFree(c->oil_guide);
  if (c->noise_buf) (void)hipFree(c->noise_buf);
  (void)hipFree(c->counter);
  (void)hipHostFree(c->scan_host);
  (void)hipEventDestroy(c->scan_ev);
  (void)hipFree(c->dilate_flags);
  if (c->tile_flags) (void)hipFree(c->tile_flags);
  for (int k = 0; k < 2; ++k) if (c->bounce[k]) (void)hipHostFree(c->bounce[k]);
  if (c->lanes_ready) {
    for (int l = 0; l < ODR_MAX_LANES; ++l) {
      (void)hipStreamDestroy(c->lane_stream[l]);
      (void)hipEventDestroy(c->lane_step[l]);
      (void)hipEventDestroy(c->lane_done[l]);
    }
    (void)hipEventDestroy(c->lane_fork);
  }
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  (void)hipStreamDestroy(c->own_stream);
  delete c;
  return 0;
}

int odr_sync(odr_ctx *c) {
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int odr_set_stream(odr_ctx *c, void *s) {
  HIPCHK(hipStreamSynchronize(c->stream));
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return 0;
}

int odr_timer_begin(odr_ctx *c) {
  HIPCHK(hipEventRecord(c->ev0, c->stream));
  return 0;
}
int odr_timer_end(odr_ctx *c, float *ms) {
  HIPCHK(hipEventRecord(c->ev1, c->stream));
  HIPCHK(hipEventSynchronize(c->ev1));
  HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return 0;
}

// ------------------------------------------------------------------ particles
int odr_particles_create(odr_ctx *c, int64_t capacity, odr_particles **out) {
  REQUIRE(capacity > 0 && out, "capacity must be positive");
  HIPCHK(hipSetDevice(c->device));
  odr_particles *p = new odr_particles();
  memset(p, 0, sizeof(*p));
  p->cap = capacity;
  p->dead_cap = capacity;
  p->scan_kept = -1;
  for (int k = 0; k < 7; ++k) HIPCHK(hipMalloc((void **)&p->d64[k], sizeof(double) * (size_t)capacity));
  for (int k = 0; k < 3; ++k) HIPCHK(hipMalloc((void **)&p->i32[k], sizeof(int) * (size_t)capacity));
  for (int k = 0; k < 4; ++k) HIPCHK(hipMalloc((void **)&p->f32[k], sizeof(float) * (size_t)capacity));
  HIPCHK(hipMalloc((void **)&p->bcount, sizeof(unsigned) * (size_t)(nblk(capacity) + 1)));
  HIPCHK(hipMalloc((void **)&p->wcount, sizeof(unsigned) * (size_t)(nblk(capacity) + 1) * (BLOCK / 64)));
  p->wcount_epoch = ~0ull;
  *out = p;
  return 0;
}

int odr_particles_destroy(odr_ctx *c, odr_particles *p) {
  if (c->red_owner == p) c->red_owner = nullptr;
  if (c->noise_owner == p) c->noise_owner = nullptr;
  if (!p) return 0;
  (void)hipStreamSynchronize(c->stream);
  auto fr = [](void *q) { if (q) (void)hipFree(q); };
  for (int k = 0; k < 7; ++k) { fr(p->d64[k]); fr(p->alt64[k]); }
  for (int k = 0; k < 3; ++k) { fr(p->i32[k]); fr(p->alti32[k]); fr(p->dead64[k]); }
  for (int k = 0; k < 4; ++k) { fr(p->f32[k]); fr(p->altf32[k]); }
  for (int k = 0; k < 2; ++k) fr(p->deadi32[k]);
  for (int k = 0; k < NVAR; ++k) { fr(p->env[k]); fr(p->altenv[k]); }
  for (int k = 0; k < 9; ++k) { fr(p->aux[k]); fr(p->altaux[k]); fr(p->aux_snap[k]); }
  fr(p->bcount);
  fr(p->wcount);
  fr(p->scratch);
  fr(p->rank); fr(p->rank_words); fr(p->rank_before); fr(p->rank_bsum);
  fr(p->wg_tab); fr(p->wg_total); fr(p->wg_list); fr(p->z_keep);
  delete p;
  return 0;
}

template <class T>
static int put(odr_ctx *c, T *dst, const T *src, long long n, T dflt) {
  if (src) {
    H2D(dst, src, sizeof(T) * (size_t)n);
  } else {
    std::vector<T> tmp((size_t)n, dflt);
    H2D(dst, tmp.data(), sizeof(T) * (size_t)n);
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  return 0;
}

int odr_particles_append(odr_ctx *c, odr_particles *p, int64_t n, const double *lon, const double *lat,
                         const double *z, const int32_t *id, const int32_t *moving, const float *wdf,
                         const float *cdf, const float *tv) {
  p->status_epoch++;
  p->epoch++;  // invalidates the cached reductions (reduce())
  REQUIRE(n >= 0 && lon && lat, "lon/lat required");
  if (p->n + n > p->cap) return fail(ODR_ERR_CAPACITY, "capacity %lld exceeded (%lld + %lld)", p->cap, p->n, (long long)n);
  if (n == 0) return 0;
  long long o = p->n;
  int rc;
  if ((rc = put<double>(c, p->d64[0] + o, lon, n, 0.0))) return rc;
  if ((rc = put<double>(c, p->d64[1] + o, lat, n, 0.0))) return rc;
  if ((rc = put<double>(c, p->d64[2] + o, z, n, 0.0))) return rc;
  if ((rc = put<double>(c, p->d64[3] + o, lon, n, 0.0))) return rc;
  if ((rc = put<double>(c, p->d64[4] + o, lat, n, 0.0))) return rc;
  if ((rc = put<double>(c, p->d64[5] + o, lon, n, 0.0))) return rc;
  if ((rc = put<double>(c, p->d64[6] + o, lat, n, 0.0))) return rc;
  if (id) {
    for (long long k = 0; k < n; ++k) {
      REQUIRE(id[k] >= 0, "element IDs must not be negative");
      p->id_max = std::max(p->id_max, (long long)id[k]);
    }
    if ((rc = put<int>(c, p->i32[0] + o, id, n, 0))) return rc;
  } else {
    std::vector<int> ids((size_t)n);
    p->id_max = std::max(p->id_max, o + p->ndead + n - 1);
    for (long long k = 0; k < n; ++k) ids[(size_t)k] = (int)(o + p->ndead + k);
    H2D(p->i32[0] + o, ids.data(), sizeof(int) * (size_t)n);
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  if ((rc = put<int>(c, p->i32[1] + o, nullptr, n, 0))) return rc;
  if ((rc = put<int>(c, p->i32[2] + o, moving, n, 1))) return rc;
  if ((rc = put<float>(c, p->f32[0] + o, wdf, n, 0.02f))) return rc;
  if ((rc = put<float>(c, p->f32[1] + o, cdf, n, 1.0f))) return rc;
  if ((rc = put<float>(c, p->f32[2] + o, tv, n, 0.0f))) return rc;
  if ((rc = put<float>(c, p->f32[3] + o, nullptr, n, 0.0f))) return rc;  // age_seconds = 0
  HIPCHK(hipStreamSynchronize(c->stream));
  p->n += n;
  return 0;
}

int odr_particles_count(odr_ctx *, odr_particles *p, int64_t *na, int64_t *nd) {
  if (na) *na = p->n;
  if (nd) *nd = p->ndead;
  return 0;
}

template <class T>
static int get(odr_ctx *c, T *dst, const T *src, long long n) {
  if (dst && n > 0) D2H(dst, src, sizeof(T) * (size_t)n);
  return 0;
}

int odr_particles_download(odr_ctx *c, odr_particles *p, double *lon, double *lat, double *z, int32_t *id,
                           int32_t *status, int32_t *moving) {
  int rc;
  if ((rc = get(c, lon, p->d64[0], p->n)) || (rc = get(c, lat, p->d64[1], p->n)) ||
      (rc = get(c, z, p->d64[2], p->n)) || (rc = get(c, id, p->i32[0], p->n)) ||
      (rc = get(c, status, p->i32[1], p->n)) || (rc = get(c, moving, p->i32[2], p->n)))
    return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int odr_particles_download_deactivated(odr_ctx *c, odr_particles *p, double *lon, double *lat, double *z,
                                       int32_t *id, int32_t *status) {
  if (p->ndead == 0) return 0;
  int rc;
  if ((rc = get(c, lon, p->dead64[0], p->ndead)) || (rc = get(c, lat, p->dead64[1], p->ndead)) ||
      (rc = get(c, z, p->dead64[2], p->ndead)) || (rc = get(c, id, p->deadi32[0], p->ndead)) ||
      (rc = get(c, status, p->deadi32[1], p->ndead)))
    return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int odr_particles_upload(odr_ctx *c, odr_particles *p, const double *lon, const double *lat, const double *z,
                         const int32_t *moving, const float *wdf, const float *cdf, const float *tv) {
  p->status_epoch++;
  p->epoch++;  // invalidates the cached reductions (reduce())
  size_t n = (size_t)p->n;
  if (lon) H2D(p->d64[0], lon, 8 * n);
  if (lat) H2D(p->d64[1], lat, 8 * n);
  if (z) H2D(p->d64[2], z, 8 * n);
  if (moving) H2D(p->i32[2], moving, 4 * n);
  if (wdf) H2D(p->f32[0], wdf, 4 * n);
  if (cdf) H2D(p->f32[1], cdf, 4 * n);
  if (tv) H2D(p->f32[2], tv, 4 * n);
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int odr_particles_device_ptr(odr_ctx *c, odr_particles *p, const char *name, void **dptr) {
  p->external = true;  // the caller may write the arrays: never reuse cached reductions
  REQUIRE(name && dptr, "name/dptr NULL");
  static const char *n64[5] = {"lon", "lat", "z", "plon", "plat"};
  static const char *n32[3] = {"id", "status", "moving"};
  for (int k = 0; k < 5; ++k) if (!strcmp(name, n64[k])) { *dptr = p->d64[k]; return 0; }
  for (int k = 0; k < 3; ++k) if (!strcmp(name, n32[k])) { *dptr = p->i32[k]; return 0; }
  if (!strncmp(name, "env:", 4)) {
    int v = atoi(name + 4);
    REQUIRE(v >= 0 && v < NVAR, "bad variable id");
    int rc = ensure_env(c, p, v);
    if (rc) return rc;
    p->env_cok[v] = false;
    *dptr = p->env[v];
    return 0;
  }
  return fail(ODR_ERR_INVALID, "unknown array '%s'", name);
}

// float32 element properties of the active set (LagrangianArray variables, elements/elements.py:71-88)
int odr_particles_download_f32(odr_ctx *c, odr_particles *p, const char *name, float *host) {
  REQUIRE(name && host, "name/host NULL");
  static const char *nf[4] = {"wind_drift_factor", "current_drift_factor", "terminal_velocity", "age_seconds"};
  for (int k = 0; k < 4; ++k)
    if (!strcmp(name, nf[k])) {
      if (p->n) D2H(host, p->f32[k], sizeof(float) * (size_t)p->n);
      HIPCHK(hipStreamSynchronize(c->stream));
      return 0;
    }
  return fail(ODR_ERR_INVALID, "unknown property '%s'", name);
}

// --------------------------------------------------------------------- sources
static void proj_init(DevProj &p, const odr_proj_desc *d) {
  memset(&p, 0, sizeof p);
  if (!d) { p.kind = PROJ_LATLONG; p.a = 1; p.k0 = 1; return; }
  p.kind = d->kind;
  p.a = d->a;
  p.es = d->es;
  p.e = sqrt(d->es);
  p.lon0 = d->lon0_deg * kDeg;
  p.lat0 = d->lat0_deg * kDeg;
  p.x0 = d->x0;
  p.y0 = d->y0;
  p.k0 = d->k0;
  p.south = d->lat0_deg < 0;
  p.akm1 = 2 * d->k0;
  {
    double e2 = d->es, e4 = e2 * e2, e6 = e4 * e2, e8 = e4 * e4;  // Snyder eq. 3-5
    p.cchi[0] = e2 / 2 + 5 * e4 / 24 + e6 / 12 + 13 * e8 / 360;
    p.cchi[1] = 7 * e4 / 48 + 29 * e6 / 240 + 811 * e8 / 11520;
    p.cchi[2] = 7 * e6 / 120 + 81 * e8 / 1120;
    p.cchi[3] = 4279 * e8 / 161280;
  }
  if (d->kind == PROJ_MERC || d->kind == PROJ_LCC) {
    // PROJ's merc / lcc set-up: k0 from +lat_ts (merc); cone constant, F and rho0 from the standard parallels (lcc)
    auto msfn = [&](double phi) { double sp = sin(phi); return cos(phi) / sqrt(1 - d->es * sp * sp); };
    auto ts = [&](double phi) { double es = p.e * sin(phi); return tan(0.5 * (kHalfPi - phi)) / pow((1 - es) / (1 + es), 0.5 * p.e); };
    if (d->kind == PROJ_MERC) {
      if (d->lat_ts_deg != 0) p.k0 = msfn(fabs(d->lat_ts_deg) * kDeg);
    } else {
      const double phi1 = d->lat1_deg * kDeg, phi2 = d->lat2_deg * kDeg;
      double n = sin(phi1);
      const double m1 = msfn(phi1), t1 = ts(phi1);
      if (fabs(phi1 - phi2) >= 1e-10) n = log(m1 / msfn(phi2)) / log(t1 / ts(phi2));
      p.cn = n;
      p.cc = m1 * pow(t1, -n) / n;
      p.crho0 = fabs(fabs(p.lat0) - kHalfPi) < 1e-10 ? 0.0 : p.cc * pow(ts(p.lat0), n);
    }
  }
  if (d->kind == PROJ_TMERC) {        // Karney 2011: eq. 14 (A), 35 (alpha), 36 (beta); xi of the origin latitude
    const double f = 1 - sqrt(1 - d->es), n = f / (2 - f), n2 = n * n, n3 = n2 * n, n4 = n2 * n2, n5 = n4 * n, n6 = n3 * n3;
    double *al = p.q + 2, *be = p.q + 8;
    p.q[0] = d->k0 / (1 + n) * (1 + n2 * (1.0 / 4 + n2 * (1.0 / 64 + n2 / 256)));
    al[0] = n / 2 - 2 * n2 / 3 + 5 * n3 / 16 + 41 * n4 / 180 - 127 * n5 / 288 + 7891 * n6 / 37800;
    al[1] = 13 * n2 / 48 - 3 * n3 / 5 + 557 * n4 / 1440 + 281 * n5 / 630 - 1983433 * n6 / 1935360;
    al[2] = 61 * n3 / 240 - 103 * n4 / 140 + 15061 * n5 / 26880 + 167603 * n6 / 181440;
    al[3] = 49561 * n4 / 161280 - 179 * n5 / 168 + 6601661 * n6 / 7257600;
    al[4] = 34729 * n5 / 80640 - 3418889 * n6 / 1995840;
    al[5] = 212378941 * n6 / 319334400;
    be[0] = n / 2 - 2 * n2 / 3 + 37 * n3 / 96 - n4 / 360 - 81 * n5 / 512 + 96199 * n6 / 604800;
    be[1] = n2 / 48 + n3 / 15 - 437 * n4 / 1440 + 46 * n5 / 105 - 1118711 * n6 / 3870720;
    be[2] = 17 * n3 / 480 - 37 * n4 / 840 - 209 * n5 / 4480 + 5569 * n6 / 90720;
    be[3] = 4397 * n4 / 161280 - 11 * n5 / 504 - 830251 * n6 / 7257600;
    be[4] = 4583 * n5 / 161280 - 108847 * n6 / 3991680;
    be[5] = 20648693 * n6 / 638668800;
    const double tau = tan(p.lat0), t1 = hypot(1.0, tau), sg = sinh(p.e * atanh(p.e * tau / t1));
    const double xip = atan(hypot(1.0, sg) * tau - sg * t1);
    double xi0 = xip;
    for (int k = 0; k < 6; ++k) xi0 += al[k] * sin(2 * (k + 1) * xip);
    p.q[1] = xi0;
  }
  if (d->kind == PROJ_LAEA) {         // Snyder 24-17..24-20, authalic latitude 3-11 / 3-12
    auto qs = [&](double sp) { if (p.e < 1e-7) return 2 * sp; const double con = p.e * sp; return (1 - d->es) * (sp / (1 - con * con) - (0.5 / p.e) * log((1 - con) / (1 + con))); };
    const double t = fabs(p.lat0);
    p.mode = fabs(t - kHalfPi) < 1e-10 ? (p.lat0 < 0 ? 1 : 0) : (t < 1e-10 ? 2 : 3);
    p.q[7] = sin(p.lat0); p.q[8] = cos(p.lat0);
    if (d->es != 0) {
      const double qp = qs(1.0), rq = sqrt(0.5 * qp);
      p.q[0] = qp; p.q[1] = rq; p.q[2] = 1; p.q[3] = 1; p.q[4] = 1;
      if (p.mode == 2) { p.q[2] = 1 / rq; p.q[3] = 1; p.q[4] = 0.5 * qp; }
      if (p.mode == 3) {
        const double sp = sin(p.lat0), sinb1 = qs(sp) / qp, cosb1 = sqrt(1 - sinb1 * sinb1);
        const double dd = cos(p.lat0) / (sqrt(1 - d->es * sp * sp) * rq * cosb1);
        p.q[5] = sinb1; p.q[6] = cosb1; p.q[2] = dd; p.q[4] = rq / dd; p.q[3] = rq * dd;
      }
    }
  }
  if (d->kind == PROJ_STERE_OBLIQUE) {   // Snyder 21-27 with 14-15; conformal latitude of the origin 3-1
    p.mode = fabs(p.lat0) > 1e-10 ? 3 : 2;
    if (d->es != 0) {
      const double sp = sin(p.lat0), X = 2 * atan(tan(0.5 * (kHalfPi + p.lat0)) * pow((1 - sp * p.e) / (1 + sp * p.e), 0.5 * p.e)) - kHalfPi;
      p.akm1 = 2 * d->k0 * cos(p.lat0) / sqrt(1 - d->es * sp * sp);
      p.q[0] = sin(X); p.q[1] = cos(X);
    } else {
      p.akm1 = 2 * d->k0;
      p.q[0] = sin(p.lat0); p.q[1] = cos(p.lat0);
    }
  }
  if (d->kind == PROJ_OB_TRAN) {      // lat1 = o_lat_p, lat2 = o_lon_p (include/odrift.h); a sphere of unit radius, no offsets
    p.q[0] = sin(d->lat1_deg * kDeg); p.q[1] = cos(d->lat1_deg * kDeg); p.q[2] = d->lat2_deg * kDeg;
    p.a = 1; p.es = 0; p.e = 0; p.x0 = p.y0 = 0; p.k0 = 1; p.lat0 = 0; p.south = 0;
  }
  if (d->kind == PROJ_STERE_POLAR) {  // Snyder 21-33/21-34 scale constant
    double phits = fabs(d->lat_ts_deg) * kDeg, e = p.e;
    if (d->es == 0) {
      p.akm1 = fabs(phits - kHalfPi) >= 1e-10 ? cos(phits) / tan(0.5 * (kHalfPi - phits)) : 2 * d->k0;
    } else if (fabs(phits - kHalfPi) < 1e-10) {
      p.akm1 = 2 * d->k0 / sqrt(pow(1 + e, 1 + e) * pow(1 - e, 1 - e));
    } else {
      double t = sin(phits), es = e * t;
      double ts = tan(0.5 * (kHalfPi - phits)) / pow((1 - es) / (1 + es), 0.5 * e);
      p.akm1 = cos(phits) / ts;
      t *= e;
      p.akm1 /= sqrt(1 - t * t);
    }
  }
}

static int new_source(odr_ctx *c, int kind, int32_t *sid) {
  int slot = -1;
  for (int k = 0; k < c->nsrc; ++k) if (c->src_free[k]) { slot = k; break; }   // a released source id first (odr_source_release)
  if (slot < 0 && c->nsrc >= MAXSRC) return fail(ODR_ERR_CAPACITY, "at most %d field sources", MAXSRC);
  if (slot >= 0) {
    c->src_free[slot] = false;
    DevSource &r = c->hw.src[slot];
    memset(&r, 0, sizeof r);
    r.kind = kind;
    proj_init(r.proj, nullptr);
    r.xmin = -180; r.xmax = 180; r.ymin = -90; r.ymax = 90;
    r.zmin = -INFINITY; r.zmax = INFINITY;
    r.tmin = -INFINITY; r.tmax = INFINITY;
    r.lon_mode = 1;
    *sid = slot;
    c->dirty = true;
    return 0;
  }
  DevSource &s = c->hw.src[c->nsrc];
  memset(&s, 0, sizeof s);
  s.kind = kind;
  proj_init(s.proj, nullptr);
  s.xmin = -180; s.xmax = 180; s.ymin = -90; s.ymax = 90;
  s.zmin = -INFINITY; s.zmax = INFINITY;
  s.tmin = -INFINITY; s.tmax = INFINITY;
  s.lon_mode = 1;
  *sid = c->nsrc++;
  c->hw.nsrc = c->nsrc;
  c->dirty = true;
  return 0;
}

int odr_source_constant(odr_ctx *c, int nvars, const int32_t *var_ids, const double *values, int32_t *sid) {
  REQUIRE(nvars > 0 && var_ids && values && sid, "bad arguments");
  int rc = new_source(c, SRC_CONSTANT, sid);
  if (rc) return rc;
  DevSource &s = c->hw.src[*sid];
  for (int k = 0; k < nvars; ++k) {
    REQUIRE(var_ids[k] >= 0 && var_ids[k] < NVAR, "bad variable id %d", var_ids[k]);
    s.const_val[var_ids[k]] = values[k];
  }
  return 0;
}

int odr_source_analytic(odr_ctx *c, int kind, const double *params, int nparams, int32_t *sid) {
  REQUIRE(params && nparams >= 4 && nparams <= 8 && sid, "bad arguments");
  REQUIRE(kind == ODR_ANALYTIC_DOUBLE_GYRE || kind == ODR_ANALYTIC_OSCILLATING, "unknown analytic kind %d", kind);
  int rc = new_source(c, kind == ODR_ANALYTIC_DOUBLE_GYRE ? SRC_DOUBLE_GYRE : SRC_OSCILLATING, sid);
  if (rc) return rc;
  DevSource &s = c->hw.src[*sid];
  for (int k = 0; k < nparams; ++k) s.params[k] = params[k];
  if (kind == ODR_ANALYTIC_DOUBLE_GYRE) {
    // reader_double_gyre.py:28-41: +proj=stere +lat_0=0 +lon_0=0 +lat_ts=0 +a=6.371e6 +e=0, x in [0,2], y in [0,1]
    odr_proj_desc d = {ODR_PROJ_STERE_EQUIT_SPHERE, 6.371e6, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0};
    proj_init(s.proj, &d);
    s.xmin = 0; s.xmax = 2; s.ymin = 0; s.ymax = 1;
    s.lon_mode = 2;
  }
  return 0;
}

// Landmask raster source (device image of reader_global_landmask.Reader): cells[iy * nx + ix] != 0 is land
int odr_source_landmask(odr_ctx *c, int32_t nx, int32_t ny, double lon0, double lat0, double dlon, double dlat,
                        const uint8_t *cells, int32_t *sid) {
  REQUIRE(cells && sid && nx > 0 && ny > 0 && dlon > 0 && dlat > 0, "bad raster");
  int rc = new_source(c, SRC_LANDMASK, sid);
  if (rc) return rc;
  DevSource &s = c->hw.src[*sid];
  const size_t nbits = (size_t)nx * (size_t)ny, nwords = (nbits + 31) / 32;
  std::vector<unsigned> words(nwords, 0u);
  for (size_t k = 0; k < nbits; ++k)
    if (cells[k]) words[k >> 5] |= 1u << (k & 31);
  unsigned *d = nullptr;
  HIPCHK(hipMalloc((void **)&d, sizeof(unsigned) * nwords));
  c->source_bufs.push_back(d);
  H2D(d, words.data(), sizeof(unsigned) * nwords);
  s.params[0] = lon0; s.params[1] = lat0; s.params[2] = dlon; s.params[3] = dlat;
  s.slot[0].nx = nx; s.slot[0].ny = ny;
  s.slot[0].data[VAR_LAND] = (const float *)d;
  s.always_valid = 1;
  return 0;
}

int odr_source_grid(odr_ctx *c, const odr_proj_desc *proj, const double *dom, int lon_mode, int mod360_x,
                    int nz, const double *z, int32_t *sid) {
  REQUIRE(dom && sid, "bad arguments");
  REQUIRE(nz <= MAXNZ, "at most %d z levels", MAXNZ);
  int rc = new_source(c, SRC_GRID, sid);
  if (rc) return rc;
  DevSource &s = c->hw.src[*sid];
  proj_init(s.proj, proj);
  s.xmin = dom[0]; s.xmax = dom[1]; s.ymin = dom[2]; s.ymax = dom[3]; s.zmin = dom[4]; s.zmax = dom[5];
  s.lon_mode = lon_mode;
  s.mod360_x = mod360_x;
  s.nz = (nz > 1 && z) ? nz : 1;
  for (int k = 0; k < s.nz && z; ++k) s.z[k] = z[k];
  for (int k = 0; k + 1 < s.nz && z; ++k) s.zmid[k] = -z[k] + 0.5 * (-z[k + 1] - (-z[k]));
  if (s.nz > 1) {  // np.gradient(Kprofiles, mixing_z, axis=0) constants for k_vmix
    const int n = s.nz;
    s.vg_d[0] = z[1] - z[0]; s.vg_d[1] = z[n - 1] - z[n - 2]; s.vg_d[2] = 2. * (z[1] - z[0]);
    for (int k = 0; k < 3; ++k) s.vg_id[k] = 1.0 / s.vg_d[k];
    s.vg_uniform = 1;
    for (int k = 1; k < n - 1; ++k)
      if ((z[k + 1] - z[k]) != (z[1] - z[0])) s.vg_uniform = 0;
    for (int k = 1; k < n - 1; ++k) {
      double dx1 = -(-z[k] - (-z[k - 1])), dx2 = -(-z[k + 1] - (-z[k]));
      s.vg_a[k] = -dx2 / (dx1 * (dx1 + dx2));
      s.vg_b[k] = (dx2 - dx1) / (dx1 * dx2);
      s.vg_c[k] = dx1 / (dx2 * (dx1 + dx2));
    }
  }
  if (s.nz > 1) {  // interp1d(zgrid, range(nz)) tables (Linear1DInterpolator, interpolators.py:174-197)
    const bool asc = z[1] > z[0];
    const int n = s.nz;
    for (int k = 0; k < n; ++k) s.zasc[k] = asc ? z[k] : z[n - 1 - k];
    for (int k = n; k < MAXNZ; ++k) s.zasc[k] = INFINITY;   // zinterp counts the levels below z four at a time: the padding never counts
    for (int lo = 0; lo + 1 < n; ++lo) {
      double yl = asc ? lo : n - 1 - lo, yh = asc ? lo + 1 : n - 2 - lo;
      s.zi_x[lo] = s.zasc[lo];
      s.zi_slope[lo] = (yh - yl) / (s.zasc[lo + 1] - s.zasc[lo]);
      s.zi_y[lo] = yl;
    }
  }
  return 0;
}

// StructuredReader without a projection (structured.py:44-113): the Delaunay triangulation of the (lon, lat) nodes
// is prepared on the host (odr_mesh.h: cell diagonals + Lawson flips) and kept in HBM for curvi_locate.
int odr_source_grid_curvilinear(odr_ctx *c, const double *lon, const double *lat, int ny, int nx,
                                const double *dom, int lon_mode, int nz, const double *z, int32_t *sid) {
  REQUIRE(lon && lat && dom && sid, "bad arguments");
  odr_mesh::Mesh m;
  if (!odr_mesh::build(m, lon, lat, ny, nx)) return fail(ODR_ERR_INVALID, "curvilinear grid: %s", m.error.c_str());
  int rc = odr_source_grid(c, nullptr, dom, lon_mode, 0, nz, z, sid);
  if (rc) return rc;
  void *dn = nullptr, *dv = nullptr, *dt = nullptr, *db = nullptr;
  HIPCHK(hipMalloc(&dn, m.nodes.size() * sizeof(double)));
  HIPCHK(hipMalloc(&dv, m.tri_v.size() * sizeof(int32_t)));
  HIPCHK(hipMalloc(&dt, m.tri_n.size() * sizeof(int32_t)));
  HIPCHK(hipMalloc(&db, m.bucket.size() * sizeof(int32_t)));
  c->source_bufs.push_back(dn); c->source_bufs.push_back(dv); c->source_bufs.push_back(dt); c->source_bufs.push_back(db);
  H2D(dn, m.nodes.data(), m.nodes.size() * sizeof(double));
  H2D(dv, m.tri_v.data(), m.tri_v.size() * sizeof(int32_t));
  H2D(dt, m.tri_n.data(), m.tri_n.size() * sizeof(int32_t));
  H2D(db, m.bucket.data(), m.bucket.size() * sizeof(int32_t));
  DevProj &p = c->hw.src[*sid].proj;
  p.kind = PROJ_CURVILINEAR;
  p.cv_nodes = (const D2 *)dn; p.cv_tri_v = (const int *)dv; p.cv_tri_n = (const int *)dt; p.cv_bucket = (const int *)db;
  p.cv_nx = nx; p.cv_maxit = 4 * (nx + ny) + 16; p.cv_nbx = m.nbx; p.cv_nby = m.nby;
  p.cv_bx0 = m.bx0; p.cv_by0 = m.by0; p.cv_ibx = m.ibx; p.cv_iby = m.iby;
  c->dirty = true;
  return 0;
}

int odr_source_lonlat2xy(odr_ctx *c, int32_t sid, int64_t n, const double *lon, const double *lat, double *x, double *y) {
  REQUIRE(sid >= 0 && sid < c->nsrc, "bad source id %d", sid);
  REQUIRE(n >= 0 && (n == 0 || (lon && lat && x && y)), "bad arguments");
  if (n == 0) return 0;
  int rc = flush_world(c);
  if (rc) return rc;
  double *d = nullptr;
  HIPCHK(hipMalloc((void **)&d, 4 * sizeof(double) * (size_t)n));
  H2D(d, lon, 8 * (size_t)n);
  H2D(d + n, lat, 8 * (size_t)n);
  hipLaunchKernelGGL(k_lonlat2xy, dim3(nblk(n)), dim3(BLOCK), 0, c->stream, c->dw, sid, (long long)n, d, d + n, d + 2 * n, d + 3 * n);
  HIPCHK(hipGetLastError());
  D2H(x, d + 2 * n, 8 * (size_t)n);
  D2H(y, d + 3 * n, 8 * (size_t)n);
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipFree(d));
  return 0;
}

// The device image of the world follows the host image WITHOUT a host synchronisation and without the copy engine: `hw` is
// copied into one of three page-locked snapshots (hw may change right after), and a small kernel on the compute
// stream moves the snapshot into `dw`.  (Rounds 1-4: hipMemcpyAsync from the pageable `hw` + hipStreamSynchronize.  In a
// run() whose reader levels are prefetched on the upload stream that copy queued behind the 100 MB transfer of the level
// staged just before it: the first level change of a run cost the host -- and the device -- 7 ms, profiles/r05_ab_variants.txt 6.
// Anything the runtime sets up at a first use -- a page-locked allocation, the first launch of a kernel -- blocked the same way
// when it fell behind a staged level: the snapshots are allocated and the kernel is launched once when the context is made.)
__global__ void k_world_copy(uint2 *__restrict__ dst, const uint2 *__restrict__ src, unsigned nwords) {
  for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < nwords; k += gridDim.x * blockDim.x) dst[k] = src[k];
}
int flush_world(odr_ctx *c) {
  if (c->dirty) {
    static_assert(sizeof(DevWorld) % sizeof(uint2) == 0, "DevWorld is copied in 8-byte words");
    const int k = c->hw_turn;
    {
      SlowSpan sp("flush_world: hipEventSynchronize");
      HIPCHK(hipEventSynchronize(c->hw_ev[k]));   // (the flush before the flush before this one: long done)
    }
    memcpy(c->hw_pin[k], &c->hw, sizeof(DevWorld));
    SlowSpan sp("flush_world: launch + event record");
    constexpr unsigned nwords = (unsigned)(sizeof(DevWorld) / sizeof(uint2));   // one 8-byte word per thread: one round trip over PCIe
    hipLaunchKernelGGL(k_world_copy, dim3((nwords + 1023) / 1024), dim3(1024), 0, c->stream, (uint2 *)c->dw, (const uint2 *)c->hw_pin[k], nwords);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->hw_ev[k], c->stream));
    c->hw_turn = (k + 1) % 3;
    c->dirty = false;
  }
  return 0;
}

int flush_world_init(odr_ctx *c) {
  for (int k = 0; k < 3; ++k) {
    HIPCHK(hipHostMalloc((void **)&c->hw_pin[k], sizeof(DevWorld), hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&c->hw_ev[k], hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->hw_ev[k], c->stream));
  }
  c->dirty = true;
  return flush_world(c);
}

static void sort_levels(DevSource &s) {
  s.nlevels = 0;
  for (int l = 0; l < MAXLEVELS; ++l) if (s.slot[l].valid) s.level_slot[s.nlevels++] = l;
  for (int a = 1; a < s.nlevels; ++a)
    for (int b = a; b > 0 && s.slot[s.level_slot[b]].t < s.slot[s.level_slot[b - 1]].t; --b) {
      int t = s.level_slot[b]; s.level_slot[b] = s.level_slot[b - 1]; s.level_slot[b - 1] = t;
    }
}

// ---- field block upload: a stream-ordered pipeline on the context's UPLOAD stream ----
// copy (DMA when the source is pinned / registered or device memory) -> mask -> fill towards the seafloor ->
// 10 NaN dilation sweeps -> transposition into node records, one variable after the other through two pooled
// scratch buffers.  odr_block_upload_async only enqueues (it returns while the copy engine and the kernels
// are still working, overlapping with the simulation on the compute stream); the staged block becomes the
// content of its slot with odr_block_commit, where the compute stream waits for the upload's event.  The
// block it replaces is freed (or recycled) once the compute stream has passed the commit.
static void reap(odr_ctx *c, size_t want_bytes, float **reuse) {
  for (size_t k = 0; k < c->graveyard.size();) {
    Retired &r = c->graveyard[k];
    if (hipEventQuery(r.ev) == hipSuccess) {
      if (reuse && !*reuse && r.bytes == want_bytes) *reuse = (float *)r.ptr;
      else { SlowSpan sp("reap: hipFree"); (void)hipFree(r.ptr); }
      (void)hipEventDestroy(r.ev);
      c->graveyard.erase(c->graveyard.begin() + (long)k);
    } else ++k;
  }
}

static int retire(odr_ctx *c, void *ptr, size_t bytes) {
  Retired r;
  r.ptr = ptr; r.bytes = bytes;
  HIPCHK(hipEventCreateWithFlags(&r.ev, hipEventDisableTiming));
  HIPCHK(hipEventRecord(r.ev, c->stream));   // the compute stream may still read it up to here
  c->graveyard.push_back(r);
  return 0;
}

// bcast_root >= 0 (odr_block_broadcast): the level's arrays are read on that rank only and reach the others by ONE broadcast of
// the staging memory they are copied into side by side
static int stage_block(odr_ctx *c, int32_t sid, int32_t slot, double t_epoch, int nvars, const int32_t *var_ids,
                       const void *const *data, const int32_t *var_nz, int ny, int nx, const double *xy8, int bcast_root = -1) {
  REQUIRE(sid >= 0 && sid < c->nsrc && c->hw.src[sid].kind == SRC_GRID, "source %d is not a grid source", sid);
  REQUIRE(slot >= 0 && slot < MAXLEVELS, "slot must be in [0,%d)", MAXLEVELS);
  const bool have_data = bcast_root < 0 || odr_i_comm_rank() == bcast_root;
  REQUIRE(nvars > 0 && var_ids && (data || !have_data) && var_nz && xy8 && ny > 1 && nx > 1, "bad block arguments");
  HIPCHK(hipSetDevice(c->device));
  const DevSource &s = c->hw.src[sid];
  Staged &S = c->staged[sid][slot];
  if (S.base) {  // an earlier staging of this slot that was never committed
    HIPCHK(hipStreamSynchronize(c->up_stream));
    HIPCHK(hipFree(S.base));
    S.base = nullptr;
  }
  DevBlock &b = S.blk;
  memset(&b, 0, sizeof b);
  memset(S.cid, 0, sizeof S.cid);
  b.ny = ny; b.nx = nx; b.valid = 1;
  b.x0 = xy8[0]; b.xspan = xy8[1]; b.y0 = xy8[2]; b.yspan = xy8[3];
  b.xmin = xy8[4]; b.xrange = xy8[5]; b.ymin = xy8[6]; b.yrange = xy8[7];
  b.ixspan = 1.0 / b.xspan; b.iyspan = 1.0 / b.yspan; b.ixrange = 1.0 / b.xrange; b.iyrange = 1.0 / b.yrange;
  b.t = t_epoch;
  const size_t plane = (size_t)ny * nx;
  size_t nmax = 0, ntot = 0, nlayers = 0;
  for (int k = 0; k < nvars; ++k) {
    int v = var_ids[k], nzv = var_nz[k] > 1 ? var_nz[k] : 1;
    REQUIRE(v >= 0 && v < NVAR, "bad variable id %d", v);
    const int mem = s.members[v] > 1 ? s.members[v] : 1;
    REQUIRE(nzv == mem || nzv == mem * s.nz, "variable %d has %d layers, source has %d levels x %d members", v, nzv, s.nz, mem);
    nmax = std::max(nmax, plane * (size_t)nzv);
    ntot += plane * (size_t)nzv;
    nlayers += (size_t)nzv;
  }
  nmax = std::max(nmax, ntot);   // the whole level is staged at once (prep_all below)
  // record layout: interleaved vector pairs first, then the other 3D variables, then the 2D ones; the record
  // length is padded to 16 bytes (odr_field.hip.h DevBlock)
  static const int pairs[4][2] = {{VAR_U, VAR_V}, {VAR_XWIND, VAR_YWIND}, {VAR_SX, VAR_SY}, {VAR_ICE_U, VAR_ICE_V}};
  std::vector<int> off((size_t)nvars, -1), es((size_t)nvars, 1), eo((size_t)nvars, 0);
  int rec = 0;
  for (int pr = 0; pr < 4; ++pr) {
    int ka = -1, kb = -1;
    for (int k = 0; k < nvars; ++k) { if (var_ids[k] == pairs[pr][0]) ka = k; if (var_ids[k] == pairs[pr][1]) kb = k; }
    if (ka < 0 || kb < 0) continue;
    int nza = var_nz[ka] > 1 ? var_nz[ka] : 1, nzb = var_nz[kb] > 1 ? var_nz[kb] : 1;
    if (nza != nzb) continue;
    off[(size_t)ka] = rec; off[(size_t)kb] = rec; es[(size_t)ka] = es[(size_t)kb] = 2; eo[(size_t)kb] = 1;
    rec += 2 * nza;
  }
  for (int pass = 0; pass < 2; ++pass)   // 3D scalars, then 2D scalars
    for (int k = 0; k < nvars; ++k) {
      int nzv = var_nz[k] > 1 ? var_nz[k] : 1;
      if (off[(size_t)k] >= 0 || (pass == 0) != (nzv > 1)) continue;
      off[(size_t)k] = rec;
      rec += nzv;
    }
  rec = (rec + 3) & ~3;
  const size_t base_bytes = sizeof(float) * plane * (size_t)rec + 64;
  float *base = nullptr;
  reap(c, base_bytes, &base);   // recycle a retired block of the same size if the compute stream is done with it
  if (!base) { SlowSpan sp("stage_block: hipMalloc of the block"); HIPCHK(hipMalloc((void **)&base, base_bytes)); }
  SlowSpan sp_rest("stage_block: everything after the block's allocation");
  if (c->prep_floats < nmax) {  // pooled scratch: one variable in flight + the dilation ping-pong buffer
    HIPCHK(hipStreamSynchronize(c->up_stream));
    if (c->prep[0]) HIPCHK(hipFree(c->prep[0]));
    if (c->prep[1]) HIPCHK(hipFree(c->prep[1]));
    c->prep[0] = c->prep[1] = nullptr;
    HIPCHK(hipMalloc((void **)&c->prep[0], sizeof(float) * nmax));
    HIPCHK(hipMalloc((void **)&c->prep[1], sizeof(float) * nmax));
    c->prep_floats = nmax;
  }
  hipStream_t st = c->up_stream;
  // device-resident sources (odr_sgrid_zslice results) may still be in the making on the compute stream: only
  // then does the upload depend on it (host sources must NOT wait for the simulation's backlog -- that is the overlap)
  bool dev_src = false;
  for (int k = 0; k < nvars && !dev_src && have_data; ++k) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, data[k]) == hipSuccess) dev_src = at.type == hipMemoryTypeDevice;
    else (void)hipGetLastError();   // plain pageable memory: not an error
  }
  if (dev_src) {
    HIPCHK(hipEventRecord(c->up_dep, c->stream));
    HIPCHK(hipStreamWaitEvent(st, c->up_dep, 0));
  }
  // Preparation of the whole level in three launches (k_blk_mask_fill / k_blk_dilate_tile / k_blk_records): every variable
  // staged side by side in the scratch pool, mask + sea-floor fill with a NaN flag per (layer, tile), the ten dilation
  // sweeps on flagged tiles only, one record writer that assembles complete node records.  ODR_ROW_DILATE /
  // ODR_PLAIN_DILATE: the per-variable whole-array sweeps of rounds 1-2 (the cross-check of tests/test_gpu_async_upload.py).
  const bool prep_all = !getenv("ODR_PLAIN_DILATE") && !getenv("ODR_ROW_DILATE") && ny < 65536 && nlayers < 65536 && nvars <= NVAR;
  if (bcast_root >= 0 && !prep_all) return fail(ODR_ERR_INVALID, "odr_block_broadcast: the level does not fit the one-pass preparation");
  if (prep_all) {
    BlkPrep Q;
    memset(&Q, 0, sizeof Q);
    Q.nvars = nvars; Q.ny = ny; Q.nx = nx; Q.rec = rec;
    Q.tiles_x = (nx + DIL_T - 1) / DIL_T;
    Q.ntiles = Q.tiles_x * ((ny + DIL_T - 1) / DIL_T);
    const size_t nflags = nlayers * (size_t)Q.ntiles;
    if (c->tile_flags_n < nflags) {
      HIPCHK(hipStreamSynchronize(st));
      if (c->tile_flags) HIPCHK(hipFree(c->tile_flags));
      c->tile_flags = nullptr;
      HIPCHK(hipMalloc((void **)&c->tile_flags, sizeof(int) * nflags));
      c->tile_flags_n = nflags;
    }
    HIPCHK(hipMemsetAsync(c->tile_flags, 0, sizeof(int) * nflags, st));
    HIPCHK(hipMemsetAsync((char *)base + base_bytes - 64, 0, 64, st));   // the spare bytes wide slot loads may touch
    size_t at_f = 0;
    int cum = 0, ndil = 0;
    for (int k = 0; k < nvars; ++k) {
      const int v = var_ids[k], nzv = var_nz[k] > 1 ? var_nz[k] : 1;
      const size_t n = plane * (size_t)nzv;
      float *buf = c->prep[0] + at_f;
      hipPointerAttribute_t at;
      bool pageable = false;
      if (!have_data) {}
      else if (hipPointerGetAttributes(&at, data[k]) != hipSuccess) { (void)hipGetLastError(); pageable = true; }
      else pageable = at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeHost && at.type != hipMemoryTypeManaged;
      if (!have_data) {}      // (arrives with the broadcast below)
      else if (pageable) { int rcb = odr_i_h2d(c, buf, data[k], sizeof(float) * n, st, 1); if (rcb) return rcb; }
      else HIPCHK(hipMemcpyAsync(buf, data[k], sizeof(float) * n, hipMemcpyDefault, st));
      Q.src[k] = buf; Q.fix[k] = c->prep[1] + at_f;
      Q.nz[k] = nzv; Q.cum[k] = cum; Q.off[k] = off[(size_t)k]; Q.es[k] = es[(size_t)k]; Q.eo[k] = eo[(size_t)k];
      Q.fill[k] = nzv > 1 && s.members[v] <= 1;     // ("Ensemble data currently not extrapolated towards seafloor", structured.py:58-60)
      Q.dil[k] = v != VAR_LAND;
      ndil += Q.dil[k];
      at_f += n; cum += nzv;
      b.data[v] = base + off[(size_t)k] + eo[(size_t)k];
      b.es[v] = es[(size_t)k];
      b.var_nz[v] = nzv;
    }
    Q.cum[nvars] = cum;
    if (bcast_root >= 0) {   // every variable of the level in one collective, in place, behind root's copies on this stream
      int rcb = odr_i_comm_bcast_floats(c->prep[0], at_f, bcast_root, st);
      if (rcb) return rcb;
    }
    hipLaunchKernelGGL(k_blk_mask_fill, dim3((unsigned)((nx + BLOCK - 1) / BLOCK), (unsigned)ny, (unsigned)nvars), dim3(BLOCK), 0, st,
                       Q, c->tile_flags);
    if (ndil) hipLaunchKernelGGL(k_blk_dilate_tile, dim3((unsigned)Q.ntiles, (unsigned)cum), dim3(BLOCK), 0, st, Q,
                                 (const int *)c->tile_flags);
    hipLaunchKernelGGL(k_blk_records, dim3((unsigned)((plane + 63) / 64)), dim3(BLOCK), sizeof(float) * 64 * (size_t)((rec < REC_CH ? rec : REC_CH) | 1), st,
                       Q, base, plane);
  } else
  HIPCHK(hipMemsetAsync(base, 0, base_bytes, st));
  const unsigned gp = (unsigned)((plane + BLOCK - 1) / BLOCK);
  for (int k = 0; k < nvars && !prep_all; ++k) {
    const int v = var_ids[k], nzv = var_nz[k] > 1 ? var_nz[k] : 1;
    const size_t n = plane * (size_t)nzv;
    float *buf = c->prep[0], *tmp = c->prep[1];
    // data[k]: device memory, page-locked host memory (odr_host_register / hipHostMalloc: a true asynchronous DMA), or
    // pageable host memory -- the latter through the upload stream's bounce buffer (see odr_i_h2d), which makes this
    // variable's copy synchronous for the host
    {
      hipPointerAttribute_t at;
      bool pageable = false;
      if (hipPointerGetAttributes(&at, data[k]) != hipSuccess) { (void)hipGetLastError(); pageable = true; }
      else pageable = at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeHost && at.type != hipMemoryTypeManaged;
      if (pageable) { int rcb = odr_i_h2d(c, buf, data[k], sizeof(float) * n, st, 1); if (rcb) return rcb; }
      else HIPCHK(hipMemcpyAsync(buf, data[k], sizeof(float) * n, hipMemcpyDefault, st));
    }
    const unsigned g = (unsigned)((n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(k_blk_mask, dim3(g), dim3(BLOCK), 0, st, buf, n);
    // ("Ensemble data currently not extrapolated towards seafloor", readers/interpolation/structured.py:58-60)
    if (nzv > 1 && s.members[v] <= 1) hipLaunchKernelGGL(k_blk_fill_seafloor, dim3(gp), dim3(BLOCK), 0, st, buf, nzv, plane);
    if (v != VAR_LAND) {
      // the reference dilates on demand, <=10 sweeps per interpolator call (interpolators.py:127-137);
      // 10 sweeps up front give identical samples.  (An LDS-tiled single-pass version of the
      // ten sweeps was measured slower than these ten bandwidth-bound launches -- 0.67 vs 0.41 ms -- and dropped.)
      // Sweeps after the first only look at the cells that are still NaN in their ping-pong target (= the state two
      // sweeps back; k_blk_dilate_row): one read of the target where the field has no NaN, instead of read + write.
      float *a = buf, *bb2 = tmp;
      const unsigned rows = (unsigned)nzv * (unsigned)ny;
      const bool by_row = !getenv("ODR_PLAIN_DILATE") && (size_t)nzv * (size_t)ny < (1ull << 31);
      // flags[it + 1] = sweep `it` gave some cell a value; a sweep that sees 0 from its predecessor returns at once
      int *flags = c->dilate_flags + 16 * k;
      if (by_row) HIPCHK(hipMemsetAsync(flags, 0, 16 * sizeof(int), st));
      for (int it = 0; it < 10; ++it) {
#define DILATE_ROW(W)                                                                                                   \
  do {                                                                                                                  \
    if (it == 0) hipLaunchKernelGGL((k_blk_dilate_row<W, true>), dim3(rows), dim3(BLOCK), 0, st, a, bb2, ny, nx, flags); \
    else hipLaunchKernelGGL((k_blk_dilate_row<W, false>), dim3(rows), dim3(BLOCK), 0, st, a, bb2, ny, nx, flags + it);    \
  } while (0)
        if (!by_row) hipLaunchKernelGGL(k_blk_dilate, dim3(g), dim3(BLOCK), 0, st, a, bb2, nzv, ny, nx);
        else if (nx % 4 == 0) DILATE_ROW(4);
        else if (nx % 2 == 0) DILATE_ROW(2);
        else DILATE_ROW(1);
#undef DILATE_ROW
        float *t2 = a; a = bb2; bb2 = t2;
      }
      // 10 swaps -> result is back in buf
    }
    const float *fin = buf;
    hipLaunchKernelGGL(k_blk_to_record, dim3((unsigned)((plane + 63) / 64)), dim3(BLOCK), 0, st, fin, base, nzv, plane, rec,
                       off[(size_t)k], es[(size_t)k], eo[(size_t)k], (const float *)nullptr);
    b.data[v] = base + off[(size_t)k] + eo[(size_t)k];
    b.es[v] = es[(size_t)k];
    b.var_nz[v] = nzv;
  }
  HIPCHK(hipGetLastError());
  b.base = base;
  b.rec = rec;
  b.small = plane < (1u << 24) && (double)plane * rec * 4.0 < 4294967296.0 && rec * 4 < (1 << 24);
  S.base = base;
  S.bytes = base_bytes;
  HIPCHK(hipEventRecord(c->up_done, st));
  return 0;
}

int odr_block_commit(odr_ctx *c, int32_t sid, int32_t slot) {
  REQUIRE(sid >= 0 && sid < c->nsrc && slot >= 0 && slot < MAXLEVELS, "bad source/slot");
  Staged &S = c->staged[sid][slot];
  if (!S.base) return fail(ODR_ERR_STATE, "no staged block for source %d slot %d", sid, slot);
  HIPCHK(hipSetDevice(c->device));
  // The caller may release (or overwrite) the host arrays of the staged level after this call: wait until the upload
  // pipeline has consumed them.  (A copy from pageable memory is pinned on the fly and read by the copy engine LATER:
  // arrays freed at commit time while it was still in flight showed up as a sporadic "Memory access fault by GPU" in
  // whatever ran next.)  Normally the upload finished a reader period ago and this returns at once.
  { SlowSpan sp("odr_block_commit: hipEventSynchronize(upload done)"); HIPCHK(hipEventSynchronize(c->up_done)); }
  // the compute stream must not read the new records before the upload pipeline has written them
  HIPCHK(hipStreamWaitEvent(c->stream, c->up_done, 0));
  DevSource &s = c->hw.src[sid];
  int rc;
  for (size_t k = 0; k < c->block_bufs[sid][slot].size(); ++k)
    if ((rc = retire(c, c->block_bufs[sid][slot][k], c->block_bytes[sid][slot]))) return rc;
  c->block_bufs[sid][slot].clear();
  s.slot[slot] = S.blk;
  memcpy(c->block_cid[sid][slot], S.cid, sizeof S.cid);
  c->block_bufs[sid][slot].push_back(S.base);
  c->block_bytes[sid][slot] = S.bytes;
  S.base = nullptr;
  sort_levels(s);
  c->dirty = true;
  return 0;
}

int odr_block_set_content_ids(odr_ctx *c, int32_t sid, int32_t slot, int nvars, const int32_t *var_ids, const uint64_t *ids) {
  REQUIRE(sid >= 0 && sid < c->nsrc && slot >= 0 && slot < MAXLEVELS, "bad source/slot");
  REQUIRE(nvars >= 0 && (nvars == 0 || (var_ids && ids)), "bad arguments");
  Staged &S = c->staged[sid][slot];
  unsigned long long *dst = S.base ? S.cid : c->block_cid[sid][slot];   // the staged level if there is one, else the resident one
  for (int k = 0; k < nvars; ++k) {
    REQUIRE(var_ids[k] >= 0 && var_ids[k] < NVAR, "bad variable id %d", var_ids[k]);
    dst[var_ids[k]] = ids[k];
  }
  return 0;
}

static int block_upload(odr_ctx *c, int32_t sid, int32_t slot, double t_epoch, int nvars, const int32_t *var_ids,
                        const void *const *data, bool on_device, const int32_t *var_nz, int ny, int nx,
                        const double *xy8) {
  (void)on_device;
  int rc = stage_block(c, sid, slot, t_epoch, nvars, var_ids, data, var_nz, ny, nx, xy8);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->up_stream));   // synchronous entry points: the caller may reuse its arrays
  return odr_block_commit(c, sid, slot);
}

int odr_block_upload(odr_ctx *c, int32_t sid, int32_t slot, double t, int nvars, const int32_t *var_ids,
                     const float *const *data, const int32_t *var_nz, int ny, int nx, const double *xy8) {
  return block_upload(c, sid, slot, t, nvars, var_ids, (const void *const *)data, false, var_nz, ny, nx, xy8);
}
int odr_block_upload_device(odr_ctx *c, int32_t sid, int32_t slot, double t, int nvars, const int32_t *var_ids,
                            const void *const *dev_data, const int32_t *var_nz, int ny, int nx, const double *xy8) {
  return block_upload(c, sid, slot, t, nvars, var_ids, dev_data, true, var_nz, ny, nx, xy8);
}
// enqueue only; the arrays must stay valid (and should be pinned: odr_host_register) until odr_block_commit
int odr_block_upload_async(odr_ctx *c, int32_t sid, int32_t slot, double t, int nvars, const int32_t *var_ids,
                           const void *const *data, const int32_t *var_nz, int ny, int nx, const double *xy8) {
  return stage_block(c, sid, slot, t, nvars, var_ids, data, var_nz, ny, nx, xy8);
}
int odr_block_broadcast(odr_ctx *c, int32_t sid, int32_t slot, double t, int nvars, const int32_t *var_ids,
                        const void *const *data, const int32_t *var_nz, int ny, int nx, const double *xy8, int32_t root) {
  if (!odr_i_comm_on()) return fail(ODR_ERR_STATE, "odr_block_broadcast: no communicator (odr_comm_init)");
  REQUIRE(root >= 0, "bad root");
  return stage_block(c, sid, slot, t, nvars, var_ids, data, var_nz, ny, nx, xy8, root);
}
// page-lock caller memory so that uploads from it are DMA transfers that overlap with the simulation
int odr_host_register(odr_ctx *c, void *ptr, uint64_t bytes) {
  REQUIRE(ptr && bytes > 0, "bad host range");
  HIPCHK(hipSetDevice(c->device));
  hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault);
  if (e == hipErrorHostMemoryAlreadyRegistered) {  // pinned by another context / the caller: nothing to do, not ours to unpin
    (void)hipGetLastError();
    return 0;
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();   // do not leave a sticky error for the next launch check
    return fail(ODR_ERR_HIP, "hipHostRegister: %s", hipGetErrorString(e));
  }
  c->registered.push_back(ptr);
  return 0;
}
int odr_host_unregister(odr_ctx *c, void *ptr) {
  REQUIRE(ptr, "NULL pointer");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->up_stream));
  for (size_t k = 0; k < c->registered.size(); ++k)
    if (c->registered[k] == ptr) {
      c->registered.erase(c->registered.begin() + (long)k);
      hipError_t e = hipHostUnregister(ptr);
      if (e != hipSuccess) (void)hipGetLastError();
      return 0;
    }
  return 0;   // not registered by this context
}

int odr_block_drop(odr_ctx *c, int32_t sid, int32_t slot) {
  REQUIRE(sid >= 0 && sid < c->nsrc && slot >= 0 && slot < MAXLEVELS, "bad source/slot");
  // no host synchronisation: the block is freed (or recycled by the next upload of the same size) once the compute
  // stream has passed this point
  int rc;
  for (void *b : c->block_bufs[sid][slot]) if ((rc = retire(c, b, c->block_bytes[sid][slot]))) return rc;
  c->block_bufs[sid][slot].clear();
  if (c->staged[sid][slot].base) {
    HIPCHK(hipStreamSynchronize(c->up_stream));
    HIPCHK(hipFree(c->staged[sid][slot].base));
    c->staged[sid][slot].base = nullptr;
  }
  memset(&c->hw.src[sid].slot[slot], 0, sizeof(DevBlock));
  memset(c->block_cid[sid][slot], 0, sizeof c->block_cid[sid][slot]);
  sort_levels(c->hw.src[sid]);
  c->dirty = true;
  return 0;
}

// A source the caller no longer uses (a gridded reader whose blocks were re-cut to a new window gets a new source): its
// resident blocks are dropped, it leaves every priority list, and its id is handed out again by the next odr_source_*.
int odr_source_release(odr_ctx *c, int32_t sid) {
  REQUIRE(sid >= 0 && sid < c->nsrc && !c->src_free[sid], "unknown source %d", sid);
  int rc;
  for (int slot = 0; slot < MAXLEVELS; ++slot)
    if (!c->block_bufs[sid][slot].empty() || c->staged[sid][slot].base)
      if ((rc = odr_block_drop(c, sid, slot))) return rc;
  for (int v = 0; v < NVAR; ++v) {
    int m = 0;
    for (int k = 0; k < c->hw.nlist[v]; ++k) if (c->hw.list[v][k] != sid) c->hw.list[v][m++] = c->hw.list[v][k];
    c->hw.nlist[v] = m;
  }
  DevSource &s = c->hw.src[sid];
  memset(&s, 0, sizeof s);
  s.kind = SRC_CONSTANT;                 // covers nothing: every const_val is NaN
  for (int v = 0; v < NVAR; ++v) s.const_val[v] = NAN;
  s.xmin = s.ymin = s.zmin = s.tmin = INFINITY; s.xmax = s.ymax = s.zmax = s.tmax = -INFINITY;
  c->src_free[sid] = true;
  c->src_gen[sid]++;
  if (c->red_owner) c->red_owner = nullptr;
  c->dirty = true;
  return 0;
}

int odr_env_bind(odr_ctx *c, int32_t var, int ns, const int32_t *sids, float fallback) {
  REQUIRE(var >= 0 && var < NVAR, "bad variable id %d", var);
  REQUIRE(ns >= 0 && ns <= MAXLIST, "at most %d readers per variable", MAXLIST);
  for (int k = 0; k < ns; ++k) REQUIRE(sids[k] >= 0 && sids[k] < c->nsrc, "unknown source %d", sids[k]);
  c->hw.nlist[var] = ns;
  for (int k = 0; k < ns; ++k) c->hw.list[var][k] = sids[k];
  c->hw.fallback[var] = fallback;
  c->dirty = true;
  return 0;
}

// fast path of odr_env_sample: a group served by one gridded reader with a uniform grid
bool odr_i_build_env_group(const odr_ctx *c, const int *grp, int ng, double t, EnvGroupDesc &G) {
  const DevWorld &w = c->hw;
  if (ng > MAXG || w.nlist[grp[0]] != 1) return false;
  int sid = w.list[grp[0]][0];
  const DevSource &s = w.src[sid];
  if (s.kind != SRC_GRID || s.nlevels < 1) return false;
  if (!s.always_valid && (t < s.tmin || t > s.tmax)) return false;
  for (int k = 0; k < ng; ++k) if (s.members[grp[k]] > 1) return false;   // ensemble data: generic kernels
  const DevBlock &g0 = s.slot[s.level_slot[0]];
  for (int k = 0; k < s.nlevels; ++k) {
    const DevBlock &b = s.slot[s.level_slot[k]];
    if (b.ny != g0.ny || b.nx != g0.nx || b.x0 != g0.x0 || b.xspan != g0.xspan || b.y0 != g0.y0 ||
        b.yspan != g0.yspan || b.xmin != g0.xmin || b.xrange != g0.xrange || b.ymin != g0.ymin || b.yrange != g0.yrange)
      return false;
  }
  int ib, ia;
  host_bracket(s, t, ib, ia);
  memset(&G, 0, sizeof G);
  // order: the caller's, except that the y-component of a vector pair follows its x-component
  static const int pairs[4][2] = {{VAR_U, VAR_V}, {VAR_XWIND, VAR_YWIND}, {VAR_SX, VAR_SY}, {VAR_ICE_U, VAR_ICE_V}};
  auto in_group = [&](int v) { for (int k = 0; k < ng; ++k) if (grp[k] == v) return true; return false; };
  auto pair_y = [&](int v) { for (auto &pr : pairs) if (pr[0] == v) return pr[1]; return -1; };
  auto pair_x = [&](int v) { for (auto &pr : pairs) if (pr[1] == v) return pr[0]; return -1; };
  int ord[NVAR], no = 0;
  for (int k = 0; k < ng; ++k) {
    int v = grp[k];
    if (pair_x(v) >= 0 && in_group(pair_x(v))) continue;  // placed with its x-component
    ord[no++] = v;
    if (pair_y(v) >= 0 && in_group(pair_y(v))) ord[no++] = pair_y(v);
  }
  G.nv = ng; G.sid = sid; G.geo_slot = s.level_slot[0];
  G.all_static = 1;
  const DevBlock &bb = s.slot[ib];
  const DevBlock *ba = (ia >= 0 && !s.always_valid) ? &s.slot[ia] : nullptr;
  if (!bb.small || bb.rec != g0.rec || (ba && (ba->rec != bb.rec || !ba->small))) return false;
  G.bb = bb.base;
  G.ba = ba ? ba->base : nullptr;
  for (int k = 0; k < ng; ++k) {
    int v = ord[k];
    if (!bb.data[v]) return false;
    G.var[k] = v; G.nz[k] = bb.var_nz[v]; G.es[k] = bb.es[v];
    G.off[k] = (int)(bb.data[v] - bb.base);
    if (ba && (!ba->data[v] || ba->var_nz[v] != bb.var_nz[v] || ba->es[v] != bb.es[v] ||
               (int)(ba->data[v] - ba->base) != G.off[k]))
      return false;
    G.fallback[k] = w.fallback[v];
    G.partner[k] = -1;
    if (v == VAR_LAND) G.has_land = 1;
    if (v != VAR_LAND && v != VAR_DEPTH) G.all_static = 0;
  }
  for (int k = 0; k < ng; ++k) {
    int pv = pair_y(G.var[k]);
    if (pv >= 0 && k + 1 < ng && G.var[k + 1] == pv) {
      G.partner[k] = k + 1;
      // interleaved in the node record: one 16-byte load serves both components
      if (G.es[k] == 2 && G.es[k + 1] == 2 && G.off[k + 1] == G.off[k] + 1 && G.nz[k] == G.nz[k + 1]) {
        G.kind[k] = 1; G.kind[k + 1] = 2;
      }
    }
  }
  for (int k = 0; k < MAXG; ++k) G.mode[k] = ENV_SKIP;
  for (int k = 0; k < ng; ++k) {      // gather / arithmetic class of each slot (env_slot_load / env_slot_math)
    if (G.kind[k] == 2) G.mode[k] = ENV_SKIP;
    else if (G.kind[k] == 1) G.mode[k] = G.nz[k] > 1 ? ENV_P3 : ENV_P2;
    else if (G.var[k] == VAR_LAND) G.mode[k] = ENV_LAND;
    else if (G.nz[k] <= 1) G.mode[k] = ENV_S2;
    else G.mode[k] = G.es[k] == 1 ? ENV_S3 : ENV_S3I;
  }
  {   // burst sampler: the group's variables in the physical slots A (<= 16 B per corner), B, C (<= 8 B), D (4 B), L (land)
    for (int q = 0; q < 5; ++q) G.bs[q] = -1;
    G.burst = getenv("ODR_ENV_SERIAL") ? 0 : 1;
    G.rotates = 0;
    auto width = [&](int m) { return (m == ENV_P3 || m == ENV_S3I) ? 4 : (m == ENV_S3 || m == ENV_P2) ? 2 : 1; };
    for (int pass = 4; pass >= 1 && G.burst; pass >>= 1)        // widest first
      for (int k = 0; k < ng && G.burst; ++k) {
        const int m = G.mode[k];
        if (m == ENV_SKIP || m == ENV_LAND || width(m) != pass) continue;
        if (G.partner[k] >= 0 && G.kind[k] != 1) { G.burst = 0; break; }   // a vector pair that is not interleaved: serial sampler
        int q = -1;
        if (G.bs[0] < 0) q = 0;
        else if (pass <= 2 && G.bs[1] < 0) q = 1;
        else if (pass <= 2 && G.bs[2] < 0) q = 2;
        else if (pass == 1 && G.bs[3] < 0) q = 3;
        if (q < 0) { G.burst = 0; break; }
        G.bs[q] = k;
        if (G.partner[k] >= 0) G.rotates = 1;
      }
    for (int k = 0; k < ng && G.burst; ++k) {
      if (G.mode[k] == ENV_LAND) G.bs[4] = k;
      if (G.kind[k] == 0 && G.partner[k] < 0) {   // the y-component of a pair that is not interleaved sits alone: serial sampler
        for (int j = 0; j < ng; ++j) if (G.partner[j] == k && G.kind[j] != 1) G.burst = 0;
      }
    }
    // a width-1 variable may have landed in slot B / C (8-byte slots hold it as ENV_S2): fine; slot D only takes ENV_S2
    G.temp_mask = 0;
    for (int k = 0; k < ng; ++k) if (G.var[k] == VAR_TEMP) G.temp_mask |= 1 << k;
    for (int q = 0; q < 5; ++q) {
      const int k = G.bs[q];
      G.ps_off[q] = k >= 0 ? G.off[k] * 4 : 0;
      G.ps_mode[q] = k >= 0 ? G.mode[k] : ENV_SKIP;
      G.ps_rot[q] = k >= 0 && G.partner[k] >= 0 ? 1 : 0;
      // a 2-D variable whose two bracketing levels carry the same content id (sea floor depth, land mask: the reader hands
      // out the same array every time) is gathered at ONE level; the time interpolation keeps its arithmetic
      if (k >= 0 && ba && ia >= 0 && (G.mode[k] == ENV_S2 || G.mode[k] == ENV_LAND) && !getenv("ODR_NO_STATIC_SKIP")) {
        const unsigned long long cb = c->block_cid[sid][ib][G.var[k]], ca = c->block_cid[sid][ia][G.var[k]];
        if (cb != 0 && cb == ca) G.ps_static |= 1 << q;
      }
    }
  }
  G.w = ia >= 0 ? (t - s.slot[ib].t) / (s.slot[ia].t - s.slot[ib].t) : 0.0;
  return true;
}

static bool launch_env_grid(odr_ctx *c, odr_particles *p, const int *grp, int ng, double t, int rec) {
  EnvGroupDesc G;
  if (!build_env_group(c, grp, ng, t, G)) return false;
  const DevSource &s = c->hw.src[G.sid];
  dim3 g(nblk(p->n)), b(BLOCK);
  PView v = view(p);
  G = env_bind_out(G, v);
  switch (odr_proj_template(s.proj)) {
    case PROJ_LATLONG: hipLaunchKernelGGL(k_env_grid<PROJ_LATLONG>, g, b, 0, c->stream, c->dw, v, G, rec); break;
    case PROJ_STERE_POLAR: hipLaunchKernelGGL(k_env_grid<PROJ_STERE_POLAR>, g, b, 0, c->stream, c->dw, v, G, rec); break;
    case PROJ_CURVILINEAR: hipLaunchKernelGGL(k_env_grid<PROJ_CURVILINEAR>, g, b, 0, c->stream, c->dw, v, G, rec); break;
    case PROJ_EXT: hipLaunchKernelGGL(k_env_grid<PROJ_EXT>, g, b, 0, c->stream, c->dw, v, G, rec); break;
    default: hipLaunchKernelGGL(k_env_grid<PROJ_STERE_EQUIT_SPHERE>, g, b, 0, c->stream, c->dw, v, G, rec); break;
  }
  return true;
}

// every element gets the same value: write only the elements that do not hold it yet (a steady run: none)
static void fill_const(odr_ctx *c, odr_particles *p, int v, float f) {
  const bool same = p->env_cok[v] && memcmp(&p->env_cval[v], &f, sizeof f) == 0;
  const long long from = same ? std::min(p->env_cn[v], p->n) : 0;
  if (from < p->n)
    hipLaunchKernelGGL(k_fill_f32, dim3(nblk(p->n - from)), dim3(BLOCK), 0, c->stream, p->env[v] + from, p->n - from, f);
  p->env_cok[v] = true; p->env_cval[v] = f; p->env_cn[v] = p->n;
}

// fast path of odr_env_sample: a group served by one constant reader that covers the globe and all times
// (reader_constant.py:60-82): every element gets the constant, cast to float32 -- a fill per variable
static bool launch_env_constant(odr_ctx *c, odr_particles *p, const int *grp, int ng, double t) {
  const DevWorld &w = c->hw;
  if (w.nlist[grp[0]] != 1) return false;
  const DevSource &s = w.src[w.list[grp[0]][0]];
  if (s.kind != SRC_CONSTANT || s.proj.kind != PROJ_LATLONG || s.lon_mode != 1) return false;
  if (!(s.xmin <= -180 && s.xmax >= 180 && s.ymin <= -90 && s.ymax >= 90 && s.zmin == -INFINITY && s.zmax == INFINITY))
    return false;
  if (!s.always_valid && (t < s.tmin || t > s.tmax)) return false;
  for (int k = 0; k < ng; ++k) {
    float f = (float)s.const_val[grp[k]];
    if (!std::isfinite(f)) f = std::isfinite(w.fallback[grp[k]]) ? w.fallback[grp[k]] : f;
    if (grp[k] == VAR_TEMP && f > 100.f) f = (float)((double)f - 273.15);   // Kelvin -> Celsius (environment.py:829-838)
    fill_const(c, p, grp[k], f);
  }
  return true;
}

// fast path of odr_env_sample: {x,y}_sea_water_velocity (and land_binary_mask) served by one double-gyre reader
bool odr_i_gyre_source(const odr_ctx *c, int var, int &sid) {
  const DevWorld &w = c->hw;
  if (w.nlist[var] != 1) return false;
  sid = w.list[var][0];
  const DevSource &s = w.src[sid];
  return s.kind == SRC_DOUBLE_GYRE && s.proj.kind == PROJ_STERE_EQUIT_SPHERE && s.proj.es == 0 && !s.mod360_x;
}
static bool launch_env_gyre(odr_ctx *c, odr_particles *p, const int *grp, int ng, double t, int rec) {
  bool has_u = false, has_v = false, has_land = false;
  for (int k = 0; k < ng; ++k) {
    if (grp[k] == VAR_U) has_u = true;
    else if (grp[k] == VAR_V) has_v = true;
    else if (grp[k] == VAR_LAND) has_land = true;
    else return false;
  }
  int sid;
  if (!has_u || !has_v || !gyre_source(c, VAR_U, sid)) return false;
  const DevSource &s = c->hw.src[sid];
  if (!s.always_valid && (t < s.tmin || t > s.tmax)) return false;
  const double snw = sin(s.params[2] * (t - s.params[3]));
  hipLaunchKernelGGL(k_env_gyre, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, c->dw, sid, view(p), snw, has_land ? 1 : 0, rec);
  return true;
}

// ----------------------------------------------------------------- environment
template <int NV>
static void launch_group(odr_ctx *c, odr_particles *p, const int *vars, double t, int rec) {
  GroupVars<NV> gv;
  for (int k = 0; k < NV; ++k) gv.v[k] = vars[k];
  hipLaunchKernelGGL(k_env_group<NV>, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, c->dw, view(p), gv, t, rec);
}

int odr_env_sample(odr_ctx *c, odr_particles *p, int nvars, const int32_t *var_ids, double t, float *const *out_host) {
  return env_sample_impl(c, p, nvars, var_ids, t, out_host, true);
}

// record_positions: remember the sample position (slon/slat) for the profiles of odr_vmix
int odr_i_env_sample(odr_ctx *c, odr_particles *p, int nvars, const int32_t *var_ids, double t,
                     float *const *out_host, bool record_positions) {
  p->epoch++;
  REQUIRE(nvars > 0 && nvars <= NVAR && var_ids, "bad variable list");
  HIPCHK(hipSetDevice(c->device));
  int rc;
  for (int k = 0; k < nvars; ++k) {
    REQUIRE(var_ids[k] >= 0 && var_ids[k] < NVAR, "bad variable id %d", var_ids[k]);
    if ((rc = ensure_env(c, p, var_ids[k]))) return rc;
  }
  if ((rc = flush_world(c))) return rc;
  if ((rc = odr_i_ensure_ranks(c, p))) return rc;   // ensemble data only
  if (record_positions) p->profiles_f32 = (c->hw.f32pos & 1) != 0;   // (odr_vmix gathers its profiles at these positions, in their class)
  if (record_positions && p->rank_on && p->n > 0) {
    // the main-loop call: an ensemble DIFFUSIVITY hands every element the column of the member its element values come
    // from (k_kmember) -- kept until odr_vmix, across the renumbering of the Runge-Kutta stage calls and the compaction
    for (int k = 0; k < c->hw.nlist[VAR_KZ]; ++k) {
      const DevSource &s = c->hw.src[c->hw.list[VAR_KZ][k]];
      if (s.kind != SRC_GRID) continue;
      if (s.members[VAR_KZ] > 1) {
        // (the member is parked in property slot 8; a model that uses all nine slots -- Leeway's `capsized` is slot 8 -- and
        // requires the diffusivity would have its property overwritten: refused)
        if (p->aux_user & (1u << AUX_KMEMBER))
          return fail(ODR_ERR_STATE, "an ensemble ocean_vertical_diffusivity parks its member in property slot %d, which the model has set", AUX_KMEMBER);
        p->kmember_on = true;
        if (!p->aux[AUX_KMEMBER]) {
          HIPCHK(hipMalloc((void **)&p->aux[AUX_KMEMBER], sizeof(float) * (size_t)p->cap));
          HIPCHK(hipMemsetAsync(p->aux[AUX_KMEMBER], 0, sizeof(float) * (size_t)p->cap, c->stream));
        }
        const PView pv = view(p);
        hipLaunchKernelGGL(k_kmember, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, pv.rank, p->n, s.members[VAR_KZ], pv.aux[AUX_KMEMBER]);
      }
      break;
    }
  }
  if (p->n > 0) {
    // variable groups = variables sharing the same priority list (get_reader_groups, environment.py:339-374)
    bool done[NVAR] = {false};
    int rec = record_positions ? 1 : 0;
    for (int a = 0; a < nvars; ++a) {
      int va = var_ids[a];
      if (done[va]) continue;
      int grp[NVAR], ng = 0;
      for (int b2 = a; b2 < nvars; ++b2) {
        int vb = var_ids[b2];
        if (done[vb]) continue;
        bool same = c->hw.nlist[vb] == c->hw.nlist[va];
        for (int k = 0; same && k < c->hw.nlist[va]; ++k) same = c->hw.list[vb][k] == c->hw.list[va][k];
        if (same) { grp[ng++] = vb; done[vb] = true; }
      }
      if (c->hw.nlist[va] == 0) {  // no reader: fallback only
        for (int k = 0; k < ng; ++k) {
          float f = c->hw.fallback[grp[k]];
          if (grp[k] == VAR_TEMP && f > 100.f) f = (float)((double)f - 273.15);   // Kelvin -> Celsius (environment.py:829-838)
          fill_const(c, p, grp[k], f);
        }
        continue;
      }
      if (!getenv("ODR_NO_FAST_PATH") && launch_env_constant(c, p, grp, ng, t)) continue;   // (positions recorded below)
      for (int k = 0; k < ng; ++k) p->env_cok[grp[k]] = false;   // written per element by the kernels below
      if (ng > MAXG && c->hw.nlist[va] == 1) {
        // more variables from ONE reader than a launch carries (OpenOil with sea ice: 10): consecutive launches of up to
        // MAXG variables, vector pairs kept together.  With a single reader in the list there is no next reader whose
        // turn would depend on the whole group's missing-data mask, so the split does not change any value.
        static const int pairs[4][2] = {{VAR_U, VAR_V}, {VAR_XWIND, VAR_YWIND}, {VAR_SX, VAR_SY}, {VAR_ICE_U, VAR_ICE_V}};
        int ord[NVAR], no = 0;
        bool placed[NVAR] = {false};
        for (auto &pr : pairs) {
          bool hx = false, hy = false;
          for (int k = 0; k < ng; ++k) { hx |= grp[k] == pr[0]; hy |= grp[k] == pr[1]; }
          if (hx && hy) { ord[no++] = pr[0]; ord[no++] = pr[1]; placed[pr[0]] = placed[pr[1]] = true; }
        }
        for (int k = 0; k < ng; ++k) if (!placed[grp[k]]) ord[no++] = grp[k];
        int at = 0, rcg = 0;
        while (at < no && !rcg) {
          const int m = std::min(MAXG, no - at);   // MAXG is even and the pairs come first, two by two: no pair is cut
          int sub[MAXG];
          for (int k = 0; k < m; ++k) sub[k] = ord[at + k];
          bool ok = !getenv("ODR_NO_FAST_PATH") && launch_env_grid(c, p, sub, m, t, rec);
          if (!ok) {
            switch (m) {
              case 1: launch_group<1>(c, p, sub, t, rec); break;
              case 2: launch_group<2>(c, p, sub, t, rec); break;
              case 3: launch_group<3>(c, p, sub, t, rec); break;
              case 4: launch_group<4>(c, p, sub, t, rec); break;
              case 5: launch_group<5>(c, p, sub, t, rec); break;
              case 6: launch_group<6>(c, p, sub, t, rec); break;
              case 7: launch_group<7>(c, p, sub, t, rec); break;
              default: launch_group<8>(c, p, sub, t, rec); break;
            }
          }
          rec = 0;
          at += m;
        }
        continue;
      }
      if (!getenv("ODR_NO_FAST_PATH") && launch_env_grid(c, p, grp, ng, t, rec)) { rec = 0; continue; }
      if (!getenv("ODR_NO_FAST_PATH") && launch_env_gyre(c, p, grp, ng, t, rec)) { rec = 0; continue; }
      // the whole group goes through one launch: the reference decides "static variables only"
      // and the missing-data mask per reader call on the full group (structured.py:224-229,
      // environment.py:727-746)
      switch (ng) {
        case 1: launch_group<1>(c, p, grp, t, rec); break;
        case 2: launch_group<2>(c, p, grp, t, rec); break;
        case 3: launch_group<3>(c, p, grp, t, rec); break;
        case 4: launch_group<4>(c, p, grp, t, rec); break;
        case 5: launch_group<5>(c, p, grp, t, rec); break;
        case 6: launch_group<6>(c, p, grp, t, rec); break;
        case 7: launch_group<7>(c, p, grp, t, rec); break;
        case 8: launch_group<8>(c, p, grp, t, rec); break;
        default: return fail(ODR_ERR_CAPACITY, "at most 8 variables may share one reader list (%d)", ng);
      }
      rec = 0;
    }
    if (rec) hipLaunchKernelGGL(k_record_prev, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), 0);
    HIPCHK(hipGetLastError());
  }
  if (out_host) {
    for (int k = 0; k < nvars; ++k)
      if (out_host[k] && p->n > 0)
        D2H(out_host[k], p->env[var_ids[k]], sizeof(float) * (size_t)p->n);
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  return 0;
}

int odr_env_download(odr_ctx *c, odr_particles *p, int32_t var, float *out) {
  REQUIRE(var >= 0 && var < NVAR && out, "bad arguments");
  if (!p->env[var]) return fail(ODR_ERR_STATE, "variable %d has not been sampled", var);
  if (p->n > 0) D2H(out, p->env[var], sizeof(float) * (size_t)p->n);
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int odr_env_upload(odr_ctx *c, odr_particles *p, int32_t var, const float *host) {
  p->epoch++;  // invalidates the cached reductions (reduce())
  REQUIRE(var >= 0 && var < NVAR && host, "bad arguments");
  p->env_cok[var] = false;
  int rc = ensure_env(c, p, var);
  if (rc) return rc;
  if (p->n > 0) H2D(p->env[var], host, sizeof(float) * (size_t)p->n);
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

static int host_to_scratch(odr_ctx *c, odr_particles *p, const double *a, const double *b, size_t n, double **da, double **db) {
  void *s;
  int rc = scratch(c, p, sizeof(double) * n * 2, &s);
  if (rc) return rc;
  *da = (double *)s;
  *db = *da + n;
  H2D(*da, a, sizeof(double) * n);
  if (b) H2D(*db, b, sizeof(double) * n);
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int odr_i_env_noise(odr_ctx *c, odr_particles *p, int vx, int vy, double std, int distribution, int rng_mode,
                    const double *dev_nx, const double *dev_ny, unsigned long long step) {
  p->epoch++;  // invalidates the cached reductions (reduce())
  REQUIRE(vx >= 0 && vx < NVAR && vy >= 0 && vy < NVAR, "bad variable ids");
  REQUIRE(distribution == ODR_NOISE_NORMAL || distribution == ODR_NOISE_UNIFORM, "unknown noise distribution %d", distribution);
  if (!p->env[vx] || !p->env[vy]) return fail(ODR_ERR_STATE, "variables not sampled");
  p->env_cok[vx] = p->env_cok[vy] = false;
  if (p->n == 0) return 0;
  hipLaunchKernelGGL(k_env_noise, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), vx, vy, std, distribution, rng_mode,
                     dev_nx, dev_ny, c->seed, step);
  HIPCHK(hipGetLastError());
  return 0;
}

int odr_env_add_noise(odr_ctx *c, odr_particles *p, int32_t vx, int32_t vy, double std, int distribution, int rng_mode,
                      const double *hnx, const double *hny, uint64_t step) {
  double *da = nullptr, *db = nullptr;
  if (rng_mode == ODR_RNG_HOST && p->n > 0) {
    REQUIRE(hnx && hny, "host draws required in ODR_RNG_HOST mode");
    int rc = host_to_scratch(c, p, hnx, hny, (size_t)p->n, &da, &db);
    if (rc) return rc;
  }
  return odr_i_env_noise(c, p, vx, vy, std, distribution, rng_mode, da, db, (unsigned long long)step);
}

// odr_advect, odr_env_coast_advect: odr_step.hip
bool odr_i_uv_fast_source(const odr_ctx *c, int &sid, double t_lo, double t_hi) {
  const DevWorld &w = c->hw;
  if (w.nlist[VAR_U] != 1 || w.nlist[VAR_V] != 1 || w.list[VAR_U][0] != w.list[VAR_V][0]) return false;
  sid = w.list[VAR_U][0];
  const DevSource &s = w.src[sid];
  if (s.kind != SRC_GRID || s.nlevels < 1) return false;
  if (s.members[VAR_U] > 1 || s.members[VAR_V] > 1) return false;         // ensemble data: generic kernels
  if (!s.always_valid && (t_lo < s.tmin || t_hi > s.tmax)) return false;  // generic path handles uncovered times
  const DevBlock &g0 = s.slot[s.level_slot[0]];
  for (int k = 0; k < s.nlevels; ++k) {
    const DevBlock &b = s.slot[s.level_slot[k]];
    if (!b.data[VAR_U] || b.es[VAR_U] != 2 || b.data[VAR_V] != b.data[VAR_U] + 1 || !b.small || b.rec != g0.rec) return false;
    if (b.ny != g0.ny || b.nx != g0.nx || b.x0 != g0.x0 || b.xspan != g0.xspan || b.y0 != g0.y0 ||
        b.yspan != g0.yspan || b.var_nz[VAR_U] != g0.var_nz[VAR_U])
      return false;
  }
  return true;
}


int odr_update_positions(odr_ctx *c, odr_particles *p, const double *u, const double *v, int is_f32, double dt) {
  REQUIRE(u && v, "velocities required");
  if (p->n == 0) return 0;
  double *da, *db;
  int rc = host_to_scratch(c, p, u, v, (size_t)p->n, &da, &db);
  if (rc) return rc;
  hipLaunchKernelGGL(k_update_positions, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), da, db, is_f32, dt);
  HIPCHK(hipGetLastError());
  return 0;
}

// A reader that hands `var` out as a list of `members` arrays (ensemble data, readers/interpolation/structured.py:119-135).
// Declared before the blocks are uploaded; their arrays then hold the members one after the other along the layer axis.
int odr_source_set_members(odr_ctx *c, int32_t sid, int32_t var, int32_t members) {
  REQUIRE(sid >= 0 && sid < c->nsrc && c->hw.src[sid].kind == SRC_GRID, "source %d is not a grid source", sid);
  REQUIRE(var >= 0 && var < NVAR && members >= 1 && members <= 1024, "bad ensemble declaration");
  c->hw.src[sid].members[var] = members > 1 ? members : 0;
  c->dirty = true;
  return 0;
}

// rank of every present element in ascending ID = its position in the reference's arrays (release order, for seed
// times that are monotonic in ID); PView::rank.  Only computed when some reader delivers ensemble data.
int odr_i_ensure_ranks(odr_ctx *c, odr_particles *p) {
  p->rank_on = 0;
  if (!any_members(c) || p->n == 0) return 0;
  const long long nw = p->id_max / 32 + 1;
  if (!p->rank) HIPCHK(hipMalloc((void **)&p->rank, sizeof(int) * (size_t)p->cap));
  if (p->rank_words_n < nw) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (p->rank_words) { HIPCHK(hipFree(p->rank_words)); HIPCHK(hipFree(p->rank_before)); HIPCHK(hipFree(p->rank_bsum)); }
    const long long cap = nw * 2;
    HIPCHK(hipMalloc((void **)&p->rank_words, sizeof(unsigned) * (size_t)cap));
    HIPCHK(hipMalloc((void **)&p->rank_before, sizeof(unsigned) * (size_t)cap));
    HIPCHK(hipMalloc((void **)&p->rank_bsum, sizeof(unsigned) * (size_t)((cap + 1023) / 1024 + 1)));
    p->rank_words_n = cap;
  }
  HIPCHK(hipMemsetAsync(p->rank_words, 0, sizeof(unsigned) * (size_t)nw, c->stream));
  // Numbered are the elements the reference HANDS to the ensemble reader's block: the ones its domain covers
  // (get_variables_interpolated, variables.py:747-765 -> ReaderBlock.interpolate, interpolation/structured.py:119-135).
  // One ensemble reader at the head of its priority lists is the case built; with several, or behind another reader (whose
  // missing-data mask would decide who is handed on), every active element is numbered as before.
  int ens_sid = -1, n_ens = 0;
  bool head = true;
  for (int s = 0; s < c->nsrc; ++s) {
    bool has = false;
    for (int v = 0; v < NVAR; ++v) has |= c->hw.src[s].members[v] > 1;
    if (!has) continue;
    ++n_ens; ens_sid = s;
    for (int v = 0; v < NVAR; ++v)
      if (c->hw.src[s].members[v] > 1 && !(c->hw.nlist[v] > 0 && c->hw.list[v][0] == s)) head = false;
  }
  // (a particle set of a sharded run -- odr_particles_set_rank_offset was called on it, with whatever offset -- numbers its
  // active elements: the covered elements of the lower ranks are not known here)
  if (n_ens != 1 || !head || p->rank_sharded || getenv("ODR_RANK_AMONG_ACTIVE")) ens_sid = -1;
  hipLaunchKernelGGL(k_rank_mark, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, c->dw, ens_sid, view(p), p->rank_words);
  hipLaunchKernelGGL(k_rank_count, dim3(nblk(nw)), dim3(BLOCK), 0, c->stream, p->rank_words, nw, p->rank_before);
  const unsigned nsb = (unsigned)((nw + 1023) / 1024);
  hipLaunchKernelGGL(k_scan_local, dim3(nsb), dim3(1024), 0, c->stream, p->rank_before, nw, p->rank_bsum);
  hipLaunchKernelGGL(k_cmp_scan, dim3(1), dim3(1024), 0, c->stream, p->rank_bsum, (long long)nsb, c->counter + 3);
  hipLaunchKernelGGL(k_scan_add, dim3(nsb), dim3(1024), 0, c->stream, p->rank_before, nw, p->rank_bsum);
  hipLaunchKernelGGL(k_rank_assign, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, p->i32[0], p->n, p->rank_words, p->rank_before,
                     p->rank, (int)p->rank_offset);
  HIPCHK(hipGetLastError());
  p->rank_on = 1;
  return 0;
}

int odr_particles_set_rank_offset(odr_ctx *c, odr_particles *p, int64_t offset) {
  (void)c;
  REQUIRE(p && offset >= 0 && offset < (1ll << 30), "bad rank offset");
  p->rank_offset = offset;
  p->rank_sharded = 1;
  return 0;
}

// Per-element factors of the following movers, derived from the sampled sea_ice_area_fraction (OpenOil.advect_oil,
// openoil.py:1179-1216).  Stays in force until set again.
int odr_set_element_factor(odr_ctx *c, odr_particles *p, int kind) {
  (void)c;
  REQUIRE(kind >= ODR_FACTOR_SCALAR && kind <= ODR_FACTOR_ICE_DRIFT, "bad element factor kind %d", kind);
  if (kind != ODR_FACTOR_SCALAR && !p->env[VAR_ICE_A]) return fail(ODR_ERR_STATE, "sea_ice_area_fraction has not been sampled");
  p->ice_kind = kind;
  return 0;
}

// advect_with_sea_ice (physics_methods.py:693-710)
int odr_advect_sea_ice(odr_ctx *c, odr_particles *p, double dt, double factor) {
  p->epoch++;
  if (p->n == 0) return 0;
  const int have = p->env[VAR_ICE_U] && p->env[VAR_ICE_V];
  if (!have) {
    if (!p->env[VAR_U] || !p->env[VAR_V]) return 0;   // "No sea ice velocity available"
    if (!p->env[VAR_XWIND] || !p->env[VAR_YWIND]) return fail(ODR_ERR_STATE, "wind has not been sampled");
  }
  if (p->ice_kind == ODR_FACTOR_ICE_DRIFT && !p->env[VAR_ICE_A]) return fail(ODR_ERR_STATE, "sea_ice_area_fraction has not been sampled");
  hipLaunchKernelGGL(k_advect_ice, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), dt, (float)factor, have);
  HIPCHK(hipGetLastError());
  return 0;
}

// model-specific float32 element properties (LagrangianArray subclasses): slots 0..8, for Leeway
// {downwind_slope, crosswind_slope, downwind_offset, crosswind_offset, downwind_eps, crosswind_eps,
//  jibe_probability, orientation, capsized} (leeway.py:50-131)
int odr_particles_set_property(odr_ctx *c, odr_particles *p, int slot, int64_t offset, int64_t count, const float *host) {
  p->epoch++;  // invalidates the cached reductions (reduce())
  REQUIRE(slot >= 0 && slot < 9 && host && offset >= 0 && count >= 0 && offset + count <= p->n, "bad property range");
  if (slot == AUX_KMEMBER && p->kmember_on)
    return fail(ODR_ERR_STATE, "property slot %d parks the member of an ensemble ocean_vertical_diffusivity on this particle set", slot);
  p->aux_user |= 1u << slot;
  if (!p->aux[slot]) {
    HIPCHK(hipMalloc((void **)&p->aux[slot], sizeof(float) * (size_t)p->cap));
    HIPCHK(hipMemsetAsync(p->aux[slot], 0, sizeof(float) * (size_t)p->cap, c->stream));
  }
  if (count) H2D(p->aux[slot] + offset, host, sizeof(float) * (size_t)count);
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
int odr_particles_get_property(odr_ctx *c, odr_particles *p, int slot, float *host) {
  REQUIRE(slot >= 0 && slot < 9 && host, "bad property slot");
  if (!p->aux[slot]) return fail(ODR_ERR_STATE, "property slot %d has not been set", slot);
  if (p->n) D2H(host, p->aux[slot], sizeof(float) * (size_t)p->n);
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// A copy of one property as it is NOW, for the next odr_history_record with ODR_HIST_PROPERTIES_FROM_SNAPSHOT: the launch
// that holds Leeway.update (odr_env_coast_leeway) jibes crosswind_slope / orientation BEFORE the step's record is taken,
// the reference's loop records first (basemodel/__init__.py:2276 state_to_buffer, :2293 update).
int odr_particles_snapshot_property(odr_ctx *c, odr_particles *p, int slot) {
  REQUIRE(slot >= 0 && slot < 9, "bad property slot");
  if (!p->aux[slot]) return fail(ODR_ERR_STATE, "property slot %d has not been set", slot);
  if (p->aux_snap_cap[slot] < p->cap) {
    if (p->aux_snap[slot]) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(p->aux_snap[slot])); p->aux_snap[slot] = nullptr; }
    p->aux_snap_cap[slot] = 0;
    HIPCHK(hipMalloc((void **)&p->aux_snap[slot], sizeof(float) * (size_t)p->cap));
    p->aux_snap_cap[slot] = p->cap;
  }
  if (p->n) HIPCHK(hipMemcpyAsync(p->aux_snap[slot], p->aux[slot], sizeof(float) * (size_t)p->n, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

// Leeway.update (models/leeway.py:430-494, capsizing off)
int odr_leeway(odr_ctx *c, odr_particles *p, double dt, double capsize_fraction, int rng_mode, const double *huni,
               uint64_t step) {
  p->status_epoch++;
  for (int k = 0; k < 9; ++k) if (!p->aux[k]) return fail(ODR_ERR_STATE, "Leeway property slot %d has not been set", k);
  if (!p->env[VAR_XWIND] || !p->env[VAR_YWIND] || !p->env[VAR_U] || !p->env[VAR_V])
    return fail(ODR_ERR_STATE, "wind and current must be sampled before odr_leeway");
  if (p->n == 0) return 0;
  double *du = nullptr, *dummy = nullptr;
  if (rng_mode == ODR_RNG_HOST) {
    REQUIRE(huni, "host uniforms required in ODR_RNG_HOST mode");
    int rc = host_to_scratch(c, p, huni, nullptr, (size_t)p->n, &du, &dummy);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_leeway, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), dt, (float)capsize_fraction, rng_mode,
                     du, c->seed, (unsigned long long)step);
  HIPCHK(hipGetLastError());
  return 0;
}

// The Leeway loop body between two compactions in one launch (k_step_leeway): Environment.get_environment of `var_ids` at t
// (environment.py:499-923) + drift:current_uncertainty / drift:wind_uncertainty (:869-891, device RNG) +
// interact_with_coastline (basemodel/__init__.py:670-746) + update_previous_state + Leeway.update (models/leeway.py:430-494,
// without capsizing).  Falls back to the separate entry points -- same results -- when wind, current (and landmask) do not
// come from ONE gridded reader that fits the burst sampler.
int odr_leeway_set_missing_code(odr_ctx *c, int32_t code) {
  REQUIRE(code >= 0, "bad status code");
  c->leeway_missing_code = code;
  return 0;
}

int odr_env_coast_leeway(odr_ctx *c, odr_particles *p, int nvars, const int32_t *var_ids, double t, int coast_action,
                         int stranded_code, int seeded_on_land_code, int store_previous, double dt, double capsize_fraction,
                         double std_current, double std_wind, uint64_t step, int64_t *n_on_land) {
  p->status_epoch++;
  p->epoch++;
  const int missing_code = c->leeway_missing_code;
  c->leeway_missing_code = 0;
  REQUIRE(nvars > 0 && nvars <= NVAR && var_ids, "bad variable list");
  REQUIRE(coast_action >= 0 && coast_action <= 2, "bad coastline action");
  REQUIRE(std_current >= 0 && std_wind >= 0, "uncertainties must not be negative");
  for (int k = 0; k < 9; ++k) if (!p->aux[k]) return fail(ODR_ERR_STATE, "Leeway property slot %d has not been set", k);
  HIPCHK(hipSetDevice(c->device));
  if (n_on_land) *n_on_land = 0;
  bool has[NVAR] = {false};
  for (int k = 0; k < nvars; ++k) { REQUIRE(var_ids[k] >= 0 && var_ids[k] < NVAR, "bad variable id %d", var_ids[k]); has[var_ids[k]] = true; }
  REQUIRE(has[VAR_XWIND] && has[VAR_YWIND] && has[VAR_U] && has[VAR_V], "the variable list must hold wind and current");
  if (coast_action && !has[VAR_LAND] && !p->env[VAR_LAND]) return fail(ODR_ERR_STATE, "land_binary_mask has not been sampled");
  int rc;
  if ((rc = flush_world(c))) return rc;
  auto same_list = [&](int a, int b) {
    if (c->hw.nlist[a] != c->hw.nlist[b]) return false;
    for (int k = 0; k < c->hw.nlist[a]; ++k) if (c->hw.list[a][k] != c->hw.list[b][k]) return false;
    return true;
  };
  // the group: wind pair, current pair, then every other variable of the list that shares their reader list
  int grp[NVAR], ng = 0, rest[NVAR], nrest = 0;
  grp[ng++] = VAR_XWIND; grp[ng++] = VAR_YWIND; grp[ng++] = VAR_U; grp[ng++] = VAR_V;
  bool seen[NVAR] = {false};
  seen[VAR_XWIND] = seen[VAR_YWIND] = seen[VAR_U] = seen[VAR_V] = true;
  for (int k = 0; k < nvars; ++k) {
    const int v = var_ids[k];
    if (seen[v]) continue;
    seen[v] = true;
    if (same_list(v, VAR_U)) grp[ng++] = v; else rest[nrest++] = v;
  }
  EnvGroupDesc G;
  const bool fuse = p->n > 0 && !c->hw.f32pos && !getenv("ODR_NO_FAST_PATH") && same_list(VAR_XWIND, VAR_U) && same_list(VAR_YWIND, VAR_U) &&
                    same_list(VAR_V, VAR_U) && ng <= MAXG && build_env_group(c, grp, ng, t, G) && G.burst;
  if (!fuse) {
    if ((rc = odr_env_sample(c, p, nvars, var_ids, t, nullptr))) return rc;
    if (std_current > 0 && (rc = odr_i_env_noise(c, p, VAR_U, VAR_V, std_current, ODR_NOISE_NORMAL, ODR_RNG_DEVICE, nullptr, nullptr, step))) return rc;
    if (std_wind > 0 && (rc = odr_i_env_noise(c, p, VAR_XWIND, VAR_YWIND, std_wind, ODR_NOISE_NORMAL, ODR_RNG_DEVICE, nullptr, nullptr, step))) return rc;
    if (missing_code && (rc = odr_deactivate_missing(c, p, nvars, var_ids, missing_code, nullptr))) return rc;
    if ((rc = odr_coastline(c, p, coast_action, stranded_code, seeded_on_land_code, n_on_land))) return rc;
    if (store_previous && (rc = odr_store_previous(c, p))) return rc;
    // (the caller compacts before odr_leeway in this lane: elements on land must not move)
    return ODR_SPLIT_LANE;
  }
  for (int k = 0; k < ng; ++k) { if ((rc = ensure_env(c, p, grp[k]))) return rc; p->env_cok[grp[k]] = false; }
  if (coast_action == 2) p->env_cok[VAR_LAND] = false;
  if (nrest && (rc = env_sample_impl(c, p, nrest, rest, t, nullptr, false))) return rc;
  LeewayStep S;
  memset(&S, 0, sizeof S);
  S.coast_action = coast_action; S.stranded_code = stranded_code; S.seeded_code = seeded_on_land_code;
  S.land_slot = -1; S.wind_slot = -1; S.uv_slot = -1;
  for (int k = 0; k < G.nv; ++k) {
    if (G.var[k] == VAR_LAND) S.land_slot = k;
    if (G.var[k] == VAR_XWIND) S.wind_slot = k;
    if (G.var[k] == VAR_U) S.uv_slot = k;
  }
  S.store_previous = store_previous;
  S.std_current = std_current; S.std_wind = std_wind;
  S.capsize_fraction = (float)capsize_fraction;
  S.seed = c->seed; S.step = (unsigned long long)step;
  if (missing_code) {   // only a variable without fallback can still be NaN after get_environment (environment.py:781-790)
    for (int k = 0; k < G.nv; ++k) if (std::isnan(c->hw.fallback[G.var[k]])) S.miss_grp[S.nmiss_grp++] = k;
    int nr = 0;
    for (int k = 0; k < nrest; ++k) if (std::isnan(c->hw.fallback[rest[k]])) { if (nr < 4) S.miss_rest[nr] = rest[k]; ++nr; }
    if (nr > 4) return fail(ODR_ERR_CAPACITY, "more than 4 variables without fallback outside the wind / current reader");
    S.nmiss_rest = nr;
    S.missing_code = (S.nmiss_grp || nr) ? missing_code : 0;
  }
  const bool count_in_launch = p->wcount && !p->external && !getenv("ODR_NO_STEP_COUNT");   // (odr_scan_status then folds the wave counts)
  if (coast_action || count_in_launch)
    HIPCHK(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long) * (count_in_launch ? 4 : 1), c->stream));
  if (count_in_launch) { S.wcount = p->wcount; S.sflags = c->counter + 2; }
  const DevSource &s = c->hw.src[G.sid];
  dim3 g(nblk(p->n)), b(BLOCK);
  PView v = view(p);
  G = env_bind_out(G, v);
  switch (odr_proj_template(s.proj)) {
    case PROJ_LATLONG: hipLaunchKernelGGL(k_step_leeway<PROJ_LATLONG>, g, b, 0, c->stream, c->dw, v, G, S, dt, c->counter); break;
    case PROJ_STERE_POLAR: hipLaunchKernelGGL(k_step_leeway<PROJ_STERE_POLAR>, g, b, 0, c->stream, c->dw, v, G, S, dt, c->counter); break;
    case PROJ_CURVILINEAR: hipLaunchKernelGGL(k_step_leeway<PROJ_CURVILINEAR>, g, b, 0, c->stream, c->dw, v, G, S, dt, c->counter); break;
    case PROJ_EXT: hipLaunchKernelGGL(k_step_leeway<PROJ_EXT>, g, b, 0, c->stream, c->dw, v, G, S, dt, c->counter); break;
    default: hipLaunchKernelGGL(k_step_leeway<PROJ_STERE_EQUIT_SPHERE>, g, b, 0, c->stream, c->dw, v, G, S, dt, c->counter); break;
  }
  HIPCHK(hipGetLastError());
  if (count_in_launch) { p->wcount_epoch = p->status_epoch; p->wcount_n = p->n; }
  return (coast_action && n_on_land) ? read_counter(c, n_on_land) : 0;   // n_on_land == NULL: no host synchronisation
}

// processes:capsizing of Leeway.update (models/leeway.py:438-455); call before odr_leeway
int odr_leeway_capsize(odr_ctx *c, odr_particles *p, double dt, double wind_threshold, double wind_threshold_sigma,
                       int rng_mode, const double *huni, uint64_t step) {
  p->status_epoch++;
  p->epoch++;
  if (!p->aux[8]) return fail(ODR_ERR_STATE, "Leeway property slot 8 (capsized) has not been set");
  if (!p->env[VAR_XWIND] || !p->env[VAR_YWIND]) return fail(ODR_ERR_STATE, "wind must be sampled before odr_leeway_capsize");
  REQUIRE(wind_threshold_sigma > 0, "capsizing:wind_threshold_sigma must be positive");
  if (p->n == 0) return 0;
  double *du = nullptr, *dummy = nullptr;
  if (rng_mode == ODR_RNG_HOST) {
    REQUIRE(huni, "host uniforms required in ODR_RNG_HOST mode");
    int rc = host_to_scratch(c, p, huni, nullptr, (size_t)p->n, &du, &dummy);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_capsize, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), dt, (float)wind_threshold,
                     (float)wind_threshold_sigma, rng_mode, du, c->seed, (unsigned long long)step);
  HIPCHK(hipGetLastError());
  return 0;
}

int odr_i_red_records(odr_ctx *c, odr_particles *p, double **rec) {
  const long long need = (long long)nblk(p->n) * (BLOCK / 64);
  if (c->red_rec_cap < need) {
    if (c->red_rec) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->red_rec)); c->red_rec = nullptr; c->red_rec_cap = 0; }
    const long long cap = (long long)nblk(p->cap > p->n ? p->cap : p->n) * (BLOCK / 64);
    HIPCHK(hipMalloc((void **)&c->red_rec, sizeof(double) * 6 * (size_t)cap));
    c->red_rec_cap = cap;
  }
  *rec = c->red_rec;
  return 0;
}
int odr_i_red_finish(odr_ctx *c, odr_particles *p) {
  const long long nrec = (long long)nblk(p->n) * (BLOCK / 64);
  hipLaunchKernelGGL(k_red_init, dim3(1), dim3(64), 0, c->stream, c->red);
  const unsigned g = (unsigned)std::min<long long>(128, (nrec + BLOCK - 1) / BLOCK);
  hipLaunchKernelGGL(k_red_finish, dim3(g ? g : 1), dim3(BLOCK), 0, c->stream, c->red_rec, nrec, c->red);
  HIPCHK(hipGetLastError());
  return 0;
}

// partial_ok: the caller only reads the slots the movers' tests need (a reduction formed by the step launch will do)
int odr_i_reduce(odr_ctx *c, odr_particles *p, double wdd, int relwind, bool wind_args_matter, bool extents, bool partial_ok) {
  if (c->red_pinned && c->red_owner == p) return 0;   // installed by the caller (odr_reduce_install): the all-rank values
  if (!p->external && c->red_owner == p && c->red_epoch == p->epoch && (c->red_extents || !extents) &&
      (!c->red_partial || partial_ok) && (!wind_args_matter || (c->red_wdd == wdd && c->red_rel == relwind)))
    return 0;
  c->red_owner = p; c->red_epoch = p->epoch; c->red_wdd = wdd; c->red_rel = relwind; c->red_extents = extents;
  c->red_partial = false;
  hipLaunchKernelGGL(k_red_init, dim3(1), dim3(64), 0, c->stream, c->red);
  const dim3 grid(nblk(p->n) < 2048u ? nblk(p->n) : 2048u);
  if (p->n > 0) {
    if (extents) hipLaunchKernelGGL(k_reduce<true>, grid, dim3(BLOCK), 0, c->stream, view(p), wdd, relwind, c->red);
    else hipLaunchKernelGGL(k_reduce<false>, grid, dim3(BLOCK), 0, c->stream, view(p), wdd, relwind, c->red);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// The movers' early-out tests formed by the launch of odr_env_coast_advect (StepDesc.red_*): persistent setting.
int odr_ctx_set_step_reduce(odr_ctx *c, int on, double wind_drift_depth, int relative_wind) {
  c->step_reduce_on = on ? 1 : 0;
  c->step_reduce_wdd = wind_drift_depth;
  c->step_reduce_rel = relative_wind ? 1 : 0;
  return 0;
}

int odr_reduce_scalars(odr_ctx *c, odr_particles *p, double wdd, double *out16) {
  REQUIRE(out16, "out16 NULL");
  c->red_owner = nullptr;  // lon/lat extremes change with every mover: always recomputed
  int rc = reduce(c, p, wdd, 0);
  if (rc) return rc;
  double r[R_N];
  D2H(r, c->red, sizeof r);
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int k = 0; k < 16; ++k) out16[k] = k < R_N ? r[k] : 0;
  out16[R_LONMIN] = -r[R_LONMIN];
  out16[R_LATMIN] = -r[R_LATMIN];
  out16[R_ZMIN] = -r[R_ZMIN];
  return 0;
}

// The global reductions of the movers (cdf / wdf extremes, wind speed maximum, Stokes maximum, D.max(), MLD.max():
// physics_methods.py:741,771-775,799-804, basemodel/__init__.py:1754, oceandrift.py:430) over the elements of THIS
// particle set, raw: slots 0 and 11 are counts, every other slot is a maximum (minima are stored negated).  A run sharded
// over several GPUs combines them over the ranks (sum / max) and hands the result back with odr_reduce_install: the
// movers that follow then see what the reference's single process would have seen, whatever the number of ranks.
int odr_reduce_local(odr_ctx *c, odr_particles *p, double wdd, int relwind, double *out16) {
  REQUIRE(out16, "out16 NULL");
  c->red_pinned = 0;
  c->red_owner = nullptr;
  int rc = reduce(c, p, wdd, relwind);
  if (rc) return rc;
  double r[R_N];
  D2H(r, c->red, sizeof r);
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int k = 0; k < 16; ++k) out16[k] = k < R_N ? r[k] : 0;
  return 0;
}
int odr_reduce_install(odr_ctx *c, odr_particles *p, const double *in16) {
  REQUIRE(in16, "in16 NULL");
  H2D(c->red, in16, sizeof(double) * R_N);
  HIPCHK(hipStreamSynchronize(c->stream));   // in16 is pageable
  c->red_owner = p; c->red_epoch = p->epoch; c->red_extents = true; c->red_partial = false;
  c->red_pinned = 1;      // until odr_reduce_unpin: the calls in between do not reduce again
  return 0;
}
int odr_reduce_unpin(odr_ctx *c) {
  c->red_pinned = 0;
  c->red_owner = nullptr;
  return 0;
}

int odr_advect_wind(odr_ctx *c, odr_particles *p, double dt, double wdd, int relwind, double factor) {
  if (!p->env[VAR_XWIND] || !p->env[VAR_YWIND]) return fail(ODR_ERR_STATE, "wind has not been sampled");
  if (relwind && (!p->env[VAR_U] || !p->env[VAR_V])) return fail(ODR_ERR_STATE, "current has not been sampled");
  if (p->n == 0) return 0;
  // wind identically 0 (no reader, fallback 0): wind_speed.max() == 0 -> "No wind for wind-sheared ocean drift" (:775-780)
  if (!relwind && env_is_const(p, VAR_XWIND, 0.0f) && env_is_const(p, VAR_YWIND, 0.0f)) return 0;
  int rc = reduce(c, p, wdd, relwind, true, false, true);
  if (rc) return rc;
  hipLaunchKernelGGL(k_advect_wind, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), dt, wdd, relwind, factor, c->red);
  HIPCHK(hipGetLastError());
  return 0;
}

int odr_stokes_drift(odr_ctx *c, odr_particles *p, double dt, int profile, int hs_mode, int tp_mode, double factor) {
  REQUIRE(profile >= 0 && profile <= 3 && hs_mode >= 0 && hs_mode <= 2 && tp_mode >= 0 && tp_mode <= 3, "bad stokes options");
  if (!p->env[VAR_SX] || !p->env[VAR_SY]) return fail(ODR_ERR_STATE, "Stokes drift has not been sampled");
  if (profile == 3) {   // windsea_swell: its own six variables instead of Hs / Tp / wind
    for (int v : {VAR_SWELL_DIR, VAR_SWELL_TP, VAR_SWELL_HS, VAR_WW_DIR, VAR_WW_TM, VAR_WW_HS})
      if (!p->env[v]) return fail(ODR_ERR_STATE, "the windsea_swell profile needs the swell / wind-sea direction, period and height (variable %d not sampled)", v);
  } else {
    if ((hs_mode == 0 && !p->env[VAR_HS]) || (tp_mode == 0 && !p->env[VAR_TP])) return fail(ODR_ERR_STATE, "Hs/Tp not sampled");
    if ((hs_mode == 1 || tp_mode == 1 || tp_mode == 3) && (!p->env[VAR_XWIND] || !p->env[VAR_YWIND])) return fail(ODR_ERR_STATE, "wind not sampled");
  }
  if (p->n == 0) return 0;
  if (env_is_const(p, VAR_SX, 0.0f) && env_is_const(p, VAR_SY, 0.0f)) return 0;   // "No Stokes drift velocity available" (:799-804)
  int rc = reduce(c, p, 0.0, 0, false, false, true);
  if (rc) return rc;
  hipLaunchKernelGGL(k_stokes, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), dt, profile, hs_mode, tp_mode, factor, c->red);
  HIPCHK(hipGetLastError());
  return 0;
}

int odr_hdiffusion(odr_ctx *c, odr_particles *p, double dt, int rng_mode, const double *hnx, const double *hny, uint64_t step) {
  if (!p->env[VAR_HDIFF]) return fail(ODR_ERR_STATE, "horizontal_diffusivity has not been sampled");
  if (p->n == 0) return 0;
  if (env_is_const(p, VAR_HDIFF, 0.0f)) return 0;   // "Horizontal diffusivity is 0, no random walk." (:1754)
  double *da = nullptr, *db = nullptr;
  int rc;
  if (rng_mode == ODR_RNG_HOST) {
    REQUIRE(hnx && hny, "host normals required in ODR_RNG_HOST mode");
    if ((rc = host_to_scratch(c, p, hnx, hny, (size_t)p->n, &da, &db))) return rc;
  }
  if ((rc = reduce(c, p, 0.0, 0, false, false, true))) return rc;
  hipLaunchKernelGGL(k_hdiff, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), dt, rng_mode, da, db, c->seed,
                     (unsigned long long)step, c->red);
  HIPCHK(hipGetLastError());
  return 0;
}

// drift:truncate_ocean_model_below_m (models/basemodel/environment.py:554-566): get_environment samples every reader at
// max(z, -depth) -- the elements keep their z.  odr_particles_truncate_z puts the clipped depths in place of z for the calls
// that SAMPLE (odr_env_sample, odr_advect: its Runge-Kutta stage calls are get_environment calls too); odr_particles_restore_z
// brings the elements' own z back before anything that moves them vertically.
__global__ __launch_bounds__(BLOCK) static void k_truncate_z(double *z, double *keep, long long n, double zmin) {
  const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const double v = z[i];
  keep[i] = v;
  z[i] = v < zmin ? zmin : v;     // z[z < -truncate_depth] = -truncate_depth
}
__global__ __launch_bounds__(BLOCK) static void k_restore_z(double *z, const double *keep, long long n) {
  const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) z[i] = keep[i];
}
int odr_particles_truncate_z(odr_ctx *c, odr_particles *p, double truncate_depth) {
  REQUIRE(truncate_depth >= 0, "truncate depth must not be negative");
  REQUIRE(!p->z_truncated, "z is already truncated: odr_particles_restore_z first");
  if (!p->z_keep) HIPCHK(hipMalloc((void **)&p->z_keep, sizeof(double) * (size_t)p->cap));
  if (p->n > 0) hipLaunchKernelGGL(k_truncate_z, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, p->d64[2], p->z_keep, p->n, -truncate_depth);
  HIPCHK(hipGetLastError());
  p->z_truncated = true; p->z_keep_n = p->n;
  p->epoch++;
  return 0;
}
int odr_particles_restore_z(odr_ctx *c, odr_particles *p) {
  if (!p->z_truncated) return 0;
  REQUIRE(p->n == p->z_keep_n, "the element set changed while z was truncated");
  if (p->n > 0) hipLaunchKernelGGL(k_restore_z, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, p->d64[2], (const double *)p->z_keep, p->n);
  HIPCHK(hipGetLastError());
  p->z_truncated = false;
  p->epoch++;
  return 0;
}

// advect_wind -> stokes_drift -> horizontal_diffusion (any subset, in that order) in one launch: k_movers.  Every check and
// early return of the three entry points above is kept per mover; what is left of `which` after them runs as one kernel
// behind one reduction.  Bit-identical to the separate calls (tests/test_gpu_movers.py).
int odr_movers(odr_ctx *c, odr_particles *p, double dt, int which, double wdd, int relwind, double wind_factor, int profile,
               int hs_mode, int tp_mode, double stokes_factor, int rng_mode, const double *hnx, const double *hny, uint64_t step) {
  REQUIRE(which >= 0 && which <= 7, "which: 1 wind | 2 Stokes drift | 4 horizontal diffusion");
  if (which & 1) {
    if (!p->env[VAR_XWIND] || !p->env[VAR_YWIND]) return fail(ODR_ERR_STATE, "wind has not been sampled");
    if (relwind && (!p->env[VAR_U] || !p->env[VAR_V])) return fail(ODR_ERR_STATE, "current has not been sampled");
    if (!relwind && env_is_const(p, VAR_XWIND, 0.0f) && env_is_const(p, VAR_YWIND, 0.0f)) which &= ~1;
  }
  if (which & 2) {
    REQUIRE(profile >= 0 && profile <= 3 && hs_mode >= 0 && hs_mode <= 2 && tp_mode >= 0 && tp_mode <= 3, "bad stokes options");
    if (!p->env[VAR_SX] || !p->env[VAR_SY]) return fail(ODR_ERR_STATE, "Stokes drift has not been sampled");
    if (profile == 3) {
      for (int v : {VAR_SWELL_DIR, VAR_SWELL_TP, VAR_SWELL_HS, VAR_WW_DIR, VAR_WW_TM, VAR_WW_HS})
        if (!p->env[v]) return fail(ODR_ERR_STATE, "the windsea_swell profile needs the swell / wind-sea direction, period and height (variable %d not sampled)", v);
    } else {
      if ((hs_mode == 0 && !p->env[VAR_HS]) || (tp_mode == 0 && !p->env[VAR_TP])) return fail(ODR_ERR_STATE, "Hs/Tp not sampled");
      if ((hs_mode == 1 || tp_mode == 1 || tp_mode == 3) && (!p->env[VAR_XWIND] || !p->env[VAR_YWIND])) return fail(ODR_ERR_STATE, "wind not sampled");
    }
    if (env_is_const(p, VAR_SX, 0.0f) && env_is_const(p, VAR_SY, 0.0f)) which &= ~2;
  }
  if (which & 4) {
    if (!p->env[VAR_HDIFF]) return fail(ODR_ERR_STATE, "horizontal_diffusivity has not been sampled");
    if (env_is_const(p, VAR_HDIFF, 0.0f)) which &= ~4;
  }
  if (p->n == 0 || which == 0) return 0;
  MoversDesc M;
  memset(&M, 0, sizeof M);
  int rc;
  if ((which & 4) && rng_mode == ODR_RNG_HOST) {
    REQUIRE(hnx && hny, "host normals required in ODR_RNG_HOST mode");
    double *da = nullptr, *db = nullptr;
    if ((rc = host_to_scratch(c, p, hnx, hny, (size_t)p->n, &da, &db))) return rc;
    M.hnx = da; M.hny = db;
  }
  // the reduction of the first mover that needs the wind arguments (the cached one is reused by the others: nothing they
  // read changes in between)
  if ((rc = (which & 1) ? reduce(c, p, wdd, relwind, true, false, true) : reduce(c, p, 0.0, 0, false, false, true))) return rc;
  M.which = which; M.relative_wind = relwind; M.profile = profile; M.hs_mode = hs_mode; M.tp_mode = tp_mode; M.rng_mode = rng_mode;
  M.dt = dt; M.wind_drift_depth = wdd; M.wind_factor = wind_factor; M.stokes_factor = stokes_factor;
  M.seed = c->seed; M.step = (unsigned long long)step;
  hipLaunchKernelGGL(k_movers, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), M, c->red);
  HIPCHK(hipGetLastError());
  return 0;
}

// odr_vmix followed by odr_vertical_advection in one kernel (same particle, same z): request
// the fusion for the next odr_vmix call
int odr_vmix_fuse_vertical_advection(odr_ctx *c, int at_surface) {
  c->fuse_vadv = at_surface ? 1 : 0;
  return 0;
}

int odr_vertical_advection(odr_ctx *c, odr_particles *p, double dt, int at_surface) {
  p->epoch++;  // invalidates the cached reductions (reduce())
  if (!p->env[VAR_W]) return fail(ODR_ERR_STATE, "upward_sea_water_velocity has not been sampled");
  if (p->n == 0) return 0;
  hipLaunchKernelGGL(k_vadvect, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), dt, at_surface);
  HIPCHK(hipGetLastError());
  return 0;
}

int odr_vertical_buoyancy(odr_ctx *c, odr_particles *p, double dt) {
  p->status_epoch++;
  p->epoch++;  // invalidates the cached reductions (reduce())
  if (!p->env[VAR_DEPTH]) return fail(ODR_ERR_STATE, "sea_floor_depth_below_sea_level has not been sampled");
  int rc = ensure_env(c, p, VAR_SSH);
  if (rc) return rc;
  if (p->n == 0) return 0;
  hipLaunchKernelGGL(k_vbuoy, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), dt, c->seafloor);
  HIPCHK(hipGetLastError());
  return 0;
}

int odr_i_read_counter(odr_ctx *c, int64_t *out) {
  if (out) {
    unsigned long long v;
    D2H(&v, c->counter, sizeof v);
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = (int64_t)v;
  }
  return 0;
}

// update_previous_state (basemodel/__init__.py:642-668): remember lon/lat for 'previous' actions
int odr_store_previous(odr_ctx *c, odr_particles *p) {
  if (p->n == 0) return 0;
  hipLaunchKernelGGL(k_record_prev, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), 1);
  HIPCHK(hipGetLastError());
  return 0;
}

int odr_source_time_coverage(odr_ctx *c, int32_t sid, double t_start, double t_end, int always_valid) {
  REQUIRE(sid >= 0 && sid < c->nsrc, "unknown source %d", sid);
  c->hw.src[sid].tmin = t_start;
  c->hw.src[sid].tmax = t_end;
  c->hw.src[sid].always_valid = always_valid;
  c->dirty = true;
  return 0;
}

int odr_deactivate_missing(odr_ctx *c, odr_particles *p, int nvars, const int32_t *var_ids, int32_t code, int64_t *n_missing) {
  p->status_epoch++;
  p->epoch++;
  REQUIRE(nvars >= 0 && nvars <= NVAR && (nvars == 0 || var_ids), "bad variable list");
  if (n_missing) *n_missing = 0;
  VarList L;
  L.n = 0;
  for (int k = 0; k < nvars; ++k) {
    REQUIRE(var_ids[k] >= 0 && var_ids[k] < NVAR, "bad variable id %d", var_ids[k]);
    // only a variable without fallback can still be NaN after get_environment (environment.py:781-790)
    if (p->env[var_ids[k]] && std::isnan(c->hw.fallback[var_ids[k]])) L.var[L.n++] = var_ids[k];
  }
  if (p->n == 0 || L.n == 0) return 0;
  HIPCHK(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(k_deactivate_missing, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), L, (int)code, c->counter);
  HIPCHK(hipGetLastError());
  return read_counter(c, n_missing);
}

int odr_increase_age(odr_ctx *c, odr_particles *p, double dt, double max_age_seconds, int retired_code) {
  if (max_age_seconds > 0) p->status_epoch++;   // only retirement deactivates
  p->epoch++;  // invalidates the cached reductions (reduce())
  if (p->n == 0) return 0;
  hipLaunchKernelGGL(k_age, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), (float)dt, (float)max_age_seconds, retired_code);
  HIPCHK(hipGetLastError());
  return 0;
}

int odr_coastline(odr_ctx *c, odr_particles *p, int action, int code, int seeded_on_land_code, int64_t *n_on_land) {
  p->status_epoch++;
  p->epoch++;  // invalidates the cached reductions (reduce())
  REQUIRE(action >= 0 && action <= 2, "bad coastline action");
  if (n_on_land) *n_on_land = 0;
  if (action == 0 || p->n == 0) return 0;
  if (!p->env[VAR_LAND]) return fail(ODR_ERR_STATE, "land_binary_mask has not been sampled");
  if (action == 2) p->env_cok[VAR_LAND] = false;   // elements on land get land_binary_mask = 0
  HIPCHK(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(k_coast, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), action, code, seeded_on_land_code, c->counter);
  HIPCHK(hipGetLastError());
  return read_counter(c, n_on_land);
}

int odr_coastline_crossing(odr_ctx *c, odr_particles *p, int action, int code, int seeded_on_land_code,
                           double precision_deg, int32_t landmask_source, int64_t *n_on_land) {
  p->status_epoch++;
  p->epoch++;  // invalidates the cached reductions (reduce())
  REQUIRE(action == ODR_COAST_STRANDING || action == ODR_COAST_PREVIOUS, "bad coastline action");
  REQUIRE(precision_deg > 0, "coastline_approximation_precision must be positive");
  REQUIRE(landmask_source >= 0 && landmask_source < c->nsrc && c->hw.src[landmask_source].kind == SRC_LANDMASK,
          "source %d is not a landmask raster", landmask_source);
  if (n_on_land) *n_on_land = 0;
  if (p->n == 0) return 0;
  if (!p->env[VAR_LAND]) return fail(ODR_ERR_STATE, "land_binary_mask has not been sampled");
  int rc = flush_world(c);
  if (rc) return rc;
  p->env_cok[VAR_LAND] = false;
  HIPCHK(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(k_coast_crossing, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, c->dw, (int)landmask_source, view(p),
                     action, code, seeded_on_land_code, precision_deg, c->counter);
  HIPCHK(hipGetLastError());
  return read_counter(c, n_on_land);
}

int odr_seafloor_action(odr_ctx *c, odr_particles *p, int action, int32_t code, int64_t *n_below) {
  p->status_epoch++;
  p->epoch++;  // invalidates the cached reductions (reduce())
  REQUIRE(action >= ODR_SEAFLOOR_LIFT && action <= ODR_SEAFLOOR_PREVIOUS, "unknown seafloor action %d", action);
  if (n_below) *n_below = 0;
  if (p->n == 0) return 0;
  if (!p->env[VAR_DEPTH]) return fail(ODR_ERR_STATE, "sea_floor_depth_below_sea_level has not been sampled");
  HIPCHK(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(k_seafloor, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), action, (int)code, c->counter);
  HIPCHK(hipGetLastError());
  return read_counter(c, n_below);
}

int odr_set_seafloor_action(odr_ctx *c, int action, int32_t code) {
  REQUIRE(action >= 0 && action <= ODR_SEAFLOOR_PREVIOUS, "unknown seafloor action %d", action);
  c->seafloor = action | ((int)code << 8);
  return 0;
}

// elements of the active set carrying status_code (flagged, not yet removed by odr_compact)
int odr_particles_count_status(odr_ctx *c, odr_particles *p, int32_t code, int64_t *n) {
  REQUIRE(n, "n NULL");
  *n = 0;
  if (p->n == 0) return 0;
  HIPCHK(hipMemsetAsync(c->counter, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(k_count_status, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), (int)code, c->counter);
  HIPCHK(hipGetLastError());
  return read_counter(c, n);
}

int odr_particles_remap_status(odr_ctx *c, odr_particles *p, int32_t from, int32_t to) {
  if (p->n == 0 || from == to) return 0;
  hipLaunchKernelGGL(k_status_remap, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), (int)from, (int)to);
  HIPCHK(hipGetLastError());
  return 0;
}

int odr_seafloor(odr_ctx *c, odr_particles *p, int64_t *n_below) {
  return odr_seafloor_action(c, p, ODR_SEAFLOOR_LIFT, 0, n_below);
}

int odr_deactivate(odr_ctx *c, odr_particles *p, const uint8_t *mask, int32_t code) {
  p->status_epoch++;
  p->epoch++;  // invalidates the cached reductions (reduce())
  REQUIRE(mask, "mask NULL");
  if (p->n == 0) return 0;
  void *s;
  int rc = scratch(c, p, (size_t)p->n, &s);
  if (rc) return rc;
  H2D(s, mask, (size_t)p->n);
  HIPCHK(hipStreamSynchronize(c->stream));
  hipLaunchKernelGGL(k_deactivate, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), (const unsigned char *)s, code);
  HIPCHK(hipGetLastError());
  return 0;
}

// deactivate_outside (basemodel/__init__.py:2354-2382); NaN bound = not set
int odr_deactivate_outside(odr_ctx *c, odr_particles *p, double west, double east, double south, double north,
                           int32_t code) {
  p->status_epoch++;
  p->epoch++;
  if (p->n == 0) return 0;
  const int uW = west == west, uE = east == east, uS = south == south, uN = north == north;
  if (!(uW || uE || uS || uN)) return 0;
  hipLaunchKernelGGL(k_deactivate_outside, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, view(p), west, east, south, north,
                     uW, uE, uS, uN, (uE && east > 180.0) ? 1 : 0, code);
  HIPCHK(hipGetLastError());
  return 0;
}

static int ensure_alt(odr_particles *p) {
  size_t cap = (size_t)p->cap;
  // lazily allocate the ping-pong set and the deactivated store
  for (int k = 0; k < 7; ++k) if (!p->alt64[k]) HIPCHK(hipMalloc((void **)&p->alt64[k], 8 * cap));
  for (int k = 0; k < 3; ++k) {
    if (!p->alti32[k]) HIPCHK(hipMalloc((void **)&p->alti32[k], 4 * cap));
    if (!p->dead64[k]) HIPCHK(hipMalloc((void **)&p->dead64[k], 8 * cap));
  }
  for (int k = 0; k < 4; ++k) if (!p->altf32[k]) HIPCHK(hipMalloc((void **)&p->altf32[k], 4 * cap));
  for (int k = 0; k < 2; ++k) if (!p->deadi32[k]) HIPCHK(hipMalloc((void **)&p->deadi32[k], 4 * cap));
  for (int k = 0; k < NVAR; ++k) if (p->env[k] && !p->altenv[k]) HIPCHK(hipMalloc((void **)&p->altenv[k], 4 * cap));
  for (int k = 0; k < 9; ++k) if (p->aux[k] && !p->altaux[k]) HIPCHK(hipMalloc((void **)&p->altaux[k], 4 * cap));
  return 0;
}

// with_env = false: without the environment and the sample position (slon / slat), which the next environment sample
// rewrites for every element -- a re-sort at the top of a step moves 68 instead of ~130 bytes per particle
static void all_arrays(odr_particles *p, CmpArrays &A, bool with_env = true) {
  memset(&A, 0, sizeof A);
  const int n64 = with_env ? 7 : 5;
  for (int k = 0; k < n64; ++k) {
    A.src64[k] = p->d64[k]; A.dst64[k] = p->alt64[k];
    A.dead64[k] = k < 3 ? p->dead64[k] : nullptr;
  }
  A.n64 = n64;
  int m = 0;
  for (int k = 0; k < 3; ++k, ++m) { A.src32[m] = p->i32[k]; A.dst32[m] = p->alti32[k]; A.dead32[m] = k < 2 ? p->deadi32[k] : nullptr; }
  for (int k = 0; k < 4; ++k, ++m) { A.src32[m] = (const int *)p->f32[k]; A.dst32[m] = (int *)p->altf32[k]; }
  if (with_env)
    for (int k = 0; k < NVAR; ++k)
      if (p->env[k]) { A.src32[m] = (const int *)p->env[k]; A.dst32[m] = (int *)p->altenv[k]; ++m; }
  for (int k = 0; k < 9; ++k)
    if (p->aux[k]) { A.src32[m] = (const int *)p->aux[k]; A.dst32[m] = (int *)p->altaux[k]; ++m; }
  A.n32 = m;
}

static void swap_sets(odr_particles *p, bool with_env = true) {
  for (int k = 0; k < (with_env ? 7 : 5); ++k) std::swap(p->d64[k], p->alt64[k]);
  for (int k = 0; k < 3; ++k) std::swap(p->i32[k], p->alti32[k]);
  for (int k = 0; k < 4; ++k) std::swap(p->f32[k], p->altf32[k]);
  if (with_env) for (int k = 0; k < NVAR; ++k) if (p->env[k]) std::swap(p->env[k], p->altenv[k]);
  for (int k = 0; k < 9; ++k) if (p->aux[k]) std::swap(p->aux[k], p->altaux[k]);
}

// First half of odr_compact: how many elements stay, and which provisional status numbers (100 + k -> bit k of *flags)
// are present -- one kernel pair, ONE host read for both (the run() loop needs both every step).
int odr_scan_status(odr_ctx *c, odr_particles *p, int64_t *n_kept, uint64_t *flags) {
  HIPCHK(hipSetDevice(c->device));
  if (n_kept) *n_kept = p->n;
  if (flags) *flags = 0;
  if (p->n == 0) return 0;
  unsigned nb = nblk(p->n);
  if (p->wcount && p->wcount_epoch == p->status_epoch && p->wcount_n == p->n) {
    // the step launch that ran last counted as it went (StepDesc.wcount; counter[1] zeroed, counter[2] collected by that call)
    int rc = odr_scan_status_begin(c, p);
    if (rc) return rc < 0 ? rc : fail(ODR_ERR_STATE, "odr_scan_status: the counts of the step launch are gone");
    return odr_scan_status_end(c, p, n_kept, flags);
  } else {
    HIPCHK(hipMemsetAsync(c->counter + 1, 0, 2 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_cmp_count, dim3(std::min(nb, 2048u)), dim3(BLOCK), 0, c->stream, p->i32[1], p->n, p->bcount, c->counter + 2, c->counter + 1);
  }
  unsigned long long out[2];
  D2H(out, c->counter + 1, sizeof out);
  HIPCHK(hipStreamSynchronize(c->stream));
  if (n_kept) *n_kept = (int64_t)out[0];
  if (flags) *flags = out[1];
  p->scan_kept = (long long)out[0];
  p->scan_epoch = p->status_epoch;
  return 0;
}

// odr_scan_status in two halves (include/odrift.h): the fold enqueued, its result read later -- the launch the loop makes in
// between (a guarded odr_vmix) does not wait for the host.
int odr_scan_status_begin(odr_ctx *c, odr_particles *p) {
  HIPCHK(hipSetDevice(c->device));
  if (!(p->n > 0 && p->wcount && p->wcount_epoch == p->status_epoch && p->wcount_n == p->n)) return 1;   // no counts of a step launch
  p->wcount_epoch = ~0ull;   // (used once: a second fold accumulates into a counter nobody zeroed)
  const long long nw = (p->n + 63) / 64;
  const unsigned wgs = (unsigned)std::min<long long>(16, ((nw + 3) / 4 + 4095) / 4096);   // (4 chunks per thread and pass)
  hipLaunchKernelGGL(k_cmp_total, dim3(wgs ? wgs : 1), dim3(1024), 0, c->stream, p->wcount, nw, p->bcount, c->counter + 1, c->scan_host,
                     (long long)p->n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(c->scan_ev, c->stream));
  c->scan_open = true;
  return 0;
}

int odr_scan_status_end(odr_ctx *c, odr_particles *p, int64_t *n_kept, uint64_t *flags) {
  if (!c->scan_open) return fail(ODR_ERR_STATE, "odr_scan_status_end without odr_scan_status_begin");
  c->scan_open = false;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipEventSynchronize(c->scan_ev));
  if (n_kept) *n_kept = (int64_t)c->scan_host[0];
  if (flags) *flags = c->scan_host[1];
  p->scan_kept = (long long)c->scan_host[0];
  p->scan_epoch = p->status_epoch;   // (a guarded mixing launch between the halves deactivates nothing the fold has not seen: it
                                     // runs only when every element stays, and then there is nothing to compact)
  return 0;
}

// The main-loop samples that follow treat the element positions as the reference's float32 arrays of the first
// get_environment of a run (DevWorld::f32pos); 0 ends it.
int odr_ctx_set_position_class(odr_ctx *c, int f32) {
  const int v = f32 ? 1 : 0;
  if ((c->hw.f32pos & 1) != v) { c->hw.f32pos = (c->hw.f32pos & ~1) | v; c->dirty = true; }
  return 0;
}
// Coordinate arrays of a grid source: float32 or not (the index maps of a geographic reader in the float32 position class)
int odr_source_set_coordinate_dtype(odr_ctx *c, int32_t sid, int x_is_f32, int y_is_f32) {
  REQUIRE(sid >= 0 && sid < c->nsrc && c->hw.src[sid].kind == SRC_GRID, "source %d is not a grid source", sid);
  c->hw.src[sid].xy_f32 = (x_is_f32 ? 1 : 0) | (y_is_f32 ? 2 : 0);
  c->dirty = true;
  return 0;
}

int odr_ctx_guard_next_vmix(odr_ctx *c, int on) {
  c->guard_next_vmix = on ? 1 : 0;
  return 0;
}

// Second half: remove the deactivated elements using the counts of the last odr_scan_status (no host read).  Valid as
// long as no element changed between active and deactivated since the scan (renumbering statuses is fine).
int odr_compact_apply(odr_ctx *c, odr_particles *p, int64_t *n_active) {
  // the cached reductions run over the ACTIVE elements only, which a compaction neither changes nor removes: a valid
  // cache stays valid (the step launch forms the movers' tests, the compaction comes between it and the movers)
  const bool keep_red = !p->external && c->red_owner == p && c->red_epoch == p->epoch && !c->red_extents;
  p->epoch++;
  if (keep_red) c->red_epoch = p->epoch;
  HIPCHK(hipSetDevice(c->device));
  if (p->n == 0) { if (n_active) *n_active = 0; return 0; }
  if (p->scan_epoch != p->status_epoch || p->scan_kept < 0)
    return fail(ODR_ERR_STATE, "odr_compact_apply: elements were deactivated since the last odr_scan_status");
  const long long kept = p->scan_kept;
  p->scan_kept = -1;
  unsigned nb = nblk(p->n);
  if (kept == p->n) { if (n_active) *n_active = p->n; return 0; }  // "No elements to deactivate"
  int rc = ensure_alt(p);  // (allocates the deactivated store)
  if (rc) return rc;
  // exclusive scan of the per-block counts of the last odr_scan_status (only now that something is to be removed)
  hipLaunchKernelGGL(k_cmp_scan, dim3(1), dim3(1024), 0, c->stream, p->bcount, (long long)nb, c->counter + 3);
  CmpArrays A;
  all_arrays(p, A);
  long long removed = p->n - kept;
  if (getenv("ODR_ORDERED_COMPACT")) {  // order-preserving variant: rewrites every array
    hipLaunchKernelGGL(k_cmp_scatter, dim3(nb), dim3(BLOCK), 0, c->stream, p->i32[1], p->n, p->bcount, A, p->ndead);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    swap_sets(p);
    for (int v = 0; v < NVAR; ++v) p->env_cn[v] = std::min(p->env_cn[v], kept);
  } else {
    void *sc;
    if ((rc = scratch(c, p, sizeof(unsigned) * 2 * (size_t)removed, &sc))) return rc;
    unsigned *holes = (unsigned *)sc, *fills = holes + removed;
    HIPCHK(hipMemsetAsync(sc, 0xFF, sizeof(unsigned) * 2 * (size_t)removed, c->stream));
    hipLaunchKernelGGL(k_cmp_lists, dim3(nb), dim3(BLOCK), 0, c->stream, p->i32[1], p->n, p->bcount, (long long)kept, A,
                       p->ndead, holes, fills);
    hipLaunchKernelGGL(k_cmp_move, dim3(nblk(removed)), dim3(BLOCK), 0, c->stream, holes, fills, removed, A);
    HIPCHK(hipGetLastError());
  }
  p->ndead += removed;
  p->n = kept;
  // (the tile step's workgroup table stays valid: an element that filled a hole outside its sort tile is outside its workgroup's
  // node rectangle and is handed to k_step_list -- tests/test_gpu_tile.py::test_tile_step_with_a_small_lds_budget_and_compaction;
  // the table dies with the GEOMETRY it was built on: odr_source_release, src_gen)
  if (n_active) *n_active = p->n;
  return 0;
}

int odr_compact(odr_ctx *c, odr_particles *p, int64_t *n_active) {
  int rc = odr_scan_status(c, p, nullptr, nullptr);
  if (rc) return rc;
  return odr_compact_apply(c, p, n_active);
}

// Re-order the particle arrays by the grid cell of one gridded reader (see k_sort_hist).
int odr_sort_particles(odr_ctx *c, odr_particles *p, int32_t sid) { return odr_sort_particles_ex(c, p, sid, 1); }

// keep_environment = 0: the sampled environment and the sample position are NOT carried along (they are left in the old
// order, i.e. meaningless): for a re-sort at the top of a step, whose environment sample rewrites them anyway.
// Environment variables that hold one value for every element stay valid.
int odr_sort_particles_ex(odr_ctx *c, odr_particles *p, int32_t sid, int keep_environment) {
  const bool with_env = keep_environment != 0;
  REQUIRE(sid >= 0 && sid < c->nsrc && c->hw.src[sid].kind == SRC_GRID && c->hw.src[sid].nlevels > 0,
          "source %d is not a gridded source with a resident block", sid);
  HIPCHK(hipSetDevice(c->device));
  if (p->n < 2) return 0;
  SlowSpan sp_all("odr_sort_particles_ex (whole call)");
  int rc = flush_world(c);
  if (rc) return rc;
  if ((rc = ensure_alt(p))) return rc;
  const DevSource &s = c->hw.src[sid];
  int slot = s.level_slot[0];
  const DevBlock &b = s.slot[slot];
  int ntx = (b.nx + 7) / 8, nty = (b.ny + 7) / 8;
  // ODR_SORT_ZBANDS=<levels per band> (3-D readers): depth bands inside every cell (sort_key)
  int lpb = getenv("ODR_SORT_ZBANDS") ? atoi(getenv("ODR_SORT_ZBANDS")) : 0, zb = 1;
  if (lpb > 0 && s.nz > 1 && (long long)ntx * nty * 64 * ((s.nz + lpb - 1) / lpb) < (1ll << 27)) zb = (s.nz + lpb - 1) / lpb;
  else lpb = 1;
  unsigned nbins = (unsigned)(ntx * nty * 64 * zb + 1);
  size_t n = (size_t)p->n;
  void *sc;
  const int ntiles = ntx * nty;
  if ((rc = scratch(c, p, sizeof(unsigned) * (2 * n + nbins + nbins / 1024 + 64 + (size_t)ntiles + 1 + (size_t)ntiles / 1024 + 64), &sc))) return rc;
  unsigned *keys = (unsigned *)sc, *perm = keys + n, *hist = perm + n;
  unsigned *wg_nw = hist + nbins + nbins / 1024 + 64;
  HIPCHK(hipMemsetAsync(hist, 0, sizeof(unsigned) * nbins, c->stream));
  hipLaunchKernelGGL(k_sort_hist, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, c->dw, sid, slot, view(p), ntx, nbins,
                     keys, hist, zb, lpb);
  {
    unsigned nsb = (nbins + 1023) / 1024;
    unsigned *bsum = hist + nbins;  // scratch tail
    hipLaunchKernelGGL(k_scan_local, dim3(nsb), dim3(1024), 0, c->stream, hist, (long long)nbins, bsum);
    hipLaunchKernelGGL(k_cmp_scan, dim3(1), dim3(1024), 0, c->stream, bsum, (long long)nsb, c->counter + 2);
    hipLaunchKernelGGL(k_scan_add, dim3(nsb), dim3(1024), 0, c->stream, hist, (long long)nbins, bsum);
  }
  hipLaunchKernelGGL(k_sort_perm, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, keys, p->n, hist, perm);
  {   // workgroup table for k_step_tile (odr_tile.hip.h): hist[] now holds the END offset of every key
    p->wg_valid = false;
    const long long cap = std::min<long long>((long long)ntiles + 1, p->n) + p->n / BLOCK + 1;
    if (!p->wg_total) {
      HIPCHK(hipMalloc((void **)&p->wg_total, sizeof(unsigned long long) * 4));
      HIPCHK(hipMemsetAsync(p->wg_total, 0, sizeof(unsigned long long) * 4, c->stream));
      p->wg_stats = p->wg_total + 2;
    }
    if (p->wg_cap < cap) {
      if (p->wg_tab) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(p->wg_tab)); p->wg_tab = nullptr; }
      HIPCHK(hipMalloc((void **)&p->wg_tab, sizeof(unsigned) * 2 * (size_t)cap));
      p->wg_cap = cap;
    }
    const unsigned nt1 = (unsigned)ntiles + 1;
    hipLaunchKernelGGL(k_wg_count, dim3((nt1 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, hist, ntiles, 64u * (unsigned)zb, wg_nw);
    hipLaunchKernelGGL(k_cmp_scan, dim3(1), dim3(1024), 0, c->stream, wg_nw, (long long)nt1, p->wg_total);   // exclusive, in place
    hipLaunchKernelGGL(k_wg_fill, dim3((nt1 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, hist, ntiles, 64u * (unsigned)zb, wg_nw, p->wg_tab, (unsigned)cap);
    p->wg_grid = cap; p->wg_n = p->n; p->wg_sid = sid; p->wg_src_gen = c->src_gen[sid]; p->wg_valid = true;
  }
  CmpArrays A;
  all_arrays(p, A, with_env);
  hipLaunchKernelGGL(k_gather_perm, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, perm, p->n, A);
  HIPCHK(hipGetLastError());
  swap_sets(p, with_env);     // host pointers only: everything after this call is ordered behind the gather on the stream
  for (int v = 0; v < NVAR; ++v) {
    if (with_env) p->env_cn[v] = std::min(p->env_cn[v], p->n);
    else if (!(p->env_cok[v] && p->env_cn[v] >= p->n)) p->env_cok[v] = false;   // not carried: only a constant survives
  }
  p->status_epoch++;
  return 0;
}

// ------------------------------------------------------------------ output history
// state_to_buffer replacement: the float32 result buffer of run()
// (basemodel/__init__.py:2084-2105) lives in HBM; odr_history_record is one scatter kernel on the
// context stream; odr_history_flush extracts every variable in the reference's (trajectory, time)
// layout on a second stream into pinned host memory, overlapping with the following steps.
struct odr_history {
  long long ntraj;
  int ntimes, nvars, stride;
  int codes[HIST_MAXV];
  float *buf;            // [ntimes][ntraj][stride]
  float *stage;          // device staging of one extracted variable chunk
  size_t stage_floats;
  float *host;           // pinned: [nvars][ntraj][flush_nt]
  size_t host_floats;
  int flush_nt;
  hipStream_t copy_stream;
  hipEvent_t recorded, flushed;
  double *red;           // 2 doubles
  long long id_base;     // element ID of trajectory row 0 (odr_history_set_id_base)
};

static int hist_fill_nan(odr_ctx *c, odr_history *h, hipStream_t st) {
  size_t n = (size_t)h->ntimes * (size_t)h->ntraj * (size_t)h->stride;
  hipLaunchKernelGGL(k_fill_u32, dim3(4096), dim3(BLOCK), 0, st, (unsigned *)h->buf, n, 0x7FC00000u);
  HIPCHK(hipGetLastError());
  return 0;
}

int odr_history_create(odr_ctx *c, int64_t n_trajectories, int32_t n_times, int32_t nvars, const int32_t *var_codes,
                       odr_history **out) {
  REQUIRE(n_trajectories > 0 && n_times > 0 && nvars > 0 && nvars <= HIST_MAXV && var_codes && out, "bad history shape");
  for (int k = 0; k < nvars; ++k) {
    int cde = var_codes[k];
    bool ok = (cde >= 0 && cde < NVAR) || (cde >= ODR_HIST_LON && cde <= ODR_HIST_TERMINAL_VELOCITY) ||
              (cde >= ODR_HIST_PROPERTY0 && cde < ODR_HIST_PROPERTY0 + 9);
    REQUIRE(ok, "unknown history variable code %d", cde);
  }
  HIPCHK(hipSetDevice(c->device));
  odr_history *h = new odr_history();
  memset(h, 0, sizeof *h);
  h->ntraj = n_trajectories; h->ntimes = n_times; h->nvars = nvars; h->stride = (nvars + 3) & ~3;
  for (int k = 0; k < nvars; ++k) h->codes[k] = var_codes[k];
  HIPCHK(hipMalloc((void **)&h->buf, sizeof(float) * (size_t)n_times * (size_t)n_trajectories * (size_t)h->stride));
  HIPCHK(hipMalloc((void **)&h->red, 2 * sizeof(double)));
  HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&h->recorded, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&h->flushed, hipEventDisableTiming));
  int rc = hist_fill_nan(c, h, c->stream);
  if (rc) return rc;
  *out = h;
  return 0;
}

// the buffer holds the trajectories [id_base, id_base + n_trajectories): the shard of one rank of a sharded run
int odr_history_set_id_base(odr_ctx *, odr_history *h, int64_t id_base) {
  h->id_base = id_base;
  return 0;
}

int odr_history_destroy(odr_ctx *c, odr_history *h) {
  if (!h) return 0;
  HIPCHK(hipSetDevice(c->device));
  (void)hipStreamSynchronize(h->copy_stream);
  (void)hipStreamSynchronize(c->stream);
  if (h->buf) (void)hipFree(h->buf);
  if (h->stage) (void)hipFree(h->stage);
  if (h->host) (void)hipHostFree(h->host);
  if (h->red) (void)hipFree(h->red);
  (void)hipStreamDestroy(h->copy_stream);
  (void)hipEventDestroy(h->recorded);
  (void)hipEventDestroy(h->flushed);
  delete h;
  return 0;
}

int odr_history_record(odr_ctx *c, odr_particles *p, odr_history *h, int32_t time_index, int only_deactivated,
                       int position_from_previous) {
  REQUIRE(h && time_index >= 0 && time_index < h->ntimes, "time index %d outside the buffer (%d)", time_index, h ? h->ntimes : 0);
  if (p->n == 0) return 0;
  HistVars H;
  memset(&H, 0, sizeof H);
  H.nvars = h->nvars; H.stride = h->stride;
  for (int k = 0; k < h->nvars; ++k) {
    int cde = h->codes[k];
    const void *src = nullptr;
    int kind = HK_F32;
    if (cde >= 0 && cde < NVAR) src = p->env[cde];
    else if (cde == ODR_HIST_LON) { src = p->d64[(position_from_previous & 1) ? 3 : 0]; kind = HK_F64; }
    else if (cde == ODR_HIST_LAT) { src = p->d64[(position_from_previous & 1) ? 4 : 1]; kind = HK_F64; }
    else if (cde == ODR_HIST_Z) { src = p->d64[2]; kind = HK_F64; }
    else if (cde == ODR_HIST_STATUS) { src = p->i32[1]; kind = HK_I32; }
    else if (cde == ODR_HIST_MOVING) { src = p->i32[2]; kind = HK_I32; }
    else if (cde == ODR_HIST_AGE_SECONDS) src = p->f32[3];
    else if (cde == ODR_HIST_WIND_DRIFT_FACTOR) src = p->f32[0];
    else if (cde == ODR_HIST_CURRENT_DRIFT_FACTOR) src = p->f32[1];
    else if (cde == ODR_HIST_TERMINAL_VELOCITY) src = p->f32[2];
    else {
      const int slot = cde - ODR_HIST_PROPERTY0;
      REQUIRE(slot >= 0 && slot < 9, "history variable %d: no such property slot", cde);
      src = ((position_from_previous & ODR_HIST_PROPERTIES_FROM_SNAPSHOT) && p->aux_snap[slot] && p->aux_snap_cap[slot] >= p->n) ? p->aux_snap[slot] : p->aux[slot];
    }
    if (!src) return fail(ODR_ERR_STATE, "history variable %d has not been sampled / set", cde);
    H.src[k] = src; H.kind[k] = kind;
  }
  // a flush in flight reads the buffer on the copy stream: records of later time slots may proceed,
  // but never overwrite a slot before its flush has finished
  HIPCHK(hipStreamWaitEvent(c->stream, h->flushed, 0));
  float *slab = h->buf + (size_t)time_index * (size_t)h->ntraj * (size_t)h->stride;
#define HIST_REC(NQ) hipLaunchKernelGGL(k_hist_record<NQ>, dim3(nblk(p->n)), dim3(BLOCK), 0, c->stream, p->n, p->i32[0], p->i32[1], \
                                        H, slab, h->ntraj, only_deactivated, h->id_base)
  switch (h->stride / 4) {
    case 1: HIST_REC(1); break;
    case 2: HIST_REC(2); break;
    case 3: HIST_REC(3); break;
    case 4: HIST_REC(4); break;
    case 5: HIST_REC(5); break;
    case 6: HIST_REC(6); break;
    case 7: HIST_REC(7); break;
    case 8: HIST_REC(8); break;
    case 9: HIST_REC(9); break;
    default: HIST_REC(10); break;
  }
#undef HIST_REC
  HIPCHK(hipGetLastError());
  return 0;
}

// Start copying time slots [t0, t0+nt) of every variable to pinned host memory in the reference's
// (trajectory, time) layout; returns immediately.  odr_history_wait blocks until the copy is done;
// odr_history_host_ptr gives the pinned array of one variable ([ntraj][nt] float32), valid until the next flush.
int odr_history_flush(odr_ctx *c, odr_history *h, int32_t t0, int32_t nt) {
  REQUIRE(h && t0 >= 0 && nt > 0 && t0 + nt <= h->ntimes, "bad time range");
  HIPCHK(hipSetDevice(c->device));
  size_t per_var = (size_t)h->ntraj * (size_t)nt;
  if (h->host_floats < per_var * (size_t)h->nvars) {
    HIPCHK(hipStreamSynchronize(h->copy_stream));
    if (h->host) HIPCHK(hipHostFree(h->host));
    HIPCHK(hipHostMalloc((void **)&h->host, sizeof(float) * per_var * (size_t)h->nvars, hipHostMallocDefault));
    h->host_floats = per_var * (size_t)h->nvars;
  }
  h->flush_nt = nt;
  const size_t chunk_tr = std::max<size_t>(1, std::min<size_t>((size_t)h->ntraj, ((size_t)1 << 27) / (size_t)nt));  // <= 512 MiB staging
  if (h->stage_floats < chunk_tr * (size_t)nt * 2) {
    HIPCHK(hipStreamSynchronize(h->copy_stream));
    if (h->stage) HIPCHK(hipFree(h->stage));
    HIPCHK(hipMalloc((void **)&h->stage, sizeof(float) * chunk_tr * (size_t)nt * 2));  // double buffered
    h->stage_floats = chunk_tr * (size_t)nt * 2;
  }
  HIPCHK(hipEventRecord(h->recorded, c->stream));
  HIPCHK(hipStreamWaitEvent(h->copy_stream, h->recorded, 0));
  int flip = 0;
  for (int v = 0; v < h->nvars; ++v)
    for (size_t tr0 = 0; tr0 < (size_t)h->ntraj; tr0 += chunk_tr) {
      size_t ntr = std::min(chunk_tr, (size_t)h->ntraj - tr0);
      float *st = h->stage + (size_t)flip * chunk_tr * (size_t)nt;
      flip ^= 1;
      hipLaunchKernelGGL(k_hist_extract, dim3(nblk((long long)(ntr * nt))), dim3(BLOCK), 0, h->copy_stream, h->buf, h->ntraj,
                         h->stride, v, t0, nt, (long long)tr0, (long long)ntr, st);
      HIPCHK(hipMemcpyAsync(h->host + (size_t)v * per_var + tr0 * (size_t)nt, st, sizeof(float) * ntr * (size_t)nt,
                            hipMemcpyDeviceToHost, h->copy_stream));
    }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(h->flushed, h->copy_stream));
  return 0;
}

int odr_history_wait(odr_ctx *c, odr_history *h) {
  REQUIRE(h, "history NULL");
  HIPCHK(hipStreamSynchronize(h->copy_stream));
  return 0;
}

int odr_history_host_ptr(odr_ctx *c, odr_history *h, int32_t var_index, float **ptr, int32_t *nt) {
  REQUIRE(h && var_index >= 0 && var_index < h->nvars && ptr, "bad variable index");
  if (!h->host) return fail(ODR_ERR_STATE, "odr_history_flush has not been called");
  *ptr = h->host + (size_t)var_index * (size_t)h->ntraj * (size_t)h->flush_nt;
  if (nt) *nt = h->flush_nt;
  return 0;
}

// NaN-fill the buffer for the next export_buffer_length output times (:2493-2499)
int odr_history_reset(odr_ctx *c, odr_history *h) {
  REQUIRE(h, "history NULL");
  HIPCHK(hipStreamWaitEvent(c->stream, h->flushed, 0));
  return hist_fill_nan(c, h, c->stream);
}

// var.min(skipna=True), var.max(skipna=True) over the whole buffer (:2409-2414); NaN when nothing was written
int odr_history_minmax(odr_ctx *c, odr_history *h, int32_t var_index, double *minval, double *maxval) {
  REQUIRE(h && var_index >= 0 && var_index < h->nvars && minval && maxval, "bad arguments");
  double init[2] = {-INFINITY, -INFINITY}, r[2];
  H2D(h->red, init, sizeof init);
  long long nrec = (long long)h->ntimes * h->ntraj;
  hipLaunchKernelGGL(k_hist_minmax, dim3(2048), dim3(BLOCK), 0, c->stream, h->buf, nrec, h->stride, var_index, h->red);
  HIPCHK(hipGetLastError());
  D2H(r, h->red, sizeof r);
  HIPCHK(hipStreamSynchronize(c->stream));
  *minval = r[0] == -INFINITY ? NAN : -r[0];
  *maxval = r[1] == -INFINITY ? NAN : r[1];
  return 0;
}

// ------------------------------------------------------------------ ROMS sigma grid
// the sigma -> z regridding of ROMS-native blocks on the device.  An odr_sgrid holds
// the depths of the s-levels of one grid (z_rho, computed once like the reader's z_rho_tot); odr_sgrid_zslice
// regrids one 3-D variable per call into a float32 [kmax][ny][nx] device array that odr_block_upload_device takes.
struct odr_sgrid {
  int N, ny, nx;
  double *zr;       // [N][ny*nx]
  double *Z;        // device copy of the last z levels
  int kmax_cap;
  void *fbuf;       // device staging of a host-supplied field
  size_t fbuf_bytes;
  float *out32[8];  // result slots (several regridded variables alive for one block upload)
  size_t out_elems[8];
  double *out64;
  size_t out64_elems;
};

int odr_sgrid_create(odr_ctx *c, int32_t ny, int32_t nx, int32_t N, const double *H, const double *zeta, double Hc,
                     const double *Cs, const double *S, int32_t vtransform, odr_sgrid **out) {
  REQUIRE(ny > 0 && nx > 0 && N >= 2 && N <= 4096 && H && Cs && out, "bad s-grid arguments");
  REQUIRE(vtransform == 1 || vtransform == 2, "Unknown Vtransform");
  HIPCHK(hipSetDevice(c->device));
  odr_sgrid *g = new odr_sgrid();
  memset(g, 0, sizeof *g);
  g->N = N; g->ny = ny; g->nx = nx;
  const long long M = (long long)ny * nx;
  std::vector<double> s((size_t)N);
  for (int k = 0; k < N; ++k) s[(size_t)k] = S ? S[k] : -1.0 + ((double)k + 0.5) / (double)N;   // depth.py:92-100
  double *dH, *dz = nullptr, *dC, *dS;
  int *flag;
  HIPCHK(hipMalloc((void **)&dH, sizeof(double) * (size_t)M));
  HIPCHK(hipMalloc((void **)&dC, sizeof(double) * (size_t)N));
  HIPCHK(hipMalloc((void **)&dS, sizeof(double) * (size_t)N));
  HIPCHK(hipMalloc((void **)&flag, sizeof(int)));
  HIPCHK(hipMalloc((void **)&g->zr, sizeof(double) * (size_t)M * (size_t)N));
  H2D(dH, H, sizeof(double) * (size_t)M);
  H2D(dC, Cs, sizeof(double) * (size_t)N);
  H2D(dS, s.data(), sizeof(double) * (size_t)N);
  HIPCHK(hipMemsetAsync(flag, 0, sizeof(int), c->stream));
  if (zeta) {
    HIPCHK(hipMalloc((void **)&dz, sizeof(double) * (size_t)M));
    H2D(dz, zeta, sizeof(double) * (size_t)M);
  }
  hipLaunchKernelGGL(k_roms_zrho, dim3(nblk(M)), dim3(BLOCK), 0, c->stream, dH, dz, Hc, dC, dS, N, M, vtransform, g->zr);
  hipLaunchKernelGGL(k_roms_zrho_positive, dim3(nblk(M * N)), dim3(BLOCK), 0, c->stream, g->zr, M * N, flag, 0);
  int hflag = 0;
  D2H(&hflag, flag, sizeof(int));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (hflag) hipLaunchKernelGGL(k_roms_zrho_positive, dim3(nblk(M * N)), dim3(BLOCK), 0, c->stream, g->zr, M * N, flag, 1);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  (void)hipFree(dH); (void)hipFree(dC); (void)hipFree(dS); (void)hipFree(flag);
  if (dz) (void)hipFree(dz);
  *out = g;
  return 0;
}

int odr_sgrid_destroy(odr_ctx *c, odr_sgrid *g) {
  if (!g) return 0;
  HIPCHK(hipSetDevice(c->device));
  (void)hipStreamSynchronize(c->stream);
  if (g->zr) (void)hipFree(g->zr);
  if (g->Z) (void)hipFree(g->Z);
  if (g->fbuf) (void)hipFree(g->fbuf);
  for (int k = 0; k < 8; ++k) if (g->out32[k]) (void)hipFree(g->out32[k]);
  if (g->out64) (void)hipFree(g->out64);
  delete g;
  return 0;
}

int odr_sgrid_download_zrho(odr_ctx *c, odr_sgrid *g, double *out) {
  REQUIRE(g && out, "bad arguments");
  D2H(out, g->zr, sizeof(double) * (size_t)g->N * g->ny * g->nx);
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// field: [N][ny][nx] float32 (is_f64 = 0) or float64, on the host or (on_device) in device memory.
// Results: *out_dev32 = device float32 [kmax][ny][nx] (valid until the next call on this s-grid; feed it to
// odr_block_upload_device); out_host64 (optional) receives the float64 values the reference computes.
int odr_sgrid_zslice(odr_ctx *c, odr_sgrid *g, const void *field, int is_f64, int on_device, int32_t kmax,
                     const double *Z, int32_t out_slot, void **out_dev32, double *out_host64) {
  REQUIRE(g && field && kmax > 0 && Z && out_slot >= 0 && out_slot < 8, "bad arguments");
  HIPCHK(hipSetDevice(c->device));
  const long long M = (long long)g->ny * g->nx;
  const size_t fbytes = (is_f64 ? 8 : 4) * (size_t)M * (size_t)g->N;
  const void *dF = field;
  if (!on_device) {
    if (g->fbuf_bytes < fbytes) {
      HIPCHK(hipStreamSynchronize(c->stream));
      if (g->fbuf) HIPCHK(hipFree(g->fbuf));
      HIPCHK(hipMalloc(&g->fbuf, fbytes));
      g->fbuf_bytes = fbytes;
    }
    H2D(g->fbuf, field, fbytes);
    dF = g->fbuf;
  }
  if (g->kmax_cap < kmax) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (g->Z) HIPCHK(hipFree(g->Z));
    HIPCHK(hipMalloc((void **)&g->Z, sizeof(double) * (size_t)kmax));
    g->kmax_cap = kmax;
  }
  H2D(g->Z, Z, sizeof(double) * (size_t)kmax);
  const size_t oel = (size_t)M * (size_t)kmax;
  if (g->out_elems[out_slot] < oel) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (g->out32[out_slot]) HIPCHK(hipFree(g->out32[out_slot]));
    HIPCHK(hipMalloc((void **)&g->out32[out_slot], sizeof(float) * oel));
    g->out_elems[out_slot] = oel;
  }
  if (out_host64 && g->out64_elems < oel) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (g->out64) HIPCHK(hipFree(g->out64));
    HIPCHK(hipMalloc((void **)&g->out64, sizeof(double) * oel));
    g->out64_elems = oel;
  }
  double *o64 = out_host64 ? g->out64 : nullptr;
  // z levels in chunks of at most 64 (compile-time register arrays of 8 / 16 / 32 / 64 counters)
  for (int j0 = 0; j0 < kmax; j0 += 64) {
    const int kc = std::min(64, kmax - j0);
    float *o32 = g->out32[out_slot] + (size_t)j0 * (size_t)M;
    double *o64c = o64 ? o64 + (size_t)j0 * (size_t)M : nullptr;
#define ZSL(T, K) hipLaunchKernelGGL((k_roms_zslice<T, K>), dim3(nblk(M)), dim3(BLOCK), 0, c->stream, (const T *)dF, g->zr, \
                                     g->Z + j0, g->N, kc, M, o32, o64c)
    if (is_f64) { if (kc <= 8) ZSL(double, 8); else if (kc <= 16) ZSL(double, 16); else if (kc <= 32) ZSL(double, 32); else ZSL(double, 64); }
    else { if (kc <= 8) ZSL(float, 8); else if (kc <= 16) ZSL(float, 16); else if (kc <= 32) ZSL(float, 32); else ZSL(float, 64); }
#undef ZSL
  }
  HIPCHK(hipGetLastError());
  if (out_host64) {
    D2H(out_host64, g->out64, sizeof(double) * oel);
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  if (out_dev32) *out_dev32 = g->out32[out_slot];
  return 0;
}
