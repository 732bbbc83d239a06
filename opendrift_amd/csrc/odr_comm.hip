// libodrift_hip.so, translation unit 9: the collectives of a run sharded over the GPUs of a node over
// RCCL -- one process per GPU, ONE communicator pair per process, no torch in the process.  The reference has no multi-process
// mode (docs/source/performance.rst:22,36: "run several simulations side by side"); what a sharded step exchanges is stated in
// include/odrift.h.  librccl.so.1 is opened on first use (dlopen): a one-GPU run never loads it, and a process that also holds
// torch (the CPU rehearsal tests over gloo) does not end up with two RCCL builds resolved against each other.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "odr_host.h"

namespace {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl R;

// The process's communicators.  `step`: the small collectives of the loop (the step summary, scalar reductions, metadata); `bulk`:
// reader levels.  Two, because operations on one communicator execute in the order they were issued: a 200 MB level in flight on
// the upload stream would hold the 200-byte summary of the step behind it.
struct Comm {
  bool on = false;
  int rank = 0, nranks = 1, device = 0, version = 0;
  uint64_t id_hash = 0;
  ncclComm_t step = nullptr, bulk = nullptr;
  hipStream_t stream = nullptr;          // the small collectives' stream
  hipEvent_t done = nullptr;
  double *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;   // device staging / page-locked mirrors
  size_t cap = 0;                        // doubles per rank the staging holds
  unsigned char *d_bytes = nullptr, *h_bytes = nullptr;
  size_t bytes_cap = 0;
  int gather_n = -1;                     // doubles per rank of the collective in flight (odr_comm_allgather_begin), -1: none
  unsigned long long collectives = 0, bulk_collectives = 0;
} G;

int load_rccl() {
  if (R.lib) return 0;
  const char *names[] = {getenv("ODR_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    if (!n || !*n) continue;
    R.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (R.lib) break;
  }
  if (!R.lib) return fail(ODR_ERR_STATE, "librccl.so.1 could not be opened: %s", dlerror());
#define SYM(field, name)                                                                         \
  do {                                                                                           \
    *(void **)(&R.field) = dlsym(R.lib, name);                                                   \
    if (!R.field) return fail(ODR_ERR_STATE, "librccl: symbol %s is missing", name);             \
  } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(CommCount, "ncclCommCount"); SYM(CommUserRank, "ncclCommUserRank"); SYM(AllGather, "ncclAllGather");
  SYM(AllReduce, "ncclAllReduce"); SYM(Broadcast, "ncclBroadcast"); SYM(GetVersion, "ncclGetVersion");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  return 0;
}

#define NCCLCHK(x)                                                                              \
  do {                                                                                          \
    ncclResult_t r_ = (x);                                                                      \
    if (r_ != ncclSuccess) return fail(ODR_ERR_HIP, "%s: %s", #x, R.GetErrorString(r_));        \
  } while (0)
#define NEED_COMM() do { if (!G.on) return fail(ODR_ERR_STATE, "no communicator: odr_comm_init has not been called"); } while (0)

int ensure_doubles(size_t n) {
  if (n <= G.cap) return 0;
  HIPCHK(hipStreamSynchronize(G.stream));
  if (G.d_in) { HIPCHK(hipFree(G.d_in)); HIPCHK(hipFree(G.d_out)); HIPCHK(hipHostFree(G.h_in)); HIPCHK(hipHostFree(G.h_out)); }
  G.cap = std::max<size_t>(64, 2 * n);
  HIPCHK(hipMalloc((void **)&G.d_in, sizeof(double) * G.cap));
  HIPCHK(hipMalloc((void **)&G.d_out, sizeof(double) * G.cap * (size_t)G.nranks));
  HIPCHK(hipHostMalloc((void **)&G.h_in, sizeof(double) * G.cap, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void **)&G.h_out, sizeof(double) * G.cap * (size_t)G.nranks, hipHostMallocDefault));
  return 0;
}

// row[0] = elements that stay, row[1 + k] = provisional status number 100 + k occurred: what the fold of the step's status scan
// (k_cmp_total) left in the context's counter words -- read on the DEVICE, so that the collective does not wait for the host
__global__ void k_summary_row(double *row, const unsigned long long *work) {
  const int k = threadIdx.x;
  if (k == 0) row[0] = (double)work[0];
  else if (k <= 8) row[k] = (double)((work[1] >> (k - 1)) & 1ull);
}

}  // namespace

extern "C" {

int odr_device_count(int32_t *n) {
  REQUIRE(n, "NULL argument");
  int k = 0;
  if (hipGetDeviceCount(&k) != hipSuccess) { (void)hipGetLastError(); k = 0; }
  *n = k;
  return 0;
}

int odr_comm_unique_id(uint8_t *id) {
  REQUIRE(id, "NULL id");
  int rc = load_rccl();
  if (rc) return rc;
  static_assert(ODR_COMM_ID_BYTES == 2 * NCCL_UNIQUE_ID_BYTES, "two communicators");
  ncclUniqueId a, b;
  NCCLCHK(R.GetUniqueId(&a));
  NCCLCHK(R.GetUniqueId(&b));
  memcpy(id, a.internal, NCCL_UNIQUE_ID_BYTES);
  memcpy(id + NCCL_UNIQUE_ID_BYTES, b.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

int odr_comm_init(odr_ctx *c, const uint8_t *id, int32_t rank, int32_t nranks) {
  REQUIRE(c && id && nranks >= 1 && rank >= 0 && rank < nranks, "bad communicator arguments");
  if (G.on) return fail(ODR_ERR_STATE, "this process already holds a communicator (one process per GPU)");
  int rc = load_rccl();
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->device));
  ncclUniqueId a, b;
  memcpy(a.internal, id, NCCL_UNIQUE_ID_BYTES);
  memcpy(b.internal, id + NCCL_UNIQUE_ID_BYTES, NCCL_UNIQUE_ID_BYTES);
  NCCLCHK(R.CommInitRank(&G.step, nranks, a, rank));
  NCCLCHK(R.CommInitRank(&G.bulk, nranks, b, rank));
  int cnt = 0, me = -1;
  NCCLCHK(R.CommCount(G.step, &cnt));
  NCCLCHK(R.CommUserRank(G.step, &me));
  if (cnt != nranks || me != rank) return fail(ODR_ERR_STATE, "RCCL reports rank %d of %d, asked for %d of %d", me, cnt, rank, nranks);
  (void)R.GetVersion(&G.version);
  G.rank = rank; G.nranks = nranks; G.device = c->device;
  uint64_t h = 1469598103934665603ull;   // FNV-1a of the id: every rank of one job prints the same number
  for (int k = 0; k < ODR_COMM_ID_BYTES; ++k) { h ^= id[k]; h *= 1099511628211ull; }
  G.id_hash = h;
  HIPCHK(hipStreamCreateWithFlags(&G.stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&G.done, hipEventDisableTiming));
  G.on = true;
  G.cap = 0;
  return ensure_doubles(64);
}

int odr_comm_info(int32_t *rank, int32_t *nranks, uint64_t *id_hash, int32_t *rccl_version, uint64_t *collectives) {
  if (rank) *rank = G.on ? G.rank : 0;
  if (nranks) *nranks = G.on ? G.nranks : 0;     // 0: no communicator
  if (id_hash) *id_hash = G.id_hash;
  if (rccl_version) *rccl_version = G.version;
  if (collectives) *collectives = G.collectives + G.bulk_collectives;
  return 0;
}

int odr_comm_destroy(void) {
  if (!G.on) return 0;
  HIPCHK(hipSetDevice(G.device));
  HIPCHK(hipStreamSynchronize(G.stream));
  (void)R.CommDestroy(G.step);
  (void)R.CommDestroy(G.bulk);
  if (G.d_in) { (void)hipFree(G.d_in); (void)hipFree(G.d_out); (void)hipHostFree(G.h_in); (void)hipHostFree(G.h_out); }
  if (G.d_bytes) { (void)hipFree(G.d_bytes); (void)hipHostFree(G.h_bytes); }
  (void)hipEventDestroy(G.done);
  (void)hipStreamDestroy(G.stream);
  G = Comm();
  return 0;
}

// in place, blocking: a handful of float64 (counts: sum; extents: min / max)
int odr_allreduce_scalars(odr_ctx *c, double *values, int32_t n, int32_t op) {
  NEED_COMM();
  REQUIRE(values && n > 0 && op >= 0 && op <= 2, "bad arguments");
  if (G.gather_n >= 0) return fail(ODR_ERR_STATE, "odr_allreduce_scalars between odr_comm_allgather_begin and _end");
  (void)c;
  HIPCHK(hipSetDevice(G.device));
  int rc = ensure_doubles((size_t)n);
  if (rc) return rc;
  memcpy(G.h_in, values, sizeof(double) * (size_t)n);
  HIPCHK(hipMemcpyAsync(G.d_in, G.h_in, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, G.stream));
  static const ncclRedOp_t ops[3] = {ncclSum, ncclMin, ncclMax};
  NCCLCHK(R.AllReduce(G.d_in, G.d_out, (size_t)n, ncclDouble, ops[op], G.step, G.stream));
  HIPCHK(hipMemcpyAsync(G.h_out, G.d_out, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, G.stream));
  HIPCHK(hipStreamSynchronize(G.stream));
  memcpy(values, G.h_out, sizeof(double) * (size_t)n);
  G.collectives++;
  return 0;
}

// The ONE collective of a sharded step, in two halves: every rank's row of n float64, gathered as [nranks][n].  _begin only
// enqueues (the comm stream: copy in, all-gather, copy out into page-locked memory, event); _end waits for the event.
// from_scan != 0: row[0 .. 8] -- the elements that stay and the eight status flags -- are taken ON THE DEVICE from the fold of the
// step's status scan (between odr_scan_status_begin and _end): the collective starts when the device has folded the counts, not
// when the host has read them.
int odr_comm_allgather_begin(odr_ctx *c, const double *row, int32_t n, int32_t from_scan) {
  NEED_COMM();
  REQUIRE(c && row && n > 0 && (!from_scan || n >= 9), "bad arguments");
  if (G.gather_n >= 0) return fail(ODR_ERR_STATE, "odr_comm_allgather_begin: the previous collective has not been finished");
  if (from_scan && !c->scan_open) return fail(ODR_ERR_STATE, "odr_comm_allgather_begin(from_scan): no status scan is open");
  HIPCHK(hipSetDevice(G.device));
  int rc = ensure_doubles((size_t)n);
  if (rc) return rc;
  memcpy(G.h_in, row, sizeof(double) * (size_t)n);
  HIPCHK(hipMemcpyAsync(G.d_in, G.h_in, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, G.stream));
  if (from_scan) {
    HIPCHK(hipStreamWaitEvent(G.stream, c->scan_ev, 0));
    hipLaunchKernelGGL(k_summary_row, dim3(1), dim3(64), 0, G.stream, G.d_in, (const unsigned long long *)(c->counter + 1));
    HIPCHK(hipGetLastError());
  }
  NCCLCHK(R.AllGather(G.d_in, G.d_out, (size_t)n, ncclDouble, G.step, G.stream));
  HIPCHK(hipMemcpyAsync(G.h_out, G.d_out, sizeof(double) * (size_t)n * (size_t)G.nranks, hipMemcpyDeviceToHost, G.stream));
  HIPCHK(hipEventRecord(G.done, G.stream));
  G.gather_n = n;
  G.collectives++;
  return 0;
}

int odr_comm_allgather_end(odr_ctx *c, double *rows) {
  NEED_COMM();
  (void)c;
  REQUIRE(rows, "NULL rows");
  if (G.gather_n < 0) return fail(ODR_ERR_STATE, "odr_comm_allgather_end without odr_comm_allgather_begin");
  HIPCHK(hipEventSynchronize(G.done));
  memcpy(rows, G.h_out, sizeof(double) * (size_t)G.gather_n * (size_t)G.nranks);
  G.gather_n = -1;
  return 0;
}

// host bytes from `root` to every rank (metadata of a reader level: shapes, coordinates, content ids -- pickled by the host);
// blocking; every rank passes the same nbytes (the host broadcasts the length first, as 8 bytes)
int odr_comm_broadcast_bytes(odr_ctx *c, void *buf, int64_t nbytes, int32_t root) {
  NEED_COMM();
  (void)c;
  REQUIRE(buf && nbytes > 0 && root >= 0 && root < G.nranks, "bad arguments");
  HIPCHK(hipSetDevice(G.device));
  if ((size_t)nbytes > G.bytes_cap) {
    HIPCHK(hipStreamSynchronize(G.stream));
    if (G.d_bytes) { HIPCHK(hipFree(G.d_bytes)); HIPCHK(hipHostFree(G.h_bytes)); }
    G.bytes_cap = std::max<size_t>(4096, 2 * (size_t)nbytes);
    HIPCHK(hipMalloc((void **)&G.d_bytes, G.bytes_cap));
    HIPCHK(hipHostMalloc((void **)&G.h_bytes, G.bytes_cap, hipHostMallocDefault));
  }
  if (G.rank == root) {
    memcpy(G.h_bytes, buf, (size_t)nbytes);
    HIPCHK(hipMemcpyAsync(G.d_bytes, G.h_bytes, (size_t)nbytes, hipMemcpyHostToDevice, G.stream));
  }
  NCCLCHK(R.Broadcast(G.d_bytes, G.d_bytes, (size_t)nbytes, ncclUint8, root, G.step, G.stream));
  HIPCHK(hipMemcpyAsync(G.h_bytes, G.d_bytes, (size_t)nbytes, hipMemcpyDeviceToHost, G.stream));
  HIPCHK(hipStreamSynchronize(G.stream));
  if (G.rank != root) memcpy(buf, G.h_bytes, (size_t)nbytes);
  G.collectives++;
  return 0;
}

int odr_comm_barrier(odr_ctx *c) {
  double one = 1.0;
  return odr_allreduce_scalars(c, &one, 1, 0);
}

}  // extern "C"

// stage_block (odrift.hip): the float32 arrays of one reader level, staged side by side in device memory on `root`, to every
// rank -- ONE broadcast per level, in place, on the upload stream, on the bulk communicator
int odr_i_comm_rank() { return G.on ? G.rank : 0; }
bool odr_i_comm_on() { return G.on; }
int odr_i_comm_bcast_floats(float *dev, size_t count, int root, hipStream_t st) {
  NEED_COMM();
  REQUIRE(root >= 0 && root < G.nranks, "bad root %d", root);
  NCCLCHK(R.Broadcast(dev, dev, count, ncclFloat, root, G.bulk, st));
  G.bulk_collectives++;
  return 0;
}
