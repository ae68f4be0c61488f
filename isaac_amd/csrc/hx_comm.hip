// hx_comm.hip -- the data-parallel exchange of the hector hot path inside libhx.so: RCCL over xGMI, called from the
// learner's own HIP stream (include/hx_ppo.h "data parallelism").  The reference has no counterpart (single process;
// SURVEY.md 2.1, 8e): environments never interact, so each rank owns its shard of robots and the only exchanges are
//   * one all-reduce(sum) of the flat [gradient | kl_sum | value_loss_sum | surrogate_sum | rows] buffer per optimiser
//     step, enqueued on the learner's stream between the backward kernels and the Adam kernel -- no host synchronisation;
//   * one all-reduce(sum) of three doubles per iteration (advantage moments), same stream;
//   * one broadcast of the parameters at start-up.
// RCCL is bound at run time: dlopen("librccl.so.1", RTLD_NOLOAD) first, so that a process which already carries the library
// (PyTorch maps it under that SONAME) shares that copy; a single-GPU run never loads it.  Types and enum values come from
// <rccl/rccl.h>; ncclGetVersion of the loaded copy must report the same major version.  Nothing here waits without a deadline:
// hx_comm_init gives up after HX_COMM_INIT_TIMEOUT_S, hx_comm_wait (used wherever the learner synchronises with a stream that
// carries collectives) after HX_COMM_TIMEOUT_S, both with an error string and a non-zero return.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>      // types, enums and NCCL_VERSION_CODE of the RCCL this library was compiled against (no link dependency)
#else
// A ROCm install without the RCCL development headers still builds the (default, single-GPU) library: the slice of the NCCL 2.x
// API used below, as published in rccl.h (stable within the major version, which load_rccl() checks at run time).
#define NCCL_MAJOR 2
#define NCCL_VERSION_CODE 20000
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5,
               ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
extern "C" {
ncclResult_t ncclGetVersion(int*);
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclCommAbort(ncclComm_t);
ncclResult_t ncclCommGetAsyncError(ncclComm_t, ncclResult_t*);
ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
ncclResult_t ncclBroadcast(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
const char* ncclGetErrorString(ncclResult_t);
}
#endif
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include "../../include/hx_ppo.h"
#include "hx_common.h"

namespace {
struct Rccl {
  void* handle = nullptr;
  int version = 0;
  bool shared_copy = false;      // true: the process already had librccl.so.1 mapped (PyTorch's) and this is that copy
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.handle) return 0;
  // 1. the copy this process already carries (torch maps librccl.so.1): RTLD_NOLOAD finds it by SONAME and never maps a second
  //    one; 2. the versioned SONAME; 3. the development names.
  g_rccl.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
  g_rccl.shared_copy = g_rccl.handle != nullptr;
  if (!g_rccl.handle) {
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) {
      g_rccl.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (g_rccl.handle) break;
    }
  }
  if (!g_rccl.handle) { hx_set_error(std::string("hx_comm: cannot load librccl.so.1: ") + dlerror()); return -10; }
  auto sym = [&](const char* n) { return dlsym(g_rccl.handle, n); };
#define HX_SYM(field, name) g_rccl.field = (decltype(g_rccl.field))sym(name)
  HX_SYM(GetVersion, "ncclGetVersion"); HX_SYM(GetUniqueId, "ncclGetUniqueId"); HX_SYM(CommInitRank, "ncclCommInitRank");
  HX_SYM(CommDestroy, "ncclCommDestroy"); HX_SYM(CommAbort, "ncclCommAbort"); HX_SYM(CommGetAsyncError, "ncclCommGetAsyncError");
  HX_SYM(AllReduce, "ncclAllReduce"); HX_SYM(Broadcast, "ncclBroadcast"); HX_SYM(GetErrorString, "ncclGetErrorString");
#undef HX_SYM
  if (!g_rccl.GetVersion || !g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.Broadcast) {
    hx_set_error("hx_comm: librccl.so lacks an nccl* entry point"); g_rccl.handle = nullptr; return -10;
  }
  // ABI check: the structs and enum values used below come from the rccl.h of this build (NCCL_VERSION_CODE); they are stable
  // within a major version of the NCCL API, so the library found at run time must report the same major version.
  int v = 0;
  if (g_rccl.GetVersion(&v) != ncclSuccess || v / 10000 != NCCL_MAJOR) {
    hx_set_error("hx_comm: librccl.so reports NCCL API version " + std::to_string(v) + ", this library was built against " +
                 std::to_string(NCCL_VERSION_CODE) + " (major versions must agree)");
    g_rccl.handle = nullptr; return -10;
  }
  g_rccl.version = v;
  return 0;
}
int check_nccl(ncclResult_t rc, const char* what) {
  if (rc == ncclSuccess) return 0;
  hx_set_error(std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
  return -200 - (int)rc;
}
double env_seconds(const char* name, double dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  char* end = nullptr;
  const double v = strtod(e, &end);
  return (end != e && v > 0.0) ? v : dflt;
}
}  // namespace

struct hx_comm { ncclComm_t comm; int rank, world; double timeout_s; bool dead; };

static_assert(HX_COMM_ID_BYTES == sizeof(ncclUniqueId), "HX_COMM_ID_BYTES must be the size of ncclUniqueId");
static_assert(NCCL_MAJOR == 2, "hx_comm.hip is written against the NCCL 2.x API");

extern "C" int hx_comm_get_unique_id(uint8_t* id_h) {
  if (!id_h) { hx_set_error("hx_comm_get_unique_id: null buffer"); return -2; }
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  if (int rc = check_nccl(g_rccl.GetUniqueId(&id), "ncclGetUniqueId")) return rc;
  memcpy(id_h, &id, sizeof(id));
  return 0;
}

// {NCCL API version of the loaded library, 1 if it is the copy the process had mapped already (RTLD_NOLOAD hit)}
extern "C" int hx_comm_library_info(int* version, int* shared_copy) {
  if (int rc = load_rccl()) return rc;
  if (version) *version = g_rccl.version;
  if (shared_copy) *shared_copy = g_rccl.shared_copy ? 1 : 0;
  return 0;
}

// ncclCommInitRank blocks until every rank of the job has arrived.  A rank that never comes (crashed child, wrong WORLD_SIZE)
// would leave the others inside it forever, so it runs on a helper thread and the caller gives up after HX_COMM_INIT_TIMEOUT_S
// (default 300 s): error string + non-zero return, the process is expected to exit (the helper thread is abandoned).
extern "C" int hx_comm_init(const uint8_t* id_h, int rank, int world, hx_comm** out) {
  if (!id_h || !out || world < 1 || rank < 0 || rank >= world) { hx_set_error("hx_comm_init: bad arguments"); return -2; }
  if (int rc = load_rccl()) return rc;
  int dev = 0;
  HX_CHECK(hipGetDevice(&dev));
  struct Job { ncclUniqueId id; int rank, world, dev; ncclComm_t comm = nullptr; ncclResult_t rc = ncclSuccess; std::atomic<int> done{0}; };
  auto job = std::make_shared<Job>();
  memcpy(&job->id, id_h, sizeof(job->id));
  job->rank = rank; job->world = world; job->dev = dev;
  std::thread([job]() {
    (void)hipSetDevice(job->dev);
    job->rc = g_rccl.CommInitRank(&job->comm, job->world, job->id, job->rank);
    job->done.store(1, std::memory_order_release);
  }).detach();
  const double limit = env_seconds("HX_COMM_INIT_TIMEOUT_S", 300.0);
  const auto t0 = std::chrono::steady_clock::now();
  while (!job->done.load(std::memory_order_acquire)) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
      hx_set_error("hx_comm_init: ncclCommInitRank did not return within " + std::to_string((int)limit) + " s (rank " + std::to_string(rank) +
                   " of " + std::to_string(world) + "): a rank of the job is missing; this process should exit");
      return -11;
    }
    usleep(2000);
  }
  if (int rc = check_nccl(job->rc, "ncclCommInitRank")) return rc;
  *out = new hx_comm{job->comm, rank, world, env_seconds("HX_COMM_TIMEOUT_S", 120.0), false};
  return 0;
}

extern "C" void hx_comm_destroy(hx_comm* c) {
  if (!c) return;
  if (c->comm) {
    if (c->dead && g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);          // a communicator with a stuck collective cannot be destroyed
    else if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  }
  delete c;
}
extern "C" int hx_comm_rank(hx_comm* c) { return c ? c->rank : 0; }
extern "C" int hx_comm_world(hx_comm* c) { return c ? c->world : 1; }

extern "C" int hx_comm_all_reduce(hx_comm* c, void* buf, size_t count, int dtype, int op, void* stream) {
  if (!c || !buf) { hx_set_error("hx_comm_all_reduce: null argument"); return -2; }
  if (c->dead) { hx_set_error("hx_comm_all_reduce: communicator was aborted after a time-out"); return -12; }
  if (dtype != HX_COMM_F32 && dtype != HX_COMM_F64) { hx_set_error("hx_comm_all_reduce: dtype must be HX_COMM_F32 or HX_COMM_F64"); return -2; }
  if (op != HX_COMM_SUM && op != HX_COMM_MAX) { hx_set_error("hx_comm_all_reduce: op must be HX_COMM_SUM or HX_COMM_MAX"); return -2; }
  return check_nccl(g_rccl.AllReduce(buf, buf, count, dtype == HX_COMM_F32 ? ncclFloat32 : ncclFloat64, op == HX_COMM_SUM ? ncclSum : ncclMax,
                                     c->comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int hx_comm_broadcast(hx_comm* c, void* buf, size_t count_f32, int root, void* stream) {
  if (!c || !buf) { hx_set_error("hx_comm_broadcast: null argument"); return -2; }
  if (c->dead) { hx_set_error("hx_comm_broadcast: communicator was aborted after a time-out"); return -12; }
  return check_nccl(g_rccl.Broadcast(buf, buf, count_f32, ncclFloat32, root, c->comm, (hipStream_t)stream), "ncclBroadcast");
}

// Watchdog: wait for everything enqueued on `stream` (collectives included) WITHOUT an unbounded hipStreamSynchronize.
// A collective whose peer died never completes; after HX_COMM_TIMEOUT_S (default 120 s; `timeout_s` > 0 overrides) or on an
// asynchronous RCCL error the communicator is aborted (ncclCommAbort releases the stuck kernel), the error string says which
// rank gave up, and the call returns non-zero -- the process is expected to exit; a launcher may start a fresh one.
extern "C" int hx_comm_wait(hx_comm* c, void* stream, double timeout_s) {
  hipStream_t st = (hipStream_t)stream;
  if (!c) { HX_CHECK(hipStreamSynchronize(st)); return 0; }
  // an aborted communicator has no handle left to query or to abort again: every later wait on it fails at once
  if (c->dead || !c->comm) { hx_set_error("hx_comm_wait: communicator was aborted after a time-out or an RCCL error (rank " + std::to_string(c->rank) + ")"); return -12; }
  const double limit = timeout_s > 0.0 ? timeout_s : c->timeout_s;
  const auto t0 = std::chrono::steady_clock::now();
  long spins = 0;
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) { HX_CHECK(q); }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ncclResult_t async = ncclSuccess;
    if ((++spins & 63) == 0 && g_rccl.CommGetAsyncError && g_rccl.CommGetAsyncError(c->comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
      c->dead = true;
      if (g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
      c->comm = nullptr;
      hx_set_error(std::string("hx_comm_wait: asynchronous RCCL error on rank ") + std::to_string(c->rank) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(async) : "?"));
      return -13;
    }
    if (el > limit) {
      c->dead = true;
      if (g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
      c->comm = nullptr;
      hx_set_error("hx_comm_wait: work on the stream (an RCCL collective) did not complete within " + std::to_string((int)limit) + " s on rank " +
                   std::to_string(c->rank) + " of " + std::to_string(c->world) + ": a peer rank is gone; communicator aborted, this process should exit");
      return -12;
    }
    if (el > 0.002) usleep(el > 0.5 ? 1000 : 50);      // spin for the first 2 ms (the common case: the stream is nearly drained)
  }
}
