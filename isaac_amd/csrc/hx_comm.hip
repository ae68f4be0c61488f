// hx_comm.hip -- the data-parallel exchange of the hector hot path inside libhx.so: RCCL over xGMI, called from the
// learner's own HIP stream (include/hx_ppo.h "data parallelism").  The reference has no counterpart (single process;
// SURVEY.md 2.1, 8e): environments never interact, so each rank owns its shard of robots and the only exchanges are
//   * one all-reduce(sum) of the flat [gradient | kl_sum | value_loss_sum | surrogate_sum | rows] buffer per optimiser
//     step, enqueued on the learner's stream between the backward kernels and the Adam kernel -- no host synchronisation;
//   * one all-reduce(sum) of three doubles per iteration (advantage moments), same stream;
//   * one broadcast of the parameters at start-up.
// RCCL is bound at run time (dlopen of librccl.so and its nccl* entry points): a single-GPU run never loads it, and a
// process that already carries a copy of the library (PyTorch ships one) shares that copy instead of a second one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstring>
#include <string>
#include "../../include/hx_ppo.h"
#include "hx_common.h"

namespace {
// the slice of rccl.h this file needs (ABI of RCCL 2.x: ncclUniqueId is 128 opaque bytes, enums as below)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclSum = 0, ncclMax = 2 };
enum { ncclFloat32 = 7, ncclFloat64 = 8 };
struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.handle) return 0;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* nm : names) {
    g_rccl.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) { hx_set_error(std::string("hx_comm: cannot load librccl.so: ") + dlerror()); return -10; }
  auto sym = [&](const char* n) { return dlsym(g_rccl.handle, n); };
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
  g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.Broadcast) {
    hx_set_error("hx_comm: librccl.so lacks an nccl* entry point"); g_rccl.handle = nullptr; return -10;
  }
  return 0;
}
int check_nccl(int rc, const char* what) {
  if (rc == ncclSuccess) return 0;
  hx_set_error(std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
  return -200 - rc;
}
}  // namespace

struct hx_comm { ncclComm_t comm; int rank, world; };

static_assert(HX_COMM_ID_BYTES == sizeof(ncclUniqueId), "HX_COMM_ID_BYTES must be the size of ncclUniqueId");

extern "C" int hx_comm_get_unique_id(uint8_t* id_h) {
  if (!id_h) { hx_set_error("hx_comm_get_unique_id: null buffer"); return -2; }
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  if (int rc = check_nccl(g_rccl.GetUniqueId(&id), "ncclGetUniqueId")) return rc;
  memcpy(id_h, &id, sizeof(id));
  return 0;
}

extern "C" int hx_comm_init(const uint8_t* id_h, int rank, int world, hx_comm** out) {
  if (!id_h || !out || world < 1 || rank < 0 || rank >= world) { hx_set_error("hx_comm_init: bad arguments"); return -2; }
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  memcpy(&id, id_h, sizeof(id));
  hx_comm* c = new hx_comm{nullptr, rank, world};
  const int rc = check_nccl(g_rccl.CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
  if (rc) { delete c; return rc; }
  *out = c;
  return 0;
}

extern "C" void hx_comm_destroy(hx_comm* c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
}
extern "C" int hx_comm_rank(hx_comm* c) { return c ? c->rank : 0; }
extern "C" int hx_comm_world(hx_comm* c) { return c ? c->world : 1; }

extern "C" int hx_comm_all_reduce(hx_comm* c, void* buf, size_t count, int dtype, int op, void* stream) {
  if (!c || !buf) { hx_set_error("hx_comm_all_reduce: null argument"); return -2; }
  if (dtype != HX_COMM_F32 && dtype != HX_COMM_F64) { hx_set_error("hx_comm_all_reduce: dtype must be HX_COMM_F32 or HX_COMM_F64"); return -2; }
  if (op != HX_COMM_SUM && op != HX_COMM_MAX) { hx_set_error("hx_comm_all_reduce: op must be HX_COMM_SUM or HX_COMM_MAX"); return -2; }
  return check_nccl(g_rccl.AllReduce(buf, buf, count, dtype == HX_COMM_F32 ? ncclFloat32 : ncclFloat64, op == HX_COMM_SUM ? ncclSum : ncclMax,
                                     c->comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int hx_comm_broadcast(hx_comm* c, void* buf, size_t count_f32, int root, void* stream) {
  if (!c || !buf) { hx_set_error("hx_comm_broadcast: null argument"); return -2; }
  return check_nccl(g_rccl.Broadcast(buf, buf, count_f32, ncclFloat32, root, c->comm, (hipStream_t)stream), "ncclBroadcast");
}
