// mansy_comm_* / mansy_allreduce_* : thin wrappers over RCCL communicators ("plus mansy_allreduce_* thin wrappers over RCCL
// communicators"; round 5).  The data-parallel hot path has exactly three collectives -- the flat-gradient average (VP 36.8 MB; PPO 1.7 /
// 1.05 MB), the SyncBN statistics (2 x d_model doubles) and the return normaliser's 3 doubles -- and these are their C-ABI forms: explicit
// communicator, explicit stream, device pointers, int status.  RCCL is bound at RUN time (dlopen of librccl.so: torch's copy when the process
// already holds it, ROCm's otherwise), so libmansy_hip.so has no link-time dependency on it and loads on a box without RCCL; the first
// mansy_comm_* call there returns MANSY_EHIP with the loader's message.  A communicator context starts with the same `kind` tag as the
// peer-memory context of csrc/xgmi.hip, so the engine entry points that take a `sync` context (mansy_ppo_minibatch_step,
// mansy_identifier_train_step) accept either.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "mansy_kernels.h"
#include "../../include/mansy_hip.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  char why[256] = "";
};

// bound once per process, read-only afterwards (a function table, not a mode)
Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { x.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (x.handle) break; }
    if (!x.handle) { snprintf(x.why, sizeof(x.why), "librccl.so not loadable: %s", dlerror()); return x; }
    auto sym = [&](const char* s) { return dlsym(x.handle, s); };
    x.GetUniqueId = (decltype(x.GetUniqueId))sym("ncclGetUniqueId");
    x.CommInitRank = (decltype(x.CommInitRank))sym("ncclCommInitRank");
    x.CommDestroy = (decltype(x.CommDestroy))sym("ncclCommDestroy");
    x.AllReduce = (decltype(x.AllReduce))sym("ncclAllReduce");
    x.AllGather = (decltype(x.AllGather))sym("ncclAllGather");
    x.GetErrorString = (decltype(x.GetErrorString))sym("ncclGetErrorString");
    x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce && x.AllGather && x.GetErrorString;
    if (!x.ok) snprintf(x.why, sizeof(x.why), "librccl.so lacks an expected symbol");
    return x;
  }();
  return r;
}

#define RCCL_CHECK(expr, what)                                                                        \
  do {                                                                                                \
    const ncclResult_t r_ = (expr);                                                                   \
    if (r_ != ncclSuccess) { mansy_set_error("%s: RCCL error %d (%s)", what, (int)r_, rccl().GetErrorString(r_)); return MANSY_EHIP; } \
  } while (0)

}  // namespace

struct MansyCommCtx { int kind; int world, rank; ncclComm_t comm; };      // kind == MANSY_SYNC_RCCL

extern "C" {

int mansy_comm_unique_id(mansy_comm_id* out) {
  MANSY_REQUIRE(out, "comm_unique_id: null");
  static_assert(sizeof(mansy_comm_id) == sizeof(ncclUniqueId), "mansy_comm_id must hold an ncclUniqueId");
  Rccl& R = rccl();
  if (!R.ok) { mansy_set_error("comm_unique_id: %s", R.why); return MANSY_EHIP; }
  ncclUniqueId id;
  RCCL_CHECK(R.GetUniqueId(&id), "comm_unique_id");
  memcpy(out->bytes, &id, sizeof(id));
  return MANSY_OK;
}

// Every rank calls this with rank 0's id (the host moves the 128 bytes: torch.distributed broadcast_object_list, MPI, a file ...), on its own
// current device.  Collective: returns once all `world` ranks have joined.
int mansy_comm_create(const mansy_comm_id* id, int world, int rank, void** comm_out) {
  MANSY_REQUIRE(id && comm_out && world >= 1 && rank >= 0 && rank < world, "comm_create: bad arguments");
  Rccl& R = rccl();
  if (!R.ok) { mansy_set_error("comm_create: %s", R.why); return MANSY_EHIP; }
  ncclUniqueId nid;
  memcpy(&nid, id->bytes, sizeof(nid));
  MansyCommCtx* c = new MansyCommCtx{MANSY_SYNC_RCCL, world, rank, nullptr};
  const ncclResult_t r = R.CommInitRank(&c->comm, world, nid, rank);
  if (r != ncclSuccess) { mansy_set_error("comm_create: ncclCommInitRank -> %d (%s)", (int)r, R.GetErrorString(r)); delete c; return MANSY_EHIP; }
  *comm_out = c;
  return MANSY_OK;
}

int mansy_comm_destroy(void* comm) {
  MansyCommCtx* c = (MansyCommCtx*)comm;
  if (!c) return MANSY_OK;
  MANSY_REQUIRE(c->kind == MANSY_SYNC_RCCL, "comm_destroy: not a communicator context");
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  delete c;
  return MANSY_OK;
}

// in place: buf <- mean over the ranks (ncclAvg: the division happens inside the collective, one launch)
int mansy_allreduce_avg_f32(void* comm, float* buf, long long n, void* stream) {
  MansyCommCtx* c = (MansyCommCtx*)comm;
  MANSY_REQUIRE(c && c->kind == MANSY_SYNC_RCCL && buf && n >= 1, "allreduce_avg_f32: bad arguments");
  RCCL_CHECK(rccl().AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclAvg, c->comm, (hipStream_t)stream), "allreduce_avg_f32");
  return MANSY_OK;
}
// in place: buf <- sum over the ranks (the SyncBN hook's 2 x d_model doubles)
int mansy_allreduce_sum_f64(void* comm, double* buf, long long n, void* stream) {
  MansyCommCtx* c = (MansyCommCtx*)comm;
  MANSY_REQUIRE(c && c->kind == MANSY_SYNC_RCCL && buf && n >= 1, "allreduce_sum_f64: bad arguments");
  RCCL_CHECK(rccl().AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclSum, c->comm, (hipStream_t)stream), "allreduce_sum_f64");
  return MANSY_OK;
}
// recv [world][n_per_rank] <- every rank's send [n_per_rank] (the return normaliser's (mean, var, count))
int mansy_allgather_f64(void* comm, const double* send, double* recv, long long n_per_rank, void* stream) {
  MansyCommCtx* c = (MansyCommCtx*)comm;
  MANSY_REQUIRE(c && c->kind == MANSY_SYNC_RCCL && send && recv && n_per_rank >= 1, "allgather_f64: bad arguments");
  RCCL_CHECK(rccl().AllGather(send, recv, (size_t)n_per_rank, ncclFloat64, c->comm, (hipStream_t)stream), "allgather_f64");
  return MANSY_OK;
}

}  // extern "C"
