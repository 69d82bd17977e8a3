// RCCL behind the C ABI: the gradient all-reduce of data-parallel training
// (replaces the implicit gradient sum over the in-graph towers of the reference,
// flypylib/multi_gpu.py:20-61 via flypylib/fplnetwork.py:124-128).
//
// One communicator per context (= per GPU).  librccl is opened with dlopen at the
// first fpl_comm_* call, so libfplhip.so itself has no link-time dependency on it
// and a process that already carries an RCCL (PyTorch's) shares that copy:
//   FPL_RCCL_LIB (explicit path)  ->  an already loaded librccl.so / librccl.so.1
//   ->  librccl.so.1 / librccl.so on the loader path  ->  /opt/rocm/lib/librccl.so.1
// The host passes the 128-byte unique id between ranks by whatever it has
// (torch.distributed object broadcast, MPI, a file); ranks of one process (one host
// thread per GPU) call fpl_comm_init concurrently, as ncclCommInitRank requires.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include <unistd.h>

#include "common.h"

namespace {

struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;     // optional symbol
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t,
                            ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t,
                            hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  char path[256] = {0};
};

RcclApi g_rccl;
std::mutex g_rccl_mutex;

int load_rccl(fpl_ctx *ctx) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle) return 0;
  const char *env = getenv("FPL_RCCL_LIB");
  struct Try { const char *name; int flags; };
  const Try tries[] = {
      {env, RTLD_NOW | RTLD_LOCAL},
      {"librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD},
      {"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD},
      {"librccl.so.1", RTLD_NOW | RTLD_LOCAL},
      {"librccl.so", RTLD_NOW | RTLD_LOCAL},
      {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL},
  };
  void *h = nullptr;
  const char *used = nullptr;
  // the FIRST failure that is not a mere "not loaded yet" is the informative one (the
  // FPL_RCCL_LIB path, or the plain soname); dlerror() clears itself when read
  char first_err[256] = {0};
  for (const Try &t : tries) {
    if (!t.name || !t.name[0]) continue;
    h = dlopen(t.name, t.flags);
    if (h) { used = t.name; break; }
    const char *e = dlerror();
    if (e && !first_err[0] && !(t.flags & RTLD_NOLOAD))
      snprintf(first_err, sizeof(first_err), "%s", e);
  }
  if (!h)
    return fpl_fail(ctx, "fpl_comm: cannot open librccl (%s); set FPL_RCCL_LIB",
                    first_err[0] ? first_err : "not found");
  RcclApi api;
  api.handle = h;
  snprintf(api.path, sizeof(api.path), "%s", used);
#define FPL_SYM(field, sym)                                                    \
  do {                                                                         \
    api.field = (decltype(api.field))dlsym(h, sym);                            \
    if (!api.field) {                                                          \
      dlclose(h);                                                              \
      return fpl_fail(ctx, "fpl_comm: %s has no symbol %s", used, sym);        \
    }                                                                          \
  } while (0)
  FPL_SYM(GetUniqueId, "ncclGetUniqueId");
  FPL_SYM(CommInitRank, "ncclCommInitRank");
  FPL_SYM(CommDestroy, "ncclCommDestroy");
  FPL_SYM(AllReduce, "ncclAllReduce");
  FPL_SYM(Broadcast, "ncclBroadcast");
  FPL_SYM(GetErrorString, "ncclGetErrorString");
#undef FPL_SYM
  api.CommAbort = (decltype(api.CommAbort))dlsym(h, "ncclCommAbort");
  g_rccl = api;
  return 0;
}

#define FPL_NCCL(ctx, expr)                                                    \
  do {                                                                         \
    ncclResult_t r__ = (expr);                                                 \
    if (r__ != ncclSuccess)                                                    \
      return fpl_fail((ctx), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,     \
                      g_rccl.GetErrorString(r__));                             \
  } while (0)

}  // namespace

// take the communicator out of the context (nobody else will see it afterwards)
static void *comm_take(fpl_ctx *ctx) {
  std::lock_guard<std::mutex> lk(ctx->comm_mu);
  void *c = ctx->comm;
  ctx->comm = nullptr;
  ctx->comm_rank = 0;
  ctx->comm_nranks = 1;
  return c;
}

void fpl_comm_release(fpl_ctx *ctx) {
  if (!ctx) return;
  void *c = comm_take(ctx);
  if (c && g_rccl.handle) g_rccl.CommDestroy((ncclComm_t)c);
}

// a collective's hold on the communicator: copied under the lock together with the count that
// fpl_comm_abort waits on, so the communicator cannot be freed between the copy and the call
struct CommUse {
  fpl_ctx *ctx;
  void *comm = nullptr;
  explicit CommUse(fpl_ctx *c) : ctx(c) {
    std::lock_guard<std::mutex> lk(ctx->comm_mu);
    if (!ctx->comm_aborting.load() && ctx->comm) {
      comm = ctx->comm;
      ++ctx->comm_inflight;
    }
  }
  ~CommUse() {
    if (comm) {
      std::lock_guard<std::mutex> lk(ctx->comm_mu);
      --ctx->comm_inflight;
    }
  }
};

extern "C" {

int fpl_comm_unique_id(uint8_t out[FPL_COMM_ID_BYTES]) {
  if (!out) return fpl_fail(nullptr, "fpl_comm_unique_id: out is NULL");
  static_assert(sizeof(ncclUniqueId) == FPL_COMM_ID_BYTES, "unique id size");
  FPL_TRY(load_rccl(nullptr));
  ncclUniqueId id;
  FPL_NCCL(nullptr, g_rccl.GetUniqueId(&id));
  memcpy(out, &id, sizeof(id));
  return 0;
}

int fpl_comm_init(fpl_ctx *ctx, int32_t rank, int32_t nranks,
                  const uint8_t unique_id[FPL_COMM_ID_BYTES]) {
  if (!ctx || !unique_id) return fpl_fail(ctx, "fpl_comm_init: NULL argument");
  FPL_REQUIRE(ctx, nranks >= 1 && rank >= 0 && rank < nranks,
              "fpl_comm_init: rank %d of %d", rank, nranks);
  FPL_REQUIRE(ctx, ctx->comm == nullptr,
              "fpl_comm_init: this context already has a communicator "
              "(fpl_comm_destroy it first)");
  FPL_TRY(load_rccl(ctx));
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  FPL_NCCL(ctx, g_rccl.CommInitRank(&comm, nranks, id, rank));
  std::lock_guard<std::mutex> lk(ctx->comm_mu);
  ctx->comm = comm;
  ctx->comm_rank = rank;
  ctx->comm_nranks = nranks;
  return 0;
}

int fpl_comm_destroy(fpl_ctx *ctx) {
  if (!ctx) return fpl_fail(nullptr, "fpl_comm_destroy: ctx is NULL");
  if (ctx->comm) {
    FPL_HIP(ctx, hipSetDevice(ctx->device));
    FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  fpl_comm_release(ctx);
  return 0;
}

int fpl_comm_abort(fpl_ctx *ctx) {
  if (!ctx) return fpl_fail(nullptr, "fpl_comm_abort: ctx is NULL");
  // no stream synchronisation: the point is to get out of a collective that will never
  // complete because a peer is gone.  May be called from another host thread: new collectives
  // are refused from here on, calls that already hold the communicator get a moment to return
  // (an enqueue takes microseconds), and one that is still inside - blocked in the collective
  // this abort is meant to break - is what ncclCommAbort exists for.
  ctx->comm_aborting.store(true);
  for (int i = 0; i < 200; ++i) {
    {
      std::lock_guard<std::mutex> lk(ctx->comm_mu);
      if (ctx->comm_inflight == 0) break;
    }
    usleep(10000);
  }
  void *c = comm_take(ctx);
  if (c && g_rccl.handle) {
    if (g_rccl.CommAbort) g_rccl.CommAbort((ncclComm_t)c);
    // else: no abort entry point in this librccl - the communicator is LEAKED; ncclCommDestroy
    // would wait for the stuck collective
  }
  ctx->comm_aborting.store(false);
  return 0;
}

int fpl_comm_info(fpl_ctx *ctx, int32_t *rank, int32_t *nranks, char *lib_path,
                  size_t lib_path_cap) {
  if (!ctx) return fpl_fail(nullptr, "fpl_comm_info: ctx is NULL");
  if (rank) *rank = ctx->comm ? ctx->comm_rank : 0;
  if (nranks) *nranks = ctx->comm ? ctx->comm_nranks : 0;
  if (lib_path && lib_path_cap) snprintf(lib_path, lib_path_cap, "%s", g_rccl.path);
  return 0;
}

int fpl_comm_allreduce_sum_f32(fpl_ctx *ctx, float *dev_ptr, int64_t n) {
  if (!ctx || !dev_ptr) return fpl_fail(ctx, "fpl_comm_allreduce_sum_f32: NULL argument");
  CommUse use(ctx);
  FPL_REQUIRE(ctx, use.comm != nullptr,
              "fpl_comm_allreduce_sum_f32: no communicator (call fpl_comm_init; or it was aborted)");
  FPL_REQUIRE(ctx, n >= 0, "fpl_comm_allreduce_sum_f32: n %lld", (long long)n);
  if (n == 0) return 0;
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  {
    TimedLaunch tl(ctx, "rccl_allreduce_f32");
    FPL_NCCL(ctx, g_rccl.AllReduce(dev_ptr, dev_ptr, (size_t)n, ncclFloat, ncclSum,
                                   (ncclComm_t)use.comm, ctx->stream));
  }
  return 0;
}

int fpl_comm_broadcast_f32(fpl_ctx *ctx, float *dev_ptr, int64_t n, int32_t root) {
  if (!ctx || !dev_ptr) return fpl_fail(ctx, "fpl_comm_broadcast_f32: NULL argument");
  CommUse use(ctx);
  FPL_REQUIRE(ctx, use.comm != nullptr,
              "fpl_comm_broadcast_f32: no communicator (call fpl_comm_init; or it was aborted)");
  FPL_REQUIRE(ctx, root >= 0 && root < ctx->comm_nranks, "fpl_comm_broadcast_f32: root %d", root);
  if (n <= 0) return 0;
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  FPL_NCCL(ctx, g_rccl.Broadcast(dev_ptr, dev_ptr, (size_t)n, ncclFloat, root,
                                 (ncclComm_t)use.comm, ctx->stream));
  return 0;
}

}  // extern "C"
