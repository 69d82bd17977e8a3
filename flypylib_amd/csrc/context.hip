// Context, error strings, device block cache, HIP-event kernel timing.
#include "common.h"

thread_local char g_fpl_err[FPL_MAX_ERR] = {0};

int fpl_fail(fpl_ctx *ctx, const char *fmt, ...) {
  char buf[FPL_MAX_ERR];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) memcpy(ctx->err, buf, sizeof(buf));
  memcpy(g_fpl_err, buf, sizeof(buf));
  return 1;
}

int fpl_fail_range(fpl_ctx *ctx, const char *fmt, ...) {
  char buf[FPL_MAX_ERR];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) memcpy(ctx->err, buf, sizeof(buf));
  memcpy(g_fpl_err, buf, sizeof(buf));
  return FPL_RC_RANGE;
}

int fpl_fail_range_call(fpl_ctx *ctx, const char *fmt, ...) {
  char buf[FPL_MAX_ERR];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) memcpy(ctx->err, buf, sizeof(buf));
  memcpy(g_fpl_err, buf, sizeof(buf));
  return FPL_RC_RANGE_CALL;
}

int fpl_range_flag(fpl_ctx *ctx, unsigned **dev) {
  if (!ctx->range_flag_dev) {
    FPL_HIP(ctx, hipMalloc((void **)&ctx->range_flag_dev, sizeof(unsigned)));
    FPL_HIP(ctx, hipHostMalloc((void **)&ctx->range_flag_host, sizeof(unsigned), hipHostMallocDefault));
    *ctx->range_flag_host = 0u;
    FPL_HIP(ctx, hipMemsetAsync(ctx->range_flag_dev, 0, sizeof(unsigned), ctx->stream));
  }
  *dev = ctx->range_flag_dev;
  return 0;
}

extern "C" {

int fpl_abi_version(void) { return FPL_ABI_VERSION; }

const char *fpl_last_error(fpl_ctx *ctx) { return ctx ? ctx->err : g_fpl_err; }

int fpl_ctx_create(int device_id, fpl_ctx **out) {
  if (!out) return fpl_fail(nullptr, "fpl_ctx_create: out is NULL");
  *out = nullptr;
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev == 0)
    return fpl_fail(nullptr, "fpl_ctx_create: no HIP device (%s)",
                    hipGetErrorString(e));
  if (device_id < 0 || device_id >= n_dev)
    return fpl_fail(nullptr, "fpl_ctx_create: device %d out of range [0,%d)",
                    device_id, n_dev);
  FPL_HIP(nullptr, hipSetDevice(device_id));
  hipDeviceProp_t prop;
  FPL_HIP(nullptr, hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fpl_fail(nullptr,
                    "fpl_ctx_create: device %d is %s; this library is built "
                    "for gfx950 (MI355X) only", device_id, prop.gcnArchName);
  // The HIP runtime in the process may not be the one this library was built against (the
  // Python binding loads PyTorch's bundled libamdhip64 first so that both share one runtime):
  // say so once when even the major versions differ; FPL_HIP_VERSION_CHECK=1 reports any
  // difference.
  {
    static bool said = false;
    int rt = 0;
    if (!said && hipRuntimeGetVersion(&rt) == hipSuccess) {
      const int built_major = HIP_VERSION_MAJOR, built_minor = HIP_VERSION_MINOR;
      const int rt_major = rt / 10000000, rt_minor = (rt / 100000) % 100;
      const bool strict = getenv("FPL_HIP_VERSION_CHECK") != nullptr;
      if (rt_major != built_major || (strict && rt_minor != built_minor))
        fprintf(stderr, "libfplhip: built against HIP %d.%d, running on the HIP %d.%d runtime "
                        "already in this process\n", built_major, built_minor, rt_major, rt_minor);
      said = true;
    }
  }
  fpl_ctx *ctx = new fpl_ctx();
  ctx->device = device_id;
  ctx->n_cu = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) !=
      hipSuccess) {
    delete ctx;
    return fpl_fail(nullptr, "fpl_ctx_create: hipStreamCreate failed");
  }
  ctx->stream = ctx->own_stream;
  *out = ctx;
  return 0;
}

int fpl_ctx_destroy(fpl_ctx *ctx) {
  if (!ctx) return 0;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  fpl_comm_release(ctx);
  for (auto &p : ctx->pending) {
    ctx->event_pool.push_back(p.start);
    ctx->event_pool.push_back(p.stop);
  }
  for (hipEvent_t ev : ctx->event_pool) hipEventDestroy(ev);
  if (ctx->v2o.smoothed) fpl_dev_release(ctx, ctx->v2o.smoothed);
  if (ctx->v2o.seg) fpl_dev_release(ctx, ctx->v2o.seg);
  if (ctx->v2o.cellmax) fpl_dev_release(ctx, ctx->v2o.cellmax);
  if (ctx->v2o.smoothed64) fpl_dev_release(ctx, ctx->v2o.smoothed64);
  if (ctx->v2o.sort_keys) fpl_dev_release(ctx, ctx->v2o.sort_keys);
  if (ctx->v2o.sort_idx) fpl_dev_release(ctx, ctx->v2o.sort_idx);
  fpl_dev_trim(ctx);
  for (auto &kv : ctx->live_blocks) hipFree(kv.first);
  if (ctx->zero_pool) hipFree(ctx->zero_pool);
  if (ctx->range_flag_dev) hipFree(ctx->range_flag_dev);
  if (ctx->range_flag_host) hipHostFree(ctx->range_flag_host);
  hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return 0;
}

int fpl_ctx_set_stream(fpl_ctx *ctx, void *hip_stream) {
  if (!ctx) return fpl_fail(nullptr, "fpl_ctx_set_stream: ctx is NULL");
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return 0;
}

int fpl_ctx_synchronize(fpl_ctx *ctx) {
  if (!ctx) return fpl_fail(nullptr, "fpl_ctx_synchronize: ctx is NULL");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int fpl_device_info(fpl_ctx *ctx, int32_t *n_cu, int64_t *hbm_bytes, char *name,
                    size_t name_cap) {
  if (!ctx) return fpl_fail(nullptr, "fpl_device_info: ctx is NULL");
  hipDeviceProp_t prop;
  FPL_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  if (name && name_cap) {
    snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
  }
  return 0;
}

int fpl_device_pci_bus_id(fpl_ctx *ctx, char *out, size_t cap) {
  if (!ctx || !out || cap < 16) return fpl_fail(ctx, "fpl_device_pci_bus_id: bad argument");
  FPL_HIP(ctx, hipDeviceGetPCIBusId(out, (int)cap, ctx->device));
  return 0;
}

int fpl_malloc(fpl_ctx *ctx, size_t bytes, void **dev_ptr) {
  if (!ctx || !dev_ptr) return fpl_fail(ctx, "fpl_malloc: NULL argument");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  FPL_HIP(ctx, hipMalloc(dev_ptr, bytes ? bytes : 16));
  return 0;
}

int fpl_free(fpl_ctx *ctx, void *dev_ptr) {
  if (!ctx) return fpl_fail(nullptr, "fpl_free: ctx is NULL");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  FPL_HIP(ctx, hipFree(dev_ptr));
  return 0;
}

int fpl_memcpy(fpl_ctx *ctx, void *dst, int dst_mem, const void *src,
               int src_mem, size_t bytes) {
  if (!ctx) return fpl_fail(nullptr, "fpl_memcpy: ctx is NULL");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  hipMemcpyKind kind =
      dst_mem == FPL_MEM_DEVICE
          ? (src_mem == FPL_MEM_DEVICE ? hipMemcpyDeviceToDevice
                                       : hipMemcpyHostToDevice)
          : (src_mem == FPL_MEM_DEVICE ? hipMemcpyDeviceToHost
                                       : hipMemcpyHostToHost);
  FPL_HIP(ctx, hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int fpl_timing_enable(fpl_ctx *ctx, int on) {
  if (!ctx) return fpl_fail(nullptr, "fpl_timing_enable: ctx is NULL");
  ctx->timing = on != 0;
  return 0;
}

static int drain_pending(fpl_ctx *ctx) {
  if (ctx->pending.empty()) return 0;
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto &p : ctx->pending) {
    float ms = 0.f;
    FPL_HIP(ctx, hipEventElapsedTime(&ms, p.start, p.stop));
    ctx->stats[p.name_id].ms += ms;
    ctx->stats[p.name_id].launches += 1;
    ctx->event_pool.push_back(p.start);
    ctx->event_pool.push_back(p.stop);
  }
  ctx->pending.clear();
  return 0;
}

int fpl_timing_reset(fpl_ctx *ctx) {
  if (!ctx) return fpl_fail(nullptr, "fpl_timing_reset: ctx is NULL");
  FPL_TRY(drain_pending(ctx));
  for (auto &s : ctx->stats) s = KernelStat();
  return 0;
}

int fpl_timing_get(fpl_ctx *ctx, char *names, double *ms, int64_t *launches,
                   int32_t cap, int32_t *n) {
  if (!ctx || !n) return fpl_fail(ctx, "fpl_timing_get: NULL argument");
  FPL_TRY(drain_pending(ctx));
  int32_t m = 0;
  for (size_t i = 0; i < ctx->stats.size() && m < cap; ++i) {
    if (ctx->stats[i].launches == 0) continue;
    if (names) {
      strncpy(names + (size_t)m * 64, ctx->stat_names[i].c_str(), 63);
      names[(size_t)m * 64 + 63] = 0;
    }
    if (ms) ms[m] = ctx->stats[i].ms;
    if (launches) launches[m] = ctx->stats[i].launches;
    ++m;
  }
  *n = m;
  return 0;
}

}  // extern "C"

// ---- device block cache ------------------------------------------------------
int fpl_dev_alloc(fpl_ctx *ctx, size_t bytes, void **out) {
  if (bytes == 0) bytes = 256;
  bytes = (bytes + 255) & ~size_t(255);
  auto it = ctx->free_blocks.lower_bound(bytes);
  // accept a cached block up to 1.25x the request
  if (it != ctx->free_blocks.end() && it->first <= bytes + bytes / 4) {
    *out = it->second;
    ctx->live_blocks[*out] = it->first;
    ctx->cached_bytes -= it->first;
    ctx->free_blocks.erase(it);
    return 0;
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    fpl_dev_trim(ctx);
    e = hipMalloc(out, bytes);
  }
  if (e != hipSuccess)
    return fpl_fail(ctx, "device allocation of %zu bytes failed: %s", bytes,
                    hipGetErrorString(e));
  ctx->live_blocks[*out] = bytes;
  return 0;
}

void fpl_dev_release(fpl_ctx *ctx, void *p) {
  if (!p) return;
  auto it = ctx->live_blocks.find(p);
  if (it == ctx->live_blocks.end()) return;
  ctx->free_blocks.emplace(it->second, p);
  ctx->cached_bytes += it->second;
  ctx->live_blocks.erase(it);
  // keep the cache bounded (long-running services): past 96 GiB give it all back
  if (ctx->cached_bytes > ((size_t)96 << 30)) fpl_dev_trim(ctx);
}

int fpl_dev_trim(fpl_ctx *ctx) {
  hipStreamSynchronize(ctx->stream);
  for (auto &kv : ctx->free_blocks) hipFree(kv.second);
  ctx->free_blocks.clear();
  ctx->cached_bytes = 0;
  return 0;
}

// ---- timing ---------------------------------------------------------------------
static hipEvent_t take_event(fpl_ctx *ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t ev = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return ev;
  }
  hipEvent_t ev = nullptr;
  hipEventCreate(&ev);
  return ev;
}

TimedLaunch::TimedLaunch(fpl_ctx *c, const char *name) : ctx(c), on(c->timing) {
  if (!on) return;
  auto it = ctx->stat_index.find(name);
  int id;
  if (it == ctx->stat_index.end()) {
    id = (int)ctx->stat_names.size();
    ctx->stat_names.push_back(name);
    ctx->stats.push_back(KernelStat());
    ctx->stat_index[name] = id;
  } else {
    id = it->second;
  }
  pt.name_id = id;
  pt.start = take_event(ctx);
  pt.stop = take_event(ctx);
  hipEventRecord(pt.start, ctx->stream);
}

TimedLaunch::~TimedLaunch() {
  if (!on) return;
  hipEventRecord(pt.stop, ctx->stream);
  ctx->pending.push_back(pt);
  // bound the number of live events
  if (ctx->pending.size() >= 4096) drain_pending(ctx);
}
