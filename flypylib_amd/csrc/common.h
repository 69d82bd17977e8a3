// Shared host-side plumbing of libfplhip.so: context, error reporting, device
// buffer cache, per-kernel HIP-event timing.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/fplhip.h"

#define FPL_MAX_ERR 1024
#define FPL_MAX_DEVICES 64

struct KernelStat {
  double ms = 0.0;
  int64_t launches = 0;
};

struct PendingTiming {
  int name_id;
  hipEvent_t start, stop;
};

// padded smoothed volume + NMS scratch kept between fpl_v2o_smooth / fpl_v2o_nms
struct V2oState {
  float *smoothed = nullptr;   // padded dims
  size_t cap_bytes = 0;
  unsigned long long *seg = nullptr;   // padded segmentation (fpl_v2o_set_seg) or null
  size_t seg_cap_bytes = 0;
  bool seg_valid = false;
  int64_t pdims[3] = {0, 0, 0};
  int32_t r = 0;
  bool valid = false;
  // per 4x4x4 cell: key of its largest positive voxel, taken by the fused y+x pass
  // while the values are in registers (v2o.hip); stale once the volume is edited
  unsigned long long *cellmax = nullptr;
  size_t cellmax_cap_bytes = 0;
  bool cellmax_valid = false;
  float floor = 0.f;            // fpl_v2o_set_floor: the NMS threshold will be >= this
  float cellmax_floor = 0.f;    // the floor the keys in `cellmax` were taken with
  // float64 predictions (v2o_f64.hip): the smoothed volume in double, its voxels sorted
  // by value (order statistics, dense ranks); `smoothed` then holds rank surrogates
  bool f64 = false;
  bool trunc_passes = false;    // fpl_v2o_set_integer: the next fpl_v2o_smooth_f64 filters an INTEGER volume
  double *smoothed64 = nullptr;
  size_t cap64_bytes = 0;
  unsigned long long *sort_keys = nullptr;     // ascending monotone keys of smoothed64
  unsigned int *sort_idx = nullptr;            // their flat indices
  size_t sort_cap = 0;                         // elements
  bool sorted = false;
};

// a training tensor restated as planar split halves (conv_mfma.hip, split build): made by the
// forward / input-gradient convolution of a step, read again by the weight-gradient kernel
struct FplSplitCopy {
  const void *key; int n, D, pad, C;   // the fp32 tensor (device pointer), patches, edge, zero shell, channels
  unsigned char *planar; float *sc;    // the copy (x s) and [s, 1 / s]
  bool sc_owned;                       // sc is an allocation of its own (the zero-record pool was spent)
  int64_t part;                        // bytes of one plane
};

struct fpl_ctx {
  int device = 0;
  std::vector<FplSplitCopy> split_copies;   // valid within one training step (fpl_tm_split_reset)
  std::vector<std::pair<const void *, unsigned *>> split_wmax;   // ... and the maxima of the weight sets seen in it
  // 64-B records (scale [0..2], maximum's bits [4]) handed out ZEROED to those copies and weight sets: one memset
  // of the used ones per step (fpl_tm_split_reset) instead of one per record - on the U-Net's small layers a
  // training step is mostly launches
  unsigned char *zero_pool = nullptr;
  int zero_next = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  char err[FPL_MAX_ERR] = {0};
  int n_cu = 0;
  // size-bucketed cache of device blocks (hipMalloc is slow and synchronising)
  std::multimap<size_t, void *> free_blocks;
  std::unordered_map<void *, size_t> live_blocks;
  size_t cached_bytes = 0;
  // timing
  bool timing = false;
  std::vector<std::string> stat_names;
  std::unordered_map<std::string, int> stat_index;
  std::vector<KernelStat> stats;
  std::vector<PendingTiming> pending;
  std::vector<hipEvent_t> event_pool;
  V2oState v2o;
  // RCCL communicator of this GPU (comm.hip); ncclComm_t kept opaque here.  fpl_comm_abort may
  // come from another host thread than the one inside a collective: `comm` changes hands under
  // comm_mu, comm_inflight counts the calls that hold a copy of it
  void *comm = nullptr;
  int comm_rank = 0, comm_nranks = 1;
  std::mutex comm_mu;
  int comm_inflight = 0;
  std::atomic<bool> comm_aborting{false};
  // executor chosen by the last fpl_infer_volume / fpl_program_forward (fpl_last_path)
  char last_path[64] = {0};
  // half-range guard of the split-operand kernels (mfma_util.h): a device word the kernels
  // OR into and its pinned host copy, read back at the end of fpl_infer_volume
  unsigned *range_flag_dev = nullptr;
  unsigned *range_flag_host = nullptr;
};

// the context's half-range flag word (allocated on first use)
int fpl_range_flag(fpl_ctx *ctx, unsigned **dev);

void fpl_comm_release(fpl_ctx *ctx);                  // comm.hip

extern thread_local char g_fpl_err[FPL_MAX_ERR];

int fpl_fail(fpl_ctx *ctx, const char *fmt, ...);
// "this network / input does not fit the IEEE-half range of the split-operand kernels":
// sets the message and returns FPL_RC_RANGE, which fpl_infer_volume turns into the fp32
// executor under FPL_PREC_AUTO and into an ordinary failure under FPL_PREC_F16S
#define FPL_RC_RANGE 3
// ... the same for a reason that belongs to THIS CALL's normalisation (mean / std folded into the
// first layer), not to the program's weights: 'auto' reruns the call in fp32 but does not pin
// the program to fp32 for later calls (fpl_fail_range_call)
#define FPL_RC_RANGE_CALL 4
int fpl_fail_range_call(fpl_ctx *ctx, const char *fmt, ...);
// bits of the half-range flag word: which kernel family raised it (for the message)
#define FPL_RANGE_INPUT 1u      // a normalised input voxel beyond the stem's input limit
#define FPL_RANGE_STEM 2u
#define FPL_RANGE_MID 4u
#define FPL_RANGE_TAIL 8u
#define FPL_RANGE_UNET 16u
int fpl_fail_range(fpl_ctx *ctx, const char *fmt, ...);

#define FPL_HIP(ctx, expr)                                                     \
  do {                                                                         \
    hipError_t e__ = (expr);                                                   \
    if (e__ != hipSuccess)                                                     \
      return fpl_fail((ctx), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,     \
                      hipGetErrorString(e__));                                 \
  } while (0)

#define FPL_TRY(expr)                                                          \
  do {                                                                         \
    int rc__ = (expr);                                                         \
    if (rc__ != 0) return rc__;                                                \
  } while (0)

#define FPL_REQUIRE(ctx, cond, ...)                                            \
  do {                                                                         \
    if (!(cond)) return fpl_fail((ctx), __VA_ARGS__);                          \
  } while (0)

// device block cache
int fpl_dev_alloc(fpl_ctx *ctx, size_t bytes, void **out);
void fpl_dev_release(fpl_ctx *ctx, void *p);          // back to the cache
int fpl_dev_trim(fpl_ctx *ctx);                       // hipFree everything cached

// RAII guard for temporaries inside one API call
struct DevTemp {
  fpl_ctx *ctx;
  std::vector<void *> ptrs;
  explicit DevTemp(fpl_ctx *c) : ctx(c) {}
  ~DevTemp() {
    for (void *p : ptrs) fpl_dev_release(ctx, p);
  }
  int alloc(size_t bytes, void **out) {
    int rc = fpl_dev_alloc(ctx, bytes, out);
    if (rc == 0) ptrs.push_back(*out);
    return rc;
  }
  void release(void *p) {
    for (size_t i = 0; i < ptrs.size(); ++i)
      if (ptrs[i] == p) {
        ptrs.erase(ptrs.begin() + i);
        fpl_dev_release(ctx, p);
        return;
      }
  }
};

// timing scope: records start/stop events around a launch when enabled
struct TimedLaunch {
  fpl_ctx *ctx;
  bool on;
  PendingTiming pt;
  TimedLaunch(fpl_ctx *c, const char *name);
  ~TimedLaunch();
};

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
