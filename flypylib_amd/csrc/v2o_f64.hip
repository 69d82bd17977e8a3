// voxel2obj on a float64 prediction volume.  The reference pads and smooths in the array's
// own dtype (flypylib/fplobjdetect.py:158-168: np.pad, scipy gaussian_filter with a float64
// output for a float64 input), takes np.percentile in float64 and compares / reports
// float64 values - results that differ from those of the float32 pipeline (no rounding to
// float32 between the three smoothing passes, ties decided on 53-bit values).  Its own
// callers never produce such a volume (FplNetwork.infer returns float32), so this is a
// completeness path, built from simple kernels around the float32 machinery:
//
//   fpl_v2o_smooth_f64   pad, three separable passes in fp64 - scipy's operation order, no
//                        FMA, NO rounding between the axes - margin zeroing -> smoothed64
//   fpl_v2o_select_f64   radix sort (rocPRIM) of all padded voxels by value, flat index
//                        as payload: exact order statistics for the percentile
//   fpl_v2o_rank_f64     the NMS only needs the ORDER of the candidates (arg-max with the
//                        lowest index winning ties, `> thresh`): every candidate gets its
//                        dense rank among the candidate values as a float32 surrogate
//                        (< 2^24 distinct values: exact), everything else 0 -> `smoothed`;
//                        fpl_v2o_nms / fpl_v2o_nms_seg then run unchanged with thresh 0.5
//   fpl_v2o_values_f64   the float64 confidences of the picked voxels
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

#pragma clang fp contract(off)      // every product and sum rounded on its own, as in v2o.hip

namespace {

__device__ __forceinline__ double mul_rn(double a, double b) { return a * b; }
__device__ __forceinline__ double add_rn(double a, double b) { return a + b; }

__device__ __forceinline__ int64_t reflect_idx(int64_t i, int64_t n) {
  const int64_t p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return i >= n ? p - 1 - i : i;
}

// zero-padded copy: np.pad(pred, r, 'constant')
__global__ void pad_f64(const double *__restrict__ pred, int64_t D0, int64_t D1, int64_t D2, int r,
                        double *__restrict__ out, int64_t P1, int64_t P2, int64_t n_pad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  const int64_t x = i % P2 - r, y = (i / P2) % P1 - r, z = i / (P2 * P1) - r;
  out[i] = (z < 0 || y < 0 || x < 0 || z >= D0 || y >= D1 || x >= D2)
               ? 0.0 : pred[(z * D1 + y) * D2 + x];
}

// one separable pass (scipy ni_filters.c, symmetric branch, 'reflect'): acc = x[0] w[0];
// for j = R .. 1: acc += (x[-j] + x[+j]) * w[j].  AXIS 2 also zeroes the outer r shell.
template <int AXIS>
__global__ __launch_bounds__(256) void gauss_pass_f64(const double *__restrict__ in,
                                                      double *__restrict__ out, int64_t P0,
                                                      int64_t P1, int64_t P2,
                                                      const double *__restrict__ w, int wr, int r,
                                                      int trunc_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P0 * P1 * P2) return;
  const int64_t x = i % P2, y = (i / P2) % P1, z = i / (P2 * P1);
  auto load = [&](int64_t d) -> double {
    if (AXIS == 0) return in[(reflect_idx(z + d, P0) * P1 + y) * P2 + x];
    if (AXIS == 1) return in[(z * P1 + reflect_idx(y + d, P1)) * P2 + x];
    return in[(z * P1 + y) * P2 + reflect_idx(x + d, P2)];
  };
  double acc = mul_rn(load(0), w[0]);
  for (int j = wr; j >= 1; --j) acc = add_rn(acc, mul_rn(add_rn(load(-j), load(j)), w[j]));
  // an integer volume: scipy stores every pass in the array's own type - a C cast of the
  // double, i.e. truncation toward zero (the values stay in range: a convex combination)
  if (trunc_out) acc = __builtin_trunc(acc);
  if (AXIS == 2 && r > 0 &&
      (z < r || y < r || x < r || z >= P0 - r || y >= P1 - r || x >= P2 - r))
    acc = 0.0;
  out[i] = acc;
}

// monotone unsigned key of a double (-0.0 and +0.0 share a key: they compare equal)
__host__ __device__ inline unsigned long long key_of(double v) {
  if (v == 0.0) v = 0.0;
  unsigned long long u;
  memcpy(&u, &v, 8);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__host__ inline double value_of(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  double v;
  memcpy(&v, &u, 8);
  return v;
}

__global__ void make_keys(const double *__restrict__ v, unsigned long long *__restrict__ keys,
                          unsigned int *__restrict__ idx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = key_of(v[i]);
  idx[i] = (unsigned int)i;
}

// flag[i - first] = 1 where sorted position i starts a new value (first itself included)
__global__ void new_value_flags(const unsigned long long *__restrict__ keys, int64_t first,
                                int64_t n, unsigned int *__restrict__ flags) {
  const int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flags[i - first] = (i == first || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ void scatter_ranks(const unsigned int *__restrict__ idx, int64_t first, int64_t n,
                              const unsigned int *__restrict__ ranks, float *__restrict__ out) {
  const int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[idx[i]] = (float)ranks[i - first];
}

__global__ void gather_f64(const double *__restrict__ v, const long long *__restrict__ flat,
                           int64_t n, double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v[flat[i]];
}

int grow(fpl_ctx *ctx, void **p, size_t *cap, size_t need) {
  if (*cap >= need) return 0;
  if (*p) fpl_dev_release(ctx, *p);
  *p = nullptr;
  *cap = 0;
  FPL_TRY(fpl_dev_alloc(ctx, need, p));
  *cap = need;
  return 0;
}

}  // namespace

extern "C" {

int fpl_v2o_smooth_f64(fpl_ctx *ctx, const double *pred, int pred_mem, const int64_t dims[3],
                       int32_t r, const double *weights, int32_t wr) {
  // integer mode (fpl_v2o_set_integer) holds for ONE call: taken and cleared before anything
  // below can fail, so that a failed call never leaves it set for the next float64 volume
  int tr = 0;
  if (ctx) {
    tr = ctx->v2o.trunc_passes ? 1 : 0;
    ctx->v2o.trunc_passes = false;
  }
  if (!ctx || !pred || !dims || !weights)
    return fpl_fail(ctx, "fpl_v2o_smooth_f64: NULL argument");
  FPL_REQUIRE(ctx, r >= 0 && wr >= 0, "fpl_v2o_smooth_f64: negative radius");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  int64_t P[3];
  for (int a = 0; a < 3; ++a) {
    FPL_REQUIRE(ctx, dims[a] > 0, "fpl_v2o_smooth_f64: dims[%d] = %lld", a, (long long)dims[a]);
    P[a] = dims[a] + 2 * (int64_t)r;
  }
  const int64_t n_pad = P[0] * P[1] * P[2];
  FPL_REQUIRE(ctx, n_pad < ((int64_t)1 << 32) - 1,
              "fpl_v2o_smooth_f64: padded volume has %lld voxels; the NMS keys hold 32-bit "
              "flat indices - process it as substacks", (long long)n_pad);
  hipStream_t st = ctx->stream;
  DevTemp tmp(ctx);
  V2oState &S = ctx->v2o;
  S.valid = false;
  S.seg_valid = false;
  S.sorted = false;
  S.cellmax_valid = false;
  S.floor = 0.f;
  {
    void *p = S.smoothed;
    FPL_TRY(grow(ctx, &p, &S.cap_bytes, (size_t)n_pad * sizeof(float)));
    S.smoothed = (float *)p;
    p = S.smoothed64;
    FPL_TRY(grow(ctx, &p, &S.cap64_bytes, (size_t)n_pad * sizeof(double)));
    S.smoothed64 = (double *)p;
    // per-cell key array of the NMS (sized as fpl_v2o_smooth does)
    const size_t need = (size_t)(ceil_div64(P[0], 4) * ceil_div64(P[1], 4) * ceil_div64(P[2], 4)) *
                        sizeof(unsigned long long);
    p = S.cellmax;
    FPL_TRY(grow(ctx, &p, &S.cellmax_cap_bytes, need));
    S.cellmax = (unsigned long long *)p;
  }
  const double *pred_dev = pred;
  void *p;
  if (pred_mem == FPL_MEM_HOST) {
    const size_t nb = (size_t)(dims[0] * dims[1] * dims[2]) * sizeof(double);
    FPL_TRY(tmp.alloc(nb, &p));
    FPL_HIP(ctx, hipMemcpyAsync(p, pred, nb, hipMemcpyHostToDevice, st));
    pred_dev = (const double *)p;
  }
  FPL_TRY(tmp.alloc((size_t)n_pad * sizeof(double), &p));
  double *scratch = (double *)p;
  FPL_TRY(tmp.alloc((size_t)(wr + 1) * sizeof(double), &p));
  double *w_dev = (double *)p;
  // weights[0..2wr] symmetric; the kernel wants w[j] by distance j
  FPL_HIP(ctx, hipMemcpyAsync(w_dev, weights + wr, (size_t)(wr + 1) * sizeof(double),
                              hipMemcpyHostToDevice, st));
  const unsigned grid = (unsigned)ceil_div64(n_pad, 256);
  {
    TimedLaunch tl(ctx, "v2o64_pad");
    pad_f64<<<grid, 256, 0, st>>>(pred_dev, dims[0], dims[1], dims[2], r, S.smoothed64, P[1], P[2], n_pad);
  }
  {
    TimedLaunch tl(ctx, "v2o64_gauss");
    gauss_pass_f64<0><<<grid, 256, 0, st>>>(S.smoothed64, scratch, P[0], P[1], P[2], w_dev, wr, r, tr);
    gauss_pass_f64<1><<<grid, 256, 0, st>>>(scratch, S.smoothed64, P[0], P[1], P[2], w_dev, wr, r, tr);
    gauss_pass_f64<2><<<grid, 256, 0, st>>>(S.smoothed64, scratch, P[0], P[1], P[2], w_dev, wr, r, tr);
  }
  FPL_HIP(ctx, hipGetLastError());
  FPL_HIP(ctx, hipMemcpyAsync(S.smoothed64, scratch, (size_t)n_pad * sizeof(double),
                              hipMemcpyDeviceToDevice, st));
  FPL_HIP(ctx, hipStreamSynchronize(st));
  for (int a = 0; a < 3; ++a) S.pdims[a] = P[a];
  S.r = r;
  S.f64 = true;
  S.valid = true;          // dims and radius are set; `smoothed` is filled by fpl_v2o_rank_f64
  return 0;
}

int fpl_v2o_set_integer(fpl_ctx *ctx, int32_t on) {
  if (!ctx) return fpl_fail(nullptr, "fpl_v2o_set_integer: ctx is NULL");
  ctx->v2o.trunc_passes = on != 0;
  return 0;
}

int fpl_v2o_select_f64(fpl_ctx *ctx, const int64_t *ranks, int32_t n_ranks, double *rank_values) {
  if (!ctx || (n_ranks > 0 && (!ranks || !rank_values)))
    return fpl_fail(ctx, "fpl_v2o_select_f64: NULL argument");
  V2oState &S = ctx->v2o;
  FPL_REQUIRE(ctx, S.valid && S.f64, "fpl_v2o_select_f64: call fpl_v2o_smooth_f64 first");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int64_t n = S.pdims[0] * S.pdims[1] * S.pdims[2];
  for (int i = 0; i < n_ranks; ++i)
    FPL_REQUIRE(ctx, ranks[i] >= 0 && ranks[i] < n, "fpl_v2o_select_f64: rank %lld out of range",
                (long long)ranks[i]);
  if (S.sort_cap < (size_t)n) {
    if (S.sort_keys) fpl_dev_release(ctx, S.sort_keys);
    if (S.sort_idx) fpl_dev_release(ctx, S.sort_idx);
    S.sort_keys = nullptr; S.sort_idx = nullptr; S.sort_cap = 0;
    void *p;
    FPL_TRY(fpl_dev_alloc(ctx, (size_t)n * 8, &p));
    S.sort_keys = (unsigned long long *)p;
    FPL_TRY(fpl_dev_alloc(ctx, (size_t)n * 4, &p));
    S.sort_idx = (unsigned int *)p;
    S.sort_cap = (size_t)n;
  }
  DevTemp tmp(ctx);
  void *p;
  FPL_TRY(tmp.alloc((size_t)n * 8, &p));
  unsigned long long *k_in = (unsigned long long *)p;
  FPL_TRY(tmp.alloc((size_t)n * 4, &p));
  unsigned int *v_in = (unsigned int *)p;
  {
    TimedLaunch tl(ctx, "v2o64_sort");
    make_keys<<<(unsigned)ceil_div64(n, 256), 256, 0, st>>>(S.smoothed64, k_in, v_in, n);
    size_t tb = 0;
    FPL_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, k_in, S.sort_keys, v_in, S.sort_idx,
                                           (size_t)n, 0, 64, st));
    FPL_TRY(tmp.alloc(tb + 16, &p));
    // equal keys keep their input order (LSD radix sort is stable): ties stay in flat-index
    // order, which is what the dense ranks and np.argmax's first-hit rule need
    FPL_HIP(ctx, rocprim::radix_sort_pairs(p, tb, k_in, S.sort_keys, v_in, S.sort_idx, (size_t)n,
                                           0, 64, st));
  }
  for (int i = 0; i < n_ranks; ++i) {
    unsigned long long k;
    FPL_HIP(ctx, hipMemcpyAsync(&k, S.sort_keys + ranks[i], 8, hipMemcpyDeviceToHost, st));
    FPL_HIP(ctx, hipStreamSynchronize(st));
    rank_values[i] = value_of(k);
  }
  FPL_HIP(ctx, hipStreamSynchronize(st));
  S.sorted = true;
  return 0;
}

int fpl_v2o_rank_f64(fpl_ctx *ctx, double thresh, int64_t *n_candidates) {
  if (!ctx) return fpl_fail(nullptr, "fpl_v2o_rank_f64: ctx is NULL");
  V2oState &S = ctx->v2o;
  FPL_REQUIRE(ctx, S.valid && S.f64 && S.sorted,
              "fpl_v2o_rank_f64: call fpl_v2o_smooth_f64 and fpl_v2o_select_f64 first");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int64_t n = S.pdims[0] * S.pdims[1] * S.pdims[2];
  // candidates: value > thresh (and > 0, the reference's stop rule): the suffix of the
  // sorted order behind the last key <= max(thresh, 0) - binary search, 8-byte probes
  const unsigned long long kt = key_of(thresh > 0.0 ? thresh : 0.0);
  int64_t lo = 0, hi = n;                 // first position with key > kt
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    unsigned long long k;
    FPL_HIP(ctx, hipMemcpyAsync(&k, S.sort_keys + mid, 8, hipMemcpyDeviceToHost, st));
    FPL_HIP(ctx, hipStreamSynchronize(st));
    if (k > kt) hi = mid; else lo = mid + 1;
  }
  const int64_t first = lo, m = n - first;
  if (n_candidates) *n_candidates = m;
  FPL_HIP(ctx, hipMemsetAsync(S.smoothed, 0, (size_t)n * sizeof(float), st));
  S.cellmax_valid = false;
  if (m > 0) {
    DevTemp tmp(ctx);
    void *p;
    FPL_TRY(tmp.alloc((size_t)m * 4, &p));
    unsigned int *flags = (unsigned int *)p;
    FPL_TRY(tmp.alloc((size_t)m * 4, &p));
    unsigned int *rk = (unsigned int *)p;
    const unsigned grid = (unsigned)ceil_div64(m, 256);
    TimedLaunch tl(ctx, "v2o64_rank");
    new_value_flags<<<grid, 256, 0, st>>>(S.sort_keys, first, n, flags);
    size_t tb = 0;
    FPL_HIP(ctx, rocprim::inclusive_scan(nullptr, tb, flags, rk, (size_t)m, rocprim::plus<unsigned int>(), st));
    FPL_TRY(tmp.alloc(tb + 16, &p));
    FPL_HIP(ctx, rocprim::inclusive_scan(p, tb, flags, rk, (size_t)m, rocprim::plus<unsigned int>(), st));
    unsigned int distinct = 0;
    FPL_HIP(ctx, hipMemcpyAsync(&distinct, rk + (m - 1), 4, hipMemcpyDeviceToHost, st));
    FPL_HIP(ctx, hipStreamSynchronize(st));
    FPL_REQUIRE(ctx, distinct < (1u << 24),
                "fpl_v2o_rank_f64: %u distinct candidate values above the threshold; the float32 "
                "rank surrogates are exact below 2^24 - process the volume as substacks", distinct);
    scatter_ranks<<<grid, 256, 0, st>>>(S.sort_idx, first, n, rk, S.smoothed);
    FPL_HIP(ctx, hipGetLastError());
    FPL_HIP(ctx, hipStreamSynchronize(st));
  }
  FPL_HIP(ctx, hipStreamSynchronize(st));
  return 0;
}

int fpl_v2o_values_f64(fpl_ctx *ctx, const int64_t *flat, int64_t n, double *out) {
  if (!ctx || (n > 0 && (!flat || !out))) return fpl_fail(ctx, "fpl_v2o_values_f64: NULL argument");
  V2oState &S = ctx->v2o;
  FPL_REQUIRE(ctx, S.f64 && S.smoothed64, "fpl_v2o_values_f64: no float64 volume");
  if (n == 0) return 0;
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int64_t n_pad = S.pdims[0] * S.pdims[1] * S.pdims[2];
  for (int64_t i = 0; i < n; ++i)
    FPL_REQUIRE(ctx, flat[i] >= 0 && flat[i] < n_pad, "fpl_v2o_values_f64: index out of range");
  DevTemp tmp(ctx);
  void *p;
  FPL_TRY(tmp.alloc((size_t)n * 8, &p));
  long long *f_dev = (long long *)p;
  FPL_TRY(tmp.alloc((size_t)n * 8, &p));
  double *o_dev = (double *)p;
  FPL_HIP(ctx, hipMemcpyAsync(f_dev, flat, (size_t)n * 8, hipMemcpyHostToDevice, st));
  gather_f64<<<(unsigned)ceil_div64(n, 256), 256, 0, st>>>(S.smoothed64, f_dev, n, o_dev);
  FPL_HIP(ctx, hipMemcpyAsync(out, o_dev, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  FPL_HIP(ctx, hipStreamSynchronize(st));
  return 0;
}

}  // extern "C"
