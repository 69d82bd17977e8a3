// Training step engine: forward in training mode, binary cross-entropy, backward,
// Adam.  Replaces the per-step work of `train_network.fit_generator`
// (flypylib/fplnetwork.py:112-122) with the Keras-layer semantics of SURVEY.md
// section 8a rows M3-M5:
//   BatchNormalization: batch statistics over (N,D,H,W), biased variance,
//     eps 1e-3, moving = 0.99*moving + 0.01*batch
//   Dropout(rate): inverted scaling, mask from a counter-based hash of
//     (seed, layer, element)  (flypylib_amd/synth.py::dropout_mask is the same)
//   binary_crossentropy: p clipped to [1e-7, 1-1e-7], evaluated through logits
//   Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)
// fp32 throughout; one simple kernel per layer (round-1 correctness path - the
// MFMA forward/backward kernels replace the convs later).
#include <algorithm>
#include <cmath>

#include "common.h"
#include "conv_direct.h"
#include "fast_paths.h"

namespace {

struct TShape {
  int d = 0, h = 0, w = 0, c = 0;
  int64_t vox() const { return (int64_t)d * h * w; }
  int64_t elems() const { return vox() * c; }
};

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

inline unsigned g1(int64_t n) { return (unsigned)ceil_div64(n, 256); }

// ---- per-channel reductions over rows of an [M][C] tensor -----------------------
// block = (C, R) threads; each block covers rows blockIdx.x*ROWS .. ; partial
// sums in double -> part[block][2][C]
// rows per block: enough blocks to fill the chip on small tensors, <= 4096 rows on big
static inline int red_rows(int64_t M) {
  const int64_t r = (ceil_div64(M, 2048) + 63) / 64 * 64;
  return (int)std::min<int64_t>(4096, std::max<int64_t>(256, r));
}

template <int MODE>   // 0: (x, x^2)   1: (dy, dy*xhat)   2: (dy, 0)
                      // 3: as 1 with dy gated by the ReLU output y3 > 0
__global__ void chan_reduce_partial(const float *__restrict__ a,
                                    const float *__restrict__ b,
                                    const float *__restrict__ mean,
                                    const float *__restrict__ invstd, int64_t M,
                                    int C, int rows, double *__restrict__ part,
                                    const float *__restrict__ y3 = nullptr) {
  extern __shared__ double red[];
  const int c = threadIdx.x, r = threadIdx.y, R = blockDim.y;
  const int64_t row0 = (int64_t)blockIdx.x * rows;
  const int64_t row1 = row0 + rows < M ? row0 + rows : M;
  double s0 = 0.0, s1 = 0.0;
  const float mu = (MODE == 1 || MODE == 3) ? mean[c] : 0.f,
              is = (MODE == 1 || MODE == 3) ? invstd[c] : 0.f;
  for (int64_t m = row0 + r; m < row1; m += R) {
    float v = a[m * C + c];
    if (MODE == 3) v = y3[m * C + c] > 0.f ? v : 0.f;
    if (MODE == 0) {
      s0 += v; s1 += (double)v * v;
    } else if (MODE == 1 || MODE == 3) {
      const float xh = (b[m * C + c] - mu) * is;
      s0 += v; s1 += (double)v * xh;
    } else {
      s0 += v;
    }
  }
  red[(r * C + c) * 2 + 0] = s0;
  red[(r * C + c) * 2 + 1] = s1;
  __syncthreads();
  if (r == 0) {
    for (int k = 1; k < R; ++k) {
      s0 += red[(k * C + c) * 2 + 0];
      s1 += red[(k * C + c) * 2 + 1];
    }
    part[((int64_t)blockIdx.x * 2 + 0) * C + c] = s0;
    part[((int64_t)blockIdx.x * 2 + 1) * C + c] = s1;
  }
}

// BN affine output with a fixed rounding sequence: the forward pass and the ReLU mask
// recomputed by the backward passes (instead of re-reading y) must agree bit for bit
__device__ __forceinline__ float bn_affine(float v, float m, float s, float g, float b) {
  return __fmaf_rn(__fmul_rn(__fsub_rn(v, m), s), g, b);
}

// float4 form of the same reduction for C % 4 == 0: block = (C/4, R) threads, a thread
// owns 4 channels and streams 16-B pieces, four rows in flight
template <int MODE>
__global__ void chan_reduce_partial4(const float *__restrict__ a,
                                     const float *__restrict__ b,
                                     const float *__restrict__ mean,
                                     const float *__restrict__ invstd, int64_t M,
                                     int C, int rows, double *__restrict__ part,
                                     const float *__restrict__ gamma = nullptr,
                                     const float *__restrict__ beta = nullptr) {
  // MODE 3 = MODE 1 behind a fused ReLU: dy counts where BN's output was positive
  extern __shared__ double red[];
  const int c4 = threadIdx.x, r = threadIdx.y, R = blockDim.y, C4 = C / 4;
  const int64_t row0 = (int64_t)blockIdx.x * rows;
  const int64_t row1 = row0 + rows < M ? row0 + rows : M;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu, ga = mu, be = mu;
  if (MODE == 1 || MODE == 3) {
    mu = reinterpret_cast<const float4 *>(mean)[c4];
    is = reinterpret_cast<const float4 *>(invstd)[c4];
  }
  if (MODE == 3) {
    ga = reinterpret_cast<const float4 *>(gamma)[c4];
    be = reinterpret_cast<const float4 *>(beta)[c4];
  }
  const float4 *a4 = reinterpret_cast<const float4 *>(a);
  const float4 *b4 = reinterpret_cast<const float4 *>(b);
  auto add = [&](const float4 &va, const float4 &vb) {
    float v[4] = {va.x, va.y, va.z, va.w};
    const float xb[4] = {vb.x, vb.y, vb.z, vb.w};
    const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, i4[4] = {is.x, is.y, is.z, is.w};
    const float g4[4] = {ga.x, ga.y, ga.z, ga.w}, e4[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (MODE == 3) v[q] = bn_affine(xb[q], m4[q], i4[q], g4[q], e4[q]) > 0.f ? v[q] : 0.f;
      if (MODE == 0) {
        s0[q] += v[q]; s1[q] += (double)v[q] * v[q];
      } else if (MODE == 1 || MODE == 3) {
        const float xh = (xb[q] - m4[q]) * i4[q];
        s0[q] += v[q]; s1[q] += (double)v[q] * xh;
      } else {
        s0[q] += v[q];
      }
    }
  };
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t m = row0 + r;
  for (; m + 3 * R < row1; m += 4 * R) {
    float4 va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t idx = (m + u * R) * C4 + c4;
      va[u] = a4[idx];
      vb[u] = (MODE == 1 || MODE == 3) ? b4[idx] : z4;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) add(va[u], vb[u]);
  }
  for (; m < row1; m += R) {
    const int64_t idx = m * C4 + c4;
    add(a4[idx], (MODE == 1 || MODE == 3) ? b4[idx] : z4);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    red[(r * C + 4 * c4 + q) * 2 + 0] = s0[q];
    red[(r * C + 4 * c4 + q) * 2 + 1] = s1[q];
  }
  __syncthreads();
  // the first C threads of the block finish one channel each
  const int t = threadIdx.y * blockDim.x + threadIdx.x;
  if (t < C) {
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < R; ++k) {
      t0 += red[(k * C + t) * 2 + 0];
      t1 += red[(k * C + t) * 2 + 1];
    }
    part[((int64_t)blockIdx.x * 2 + 0) * C + t] = t0;
    part[((int64_t)blockIdx.x * 2 + 1) * C + t] = t1;
  }
}

// float4 forms of the BN (+ReLU) apply / backward passes (C % 4 == 0; n4 = n / 4)
template <bool RELU>
__global__ void bn_apply4(const float4 *__restrict__ x, const float4 *__restrict__ mean,
                          const float4 *__restrict__ invstd, const float4 *__restrict__ gamma,
                          const float4 *__restrict__ beta, float4 *__restrict__ y,
                          int64_t n4, int C4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)(i % C4);
  const float4 v = x[i], m = mean[c], s = invstd[c], g = gamma[c], b = beta[c];
  float4 o;
  o.x = bn_affine(v.x, m.x, s.x, g.x, b.x); o.y = bn_affine(v.y, m.y, s.y, g.y, b.y);
  o.z = bn_affine(v.z, m.z, s.z, g.z, b.z); o.w = bn_affine(v.w, m.w, s.w, g.w, b.w);
  if (RELU) {
    o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
  }
  y[i] = o;
}

template <bool RELU, bool ACC>
__global__ void bn_backward4(const float4 *__restrict__ dy, const float4 *__restrict__ beta,
                             const float4 *__restrict__ x, const float4 *__restrict__ mean,
                             const float4 *__restrict__ invstd,
                             const float4 *__restrict__ gamma,
                             const float4 *__restrict__ sum_g,
                             const float4 *__restrict__ sum_g_xhat, float4 *__restrict__ dx,
                             int64_t n4, int C4, float inv_m) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)(i % C4);
  const float4 d = dy[i], xv = x[i], m = mean[c], s = invstd[c], g = gamma[c],
               sg = sum_g[c], sx = sum_g_xhat[c];
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (RELU) bv = beta[c];                  // the ReLU mask is recomputed from x
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ACC) o = dx[i];                      // else this pass is the tensor's only writer
  const float dd[4] = {d.x, d.y, d.z, d.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w},
              mm[4] = {m.x, m.y, m.z, m.w}, ss[4] = {s.x, s.y, s.z, s.w},
              gg[4] = {g.x, g.y, g.z, g.w}, s0[4] = {sg.x, sg.y, sg.z, sg.w},
              s1[4] = {sx.x, sx.y, sx.z, sx.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
  float oo[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float gq = (!RELU || bn_affine(xx[q], mm[q], ss[q], gg[q], bb[q]) > 0.f) ? dd[q] : 0.f;
    const float xh = (xx[q] - mm[q]) * ss[q];
    oo[q] += gg[q] * ss[q] * (gq - inv_m * s0[q] - xh * inv_m * s1[q]);
  }
  dx[i] = make_float4(oo[0], oo[1], oo[2], oo[3]);
}

// ---- BN + ReLU + MaxPooling3D(2) as one layer (exactly tiling windows, C % 4 == 0).
// Forward: a thread owns four channels of one pooling window, reads the eight BN
// inputs, and writes the pooled ReLU output and the arg-max bytes; the full-resolution
// ReLU output is never stored.  Backward: the gradient of that output is non-zero only
// at the arg-max position (and only where the output was positive), so both BN passes
// read x once plus the pooled gradient (1/8 of the size) and nothing else.
struct PoolGeo { int D, H, W, od, oh, ow; };      // BN tensor dims, pooled dims

__device__ __forceinline__ int64_t pool_base4(const PoolGeo &g, int64_t row, int C4) {
  // row = pooled voxel (t, oz, oy, ox) -> float4 index of window position 0, channel 0
  int64_t t = row;
  const int ox = (int)(t % g.ow); t /= g.ow;
  const int oy = (int)(t % g.oh); t /= g.oh;
  const int oz = (int)(t % g.od); t /= g.od;
  return ((((t * g.D + 2 * oz) * g.H + 2 * oy) * (int64_t)g.W + 2 * ox)) * C4;
}
__device__ __forceinline__ int64_t pool_pos4(const PoolGeo &g, int p, int C4) {
  return ((int64_t)((p >> 2) * g.H + ((p >> 1) & 1)) * g.W + (p & 1)) * C4;
}

__global__ void bn_relu_pool4(const float4 *__restrict__ x, const float4 *__restrict__ mean,
                              const float4 *__restrict__ invstd, const float4 *__restrict__ gamma,
                              const float4 *__restrict__ beta, float4 *__restrict__ y,
                              uint32_t *__restrict__ arg, int64_t n4, int C4, PoolGeo g) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)(i % C4);
  const int64_t base = pool_base4(g, i / C4, C4) + c;
  const float4 m = mean[c], s = invstd[c], ga = gamma[c], be = beta[c];
  float4 v[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) v[p] = x[base + pool_pos4(g, p, C4)];
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  uint32_t am[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const float o[4] = {fmaxf(bn_affine(v[p].x, m.x, s.x, ga.x, be.x), 0.f),
                        fmaxf(bn_affine(v[p].y, m.y, s.y, ga.y, be.y), 0.f),
                        fmaxf(bn_affine(v[p].z, m.z, s.z, ga.z, be.z), 0.f),
                        fmaxf(bn_affine(v[p].w, m.w, s.w, ga.w, be.w), 0.f)};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (o[q] > best[q]) { best[q] = o[q]; am[q] = (uint32_t)p; }   // first maximum wins
  }
  y[i] = make_float4(best[0], best[1], best[2], best[3]);
  arg[i] = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
}

// partial sums (sum g, sum g * xhat) of the masked gradient, rows = pooling windows
__global__ void chan_reduce_pool4(const float4 *__restrict__ dyp, const uint32_t *__restrict__ arg,
                                  const float4 *__restrict__ x, const float *__restrict__ mean,
                                  const float *__restrict__ invstd, const float *__restrict__ gamma,
                                  const float *__restrict__ beta, int64_t Mp, int C, int rows,
                                  double *__restrict__ part, PoolGeo g) {
  extern __shared__ double red[];
  const int c4 = threadIdx.x, r = threadIdx.y, R = blockDim.y, C4 = C / 4;
  const int64_t row0 = (int64_t)blockIdx.x * rows;
  const int64_t row1 = row0 + rows < Mp ? row0 + rows : Mp;
  const float4 mu = reinterpret_cast<const float4 *>(mean)[c4];
  const float4 is = reinterpret_cast<const float4 *>(invstd)[c4];
  const float4 ga = reinterpret_cast<const float4 *>(gamma)[c4];
  const float4 be = reinterpret_cast<const float4 *>(beta)[c4];
  const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, i4[4] = {is.x, is.y, is.z, is.w};
  const float g4[4] = {ga.x, ga.y, ga.z, ga.w}, e4[4] = {be.x, be.y, be.z, be.w};
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  for (int64_t m = row0 + r; m < row1; m += R) {
    const int64_t base = pool_base4(g, m, C4) + c4;
    float4 v[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) v[p] = x[base + pool_pos4(g, p, C4)];
    const float4 d = dyp[m * C4 + c4];
    const uint32_t a = arg[m * C4 + c4];
    const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = (int)((a >> (8 * q)) & 255u);
      float xv = 0.f;
#pragma unroll
      for (int pp = 0; pp < 8; ++pp) {
        const float cand = q == 0 ? v[pp].x : q == 1 ? v[pp].y : q == 2 ? v[pp].z : v[pp].w;
        xv = pp == p ? cand : xv;
      }
      const float gq = bn_affine(xv, m4[q], i4[q], g4[q], e4[q]) > 0.f ? dd[q] : 0.f;
      const float xh = (xv - m4[q]) * i4[q];
      s0[q] += gq; s1[q] += (double)gq * xh;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    red[(r * C + 4 * c4 + q) * 2 + 0] = s0[q];
    red[(r * C + 4 * c4 + q) * 2 + 1] = s1[q];
  }
  __syncthreads();
  const int t = threadIdx.y * blockDim.x + threadIdx.x;
  if (t < C) {
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < R; ++k) {
      t0 += red[(k * C + t) * 2 + 0];
      t1 += red[(k * C + t) * 2 + 1];
    }
    part[((int64_t)blockIdx.x * 2 + 0) * C + t] = t0;
    part[((int64_t)blockIdx.x * 2 + 1) * C + t] = t1;
  }
}

template <bool ACC>
__global__ void bn_backward_pool4(const float4 *__restrict__ dyp, const uint32_t *__restrict__ arg,
                                  const float4 *__restrict__ x, const float4 *__restrict__ mean,
                                  const float4 *__restrict__ invstd, const float4 *__restrict__ gamma,
                                  const float4 *__restrict__ beta, const float4 *__restrict__ sum_g,
                                  const float4 *__restrict__ sum_g_xhat, float4 *__restrict__ dx,
                                  int64_t n4, int C4, float inv_m, PoolGeo g) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)(i % C4);
  const int64_t base = pool_base4(g, i / C4, C4) + c;
  const float4 m = mean[c], s = invstd[c], ga = gamma[c], be = beta[c], sg = sum_g[c],
               sx = sum_g_xhat[c], d = dyp[i];
  const uint32_t a = arg[i];
  const float mm[4] = {m.x, m.y, m.z, m.w}, ss[4] = {s.x, s.y, s.z, s.w},
              gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w},
              s0[4] = {sg.x, sg.y, sg.z, sg.w}, s1[4] = {sx.x, sx.y, sx.z, sx.w},
              dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int64_t idx = base + pool_pos4(g, p, C4);
    const float4 xv = x[idx];
    const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ACC) o = dx[idx];
    float oo[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool hit = (int)((a >> (8 * q)) & 255u) == p &&
                       bn_affine(xx[q], mm[q], ss[q], gg[q], bb[q]) > 0.f;
      const float gq = hit ? dd[q] : 0.f;
      const float xh = (xx[q] - mm[q]) * ss[q];
      oo[q] += gg[q] * ss[q] * (gq - inv_m * s0[q] - xh * inv_m * s1[q]);
    }
    dx[idx] = make_float4(oo[0], oo[1], oo[2], oo[3]);
  }
}

// sum of the nb partials of channel c, by one 256-thread block (fixed tree: the result
// does not depend on scheduling); valid in thread 0
__device__ __forceinline__ void block_sum_partials(const double *__restrict__ part, int nb,
                                                   int C, int c, double &s0, double &s1) {
  __shared__ double sh[2][256];
  double a0 = 0.0, a1 = 0.0;
  for (int b = threadIdx.x; b < nb; b += 256) {
    a0 += part[((int64_t)b * 2 + 0) * C + c];
    a1 += part[((int64_t)b * 2 + 1) * C + c];
  }
  sh[0][threadIdx.x] = a0; sh[1][threadIdx.x] = a1;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + w];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  s0 = sh[0][0]; s1 = sh[1][0];
}

// BN forward statistics from the partials: mean, invstd (biased var), and the
// moving-average deltas into the gradient arena
__global__ void bn_finish_stats(const double *__restrict__ part, int nb, int C,
                                int64_t M, float eps, float momentum,
                                const float *__restrict__ mov_mean,
                                const float *__restrict__ mov_var,
                                float *__restrict__ mean, float *__restrict__ invstd,
                                float *__restrict__ d_mov_mean,
                                float *__restrict__ d_mov_var) {
  const int c = blockIdx.x;                     // one 256-thread block per channel
  double s0, s1;
  block_sum_partials(part, nb, C, c, s0, s1);
  if (threadIdx.x != 0) return;
  const double mu = s0 / (double)M;
  double var = s1 / (double)M - mu * mu;
  var = var > 0.0 ? var : 0.0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  d_mov_mean[c] = ((float)mu - mov_mean[c]) * (1.f - momentum);
  d_mov_var[c] = ((float)var - mov_var[c]) * (1.f - momentum);
}

__global__ void finish_sums(const double *__restrict__ part, int nb, int C,
                            float *__restrict__ s0_out, float *__restrict__ s1_out,
                            float scale) {
  const int c = blockIdx.x;                     // one 256-thread block per channel
  double s0, s1;
  block_sum_partials(part, nb, C, c, s0, s1);
  if (threadIdx.x != 0) return;
  if (s0_out) s0_out[c] += (float)(s0 * scale);
  if (s1_out) s1_out[c] += (float)(s1 * scale);
}

__global__ void bn_apply(const float *__restrict__ x, const float *__restrict__ mean,
                         const float *__restrict__ invstd,
                         const float *__restrict__ gamma,
                         const float *__restrict__ beta, float *__restrict__ y,
                         int64_t n, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  y[i] = (x[i] - mean[c]) * invstd[c] * gamma[c] + beta[c];
}

// dx += gamma*invstd/M * (M*dy - sum_dy - xhat*sum_dy_xhat)
__global__ void bn_backward(const float *__restrict__ dy, const float *__restrict__ x,
                            const float *__restrict__ mean,
                            const float *__restrict__ invstd,
                            const float *__restrict__ gamma,
                            const float *__restrict__ sum_dy,
                            const float *__restrict__ sum_dy_xhat,
                            float *__restrict__ dx, int64_t n, int C, float inv_m,
                            int acc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const float xh = (x[i] - mean[c]) * invstd[c];
  const float v = gamma[c] * invstd[c] *
                  (dy[i] - inv_m * sum_dy[c] - xh * inv_m * sum_dy_xhat[c]);
  dx[i] = acc ? dx[i] + v : v;
}

// BatchNorm + ReLU in one pass (the pair always appears together, fplmodels.py:67-71)
__global__ void bn_relu_apply(const float *__restrict__ x, const float *__restrict__ mean,
                              const float *__restrict__ invstd,
                              const float *__restrict__ gamma,
                              const float *__restrict__ beta, float *__restrict__ y,
                              int64_t n, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  y[i] = fmaxf((x[i] - mean[c]) * invstd[c] * gamma[c] + beta[c], 0.f);
}
// backward of the pair: g = dy * (y > 0), then the BatchNorm input gradient
__global__ void bn_relu_backward(const float *__restrict__ dy, const float *__restrict__ y,
                                 const float *__restrict__ x,
                                 const float *__restrict__ mean,
                                 const float *__restrict__ invstd,
                                 const float *__restrict__ gamma,
                                 const float *__restrict__ sum_g,
                                 const float *__restrict__ sum_g_xhat,
                                 float *__restrict__ dx, int64_t n, int C, float inv_m,
                                 int acc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const float gg = y[i] > 0.f ? dy[i] : 0.f;
  const float xh = (x[i] - mean[c]) * invstd[c];
  const float v = gamma[c] * invstd[c] * (gg - inv_m * sum_g[c] - xh * inv_m * sum_g_xhat[c]);
  dx[i] = acc ? dx[i] + v : v;
}

__global__ void relu_fwd(const float *__restrict__ x, float *__restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = fmaxf(x[i], 0.f);
}
__global__ void relu_bwd(const float *__restrict__ dy, const float *__restrict__ y,
                         float *__restrict__ dx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] += y[i] > 0.f ? dy[i] : 0.f;
}

__global__ void pool2_fwd(const float *__restrict__ x, float *__restrict__ y,
                          uint8_t *__restrict__ arg, int64_t n_out, int D, int H,
                          int W, int C, int od, int oh, int ow) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  int64_t t = i;
  const int c = (int)(t % C); t /= C;
  const int ox = (int)(t % ow); t /= ow;
  const int oy = (int)(t % oh); t /= oh;
  const int oz = (int)(t % od); t /= od;
  float m = -INFINITY;
  int am = 0;
  for (int p = 0; p < 8; ++p) {
    const float v = x[((((t * D + 2 * oz + (p >> 2)) * H + 2 * oy + ((p >> 1) & 1)) *
                        (int64_t)W + 2 * ox + (p & 1)) * C) + c];
    if (v > m) { m = v; am = p; }      // first maximum wins (TF MaxPoolGrad)
  }
  y[i] = m;
  arg[i] = (uint8_t)am;
}
__global__ void pool2_bwd(const float *__restrict__ dy, const uint8_t *__restrict__ arg,
                          float *__restrict__ dx, int64_t n_out, int D, int H, int W,
                          int C, int od, int oh, int ow) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  int64_t t = i;
  const int c = (int)(t % C); t /= C;
  const int ox = (int)(t % ow); t /= ow;
  const int oy = (int)(t % oh); t /= oh;
  const int oz = (int)(t % od); t /= od;
  const int p = arg[i];
  dx[((((t * D + 2 * oz + (p >> 2)) * H + 2 * oy + ((p >> 1) & 1)) * (int64_t)W +
       2 * ox + (p & 1)) * C) + c] += dy[i];      // windows do not overlap
}

// Pool gradient when the 2x2x2 windows tile the input exactly (even D, H, W) and
// C % 4 == 0: a thread owns four channels of one window and WRITES all eight positions
// (dy at the arg-max, zero elsewhere) - dx needs no zero-fill and no read-modify-write.
__global__ void pool2_bwd_assign4(const float4 *__restrict__ dy, const uint32_t *__restrict__ arg,
                                  float4 *__restrict__ dx, int64_t n4, int D, int H, int W,
                                  int C4, int od, int oh, int ow) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  int64_t t = i;
  const int c = (int)(t % C4); t /= C4;
  const int ox = (int)(t % ow); t /= ow;
  const int oy = (int)(t % oh); t /= oh;
  const int oz = (int)(t % od); t /= od;
  const float4 g = dy[i];
  const uint32_t a = arg[i];
  const int a0 = a & 255, a1 = (a >> 8) & 255, a2 = (a >> 16) & 255, a3 = a >> 24;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    float4 o;
    o.x = a0 == p ? g.x : 0.f; o.y = a1 == p ? g.y : 0.f;
    o.z = a2 == p ? g.z : 0.f; o.w = a3 == p ? g.w : 0.f;
    dx[((((t * D + 2 * oz + (p >> 2)) * H + 2 * oy + ((p >> 1) & 1)) * (int64_t)W +
         2 * ox + (p & 1)) * C4) + c] = o;
  }
}

__device__ __forceinline__ bool drop_keep(uint64_t seed, int layer, int64_t i,
                                          float rate) {
  const uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)layer << 48) ^ (uint64_t)i));
  return (float)(h >> 40) * (1.f / 16777216.f) >= rate;
}
__global__ void dropout_fwd(const float *__restrict__ x, float *__restrict__ y,
                            int64_t n, uint64_t seed, int layer, float rate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = drop_keep(seed, layer, i, rate) ? x[i] / (1.f - rate) : 0.f;
}
__global__ void dropout_bwd(const float *__restrict__ dy, float *__restrict__ dx,
                            int64_t n, uint64_t seed, int layer, float rate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] += drop_keep(seed, layer, i, rate) ? dy[i] / (1.f - rate) : 0.f;
}

// y[b][z][y][x][c_off + c] = x[b][(z+lo)/f..][c]   (crop / upsample / concat half)
__global__ void remap_fwd(const float *__restrict__ x, float *__restrict__ y,
                          int64_t n, int D, int H, int W, int C, int od, int oh,
                          int ow, int lo0, int lo1, int lo2, int f0, int f1, int f2,
                          int c_off, int c_total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t t = i;
  const int c = (int)(t % C); t /= C;
  const int ox = (int)(t % ow); t /= ow;
  const int oy = (int)(t % oh); t /= oh;
  const int oz = (int)(t % od); t /= od;
  y[((((t * od + oz) * oh + oy) * (int64_t)ow + ox) * c_total) + c_off + c] =
      x[((((t * D + (oz + lo0) / f0) * H + (oy + lo1) / f1) * (int64_t)W +
          (ox + lo2) / f2) * C) + c];
}
// transpose of remap_fwd: dx[src] += dy[dst] (atomic: upsampling maps many -> one)
__global__ void remap_bwd(const float *__restrict__ dy, float *__restrict__ dx,
                          int64_t n, int D, int H, int W, int C, int od, int oh,
                          int ow, int lo0, int lo1, int lo2, int f0, int f1, int f2,
                          int c_off, int c_total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t t = i;
  const int c = (int)(t % C); t /= C;
  const int ox = (int)(t % ow); t /= ow;
  const int oy = (int)(t % oh); t /= oh;
  const int oz = (int)(t % od); t /= od;
  const float g =
      dy[((((t * od + oz) * oh + oy) * (int64_t)ow + ox) * c_total) + c_off + c];
  float *dst = &dx[((((t * D + (oz + lo0) / f0) * H + (oy + lo1) / f1) * (int64_t)W +
                     (ox + lo2) / f2) * C) + c];
  if (f0 * f1 * f2 == 1) *dst += g; else atomicAdd(dst, g);
}
__global__ void add_fwd(const float *a, const float *b, float *y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a[i] + b[i];
}
__global__ void accum(const float *dy, float *dx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] += dy[i];
}

// Sigmoid-output losses (include/fplhip.h fpl_loss): per-element loss, metric sums
// and dL/dlogit with the 1/n of the mean folded in.  The network's last op is the
// sigmoid, so dL/dlogit = dL/dp * p (1 - p).
//   BCE (Keras K.binary_crossentropy): p clipped to [1e-7, 1-1e-7], re-expressed
//     through the logit as TF does; zero gradient where the clip is active.
//   MASKED_BCE (fplmodels.py:41-44): the same on (p * mask, y * mask), mask = y != 2;
//     a masked voxel contributes -log(1 - 1e-7) and no gradient.
//   MASKED_WEIGHTED_BCE (fplmodels.py:28-39): tf.nn.weighted_cross_entropy_with_logits
//     with pos_weight q = 100 on z = logit(clip(p * mask)):
//     (1 - t) z + (1 + (q - 1) t) (log1p(exp(-|z|)) + max(-z, 0)).
//   MASKED_FOCAL (fplmodels.py:45-50): -mask (1 - pt)^2 log(pt + 1e-7),
//     pt = p where y == 1, else 1 - p; mask = y < 2.
template <int KIND>
__global__ void loss_grad(const float *__restrict__ p, const uint8_t *__restrict__ lab,
                          float *__restrict__ dlogit, int64_t n, float inv_n,
                          double *__restrict__ sums) {
  __shared__ double sh[7][256];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double v[7] = {0, 0, 0, 0, 0, 0, 0};
  if (i < n) {
    const int yl = lab[i];
    const float y = (float)yl;
    const float pi = p[i];
    const float eps = 1e-7f;
    const float mask = KIND == FPL_LOSS_BCE ? 1.f : (yl != 2 ? 1.f : 0.f);
    float l, d;
    if (KIND == FPL_LOSS_MASKED_FOCAL) {
      const float m = yl < 2 ? 1.f : 0.f;
      const bool pos = yl == 1;
      const float pt = pos ? pi : 1.f - pi;
      const float lg = logf(pt + eps);
      l = -m * (1.f - pt) * (1.f - pt) * lg;
      // d/dpt of -(1-pt)^2 log(pt+eps); dpt/dp = +-1
      const float dpt = 2.f * (1.f - pt) * lg - (1.f - pt) * (1.f - pt) / (pt + eps);
      d = m * (pos ? dpt : -dpt) * pi * (1.f - pi);
    } else {
      const float t = y * mask, pm = pi * mask;
      const float pc = fminf(fmaxf(pm, eps), 1.f - eps);
      const float z = logf(pc / (1.f - pc));
      const bool open = pm > eps && pm < 1.f - eps;     // clip inactive
      if (KIND == FPL_LOSS_MASKED_WEIGHTED_BCE) {
        const float w = 1.f + 99.f * t;
        l = (1.f - t) * z + w * (log1pf(expf(-fabsf(z))) + fmaxf(-z, 0.f));
        d = open ? ((1.f - t) - w * (1.f - pc)) : 0.f;
      } else {
        l = fmaxf(z, 0.f) - z * t + log1pf(expf(-fabsf(z)));
        d = open ? (pc - t) : 0.f;
      }
    }
    dlogit[i] = d * inv_n;
    const float mk = yl != 2 ? 1.f : 0.f;
    v[0] = (double)l;
    v[1] = (rintf(pi) == y) ? 1.0 : 0.0;
    v[2] = (rintf(pi * mk) == y * mk) ? 1.0 : 0.0;
    v[3] = yl == 0 ? (double)pi : 0.0;
    v[4] = yl == 0 ? 1.0 : 0.0;
    v[5] = yl == 1 ? (double)(1.f - pi) : 0.0;
    v[6] = yl == 1 ? 1.0 : 0.0;
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) sh[k][threadIdx.x] = v[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int k = 0; k < 7; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x < 7) atomicAdd(&sums[threadIdx.x], sh[threadIdx.x][0]);
}

// dX[in][ci] += sum_tap sum_co dY[in - tap][co] * W[tap][ci][co]
template <int CT>
__global__ __launch_bounds__(256) void conv_dgrad_f32(
    const float *__restrict__ dy, const float *__restrict__ w, float *__restrict__ dx,
    int64_t n_vox, int D, int H, int W, int cin, int od, int oh, int ow, int cout,
    int k) {
  const int64_t vox = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vox >= n_vox) return;
  const int ci0 = blockIdx.y * CT;
  int64_t t = vox;
  const int ix = (int)(t % W); t /= W;
  const int iy = (int)(t % H); t /= H;
  const int iz = (int)(t % D); t /= D;
  float acc[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) acc[j] = 0.f;
  for (int dz = 0; dz < k; ++dz) {
    const int oz = iz - dz;
    if (oz < 0 || oz >= od) continue;
    for (int dyy = 0; dyy < k; ++dyy) {
      const int oy = iy - dyy;
      if (oy < 0 || oy >= oh) continue;
      for (int dxx = 0; dxx < k; ++dxx) {
        const int ox = ix - dxx;
        if (ox < 0 || ox >= ow) continue;
        const float *gp = dy + ((((t * od + oz) * oh + oy) * (int64_t)ow + ox) * cout);
        const float *wp = w + ((int64_t)((dz * k + dyy) * k + dxx) * cin + ci0) * cout;
        for (int co = 0; co < cout; ++co) {
          const float g = gp[co];
#pragma unroll
          for (int j = 0; j < CT; ++j)
            if (CT == 1 || ci0 + j < cin) acc[j] = fmaf(g, wp[(int64_t)j * cout + co], acc[j]);
        }
      }
    }
  }
  float *xp = dx + vox * cin + ci0;
#pragma unroll
  for (int j = 0; j < CT; ++j)
    if (ci0 + j < cin) xp[j] += acc[j];
}

// dW[tap][ci][co] += sum_m X[m+tap][ci] * dY[m][co].  Block = 16x16 threads, one
// tap, one 48x48 (ci,co) tile, one chunk of output voxels; 3x3 register tile.
constexpr int WG_VS = 16;            // voxels staged per LDS round
constexpr int WG_CHUNK = 4096;       // output voxels per block
__global__ __launch_bounds__(256) void conv_wgrad_f32(
    const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dw,
    int64_t n_vox, int D, int H, int W, int cin, int od, int oh, int ow, int cout,
    int k, int ci_tiles, int co_tiles) {
  __shared__ float xs[WG_VS][48], gs[WG_VS][48];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  int bid = blockIdx.x;
  const int cot = bid % co_tiles; bid /= co_tiles;
  const int cit = bid % ci_tiles; bid /= ci_tiles;
  const int tap = bid;
  const int dz = tap / (k * k), dyy = (tap / k) % k, dxx = tap % k;
  const int ci0 = cit * 48, co0 = cot * 48;
  const int64_t m0 = (int64_t)blockIdx.y * WG_CHUNK;
  const int64_t m1 = m0 + WG_CHUNK < n_vox ? m0 + WG_CHUNK : n_vox;
  float acc[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  for (int64_t mb = m0; mb < m1; mb += WG_VS) {
    // stage WG_VS rows: 16 x 48 values each for x and dy -> 3 per thread each
    for (int e = threadIdx.x; e < WG_VS * 48; e += 256) {
      const int v = e / 48, c = e % 48;
      const int64_t m = mb + v;
      float xv = 0.f, gv = 0.f;
      if (m < m1) {
        int64_t t = m;
        const int ox = (int)(t % ow); t /= ow;
        const int oy = (int)(t % oh); t /= oh;
        const int oz = (int)(t % od); t /= od;
        if (ci0 + c < cin)
          xv = x[((((t * D + oz + dz) * H + oy + dyy) * (int64_t)W + ox + dxx) * cin) +
                 ci0 + c];
        if (co0 + c < cout) gv = dy[m * cout + co0 + c];
      }
      xs[v][c] = xv;
      gs[v][c] = gv;
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < WG_VS; ++v) {
      float xv[3], gv[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { xv[i] = xs[v][ty + 16 * i]; gv[i] = gs[v][tx + 16 * i]; }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = fmaf(xv[i], gv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ci = ci0 + ty + 16 * i, co = co0 + tx + 16 * j;
      if (ci < cin && co < cout && acc[i][j] != 0.f)
        atomicAdd(&dw[((int64_t)tap * cin + ci) * cout + co], acc[i][j]);
    }
}

__global__ void adam_update(float *__restrict__ w, const float *__restrict__ g,
                            float *__restrict__ m, float *__restrict__ v,
                            const uint8_t *__restrict__ kind, int64_t n,
                            float grad_scale, float lr_t, float b1, float b2,
                            float eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * grad_scale;
  if (kind[i] == 1) {            // moving statistic: additive averaged delta
    w[i] += gi;
    return;
  }
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  w[i] -= lr_t * mi / (sqrtf(vi) + eps);
}

__global__ void fill_f32(float *p, float v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

struct fpl_trainer {
  fpl_ctx *ctx = nullptr;
  std::vector<fpl_layer> layers;
  int n_tensors = 0, out_tensor = 0;
  int64_t n_w = 0;
  float *w = nullptr, *g = nullptr, *m = nullptr, *v = nullptr;
  uint8_t *kind = nullptr;
  float *ones = nullptr, *zeros = nullptr;        // 256 floats each
  float lr = 1e-3f, b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
  int64_t step_count = 0;
  int loss_kind = FPL_LOSS_BCE;
  double last_sums[FPL_N_METRIC_SUMS] = {0};
};

namespace {

int shapes_for(fpl_ctx *ctx, const fpl_trainer *t, const int32_t patch[3],
               std::vector<TShape> *shp) {
  shp->assign(t->n_tensors, TShape());
  (*shp)[0] = TShape{patch[0], patch[1], patch[2], 1};
  for (size_t i = 0; i < t->layers.size(); ++i) {
    const fpl_layer &L = t->layers[i];
    const TShape a = (*shp)[L.src0];
    FPL_REQUIRE(ctx, a.c > 0, "layer %zu reads tensor %d before it is produced", i, L.src0);
    TShape o = a;
    switch (L.kind) {
      case FPL_L_CONV:
        FPL_REQUIRE(ctx, a.c == L.cin, "layer %zu: conv cin mismatch", i);
        o.d = a.d - L.k + 1; o.h = a.h - L.k + 1; o.w = a.w - L.k + 1; o.c = L.cout;
        break;
      case FPL_L_POOL:
        FPL_REQUIRE(ctx, L.p[0] == 2 && L.p[1] == 2 && L.p[2] == 2,
                    "layer %zu: only MaxPooling3D(2) is trainable", i);
        o.d = a.d / 2; o.h = a.h / 2; o.w = a.w / 2;
        break;
      case FPL_L_UP: o.d = a.d * L.p[0]; o.h = a.h * L.p[1]; o.w = a.w * L.p[2]; break;
      case FPL_L_CROP:
        o.d = a.d - L.p[0] - L.p[1]; o.h = a.h - L.p[2] - L.p[3]; o.w = a.w - L.p[4] - L.p[5];
        break;
      case FPL_L_CONCAT: {
        const TShape b = (*shp)[L.src1];
        FPL_REQUIRE(ctx, a.d == b.d && a.h == b.h && a.w == b.w,
                    "layer %zu: concatenate shape mismatch - patch size is not "
                    "compatible with this architecture", i);
        o.c = a.c + b.c;
        break;
      }
      case FPL_L_ADD: case FPL_L_BN: case FPL_L_RELU: case FPL_L_DROPOUT: break;
      default: return fpl_fail(ctx, "layer %zu: unknown kind %d", i, L.kind);
    }
    FPL_REQUIRE(ctx, o.d > 0 && o.h > 0 && o.w > 0, "layer %zu: patch too small", i);
    (*shp)[L.dst] = o;
  }
  return 0;
}

}  // namespace

extern "C" {

int fpl_trainer_create(fpl_ctx *ctx, const fpl_layer *layers, int32_t n_layers,
                       int32_t n_tensors, int32_t out_tensor, const float *weights,
                       int64_t n_weights, float lr, float beta1, float beta2,
                       float eps, fpl_trainer **out) {
  if (!ctx || !layers || !weights || !out)
    return fpl_fail(ctx, "fpl_trainer_create: NULL argument");
  *out = nullptr;
  FPL_REQUIRE(ctx, n_layers > 0 && n_tensors > 1 && out_tensor > 0 && out_tensor < n_tensors,
              "fpl_trainer_create: bad sizes");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<uint8_t> kind((size_t)n_weights, 0);
  for (int i = 0; i < n_layers; ++i) {
    const fpl_layer &L = layers[i];
    FPL_REQUIRE(ctx, L.src0 >= 0 && L.src0 < n_tensors && L.dst > 0 && L.dst < n_tensors &&
                         L.src1 < n_tensors, "fpl_trainer_create: layer %d tensor ids", i);
    if (L.kind == FPL_L_CONV) {
      FPL_REQUIRE(ctx, L.k == 1 || L.k == 3, "layer %d: conv kernel %d", i, L.k);
      FPL_REQUIRE(ctx, L.cin <= 256 && L.cout <= 256, "layer %d: > 256 channels", i);
      const int64_t kk = (int64_t)L.k * L.k * L.k * L.cin * L.cout;
      FPL_REQUIRE(ctx, L.w_off[0] >= 0 && L.w_off[0] + kk <= n_weights &&
                           (!L.use_bias || (L.w_off[1] >= 0 && L.w_off[1] + L.cout <= n_weights)),
                  "layer %d: weight offsets exceed the arena", i);
      FPL_REQUIRE(ctx, L.act == FPL_ACT_NONE || (L.act == FPL_ACT_SIGMOID && L.dst == out_tensor),
                  "layer %d: only a sigmoid head may carry a conv activation", i);
    } else if (L.kind == FPL_L_BN) {
      FPL_REQUIRE(ctx, L.cin <= 256, "layer %d: > 256 channels", i);
      for (int q = 0; q < 4; ++q)
        FPL_REQUIRE(ctx, L.w_off[q] >= 0 && L.w_off[q] + L.cin <= n_weights,
                    "layer %d: BN offsets exceed the arena", i);
      for (int q = 2; q < 4; ++q)
        for (int c = 0; c < L.cin; ++c) kind[L.w_off[q] + c] = 1;
    }
  }
  fpl_trainer *t = new fpl_trainer();
  t->ctx = ctx;
  t->layers.assign(layers, layers + n_layers);
  t->n_tensors = n_tensors;
  t->out_tensor = out_tensor;
  t->n_w = n_weights;
  t->lr = lr; t->b1 = beta1; t->b2 = beta2; t->eps = eps;
  const size_t nb = (size_t)n_weights * sizeof(float);
  bool ok = hipMalloc((void **)&t->w, nb) == hipSuccess &&
            hipMalloc((void **)&t->g, nb) == hipSuccess &&
            hipMalloc((void **)&t->m, nb) == hipSuccess &&
            hipMalloc((void **)&t->v, nb) == hipSuccess &&
            hipMalloc((void **)&t->kind, (size_t)n_weights) == hipSuccess &&
            hipMalloc((void **)&t->ones, 256 * sizeof(float)) == hipSuccess &&
            hipMalloc((void **)&t->zeros, 256 * sizeof(float)) == hipSuccess;
  if (!ok) {
    fpl_trainer_destroy(t);
    return fpl_fail(ctx, "fpl_trainer_create: device allocation failed");
  }
  hipStream_t st = ctx->stream;
  FPL_HIP(ctx, hipMemcpyAsync(t->w, weights, nb, hipMemcpyHostToDevice, st));
  FPL_HIP(ctx, hipMemcpyAsync(t->kind, kind.data(), (size_t)n_weights, hipMemcpyHostToDevice, st));
  FPL_HIP(ctx, hipMemsetAsync(t->g, 0, nb, st));
  FPL_HIP(ctx, hipMemsetAsync(t->m, 0, nb, st));
  FPL_HIP(ctx, hipMemsetAsync(t->v, 0, nb, st));
  FPL_HIP(ctx, hipMemsetAsync(t->zeros, 0, 256 * sizeof(float), st));
  fill_f32<<<1, 256, 0, st>>>(t->ones, 1.f, 256);
  FPL_HIP(ctx, hipStreamSynchronize(st));
  *out = t;
  return 0;
}

int fpl_trainer_destroy(fpl_trainer *t) {
  if (!t) return 0;
  hipSetDevice(t->ctx->device);
  hipStreamSynchronize(t->ctx->stream);
  for (void *p : {(void *)t->w, (void *)t->g, (void *)t->m, (void *)t->v, (void *)t->kind,
                  (void *)t->ones, (void *)t->zeros})
    if (p) hipFree(p);
  delete t;
  return 0;
}

int fpl_trainer_grad_ptr(fpl_trainer *t, void **dev_ptr, int64_t *n_floats) {
  if (!t || !dev_ptr || !n_floats) return fpl_fail(nullptr, "fpl_trainer_grad_ptr: NULL");
  *dev_ptr = t->g;
  *n_floats = t->n_w;
  return 0;
}

static int copy_arena(fpl_trainer *t, float *dev, float *host, int64_t n, bool to_host) {
  fpl_ctx *ctx = t->ctx;
  FPL_REQUIRE(ctx, n == t->n_w, "arena has %lld floats, trainer expects %lld",
              (long long)n, (long long)t->n_w);
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  if (to_host)
    FPL_HIP(ctx, hipMemcpyAsync(host, dev, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  else
    FPL_HIP(ctx, hipMemcpyAsync(dev, host, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}
int fpl_trainer_get_weights(fpl_trainer *t, float *out, int64_t n) {
  if (!t || !out) return fpl_fail(nullptr, "fpl_trainer_get_weights: NULL");
  return copy_arena(t, t->w, out, n, true);
}
int fpl_trainer_get_grads(fpl_trainer *t, float *out, int64_t n) {
  if (!t || !out) return fpl_fail(nullptr, "fpl_trainer_get_grads: NULL");
  return copy_arena(t, t->g, out, n, true);
}
int fpl_trainer_set_weights(fpl_trainer *t, const float *w, int64_t n) {
  if (!t || !w) return fpl_fail(nullptr, "fpl_trainer_set_weights: NULL");
  return copy_arena(t, t->w, const_cast<float *>(w), n, false);
}

// Adam state: first / second moments in the weight arena's layout (moving-statistics slots
// unused) and the number of updates applied - what Keras' model.save keeps as
// `optimizer_weights` (flypylib/fplnetwork.py:15-17,81-97 save through it)
int fpl_trainer_get_opt_state(fpl_trainer *t, float *m, float *v, int64_t n, int64_t *steps) {
  if (!t || !m || !v || !steps) return fpl_fail(nullptr, "fpl_trainer_get_opt_state: NULL");
  FPL_TRY(copy_arena(t, t->m, m, n, true));
  FPL_TRY(copy_arena(t, t->v, v, n, true));
  *steps = t->step_count;
  return 0;
}
int fpl_trainer_set_opt_state(fpl_trainer *t, const float *m, const float *v, int64_t n, int64_t steps) {
  if (!t || !m || !v) return fpl_fail(nullptr, "fpl_trainer_set_opt_state: NULL");
  if (steps < 0) return fpl_fail(t->ctx, "fpl_trainer_set_opt_state: %lld steps", (long long)steps);
  FPL_TRY(copy_arena(t, t->m, const_cast<float *>(m), n, false));
  FPL_TRY(copy_arena(t, t->v, const_cast<float *>(v), n, false));
  t->step_count = steps;
  return 0;
}

int fpl_trainer_set_grads(fpl_trainer *t, const float *g, int64_t n) {
  if (!t || !g) return fpl_fail(nullptr, "fpl_trainer_set_grads: NULL");
  return copy_arena(t, t->g, const_cast<float *>(g), n, false);
}

// one RCCL all-reduce (sum) of the flat gradient arena on the context's stream; the
// Adam kernel of fpl_trainer_apply is ordered behind it on the same stream
int fpl_allreduce_grads(fpl_trainer *t) {
  if (!t) return fpl_fail(nullptr, "fpl_allreduce_grads: NULL");
  return fpl_comm_allreduce_sum_f32(t->ctx, t->g, t->n_w);
}

// every rank starts from rank `root`'s weights and Adam state
int fpl_trainer_broadcast_state(fpl_trainer *t, int32_t root) {
  if (!t) return fpl_fail(nullptr, "fpl_trainer_broadcast_state: NULL");
  FPL_TRY(fpl_comm_broadcast_f32(t->ctx, t->w, t->n_w, root));
  FPL_TRY(fpl_comm_broadcast_f32(t->ctx, t->m, t->n_w, root));
  FPL_TRY(fpl_comm_broadcast_f32(t->ctx, t->v, t->n_w, root));
  FPL_HIP(t->ctx, hipStreamSynchronize(t->ctx->stream));
  return 0;
}

int fpl_trainer_apply(fpl_trainer *t, float grad_scale) {
  if (!t) return fpl_fail(nullptr, "fpl_trainer_apply: NULL");
  fpl_ctx *ctx = t->ctx;
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  t->step_count += 1;
  const double tt = (double)t->step_count;
  const float lr_t = (float)(t->lr * std::sqrt(1.0 - std::pow((double)t->b2, tt)) /
                             (1.0 - std::pow((double)t->b1, tt)));
  {
    TimedLaunch tl(ctx, "train_adam");
    adam_update<<<g1(t->n_w), 256, 0, ctx->stream>>>(t->w, t->g, t->m, t->v, t->kind, t->n_w,
                                                     grad_scale, lr_t, t->b1, t->b2, t->eps);
  }
  FPL_HIP(ctx, hipGetLastError());
  return 0;
}

int fpl_trainer_step(fpl_trainer *t, const float *data, int data_mem,
                     const uint8_t *labels, int labels_mem, int32_t batch,
                     const int32_t patch[3], uint64_t seed, float *loss,
                     float *accuracy) {
  if (!t || !data || !labels || !patch) return fpl_fail(nullptr, "fpl_trainer_step: NULL");
  fpl_ctx *ctx = t->ctx;
  FPL_REQUIRE(ctx, batch > 0, "fpl_trainer_step: batch %d", batch);
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // the split-half copies of the previous step's tensors (conv_mfma.hip) mirror buffers this step recycles
  struct SplitReset {
    fpl_ctx *c;
    explicit SplitReset(fpl_ctx *c_) : c(c_) { fpl_tm_split_reset(c); }
    ~SplitReset() { fpl_tm_split_reset(c); }
  } split_reset(ctx);
  // fp32 MFMA convolutions; FPL_TRAIN_DIRECT = bitmask of what falls back to the
  // direct kernels (1 forward, 2 backward; anything else = both) - a debug switch
  const char *direct_env = getenv("FPL_TRAIN_DIRECT");
  const int direct_mask = direct_env ? ((atoi(direct_env) & 3) ? (atoi(direct_env) & 3) : 3) : 0;
  const bool use_mfma = !(direct_mask & 1), use_mfma_bwd = !(direct_mask & 2);
  std::vector<TShape> shp;
  FPL_TRY(shapes_for(ctx, t, patch, &shp));
  const int nt = t->n_tensors;
  const int nl = (int)t->layers.size();
  DevTemp tmp(ctx);
  std::vector<float *> val(nt, nullptr), grad(nt, nullptr);
  std::vector<uint8_t *> arg(nl, nullptr);
  std::vector<float *> bn_mean(nl, nullptr), bn_invstd(nl, nullptr);
  std::vector<double *> conv_stats(nt, nullptr);     // per tensor: statistics partials
  std::vector<int> conv_stats_rows(nt, 0);
  auto alloc_f = [&](int64_t n, float **p) -> int {
    void *q;
    int rc = tmp.alloc((size_t)n * sizeof(float), &q);
    *p = (float *)q;
    return rc;
  };
  // input
  const int64_t in_elems = (int64_t)batch * shp[0].elems();
  if (data_mem == FPL_MEM_HOST) {
    FPL_TRY(alloc_f(in_elems, &val[0]));
    FPL_HIP(ctx, hipMemcpyAsync(val[0], data, (size_t)in_elems * 4, hipMemcpyHostToDevice, st));
  } else {
    val[0] = const_cast<float *>(data);
  }
  const TShape oshape = shp[t->out_tensor];
  FPL_REQUIRE(ctx, oshape.c == 1, "fpl_trainer_step: network output has %d channels", oshape.c);
  const int64_t n_out = (int64_t)batch * oshape.elems();
  const uint8_t *lab_dev = labels;
  if (labels_mem == FPL_MEM_HOST) {
    void *q;
    FPL_TRY(tmp.alloc((size_t)n_out, &q));
    FPL_HIP(ctx, hipMemcpyAsync(q, labels, (size_t)n_out, hipMemcpyHostToDevice, st));
    lab_dev = (const uint8_t *)q;
  }
  FPL_HIP(ctx, hipMemsetAsync(t->g, 0, (size_t)t->n_w * 4, st));
  void *partv;
  const int64_t max_rows = (int64_t)batch * shp[0].vox();
  const int max_nb = (int)std::max<int64_t>(ceil_div64(max_rows, 4096), 2048) + 1;
  FPL_TRY(tmp.alloc((size_t)max_nb * 2 * 256 * sizeof(double), &partv));
  double *part = (double *)partv;
  void *sumsv;
  FPL_TRY(tmp.alloc(8 * sizeof(double), &sumsv));
  double *sums = (double *)sumsv;
  FPL_HIP(ctx, hipMemsetAsync(sums, 0, 8 * sizeof(double), st));

  // BatchNorm directly followed by its only consumer ReLU runs as one fused pass
  std::vector<int> n_cons(nt, 0);
  for (int li = 0; li < nl; ++li) {
    n_cons[t->layers[li].src0]++;
    if (t->layers[li].src1 >= 0) n_cons[t->layers[li].src1]++;
  }
  std::vector<char> bn_fused(nl, 0), relu_fused(nl, 0);
  for (int li = 0; li + 1 < nl; ++li)
    if (t->layers[li].kind == FPL_L_BN && t->layers[li + 1].kind == FPL_L_RELU &&
        t->layers[li + 1].src0 == t->layers[li].dst && n_cons[t->layers[li].dst] == 1 &&
        t->layers[li].dst != t->out_tensor && !getenv("FPL_TRAIN_UNFUSED")) {
      bn_fused[li] = 1;
      relu_fused[li + 1] = 1;
    }
  // ... and when that ReLU's only consumer is an exactly tiling 2x2x2 pool, the three run
  // as one layer: the full-resolution ReLU output and its gradient never exist
  std::vector<char> pool_fused(nl, 0), pool_skipped(nl, 0);
  for (int li = 0; li + 2 < nl; ++li) {
    if (!bn_fused[li]) continue;
    const fpl_layer &B = t->layers[li], &R = t->layers[li + 1], &P = t->layers[li + 2];
    const TShape x = shp[B.src0];
    if (P.kind == FPL_L_POOL && P.src0 == R.dst && n_cons[R.dst] == 1 && R.dst != t->out_tensor &&
        P.p[0] == 2 && P.p[1] == 2 && P.p[2] == 2 &&
        x.d % 2 == 0 && x.h % 2 == 0 && x.w % 2 == 0 && x.c % 4 == 0 && x.c <= 256 &&
        B.w_off[0] % 4 == 0 && B.w_off[1] % 4 == 0) {
      pool_fused[li] = 1;
      pool_skipped[li + 2] = 1;
    }
  }

  // ... and when it is a 1x1x1 convolution with a kernel for it, the convolution (and
  // its weight gradient) apply BN + ReLU while loading: that tensor is never written
  // (vgg_like's first block: 1.46 GB of a 32 x 64^3 step written and read back)
  std::vector<int> bn_view(nt, -1);            // tensor -> BN layer whose relu output it is
  std::vector<double *> bn_bstat(nl, nullptr); // BN layer -> backward sums made by its consumer
  std::vector<int> bn_bstat_rows(nl, 0);
  // tensor -> its gradient is not written: the convolution that made the tensor forms it in
  // its weight-gradient loader from the BatchNorm's output gradient (FplBnGrad, fast_paths.h)
  std::vector<FplBnGrad> bn_grad(nt, FplBnGrad{});
  // ... or, behind a BatchNorm + ReLU + pool layer, the 1x1x1 convolution's weight- AND
  // input-gradient loaders form it from the pooled gradient (FplPoolGrad)
  std::vector<FplPoolGrad> pool_grad(nt, FplPoolGrad{});
  for (int li = 0; li + 2 < nl; ++li) {
    if (!bn_fused[li] || pool_fused[li] || !use_mfma || !use_mfma_bwd) continue;
    const fpl_layer &B = t->layers[li], &R = t->layers[li + 1];
    const bool v4 = shp[B.src0].c % 4 == 0 && B.w_off[0] % 4 == 0 && B.w_off[1] % 4 == 0;
    if (!v4 || n_cons[R.dst] != 1 || R.dst == t->out_tensor) continue;
    for (int lc = li + 2; lc < nl; ++lc) {
      const fpl_layer &Cv = t->layers[lc];
      if (Cv.src0 != R.dst) continue;
      if (Cv.kind == FPL_L_CONV && fpl_tm_bn_view_supported(Cv.k, Cv.cin, Cv.cout) &&
          fpl_tm_supported(Cv.k, Cv.cin, Cv.cout) && fpl_tm_bwd_supported(Cv.k, Cv.cin, Cv.cout))
        bn_view[R.dst] = li;
      break;
    }
  }
  auto view_of = [&](int tensor, FplBnView *v) -> const float * {
    const int li = bn_view[tensor];
    const fpl_layer &B = t->layers[li];
    v->mean = bn_mean[li]; v->invstd = bn_invstd[li];
    v->gamma = t->w + B.w_off[0]; v->beta = t->w + B.w_off[1];
    return val[B.src0];
  };

  // ------------------------------ forward ------------------------------------
  for (int li = 0; li < nl; ++li) {
    const fpl_layer &L = t->layers[li];
    const TShape a = shp[L.src0], o = shp[L.dst];
    const int64_t n = (int64_t)batch * o.elems();
    if (relu_fused[li] || pool_skipped[li]) continue;   // produced by the BN before it
    if (pool_fused[li]) {
      const int pdst = t->layers[li + 2].dst;
      FPL_TRY(alloc_f((int64_t)batch * shp[pdst].elems(), &val[pdst]));
    } else if (bn_fused[li]) {
      FPL_TRY(alloc_f(n, &val[t->layers[li + 1].dst]));
    } else {
      FPL_TRY(alloc_f(n, &val[L.dst]));
    }
    switch (L.kind) {
      case FPL_L_CONV: {
        const int64_t n_vox = (int64_t)batch * o.vox();
        const float *bias = L.use_bias ? t->w + L.w_off[1] : t->zeros;
        if (use_mfma && fpl_tm_supported(L.k, L.cin, L.cout)) {
          // a BatchNorm next in line gets its batch statistics from this kernel's epilogue
          double *st_part = nullptr;
          if (li + 1 < nl && t->layers[li + 1].kind == FPL_L_BN &&
              t->layers[li + 1].src0 == L.dst && !getenv("FPL_TRAIN_UNFUSED")) {
            const int64_t rows = fpl_tm_conv_stats_rows(ctx, batch, a.d, a.h, a.w, a.c, L.k, L.cout);
            if (rows > 0 && rows < (1 << 30)) {
              void *q;
              FPL_TRY(tmp.alloc((size_t)rows * 2 * L.cout * sizeof(double), &q));
              st_part = (double *)q;
              conv_stats[L.dst] = st_part;
              conv_stats_rows[L.dst] = (int)rows;
            }
          }
          FplBnView bv;
          const bool viewed = bn_view[L.src0] >= 0;
          const float *xin = viewed ? view_of(L.src0, &bv) : val[L.src0];
          FPL_TRY(fpl_tm_conv_fwd(ctx, xin, batch, a.d, a.h, a.w, a.c, L.k, L.cout,
                                  t->w + L.w_off[0], bias, L.act, val[L.dst], st_part,
                                  viewed ? &bv : nullptr));
          break;
        }
        TimedLaunch tl(ctx, "train_conv_fwd");
        if (L.cout % 16 == 0) {
          dim3 g((unsigned)ceil_div64(n_vox, 256), L.cout / 16);
          conv3d_direct_f32<16><<<g, 256, 0, st>>>(val[L.src0], t->w + L.w_off[0], t->ones, bias,
              val[L.dst], n_vox, a.d, a.h, a.w, a.c, o.d, o.h, o.w, o.c, L.k, L.act);
        } else {
          dim3 g((unsigned)ceil_div64(n_vox, 256), L.cout);
          conv3d_direct_f32<1><<<g, 256, 0, st>>>(val[L.src0], t->w + L.w_off[0], t->ones, bias,
              val[L.dst], n_vox, a.d, a.h, a.w, a.c, o.d, o.h, o.w, o.c, L.k, L.act);
        }
        break;
      }
      case FPL_L_BN: {
        const int C = a.c;
        const int64_t M = (int64_t)batch * a.vox();
        const int rr = red_rows(M), nb = (int)ceil_div64(M, rr);
        const int R = std::max(1, 256 / C);
        FPL_TRY(alloc_f(C, &bn_mean[li]));
        FPL_TRY(alloc_f(C, &bn_invstd[li]));
        TimedLaunch tl(ctx, "train_bn_fwd");
        // float4 kernels when the per-channel vectors are 16-B aligned in the arena
        const bool v4 = C % 4 == 0 && (L.w_off[0] % 4) == 0 && (L.w_off[1] % 4) == 0;
        const int R4 = v4 ? std::max(1, 256 / (C / 4)) : 0;
        const double *spart = part;
        int snb = nb;
        if (conv_stats[L.src0]) {                  // written by the producing convolution
          spart = conv_stats[L.src0];
          snb = conv_stats_rows[L.src0];
        } else if (v4)
          chan_reduce_partial4<0><<<nb, dim3(C / 4, R4), (size_t)C * R4 * 2 * sizeof(double), st>>>(
              val[L.src0], nullptr, nullptr, nullptr, M, C, rr, part);
        else
          chan_reduce_partial<0><<<nb, dim3(C, R), (size_t)C * R * 2 * sizeof(double), st>>>(
              val[L.src0], nullptr, nullptr, nullptr, M, C, rr, part);
        bn_finish_stats<<<C, 256, 0, st>>>(spart, snb, C, M, 1e-3f, 0.99f,
            t->w + L.w_off[2], t->w + L.w_off[3], bn_mean[li], bn_invstd[li],
            t->g + L.w_off[2], t->g + L.w_off[3]);
        if (pool_fused[li]) {
          typedef const float4 *cf4;
          const int pl = li + 2, pdst = t->layers[pl].dst;
          const TShape po = shp[pdst];
          const int64_t np4 = (int64_t)batch * po.elems() / 4;
          void *q;
          FPL_TRY(tmp.alloc((size_t)np4 * 4, &q));
          arg[pl] = (uint8_t *)q;
          const PoolGeo geo = {a.d, a.h, a.w, po.d, po.h, po.w};
          bn_relu_pool4<<<g1(np4), 256, 0, st>>>((cf4)val[L.src0], (cf4)bn_mean[li],
              (cf4)bn_invstd[li], (cf4)(t->w + L.w_off[0]), (cf4)(t->w + L.w_off[1]),
              (float4 *)val[pdst], (uint32_t *)arg[pl], np4, C / 4, geo);
        } else if (bn_fused[li] && bn_view[t->layers[li + 1].dst] == li) {
          // applied by the consuming convolution's loader
        } else if (v4) {
          typedef const float4 *cf4;
          if (bn_fused[li])
            bn_apply4<true><<<g1(n / 4), 256, 0, st>>>((cf4)val[L.src0], (cf4)bn_mean[li],
                (cf4)bn_invstd[li], (cf4)(t->w + L.w_off[0]), (cf4)(t->w + L.w_off[1]),
                (float4 *)val[t->layers[li + 1].dst], n / 4, C / 4);
          else
            bn_apply4<false><<<g1(n / 4), 256, 0, st>>>((cf4)val[L.src0], (cf4)bn_mean[li],
                (cf4)bn_invstd[li], (cf4)(t->w + L.w_off[0]), (cf4)(t->w + L.w_off[1]),
                (float4 *)val[L.dst], n / 4, C / 4);
        } else if (bn_fused[li])
          bn_relu_apply<<<g1(n), 256, 0, st>>>(val[L.src0], bn_mean[li], bn_invstd[li],
              t->w + L.w_off[0], t->w + L.w_off[1], val[t->layers[li + 1].dst], n, C);
        else
          bn_apply<<<g1(n), 256, 0, st>>>(val[L.src0], bn_mean[li], bn_invstd[li],
              t->w + L.w_off[0], t->w + L.w_off[1], val[L.dst], n, C);
        break;
      }
      case FPL_L_RELU: {
        TimedLaunch tl(ctx, "train_elementwise");
        relu_fwd<<<g1(n), 256, 0, st>>>(val[L.src0], val[L.dst], n);
        break;
      }
      case FPL_L_POOL: {
        void *q;
        FPL_TRY(tmp.alloc((size_t)n, &q));
        arg[li] = (uint8_t *)q;
        TimedLaunch tl(ctx, "train_pool");
        pool2_fwd<<<g1(n), 256, 0, st>>>(val[L.src0], val[L.dst], arg[li], n, a.d, a.h, a.w,
                                         a.c, o.d, o.h, o.w);
        break;
      }
      case FPL_L_DROPOUT: {
        TimedLaunch tl(ctx, "train_elementwise");
        dropout_fwd<<<g1(n), 256, 0, st>>>(val[L.src0], val[L.dst], n, seed, li, L.rate);
        break;
      }
      case FPL_L_UP: {
        TimedLaunch tl(ctx, "train_elementwise");
        remap_fwd<<<g1(n), 256, 0, st>>>(val[L.src0], val[L.dst], n, a.d, a.h, a.w, a.c, o.d,
            o.h, o.w, 0, 0, 0, L.p[0], L.p[1], L.p[2], 0, o.c);
        break;
      }
      case FPL_L_CROP: {
        TimedLaunch tl(ctx, "train_elementwise");
        remap_fwd<<<g1(n), 256, 0, st>>>(val[L.src0], val[L.dst], n, a.d, a.h, a.w, a.c, o.d,
            o.h, o.w, L.p[0], L.p[2], L.p[4], 1, 1, 1, 0, o.c);
        break;
      }
      case FPL_L_CONCAT: {
        const TShape b = shp[L.src1];
        const int64_t na = (int64_t)batch * a.elems(), nbb = (int64_t)batch * b.elems();
        TimedLaunch tl(ctx, "train_elementwise");
        remap_fwd<<<g1(na), 256, 0, st>>>(val[L.src0], val[L.dst], na, a.d, a.h, a.w, a.c, o.d,
            o.h, o.w, 0, 0, 0, 1, 1, 1, 0, o.c);
        remap_fwd<<<g1(nbb), 256, 0, st>>>(val[L.src1], val[L.dst], nbb, b.d, b.h, b.w, b.c, o.d,
            o.h, o.w, 0, 0, 0, 1, 1, 1, a.c, o.c);
        break;
      }
      case FPL_L_ADD: {
        TimedLaunch tl(ctx, "train_elementwise");
        add_fwd<<<g1(n), 256, 0, st>>>(val[L.src0], val[L.src1], val[L.dst], n);
        break;
      }
    }
    FPL_HIP(ctx, hipGetLastError());
  }

  // ------------------------------ loss ---------------------------------------
  const fpl_layer *head = nullptr;
  for (int li = 0; li < nl; ++li)
    if (t->layers[li].dst == t->out_tensor) head = &t->layers[li];
  FPL_REQUIRE(ctx, head && head->kind == FPL_L_CONV && head->act == FPL_ACT_SIGMOID,
              "fpl_trainer_step: the output must be a sigmoid conv head");
  // A tensor whose only consumer's backward pass writes every element of its gradient
  // (BatchNorm; an MFMA-path convolution) gets that gradient by assignment: no
  // zero-fill and no read-modify-write.  Everything else accumulates into zeros.
  std::vector<int> n_use(nt, 0), cons(nt, -1);
  for (int li = 0; li < nl; ++li) {
    const fpl_layer &L = t->layers[li];
    if (relu_fused[li]) continue;                 // consumes the unallocated BN output
    ++n_use[L.src0]; cons[L.src0] = li;
    if (L.src1 > 0) { ++n_use[L.src1]; cons[L.src1] = -2; }
  }
  std::vector<char> assign(nt, 0);
  auto pool_tiles = [](const TShape &x) {
    return x.d % 2 == 0 && x.h % 2 == 0 && x.w % 2 == 0 && x.c % 4 == 0;
  };
  for (int ti = 1; ti < nt; ++ti) {
    if (n_use[ti] != 1 || cons[ti] < 0) continue;
    const fpl_layer &L = t->layers[cons[ti]];
    if (L.kind == FPL_L_BN) assign[ti] = 1;
    if (L.kind == FPL_L_POOL && pool_tiles(shp[ti])) assign[ti] = 1;
    if (L.kind == FPL_L_CONV && use_mfma_bwd && fpl_tm_bwd_supported(L.k, L.cin, L.cout))
      assign[ti] = 1;
  }
  for (int ti = 1; ti < nt; ++ti) {
    if (!val[ti]) continue;
    const int64_t n = (int64_t)batch * shp[ti].elems();
    FPL_TRY(alloc_f(n, &grad[ti]));
    if (!assign[ti]) FPL_HIP(ctx, hipMemsetAsync(grad[ti], 0, (size_t)n * 4, st));
  }
  {
    TimedLaunch tl(ctx, "train_loss");
    const float inv_n = 1.f / (float)n_out;
    float *po = val[t->out_tensor], *go = grad[t->out_tensor];
    switch (t->loss_kind) {
      case FPL_LOSS_MASKED_BCE:
        loss_grad<FPL_LOSS_MASKED_BCE><<<g1(n_out), 256, 0, st>>>(po, lab_dev, go, n_out, inv_n, sums);
        break;
      case FPL_LOSS_MASKED_WEIGHTED_BCE:
        loss_grad<FPL_LOSS_MASKED_WEIGHTED_BCE><<<g1(n_out), 256, 0, st>>>(po, lab_dev, go, n_out, inv_n, sums);
        break;
      case FPL_LOSS_MASKED_FOCAL:
        loss_grad<FPL_LOSS_MASKED_FOCAL><<<g1(n_out), 256, 0, st>>>(po, lab_dev, go, n_out, inv_n, sums);
        break;
      default:
        loss_grad<FPL_LOSS_BCE><<<g1(n_out), 256, 0, st>>>(po, lab_dev, go, n_out, inv_n, sums);
    }
  }

  // ------------------------------ backward -----------------------------------
  for (int li = nl - 1; li >= 0; --li) {
    const fpl_layer &L = t->layers[li];
    const TShape a = shp[L.src0], o = shp[L.dst];
    const int64_t n = (int64_t)batch * o.elems();
    if (relu_fused[li] || pool_skipped[li]) continue;   // handled with the BN before it
    float *dy = pool_fused[li] ? grad[t->layers[li + 2].dst]
                : bn_fused[li] ? grad[t->layers[li + 1].dst] : grad[L.dst];
    float *dx = L.src0 > 0 ? grad[L.src0] : nullptr;
    switch (L.kind) {
      case FPL_L_CONV: {
        const int64_t n_vox = (int64_t)batch * o.vox();
        const int taps = L.k * L.k * L.k;
        if (use_mfma_bwd && fpl_tm_bwd_supported(L.k, L.cin, L.cout)) {
          FplBnView bv;
          const bool viewed = bn_view[L.src0] >= 0;
          const float *xin = viewed ? view_of(L.src0, &bv) : val[L.src0];
          const FplBnGrad *bgv = bn_grad[L.dst].g ? &bn_grad[L.dst] : nullptr;
          const FplPoolGrad *pgv = pool_grad[L.dst].x ? &pool_grad[L.dst] : nullptr;
          FPL_TRY(fpl_tm_conv_wgrad(ctx, xin, batch, a.d, a.h, a.w, a.c, dy, L.k,
                                    L.cout, t->g + L.w_off[0], viewed ? &bv : nullptr, bgv, pgv));
          if (L.use_bias) {
            const int rr = red_rows(n_vox), nb = (int)ceil_div64(n_vox, rr);
            const int R = std::max(1, 256 / L.cout);
            TimedLaunch tl(ctx, "train_bias_grad");
            chan_reduce_partial<2><<<nb, dim3(L.cout, R), (size_t)L.cout * R * 2 * sizeof(double), st>>>(
                dy, nullptr, nullptr, nullptr, n_vox, L.cout, rr, part);
            finish_sums<<<L.cout, 256, 0, st>>>(part, nb, L.cout, t->g + L.w_off[1],
                                                           nullptr, 1.f);
          }
          if (dx && assign[L.src0] && viewed && !getenv("FPL_TRAIN_BNSTAT_SEPARATE")) {
            // ... and the input-gradient kernel's epilogue makes that BatchNorm's backward
            // sums while the gradient is in registers (the separate pass read it back)
            const int bl = bn_view[L.src0];
            const int64_t rows = fpl_tm_conv_stats_rows(ctx, batch, a.d, a.h, a.w, a.c, 1, L.cin);
            void *q;
            FPL_TRY(tmp.alloc((size_t)rows * 2 * L.cin * sizeof(double), &q));
            FplBnStat bs;
            bs.x = xin; bs.bn = bv; bs.part = (double *)q;
            bn_bstat[bl] = (double *)q;
            bn_bstat_rows[bl] = (int)rows;
            FPL_TRY(fpl_tm_conv_dgrad(ctx, dy, batch, o.d, o.h, o.w, o.c, L.k, L.cin,
                                      t->w + L.w_off[0], t->zeros, dx, &bs, pgv));
          } else if (dx && assign[L.src0]) {
            FPL_TRY(fpl_tm_conv_dgrad(ctx, dy, batch, o.d, o.h, o.w, o.c, L.k, L.cin,
                                      t->w + L.w_off[0], t->zeros, dx));
          } else if (dx) {
            const int64_t in_el = (int64_t)batch * a.elems();
            float *tmpdx;
            FPL_TRY(alloc_f(in_el, &tmpdx));
            FPL_TRY(fpl_tm_conv_dgrad(ctx, dy, batch, o.d, o.h, o.w, o.c, L.k, L.cin,
                                      t->w + L.w_off[0], t->zeros, tmpdx));
            TimedLaunch tl(ctx, "train_elementwise");
            accum<<<g1(in_el), 256, 0, st>>>(tmpdx, dx, in_el);
            tmp.release(tmpdx);
          }
          break;
        }
        {
          const int cit = (L.cin + 47) / 48, cot = (L.cout + 47) / 48;
          dim3 g((unsigned)(taps * cit * cot), (unsigned)ceil_div64(n_vox, WG_CHUNK));
          TimedLaunch tl(ctx, "train_conv_wgrad");
          conv_wgrad_f32<<<g, 256, 0, st>>>(val[L.src0], dy, t->g + L.w_off[0], n_vox, a.d, a.h,
              a.w, a.c, o.d, o.h, o.w, o.c, L.k, cit, cot);
        }
        if (L.use_bias) {
          const int rr = red_rows(n_vox), nb = (int)ceil_div64(n_vox, rr);
          const int R = std::max(1, 256 / L.cout);
          TimedLaunch tl(ctx, "train_bias_grad");
          chan_reduce_partial<2><<<nb, dim3(L.cout, R), (size_t)L.cout * R * 2 * sizeof(double), st>>>(
              dy, nullptr, nullptr, nullptr, n_vox, L.cout, rr, part);
          finish_sums<<<L.cout, 256, 0, st>>>(part, nb, L.cout, t->g + L.w_off[1],
                                                         nullptr, 1.f);
        }
        if (dx) {
          const int64_t in_vox = (int64_t)batch * a.vox();
          TimedLaunch tl(ctx, "train_conv_dgrad");
          if (L.cin % 16 == 0) {
            dim3 g((unsigned)ceil_div64(in_vox, 256), L.cin / 16);
            conv_dgrad_f32<16><<<g, 256, 0, st>>>(dy, t->w + L.w_off[0], dx, in_vox, a.d, a.h,
                a.w, a.c, o.d, o.h, o.w, o.c, L.k);
          } else {
            dim3 g((unsigned)ceil_div64(in_vox, 256), L.cin);
            conv_dgrad_f32<1><<<g, 256, 0, st>>>(dy, t->w + L.w_off[0], dx, in_vox, a.d, a.h,
                a.w, a.c, o.d, o.h, o.w, o.c, L.k);
          }
        }
        break;
      }
      case FPL_L_BN: {
        const int C = a.c;
        const int64_t M = (int64_t)batch * a.vox();
        const int rr = red_rows(M), nb = (int)ceil_div64(M, rr);
        const int R = std::max(1, 256 / C);
        float *sdy, *sdyx;
        FPL_TRY(alloc_f(C, &sdy));
        FPL_TRY(alloc_f(C, &sdyx));
        FPL_HIP(ctx, hipMemsetAsync(sdy, 0, (size_t)C * 4, st));
        FPL_HIP(ctx, hipMemsetAsync(sdyx, 0, (size_t)C * 4, st));
        TimedLaunch tl(ctx, "train_bn_bwd");
        const float *yrelu = bn_fused[li] ? val[t->layers[li + 1].dst] : nullptr;
        const bool v4 = C % 4 == 0 && (L.w_off[0] % 4) == 0 && (L.w_off[1] % 4) == 0;
        const int R4 = v4 ? std::max(1, 256 / (C / 4)) : 0;
        if (pool_fused[li]) {
          // dy is the POOLED gradient; rows = pooling windows
          typedef const float4 *cf4;
          const int pl = li + 2;
          const TShape po = shp[t->layers[pl].dst];
          const int64_t Mp = (int64_t)batch * po.vox();
          const int rp = red_rows(Mp), nbp = (int)ceil_div64(Mp, rp);
          const PoolGeo geo = {a.d, a.h, a.w, po.d, po.h, po.w};
          chan_reduce_pool4<<<nbp, dim3(C / 4, R4), (size_t)C * R4 * 2 * sizeof(double), st>>>(
              (cf4)dy, (const uint32_t *)arg[pl], (cf4)val[L.src0], bn_mean[li], bn_invstd[li],
              t->w + L.w_off[0], t->w + L.w_off[1], Mp, C, rp, part, geo);
          finish_sums<<<C, 256, 0, st>>>(part, nbp, C, sdy, sdyx, 1.f);
          accum<<<1, 256, 0, st>>>(sdy, t->g + L.w_off[1], C);
          accum<<<1, 256, 0, st>>>(sdyx, t->g + L.w_off[0], C);
          // The tensor's only reader is the 1x1x1 convolution that made it, whose weight- and
          // input-gradient kernels can form dx from the pooled gradient themselves (the input
          // gradient in its BatchNorm-statistics form, i.e. behind a viewed BN): this pass's
          // write of the layer's full-resolution tensor and one read of it go
          {
            int prod = -1;
            for (int lp = 0; lp < li; ++lp)
              if (t->layers[lp].dst == L.src0) prod = lp;
            if (dx && assign[L.src0] && prod >= 0 && t->layers[prod].kind == FPL_L_CONV &&
                !t->layers[prod].use_bias && use_mfma_bwd && bn_view[t->layers[prod].src0] >= 0 &&
                assign[t->layers[prod].src0] && grad[t->layers[prod].src0] &&
                fpl_tm_bwd_supported(t->layers[prod].k, t->layers[prod].cin, t->layers[prod].cout) &&
                fpl_tm_pool_grad_supported(t->layers[prod].k, t->layers[prod].cin, t->layers[prod].cout) &&
                !getenv("FPL_TRAIN_BNSTAT_SEPARATE") && !getenv("FPL_TRAIN_POOLGRAD_SEPARATE") &&
                (int64_t)batch * a.vox() < ((int64_t)1 << 31)) {
              FplPoolGrad &pgv = pool_grad[L.src0];
              pgv.dyp = dy; pgv.arg = (const uint32_t *)arg[pl]; pgv.x = val[L.src0];
              pgv.bn.mean = bn_mean[li]; pgv.bn.invstd = bn_invstd[li];
              pgv.bn.gamma = t->w + L.w_off[0]; pgv.bn.beta = t->w + L.w_off[1];
              pgv.sum_g = sdy; pgv.sum_gx = sdyx; pgv.inv_m = 1.f / (float)M;
              pgv.D = a.d; pgv.H = a.h; pgv.W = a.w;
              break;
            }
          }
          if (dx) {
            const int64_t np4 = Mp * (C / 4);
#define FPL_BNP4(ACC)                                                                         \
  bn_backward_pool4<ACC><<<g1(np4), 256, 0, st>>>((cf4)dy, (const uint32_t *)arg[pl],          \
      (cf4)val[L.src0], (cf4)bn_mean[li], (cf4)bn_invstd[li], (cf4)(t->w + L.w_off[0]),        \
      (cf4)(t->w + L.w_off[1]), (cf4)sdy, (cf4)sdyx, (float4 *)dx, np4, C / 4, 1.f / (float)M, geo)
            if (assign[L.src0]) FPL_BNP4(false); else FPL_BNP4(true);
#undef FPL_BNP4
          }
          break;
        }
        const double *bpart = part;
        int bnb = nb;
        if (bn_bstat[li]) {                      // made by the consumer's input-gradient kernel
          bpart = bn_bstat[li];
          bnb = bn_bstat_rows[li];
        } else if (v4 && bn_fused[li])
          chan_reduce_partial4<3><<<nb, dim3(C / 4, R4), (size_t)C * R4 * 2 * sizeof(double), st>>>(
              dy, val[L.src0], bn_mean[li], bn_invstd[li], M, C, rr, part, t->w + L.w_off[0],
              t->w + L.w_off[1]);
        else if (v4)
          chan_reduce_partial4<1><<<nb, dim3(C / 4, R4), (size_t)C * R4 * 2 * sizeof(double), st>>>(
              dy, val[L.src0], bn_mean[li], bn_invstd[li], M, C, rr, part);
        else if (bn_fused[li])
          chan_reduce_partial<3><<<nb, dim3(C, R), (size_t)C * R * 2 * sizeof(double), st>>>(
              dy, val[L.src0], bn_mean[li], bn_invstd[li], M, C, rr, part, yrelu);
        else
          chan_reduce_partial<1><<<nb, dim3(C, R), (size_t)C * R * 2 * sizeof(double), st>>>(
              dy, val[L.src0], bn_mean[li], bn_invstd[li], M, C, rr, part);
        finish_sums<<<C, 256, 0, st>>>(bpart, bnb, C, sdy, sdyx, 1.f);
        // dbeta = sum dy, dgamma = sum dy*xhat
        accum<<<1, 256, 0, st>>>(sdy, t->g + L.w_off[1], C);
        accum<<<1, 256, 0, st>>>(sdyx, t->g + L.w_off[0], C);
        const int acc = dx && !assign[L.src0];
        // The tensor's only reader is the weight gradient of the first convolution (cin = 1:
        // no input gradient): that kernel makes dx from dy and x itself - this pass's write
        // and that kernel's read of the step's largest tensor (1.46 GB each at 32 x 64^3) go
        int prod = -1;
        for (int lp = 0; lp < li; ++lp)
          if (t->layers[lp].dst == L.src0) prod = lp;
        if (dx && v4 && bn_fused[li] && !acc && prod >= 0 && t->layers[prod].kind == FPL_L_CONV &&
            t->layers[prod].src0 == 0 && !t->layers[prod].use_bias && use_mfma_bwd &&
            fpl_tm_bwd_supported(t->layers[prod].k, t->layers[prod].cin, t->layers[prod].cout) &&
            fpl_tm_bn_grad_supported(t->layers[prod].k, t->layers[prod].cin, t->layers[prod].cout) &&
            !getenv("FPL_TRAIN_BNGRAD_SEPARATE")) {
          FplBnGrad &bgv = bn_grad[L.src0];
          bgv.g = dy; bgv.x = val[L.src0];
          bgv.bn.mean = bn_mean[li]; bgv.bn.invstd = bn_invstd[li];
          bgv.bn.gamma = t->w + L.w_off[0]; bgv.bn.beta = t->w + L.w_off[1];
          bgv.sum_g = sdy; bgv.sum_gx = sdyx; bgv.inv_m = 1.f / (float)M;
          break;
        }
        if (dx && v4) {
          typedef const float4 *cf4;
#define FPL_BNB4(RELU, ACC, YP)                                                            \
  bn_backward4<RELU, ACC><<<g1(n / 4), 256, 0, st>>>((cf4)dy, (cf4)(YP), (cf4)val[L.src0],    \
      (cf4)bn_mean[li], (cf4)bn_invstd[li], (cf4)(t->w + L.w_off[0]), (cf4)sdy, (cf4)sdyx,     \
      (float4 *)dx, n / 4, C / 4, 1.f / (float)M)
          if (bn_fused[li]) { if (acc) FPL_BNB4(true, true, t->w + L.w_off[1]); else FPL_BNB4(true, false, t->w + L.w_off[1]); }
          else { if (acc) FPL_BNB4(false, true, nullptr); else FPL_BNB4(false, false, nullptr); }
#undef FPL_BNB4
        } else if (dx && bn_fused[li])
          bn_relu_backward<<<g1(n), 256, 0, st>>>(dy, yrelu, val[L.src0], bn_mean[li],
              bn_invstd[li], t->w + L.w_off[0], sdy, sdyx, dx, n, C, 1.f / (float)M, acc);
        else if (dx)
          bn_backward<<<g1(n), 256, 0, st>>>(dy, val[L.src0], bn_mean[li], bn_invstd[li],
              t->w + L.w_off[0], sdy, sdyx, dx, n, C, 1.f / (float)M, acc);
        break;
      }
      case FPL_L_RELU:
        if (dx) {
          TimedLaunch tl(ctx, "train_elementwise");
          relu_bwd<<<g1(n), 256, 0, st>>>(dy, val[L.dst], dx, n);
        }
        break;
      case FPL_L_POOL:
        if (dx) {
          TimedLaunch tl(ctx, "train_pool");
          if (assign[L.src0])
            pool2_bwd_assign4<<<g1(n / 4), 256, 0, st>>>((const float4 *)dy, (const uint32_t *)arg[li],
                (float4 *)dx, n / 4, a.d, a.h, a.w, a.c / 4, o.d, o.h, o.w);
          else
            pool2_bwd<<<g1(n), 256, 0, st>>>(dy, arg[li], dx, n, a.d, a.h, a.w, a.c, o.d, o.h, o.w);
        }
        break;
      case FPL_L_DROPOUT:
        if (dx) {
          TimedLaunch tl(ctx, "train_elementwise");
          dropout_bwd<<<g1(n), 256, 0, st>>>(dy, dx, n, seed, li, L.rate);
        }
        break;
      case FPL_L_UP:
        if (dx) {
          TimedLaunch tl(ctx, "train_elementwise");
          remap_bwd<<<g1(n), 256, 0, st>>>(dy, dx, n, a.d, a.h, a.w, a.c, o.d, o.h, o.w, 0, 0, 0,
              L.p[0], L.p[1], L.p[2], 0, o.c);
        }
        break;
      case FPL_L_CROP:
        if (dx) {
          TimedLaunch tl(ctx, "train_elementwise");
          remap_bwd<<<g1(n), 256, 0, st>>>(dy, dx, n, a.d, a.h, a.w, a.c, o.d, o.h, o.w, L.p[0],
              L.p[2], L.p[4], 1, 1, 1, 0, o.c);
        }
        break;
      case FPL_L_CONCAT: {
        const TShape b = shp[L.src1];
        const int64_t na = (int64_t)batch * a.elems(), nbb = (int64_t)batch * b.elems();
        TimedLaunch tl(ctx, "train_elementwise");
        if (dx)
          remap_bwd<<<g1(na), 256, 0, st>>>(dy, dx, na, a.d, a.h, a.w, a.c, o.d, o.h, o.w, 0, 0,
              0, 1, 1, 1, 0, o.c);
        if (L.src1 > 0)
          remap_bwd<<<g1(nbb), 256, 0, st>>>(dy, grad[L.src1], nbb, b.d, b.h, b.w, b.c, o.d, o.h,
              o.w, 0, 0, 0, 1, 1, 1, a.c, o.c);
        break;
      }
      case FPL_L_ADD: {
        TimedLaunch tl(ctx, "train_elementwise");
        if (dx) accum<<<g1(n), 256, 0, st>>>(dy, dx, n);
        if (L.src1 > 0) accum<<<g1(n), 256, 0, st>>>(dy, grad[L.src1], n);
        break;
      }
    }
    FPL_HIP(ctx, hipGetLastError());
  }
  if (const char *dump = getenv("FPL_TRAIN_DUMP")) {
    // debug: raw fp32 dump of every activation / activation-gradient tensor
    FPL_HIP(ctx, hipStreamSynchronize(st));
    for (int ti = 1; ti < nt; ++ti) {
      for (int which = 0; which < 2; ++which) {
        const float *src = which ? grad[ti] : val[ti];
        if (!src) continue;
        const int64_t n = (int64_t)batch * shp[ti].elems();
        std::vector<float> h((size_t)n);
        FPL_HIP(ctx, hipMemcpy(h.data(), src, (size_t)n * 4, hipMemcpyDeviceToHost));
        char path[512];
        snprintf(path, sizeof(path), "%s/%s_%03d_c%d.f32", dump, which ? "grad" : "val", ti, shp[ti].c);
        if (FILE *f = fopen(path, "wb")) { fwrite(h.data(), 4, (size_t)n, f); fclose(f); }
      }
    }
  }
  double hs[8];
  FPL_HIP(ctx, hipMemcpyAsync(hs, sums, sizeof(hs), hipMemcpyDeviceToHost, st));
  FPL_HIP(ctx, hipStreamSynchronize(st));
  hs[7] = (double)n_out;
  memcpy(t->last_sums, hs, sizeof(hs));
  if (loss) *loss = (float)(hs[0] / (double)n_out);
  if (accuracy) *accuracy = (float)(hs[1] / (double)n_out);
  return 0;
}

int fpl_trainer_set_loss(fpl_trainer *t, int loss_kind) {
  if (!t) return fpl_fail(nullptr, "fpl_trainer_set_loss: trainer is NULL");
  FPL_REQUIRE(t->ctx, loss_kind >= FPL_LOSS_BCE && loss_kind <= FPL_LOSS_MASKED_FOCAL,
              "fpl_trainer_set_loss: unknown loss %d", loss_kind);
  t->loss_kind = loss_kind;
  return 0;
}

int fpl_trainer_metric_sums(fpl_trainer *t, double out[FPL_N_METRIC_SUMS]) {
  if (!t || !out) return fpl_fail(nullptr, "fpl_trainer_metric_sums: NULL argument");
  memcpy(out, t->last_sums, sizeof(t->last_sums));
  return 0;
}

}  // extern "C"
