// fp32 MFMA (v_mfma_f32_16x16x4_f32: exact f32, k-ordered fmaf chain) kernels and a
// generic executor over lowered layer programs.  This is the fast form of the
// PARITY path (1e-3 gate of the north star): same arithmetic type as the
// reference's fp32 Keras graph, ~30x the direct per-op kernels of generic.hip.
//
// Fragment maps (lane l: c = l & 15, g = l >> 4): A[row c][k = g], B[k = g][col c],
// D[row 4g + r][col c].  Four consecutive K-steps form a K-block of 16 k-values;
// k-slot (j, g) of a block is bound to k = 4g + j, so a lane's operands for the
// whole block are ONE 16-byte read (4 consecutive channels of a tap / 4 floats of
// a fragment).  conv3_f32 mirrors conv3_bf16 (conv_mfma.hip): LDS tile of 16
// channels x (6 x 6 x 18) voxels at 96 B pitch; the weight fragments (MB x 1 KiB per
// tap, the same for every wave) go from L2 straight into each wave's registers, three
// taps ahead of their use - no LDS ring and no barrier inside the tap loop (round 3; the
// ring cost a barrier every three taps); two workgroups per CU.
#include <algorithm>

#include "fast_paths.h"
#include "mfma_util.h"

namespace {

constexpr int PITCH = 96;
constexpr int TZ = 6, TY = 6, TX = 18;
constexpr int TILE_BYTES = TZ * TY * TX * PITCH;
constexpr int CC = 16;                    // channels per LDS tile chunk
constexpr int KB = 27;                    // K-blocks (taps) per channel chunk
constexpr int WQF = 3;                    // taps of weight fragments in flight per wave
static_assert(KB % WQF == 0, "the fragment queue's phase is static inside a channel chunk");

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct SrcF {
  const float *p;        // (n, D, H, W, C) f32
  int D, H, W, C;
  int ch0, up, crop;
  int pad;               // zero padding around the source (dgrad: k-1)
};

struct Conv3F {
  SrcF src[12];
  int ncc;
  const float *w;        // fragments [cc][tap][mb][lane][4]
  const float *shift;
  int act;               // fpl_act
  float *out;            // (n, OD, OH, OW, cout)
  int cout;              // real output channels (<= 16*MB)
  int opitch;            // channel pitch of `out` (>= cout; a slice of a wider tensor)
  int OD, OH, OW, zblocks;
  // fused epilogue (inference): act -> 1x1x1 conv `w1` [q][mb1][lane][4] + shift1, act1
  // -> optional MaxPooling3D(2); `out` / `cout` / `opitch` then describe THAT result
  // (pooled: (n, OD/2, OH/2, OW/2, cout))
  const float *w1;
  const float *shift1;
  int act1, cout1;
};

__device__ __forceinline__ float act_f(float v, int act);

// The accumulators of one fp32 layer are the B operands of the next: lane (c, g) holds
// rows 4g..4g+3 of every 16-row block, and k-slot (j, g) of K-block q is channel
// 16q + 4g + j - register j of block q, no lane movement (the fp32 form of the 16-bit
// kernels' register chaining).  acc1[sub][m] += W1 frag (q, m) x act(acc[sub][q]).
template <int MB, int MB1>
struct Chain1F {
  f32x4 w[MB][MB1];        // W1 fragments (q, m), this lane's 4 k-slots
  f32x4 sh[MB1];           // shift1 of the lane's rows
  __device__ __forceinline__ void load(const float *w1, const float *shift1, int cout1, int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int q = 0; q < MB; ++q)
#pragma unroll
      for (int m = 0; m < MB1; ++m)
        w[q][m] = *reinterpret_cast<const f32x4 *>(w1 + ((size_t)(q * MB1 + m) * 64 + lane) * 4);
#pragma unroll
    for (int m = 0; m < MB1; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 16 * m + 4 * g + r;
        sh[m][r] = co < cout1 ? shift1[co] : 0.f;
      }
  }
  template <int NSUB>
  __device__ __forceinline__ void apply(const f32x4 (&acc)[NSUB][MB], int act,
                                        f32x4 (&acc1)[NSUB][MB1]) const {
#pragma unroll
    for (int m = 0; m < MB1; ++m)
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) acc1[sub][m] = sh[m];
#pragma unroll
    for (int q = 0; q < MB; ++q) {
      f32x4 bq[NSUB];
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) bq[sub][r] = act_f(acc[sub][q][r], act);
#pragma unroll
      for (int m = 0; m < MB1; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int sub = 0; sub < NSUB; ++sub)
            acc1[sub][m] = mfma4(w[q][m][j], bq[sub][j], acc1[sub][m]);
    }
  }
};

__device__ __forceinline__ float act_f(float v, int act) {
  if (act == FPL_ACT_RELU) return fmaxf(v, 0.f);
  if (act == FPL_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}

// store the four consecutive output channels a lane holds for m-block b: one 16-B
// store when the channel count allows it (a wave then writes whole 64-B pieces
// instead of 256 scattered dwords)
__device__ __forceinline__ void store_quad(float *dst, int co0, int cout, const f32x4 &v, int act) {
  if ((cout & 3) == 0) {
    if (co0 < cout) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = act_f(v[r], act);
      *reinterpret_cast<f32x4 *>(dst + co0) = o;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (co0 + r < cout) dst[co0 + r] = act_f(v[r], act);
  }
}

// Per-channel (sum, sum of squares) of a kernel's OUTPUT, for the BatchNorm that follows
// in training: every lane accumulates its own voxels in fp64, the 16 voxel lanes of a
// channel are summed once at the end of the kernel, the four waves meet in LDS and the
// workgroup writes one row of partials part[block][2][C] (the layout of train.hip's
// channel reductions, finished there by bn_finish_stats).
template <int MB>
struct ChanStats {
  double s0[MB][4], s1[MB][4];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s0[b][r] = 0.0; s1[b][r] = 0.0; }
  }
  __device__ __forceinline__ void add(int b, const f32x4 &v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { s0[b][r] += v[r]; s1[b][r] += (double)v[r] * v[r]; }
  }
  // red: 4 * 2 * 16 * MB doubles of LDS
  __device__ __forceinline__ void finish(double *red, double *part, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double a0 = s0[b][r], a1 = s1[b][r];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
          a0 += __shfl_xor(a0, off);
          a1 += __shfl_xor(a1, off);
        }
        if (c == 0) {
          red[(wave * 2 + 0) * 16 * MB + 16 * b + 4 * g + r] = a0;
          red[(wave * 2 + 1) * 16 * MB + 16 * b + 4 * g + r] = a1;
        }
      }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < C) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        part[((int64_t)blockIdx.x * 2 + k) * C + t] =
            (red[(0 * 2 + k) * 16 * MB + t] + red[(1 * 2 + k) * 16 * MB + t]) +
            (red[(2 * 2 + k) * 16 * MB + t] + red[(3 * 2 + k) * 16 * MB + t]);
    }
  }
};
__device__ __forceinline__ f32x4 act4(const f32x4 &v, int act) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = act_f(v[r], act);
  return o;
}

template <int MB, int MB1 = 0, bool POOL = false>
__global__ __launch_bounds__(256, 2) void conv3_f32(Conv3F a) {
  constexpr int PIECES = TZ * TY * TX * 4;          // 4 x 16 B per voxel (16 ch)
  constexpr int NT = (PIECES + 255) / 256;
  unsigned char *tile = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int x0 = blockIdx.x * 16, y0 = blockIdx.y * 4;
  const int n = blockIdx.z / a.zblocks, z0 = (blockIdx.z % a.zblocks) * 4;
  const unsigned vbase = (unsigned)(((wave * TY) * TX + c) * PITCH + g * 16);
  f32x4 acc[4][MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) {
    f32x4 sh;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = 16 * b + 4 * g + r;
      sh[r] = co < a.cout ? a.shift[co] : 0.f;
    }
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) acc[sub][b] = sh;
  }
  const int64_t total_steps = (int64_t)a.ncc * KB;
  const unsigned char *wl = reinterpret_cast<const unsigned char *>(a.w) + lane * 16;
  f32x4 wq[WQF][MB];
#pragma unroll
  for (int d = 0; d < WQF; ++d)
#pragma unroll
    for (int b = 0; b < MB; ++b)
      wq[d][b] = *reinterpret_cast<const f32x4 *>(wl + (size_t)(d * MB + b) * 1024);

  for (int cc = 0; cc < a.ncc; ++cc) {
    const SrcF s = a.src[cc];
    __syncthreads();
    constexpr int NB = 8;
#pragma unroll 1
    for (int j0 = 0; j0 < NT; j0 += NB) {
      u32x4 nt[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        int p = tid + 256 * (j0 + j);
        p = p < PIECES ? p : PIECES - 1;
        const int vox = p >> 2, pc = p & 3;
        int z = z0 + vox / (TY * TX) - s.pad, y = y0 + (vox / TX) % TY - s.pad,
            x = x0 + vox % TX - s.pad;
        const bool inside = z >= 0 && y >= 0 && x >= 0;
        z = (z + s.crop) / s.up; y = (y + s.crop) / s.up; x = (x + s.crop) / s.up;
        const bool ok = inside && z < s.D && y < s.H && x < s.W;
        z = ok ? z : 0; y = ok ? y : 0; x = ok ? x : 0;
        const int ch = s.ch0 + pc * 4;
        u32x4 v = {0u, 0u, 0u, 0u};
        const float *gp = s.p + ((((int64_t)n * s.D + z) * s.H + y) * s.W + x) * s.C + ch;
        if (ok) {
          if (ch + 3 < s.C && (s.C & 3) == 0) {
            v = *reinterpret_cast<const u32x4 *>(gp);
          } else {                               // ragged channel tail (C = 1, ...)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (ch + q < s.C) v[q] = __float_as_uint(gp[q]);
          }
        }
        nt[j] = v;
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        int p = tid + 256 * (j0 + j);
        p = p < PIECES ? p : PIECES - 1;
        *reinterpret_cast<u32x4 *>(tile + (size_t)(p >> 2) * PITCH + (p & 3) * 16) = nt[j];
      }
    }
    __syncthreads();
    const int64_t gs0 = (int64_t)cc * KB;
    // a z plane past the output extent (ragged last block): nothing to multiply - the
    // SIMD goes to the CU's other workgroup
    if (z0 + wave >= a.OD) continue;
#pragma unroll
    for (int tap = 0; tap < KB; ++tap) {
      const unsigned toff = (unsigned)((((tap / 9) * TY + (tap / 3) % 3) * TX + tap % 3) * PITCH);
      f32x4 bf[4];
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
        bf[sub] = *reinterpret_cast<const f32x4 *>(tile + vbase + toff + sub * TX * PITCH);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
          for (int b = 0; b < MB; ++b)
            acc[sub][b] = mfma4(wq[tap % WQF][b][j], bf[sub][j], acc[sub][b]);
      __builtin_amdgcn_s_setprio(0);
      {
        int64_t nxt = gs0 + tap + WQF;
        nxt = nxt < total_steps ? nxt : 0;             // past the end: a harmless reload
#pragma unroll
        for (int b = 0; b < MB; ++b)
          wq[tap % WQF][b] = *reinterpret_cast<const f32x4 *>(wl + ((size_t)nxt * MB + b) * 1024);
      }
    }
  }
  const int oz = z0 + wave, ox = x0 + c;
  if (MB1 == 0) {
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int oy = y0 + sub;
      if (oz < a.OD && oy < a.OH && ox < a.OW) {
        float *dst = a.out + ((((int64_t)n * a.OD + oz) * a.OH + oy) * a.OW + ox) * a.opitch;
#pragma unroll
        for (int b = 0; b < MB; ++b)
          store_quad(dst, 16 * b + 4 * g, a.cout, acc[sub][b], a.act);
      }
    }
    return;
  }
  // fused epilogue: the 1x1x1 conv that follows, chained in registers
  constexpr int M1 = MB1 > 0 ? MB1 : 1;
  f32x4 acc1[4][M1];
  {
    Chain1F<MB, M1> ch;
    ch.load(a.w1, a.shift1, a.cout1, lane);
    ch.template apply<4>(acc, a.act, acc1);
  }
  if (!POOL) {
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int oy = y0 + sub;
      if (oz < a.OD && oy < a.OH && ox < a.OW) {
        float *dst = a.out + ((((int64_t)n * a.OD + oz) * a.OH + oy) * a.OW + ox) * a.opitch;
#pragma unroll
        for (int m = 0; m < M1; ++m) store_quad(dst, 16 * m + 4 * g, a.cout1, acc1[sub][m], a.act1);
      }
    }
    return;
  }
  // ... and MaxPooling3D(2) of the 4 x 4 x 16 block: y pairs are sub-steps of a lane, x
  // pairs neighbouring lanes, z pairs neighbouring waves (through the idle tile LDS).
  // act1 (ReLU or none) is monotone, so it is applied once, to the maximum.
  f32x4 pm[2][M1];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int m = 0; m < M1; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = fmaxf(acc1[2 * s2][m][r], acc1[2 * s2 + 1][m][r]);
        v = fmaxf(v, __shfl_xor(v, 1));
        pm[s2][m][r] = v;
      }
  __syncthreads();                              // every wave is done with the tile
  float *xch = reinterpret_cast<float *>(smem);   // [2 odd waves][2][M1][64 lanes][4]
  if (wave & 1) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int m = 0; m < M1; ++m)
        *reinterpret_cast<f32x4 *>(xch + ((((wave >> 1) * 2 + s2) * M1 + m) * 64 + lane) * 4) = pm[s2][m];
  }
  __syncthreads();
  if ((wave & 1) == 0 && (c & 1) == 0) {
    const int pz = (z0 + wave) / 2, px = (x0 + c) / 2;
    const int PD = a.OD / 2, PH = a.OH / 2, PW = a.OW / 2;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int py = y0 / 2 + s2;
      if (pz < PD && py < PH && px < PW) {
        float *dst = a.out + ((((int64_t)n * PD + pz) * PH + py) * PW + px) * a.opitch;
#pragma unroll
        for (int m = 0; m < M1; ++m) {
          const f32x4 other = *reinterpret_cast<const f32x4 *>(
              xch + ((((wave >> 1) * 2 + s2) * M1 + m) * 64 + lane) * 4);
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(pm[s2][m][r], other[r]);
          store_quad(dst, 16 * m + 4 * g, a.cout1, v, a.act1);
        }
      }
    }
  }
}

// ---- 1x1x1 conv as a voxel GEMM (fp32) ----------------------------------------------
// ---- the pooled-gradient view (FplPoolGrad, fast_paths.h) on the device ----------------
// unsigned division by a run-time constant: q = (t + ((n - t) >> s1)) >> s2, t = umulhi(n, m)
struct FastDiv { unsigned m, s1, s2, d; };
static FastDiv fast_div(unsigned d) {
  FastDiv f;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  f.s1 = l < 1 ? l : 1; f.s2 = l > 1 ? l - 1 : 0; f.d = d;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv &f) {
  const unsigned t = __umulhi(n, f.m);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}
struct PoolGradDev {
  const float *dyp; const uint32_t *arg; const float *x;
  const float *mean, *invstd, *gamma, *beta, *sum_g, *sum_gx;
  float inv_m;
  FastDiv dW, dH, dD;
  int oh, ow, od;
};
static PoolGradDev pool_grad_dev(const FplPoolGrad &g) {
  PoolGradDev p;
  p.dyp = g.dyp; p.arg = g.arg; p.x = g.x;
  p.mean = g.bn.mean; p.invstd = g.bn.invstd; p.gamma = g.bn.gamma; p.beta = g.bn.beta;
  p.sum_g = g.sum_g; p.sum_gx = g.sum_gx; p.inv_m = g.inv_m;
  p.dW = fast_div((unsigned)g.W); p.dH = fast_div((unsigned)g.H); p.dD = fast_div((unsigned)g.D);
  p.od = g.D / 2; p.oh = g.H / 2; p.ow = g.W / 2;
  return p;
}
// voxel m of (n, D, H, W) -> its pooling window (row of dyp / arg) and its position 0..7 in it
struct VoxPos { unsigned x, y, z, t; };
__device__ __forceinline__ VoxPos pool_coords(const PoolGradDev &p, unsigned m) {
  VoxPos v;
  const unsigned r1 = fdiv(m, p.dW);
  v.x = m - r1 * p.dW.d;
  const unsigned r2 = fdiv(r1, p.dH);
  v.y = r1 - r2 * p.dH.d;
  v.t = fdiv(r2, p.dD);
  v.z = r2 - v.t * p.dD.d;
  return v;
}
// ... of the voxel j < W places further along x (row, plane and volume wrap)
__device__ __forceinline__ VoxPos pool_coords_step(const PoolGradDev &p, VoxPos v, unsigned j) {
  v.x += j;
  if (v.x >= p.dW.d) {
    v.x -= p.dW.d; v.y += 1;
    if (v.y >= p.dH.d) {
      v.y = 0; v.z += 1;
      if (v.z >= p.dD.d) { v.z = 0; v.t += 1; }
    }
  }
  return v;
}
__device__ __forceinline__ void pool_window(const PoolGradDev &p, const VoxPos &v, unsigned &win, unsigned &pos) {
  win = ((v.t * p.od + (v.z >> 1)) * p.oh + (v.y >> 1)) * p.ow + (v.x >> 1);
  pos = ((v.z & 1) << 2) | ((v.y & 1) << 1) | (v.x & 1);
}
__device__ __forceinline__ void pool_locate(const PoolGradDev &p, unsigned m, unsigned &win, unsigned &pos) {
  pool_window(p, pool_coords(p, m), win, pos);
}
// one value of train.hip::bn_backward_pool4 (ACC = false): ga * is * (g' - s0 / M - xhat * s1 / M),
// g' = d where this voxel is the window's arg-max and bn(x) > 0 (the forward pass's own rounding
// sequence for the mask).  The per-channel factors are folded once per kernel: A = ga * is,
// K0 = s0 / M, K1 = s1 / M (the elementwise pass multiplies them out per value; the results
// agree to a rounding or two, tests/test_gpu_train.py holds both against each other).
struct PoolGradK { float mean, is, ga, be, A, K0, K1; };
__device__ __forceinline__ PoolGradK pool_grad_k(float mean, float is, float ga, float be, float s0, float s1,
                                                 float inv_m) {
  return PoolGradK{mean, is, ga, be, ga * is, inv_m * s0, inv_m * s1};
}
__device__ __forceinline__ float pool_grad_value(float d, bool is_max, float xq, const PoolGradK &k) {
  const float xh = __fmul_rn(__fsub_rn(xq, k.mean), k.is);
  const bool hit = is_max && __fmaf_rn(xh, k.ga, k.be) > 0.f;
  const float gq = hit ? d : 0.f;
  return k.A * ((gq - k.K0) - xh * k.K1);
}

struct Conv1F {
  const float *in; int64_t M; int cin;     // cin padded to 16 in the fragments
  const float *w;                          // fragments [kblock][mb][lane][4]
  const float *shift;
  int act;
  float *out; int cout;
  double *stats;                           // STATS: part[grid][2][cout]
  FplBnView bn = {nullptr, nullptr, nullptr, nullptr};   // conv1_f32_wreg<.., BN>: in -> relu(bn(in))
  const float *bsx = nullptr;    // conv1_f32_wreg<.., BSTAT>: BN input at the OUTPUT positions; stats = BN backward sums
  PoolGradDev pg = {};           // conv1_f32_wreg<.., PG>: `in` is not read, the input rows are made from this view
};

// train.hip's bn_affine: the forward value and every recomputation of the ReLU mask in the
// backward passes use this one rounding sequence
__device__ __forceinline__ float bn_relu_f(float v, float m, float s, float g, float b) {
  return fmaxf(__fmaf_rn(__fmul_rn(__fsub_rn(v, m), s), g, b), 0.f);
}

template <int MB, bool STATS>
__global__ __launch_bounds__(256) void conv1_f32(Conv1F a) {
  __shared__ double red[STATS ? 4 * 2 * 16 * MB : 1];
  ChanStats<STATS ? MB : 1> cs;
  if (STATS) cs.clear();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int nkb = (a.cin + 15) / 16;
  f32x4 sh[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = 16 * b + 4 * g + r;
      sh[b][r] = co < a.cout ? a.shift[co] : 0.f;
    }
  const int64_t groups = (a.M + 15) / 16;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < groups; grp += (int64_t)gridDim.x * 4) {
    int64_t m = grp * 16 + c;
    const bool ok = m < a.M;
    m = ok ? m : a.M - 1;
    f32x4 acc[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[b] = sh[b];
    for (int kb = 0; kb < nkb; ++kb) {
      const int ch = 16 * kb + 4 * g;
      f32x4 bf = {0.f, 0.f, 0.f, 0.f};
      const float *gp = a.in + m * a.cin + ch;
      if (ch + 3 < a.cin && (a.cin & 3) == 0) {
        bf = *reinterpret_cast<const f32x4 *>(gp);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (ch + q < a.cin) bf[q] = gp[q];
      }
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        const f32x4 wf = *reinterpret_cast<const f32x4 *>(a.w + ((int64_t)(kb * MB + b) * 64 + lane) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[b] = mfma4(wf[j], bf[j], acc[b]);
      }
    }
    if (ok) {
      float *dst = a.out + m * a.cout;
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        const f32x4 o = act4(acc[b], a.act);
        store_quad(dst, 16 * b + 4 * g, a.cout, o, FPL_ACT_NONE);
        if (STATS) cs.add(b, o);
      }
    }
  }
  if (STATS) cs.finish(red, a.stats, a.cout);
}

// The same GEMM when cin is exactly 16 * NKB and the fragments fit the register file: the
// weights are loaded once per wave instead of once per 16 voxels (the generic kernel pulls
// 3 x its input bytes through the vector-memory path as weight fragments), and the next
// group's voxels are in flight while this one is multiplied.  Same group order, same K
// order: bit-identical outputs and statistics.
// BSTAT (an input-gradient launch): the outputs are the gradient of relu(bn(x)); `stats`
// then receives the BN backward sums (sum g, sum g * xhat; g = output where bn(x) > 0)
// instead of the outputs' moments - x is read at the output positions (cout = 16 * NKB)
template <int MB, int NKB, bool STATS, bool BN = false, bool BSTAT = false, bool PG = false>
__global__ __launch_bounds__(256, BSTAT ? 2 : 1) void conv1_f32_wreg(Conv1F a) {
  static_assert(!BSTAT || (STATS && BN && MB == NKB), "BSTAT: statistics of a BN view, cout = cin");
  static_assert(!PG || BSTAT, "PG: the input-gradient form");
  // PG: the six per-channel vectors of the pooled layer's BatchNorm, as prm below
  __shared__ f32x4 prg[PG ? 7 * NKB * 4 : 1];       // rows: mean, invstd, gamma, beta, A, K0, K1 (PoolGradK)
  if (PG) {
    for (int ch = threadIdx.x; ch < NKB * 16; ch += 256) {
      const PoolGradK k = pool_grad_k(a.pg.mean[ch], a.pg.invstd[ch], a.pg.gamma[ch], a.pg.beta[ch],
                                      a.pg.sum_g[ch], a.pg.sum_gx[ch], a.pg.inv_m);
      float *t = reinterpret_cast<float *>(prg);
      t[0 * NKB * 16 + ch] = k.mean; t[1 * NKB * 16 + ch] = k.is; t[2 * NKB * 16 + ch] = k.ga;
      t[3 * NKB * 16 + ch] = k.be; t[4 * NKB * 16 + ch] = k.A; t[5 * NKB * 16 + ch] = k.K0;
      t[6 * NKB * 16 + ch] = k.K1;
    }
  }
  __shared__ double red[STATS ? 4 * 2 * 16 * MB : 1];
  // BN: the four per-channel vectors, read back as 16-B pieces of a lane's four channels
  __shared__ f32x4 prm[BN ? 4 * NKB * 4 : 1];
  if (BN) {
    const float *src[4] = {a.bn.mean, a.bn.invstd, a.bn.gamma, a.bn.beta};
    for (int i = threadIdx.x; i < 4 * NKB * 16; i += 256)
      reinterpret_cast<float *>(prm)[i] = src[i / (NKB * 16)][i % (NKB * 16)];
    __syncthreads();
  }
  ChanStats<STATS ? MB : 1> cs;
  if (STATS) cs.clear();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  f32x4 sh[MB], wr[NKB][MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = 16 * b + 4 * g + r;
      sh[b][r] = co < a.cout ? a.shift[co] : 0.f;
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
      wr[kb][b] = *reinterpret_cast<const f32x4 *>(a.w + ((int64_t)(kb * MB + b) * 64 + lane) * 4);
  }
  const int64_t groups = (a.M + 15) / 16, stride = (int64_t)gridDim.x * 4;
  int64_t grp = (int64_t)blockIdx.x * 4 + wave;
  f32x4 nx[NKB];
  // PG: the raw operands of the next group's rows (pooled gradient, arg-max word, BN input)
  f32x4 nd[PG ? NKB : 1];
  unsigned na[PG ? NKB : 1], npos = 0u;
  auto fetch = [&](int64_t fg) {
    const int64_t m = min(fg * 16 + c, a.M - 1);
    if (PG) {
      unsigned win;
      pool_locate(a.pg, (unsigned)m, win, npos);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        nx[kb] = *reinterpret_cast<const f32x4 *>(a.pg.x + m * (16 * NKB) + 16 * kb + 4 * g);
        nd[kb] = *reinterpret_cast<const f32x4 *>(a.pg.dyp + (int64_t)win * (16 * NKB) + 16 * kb + 4 * g);
        na[kb] = a.pg.arg[(int64_t)win * (4 * NKB) + 4 * kb + g];
      }
      return;
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
      nx[kb] = *reinterpret_cast<const f32x4 *>(a.in + m * (16 * NKB) + 16 * kb + 4 * g);
  };
  if (grp < groups) fetch(grp);
  for (; grp < groups; grp += stride) {
    const int64_t m = grp * 16 + c;
    const bool ok = m < a.M;
    f32x4 bf[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) bf[kb] = nx[kb];
    if (PG) {
      // (re-read per group: hoisted out of the loop the 21 table vectors cost 84 registers and a
      // wave per SIMD)
      int zero = 0;
      asm volatile("" : "+s"(zero));
      const f32x4 *prq = prg + zero;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const f32x4 pm = prq[(0 * NKB + kb) * 4 + g], ps = prq[(1 * NKB + kb) * 4 + g],
                    pg = prq[(2 * NKB + kb) * 4 + g], pb = prq[(3 * NKB + kb) * 4 + g],
                    pA = prq[(4 * NKB + kb) * 4 + g], p0 = prq[(5 * NKB + kb) * 4 + g],
                    p1 = prq[(6 * NKB + kb) * 4 + g];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          bf[kb][q] = pool_grad_value(nd[kb][q], ((na[kb] >> (8 * q)) & 255u) == npos, bf[kb][q],
                                      PoolGradK{pm[q], ps[q], pg[q], pb[q], pA[q], p0[q], p1[q]});
      }
    }
    if (BN && !BSTAT) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const f32x4 pm = prm[(0 * NKB + kb) * 4 + g], ps = prm[(1 * NKB + kb) * 4 + g],
                    pg = prm[(2 * NKB + kb) * 4 + g], pb = prm[(3 * NKB + kb) * 4 + g];
#pragma unroll
        for (int q = 0; q < 4; ++q) bf[kb][q] = bn_relu_f(bf[kb][q], pm[q], ps[q], pg[q], pb[q]);
      }
    }
    if (grp + stride < groups) fetch(grp + stride);
    f32x4 acc[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[b] = PG ? f32x4{0.f, 0.f, 0.f, 0.f} : sh[b];   // (an input gradient has no shift)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[b] = mfma4(wr[kb][b][j], bf[kb][j], acc[b]);
    if (ok) {
      float *dst = a.out + m * a.cout;
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        const f32x4 o = act4(acc[b], a.act);
        store_quad(dst, 16 * b + 4 * g, a.cout, o, FPL_ACT_NONE);
        if (BSTAT) {
          const f32x4 xv = *reinterpret_cast<const f32x4 *>(a.bsx + m * a.cout + 16 * b + 4 * g);
          int zero = 0;                                     // (as above: keep the table in LDS)
          asm volatile("" : "+s"(zero));
          const f32x4 *prn = prm + zero;
          const f32x4 pm = prn[(0 * NKB + b) * 4 + g], ps = prn[(1 * NKB + b) * 4 + g],
                      pg = prn[(2 * NKB + b) * 4 + g], pb = prn[(3 * NKB + b) * 4 + g];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gq = bn_relu_f(xv[r], pm[r], ps[r], pg[r], pb[r]) > 0.f ? o[r] : 0.f;
            const float xh = (xv[r] - pm[r]) * ps[r];
            cs.s0[b][r] += gq;
            cs.s1[b][r] += (double)gq * xh;
          }
        } else if (STATS) {
          cs.add(b, o);
        }
      }
    }
  }
  if (STATS) cs.finish(red, a.stats, a.cout);
}

// ---- conv3 1 -> cout (fp32): 27 taps = 2 K-blocks (k-slot (j,g) of block q = tap
// 16q + 4g + j), gathered straight from an f32 LDS tile ------------------------------
constexpr int ST_Z = 4, ST_Y = 8, ST_X = 64;
constexpr int ST_TZ = ST_Z + 2, ST_TY = ST_Y + 2, ST_TX = ST_X + 2;

struct StemF {
  const float *in; int D, H, W;
  const float *w;                // fragments [2][mb][lane][4]
  const float *shift;
  int act;
  float *out; int cout; int OD, OH, OW, zblocks;
  double *stats;                 // STATS: part[blocks][2][cout], blocks in launch order
  // fused form (stem_conv1_pool_f32): act -> 1x1x1 conv w1 + shift1, act1 -> pool(2);
  // `out` is then the pooled (n, OD/2, OH/2, OW/2, cout1) tensor
  const float *w1;
  const float *shift1;
  int act1, cout1;
};

template <int MB, bool STATS>
__global__ __launch_bounds__(256) void stem_cin1_f32(StemF a) {
  __shared__ double red[STATS ? 4 * 2 * 16 * MB : 1];
  ChanStats<STATS ? MB : 1> cs;
  if (STATS) cs.clear();
  __shared__ float tile[ST_TZ * ST_TY * ST_TX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int x0 = blockIdx.x * ST_X, y0 = blockIdx.y * ST_Y;
  const int n = blockIdx.z / a.zblocks, z0 = (blockIdx.z % a.zblocks) * ST_Z;
  {
    // all of a thread's tile loads in flight before the first LDS store (the rolled loop ran
    // load -> wait -> store 16 times in a row)
    constexpr int NLD = (ST_TZ * ST_TY * ST_TX + 255) / 256;
    float tv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + 256 * k;
      const int tx = i % ST_TX, ty = (i / ST_TX) % ST_TY, tz = i / (ST_TX * ST_TY);
      const int z = z0 + tz, y = y0 + ty, x = x0 + tx;
      float v = 0.f;
      if (i < ST_TZ * ST_TY * ST_TX && z < a.D && y < a.H && x < a.W)
        v = a.in[(((int64_t)n * a.D + z) * a.H + y) * a.W + x];
      tv[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      if (tid + 256 * k < ST_TZ * ST_TY * ST_TX) tile[tid + 256 * k] = tv[k];
  }
  int toff[8];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = 16 * q + 4 * g + j;
      toff[4 * q + j] = t < 27 ? ((t / 9) * ST_TY + (t / 3) % 3) * ST_TX + t % 3 : 0;
    }
  f32x4 w[2][MB], sh[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) {
    w[0][b] = *reinterpret_cast<const f32x4 *>(a.w + ((0 * MB + b) * 64 + lane) * 4);
    w[1][b] = *reinterpret_cast<const f32x4 *>(a.w + ((1 * MB + b) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = 16 * b + 4 * g + r;
      sh[b][r] = co < a.cout ? a.shift[co] : 0.f;
    }
  }
  __syncthreads();
  for (int task = wave; task < ST_Z * ST_Y * (ST_X / 16); task += 4) {
    const int xg = task % (ST_X / 16), yl = (task / (ST_X / 16)) % ST_Y, zl = task / (ST_X / 16 * ST_Y);
    const int base = (zl * ST_TY + yl) * ST_TX + 16 * xg + c;
    float bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = tile[base + toff[j]];
    const int oz = z0 + zl, oy = y0 + yl, ox = x0 + 16 * xg + c;
    const bool ok = oz < a.OD && oy < a.OH && ox < a.OW;
    float *dst = a.out + ((((int64_t)n * a.OD + oz) * a.OH + oy) * a.OW + ox) * a.cout;
    f32x4 accb[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) accb[b] = sh[b];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int b = 0; b < MB; ++b) accb[b] = mfma4(w[q][b][j], bv[4 * q + j], accb[b]);   // MB independent chains
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      if (ok) {
        const f32x4 o = act4(accb[b], a.act);
        store_quad(dst, 16 * b + 4 * g, a.cout, o, FPL_ACT_NONE);
        if (STATS) cs.add(b, o);
      }
    }
  }
  if (STATS) {
    // partials row = linear block index
    double *part = a.stats;
    const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    __syncthreads();
    // finish() indexes rows by blockIdx.x: shift the base instead
    cs.finish(red, part + (blk - blockIdx.x) * 2 * a.cout, a.cout);
  }
}

// conv3 1 -> cout, act, 1x1x1 conv, act1 and MaxPooling3D(2) in one kernel (the first
// block of baseline / vgg-style / U-Net models): a wave task is 16 pre-pool x of one
// POOLED (z, y) row, its four sub-steps walk the (dz, dy) window positions, so the pool
// is an element-wise max of accumulators plus one lane exchange for the x pair.  The
// full-resolution tensors (48 x 4 B per voxel, twice) never reach HBM.
template <int MB, int MB1>
__global__ __launch_bounds__(256) void stem_conv1_pool_f32(StemF a) {
  __shared__ float tile[ST_TZ * ST_TY * ST_TX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int x0 = blockIdx.x * ST_X, y0 = blockIdx.y * ST_Y;
  const int n = blockIdx.z / a.zblocks, z0 = (blockIdx.z % a.zblocks) * ST_Z;
  {
    // all of a thread's tile loads in flight before the first LDS store (the rolled loop ran
    // load -> wait -> store 16 times in a row)
    constexpr int NLD = (ST_TZ * ST_TY * ST_TX + 255) / 256;
    float tv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + 256 * k;
      const int tx = i % ST_TX, ty = (i / ST_TX) % ST_TY, tz = i / (ST_TX * ST_TY);
      const int z = z0 + tz, y = y0 + ty, x = x0 + tx;
      float v = 0.f;
      if (i < ST_TZ * ST_TY * ST_TX && z < a.D && y < a.H && x < a.W)
        v = a.in[(((int64_t)n * a.D + z) * a.H + y) * a.W + x];
      tv[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      if (tid + 256 * k < ST_TZ * ST_TY * ST_TX) tile[tid + 256 * k] = tv[k];
  }
  int toff[8];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = 16 * q + 4 * g + j;
      toff[4 * q + j] = t < 27 ? ((t / 9) * ST_TY + (t / 3) % 3) * ST_TX + t % 3 : 0;
    }
  f32x4 w[2][MB], sh[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) {
    w[0][b] = *reinterpret_cast<const f32x4 *>(a.w + ((0 * MB + b) * 64 + lane) * 4);
    w[1][b] = *reinterpret_cast<const f32x4 *>(a.w + ((1 * MB + b) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = 16 * b + 4 * g + r;
      sh[b][r] = co < a.cout ? a.shift[co] : 0.f;
    }
  }
  // the chained conv's fragments stay in registers for the whole kernel (loaded inside
  // the task loop they could not be hoisted past the output stores)
  Chain1F<MB, MB1> ch;
  ch.load(a.w1, a.shift1, a.cout1, lane);
  __syncthreads();
  const int PD = a.OD / 2, PH = a.OH / 2, PW = a.OW / 2;
  for (int task = wave; task < (ST_Z / 2) * (ST_Y / 2) * (ST_X / 16); task += 4) {
    const int xg = task % (ST_X / 16), pyl = (task / (ST_X / 16)) % (ST_Y / 2),
              pzl = task / (ST_X / 16 * (ST_Y / 2));
    // the four window positions side by side: 4 * MB independent accumulators, so no MFMA
    // waits for the one before it; every accumulator still sums in the same order.  (Measured
    // against one position at a time - MB chains of 8 dependent MFMAs: the same 119 ms at
    // 1024^3, so dependent issue was not what holds this kernel at half the fp32 MFMA rate.)
    f32x4 pm[MB1];
    float bv[4][8];
    f32x4 acc[4][MB];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int zl = 2 * pzl + (sub >> 1), yl = 2 * pyl + (sub & 1);
      const int base = (zl * ST_TY + yl) * ST_TX + 16 * xg + c;
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[sub][j] = tile[base + toff[j]];
#pragma unroll
      for (int b = 0; b < MB; ++b) acc[sub][b] = sh[b];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
          for (int b = 0; b < MB; ++b)
            acc[sub][b] = mfma4(w[q][b][j], bv[sub][4 * q + j], acc[sub][b]);
    f32x4 acc1[4][MB1];
    ch.template apply<4>(acc, a.act, acc1);
#pragma unroll
    for (int m = 0; m < MB1; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pm[m][r] = fmaxf(fmaxf(acc1[0][m][r], acc1[1][m][r]), fmaxf(acc1[2][m][r], acc1[3][m][r]));
#pragma unroll
    for (int m = 0; m < MB1; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) pm[m][r] = fmaxf(pm[m][r], __shfl_xor(pm[m][r], 1));
    const int pz = z0 / 2 + pzl, py = y0 / 2 + pyl, px = (x0 + 16 * xg + c) / 2;
    if ((c & 1) == 0 && pz < PD && py < PH && px < PW) {
      float *dst = a.out + ((((int64_t)n * PD + pz) * PH + py) * PW + px) * a.cout1;
#pragma unroll
      for (int m = 0; m < MB1; ++m) store_quad(dst, 16 * m + 4 * g, a.cout1, pm[m], a.act1);
    }
  }
}

// out = act(a + b), either operand seen through a centre crop of a larger cubic tile
// (resnet_like's shortcuts, flypylib/fplmodels.py:174-208); channels-last, C % 4 == 0 not
// required
__global__ void add_view_f32(const float *__restrict__ a, int aD, int acrop,
                             const float *__restrict__ b, int bD, int bcrop,
                             float *__restrict__ out, int64_t n_out, int od, int C, int act) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  int64_t t = i;
  const int c = (int)(t % C); t /= C;
  const int x = (int)(t % od); t /= od;
  const int y = (int)(t % od); t /= od;
  const int z = (int)(t % od); t /= od;
  const float va = a[((((t * aD + z + acrop) * aD + y + acrop) * (int64_t)aD + x + acrop) * C) + c];
  const float vb = b[((((t * bD + z + bcrop) * bD + y + bcrop) * (int64_t)bD + x + bcrop) * C) + c];
  out[i] = act_f(va + vb, act);
}

__global__ void pool2_f32v(const float *__restrict__ x, float *__restrict__ y, int64_t n_out,
                           int D, int H, int W, int C, int od, int oh, int ow) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  int64_t t = i;
  const int c = (int)(t % C); t /= C;
  const int ox = (int)(t % ow); t /= ow;
  const int oy = (int)(t % oh); t /= oh;
  const int oz = (int)(t % od); t /= od;
  float m = -INFINITY;
#pragma unroll
  for (int p = 0; p < 8; ++p)
    m = fmaxf(m, x[((((t * D + 2 * oz + (p >> 2)) * H + 2 * oy + ((p >> 1) & 1)) * (int64_t)W +
                     2 * ox + (p & 1)) * C) + c]);
  y[i] = m;
}

// ---- host: fragment packing + generic executor --------------------------------------
// conv3 fragments: [cc][tap][mb][lane][j] = W[tap][16cc + 4g + j][16mb + (lane&15)] * scale
// output channels [co0, co0 + 16 * mb) of a conv with `cout` channels (slices of 64 for
// wider convs)
void pack_conv3_f32(const float *W, const float *scale, int cin, int cout, int mb,
                    std::vector<float> *out, int co0 = 0) {
  const int ncc = (cin + 15) / 16;
  out->assign((size_t)ncc * 27 * mb * 256, 0.f);
  for (int cc = 0; cc < ncc; ++cc)
    for (int tap = 0; tap < 27; ++tap)
      for (int b = 0; b < mb; ++b)
        for (int lane = 0; lane < 64; ++lane) {
          const int co = co0 + 16 * b + (lane & 15), g = lane >> 4;
          if (co >= cout) continue;
          for (int j = 0; j < 4; ++j) {
            const int ci = 16 * cc + 4 * g + j;
            if (ci >= cin) continue;
            (*out)[((((size_t)cc * 27 + tap) * mb + b) * 64 + lane) * 4 + j] =
                W[((size_t)tap * cin + ci) * cout + co] * scale[co];
          }
        }
}

void pack_conv1_f32(const float *W, const float *scale, int cin, int cout, int mb,
                    std::vector<float> *out) {
  const int nkb = (cin + 15) / 16;
  out->assign((size_t)nkb * mb * 256, 0.f);
  for (int kb = 0; kb < nkb; ++kb)
    for (int b = 0; b < mb; ++b)
      for (int lane = 0; lane < 64; ++lane) {
        const int co = 16 * b + (lane & 15), g = lane >> 4;
        if (co >= cout) continue;
        for (int j = 0; j < 4; ++j) {
          const int ci = 16 * kb + 4 * g + j;
          if (ci >= cin) continue;
          (*out)[(((size_t)kb * mb + b) * 64 + lane) * 4 + j] = W[(size_t)ci * cout + co] * scale[co];
        }
      }
}

// stem fragments [q][mb][lane][j] = W[tap = 16q + 4g + j][0][16mb + (lane&15)] * scale
void pack_stem_f32(const float *W, const float *scale, int cout, int mb, std::vector<float> *out) {
  out->assign((size_t)2 * mb * 256, 0.f);
  for (int q = 0; q < 2; ++q)
    for (int b = 0; b < mb; ++b)
      for (int lane = 0; lane < 64; ++lane) {
        const int co = 16 * b + (lane & 15), g = lane >> 4;
        if (co >= cout) continue;
        for (int j = 0; j < 4; ++j) {
          const int tap = 16 * q + 4 * g + j;
          if (tap < 27)
            (*out)[(((size_t)q * mb + b) * 64 + lane) * 4 + j] = W[(size_t)tap * cout + co] * scale[co];
        }
      }
}

template <int MB>
int launch_stem(fpl_ctx *ctx, StemF &a, int n) {
  a.zblocks = (int)ceil_div64(a.OD, ST_Z);
  dim3 grid((unsigned)ceil_div64(a.OW, ST_X), (unsigned)ceil_div64(a.OH, ST_Y), (unsigned)(n * a.zblocks));
  TimedLaunch tl(ctx, "mfma_stem_f32");
  if (a.stats) stem_cin1_f32<MB, true><<<grid, 256, 0, ctx->stream>>>(a);
  else stem_cin1_f32<MB, false><<<grid, 256, 0, ctx->stream>>>(a);
  return 0;
}

struct F32State {
  uint64_t version = ~0ull;
  float *frags = nullptr;
  std::vector<size_t> off;       // per op (float offset), conv ops only
};

void f32_state_free(fpl_ctx *, void *p) {
  F32State *s = (F32State *)p;
  if (s->frags) hipFree(s->frags);
  delete s;
}

// virtual tensor view: a real f32 buffer seen through up / crop
struct View {
  const float *p = nullptr;
  int D = 0, C = 0;              // real buffer dims (cubic tiles) and channels
  int up = 1, crop = 0;
  int dim = 0;                   // logical (viewed) edge
};

template <int MB>
int launch3(fpl_ctx *ctx, Conv3F &a, int n) {
  constexpr int SMEM = TILE_BYTES;
  // function attributes belong to the current device: one flag per device (a process may
  // drive several GPUs, one context each; setting it twice is harmless)
  static bool attr_set[FPL_MAX_DEVICES] = {false};
  if (!attr_set[ctx->device % FPL_MAX_DEVICES]) {
    FPL_HIP(ctx, hipFuncSetAttribute((const void *)conv3_f32<MB>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_set[ctx->device % FPL_MAX_DEVICES] = true;
  }
  a.zblocks = (int)ceil_div64(a.OD, 4);
  dim3 grid((unsigned)ceil_div64(a.OW, 16), (unsigned)ceil_div64(a.OH, 4), (unsigned)(n * a.zblocks));
  TimedLaunch tl(ctx, "mfma_conv3_f32");
  conv3_f32<MB><<<grid, 256, SMEM, ctx->stream>>>(a);
  return 0;
}

// conv3 + chained conv1 (+ pool): the (MB, MB1) pairs the reference's architectures need
template <int MB, int MB1, bool POOL>
int launch3_fused(fpl_ctx *ctx, Conv3F &a, int n) {
  constexpr int SMEM = TILE_BYTES;
  static bool attr_set[FPL_MAX_DEVICES] = {false};
  if (!attr_set[ctx->device % FPL_MAX_DEVICES]) {
    FPL_HIP(ctx, hipFuncSetAttribute((const void *)conv3_f32<MB, MB1, POOL>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_set[ctx->device % FPL_MAX_DEVICES] = true;
  }
  a.zblocks = (int)ceil_div64(a.OD, 4);
  dim3 grid((unsigned)ceil_div64(a.OW, 16), (unsigned)ceil_div64(a.OH, 4), (unsigned)(n * a.zblocks));
  TimedLaunch tl(ctx, POOL ? "mfma_conv3_conv1_pool_f32" : "mfma_conv3_conv1_f32");
  conv3_f32<MB, MB1, POOL><<<grid, 256, SMEM, ctx->stream>>>(a);
  return 0;
}

template <int MB, int MB1>
int launch_stem_fused(fpl_ctx *ctx, StemF &a, int n) {
  a.zblocks = (int)ceil_div64(a.OD, ST_Z);
  dim3 grid((unsigned)ceil_div64(a.OW, ST_X), (unsigned)ceil_div64(a.OH, ST_Y), (unsigned)(n * a.zblocks));
  TimedLaunch tl(ctx, "mfma_stem_conv1_pool_f32");
  stem_conv1_pool_f32<MB, MB1><<<grid, 256, 0, ctx->stream>>>(a);
  return 0;
}

template <int MB>
int launch1(fpl_ctx *ctx, Conv1F &a) {
  const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(a.M, 64), (int64_t)ctx->n_cu * 8);
  TimedLaunch tl(ctx, "mfma_conv1_f32");
  // weights-in-registers form for the 48- and 96-channel inputs of the vgg / U-Net blocks
  if constexpr (MB == 3) {
    if (a.cin == 48 && a.bsx) {
      if (a.pg.x) conv1_f32_wreg<3, 3, true, true, true, true><<<grid, 256, 0, ctx->stream>>>(a);
      else conv1_f32_wreg<3, 3, true, true, true><<<grid, 256, 0, ctx->stream>>>(a);
      return 0;
    }
    if (a.pg.x) return fpl_fail(ctx, "conv1: the pooled-gradient view needs the BatchNorm-statistics form");
  }
  if (a.bsx) return fpl_fail(ctx, "conv1: no BatchNorm-statistics epilogue for %d -> %d", a.cin, a.cout);
  if constexpr (MB <= 3) {
    if (a.cin == 48 && a.bn.mean) {
      if (a.stats) conv1_f32_wreg<MB, 3, true, true><<<grid, 256, 0, ctx->stream>>>(a);
      else conv1_f32_wreg<MB, 3, false, true><<<grid, 256, 0, ctx->stream>>>(a);
      return 0;
    }
    if (a.cin == 48 && !getenv("FPL_CONV1_GENERIC")) {
      if (a.stats) conv1_f32_wreg<MB, 3, true><<<grid, 256, 0, ctx->stream>>>(a);
      else conv1_f32_wreg<MB, 3, false><<<grid, 256, 0, ctx->stream>>>(a);
      return 0;
    }
  }
  if (a.bn.mean) return fpl_fail(ctx, "conv1: no BatchNorm-view kernel for %d -> %d", a.cin, a.cout);
  if (a.stats) conv1_f32<MB, true><<<grid, 256, 0, ctx->stream>>>(a);
  else conv1_f32<MB, false><<<grid, 256, 0, ctx->stream>>>(a);
  return 0;
}

}  // namespace

// every op kind; conv3 needs cout <= 64 and <= 12 channel chunks
bool fpl_mfma_f32_supported(const fpl_program *prog) {
  // tensors that exist only as an index remap (up / crop / concat): consumable by a
  // multi-channel conv3 (its tile loader applies the remap), nothing else
  std::vector<char> is_view(prog->n_tensors, 0);
  for (auto &op : prog->ops) {
    const bool v0 = is_view[op.src0], v1 = op.src1 >= 0 && is_view[op.src1];
    if (op.kind == FPL_OP_UP || op.kind == FPL_OP_CROP) {
      if (v0) return false;
      is_view[op.dst] = 1;
    } else if (op.kind == FPL_OP_CONCAT) {
      is_view[op.dst] = 1;
    } else if (op.kind == FPL_OP_CONV && op.k == 3 && op.cin > 1) {
      if (v1) return false;
    } else if (op.kind == FPL_OP_ADD) {
      // operands may be cropped views (checked for up / concat at run time)
    } else if (v0 || v1) {
      return false;
    }
    if (prog->out_tensor == op.dst && is_view[op.dst]) return false;
  }
  for (auto &op : prog->ops) {
    if (op.kind == FPL_OP_POOL && (op.p[0] != 2 || op.p[1] != 2 || op.p[2] != 2)) return false;
    if (op.kind == FPL_OP_UP && (op.p[0] != op.p[1] || op.p[1] != op.p[2] || (op.p[0] != 1 && op.p[0] != 2)))
      return false;
    if (op.kind == FPL_OP_CROP && !(op.p[0] == op.p[1] && op.p[1] == op.p[2] && op.p[2] == op.p[3] &&
                                    op.p[3] == op.p[4] && op.p[4] == op.p[5]))
      return false;
    if (op.kind == FPL_OP_CONV) {
      if (op.k == 3 && (op.cout > 256 || op.cin > 12 * 16)) return false;
      if (op.k == 1 && op.cout > 128) return false;
    }
  }
  return true;
}

// in: (n, T,T,T) f32 on the device (cubic tiles); out: (n, d,d,d, c) f32
int fpl_forward_mfma_f32(fpl_ctx *ctx, fpl_program *prog, const float *in, int n, int T,
                         float *out) {
  FPL_REQUIRE(ctx, fpl_mfma_f32_supported(prog), "program has ops the fp32 MFMA executor lacks");
  F32State *st = (F32State *)prog->fast_state_f32;
  if (!st) {
    st = new F32State();
    prog->fast_state_f32 = st;
    prog->fast_state_f32_free = f32_state_free;
  }
  const float *A = prog->arena_host.data();
  if (st->version != prog->arena_version) {
    std::vector<float> all;
    st->off.assign(prog->ops.size(), 0);
    for (size_t i = 0; i < prog->ops.size(); ++i) {
      const fpl_op &op = prog->ops[i];
      if (op.kind != FPL_OP_CONV) continue;
      std::vector<float> f;
      const int mb = (op.cout + 15) / 16;
      if (op.k == 3 && op.cin == 1) pack_stem_f32(A + op.w_off, A + op.scale_off, op.cout, mb, &f);
      else if (op.k == 3) {
        // 64 output channels per launch: slices back to back
        for (int c0 = 0; c0 < op.cout; c0 += 64) {
          std::vector<float> fs;
          pack_conv3_f32(A + op.w_off, A + op.scale_off, op.cin, op.cout,
                         (std::min(64, op.cout - c0) + 15) / 16, &fs, c0);
          f.insert(f.end(), fs.begin(), fs.end());
        }
      }
      else pack_conv1_f32(A + op.w_off, A + op.scale_off, op.cin, op.cout, mb, &f);
      st->off[i] = all.size();
      all.insert(all.end(), f.begin(), f.end());
    }
    if (st->frags) FPL_HIP(ctx, hipFree(st->frags));
    st->frags = nullptr;
    FPL_HIP(ctx, hipMalloc((void **)&st->frags, all.size() * sizeof(float)));
    FPL_HIP(ctx, hipMemcpy(st->frags, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice));
    st->version = prog->arena_version;
  }
  DevTemp tmp(ctx);
  std::vector<View> view(prog->n_tensors);
  view[0].p = in; view[0].D = T; view[0].C = 1; view[0].dim = T;
  hipStream_t stm = ctx->stream;
  auto cube = [](int d) { return (int64_t)d * d * d; };
  for (size_t i = 0; i < prog->ops.size(); ++i) {
    const fpl_op &op = prog->ops[i];
    const View a = view[op.src0];
    FPL_REQUIRE(ctx, a.dim > 0, "op %zu reads a tensor before it is produced", i);
    View o;
    switch (op.kind) {
      case FPL_OP_UP:
        FPL_REQUIRE(ctx, a.up == 1 && a.crop == 0, "op %zu: nested views", i);
        o = a; o.up = op.p[0]; o.dim = a.dim * op.p[0];
        view[op.dst] = o;
        continue;
      case FPL_OP_CROP:
        FPL_REQUIRE(ctx, a.up == 1 && a.crop == 0, "op %zu: nested views", i);
        o = a; o.crop = op.p[0]; o.dim = a.dim - 2 * op.p[0];
        view[op.dst] = o;
        continue;
      case FPL_OP_CONCAT: {
        // kept virtual: remembered as two views; only a conv3 may consume it
        const View b = view[op.src1];
        FPL_REQUIRE(ctx, a.dim == b.dim, "op %zu: concatenate of %d^3 with %d^3 - input size is "
                    "not compatible with this architecture", i, a.dim, b.dim);
        o = a; o.p = nullptr; o.dim = a.dim; o.C = a.C + b.C;
        view[op.dst] = o;
        continue;
      }
      default: break;
    }
    // conv3 -> conv1 (-> pool2) run as ONE kernel when the intermediate tensors have no
    // other reader: the 1x1x1 conv is chained in registers, the pool is an epilogue
    // (FPL_F32_UNFUSED=1: one kernel per op, for A/B runs)
    int fuse1 = -1, fusep = -1;
    if (op.kind == FPL_OP_CONV && op.k == 3 && i + 1 < prog->ops.size() && !getenv("FPL_F32_UNFUSED")) {
      auto readers = [&](int t) {
        int cnt = t == prog->out_tensor ? 1 : 0;
        for (auto &o2 : prog->ops) cnt += (o2.src0 == t) + (o2.src1 == t);
        return cnt;
      };
      const fpl_op &n1 = prog->ops[i + 1];
      const int mb0 = (op.cout + 15) / 16, mb1 = (n1.cout + 15) / 16;
      const bool pair_ok = (mb0 == mb1) && (mb0 == 2 || mb0 == 3 || mb0 == 4) && op.cout % 4 == 0;
      if (n1.kind == FPL_OP_CONV && n1.k == 1 && n1.src0 == op.dst && readers(op.dst) == 1 &&
          pair_ok && (op.act == FPL_ACT_RELU || op.act == FPL_ACT_NONE)) {
        fuse1 = (int)i + 1;
        if (i + 2 < prog->ops.size()) {
          const fpl_op &n2 = prog->ops[i + 2];
          if (n2.kind == FPL_OP_POOL && n2.src0 == n1.dst && readers(n1.dst) == 1 &&
              n2.p[0] == 2 && n2.p[1] == 2 && n2.p[2] == 2 &&
              (n1.act == FPL_ACT_RELU || n1.act == FPL_ACT_NONE))
            fusep = (int)i + 2;
        }
        // the Cin = 1 kernel exists only in its pooled form (mb 2 / 3); a virtual input
        // (up / crop / concat views are fine for the multi-channel kernel)
        if (op.cin == 1 && (fusep < 0 || mb0 == 4)) fuse1 = fusep = -1;
      }
    }
    float *dst;
    int od = 0, oc = 0;
    if (op.kind == FPL_OP_CONV) { od = a.dim - (op.k - 1); oc = op.cout; }
    if (op.kind == FPL_OP_POOL) { od = a.dim / 2; oc = a.C; }
    if (op.kind == FPL_OP_ADD) { od = a.dim; oc = a.C; }
    FPL_REQUIRE(ctx, od > 0, "op %zu: tile %d is too small for this architecture", i, T);
    const int od_conv = od;                       // conv3's own output edge
    const fpl_op &last = fusep >= 0 ? prog->ops[fusep] : (fuse1 >= 0 ? prog->ops[fuse1] : op);
    if (fuse1 >= 0) { oc = prog->ops[fuse1].cout; if (fusep >= 0) od = od / 2; }
    FPL_REQUIRE(ctx, od > 0, "op %zu: tile %d is too small for this architecture", i, T);
    if (last.dst == prog->out_tensor) {
      dst = out;
    } else {
      void *q;
      FPL_TRY(tmp.alloc((size_t)n * cube(od) * oc * sizeof(float) + 64, &q));
      dst = (float *)q;
    }
    o.p = dst; o.D = od; o.C = oc; o.dim = od;
    if (op.kind == FPL_OP_ADD) {
      const View b = view[op.src1];
      FPL_REQUIRE(ctx, a.p && b.p && a.up == 1 && b.up == 1 && a.dim == b.dim && a.C == b.C,
                  "op %zu: add of incompatible tensors", i);
      const int64_t no = (int64_t)n * cube(od) * oc;
      TimedLaunch tl(ctx, "mfma_add_f32");
      add_view_f32<<<(unsigned)ceil_div64(no, 256), 256, 0, stm>>>(a.p, a.D, a.crop, b.p, b.D, b.crop,
                                                                   dst, no, od, oc, op.act);
    } else if (op.kind == FPL_OP_POOL) {
      FPL_REQUIRE(ctx, a.p && a.up == 1 && a.crop == 0, "op %zu: pool of a view", i);
      const int64_t no = (int64_t)n * cube(od) * oc;
      TimedLaunch tl(ctx, "mfma_pool_f32");
      pool2_f32v<<<(unsigned)ceil_div64(no, 256), 256, 0, stm>>>(a.p, dst, no, a.D, a.D, a.D, a.C, od, od, od);
    } else if (op.k == 1) {
      FPL_REQUIRE(ctx, a.p && a.up == 1 && a.crop == 0, "op %zu: conv1 of a view", i);
      Conv1F c;
      c.in = a.p; c.M = (int64_t)n * cube(a.D); c.cin = op.cin;
      c.w = st->frags + st->off[i]; c.shift = prog->arena_dev + op.shift_off; c.act = op.act;
      c.out = dst; c.cout = op.cout; c.stats = nullptr;
      const int mb = (op.cout + 15) / 16;
      switch (mb) {
        case 1: FPL_TRY(launch1<1>(ctx, c)); break;
        case 2: FPL_TRY(launch1<2>(ctx, c)); break;
        case 3: FPL_TRY(launch1<3>(ctx, c)); break;
        case 4: FPL_TRY(launch1<4>(ctx, c)); break;
        case 6: FPL_TRY(launch1<6>(ctx, c)); break;
        case 8: FPL_TRY(launch1<8>(ctx, c)); break;
        default: return fpl_fail(ctx, "op %zu: conv1 with %d output channels", i, op.cout);
      }
    } else if (op.cin == 1) {
      FPL_REQUIRE(ctx, a.p && a.up == 1 && a.crop == 0 && a.C == 1, "op %zu: stem of a view", i);
      StemF c;
      c.in = a.p; c.D = c.H = c.W = a.D;
      c.w = st->frags + st->off[i]; c.shift = prog->arena_dev + op.shift_off; c.act = op.act;
      c.out = dst; c.cout = op.cout; c.OD = c.OH = c.OW = od_conv; c.stats = nullptr;
      const int mb = (op.cout + 15) / 16;
      if (fusep >= 0) {
        const fpl_op &n1 = prog->ops[fuse1];
        c.w1 = st->frags + st->off[fuse1]; c.shift1 = prog->arena_dev + n1.shift_off;
        c.act1 = n1.act; c.cout1 = n1.cout;
        if (mb == 2) FPL_TRY((launch_stem_fused<2, 2>(ctx, c, n)));
        else FPL_TRY((launch_stem_fused<3, 3>(ctx, c, n)));
      } else
      switch (mb) {
        case 1: FPL_TRY(launch_stem<1>(ctx, c, n)); break;
        case 2: FPL_TRY(launch_stem<2>(ctx, c, n)); break;
        case 3: FPL_TRY(launch_stem<3>(ctx, c, n)); break;
        case 4: FPL_TRY(launch_stem<4>(ctx, c, n)); break;
        default: return fpl_fail(ctx, "op %zu: stem with %d output channels", i, op.cout);
      }
    } else {
      Conv3F c;
      c.ncc = 0;
      auto add_src = [&](const View &v) {
        const int chunks = (v.C + 15) / 16;
        for (int q = 0; q < chunks; ++q) {
          SrcF s;
          s.p = v.p; s.D = s.H = s.W = v.D; s.C = v.C; s.ch0 = 16 * q;
          s.up = v.up; s.crop = v.crop; s.pad = 0;
          c.src[c.ncc++] = s;
        }
      };
      // a concat input is resolved through the producing op
      bool done = false;
      for (size_t j = 0; j < i && !done; ++j)
        if (prog->ops[j].dst == op.src0 && prog->ops[j].kind == FPL_OP_CONCAT) {
          const View va = view[prog->ops[j].src0], vb = view[prog->ops[j].src1];
          FPL_REQUIRE(ctx, va.p && vb.p && va.C % 16 == 0, "op %zu: unsupported concat", i);
          add_src(va);
          add_src(vb);
          done = true;
        }
      if (!done) {
        FPL_REQUIRE(ctx, a.p, "op %zu: conv3 of an unmaterialised tensor", i);
        add_src(a);
      }
      // the packed fragments assume channel chunks of the concatenated tensor in
      // order, each source starting on a 16-channel boundary
      c.act = op.act; c.opitch = op.cout; c.OD = c.OH = c.OW = od_conv;
      c.w1 = nullptr; c.shift1 = nullptr; c.act1 = 0; c.cout1 = 0;
      if (fuse1 >= 0) {
        const fpl_op &n1 = prog->ops[fuse1];
        const int mb = (op.cout + 15) / 16;
        c.w = st->frags + st->off[i]; c.shift = prog->arena_dev + op.shift_off;
        c.out = dst; c.cout = op.cout; c.opitch = n1.cout;
        c.w1 = st->frags + st->off[fuse1]; c.shift1 = prog->arena_dev + n1.shift_off;
        c.act1 = n1.act; c.cout1 = n1.cout;
        const bool pl = fusep >= 0;
        if (mb == 2) FPL_TRY(pl ? (launch3_fused<2, 2, true>(ctx, c, n)) : (launch3_fused<2, 2, false>(ctx, c, n)));
        else if (mb == 3) FPL_TRY(pl ? (launch3_fused<3, 3, true>(ctx, c, n)) : (launch3_fused<3, 3, false>(ctx, c, n)));
        else FPL_TRY(pl ? (launch3_fused<4, 4, true>(ctx, c, n)) : (launch3_fused<4, 4, false>(ctx, c, n)));
      } else {
      size_t woff = st->off[i];
      for (int c0 = 0; c0 < op.cout; c0 += 64) {      // 64 output channels per launch
        const int cs = std::min(64, op.cout - c0), mb = (cs + 15) / 16;
        c.w = st->frags + woff; c.shift = prog->arena_dev + op.shift_off + c0;
        c.out = dst + c0; c.cout = cs;
        woff += (size_t)c.ncc * 27 * mb * 256;
        switch (mb) {
          case 1: FPL_TRY(launch3<1>(ctx, c, n)); break;
          case 2: FPL_TRY(launch3<2>(ctx, c, n)); break;
          case 3: FPL_TRY(launch3<3>(ctx, c, n)); break;
          case 4: FPL_TRY(launch3<4>(ctx, c, n)); break;
        }
      }
      }
    }
    FPL_HIP(ctx, hipGetLastError());
    if (fuse1 >= 0) {                 // the fused ops are done: their result is `last`'s
      view[last.dst] = o;
      i = (size_t)(fusep >= 0 ? fusep : fuse1);
      continue;
    }
    view[op.dst] = o;
  }
  return 0;
}

// =====================================================================================
// Training-side entry points (used by train.hip): forward / input-gradient / weight-
// gradient convolutions on the fp32 MFMA kernels.  Weights change every step, so the
// fragments are re-packed on the device per call (a few hundred KB).
// =====================================================================================
namespace {

// mode 0: forward fragments of W[tap][cin][cout]
// mode 1: input-gradient fragments: W'[tap'][cout][cin] with tap' the mirrored tap
//         (dX = valid correlation of the (k-1)-padded dY with the flipped kernel)
__global__ void pack_frags_dev(const float *__restrict__ W, float *__restrict__ out, int k3,
                               int cin, int cout, int mb, int mode, int64_t total,
                               int co_off = 0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
  int64_t t = i >> 8;
  const int b = (int)(t % mb); t /= mb;
  const int tap = (int)(t % k3); const int cc = (int)(t / k3);
  const int g = lane >> 4;
  const int kin = mode == 0 ? cin : cout, kout = mode == 0 ? cout : cin;
  const int ci = 16 * cc + 4 * g + j, co = co_off + 16 * b + (lane & 15);   // a slice of the outputs
  float v = 0.f;
  if (ci < kin && co < kout) {
    if (mode == 0) v = W[((int64_t)tap * cin + ci) * cout + co];
    else v = W[((int64_t)(k3 - 1 - tap) * cin + co) * cout + ci];
  }
  out[i] = v;
}

__global__ void pack_stem_dev(const float *__restrict__ W, float *__restrict__ out, int cout,
                              int mb, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
  int64_t t = i >> 8;
  const int b = (int)(t % mb); const int q = (int)(t / mb);
  const int tap = 16 * q + 4 * (lane >> 4) + j, co = 16 * b + (lane & 15);
  out[i] = (tap < 27 && co < cout) ? W[(int64_t)tap * cout + co] : 0.f;
}

// ---- weight gradient, 3x3x3: dW[tap][ci][co] += sum_m X[m + tap][ci] * dY[m][co].
// One workgroup = one 4 x 4 x 16 block of output voxels x one 16-channel chunk of the
// input x up to 48 output channels; the 27 taps are split over the 8 waves, each
// wave sweeps all 16 K-blocks (rows of 16 x) for its taps.  A = X^T (ci x voxel),
// B = dY (voxel x co), k-slot (j,g) of a K-block = voxel 4g + j of the row.
constexpr int WG_YP = 48 * 4;                 // dY tile pitch (bytes): 48 floats
struct WgradArgs {
  const float *x; int D, H, W, cin;
  const float *dy; int od, oh, ow, cout;
  float *dw;
  int zblocks, ncc, nco;                      // ci chunks of 16, co chunks of 48
};

// Persistent: workgroup (chunk pair, p) walks the voxel blocks p, p + P, ... with its
// accumulators in registers and adds them to dW once at the end.
__global__ __launch_bounds__(512) void conv3_wgrad_f32(WgradArgs a, int64_t total_blocks,
                                                      int nbx, int nby, int P) {
  unsigned char *xt = smem;                               // (6,6,18) x 96 B
  unsigned char *yt = smem + TILE_BYTES;                  // 256 voxels x 192 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  int bx = blockIdx.x;
  const int coc = bx % a.nco; bx /= a.nco;
  const int cc = bx % a.ncc; bx /= a.ncc;
  const int co0 = coc * 48;
  const int tap0 = wave < 3 ? 4 * wave : 12 + 3 * (wave - 3), ntap = wave < 3 ? 4 : 3;   // 8 waves
  f32x4 acc[4][3];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int64_t blk = bx; blk < total_blocks; blk += P) {
  const int x0 = (int)(blk % nbx) * 16, y0 = (int)((blk / nbx) % nby) * 4;
  const int64_t bz = blk / ((int64_t)nbx * nby);
  const int n = (int)(bz / a.zblocks), z0 = (int)(bz % a.zblocks) * 4;
  __syncthreads();                                 // previous block consumed
  // stage X chunk (16 channels) and dY (48 channels), zero outside
  for (int p = tid; p < TZ * TY * TX * 4; p += 512) {
    const int vox = p >> 2, pc = p & 3;
    const int z = z0 + vox / (TY * TX), y = y0 + (vox / TX) % TY, x = x0 + vox % TX;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (z < a.D && y < a.H && x < a.W) {
      const float *gp = a.x + ((((int64_t)n * a.D + z) * a.H + y) * a.W + x) * a.cin;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = 16 * cc + 4 * pc + q;
        if (ch < a.cin) v[q] = gp[ch];
      }
    }
    *reinterpret_cast<f32x4 *>(xt + (size_t)vox * PITCH + pc * 16) = v;
  }
  for (int p = tid; p < 256 * 12; p += 512) {
    const int vox = p / 12, pc = p % 12;
    const int z = z0 + vox / 64, y = y0 + (vox / 16) % 4, x = x0 + vox % 16;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (z < a.od && y < a.oh && x < a.ow) {
      const float *gp = a.dy + ((((int64_t)n * a.od + z) * a.oh + y) * a.ow + x) * a.cout;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = co0 + 4 * pc + q;
        if (ch < a.cout) v[q] = gp[ch];
      }
    }
    *reinterpret_cast<f32x4 *>(yt + (size_t)vox * WG_YP + pc * 16) = v;
  }
  __syncthreads();
  for (int row = 0; row < 16; ++row) {             // (z,y) rows of 16 x
    const int vz = row >> 2, vy = row & 3;
    if (z0 + vz >= a.od || y0 + vy >= a.oh) continue;   // a row of zeros (ragged extent)
    float bv[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        bv[b][j] = *reinterpret_cast<const float *>(
            yt + (size_t)((vz * 4 + vy) * 16 + 4 * g + j) * WG_YP + (16 * b + c) * 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < ntap) {
        const int tap = tap0 + t;
        const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
        float av[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          av[j] = *reinterpret_cast<const float *>(
              xt + (size_t)(((vz + tz) * TY + vy + ty) * TX + 4 * g + j + tx) * PITCH + c * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int b = 0; b < 3; ++b) acc[t][b] = mfma4(av[j], bv[b][j], acc[t][b]);
      }
    }
  }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < ntap) {
      const int tap = tap0 + t;
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ci = 16 * cc + 4 * g + r, co = co0 + 16 * b + c;
          if (ci < a.cin && co < a.cout && acc[t][b][r] != 0.f)
            atomicAdd(&a.dw[((int64_t)tap * a.cin + ci) * a.cout + co], acc[t][b][r]);
        }
    }
  }
}


// Weight gradient of a first layer (cin = 1): dW[tap][co] += sum_m x[m + tap] * dY[m][co], a
// (27 x voxels) x (voxels x cout) product, HBM-bound on dY.  No LDS, no barriers (round 3;
// the LDS-tile form of round 2 was removed in round 5): a wave walks rows of 16 output voxels;
// lane (c, g) of K-step j reads its dY scalars (voxel 4j + g, channels 16b + c: 64
// contiguous bytes per 16 lanes) and its two shifted input scalars (taps c and 16 + c; the
// input volume stays in L2) straight from global memory, one row ahead of the MFMAs.  (The
// LDS form ran load -> barrier -> multiply -> barrier and reached a quarter of the dY
// stream's HBM rate.)
// BG: dY is the input gradient of the BatchNorm (+ ReLU) that follows this convolution, made
// from that layer's output gradient and input while loading (FplBnGrad, fast_paths.h).
template <bool BG>
__global__ __launch_bounds__(256) void conv3_wgrad_cin1_direct_f32(WgradArgs a, int64_t rows, int nbx,
                                                                   FplBnGrad bg) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  // BG: this lane's three channels' constants
  float bm[3], bs[3], bgam[3], bbet[3], bs0[3], bs1[3];
  if (BG) {
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int co = 16 * b + c;
      const bool ok = co < a.cout;
      bm[b] = ok ? bg.bn.mean[co] : 0.f; bs[b] = ok ? bg.bn.invstd[co] : 0.f;
      bgam[b] = ok ? bg.bn.gamma[co] : 0.f; bbet[b] = ok ? bg.bn.beta[co] : 0.f;
      bs0[b] = ok ? bg.sum_g[co] : 0.f; bs1[b] = ok ? bg.sum_gx[co] : 0.f;
    }
  }
  int64_t toff[2];
  bool tap_ok[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int t = 16 * mb + c;
    tap_ok[mb] = t < 27;
    toff[mb] = tap_ok[mb] ? ((int64_t)(t / 9) * a.H + (t / 3) % 3) * a.W + t % 3 : 0;
  }
  f32x4 acc[2][3];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[mb][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float an[4][2], bn[4][3];
  auto fetch = [&](int64_t row) {
    const int x0 = (int)(row % nbx) * 16;
    int64_t t = row / nbx;
    const int y = (int)(t % a.oh); t /= a.oh;
    const int z = (int)(t % a.od);
    const int64_t n = t / a.od;
    const float *xr = a.x + (((int64_t)n * a.D + z) * a.H + y) * a.W;
    const int64_t yoff = ((((int64_t)n * a.od + z) * a.oh + y) * a.ow) * a.cout;
    const float *yr = (BG ? bg.g : a.dy) + yoff;
    const float *yx = BG ? bg.x + yoff : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xv = x0 + 4 * j + g;
      const bool ok = xv < a.ow;
      const int xs = ok ? xv : 0;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int co = 16 * b + c;
        const bool okc = ok && co < a.cout;
        if (BG) {
          // train.hip::bn_backward4<RELU = true>, the same operations in the same order
          const float d = okc ? yr[(int64_t)xs * a.cout + co] : 0.f;
          const float xq = okc ? yx[(int64_t)xs * a.cout + co] : 0.f;
          const float gq = __fmaf_rn(__fmul_rn(__fsub_rn(xq, bm[b]), bs[b]), bgam[b], bbet[b]) > 0.f ? d : 0.f;
          const float xh = (xq - bm[b]) * bs[b];
          const float v = bgam[b] * bs[b] * (gq - bg.inv_m * bs0[b] - xh * bg.inv_m * bs1[b]);
          bn[j][b] = okc ? v : 0.f;
        } else {
          bn[j][b] = okc ? yr[(int64_t)xs * a.cout + co] : 0.f;
        }
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) an[j][mb] = (ok && tap_ok[mb]) ? xr[xs + toff[mb]] : 0.f;
    }
  };
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row < rows) fetch(row);
  for (; row < rows; row += stride) {
    float av[4][2], bv[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) av[j][mb] = an[j][mb];
#pragma unroll
      for (int b = 0; b < 3; ++b) bv[j][b] = bn[j][b];
    }
    if (row + stride < rows) fetch(row + stride);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[mb][b] = mfma4(av[j][mb], bv[j][b], acc[mb][b]);
  }
  // one sum per workgroup, then one atomic per element
  __shared__ float red[4][2 * 3 * 256];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][((mb * 3 + b) * 4 + r) * 64 + lane] = acc[mb][b][r];
  __syncthreads();
  for (int i = tid; i < 2 * 3 * 256; i += 256) {
    const float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    const int ln = i & 63, r = (i >> 6) & 3, mbb = i >> 8;
    const int mb = mbb / 3, b = mbb % 3;
    const int tap = 16 * mb + 4 * (ln >> 4) + r, co = 16 * b + (ln & 15);
    if (tap < 27 && co < a.cout && v != 0.f) atomicAdd(&a.dw[(int64_t)tap * a.cout + co], v);
  }
}

// ---- weight gradient, 1x1x1: dW[ci][co] += sum_m X[m][ci] * dY[m][co]; one workgroup
// = 1024 voxels (256 per wave) x one 16-channel input chunk x up to 48 output channels
struct Wgrad1Args {
  const float *x; const float *dy; int64_t M; int cin, cout;
  float *dw; int ncc, nco;
};

__global__ __launch_bounds__(256) void conv1_wgrad_f32(Wgrad1Args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  int bx = blockIdx.x;
  const int coc = bx % a.nco; bx /= a.nco;
  const int cc = bx % a.ncc; bx /= a.ncc;
  const int co0 = coc * 48;
  const int64_t m0 = ((int64_t)bx * 4 + wave) * 256;
  f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const int ci = 16 * cc + c;
  for (int kb = 0; kb < 16; ++kb) {
    float av[4], bv[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t m = m0 + kb * 16 + 4 * g + j;
      const bool ok = m < a.M;
      av[j] = (ok && ci < a.cin) ? a.x[m * a.cin + ci] : 0.f;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int co = co0 + 16 * b + c;
        bv[b][j] = (ok && co < a.cout) ? a.dy[m * a.cout + co] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[b] = mfma4(av[j], bv[b][j], acc[b]);
  }
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cir = 16 * cc + 4 * g + r, co = co0 + 16 * b + c;
      if (cir < a.cin && co < a.cout && acc[b][r] != 0.f)
        atomicAdd(&a.dw[(int64_t)cir * a.cout + co], acc[b][r]);
    }
}

// 48 -> 48 (vgg_like's 1x1 convs): persistent workgroups, a wave keeps the whole 48 x 48
// gradient (9 accumulator tiles) and streams voxels.  Row c of A-block q is channel
// 3c + q (and column c of B-block b is channel 3c + b), so a lane's operands for one
// voxel are 12 contiguous bytes of x and of dy: every row is read once, coalesced.
// Partials per workgroup, then a deterministic sum (no float atomics).
typedef float f32x3 __attribute__((ext_vector_type(3)));

// PG: dY is the input gradient of the BatchNorm + ReLU + pool layer behind this convolution,
// made from the pooled gradient while loading (PoolGradDev above)
template <bool BN, bool PG = false>
__global__ __launch_bounds__(256) void conv1_wgrad48_f32(const float *__restrict__ x,
                                                         const float *__restrict__ dy,
                                                         int64_t M, float *__restrict__ part,
                                                         FplBnView bn, PoolGradDev pgd) {
  __shared__ float red[4][2304];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  f32x3 pm = {0.f, 0.f, 0.f}, ps = pm, pg = pm, pb = pm;     // BN: x -> relu(bn(x))
  if (BN) {
    pm = *reinterpret_cast<const f32x3 *>(bn.mean + 3 * c);
    ps = *reinterpret_cast<const f32x3 *>(bn.invstd + 3 * c);
    pg = *reinterpret_cast<const f32x3 *>(bn.gamma + 3 * c);
    pb = *reinterpret_cast<const f32x3 *>(bn.beta + 3 * c);
  }
  PoolGradK qk[3] = {};                 // PG: the pooled layer's BN and sums, channels 3c .. 3c + 2
  if (PG) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
      qk[q] = pool_grad_k(pgd.mean[3 * c + q], pgd.invstd[3 * c + q], pgd.gamma[3 * c + q], pgd.beta[3 * c + q],
                          pgd.sum_g[3 * c + q], pgd.sum_gx[3 * c + q], pgd.inv_m);
  }
  f32x4 acc[3][3];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[q][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t groups = (M + 15) / 16;
  const int64_t stride = (int64_t)gridDim.x * 4;
  // the next group's rows are in flight while this one is multiplied (same group order as
  // the two-groups-per-iteration form it replaces: same sums)
  f32x3 an[4], bn_[4];
  f32x3 xn[PG ? 4 : 1];            // PG: the pooled layer's BN input rows
  unsigned hitn[PG ? 4 : 1];       // PG: bit q = this voxel is the arg-max of channel 3c + q's window
  auto fetch = [&](int64_t grp) {
    VoxPos v0 = {};
    if (PG) v0 = pool_coords(pgd, (unsigned)min(grp * 16 + 4 * g, M - 1));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t m = grp * 16 + 4 * g + j;                    // k-slot (g, j) = voxel
      const bool ok = m < M;
      const int64_t mm = ok ? m : 0;
      an[j] = *reinterpret_cast<const f32x3 *>(x + mm * 48 + 3 * c);
      if (PG) {
        unsigned win, pos;
        // (rows past M: any window - their x rows are zeroed below)
        pool_window(pgd, ok ? pool_coords_step(pgd, v0, (unsigned)j) : VoxPos{0u, 0u, 0u, 0u}, win, pos);
        bn_[j] = *reinterpret_cast<const f32x3 *>(pgd.dyp + (int64_t)win * 48 + 3 * c);
        xn[j] = *reinterpret_cast<const f32x3 *>(pgd.x + mm * 48 + 3 * c);
        // the three arg-max bytes of channels 3c .. 3c + 2 as one (unaligned) 32-bit load; its
        // fourth byte is the next channel's, or - behind the very last window - allocator padding
        const unsigned char *ab = reinterpret_cast<const unsigned char *>(pgd.arg) + (int64_t)win * 48 + 3 * c;
        unsigned aw;
        __builtin_memcpy(&aw, ab, 4);
        hitn[j] = ((aw & 255u) == pos ? 1u : 0u) | (((aw >> 8) & 255u) == pos ? 2u : 0u) |
                  (((aw >> 16) & 255u) == pos ? 4u : 0u);
      } else {
        bn_[j] = *reinterpret_cast<const f32x3 *>(dy + mm * 48 + 3 * c);
      }
      if (!ok) an[j] = f32x3{0.f, 0.f, 0.f};
    }
  };
  int64_t grp = (int64_t)blockIdx.x * 4 + wave;
  if (grp < groups) fetch(grp);
  for (; grp < groups; grp += stride) {
    f32x3 av[4], bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { av[j] = an[j]; bv[j] = bn_[j]; }
    if (PG) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          bv[j][q] = pool_grad_value(bv[j][q], ((hitn[j] >> q) & 1u) != 0u, xn[j][q], qk[q]);
    }
    if (BN) {
      // rows past M were zeroed by fetch(); relu(bn(0)) is not 0, but their dY rows are
      // loaded from voxel 0 - zero THOSE instead
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = grp * 16 + 4 * g + j < M;
#pragma unroll
        for (int q = 0; q < 3; ++q)
          av[j][q] = ok ? bn_relu_f(av[j][q], pm[q], ps[q], pg[q], pb[q]) : 0.f;
      }
    }
    if (grp + stride < groups) fetch(grp + stride);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[q][b] = mfma4(av[j][q], bv[j][b], acc[q][b]);
  }
  // D row 4g + r of block (q, b) is ci = 3 (4g + r) + q, column c is co = 3c + b
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[wave][(3 * (4 * g + r) + q) * 48 + 3 * c + b] = acc[q][b][r];
  __syncthreads();
  for (int i = threadIdx.x; i < 2304; i += 256)
    part[(int64_t)blockIdx.x * 2304 + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// dw[i] += sum over the nblk partials: 64 consecutive i per workgroup, four rows of
// partials in flight per wave, fixed order
__global__ __launch_bounds__(256) void wgrad_partials_add(const float *__restrict__ part, int nblk,
                                                          int n, float *__restrict__ dw) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (i < n)
    for (int k = wave; k < nblk; k += 4) s += part[(int64_t)k * n + i];
  sh[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && i < n) dw[i] += (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}

}  // namespace

bool fpl_tm_supported(int k, int cin, int cout) {
  // 3x3x3: 64 output channels per launch (wider ones in slices, which need 16-B aligned
  // channel offsets), <= 12 input chunks of 16
  if (k == 3) return (cout <= 64 || cout % 4 == 0) && cout <= 256 && cin <= 12 * 16;
  return k == 1 && cout <= 128;
}

// input + weight gradients of a conv the forward supports: the 3x3x3 input gradient
// comes in 64-channel slices, so cin may be wide; its K (= cout) is <= 12 chunks
bool fpl_tm_bwd_supported(int k, int cin, int cout) {
  if (!fpl_tm_supported(k, cin, cout)) return false;
  if (k == 3) return cout <= 12 * 16 && cin % 4 == 0 ? true : fpl_tm_supported(k, cout, cin);
  return fpl_tm_supported(k, cout, cin);
}

// y = act(conv(x, W) + bias); x (n,D,H,W,cin), W [k^3][cin][cout] on the device
// rows of per-channel statistics partials fpl_tm_conv_fwd writes for this shape when
// given a `stats` buffer (rows x 2 x cout doubles); 0 = that kernel has no fused statistics
int64_t fpl_tm_conv_stats_rows(fpl_ctx *ctx, int n, int D, int H, int W_, int cin, int k, int cout) {
  if (cout > 64) return 0;
  const int od = D - k + 1, oh = H - k + 1, ow = W_ - k + 1;
  if (k == 3 && cin == 1)
    return ceil_div64(ow, ST_X) * ceil_div64(oh, ST_Y) * n * ceil_div64(od, ST_Z);
  if (k == 1) return std::min<int64_t>(ceil_div64((int64_t)n * D * H * W_, 64), (int64_t)ctx->n_cu * 8);
  return 0;
}

// both the forward and the weight-gradient kernel must take the view
bool fpl_tm_bn_view_supported(int k, int cin, int cout) { return k == 1 && cin == 48 && cout == 48; }

int fpl_tm_conv_fwd(fpl_ctx *ctx, const float *x, int n, int D, int H, int W_, int cin, int k,
                    int cout, const float *Wd, const float *bias, int act, float *y,
                    double *stats, const FplBnView *bn) {
  FPL_REQUIRE(ctx, !bn || fpl_tm_bn_view_supported(k, cin, cout),
              "conv fwd: no BatchNorm-view kernel for k %d, %d -> %d", k, cin, cout);
  DevTemp tmp(ctx);
  const int mb = (cout + 15) / 16;
  const int od = D - k + 1, oh = H - k + 1, ow = W_ - k + 1;
  void *fr;
  if (k == 3 && cin == 1) {
    const int64_t tot = (int64_t)2 * mb * 256;
    FPL_TRY(tmp.alloc(tot * 4, &fr));
    pack_stem_dev<<<(unsigned)ceil_div64(tot, 256), 256, 0, ctx->stream>>>(Wd, (float *)fr, cout, mb, tot);
    StemF a;
    a.in = x; a.D = D; a.H = H; a.W = W_; a.w = (const float *)fr; a.shift = bias; a.act = act;
    a.out = y; a.cout = cout; a.OD = od; a.OH = oh; a.OW = ow; a.stats = stats;
    switch (mb) {
      case 1: return launch_stem<1>(ctx, a, n);
      case 2: return launch_stem<2>(ctx, a, n);
      case 3: return launch_stem<3>(ctx, a, n);
      case 4: return launch_stem<4>(ctx, a, n);
    }
    return fpl_fail(ctx, "stem with %d channels", cout);
  }
  // 32 - 192 channels: split halves (FPL_TRAIN_F32CONV=1: the fp32 MFMA kernel, A/B)
  if (fpl_tm_conv3_split_supported(k, cin, cout) && D == H && H == W_ && (act == FPL_ACT_NONE || act == FPL_ACT_RELU) &&
      !getenv("FPL_TRAIN_F32CONV"))
    return fpl_tm_conv3_split(ctx, x, n, D, H, W_, cin, cout, Wd, bias, 0, act == FPL_ACT_RELU, y);
  const int ncc = (cin + 15) / 16, k3 = k * k * k;
  const int64_t tot = (int64_t)ncc * k3 * mb * 256;
  FPL_TRY(tmp.alloc(tot * 4, &fr));
  pack_frags_dev<<<(unsigned)ceil_div64(tot, 256), 256, 0, ctx->stream>>>(Wd, (float *)fr, k3, cin, cout, mb, 0, tot);
  if (k == 1) {
    Conv1F c;
    c.in = x; c.M = (int64_t)n * D * H * W_; c.cin = cin; c.w = (const float *)fr; c.shift = bias;
    c.act = act; c.out = y; c.cout = cout; c.stats = stats;
    if (bn) c.bn = *bn;
    switch (mb) {
      case 1: return launch1<1>(ctx, c);
      case 2: return launch1<2>(ctx, c);
      case 3: return launch1<3>(ctx, c);
      case 4: return launch1<4>(ctx, c);
      case 6: return launch1<6>(ctx, c);
      case 8: return launch1<8>(ctx, c);
    }
    return fpl_fail(ctx, "conv1 with %d channels", cout);
  }
  for (int c0 = 0; c0 < cout; c0 += 64) {          // 64 output channels per launch
    const int cs = std::min(64, cout - c0), mbs = (cs + 15) / 16;
    const int64_t tots = (int64_t)ncc * k3 * mbs * 256;
    void *frs;
    FPL_TRY(tmp.alloc(tots * 4, &frs));
    pack_frags_dev<<<(unsigned)ceil_div64(tots, 256), 256, 0, ctx->stream>>>(Wd, (float *)frs, k3, cin, cout, mbs, 0, tots, c0);
    Conv3F c;
    c.ncc = ncc;
    for (int q = 0; q < ncc; ++q) {
      SrcF s;
      s.p = x; s.D = D; s.H = H; s.W = W_; s.C = cin; s.ch0 = 16 * q; s.up = 1; s.crop = 0; s.pad = 0;
      c.src[q] = s;
    }
    c.w = (const float *)frs; c.shift = bias + c0; c.act = act; c.out = y + c0; c.cout = cs; c.opitch = cout;
    c.OD = od; c.OH = oh; c.OW = ow;
    switch (mbs) {
      case 1: FPL_TRY(launch3<1>(ctx, c, n)); break;
      case 2: FPL_TRY(launch3<2>(ctx, c, n)); break;
      case 3: FPL_TRY(launch3<3>(ctx, c, n)); break;
      case 4: FPL_TRY(launch3<4>(ctx, c, n)); break;
    }
  }
  return 0;
}

// dx (n,D,H,W,cin) = input gradient of the valid conv for dy (n,od,oh,ow,cout);
// dx is OVERWRITTEN (the caller accumulates when a tensor has several consumers)
int fpl_tm_conv_dgrad(fpl_ctx *ctx, const float *dy, int n, int od, int oh, int ow, int cout,
                      int k, int cin, const float *Wd, const float *zeros, float *dx,
                      const FplBnStat *bstat, const FplPoolGrad *pg) {
  FPL_REQUIRE(ctx, !bstat || fpl_tm_bn_view_supported(k, cin, cout),
              "conv dgrad: no BatchNorm-statistics epilogue for k %d, %d -> %d", k, cin, cout);
  FPL_REQUIRE(ctx, !pg || (bstat && fpl_tm_pool_grad_supported(k, cin, cout)),
              "conv dgrad: no pooled-gradient view for k %d, %d -> %d", k, cin, cout);
  DevTemp tmp(ctx);
  const int ncc = (cout + 15) / 16, k3 = k * k * k;
  if (k == 1) {
    const int mb = (cin + 15) / 16;            // outputs of this "conv" = cin
    const int64_t tot = (int64_t)ncc * k3 * mb * 256;
    void *fr;
    FPL_TRY(tmp.alloc(tot * 4, &fr));
    pack_frags_dev<<<(unsigned)ceil_div64(tot, 256), 256, 0, ctx->stream>>>(Wd, (float *)fr, k3, cin, cout, mb, 1, tot);
    Conv1F c;
    c.in = dy; c.M = (int64_t)n * od * oh * ow; c.cin = cout; c.w = (const float *)fr;
    c.shift = zeros; c.act = FPL_ACT_NONE; c.out = dx; c.cout = cin; c.stats = nullptr;
    if (bstat) { c.bsx = bstat->x; c.bn = bstat->bn; c.stats = bstat->part; }
    if (pg) c.pg = pool_grad_dev(*pg);
    switch (mb) {
      case 1: return launch1<1>(ctx, c);
      case 2: return launch1<2>(ctx, c);
      case 3: return launch1<3>(ctx, c);
      case 4: return launch1<4>(ctx, c);
      case 6: return launch1<6>(ctx, c);
      case 8: return launch1<8>(ctx, c);
    }
    return fpl_fail(ctx, "conv1 dgrad with %d channels", cin);
  }
  if (fpl_tm_conv3_split_supported(k, cin, cout) && od == oh && oh == ow && !getenv("FPL_TRAIN_F32CONV"))
    return fpl_tm_conv3_split(ctx, dy, n, od, oh, ow, cin, cout, Wd, zeros, 1, 0, dx);
  // 3x3x3: the kernel makes up to 64 channels per launch; a wider input gradient (unet's
  // 192- and 96-channel concats) is produced in 64-channel slices of the same tensor
  FPL_REQUIRE(ctx, ncc <= 12, "conv3 dgrad: %d > 192 output-gradient channels", cout);
  for (int c0 = 0; c0 < cin; c0 += 64) {
    const int cs = std::min(64, cin - c0), mb = (cs + 15) / 16;
    const int64_t tot = (int64_t)ncc * k3 * mb * 256;
    void *fr;
    FPL_TRY(tmp.alloc(tot * 4, &fr));
    pack_frags_dev<<<(unsigned)ceil_div64(tot, 256), 256, 0, ctx->stream>>>(Wd, (float *)fr, k3, cin, cout, mb, 1, tot, c0);
    Conv3F c;
    c.ncc = ncc;
    for (int q = 0; q < ncc; ++q) {
      SrcF s;
      s.p = dy; s.D = od; s.H = oh; s.W = ow; s.C = cout; s.ch0 = 16 * q; s.up = 1; s.crop = 0;
      s.pad = k - 1;
      c.src[q] = s;
    }
    c.w = (const float *)fr; c.shift = zeros; c.act = FPL_ACT_NONE;
    c.out = dx + c0; c.cout = cs; c.opitch = cin;
    c.OD = od + k - 1; c.OH = oh + k - 1; c.OW = ow + k - 1;
    switch (mb) {
      case 1: FPL_TRY(launch3<1>(ctx, c, n)); break;
      case 2: FPL_TRY(launch3<2>(ctx, c, n)); break;
      case 3: FPL_TRY(launch3<3>(ctx, c, n)); break;
      case 4: FPL_TRY(launch3<4>(ctx, c, n)); break;
      default: return fpl_fail(ctx, "conv3 dgrad slice with %d channels", cs);
    }
  }
  return 0;
}

// dw [k^3][cin][cout] += weight gradient (float atomics)
bool fpl_tm_pool_grad_supported(int k, int cin, int cout) { return k == 1 && cin == 48 && cout == 48; }

bool fpl_tm_bn_grad_supported(int k, int cin, int cout) {
  return k == 3 && cin == 1 && cout <= 48;
}

int fpl_tm_conv_wgrad(fpl_ctx *ctx, const float *x, int n, int D, int H, int W_, int cin,
                      const float *dy, int k, int cout, float *dw, const FplBnView *bn,
                      const FplBnGrad *bg, const FplPoolGrad *pg) {
  FPL_REQUIRE(ctx, !pg || fpl_tm_pool_grad_supported(k, cin, cout),
              "conv wgrad: no pooled-gradient view for k %d, %d -> %d", k, cin, cout);
  FPL_REQUIRE(ctx, !bn || fpl_tm_bn_view_supported(k, cin, cout),
              "conv wgrad: no BatchNorm-view kernel for k %d, %d -> %d", k, cin, cout);
  FPL_REQUIRE(ctx, !bg || fpl_tm_bn_grad_supported(k, cin, cout),
              "conv wgrad: no BatchNorm-gradient kernel for k %d, %d -> %d", k, cin, cout);
  const int od = D - k + 1, oh = H - k + 1, ow = W_ - k + 1;
  const int ncc = (cin + 15) / 16, nco = (cout + 47) / 48;
  // rows of up to 32 outputs: split halves, voxel-major MFMAs (FPL_TRAIN_F32CONV=1: the fp32 kernel)
  // (more than two launches of 32 input channels over rows shorter than 12 outputs lose to the fp32 kernel:
  // unet_like2's 192 -> 64 at 4^3 0.36 against 0.19 ms, 96 -> 32 at 6^3 0.17 against 0.16)
  if (fpl_tm_conv3_wgrad_split_supported(k, cin, cout) && D == H && H == W_ && od <= 32 && (cin <= 64 || od >= 12) && !bn && !bg && !pg &&
      !getenv("FPL_TRAIN_F32CONV"))
    return fpl_tm_conv3_wgrad_split(ctx, x, n, D, cin, cout, dy, dw);
  DevTemp tmp(ctx);
  if (k == 1) {
    Wgrad1Args a;
    a.x = x; a.dy = dy; a.M = (int64_t)n * D * H * W_; a.cin = cin; a.cout = cout; a.dw = dw;
    a.ncc = ncc; a.nco = nco;
    if (cin == 48 && cout == 48) {
      const int nblk = (int)std::min<int64_t>((int64_t)ctx->n_cu * 4, ceil_div64(a.M, 128));
      void *part;
      FPL_TRY(tmp.alloc((size_t)nblk * 2304 * 4, &part));
      TimedLaunch tl(ctx, "mfma_wgrad1_f32");
      const PoolGradDev pgd = pg ? pool_grad_dev(*pg) : PoolGradDev{};
      if (bn && pg) conv1_wgrad48_f32<true, true><<<nblk, 256, 0, ctx->stream>>>(x, dy, a.M, (float *)part, *bn, pgd);
      else if (pg) conv1_wgrad48_f32<false, true><<<nblk, 256, 0, ctx->stream>>>(x, dy, a.M, (float *)part, FplBnView{}, pgd);
      else if (bn) conv1_wgrad48_f32<true><<<nblk, 256, 0, ctx->stream>>>(x, dy, a.M, (float *)part, *bn, pgd);
      else conv1_wgrad48_f32<false><<<nblk, 256, 0, ctx->stream>>>(x, dy, a.M, (float *)part, FplBnView{}, pgd);
      wgrad_partials_add<<<36, 256, 0, ctx->stream>>>((const float *)part, nblk, 2304, dw);
      return 0;
    }
    const int64_t vb = ceil_div64(a.M, 1024);
    TimedLaunch tl(ctx, "mfma_wgrad1_f32");
    conv1_wgrad_f32<<<(unsigned)(vb * ncc * nco), 256, 0, ctx->stream>>>(a);
    return 0;
  }
  WgradArgs a;
  a.x = x; a.D = D; a.H = H; a.W = W_; a.cin = cin; a.dy = dy; a.od = od; a.oh = oh; a.ow = ow;
  a.cout = cout; a.dw = dw; a.zblocks = (int)ceil_div64(od, 4); a.ncc = ncc; a.nco = nco;
  if (cin == 1 && cout <= 48) {
    const int nbx = (int)ceil_div64(ow, 16);
    TimedLaunch tl(ctx, "mfma_wgrad3_f32_cin1");
    const int64_t rows = (int64_t)n * od * oh * nbx;
    const unsigned gridd = (unsigned)std::min<int64_t>(ceil_div64(rows, 4), (int64_t)ctx->n_cu * 8);
    if (bg) conv3_wgrad_cin1_direct_f32<true><<<gridd, 256, 0, ctx->stream>>>(a, rows, nbx, *bg);
    else conv3_wgrad_cin1_direct_f32<false><<<gridd, 256, 0, ctx->stream>>>(a, rows, nbx, FplBnGrad{});
    return 0;
  }
  constexpr int SMEM = TILE_BYTES + 256 * WG_YP;
  // function attributes belong to the current device: one flag per device (a process may
  // drive several GPUs, one context each; setting it twice is harmless)
  static bool attr_set[FPL_MAX_DEVICES] = {false};
  if (!attr_set[ctx->device % FPL_MAX_DEVICES]) {
    FPL_HIP(ctx, hipFuncSetAttribute((const void *)conv3_wgrad_f32,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_set[ctx->device % FPL_MAX_DEVICES] = true;
  }
  const int nbx = (int)ceil_div64(ow, 16), nby = (int)ceil_div64(oh, 4);
  const int64_t total = (int64_t)nbx * nby * n * a.zblocks;
  // one workgroup per CU (110 KiB of LDS), shared out over the (ci chunk, co chunk) pairs
  const int per_cu = (160 * 1024) / SMEM;
  const int P = (int)std::max<int64_t>(1, std::min<int64_t>(total, (int64_t)ctx->n_cu * per_cu / (ncc * nco)));
  char tname[64];
  snprintf(tname, sizeof(tname), "mfma_wgrad3_f32_%dto%d", cin, cout);
  TimedLaunch tl(ctx, tname);
  conv3_wgrad_f32<<<(unsigned)(P * ncc * nco), 512, SMEM, ctx->stream>>>(a, total, nbx, nby, P);
  return 0;
}
