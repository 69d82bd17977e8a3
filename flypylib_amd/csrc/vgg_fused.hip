// Fused bf16 MFMA kernels for vgg_like inference (flypylib/fplmodels.py:102-136)
// over a whole Z-slab of the volume - three launches per slab chunk:
//
//   vgg_stem_pool_bf16  u8/f32 volume -> normalise -> conv3 1->48 +BN+ReLU ->
//                       conv1 48->48 +BN+ReLU -> maxpool2            -> P1 (bf16)
//   vgg_mid_pool_bf16   P1 -> conv3 48->48 +BN+ReLU -> conv1 48->48 +BN+ReLU ->
//                       maxpool2                                     -> P2 (bf16)
//   vgg_head_bf16       P2 -> conv3 48->48 -> conv1 48->96 -> conv1 96->96 ->
//                       conv1 96->1 +bias -> sigmoid -> x4 nearest upsample,
//                       stored straight into the (Z,Y,X) f32 prediction volume
//
// The 100^3 x 48 full-resolution activations never leave registers; P1 (12 B per
// output voxel) and P2 (1.5 B) are the only intermediates in HBM.  BN is folded:
// scale into the bf16 weights, shift as the accumulators' initial value.
//
// Lattice equivalence with FplNetwork.infer (flypylib/fplnetwork.py:146-187):
// with out = 88 = 4*22 every reference tile's input origin is a multiple of the
// network stride 4, so the coarse grid is anchored at the volume origin:
// pred[7+p] = O[p/4] with O[i] seeing input [4i, 4i+18), zero (normalised) past
// the volume end - independent of the tiling.  The kernels compute O directly.
#include <algorithm>

#include "fast_paths.h"
#include "mfma_util.h"
#include "pack_weights.h"

namespace {

constexpr int CH = 48;                 // channels of P1 / P2
constexpr int VOX_BYTES = CH * 2;      // 96 B per voxel (bf16)
constexpr int KSTEPS = 42;             // 27*48 = 1296 -> 41 K-steps of 32, padded
constexpr int KCHUNK = 6;              // K-steps per weight ring slot
constexpr int NCHUNK = KSTEPS / KCHUNK;
constexpr int RING_BYTES = KCHUNK * 3 * 1024;   // 3 M-blocks x 1 KiB per K-step

// -------------------------------------------------------------------------------
// K1: stem.  WG = 4 waves; pooled block 2 x 4 x 32; wave task = 16 pooled x of one
// (pz,py) row; 8 sub-steps walk the 2x2x2 pooling window so the pool is an
// element-wise max over accumulators (no cross-lane traffic).
// -------------------------------------------------------------------------------
constexpr int S_PZ = 2, S_PY = 4, S_PX = 32;
constexpr int S_TZ = 2 * S_PZ + 2, S_TY = 2 * S_PY + 2, S_TX = 2 * S_PX + 2;

struct StemArgs {
  const void *src;
  int64_t SZ, SY, SX;      // volume dims
  float mean, sd;
  int64_t p1z0;            // global P1 row of chunk-local row 0
  const bf16x8 *w1, *w2;   // fragments [s][b][lane]
  const float *shift1, *shift2;
  __bf16 *p1;
  int P1Z, P1Y, P1X;       // chunk-local dims
};

template <typename SRC>
__global__ __launch_bounds__(256) void vgg_stem_pool_bf16(StemArgs a) {
  __shared__ unsigned short tile[S_TZ * S_TY * S_TX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int px0 = blockIdx.x * S_PX, py0 = blockIdx.y * S_PY, pz0 = blockIdx.z * S_PZ;

  // ---- input tile: normalise, round to bf16; zero past the volume end
  {
    const SRC *src = (const SRC *)a.src;
    const int64_t gz0 = 2 * (a.p1z0 + pz0), gy0 = 2 * (int64_t)py0, gx0 = 2 * (int64_t)px0;
    for (int i = tid; i < S_TZ * S_TY * S_TX; i += 256) {
      const int tx = i % S_TX, ty = (i / S_TX) % S_TY, tz = i / (S_TX * S_TY);
      const int64_t z = gz0 + tz, y = gy0 + ty, x = gx0 + tx;
      float v = 0.f;
      if (z < a.SZ && y < a.SY && x < a.SX)
        v = ((float)src[(z * a.SY + y) * a.SX + x] - a.mean) / a.sd;
      tile[i] = bf16_bits(v);
    }
  }

  // ---- per-lane constants
  int toff[8];                       // tap offsets (elements) of k-slots 8g..8g+7
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int t = 8 * g + j;
    toff[j] = t < 27 ? ((t / 9) * S_TY + (t / 3) % 3) * S_TX + t % 3 : 0;
  }
  bf16x8 w1[3], w2[2][3];
  f32x4 sh1[3], sh2[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    w1[b] = a.w1[b * 64 + lane];
    w2[0][b] = a.w2[(0 * 3 + b) * 64 + lane];
    w2[1][b] = a.w2[(1 * 3 + b) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sh1[b][r] = a.shift1[16 * b + 4 * g + r];
      sh2[b][r] = a.shift2[16 * b + 4 * g + r];
    }
  }
  __syncthreads();

  for (int task = wave; task < S_PZ * S_PY * 2; task += 4) {
    const int row = task >> 1, xh = task & 1;
    const int pzl = row / S_PY, pyl = row % S_PY;
    const int base = ((2 * pzl) * S_TY + 2 * pyl) * S_TX + 2 * (16 * xh + c);
    f32x4 pooled[3];
#pragma unroll
    for (int sub = 0; sub < 8; ++sub) {
      const int so = (((sub >> 2) & 1) * S_TY + ((sub >> 1) & 1)) * S_TX + (sub & 1);
      u16x8 raw;
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[j] = tile[base + so + toff[j]];
      const bf16x8 bfrag = __builtin_bit_cast(bf16x8, raw);
      f32x4 a1[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) a1[b] = mfma16(w1[b], bfrag, sh1[b]);
      const bf16x8 h0 = pack_relu(a1[0], a1[1]);
      const bf16x8 h1 = pack_relu_lo(a1[2]);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        f32x4 a2 = {0.f, 0.f, 0.f, 0.f};
        a2 = mfma16(w2[0][b], h0, a2);
        a2 = mfma16(w2[1][b], h1, a2);
        if (sub == 0) {
          pooled[b] = a2;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) pooled[b][r] = max1(pooled[b][r], a2[r]);
        }
      }
    }
    // relu(max(conv) + shift) == max over the window of relu(conv + shift)
    const int pz = pz0 + pzl, py = py0 + pyl, px = px0 + 16 * xh + c;
    if (pz < a.P1Z && py < a.P1Y && px < a.P1X) {
      __bf16 *dst = a.p1 + (((int64_t)pz * a.P1Y + py) * a.P1X + px) * CH + 4 * g;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (__bf16)relu1(pooled[b][r] + sh2[b][r]);
        *reinterpret_cast<bf16x4 *>(dst + 16 * b) = o;
      }
    }
  }
}

// -------------------------------------------------------------------------------
// Shared 3x3x3 48->48 implicit-GEMM K loop (K2 and K3).  The activation tile
// (TZ x TY x TX voxels x 96 B) is resident in LDS; the 126 KiB of weight
// fragments stream through a 2-slot LDS ring by LDS-DMA, one barrier per slot.
// Lane (c,g) reads, per K-step, the 16 B of its voxel (+tap) that hold k-slots
// 8g..8g+7: flat k = 32s + 8g + j over (tap, channel) -> tap = k/48, ch = k%48.
// -------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

__device__ __forceinline__ void glds16(const void *g, void *l) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void *)g,
      (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// byte offset inside the activation tile of k-slot group (s, g)
template <int TY, int TX>
__device__ __forceinline__ unsigned kslot_offset(int s, int g) {
  const int f0 = 32 * s + 8 * g;
  const int tap = f0 / CH, ch0 = f0 % CH;
  if (tap >= 27) return 0u;           // zero weights; any valid address will do
  return (unsigned)((((tap / 9) * TY + (tap / 3) % 3) * TX + tap % 3) * VOX_BYTES +
                    ch0 * 2);
}

// stage one ring slot (KCHUNK K-steps x 3 fragments, contiguous in global)
__device__ __forceinline__ void stage_weights(const unsigned char *wglobal,
                                              unsigned char *slot, int chunk,
                                              int wave, int lane) {
  const unsigned char *srcp = wglobal + (size_t)chunk * RING_BYTES;
  for (int i = wave; i < RING_BYTES / 1024; i += 4)
    glds16(srcp + i * 1024 + lane * 16, slot + i * 1024);
}

// fill the activation tile by LDS-DMA: tile is TZ*TY rows of TX voxels (96 B)
template <int TZ, int TY, int TX>
__device__ __forceinline__ void stage_tile(const __bf16 *act, int AZ, int AY, int AX,
                                           int z0, int y0, int x0,
                                           unsigned char *tile, int wave, int lane) {
  constexpr int ROW_CHUNKS = TX * VOX_BYTES / 16;          // 16-B pieces per row
  constexpr int TOTAL = TZ * TY * ROW_CHUNKS;
  constexpr int PIECES = (TOTAL + 63) / 64;
  for (int p = wave; p < PIECES; p += 4) {
    int idx = p * 64 + lane;
    idx = idx < TOTAL ? idx : TOTAL - 1;                   // tail lanes re-read
    const int row = idx / ROW_CHUNKS, cw = idx % ROW_CHUNKS;
    int z = z0 + row / TY, y = y0 + row % TY, x = x0 + cw / 6;
    z = z < AZ ? z : AZ - 1;                               // clamp: edge blocks
    y = y < AY ? y : AY - 1;                               // only feed masked
    x = x < AX ? x : AX - 1;                               // outputs
    const __bf16 *gp = act + (((int64_t)z * AY + y) * AX + x) * CH + (cw % 6) * 8;
    glds16(gp, tile + (size_t)p * 1024);
  }
}

template <int NSUB, int TY, int TX, typename SubOff>
__device__ __forceinline__ void conv3_kloop(const unsigned char *tile,
                                            unsigned char *ring,
                                            const unsigned *kofftab,
                                            const unsigned char *wglobal,
                                            unsigned vbase, SubOff sub_off,
                                            f32x4 (&acc)[NSUB][3], int wave,
                                            int lane) {
  const int g = lane >> 4;
  for (int ck = 0; ck < NCHUNK; ++ck) {
    __syncthreads();          // slot ck landed (vmcnt(0) + barrier); slot ck^1 free
    if (ck + 1 < NCHUNK)
      stage_weights(wglobal, ring + ((ck + 1) & 1) * RING_BYTES, ck + 1, wave, lane);
    const unsigned char *wslot = ring + (ck & 1) * RING_BYTES;
#pragma unroll
    for (int ks = 0; ks < KCHUNK; ++ks) {
      const int s = ck * KCHUNK + ks;
      const unsigned koff = kofftab[s * 4 + g];
      bf16x8 wf[3];
#pragma unroll
      for (int b = 0; b < 3; ++b)
        wf[b] = *reinterpret_cast<const bf16x8 *>(wslot + (ks * 3 + b) * 1024 +
                                                  lane * 16);
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        const bf16x8 bf = *reinterpret_cast<const bf16x8 *>(
            tile + vbase + koff + sub_off(sub));
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[sub][b] = mfma16(wf[b], bf, acc[sub][b]);
      }
    }
  }
}

// -------------------------------------------------------------------------------
// K2: conv3 48->48 + conv1 48->48 + maxpool2.  WG = 4 waves, pooled block
// 2 x 2 x 16; wave = one (pz,py) row; 8 sub-steps = pooling window positions.
// -------------------------------------------------------------------------------
constexpr int M_TZ = 6, M_TY = 6, M_TX = 34;
constexpr int M_TILE_BYTES = ((M_TZ * M_TY * M_TX * VOX_BYTES + 1023) / 1024) * 1024;
constexpr int M_SMEM = M_TILE_BYTES + 2 * RING_BYTES + KSTEPS * 4 * 4;

struct MidArgs {
  const __bf16 *p1;
  int P1Z, P1Y, P1X;
  const unsigned char *w3;       // 42 x 3 fragments
  const bf16x8 *w4;              // [s][b][lane]
  const float *shift3, *shift4;
  __bf16 *p2;
  int P2Z, P2Y, P2X;
};

__global__ __launch_bounds__(256) void vgg_mid_pool_bf16(MidArgs a) {
  unsigned char *tile = smem;
  unsigned char *ring = smem + M_TILE_BYTES;
  unsigned *kofftab = reinterpret_cast<unsigned *>(smem + M_TILE_BYTES + 2 * RING_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int px0 = blockIdx.x * 16, py0 = blockIdx.y * 2, pz0 = blockIdx.z * 2;

  if (tid < KSTEPS * 4) kofftab[tid] = kslot_offset<M_TY, M_TX>(tid >> 2, tid & 3);
  stage_tile<M_TZ, M_TY, M_TX>(a.p1, a.P1Z, a.P1Y, a.P1X, 2 * pz0, 2 * py0, 2 * px0,
                               tile, wave, lane);
  stage_weights(a.w3, ring, 0, wave, lane);

  const int pzl = wave >> 1, pyl = wave & 1;
  const unsigned vbase =
      (unsigned)((((2 * pzl) * M_TY + 2 * pyl) * M_TX + 2 * c) * VOX_BYTES);
  f32x4 acc[8][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    f32x4 sh;
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[r] = a.shift3[16 * b + 4 * g + r];
#pragma unroll
    for (int sub = 0; sub < 8; ++sub) acc[sub][b] = sh;
  }
  auto sub_off = [](int sub) -> unsigned {
    return (unsigned)(((((sub >> 2) & 1) * M_TY + ((sub >> 1) & 1)) * M_TX + (sub & 1)) *
                      VOX_BYTES);
  };
  conv3_kloop<8, M_TY, M_TX>(tile, ring, kofftab, a.w3, vbase, sub_off, acc, wave, lane);

  // conv1 48->48 chained in registers, pooled over the 8 window positions
  bf16x8 w4[2][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    w4[0][b] = a.w4[(0 * 3 + b) * 64 + lane];
    w4[1][b] = a.w4[(1 * 3 + b) * 64 + lane];
  }
  f32x4 pooled[3];
#pragma unroll
  for (int sub = 0; sub < 8; ++sub) {
    const bf16x8 h0 = pack_relu(acc[sub][0], acc[sub][1]);
    const bf16x8 h1 = pack_relu_lo(acc[sub][2]);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
      a4 = mfma16(w4[0][b], h0, a4);
      a4 = mfma16(w4[1][b], h1, a4);
      if (sub == 0) {
        pooled[b] = a4;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) pooled[b][r] = max1(pooled[b][r], a4[r]);
      }
    }
  }
  const int pz = pz0 + pzl, py = py0 + pyl, px = px0 + c;
  if (pz < a.P2Z && py < a.P2Y && px < a.P2X) {
    __bf16 *dst = a.p2 + (((int64_t)pz * a.P2Y + py) * a.P2X + px) * CH + 4 * g;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        o[r] = (__bf16)relu1(pooled[b][r] + a.shift4[16 * b + 4 * g + r]);
      *reinterpret_cast<bf16x4 *>(dst + 16 * b) = o;
    }
  }
}

// -------------------------------------------------------------------------------
// K3: conv3 48->48 -> conv1 48->96 -> conv1 96->96 -> conv1 96->1 + bias ->
// sigmoid -> x4 upsample store.  WG = 4 waves, coarse block 4(z) x 4(y) x 16(x);
// wave = one z, sub-steps = the 4 y rows.
// -------------------------------------------------------------------------------
constexpr int H_TZ = 6, H_TY = 6, H_TX = 18;
constexpr int H_TILE_BYTES = ((H_TZ * H_TY * H_TX * VOX_BYTES + 1023) / 1024) * 1024;
constexpr int H_W6 = 12, H_W7 = 18, H_W8 = 3;       // fragment counts
constexpr int H_WTAIL_BYTES = (H_W6 + H_W7 + H_W8) * 1024;
constexpr int H_SMEM = H_TILE_BYTES + 2 * RING_BYTES + H_WTAIL_BYTES + KSTEPS * 4 * 4;

struct HeadArgs {
  const __bf16 *p2;
  int P2Z, P2Y, P2X;
  const unsigned char *w5;       // 42 x 3 fragments
  const unsigned char *wtail;    // L6 [2][6], L7 [3][6], L8 [3][1] fragments
  const float *shift5, *shift6, *shift7;
  float bias8;
  float *dst;                    // (Z,Y,X) prediction volume, row 0
  int64_t DY, DX;                // its pitches
  int64_t cz0;                   // global coarse z of chunk-local coarse row 0
  int CZ, CY, CX;                // chunk-local coarse dims
  int64_t VZ, VY, VX;            // valid fine extents (dim - 14)
};

__global__ __launch_bounds__(256) void vgg_head_bf16(HeadArgs a) {
  unsigned char *tile = smem;
  unsigned char *ring = smem + H_TILE_BYTES;
  unsigned char *wtail = smem + H_TILE_BYTES + 2 * RING_BYTES;
  unsigned *kofftab =
      reinterpret_cast<unsigned *>(smem + H_TILE_BYTES + 2 * RING_BYTES + H_WTAIL_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int cx0 = blockIdx.x * 16, cy0 = blockIdx.y * 4, cz0 = blockIdx.z * 4;

  if (tid < KSTEPS * 4) kofftab[tid] = kslot_offset<H_TY, H_TX>(tid >> 2, tid & 3);
  stage_tile<H_TZ, H_TY, H_TX>(a.p2, a.P2Z, a.P2Y, a.P2X, cz0, cy0, cx0, tile, wave, lane);
  for (int i = wave; i < H_WTAIL_BYTES / 1024; i += 4)
    glds16(a.wtail + i * 1024 + lane * 16, wtail + i * 1024);
  stage_weights(a.w5, ring, 0, wave, lane);

  const unsigned vbase = (unsigned)(((wave * H_TY) * H_TX + c) * VOX_BYTES);
  f32x4 acc[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    f32x4 sh;
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[r] = a.shift5[16 * b + 4 * g + r];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) acc[sub][b] = sh;
  }
  auto sub_off = [](int sub) -> unsigned { return (unsigned)(sub * H_TX * VOX_BYTES); };
  conv3_kloop<4, H_TY, H_TX>(tile, ring, kofftab, a.w5, vbase, sub_off, acc, wave, lane);

  const bf16x8 *w6 = reinterpret_cast<const bf16x8 *>(wtail);
  const bf16x8 *w7 = w6 + H_W6 * 64;
  const bf16x8 *w8 = w7 + H_W7 * 64;
  f32x4 sh6[6], sh7[6];
#pragma unroll
  for (int b = 0; b < 6; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sh6[b][r] = a.shift6[16 * b + 4 * g + r];
      sh7[b][r] = a.shift7[16 * b + 4 * g + r];
    }

#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    // L6: 48 -> 96
    bf16x8 h[3];
    h[0] = pack_relu(acc[sub][0], acc[sub][1]);
    h[1] = pack_relu_lo(acc[sub][2]);
    f32x4 a6[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      a6[b] = mfma16(w6[(0 * 6 + b) * 64 + lane], h[0], sh6[b]);
      a6[b] = mfma16(w6[(1 * 6 + b) * 64 + lane], h[1], a6[b]);
    }
    // L7: 96 -> 96
#pragma unroll
    for (int s = 0; s < 3; ++s) h[s] = pack_relu(a6[2 * s], a6[2 * s + 1]);
    f32x4 a7[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      a7[b] = sh7[b];
#pragma unroll
      for (int s = 0; s < 3; ++s) a7[b] = mfma16(w7[(s * 6 + b) * 64 + lane], h[s], a7[b]);
    }
    // L8: 96 -> 1 (row 0 of one M-block), bias, sigmoid
#pragma unroll
    for (int s = 0; s < 3; ++s) h[s] = pack_relu(a7[2 * s], a7[2 * s + 1]);
    f32x4 a8 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 3; ++s) a8 = mfma16(w8[s * 64 + lane], h[s], a8);
    // lane (c, g=0) register 0 holds the logit of coarse voxel c
    const float logit = __shfl(a8[0], c) + a.bias8;
    const float p = 1.f / (1.f + __expf(-logit));

    // x4 upsample store: lane (c,g) writes 4 fine x of fine row (4cy+g), 4 z rows
    const int cz = cz0 + wave, cy = cy0 + sub, cx = cx0 + c;
    if (cz < a.CZ && cy < a.CY && cx < a.CX) {
      const int64_t fz0 = 4 * (a.cz0 + cz), fy = 4 * (int64_t)cy + g, fx0 = 4 * (int64_t)cx;
      if (fy < a.VY && fx0 < a.VX) {
        const int nx = (int)(a.VX - fx0 < 4 ? a.VX - fx0 : 4);
#pragma unroll
        for (int dz = 0; dz < 4; ++dz) {
          const int64_t fz = fz0 + dz;
          if (fz >= a.VZ) break;
          float *o = a.dst + ((fz + 7) * a.DY + fy + 7) * a.DX + fx0 + 7;
          if (nx == 4) {
            *reinterpret_cast<f32x4_a4 *>(o) = f32x4_a4{p, p, p, p};
          } else {
            for (int i = 0; i < nx; ++i) o[i] = p;
          }
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------
// host side: pattern match, weight packing, slab orchestration
// -------------------------------------------------------------------------------
struct VggFastState {
  uint64_t version = ~0ull;
  unsigned char *frags = nullptr;     // all bf16 fragments
  float *shifts = nullptr;            // all shift vectors
  size_t off_w[8] = {0};              // byte offsets of L1..L8 fragments
  size_t off_s[8] = {0};              // float offsets of shift1..shift8
  float bias8 = 0.f;
};

void vgg_state_free(fpl_ctx *ctx, void *p) {
  VggFastState *s = (VggFastState *)p;
  if (s->frags) hipFree(s->frags);
  if (s->shifts) hipFree(s->shifts);
  delete s;
}

bool is_vgg_like(const fpl_program *prog) {
  static const int kinds[10] = {0, 0, 1, 0, 0, 1, 0, 0, 0, 0};
  static const int ks[10] = {3, 1, 0, 3, 1, 0, 3, 1, 1, 1};
  static const int cin[10] = {1, 48, 48, 48, 48, 48, 48, 48, 96, 96};
  static const int cout[10] = {48, 48, 48, 48, 48, 48, 48, 96, 96, 1};
  if (prog->ops.size() != 10) return false;
  if (prog->stride[0] != 4 || prog->stride[1] != 4 || prog->stride[2] != 4) return false;
  for (int i = 0; i < 10; ++i) {
    const fpl_op &op = prog->ops[i];
    if (op.kind != kinds[i]) return false;
    if (op.src0 != (i == 0 ? 0 : prog->ops[i - 1].dst)) return false;
    if (op.kind == FPL_OP_CONV) {
      if (op.k != ks[i] || op.cin != cin[i] || op.cout != cout[i]) return false;
      if (op.act != (i == 9 ? FPL_ACT_SIGMOID : FPL_ACT_RELU)) return false;
    } else if (op.p[0] != 2 || op.p[1] != 2 || op.p[2] != 2) {
      return false;
    }
  }
  return prog->out_tensor == prog->ops[9].dst;
}

int vgg_prepare(fpl_ctx *ctx, fpl_program *prog, VggFastState **out) {
  VggFastState *st = (VggFastState *)prog->fast_state;
  if (!st) {
    st = new VggFastState();
    prog->fast_state = st;
    prog->fast_state_free = vgg_state_free;
  }
  *out = st;
  if (st->version == prog->arena_version) return 0;
  static const int conv_ops[8] = {0, 1, 3, 4, 6, 7, 8, 9};
  static const int mblocks[8] = {3, 3, 3, 3, 3, 6, 6, 1};
  static const int ksteps[8] = {1, 2, KSTEPS, 2, KSTEPS, 2, 3, 3};
  static const FplSlotMap maps[8] = {SLOT_STEM, SLOT_CHAIN, SLOT_SPATIAL, SLOT_CHAIN,
                                     SLOT_SPATIAL, SLOT_CHAIN, SLOT_CHAIN, SLOT_CHAIN};
  std::vector<uint16_t> all;
  std::vector<float> shifts;
  const float *A = prog->arena_host.data();
  for (int l = 0; l < 8; ++l) {
    const fpl_op &op = prog->ops[conv_ops[l]];
    std::vector<uint16_t> f;
    std::vector<float> scale(A + op.scale_off, A + op.scale_off + op.cout);
    if (l == 7) {
      // sigmoid head: scale is 1 (no BN); keep it explicit anyway
    }
    fpl_pack_frags(A + op.w_off, scale.data(), op.k * op.k * op.k, op.cin, op.cout,
                   mblocks[l], ksteps[l], maps[l], &f);
    st->off_w[l] = all.size() * sizeof(uint16_t);
    all.insert(all.end(), f.begin(), f.end());
    st->off_s[l] = shifts.size();
    shifts.insert(shifts.end(), A + op.shift_off, A + op.shift_off + op.cout);
    while (shifts.size() % 4) shifts.push_back(0.f);
  }
  st->bias8 = A[prog->ops[9].shift_off];
  if (st->frags) FPL_HIP(ctx, hipFree(st->frags));
  if (st->shifts) FPL_HIP(ctx, hipFree(st->shifts));
  st->frags = nullptr;
  st->shifts = nullptr;
  FPL_HIP(ctx, hipMalloc((void **)&st->frags, all.size() * sizeof(uint16_t)));
  FPL_HIP(ctx, hipMalloc((void **)&st->shifts, shifts.size() * sizeof(float)));
  FPL_HIP(ctx, hipMemcpy(st->frags, all.data(), all.size() * sizeof(uint16_t),
                         hipMemcpyHostToDevice));
  FPL_HIP(ctx, hipMemcpy(st->shifts, shifts.data(), shifts.size() * sizeof(float),
                         hipMemcpyHostToDevice));
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)vgg_mid_pool_bf16,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, M_SMEM));
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)vgg_head_bf16,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, H_SMEM));
  st->version = prog->arena_version;
  return 0;
}

}  // namespace

int fpl_fast_infer_volume(fpl_ctx *ctx, fpl_program *prog, const void *src,
                          int src_dtype, float mean, float sd,
                          const int64_t dims[3], const int32_t tile_in[3],
                          const int32_t offset[3], int precision,
                          const std::vector<int32_t> origins[3],
                          const int32_t out_sz[3], int32_t zb, int32_t ze,
                          float *dst, bool *handled) {
  *handled = false;
  if (precision != FPL_PREC_BF16 || !is_vgg_like(prog)) return 0;
  for (int a = 0; a < 3; ++a)
    if (offset[a] != 7 || out_sz[a] % 4 != 0) return 0;
  VggFastState *st;
  FPL_TRY(vgg_prepare(ctx, prog, &st));
  hipStream_t stream = ctx->stream;
  const int64_t SZ = dims[0], SY = dims[1], SX = dims[2];
  const int64_t VZ = SZ - 14, VY = SY - 14, VX = SX - 14;
  // coarse rows this slab owns (tile rows zb..ze-1 of the reference lattice)
  const int64_t fz_lo = (int64_t)origins[0][zb] - 7;
  const int64_t fz_hi = std::min<int64_t>((int64_t)origins[0][ze - 1] - 7 + out_sz[0], VZ);
  const int64_t cz_lo = fz_lo / 4, cz_hi = ceil_div64(fz_hi, 4);
  const int CY = (int)ceil_div64(VY, 4), CX = (int)ceil_div64(VX, 4);
  const int P2Y = CY + 2, P2X = CX + 2, P1Y = 2 * P2Y + 2, P1X = 2 * P2X + 2;
  // chunk of coarse rows bounded by a scratch budget (P1 dominates)
  const int64_t p1_row_bytes = (int64_t)P1Y * P1X * VOX_BYTES;
  const int64_t budget = (int64_t)48 << 30;
  int64_t cz_chunk = std::max<int64_t>(4, (budget / p1_row_bytes - 6) / 2);
  cz_chunk = std::min<int64_t>(cz_chunk, cz_hi - cz_lo);
  cz_chunk = (cz_chunk + 3) / 4 * 4;
  DevTemp tmp(ctx);
  void *p1v, *p2v;
  FPL_TRY(tmp.alloc((size_t)(2 * cz_chunk + 6) * p1_row_bytes, &p1v));
  FPL_TRY(tmp.alloc((size_t)(cz_chunk + 2) * P2Y * P2X * VOX_BYTES, &p2v));
  const unsigned char *F = st->frags;
  const float *S = st->shifts;
  for (int64_t c0 = cz_lo; c0 < cz_hi; c0 += cz_chunk) {
    const int CZ = (int)std::min<int64_t>(cz_chunk, cz_hi - c0);
    const int P2Z = CZ + 2, P1Z = 2 * P2Z + 2;
    {
      StemArgs a;
      a.src = src; a.SZ = SZ; a.SY = SY; a.SX = SX; a.mean = mean; a.sd = sd;
      a.p1z0 = 2 * c0;
      a.w1 = (const bf16x8 *)(F + st->off_w[0]);
      a.w2 = (const bf16x8 *)(F + st->off_w[1]);
      a.shift1 = S + st->off_s[0]; a.shift2 = S + st->off_s[1];
      a.p1 = (__bf16 *)p1v; a.P1Z = P1Z; a.P1Y = P1Y; a.P1X = P1X;
      dim3 grid((unsigned)ceil_div64(P1X, S_PX), (unsigned)ceil_div64(P1Y, S_PY),
                (unsigned)ceil_div64(P1Z, S_PZ));
      TimedLaunch tl(ctx, "vgg_stem_pool_bf16");
      if (src_dtype == FPL_U8)
        vgg_stem_pool_bf16<uint8_t><<<grid, 256, 0, stream>>>(a);
      else
        vgg_stem_pool_bf16<float><<<grid, 256, 0, stream>>>(a);
    }
    {
      MidArgs a;
      a.p1 = (const __bf16 *)p1v; a.P1Z = P1Z; a.P1Y = P1Y; a.P1X = P1X;
      a.w3 = F + st->off_w[2];
      a.w4 = (const bf16x8 *)(F + st->off_w[3]);
      a.shift3 = S + st->off_s[2]; a.shift4 = S + st->off_s[3];
      a.p2 = (__bf16 *)p2v; a.P2Z = P2Z; a.P2Y = P2Y; a.P2X = P2X;
      dim3 grid((unsigned)ceil_div64(P2X, 16), (unsigned)ceil_div64(P2Y, 2),
                (unsigned)ceil_div64(P2Z, 2));
      TimedLaunch tl(ctx, "vgg_mid_pool_bf16");
      vgg_mid_pool_bf16<<<grid, 256, M_SMEM, stream>>>(a);
    }
    {
      HeadArgs a;
      a.p2 = (const __bf16 *)p2v; a.P2Z = P2Z; a.P2Y = P2Y; a.P2X = P2X;
      a.w5 = F + st->off_w[4];
      a.wtail = F + st->off_w[5];      // L6, L7, L8 fragments are contiguous
      a.shift5 = S + st->off_s[4]; a.shift6 = S + st->off_s[5];
      a.shift7 = S + st->off_s[6]; a.bias8 = st->bias8;
      a.dst = dst; a.DY = SY; a.DX = SX; a.cz0 = c0;
      a.CZ = CZ; a.CY = CY; a.CX = CX;
      a.VZ = std::min<int64_t>(fz_hi, VZ); a.VY = VY; a.VX = VX;
      dim3 grid((unsigned)ceil_div64(CX, 16), (unsigned)ceil_div64(CY, 4),
                (unsigned)ceil_div64(CZ, 4));
      TimedLaunch tl(ctx, "vgg_head_bf16");
      vgg_head_bf16<<<grid, 256, H_SMEM, stream>>>(a);
    }
    FPL_HIP(ctx, hipGetLastError());
  }
  *handled = true;
  return 0;
}
