// Fused bf16 MFMA kernels for vgg_like inference (flypylib/fplmodels.py:102-136)
// over a whole Z-slab of the volume - three launches per slab chunk:
//
//   vgg_stem_pool_bf16  u8/f32 volume -> normalise -> conv3 1->48 +BN+ReLU ->
//                       conv1 48->48 +BN+ReLU -> maxpool2            -> P1 (bf16)
//   vgg_mid_pool_bf16   P1 -> conv3 48->48 +BN+ReLU -> conv1 48->48 +BN+ReLU ->
//                       maxpool2                                     -> P2 (bf16)
//   vgg_c5_tail_bf16    P2 -> conv3 48->48 +BN+ReLU -> conv1 48->96 -> conv1 96->96 ->
//                       conv1 96->1 +bias -> sigmoid -> x4 nearest upsample, stored
//                       straight into the (Z,Y,X) f32 prediction volume
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
#include <cmath>

#include "fast_paths.h"
#include "mfma_util.h"
#include "pack_weights.h"
#include "vgg_tiles.h"

namespace {

constexpr int CH = 48;                 // channels of P1 / P2
constexpr int VOX_BYTES = CH * 2;      // 96 B per voxel (bf16)
constexpr int KSTEPS = 41;             // 27*48 = 1296 -> 40.5 K-steps of 32

// -------------------------------------------------------------------------------
// K1: stem.  WG = 4 waves; pooled block 4 x 8 x 32; wave task = 16 pooled x of one
// (pz,py) row; 8 sub-steps walk the 2x2x2 pooling window so the pool is an
// element-wise max over accumulators (no cross-lane traffic).
// -------------------------------------------------------------------------------
#ifndef STEM_WPS
#define STEM_WPS 2
#endif
constexpr int S_PZ = 4, S_PY = 8, S_PX = 32;
constexpr int S_TZ = 2 * S_PZ + 2, S_TY = 2 * S_PY + 2, S_TX = 2 * S_PX + 2;
// Row pitch of the LDS tile in elements.  The gather reads of a half-wave (lane groups
// g = 0,1 and g = 2,3) hit rows that are 2 rows, 1 row +- 1 plane and 1 plane apart:
// with a pitch of 40 dwords and 18 rows per plane those are 16, 24 and 16 banks apart,
// so the two 16-lane groups never share a bank (at the natural pitch of 33 dwords
// three of the five reads were 2-way conflicted: 38 % of the LDS cycles).
constexpr int S_TP = 80;
static_assert(S_TP >= S_TX && (2 * (S_TP / 2)) % 64 >= 16 && (S_TY * (S_TP / 2)) % 64 == 16,
              "stem tile pitch: bank spread of the gather");

struct StemArgs {
  const void *src;
  int64_t SZ, SY, SX;      // volume dims
  int64_t z_hi;            // rows >= z_hi are not needed (and may not be resident)
  float mean, sd;
  int64_t p1z0;            // global P1 row of chunk-local row 0
  const h16x8 *w1, *w2;   // fragments [s][b][lane]
  const float *shift1, *shift2;
  h16_t *p1;
  int P1Z, P1Y, P1X;       // chunk-local dims
  int nbx, nby, nbz;       // blocks of S_PX x S_PY x S_PZ pooled voxels
  float in_scale;          // power of two applied to the normalised input (1 = none)
};

// element offset of tap row `row` = (tz,ty) inside the input tile
__device__ __forceinline__ int stem_row_off(int row) {
  return ((row / 3) * S_TY + row % 3) * S_TP;
}

// The kernel is persistent (two workgroups per CU walk the blocks q, q + grid, ...) and
// software-pipelined: while a wave computes the 16 tasks of block i from tile[cur] it
// also loads, normalises and stores its 45 rows of block i+1's input tile into
// tile[cur ^ 1], three rows per task, the global loads one task ahead of their use.
// Measured on the one-block-per-workgroup form: the fill (3.2 ms at 1024^3) and the
// MFMA phase (6.0 ms) simply added up - the two workgroups of a CU start together and
// stay in lockstep, so "one fills while the other computes" never happened.  Spread over
// the task loop the fill is ~1 LDS read + 1 LDS write + 3 VALU per 9 MFMAs.
constexpr int S_ROWS = S_TZ * S_TY;            // 180 tile rows of 66 voxels
constexpr int S_WROWS = S_ROWS / 4;            // 45 per wave
constexpr int S_TASKS = S_PZ * S_PY * 2 / 4;   // 16 tasks per wave and block
constexpr int S_RPT = 3;                       // rows per task iteration
static_assert(S_ROWS % 4 == 0 && S_RPT * (S_TASKS - 1) == S_WROWS && S_WROWS <= 64,
              "stem fill schedule");

// The fill code is branch-free (clamped addresses, selects, whole-wave stores): a
// conditional load or store is a basic-block boundary, and the task body must stay one
// block for the MFMAs to be scheduled across the fill instructions.
template <typename SRC>
struct StemRows {
  SRC v[S_RPT];
  bool ok[S_RPT];          // row inside the volume (uniform)
};

struct StemBlock {
  int px0, py0, pz0;       // pooled origin
  int64_t gz0, gy0, gx0;   // origin of its input tile in the volume
  unsigned xc;             // this lane's column 0..63 of the tile, clamped into the volume
  bool x_ok;
};

__device__ __forceinline__ StemBlock stem_block(const StemArgs &a, int q, int lane) {
  StemBlock b;
  const int xb = q % a.nbx, t = q / a.nbx;
  const int yb = t % a.nby, zb = t / a.nby;
  b.px0 = xb * S_PX; b.py0 = yb * S_PY; b.pz0 = zb * S_PZ;
  b.gz0 = 2 * (a.p1z0 + b.pz0); b.gy0 = 2 * (int64_t)b.py0; b.gx0 = 2 * (int64_t)b.px0;
  const int64_t x = b.gx0 + lane;
  b.x_ok = x < a.SX;
  b.xc = (unsigned)(b.x_ok ? x : a.SX - 1);
  return b;
}

// uniform row pointer (clamped into the volume) + whether the row exists
template <typename SRC>
__device__ __forceinline__ const SRC *stem_row_ptr(const StemArgs &a, const StemBlock &b, int row,
                                                   bool &ok) {
  const int64_t z = b.gz0 + row / S_TY, y = b.gy0 + row % S_TY;
  ok = z < a.z_hi && y < a.SY;
  const int64_t zc = z < a.z_hi ? z : a.z_hi - 1, yc = y < a.SY ? y : a.SY - 1;
  return (const SRC *)a.src + (zc * a.SY + yc) * a.SX;
}

// issue the loads of columns 0..63 of tile rows [row0, row0 + S_RPT)
template <typename SRC>
__device__ __forceinline__ void stem_load_rows(const StemArgs &a, const StemBlock &b, int row0,
                                               StemRows<SRC> &r) {
#pragma unroll
  for (int k = 0; k < S_RPT; ++k) {
    const int row = __builtin_amdgcn_readfirstlane(row0 + k);
    bool ok;
    const SRC *rp = stem_row_ptr<SRC>(a, b, row, ok);
    r.ok[k] = ok;
    r.v[k] = rp[b.xc];
  }
}

// Row addresses inside the task loop.  stem_row_ptr per row is ~35 scalar instructions
// (a division by 18, 64-bit products, clamps) - 105 of the ~460 instructions a wave
// issues per task.  Per block they are computed once, vectorised: lane l holds, for tile
// row wrow0 + l, the element offset of the (clamped) row from the block's (clamped)
// origin row, bit 31 = the row exists.  A task then needs one v_readlane per row.
template <typename SRC>
__device__ __forceinline__ unsigned stem_row_tab(const StemArgs &a, const StemBlock &b, int wrow0,
                                                 int lane, const SRC *&base) {
  const int row = wrow0 + (lane < S_WROWS ? lane : S_WROWS - 1);
  const int64_t z = b.gz0 + row / S_TY, y = b.gy0 + row % S_TY;
  const bool ok = z < a.z_hi && y < a.SY;
  const int64_t zc = z < a.z_hi ? z : a.z_hi - 1, yc = y < a.SY ? y : a.SY - 1;
  const int64_t zb = b.gz0 < a.z_hi ? b.gz0 : a.z_hi - 1, yb = b.gy0 < a.SY ? b.gy0 : a.SY - 1;
  base = (const SRC *)a.src + (zb * a.SY + yb) * a.SX;
  // <= S_TZ planes from the origin: fits 31 bits (checked at launch)
  const unsigned rel = (unsigned)(((zc - zb) * a.SY + (yc - yb)) * a.SX);
  return rel | (ok ? 0x80000000u : 0u);
}

// the loads of tile rows wrow0 + idx0 .. + S_RPT - 1 through the table
template <typename SRC>
__device__ __forceinline__ void stem_load_rows_tab(const SRC *base, unsigned tab, unsigned xc,
                                                   int idx0, StemRows<SRC> &r) {
#pragma unroll
  for (int k = 0; k < S_RPT; ++k) {
    const unsigned t = (unsigned)__builtin_amdgcn_readlane((int)tab, idx0 + k);
    r.ok[k] = (t >> 31) != 0u;
    r.v[k] = (base + (t & 0x7FFFFFFFu))[xc];
  }
}

// normalise (v - mean) / sd, round to 16 bits; zero past the volume end.  For u8
// sources the 256 possible values go through a per-WG lookup table: the divide +
// convert happen once per value, not once per voxel.
template <typename SRC>
__device__ __forceinline__ unsigned short stem_norm(const StemArgs &a, const unsigned short *lut,
                                                    SRC v, bool ok) {
  unsigned short t;
  if (sizeof(SRC) == 1) t = lut[(unsigned)v & 255u];
  else t = h16_bits(((float)v - a.mean) / a.sd * a.in_scale);
  return ok ? t : (unsigned short)0;
}

// conversion (LDS reads of the table) and store are separate steps so that the caller
// can put MFMA work between them
struct StemBits { unsigned short b[S_RPT]; };

template <typename SRC>
__device__ __forceinline__ void stem_convert_rows(const StemArgs &a, const StemBlock &b,
                                                  const unsigned short *lut,
                                                  const StemRows<SRC> &r, StemBits &o) {
#pragma unroll
  for (int k = 0; k < S_RPT; ++k) o.b[k] = stem_norm<SRC>(a, lut, r.v[k], r.ok[k] && b.x_ok);
}

__device__ __forceinline__ void stem_write_rows(unsigned short *tile, int row0, int lane,
                                                const StemBits &o) {
#pragma unroll
  for (int k = 0; k < S_RPT; ++k) tile[(row0 + k) * S_TP + lane] = o.b[k];
}

// columns 64 and 65 of the wave's 45 rows: lane l < 45 takes row wrow0 + l, once per block
template <typename SRC>
struct StemEdge { SRC v[2]; bool ok[2]; };

template <typename SRC>
__device__ __forceinline__ void stem_load_edge(const StemArgs &a, const StemBlock &b, int wrow0,
                                               int lane, StemEdge<SRC> &e) {
  const int row = wrow0 + (lane < S_WROWS ? lane : S_WROWS - 1);
  const int64_t z = b.gz0 + row / S_TY, y = b.gy0 + row % S_TY;
  const bool rok = z < a.z_hi && y < a.SY;
  const int64_t zc = z < a.z_hi ? z : a.z_hi - 1, yc = y < a.SY ? y : a.SY - 1;
  const SRC *rp = (const SRC *)a.src + (zc * a.SY + yc) * a.SX;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t x = b.gx0 + 64 + j;
    e.ok[j] = rok && x < a.SX;
    e.v[j] = rp[x < a.SX ? x : a.SX - 1];
  }
}

template <typename SRC>
__device__ __forceinline__ void stem_store_edge(const StemArgs &a, unsigned short *tile,
                                                const unsigned short *lut, int wrow0, int lane,
                                                const StemEdge<SRC> &e) {
  // lanes >= 45 repeat row 44's two stores (same address, same value)
  const int row = wrow0 + (lane < S_WROWS ? lane : S_WROWS - 1);
  const unsigned lo = stem_norm<SRC>(a, lut, e.v[0], e.ok[0]);
  const unsigned hi = stem_norm<SRC>(a, lut, e.v[1], e.ok[1]);
  *reinterpret_cast<unsigned *>(&tile[row * S_TP + 64]) = lo | (hi << 16);
}

// CLAMP01: conv3's activations are scaled below 1 (StemArgs w1 / shift1 / w2 are the
// scaled set, vgg_prepare), so ReLU + conversion is one instruction per pair
template <typename SRC, bool CLAMP01>
__global__ __launch_bounds__(256, STEM_WPS) void FPLK(vgg_stem_pool)(StemArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned short tiles[2][S_TZ * S_TY * S_TP];
  __shared__ unsigned short lut[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int nblocks = a.nbx * a.nby * a.nbz;
  int q = blockIdx.x;
  if (q >= nblocks) return;
  if (sizeof(SRC) == 1) lut[tid] = h16_bits(((float)tid - a.mean) / a.sd * a.in_scale);

  // ---- per-lane constants: byte offsets of the 3 pair reads and 2 single reads
  // for sub-step parity e = dx (k-slot layout: pack_weights.h::fpl_stem_slot_tap)
  int offP[2][3], offS[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      offP[e][i] = 2 * (g < 3 ? stem_row_off(3 * g + i) + (e == 0 ? 0 : 2)
                              : stem_row_off(6 + i) + (e == 0 ? 2 : 0));
#pragma unroll
    for (int h = 0; h < 2; ++h)
      // g = 3 (k-slots 30, 31: zero weights) mirrors g = 2: same address, no bank conflict
      offS[e][h] = 2 * (stem_row_off(2 * (g < 3 ? g : 2) + h) + (e == 0 ? 2 : 1));
  }
  h16x8 w1[2][3], w2[2][3];
  f32x4 sh1[3], sh2[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    w1[0][b] = a.w1[(0 * 3 + b) * 64 + lane];
    w1[1][b] = a.w1[(1 * 3 + b) * 64 + lane];
    w2[0][b] = a.w2[(0 * 3 + b) * 64 + lane];
    w2[1][b] = a.w2[(1 * 3 + b) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sh1[b][r] = a.shift1[16 * b + 4 * g + r];
      sh2[b][r] = a.shift2[16 * b + 4 * g + r];
    }
  }
  __syncthreads();                      // lut

  // ---- the first block's tile: plain fill, once per workgroup
  const int wrow0 = wave * S_WROWS, wrow1 = wrow0 + S_WROWS;
  StemBlock blk = stem_block(a, q, lane);
  for (int r0 = wrow0; r0 < wrow1; r0 += S_RPT) {
    StemRows<SRC> rr;
    StemBits hb;
    stem_load_rows<SRC>(a, blk, r0, rr);
    stem_convert_rows<SRC>(a, blk, lut, rr, hb);
    stem_write_rows(tiles[0], r0, lane, hb);
  }
  {
    StemEdge<SRC> ee;
    stem_load_edge<SRC>(a, blk, wrow0, lane, ee);
    stem_store_edge<SRC>(a, tiles[0], lut, wrow0, lane, ee);
  }
  __syncthreads();

  // fill pipeline state: rr = rows of group 0 of the next block, ee = its edge columns;
  // a block index past the end is clamped to the last block of this workgroup - its
  // fill then lands, unused, in the idle buffer (no branches in the task loop)
  const int G = (int)gridDim.x;
  auto clampq = [&](int qq, int qlast) { return qq < nblocks ? qq : qlast; };
  StemRows<SRC> rr;
  StemEdge<SRC> ee;
  {
    const StemBlock n0 = stem_block(a, clampq(q + G, q), lane);
    stem_load_rows<SRC>(a, n0, wrow0, rr);
    stem_load_edge<SRC>(a, n0, wrow0, lane, ee);
  }
  int cur = 0;
  for (;;) {
    const int qn = q + G;
    const bool has_next = qn < nblocks;                 // uniform
    const StemBlock nxt = stem_block(a, has_next ? qn : q, lane);
    const StemBlock nx2 = stem_block(a, clampq(qn + G, has_next ? qn : q), lane);
    const SRC *base_n, *base_2;
    const unsigned tab_n = stem_row_tab<SRC>(a, nxt, wrow0, lane, base_n);
    const unsigned tab_2 = stem_row_tab<SRC>(a, nx2, wrow0, lane, base_2);
    const unsigned char *tb = reinterpret_cast<const unsigned char *>(tiles[cur]);
    unsigned short *tnext = tiles[cur ^ 1];
#pragma unroll 1
    for (int ti = 0; ti < S_TASKS; ++ti) {
      // next block's tile, 16 groups of 3 rows (the 16th repeats row 44): group ti was
      // loaded one task ago; it is converted here (table reads) and written after the
      // first half of this task's MFMAs, where the loads of group ti + 1 - or, in the
      // last task, of group 0 of the block after next - are issued
      StemBits hb;
      stem_convert_rows<SRC>(a, nxt, lut, rr, hb);
      const int grow = wrow0 + (S_RPT * ti < S_WROWS ? S_RPT * ti : S_WROWS - S_RPT);
      const bool last = ti + 1 == S_TASKS;
      const SRC *lbase = last ? base_2 : base_n;
      const unsigned ltab = last ? tab_2 : tab_n, lxc = last ? nx2.xc : nxt.xc;
      int lidx = S_RPT * (ti + 1);                    // row index inside the wave's 45
      lidx = last ? 0 : (lidx < S_WROWS ? lidx : S_WROWS - S_RPT);
      const int task = wave + 4 * ti;
      const int row = task >> 1, xh = task & 1;
      const int pzl = row / S_PY, pyl = row % S_PY;
      const int base = 2 * (((2 * pzl) * S_TY + 2 * pyl) * S_TP + 2 * (16 * xh + c));
      // max-pool in fp32, two window positions per v_max3_f32 (rounding to bf16 is
      // monotonic, so rounding the fp32 max equals the max of the rounded values);
      // the initial 0 is the ReLU
      f32x4 poolf[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        f32x4 a2[2][3];
#pragma unroll
        for (int e = 0; e < 2; ++e) {                 // sub = 2 sp + e: x parity e
          const int so = 2 * ((((sp >> 1) & 1) * S_TY + (sp & 1)) * S_TP);
          u32x4 raw;
#pragma unroll
          for (int i = 0; i < 3; ++i)
            raw[i] = *reinterpret_cast<const unsigned *>(tb + base + so + offP[e][i]);
          const unsigned s0 = *reinterpret_cast<const unsigned short *>(tb + base + so + offS[e][0]);
          const unsigned s1 = *reinterpret_cast<const unsigned short *>(tb + base + so + offS[e][1]);
          raw[3] = s0 | (s1 << 16);
          const h16x8 bfrag = __builtin_bit_cast(h16x8, raw);
          f32x4 a1[3];
#pragma unroll
          for (int b = 0; b < 3; ++b) a1[b] = mfma16(w1[e][b], bfrag, sh1[b]);
          const h16x8 h0 = pack_relu_t<CLAMP01>(a1[0], a1[1]);
          const h16x8 h1 = pack_relu_lo_t<CLAMP01>(a1[2]);
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            a2[e][b] = mfma16(w2[0][b], h0, sh2[b]);
            a2[e][b] = mfma16(w2[1][b], h1, a2[e][b]);
          }

        }
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            poolf[b][r] = __builtin_fmaxf(__builtin_fmaxf(poolf[b][r], a2[0][b][r]), a2[1][b][r]);
        if (sp == 1) {
          stem_write_rows(tnext, grow, lane, hb);
          stem_load_rows_tab<SRC>(lbase, ltab, lxc, lidx, rr);
        }
      }
      u32x2 pooled[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        pooled[b][0] = cvt_pk_h16(poolf[b][0], poolf[b][1]);
        pooled[b][1] = cvt_pk_h16(poolf[b][2], poolf[b][3]);
      }
      const int pz = blk.pz0 + pzl, py = blk.py0 + pyl, px = blk.px0 + 16 * xh + c;
      if (pz < a.P1Z && py < a.P1Y && px < a.P1X) {
        h16_t *dst = a.p1 + (((int64_t)pz * a.P1Y + py) * a.P1X + px) * CH + 4 * g;
#pragma unroll
        for (int b = 0; b < 3; ++b) *reinterpret_cast<u32x2 *>(dst + 16 * b) = pooled[b];
      }
    }
    if (!has_next) break;
    stem_store_edge<SRC>(a, tnext, lut, wrow0, lane, ee);
    stem_load_edge<SRC>(a, nx2, wrow0, lane, ee);
    __syncthreads();        // tile[cur] consumed by every wave, tile[cur ^ 1] complete
    q = qn;
    blk = nxt;
    cur ^= 1;
  }
}

// -------------------------------------------------------------------------------
// Shared 3x3x3 48->48 implicit-GEMM K loop (K2 and K3).  The activation tile
// (TZ x TY x TX voxels x 96 B) is resident in LDS; the 123 KiB of weight
// fragments come from L2 into registers, per wave (see conv3_kloop).
// Lane (c,g) reads, per K-step, the 16 B of its voxel (+tap) that hold k-slots
// 8g..8g+7: flat k = 32s + 8g + j over (tap, channel) -> tap = k/48, ch = k%48.
// -------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// byte offset inside the activation tile of k-slot group (s, g)
template <int TY, int TX>
__device__ __forceinline__ unsigned kslot_offset(int s, int g) {
  const int f0 = 32 * s + 8 * g;
  const int tap = f0 / CH, ch0 = f0 % CH;
  if (tap >= 27) return 0u;           // zero weights; any valid address will do
  return (unsigned)((((tap / 9) * TY + (tap / 3) % 3) * TX + tap % 3) * VOX_BYTES +
                    ch0 * 2);
}

// k-slot offset table [g][KTAB]: entry s = tile offset of K-step s for lane group g
// (entries past the last step repeat it), read four at a time
constexpr int KTAB = (KSTEPS + 4) / 4 * 4;
constexpr int KTAB_BYTES = 4 * KTAB * 4;

template <int TY, int TX>
__device__ __forceinline__ unsigned kslot_entry(int idx) {
  const int g = idx / KTAB;
  int s = idx % KTAB;
  s = s < KSTEPS ? s : KSTEPS - 1;
  return kslot_offset<TY, TX>(s, g);
}

// The K loop.  Every wave takes the weight fragments (3 x 1 KiB per K-step, the
// same for all waves) straight from global memory - L2 / L1 hits - into registers,
// WQ K-steps ahead of their use: no LDS ring and no barrier inside the loop, so the
// waves of a workgroup drift apart and keep the MFMA pipe fed.  (An LDS ring staged
// through registers cost a barrier every three steps and was 6% slower; the 4x L2
// weight traffic, ~9 TB/s chip-wide, is well inside what L2 delivers.)
constexpr int WQ = 4;

template <int NSUB, bool DIAG = false, typename SubOff>
__device__ __forceinline__ void conv3_kloop(const unsigned char *tile,
                                            const unsigned *kofftab,
                                            const unsigned char *wglobal,
                                            unsigned vbase, SubOff sub_off,
                                            f32x4 (&acc)[NSUB][3], int tid,
                                            unsigned long long *t_ready = nullptr) {
  const int lane = tid & 63, g = lane >> 4;
  const unsigned char *wl = wglobal + lane * 16;
  h16x8 wq[WQ][3];
#pragma unroll
  for (int d = 0; d < WQ; ++d)
#pragma unroll
    for (int b = 0; b < 3; ++b)
      wq[d][b] = *reinterpret_cast<const h16x8 *>(wl + (size_t)(d * 3 + b) * 1024);
  __syncthreads();            // activation tile (LDS-DMA) + table visible
  if (DIAG) *t_ready = __builtin_amdgcn_s_memtime();
  // activation fragments run one K-step ahead of the MFMAs
  h16x8 bcur[NSUB], bnxt[NSUB];
  const unsigned *ktab = kofftab + g * KTAB;
  u32x4 kv = *reinterpret_cast<const u32x4 *>(ktab);
#pragma unroll
  for (int sub = 0; sub < NSUB; ++sub)
    bcur[sub] = *reinterpret_cast<const h16x8 *>(tile + vbase + kv[0] + sub_off(sub));
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    // prefetch K-step s+1 (the final prefetch re-reads the last step: harmless)
    if ((s + 1) % 4 == 0) kv = *reinterpret_cast<const u32x4 *>(ktab + s + 1);
    const unsigned koff = kv[(s + 1) % 4];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
      bnxt[sub] = *reinterpret_cast<const h16x8 *>(tile + vbase + koff + sub_off(sub));
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        acc[sub][b] = mfma16(wq[s % WQ][b], bcur[sub], acc[sub][b]);
    __builtin_amdgcn_s_setprio(0);
    if (s + WQ < KSTEPS) {
#pragma unroll
      for (int b = 0; b < 3; ++b)
        wq[s % WQ][b] =
            *reinterpret_cast<const h16x8 *>(wl + (size_t)((s + WQ) * 3 + b) * 1024);
    }
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) bcur[sub] = bnxt[sub];
  }
}

// -------------------------------------------------------------------------------
// K2: conv3 48->48 + conv1 48->48 + maxpool2.  WG = 4 waves, pre-pool block
// 4 x 4 x 16 (pooled 2 x 2 x 8); wave = one pooled (pz,py) row; lanes = 16
// consecutive pre-pool x; 4 sub-steps = the (dz,dy) pooling window positions, the
// x pair is pooled across adjacent lanes at the very end.  63 KiB of LDS per WG
// so two WGs share a CU: one fills its tile while the other computes.
// -------------------------------------------------------------------------------
constexpr int M_TZ = 6, M_TY = 6, M_TX = 18;
constexpr int M_TILE_BYTES = ((M_TZ * M_TY * M_TX * VOX_BYTES + 1023) / 1024) * 1024;
constexpr int M_SMEM = M_TILE_BYTES + KTAB_BYTES;
static_assert(2 * M_SMEM <= 160 * 1024, "two mid workgroups must fit one CU");

struct MidArgs {
  const h16_t *p1;
  int P1Z, P1Y, P1X;
  const unsigned char *w3;       // KSTEPS x 3 fragments
  const h16x8 *w4;              // [s][b][lane]
  const float *shift3, *shift4;
  h16_t *p2;
  int P2Z, P2Y, P2X;
  unsigned long long *dbg;       // diagnostic build only: 4 stamps per workgroup
  BlockGrid bg;
};

template <bool DIAG>
__global__ __launch_bounds__(256, 2) void FPLK(vgg_mid_pool)(MidArgs a) {
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  if (DIAG) t0 = __builtin_amdgcn_s_memtime();
  unsigned char *tile = smem;
  unsigned *kofftab = reinterpret_cast<unsigned *>(smem + M_TILE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  int xb, yb, zb;
  if (!block_coords(a.bg, xb, yb, zb)) return;
  const int px0 = xb * 8, py0 = yb * 2, pz0 = zb * 2;

  if (tid < 4 * KTAB) kofftab[tid] = kslot_entry<M_TY, M_TX>(tid);
  stage_tile<M_TZ, M_TY, M_TX>(a.p1, a.P1Z, a.P1Y, a.P1X, 2 * pz0, 2 * py0, 2 * px0,
                               tile, wave, lane);

  const int pzl = wave >> 1, pyl = wave & 1;
  const unsigned vbase =
      (unsigned)((((2 * pzl) * M_TY + 2 * pyl) * M_TX + c) * VOX_BYTES);
  f32x4 acc[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    f32x4 sh;
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[r] = a.shift3[16 * b + 4 * g + r];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) acc[sub][b] = sh;
  }
  auto sub_off = [](int sub) -> unsigned {
    return (unsigned)(((((sub >> 1) & 1) * M_TY + (sub & 1)) * M_TX) * VOX_BYTES);
  };
  conv3_kloop<4, DIAG>(tile, kofftab, a.w3, vbase, sub_off, acc, tid, &t1);
  if (DIAG) t2 = __builtin_amdgcn_s_memtime();

  // conv1 48->48 chained in registers, pooled over the 4 (dz,dy) window positions
  h16x8 w4[2][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    w4[0][b] = a.w4[(0 * 3 + b) * 64 + lane];
    w4[1][b] = a.w4[(1 * 3 + b) * 64 + lane];
  }
  f32x4 sh4[3];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) sh4[b][r] = a.shift4[16 * b + 4 * g + r];
  u32x2 pooled[3] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    const h16x8 h0 = pack_relu(acc[sub][0], acc[sub][1]);
    const h16x8 h1 = pack_relu_lo(acc[sub][2]);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      f32x4 a4 = mfma16(w4[0][b], h0, sh4[b]);
      a4 = mfma16(w4[1][b], h1, a4);
      pool_relu_h16(pooled[b], a4);
    }
  }
  // pool the x pair: lanes c and c^1 hold neighbouring pre-pool x
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    pooled[b][0] = pk_max_i16(pooled[b][0], (unsigned)__shfl_xor((int)pooled[b][0], 1));
    pooled[b][1] = pk_max_i16(pooled[b][1], (unsigned)__shfl_xor((int)pooled[b][1], 1));
  }
  const int pz = pz0 + pzl, py = py0 + pyl, px = px0 + (c >> 1);
  if ((c & 1) == 0 && pz < a.P2Z && py < a.P2Y && px < a.P2X) {
    h16_t *dst = a.p2 + (((int64_t)pz * a.P2Y + py) * a.P2X + px) * CH + 4 * g;
#pragma unroll
    for (int b = 0; b < 3; ++b) *reinterpret_cast<u32x2 *>(dst + 16 * b) = pooled[b];
  }
  if (DIAG && tid == 0) {
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    const size_t wg = blockIdx.x;
    a.dbg[4 * wg + 0] = t0; a.dbg[4 * wg + 1] = t1;
    a.dbg[4 * wg + 2] = t2; a.dbg[4 * wg + 3] = t3;
  }
}

// -------------------------------------------------------------------------------
// K3: conv3 48->48 + BN + ReLU on P2 and the 1x1 head.  Same structure as K2 (two
// workgroups per CU) without the pool: block 4(z) x 4(y) x 16(x), wave = z,
// sub-steps = the 4 y rows.
// -------------------------------------------------------------------------------
constexpr int H_TZ = 6, H_TY = 6, H_TX = 18;
constexpr int H_TILE_BYTES = ((H_TZ * H_TY * H_TX * VOX_BYTES + 1023) / 1024) * 1024;
constexpr int H_SMEM = H_TILE_BYTES + KTAB_BYTES;
static_assert(2 * H_SMEM <= 160 * 1024, "two c5 workgroups must fit one CU");

constexpr int H_W6 = 12, H_W7 = 18, H_W8 = 3;       // fragment counts of L6, L7, L8

struct C5TailArgs {
  const h16_t *p2;
  int P2Z, P2Y, P2X;
  const unsigned char *w5;       // KSTEPS x 3 fragments
  const float *shift5;
  int CZ, CY, CX;                // chunk-local coarse dims
  const unsigned char *wtail;    // L6 [2][6], L7 [3][6], L8 [3][1] fragments
  const float *shift6, *shift7;
  float bias8;
  float *dst;                    // (Z,Y,X) prediction volume, row 0
  int64_t DY, DX;                // its pitches
  int64_t gz0;                   // global coarse z of chunk-local coarse row 0
  int64_t VZ, VY, VX;            // valid fine extents (dim - 2 * off)
  int off;                       // rf offset of the network: 7 (vgg_like), 10 (vgg_like2)
  BlockGrid bg;
};

// conv3 48->48 +BN+ReLU on P2, then - in registers, on the wave's four sub-steps in
// lockstep so that every weight fragment is loaded once per wave - conv1 48->96,
// conv1 96->96, conv1 96->1 + bias, sigmoid, and the x4 nearest upsample stored
// straight into the (Z,Y,X) f32 prediction volume.  The accumulators of one layer are
// the B fragments of the next (k-slot (s,g,j) = channel 16(2s + (j>>2)) + 4g + (j&3)),
// so nothing is exchanged between lanes until the logit (lane (c, g=0), register 0).
__global__ __launch_bounds__(256, 2) void FPLK(vgg_c5_tail)(C5TailArgs a) {
  unsigned char *tile = smem;
  unsigned *kofftab = reinterpret_cast<unsigned *>(smem + H_TILE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  int xb, yb, zb;
  if (!block_coords(a.bg, xb, yb, zb)) return;
  const int cx0 = xb * 16, cy0 = yb * 4, cz0 = zb * 4;

  if (tid < 4 * KTAB) kofftab[tid] = kslot_entry<H_TY, H_TX>(tid);
  stage_tile<H_TZ, H_TY, H_TX>(a.p2, a.P2Z, a.P2Y, a.P2X, cz0, cy0, cx0, tile, wave, lane);

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
  conv3_kloop<4>(tile, kofftab, a.w5, vbase, sub_off, acc, tid);

  const unsigned char *wt = a.wtail + lane * 16;
  auto wfrag = [&](int f) { return *reinterpret_cast<const h16x8 *>(wt + (size_t)f * 1024); };
  // C5 = ReLU(acc) as B fragments
  h16x8 h5[4][2];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    h5[sub][0] = pack_relu(acc[sub][0], acc[sub][1]);
    h5[sub][1] = pack_relu_lo(acc[sub][2]);
  }
  h16x8 h6[4][3];
  {
    f32x4 a6[4][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      f32x4 sh;
#pragma unroll
      for (int r = 0; r < 4; ++r) sh[r] = a.shift6[16 * b + 4 * g + r];
      const h16x8 w0 = wfrag(0 * 6 + b), w1 = wfrag(1 * 6 + b);
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
        a6[sub][b] = mfma16(w1, h5[sub][1], mfma16(w0, h5[sub][0], sh));
    }
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int s = 0; s < 3; ++s) h6[sub][s] = pack_relu(a6[sub][2 * s], a6[sub][2 * s + 1]);
  }
  h16x8 h7[4][3];
  {
    f32x4 a7[4][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      f32x4 sh;
#pragma unroll
      for (int r = 0; r < 4; ++r) sh[r] = a.shift7[16 * b + 4 * g + r];
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) a7[sub][b] = sh;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const h16x8 w = wfrag(H_W6 + s * 6 + b);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) a7[sub][b] = mfma16(w, h6[sub][s], a7[sub][b]);
      }
    }
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int s = 0; s < 3; ++s) h7[sub][s] = pack_relu(a7[sub][2 * s], a7[sub][2 * s + 1]);
  }
  f32x4 a8[4];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) a8[sub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const h16x8 w = wfrag(H_W6 + H_W7 + s);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) a8[sub] = mfma16(w, h7[sub][s], a8[sub]);
  }

  const int cz = cz0 + wave, cx = cx0 + c;
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    const int cy = cy0 + sub;
    // lane (c, g=0) register 0 holds the logit of coarse voxel c
    const float logit = __shfl(a8[sub][0], c) + a.bias8;
    const float p = 1.f / (1.f + __expf(-logit));
    // x4 upsample store: lane (c,g) writes 4 fine x of fine row (4cy+g), 4 z rows
    if (cz < a.CZ && cy < a.CY && cx < a.CX) {
      const int64_t fz0 = 4 * (a.gz0 + cz), fy = 4 * (int64_t)cy + g, fx0 = 4 * (int64_t)cx;
      if (fy < a.VY && fx0 < a.VX) {
        const int nx = (int)(a.VX - fx0 < 4 ? a.VX - fx0 : 4);
#pragma unroll
        for (int dz = 0; dz < 4; ++dz) {
          const int64_t fz = fz0 + dz;
          if (fz >= a.VZ) break;
          float *o = a.dst + ((fz + a.off) * a.DY + fy + a.off) * a.DX + fx0 + a.off;
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
// vgg_like2 (flypylib/fplmodels.py:138-172): every second conv is 3x3x3, so the stack is
//   [conv3 1->48, conv3 48->48, pool] [conv3 48->48, conv3 48->48, pool] conv3 48->48, head
// and the 1x1 chaining of vgg_like does not apply.  One kernel template covers the four
// 48->48 convolutions (the fifth is vgg_c5_tail): block 4 x 4 x 16 outputs, the K loop and
// tile of K2 / K3, and
//   STEM  the 6 x 6 x 18 x 48 input tile is not read but COMPUTED: conv3 1->48 + BN + ReLU
//         of the raw 8 x 8 x 20 (normalised, zero past the volume end) input tile, 41
//         groups of 16 tile voxels x 3 MFMAs (27 taps = one K-step) - the full-resolution
//         48-channel tensor (96 B per voxel) never exists in HBM;
//   POOL  ReLU + 2x2x2 max pool in the epilogue (wave = pooled (z,y) row as in K2),
//         otherwise ReLU and a plain channels-last store (wave = z as in K3).
// -------------------------------------------------------------------------------
constexpr int V2_RZ = M_TZ + 2, V2_RY = M_TY + 2, V2_RX = M_TX + 2;     // raw tile 8 x 8 x 20
constexpr int V2_NRAW = V2_RZ * V2_RY * V2_RX;
static_assert(V2_NRAW % 256 == 0, "raw tile pieces per thread");
constexpr int V2_SMEM = M_TILE_BYTES + KTAB_BYTES + V2_NRAW * 2 + 256 * 2;
static_assert(2 * V2_SMEM <= 160 * 1024, "two vgg_like2 workgroups must fit one CU");

struct V2Args {
  // STEM: the raw volume
  const void *src;
  int64_t SZ, SY, SX;
  float mean, sd;
  int64_t gz0;                   // raw z of chunk-local conv row 0
  const h16x8 *wstem;            // 3 fragments, k-slot (g,j) = tap 8g + j
  const float *shstem;
  // otherwise: a 48-channel tensor
  const h16_t *in;
  int IZ, IY, IX;
  const unsigned char *w;        // KSTEPS x 3 fragments
  const float *shift;
  h16_t *out;
  int OZ, OY, OX;                // output dims (pooled dims with POOL)
  BlockGrid bg;
};

template <bool STEM, bool POOL, typename SRC>
__global__ __launch_bounds__(256, 2) void FPLK(vgg2_conv3)(V2Args a) {
  unsigned char *tile = smem;
  unsigned *kofftab = reinterpret_cast<unsigned *>(smem + M_TILE_BYTES);
  unsigned short *rawt = reinterpret_cast<unsigned short *>(smem + M_TILE_BYTES + KTAB_BYTES);
  unsigned short *lut = rawt + V2_NRAW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  int xb, yb, zb;
  if (!block_coords(a.bg, xb, yb, zb)) return;
  // origin of the block's 4 x 4 x 16 conv outputs (= of its input tile)
  const int x0 = xb * 16, y0 = yb * 4, z0 = zb * 4;

  if (tid < 4 * KTAB) kofftab[tid] = kslot_entry<M_TY, M_TX>(tid);
  if (STEM) {
    const SRC *src = (const SRC *)a.src;
    if (sizeof(SRC) == 1) {
      lut[tid] = h16_bits(((float)tid - a.mean) / a.sd);
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < V2_NRAW / 256; ++j) {
      const int p = tid + 256 * j;
      const int64_t z = a.gz0 + z0 + p / (V2_RY * V2_RX), y = y0 + (p / V2_RX) % V2_RY, x = x0 + p % V2_RX;
      unsigned short b = 0;                              // zero past the volume end
      if (z < a.SZ && y < a.SY && x < a.SX) {
        const SRC v = src[(z * a.SY + y) * a.SX + x];
        b = sizeof(SRC) == 1 ? lut[(int)v] : h16_bits(((float)v - a.mean) / a.sd);
      }
      rawt[p] = b;
    }
    int toff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 8 * g + j;
      toff[j] = t < 27 ? ((t / 9) * V2_RY + (t / 3) % 3) * V2_RX + t % 3 : 0;
    }
    h16x8 w1[3];
    f32x4 sh1[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      w1[b] = a.wstem[b * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) sh1[b][r] = a.shstem[16 * b + 4 * g + r];
    }
    __syncthreads();                                    // raw tile visible
    constexpr int NVOX = M_TZ * M_TY * M_TX, NGRP = (NVOX + 15) / 16;
    for (int grp = wave; grp < NGRP; grp += 4) {
      const int v = 16 * grp + c, vv = v < NVOX ? v : NVOX - 1;
      const int tz = vv / (M_TY * M_TX), ty = (vv / M_TX) % M_TY, tx = vv % M_TX;
      const int ro = (tz * V2_RY + ty) * V2_RX + tx;
      u16x8 rw;
#pragma unroll
      for (int j = 0; j < 8; ++j) rw[j] = rawt[ro + toff[j]];
      const h16x8 bf = __builtin_bit_cast(h16x8, rw);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const f32x4 a1 = mfma16(w1[b], bf, sh1[b]);
        u32x2 o;
        o[0] = pk_max_i16(cvt_pk_h16(a1[0], a1[1]), 0u);
        o[1] = pk_max_i16(cvt_pk_h16(a1[2], a1[3]), 0u);
        if (v < NVOX) *reinterpret_cast<u32x2 *>(tile + v * VOX_BYTES + (16 * b + 4 * g) * 2) = o;
      }
    }
  } else {
    stage_tile<M_TZ, M_TY, M_TX>(a.in, a.IZ, a.IY, a.IX, z0, y0, x0, tile, wave, lane);
  }

  // POOL: wave = pooled (z,y) row, sub-steps = the (dz,dy) window; else wave = z, subs = y
  const int pzl = wave >> 1, pyl = wave & 1;
  const unsigned vbase = POOL ? (unsigned)((((2 * pzl) * M_TY + 2 * pyl) * M_TX + c) * VOX_BYTES)
                              : (unsigned)(((wave * M_TY) * M_TX + c) * VOX_BYTES);
  f32x4 acc[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    f32x4 sh;
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[r] = a.shift[16 * b + 4 * g + r];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) acc[sub][b] = sh;
  }
  auto sub_off = [](int sub) -> unsigned {
    return POOL ? (unsigned)(((((sub >> 1) & 1) * M_TY + (sub & 1)) * M_TX) * VOX_BYTES)
                : (unsigned)(sub * M_TX * VOX_BYTES);
  };
  conv3_kloop<4>(tile, kofftab, a.w, vbase, sub_off, acc, tid);

  if (POOL) {
    u32x2 pooled[3] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int b = 0; b < 3; ++b) pool_relu_h16(pooled[b], acc[sub][b]);
#pragma unroll
    for (int b = 0; b < 3; ++b) {      // the x pair: lanes c and c^1
      pooled[b][0] = pk_max_i16(pooled[b][0], (unsigned)__shfl_xor((int)pooled[b][0], 1));
      pooled[b][1] = pk_max_i16(pooled[b][1], (unsigned)__shfl_xor((int)pooled[b][1], 1));
    }
    const int pz = zb * 2 + pzl, py = yb * 2 + pyl, px = xb * 8 + (c >> 1);
    if ((c & 1) == 0 && pz < a.OZ && py < a.OY && px < a.OX) {
      h16_t *dst = a.out + (((int64_t)pz * a.OY + py) * a.OX + px) * CH + 4 * g;
#pragma unroll
      for (int b = 0; b < 3; ++b) *reinterpret_cast<u32x2 *>(dst + 16 * b) = pooled[b];
    }
  } else {
    const int oz = z0 + wave, ox = x0 + c;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int oy = y0 + sub;
      if (oz < a.OZ && oy < a.OY && ox < a.OX) {
        h16_t *dst = a.out + (((int64_t)oz * a.OY + oy) * a.OX + ox) * CH + 4 * g;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          u32x2 o;
          o[0] = pk_max_i16(cvt_pk_h16(acc[sub][b][0], acc[sub][b][1]), 0u);
          o[1] = pk_max_i16(cvt_pk_h16(acc[sub][b][2], acc[sub][b][3]), 0u);
          *reinterpret_cast<u32x2 *>(dst + 16 * b) = o;
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
  // vgg_like, IEEE-half build: a second set of L1 / L2 fragments and shift1 with L1's
  // output channel c scaled by 2^-e[c] and L2's input channel c by 2^e[c], e[c] the
  // smallest exponent with  sum_taps |w| * STEM_XMAX + |shift| <= 2^e  - for inputs
  // |x| <= STEM_XMAX conv3's activations then lie below 1 and ReLU + conversion is ONE
  // v_cvt_pk_f16_f32 ... clamp (mfma_util.h).  Powers of two throughout: L2's sums are
  // the unscaled ones (half subnormals of the activations aside)
  bool have_scaled = false;
  size_t off_w1s = 0, off_w2s = 0, off_s1s = 0;
  float stem_in_scale = 1.f;
};

constexpr float STEM_XMAX = 8.f;   // |(v - mean) / sd| bound the scaled set is built for

void vgg_state_free(fpl_ctx *ctx, void *p) {
  VggFastState *s = (VggFastState *)p;
  if (s->frags) hipFree(s->frags);
  if (s->shifts) hipFree(s->shifts);
  delete s;
}

bool is_vgg_like(const fpl_program *prog) { return fpl_vgg_variant(prog) == 1; }
bool is_vgg_like2(const fpl_program *prog) { return fpl_vgg_variant(prog) == 2; }

int vgg_prepare(fpl_ctx *ctx, fpl_program *prog, VggFastState **out) {
  VggFastState *st = (VggFastState *)prog->fast_state_h16[FPL_H16_SLOT];
  if (!st) {
    st = new VggFastState();
    prog->fast_state_h16[FPL_H16_SLOT] = st;
    prog->fast_state_h16_free[FPL_H16_SLOT] = vgg_state_free;
  }
  *out = st;
  if (st->version == prog->arena_version) return 0;
  const bool v2 = is_vgg_like2(prog);
  static const int conv_ops[8] = {0, 1, 3, 4, 6, 7, 8, 9};
  static const int mblocks[8] = {3, 3, 3, 3, 3, 6, 6, 1};
  static const int ksteps1[8] = {1, 2, KSTEPS, 2, KSTEPS, 2, 3, 3};
  static const FplSlotMap maps1[8] = {SLOT_STEM, SLOT_CHAIN, SLOT_SPATIAL, SLOT_CHAIN,
                                      SLOT_SPATIAL, SLOT_CHAIN, SLOT_CHAIN, SLOT_CHAIN};
  // vgg_like2: L1 as one K-step of 27 taps (k-slot (g,j) = tap 8g + j), L2..L5 3x3x3
  static const int ksteps2[8] = {1, KSTEPS, KSTEPS, KSTEPS, KSTEPS, 2, 3, 3};
  static const FplSlotMap maps2[8] = {SLOT_SPATIAL, SLOT_SPATIAL, SLOT_SPATIAL, SLOT_SPATIAL,
                                      SLOT_SPATIAL, SLOT_CHAIN, SLOT_CHAIN, SLOT_CHAIN};
  const int *ksteps = v2 ? ksteps2 : ksteps1;
  const FplSlotMap *maps = v2 ? maps2 : maps1;
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
    if (l == 0 && !v2)
      fpl_pack_stem(A + op.w_off, scale.data(), op.cout, &f);
    else
      fpl_pack_frags(A + op.w_off, scale.data(), op.k * op.k * op.k, op.cin, op.cout,
                     mblocks[l], ksteps[l], maps[l], &f);
    st->off_w[l] = all.size() * sizeof(uint16_t);
    all.insert(all.end(), f.begin(), f.end());
    st->off_s[l] = shifts.size();
    shifts.insert(shifts.end(), A + op.shift_off, A + op.shift_off + op.cout);
    while (shifts.size() % 4) shifts.push_back(0.f);
  }
  st->have_scaled = false;
#ifdef FPL_F16
  if (!v2) {
    const fpl_op &o1 = prog->ops[conv_ops[0]], &o2 = prog->ops[conv_ops[1]];
    std::vector<float> sc1(A + o1.scale_off, A + o1.scale_off + o1.cout);
    std::vector<float> sh1(A + o1.shift_off, A + o1.shift_off + o1.cout);
    std::vector<float> up(o1.cout);
    std::vector<int> ex(o1.cout);
    bool ok = true;
    int kmax = 0;
    for (int c = 0; c < o1.cout; ++c) {
      double bound = std::fabs((double)sh1[c]);
      for (int t = 0; t < 27; ++t)
        bound += std::fabs((double)A[o1.w_off + (size_t)t * o1.cout + c] * sc1[c]) * STEM_XMAX;
      bound *= 1.0 + 1e-3;                       // fp32 accumulation, 16-bit weight rounding
      int e = 0;
      while (std::ldexp(1.0, e) < bound) ++e;
      ok = ok && e <= 10;                        // keeps w2 * 2^e and x * 2^-k inside the half range
      ex[c] = e;
      kmax = std::max(kmax, e);
    }
    // The 2^-e[c] goes on as (input * 2^-k) * (weight * 2^(k - e[c])) with k = max e: the
    // weights only grow (no half subnormals), the input's smallest step 1/|sd| * 2^-k
    // stays a normal half - every factor is a power of two, so the fp32 accumulators are
    // exactly 2^-e[c] times the unscaled ones.
    for (int c = 0; c < o1.cout; ++c) {
      sc1[c] = (float)std::ldexp((double)sc1[c], kmax - ex[c]);
      sh1[c] = (float)std::ldexp((double)sh1[c], -ex[c]);
      up[c] = (float)std::ldexp(1.0, ex[c]);
    }
    st->stem_in_scale = (float)std::ldexp(1.0, -kmax);
    if (ok) {
      std::vector<uint16_t> f;
      fpl_pack_stem(A + o1.w_off, sc1.data(), o1.cout, &f);
      st->off_w1s = all.size() * sizeof(uint16_t);
      all.insert(all.end(), f.begin(), f.end());
      // L2 with its input channels scaled up: W2[c][o] * up[c]
      std::vector<float> w2((size_t)o2.cin * o2.cout);
      for (int c = 0; c < o2.cin; ++c)
        for (int o = 0; o < o2.cout; ++o)
          w2[(size_t)c * o2.cout + o] = A[o2.w_off + (size_t)c * o2.cout + o] * up[c];
      std::vector<float> sc2(A + o2.scale_off, A + o2.scale_off + o2.cout);
      fpl_pack_frags(w2.data(), sc2.data(), 1, o2.cin, o2.cout, mblocks[1], ksteps[1], maps[1], &f);
      st->off_w2s = all.size() * sizeof(uint16_t);
      all.insert(all.end(), f.begin(), f.end());
      st->off_s1s = shifts.size();
      shifts.insert(shifts.end(), sh1.begin(), sh1.end());
      while (shifts.size() % 4) shifts.push_back(0.f);
      st->have_scaled = true;
      for (uint16_t h : f) st->have_scaled = st->have_scaled && (h & 0x7C00u) != 0x7C00u;
    }
  }
  for (uint16_t h : all)
    FPL_REQUIRE(ctx, (h & 0x7C00u) != 0x7C00u,
                "a folded weight exceeds the IEEE-half range (65504); use precision "
                "bf16 or f32 for this network");
#endif
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
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)FPLK(vgg_mid_pool)<false>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, M_SMEM));
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)FPLK(vgg_mid_pool)<true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, M_SMEM));
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)FPLK(vgg_c5_tail),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, H_SMEM));
  if (v2) {
    FPL_HIP(ctx, hipFuncSetAttribute((const void *)FPLK(vgg2_conv3)<true, true, uint8_t>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, V2_SMEM));
    FPL_HIP(ctx, hipFuncSetAttribute((const void *)FPLK(vgg2_conv3)<true, true, float>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, V2_SMEM));
    FPL_HIP(ctx, hipFuncSetAttribute((const void *)FPLK(vgg2_conv3)<false, false, uint8_t>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, V2_SMEM));
    FPL_HIP(ctx, hipFuncSetAttribute((const void *)FPLK(vgg2_conv3)<false, true, uint8_t>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, V2_SMEM));
  }
  st->version = prog->arena_version;
  return 0;
}

}  // namespace

bool FPLK(fpl_fast_path_available)(const fpl_program *prog, int precision,
                             const int32_t offset[3], const int32_t out_sz[3]) {
  if (precision != FPL_THIS_PREC) return false;
  const bool v1 = is_vgg_like(prog), v2 = !v1 && is_vgg_like2(prog);
  if (!v1 && !v2) return false;
  for (int a = 0; a < 3; ++a)
    if (offset[a] != (v1 ? 7 : 10) || out_sz[a] % 4 != 0) return false;
  return true;
}

namespace {

// vgg_like2 over the coarse rows of a slab: four launches per chunk
//   H1 = pool(conv3(conv3(volume)))   vgg2_conv3<STEM, POOL>   (half resolution)
//   L3 = conv3(H1)                    vgg2_conv3<>
//   Q  = pool(conv3(L3))              vgg2_conv3<POOL>         (quarter resolution)
//   prediction = head(conv3(Q))       vgg_c5_tail
// With out = 80 = 4 * 20 every reference tile's input origin is a multiple of the stride,
// so as for vgg_like the coarse grid is anchored at the volume origin: pred[10 + p] =
// O[p / 4], O[i] seeing input [4i, 4i + 24), zero (normalised) past the volume end.
int vgg2_infer(fpl_ctx *ctx, VggFastState *st, const void *src, int src_dtype, float mean, float sd,
               const int64_t dims[3], const std::vector<int32_t> origins[3],
               const int32_t out_sz[3], int32_t zb, int32_t ze, float *dst) {
  hipStream_t stream = ctx->stream;
  constexpr int OFF = 10;
  const int64_t SZ = dims[0], SY = dims[1], SX = dims[2];
  const int64_t VZ = SZ - 2 * OFF, VY = SY - 2 * OFF, VX = SX - 2 * OFF;
  if (VZ <= 0 || VY <= 0 || VX <= 0 || zb >= ze) return 0;
  const int64_t fz_lo = (int64_t)origins[0][zb] - OFF;
  const int64_t fz_hi = std::min<int64_t>((int64_t)origins[0][ze - 1] - OFF + out_sz[0], VZ);
  const int64_t cz_lo = fz_lo / 4, cz_hi = ceil_div64(fz_hi, 4);
  const int CY = (int)ceil_div64(VY, 4), CX = (int)ceil_div64(VX, 4);
  const int QY = CY + 2, QX = CX + 2, T3Y = 2 * QY + 2, T3X = 2 * QX + 2, HY = T3Y + 2, HX = T3X + 2;
  const int64_t h_row = (int64_t)HY * HX * VOX_BYTES, t_row = (int64_t)T3Y * T3X * VOX_BYTES;
  const char *budget_env = getenv("FPL_VGG_SCRATCH_MB");
  const int64_t budget = budget_env ? (int64_t)atoll(budget_env) << 20 : (int64_t)64 << 30;
  // H1 has 2 (cz + 2) + 4 rows, L3 two fewer
  int64_t cz_chunk = std::max<int64_t>(4, (budget / (h_row + t_row) - 8) / 2);
  cz_chunk = std::min<int64_t>(cz_chunk, cz_hi - cz_lo);
  cz_chunk = (cz_chunk + 3) / 4 * 4;
  DevTemp tmp(ctx);
  void *h1v, *l3v, *qv;
  FPL_TRY(tmp.alloc((size_t)(2 * cz_chunk + 8) * h_row, &h1v));
  FPL_TRY(tmp.alloc((size_t)(2 * cz_chunk + 6) * t_row, &l3v));
  FPL_TRY(tmp.alloc((size_t)(cz_chunk + 2) * QY * QX * VOX_BYTES, &qv));
  const unsigned char *F = st->frags;
  const float *S = st->shifts;
  for (int64_t c0 = cz_lo; c0 < cz_hi; c0 += cz_chunk) {
    const int CZ = (int)std::min<int64_t>(cz_chunk, cz_hi - c0);
    const int QZ = CZ + 2, T3Z = 2 * QZ + 2, HZ = T3Z + 2;
    {
      V2Args a = {};
      a.src = src; a.SZ = SZ; a.SY = SY; a.SX = SX; a.mean = mean; a.sd = sd;
      a.gz0 = 4 * c0;
      a.wstem = (const h16x8 *)(F + st->off_w[0]); a.shstem = S + st->off_s[0];
      a.w = F + st->off_w[1]; a.shift = S + st->off_s[1];
      a.out = (h16_t *)h1v; a.OZ = HZ; a.OY = HY; a.OX = HX;
      a.bg = BlockGrid{(int)ceil_div64(HX, 8), (int)ceil_div64(HY, 2), (int)ceil_div64(HZ, 2)};
      TimedLaunch tl(ctx, "vgg2_stem_conv3_pool_" FPL_PREC_STR);
      if (src_dtype == FPL_U8)
        FPLK(vgg2_conv3)<true, true, uint8_t><<<block_grid_size(a.bg), 256, V2_SMEM, stream>>>(a);
      else
        FPLK(vgg2_conv3)<true, true, float><<<block_grid_size(a.bg), 256, V2_SMEM, stream>>>(a);
    }
    {
      V2Args a = {};
      a.in = (const h16_t *)h1v; a.IZ = HZ; a.IY = HY; a.IX = HX;
      a.w = F + st->off_w[2]; a.shift = S + st->off_s[2];
      a.out = (h16_t *)l3v; a.OZ = T3Z; a.OY = T3Y; a.OX = T3X;
      a.bg = BlockGrid{(int)ceil_div64(T3X, 16), (int)ceil_div64(T3Y, 4), (int)ceil_div64(T3Z, 4)};
      TimedLaunch tl(ctx, "vgg2_conv3_" FPL_PREC_STR);
      FPLK(vgg2_conv3)<false, false, uint8_t><<<block_grid_size(a.bg), 256, V2_SMEM, stream>>>(a);
    }
    {
      V2Args a = {};
      a.in = (const h16_t *)l3v; a.IZ = T3Z; a.IY = T3Y; a.IX = T3X;
      a.w = F + st->off_w[3]; a.shift = S + st->off_s[3];
      a.out = (h16_t *)qv; a.OZ = QZ; a.OY = QY; a.OX = QX;
      a.bg = BlockGrid{(int)ceil_div64(QX, 8), (int)ceil_div64(QY, 2), (int)ceil_div64(QZ, 2)};
      TimedLaunch tl(ctx, "vgg2_conv3_pool_" FPL_PREC_STR);
      FPLK(vgg2_conv3)<false, true, uint8_t><<<block_grid_size(a.bg), 256, V2_SMEM, stream>>>(a);
    }
    {
      C5TailArgs a;
      a.p2 = (const h16_t *)qv; a.P2Z = QZ; a.P2Y = QY; a.P2X = QX;
      a.w5 = F + st->off_w[4]; a.shift5 = S + st->off_s[4];
      a.CZ = CZ; a.CY = CY; a.CX = CX;
      a.wtail = F + st->off_w[5];
      a.shift6 = S + st->off_s[5]; a.shift7 = S + st->off_s[6]; a.bias8 = st->bias8;
      a.dst = dst; a.DY = SY; a.DX = SX; a.gz0 = c0;
      a.VZ = std::min<int64_t>(fz_hi, VZ); a.VY = VY; a.VX = VX; a.off = OFF;
      a.bg = BlockGrid{(int)ceil_div64(CX, 16), (int)ceil_div64(CY, 4), (int)ceil_div64(CZ, 4)};
      TimedLaunch tl(ctx, "vgg_c5_tail_" FPL_PREC_STR);
      FPLK(vgg_c5_tail)<<<block_grid_size(a.bg), 256, H_SMEM, stream>>>(a);
    }
    FPL_HIP(ctx, hipGetLastError());
  }
  return 0;
}

}  // namespace

int FPLK(fpl_fast_infer_volume)(fpl_ctx *ctx, fpl_program *prog, const void *src,
                          int src_dtype, float mean, float sd,
                          const int64_t dims[3], const int32_t tile_in[3],
                          const int32_t offset[3], int precision,
                          const std::vector<int32_t> origins[3],
                          const int32_t out_sz[3], int32_t zb, int32_t ze,
                          float *dst, bool *handled) {
  *handled = false;
  if (!FPLK(fpl_fast_path_available)(prog, precision, offset, out_sz)) return 0;
  VggFastState *st;
  FPL_TRY(vgg_prepare(ctx, prog, &st));
  if (is_vgg_like2(prog)) {
    FPL_TRY(vgg2_infer(ctx, st, src, src_dtype, mean, sd, dims, origins, out_sz, zb, ze, dst));
    *handled = true;
    return 0;
  }
  hipStream_t stream = ctx->stream;
  const int64_t SZ = dims[0], SY = dims[1], SX = dims[2];
  const int64_t VZ = SZ - 14, VY = SY - 14, VX = SX - 14;
  if (VZ <= 0 || VY <= 0 || VX <= 0 || zb >= ze) {   // no valid output voxel
    *handled = true;
    return 0;
  }
  // coarse rows this slab owns (tile rows zb..ze-1 of the reference lattice)
  const int64_t fz_lo = (int64_t)origins[0][zb] - 7;
  const int64_t fz_hi = std::min<int64_t>((int64_t)origins[0][ze - 1] - 7 + out_sz[0], VZ);
  const int64_t cz_lo = fz_lo / 4, cz_hi = ceil_div64(fz_hi, 4);
  const int CY = (int)ceil_div64(VY, 4), CX = (int)ceil_div64(VX, 4);
  const int P2Y = CY + 2, P2X = CX + 2, P1Y = 2 * P2Y + 2, P1X = 2 * P2X + 2;
  // chunk of coarse rows bounded by a scratch budget (P1 dominates)
  const int64_t p1_row_bytes = (int64_t)P1Y * P1X * VOX_BYTES;
  // (FPL_VGG_SCRATCH_MB shrinks it so that tests can force several chunks)
  const char *budget_env = getenv("FPL_VGG_SCRATCH_MB");
  const int64_t budget = budget_env ? (int64_t)atoll(budget_env) << 20 : (int64_t)48 << 30;
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
      // block rounding may reach past the rows this slab stages: they only
      // feed masked outputs, so they read as zero
      a.z_hi = std::min<int64_t>(SZ, 4 * (c0 + CZ) + 14);
      a.w1 = (const h16x8 *)(F + st->off_w[0]);
      a.w2 = (const h16x8 *)(F + st->off_w[1]);
      a.shift1 = S + st->off_s[0]; a.shift2 = S + st->off_s[1];
      // u8 input inside the bound the scaled set was built for: one-instruction ReLU
      const float xmax = std::max(std::fabs(0.f - mean), std::fabs(255.f - mean)) / std::fabs(sd);
      const bool clamp01 = st->have_scaled && src_dtype == FPL_U8 && xmax <= STEM_XMAX &&
                           !getenv("FPL_STEM_NOCLAMP");
      a.in_scale = 1.f;
      if (clamp01) {
        a.in_scale = st->stem_in_scale;
        a.w1 = (const h16x8 *)(F + st->off_w1s);
        a.w2 = (const h16x8 *)(F + st->off_w2s);
        a.shift1 = S + st->off_s1s;
      }
      a.p1 = (h16_t *)p1v; a.P1Z = P1Z; a.P1Y = P1Y; a.P1X = P1X;
      a.nbx = (int)ceil_div64(P1X, S_PX); a.nby = (int)ceil_div64(P1Y, S_PY);
      a.nbz = (int)ceil_div64(P1Z, S_PZ);
      FPL_REQUIRE(ctx, (int64_t)S_TZ * SY * SX < ((int64_t)1 << 31),
                  "vgg fused path: a %lld x %lld plane is too large for the stem's 31-bit row "
                  "offsets", (long long)SY, (long long)SX);
      // persistent: two workgroups per CU walk the blocks
      const unsigned grid = (unsigned)std::min<int64_t>((int64_t)a.nbx * a.nby * a.nbz,
                                                        (int64_t)ctx->n_cu * 2);
      TimedLaunch tl(ctx, "vgg_stem_pool_" FPL_PREC_STR);
      if (clamp01)
        FPLK(vgg_stem_pool)<uint8_t, true><<<grid, 256, 0, stream>>>(a);
      else if (src_dtype == FPL_U8)
        FPLK(vgg_stem_pool)<uint8_t, false><<<grid, 256, 0, stream>>>(a);
      else
        FPLK(vgg_stem_pool)<float, false><<<grid, 256, 0, stream>>>(a);
    }
    {
      MidArgs a;
      a.p1 = (const h16_t *)p1v; a.P1Z = P1Z; a.P1Y = P1Y; a.P1X = P1X;
      a.w3 = F + st->off_w[2];
      a.w4 = (const h16x8 *)(F + st->off_w[3]);
      a.shift3 = S + st->off_s[2]; a.shift4 = S + st->off_s[3];
      a.p2 = (h16_t *)p2v; a.P2Z = P2Z; a.P2Y = P2Y; a.P2X = P2X;
      a.bg = BlockGrid{(int)ceil_div64(P2X, 8), (int)ceil_div64(P2Y, 2), (int)ceil_div64(P2Z, 2)};
      const unsigned grid = block_grid_size(a.bg);
      a.dbg = nullptr;
      if (getenv("FPL_DIAG_MID")) {
        // diagnostic build: per-workgroup s_memtime stamps (never timed/shipped)
        const size_t nwg = grid;
        void *dbg;
        FPL_TRY(tmp.alloc(nwg * 32, &dbg));
        a.dbg = (unsigned long long *)dbg;
        FPL_HIP(ctx, hipMemsetAsync(dbg, 0, nwg * 32, stream));   // padding blocks leave zeros
        FPLK(vgg_mid_pool)<true><<<grid, 256, M_SMEM, stream>>>(a);
        std::vector<unsigned long long> h(nwg * 4);
        FPL_HIP(ctx, hipMemcpyAsync(h.data(), dbg, nwg * 32, hipMemcpyDeviceToHost, stream));
        FPL_HIP(ctx, hipStreamSynchronize(stream));
        double fill = 0, loop = 0, epi = 0, tot = 0;
        size_t real = 0;
        for (size_t i = 0; i < nwg; ++i) {
          if (!h[4 * i + 3]) continue;
          ++real;
          fill += (double)(h[4 * i + 1] - h[4 * i]);
          loop += (double)(h[4 * i + 2] - h[4 * i + 1]);
          epi += (double)(h[4 * i + 3] - h[4 * i + 2]);
          tot += (double)(h[4 * i + 3] - h[4 * i]);
        }
        fprintf(stderr, "[FPL_DIAG_MID] %zu WGs: mean cycles fill %.0f  kloop %.0f  epilogue %.0f  total %.0f\n",
                real, fill / real, loop / real, epi / real, tot / real);
      } else {
        TimedLaunch tl(ctx, "vgg_mid_pool_" FPL_PREC_STR);
        FPLK(vgg_mid_pool)<false><<<grid, 256, M_SMEM, stream>>>(a);
      }
    }
    {
      C5TailArgs a;
      a.p2 = (const h16_t *)p2v; a.P2Z = P2Z; a.P2Y = P2Y; a.P2X = P2X;
      a.w5 = F + st->off_w[4]; a.shift5 = S + st->off_s[4];
      a.CZ = CZ; a.CY = CY; a.CX = CX;
      a.wtail = F + st->off_w[5];      // L6, L7, L8 fragments are contiguous
      a.shift6 = S + st->off_s[5]; a.shift7 = S + st->off_s[6]; a.bias8 = st->bias8;
      a.dst = dst; a.DY = SY; a.DX = SX; a.gz0 = c0;
      a.VZ = std::min<int64_t>(fz_hi, VZ); a.VY = VY; a.VX = VX; a.off = 7;
      a.bg = BlockGrid{(int)ceil_div64(CX, 16), (int)ceil_div64(CY, 4), (int)ceil_div64(CZ, 4)};
      TimedLaunch tl(ctx, "vgg_c5_tail_" FPL_PREC_STR);
      FPLK(vgg_c5_tail)<<<block_grid_size(a.bg), 256, H_SMEM, stream>>>(a);
    }
    FPL_HIP(ctx, hipGetLastError());
  }
  *handled = true;
  return 0;
}
