// Split-operand MFMA kernels for vgg_like inference (flypylib/fplmodels.py:102-136):
// the fused structure of vgg_fused.hip - stem+pool -> P1, conv3+conv1+pool -> P2,
// conv3+head -> prediction volume - at fp32-grade accuracy (FPL_PREC_F16S).
//
// Why: a 16-bit operand type cannot hold the north star's gate on trained weights.  The
// error budget (tools/dev/error_budget.py, profiles/r03_error_budget_vgg.txt) shows the
// 17 rounding points of the plain IEEE-half path (input, 8 weight tensors, 7 activation
// tensors) each contribute 0.2 - 3e-4 to the worst probability: no subset of them can be
// spared.  Here every operand is carried as TWO halves,
//     v  ~  hi + lo,   hi = half(v),  lo = half(v - hi)          (~22 significant bits)
// and every product a * w as THREE MFMAs  a_hi w_hi + a_hi w_lo + a_lo w_hi  into the
// same fp32 accumulator (the fourth product is below fp32 resolution).  v - hi is exact in
// fp32; lo may be a subnormal half, which v_mfma_f32_16x16x32_f16 keeps
// (profiles/r03_mfma_f16_denorm.txt), so the representation error is <= 2^-22 |v| or
// 2^-25 absolute.  Observed: probabilities within 2e-6 of the fp32 oracle on the trained
// fixture, at 3x the MFMA work of the 16-bit path instead of the 16x of fp32 MFMAs.
//
// Two generations of the 3x3x3 48 -> 48 K loop live here.  vgg_like (round 4): P1 / P2 as planes
// of 8-channel passes and the all-LDS K loop of vgg_split_lds.h (one persistent 8-wave
// workgroup per CU, tile AND weight fragments double-buffered in LDS by LDS-DMA): vggs_mid_pool,
// vggs_c5_tail.  vgg_like2 (round 3, the constants below): per voxel NPASS = 2 passes of CHP =
// 24 channels, each pass [hi 24][lo 24] halves = 96 B, 192 B per voxel; the K loop runs once
// per pass on a 6 x 6 x 18 x 96 B tile (two 4-wave workgroups per CU, weight fragments per wave
// from L2), 21 K-steps per pass: vggs2_conv3, vggs_c5_tail_p24.
#define FPL_F16 1   // the operand halves are IEEE halves (mfma_util.h, pack_weights.h)
#include <algorithm>
#include <cmath>

#include "fast_paths.h"
#include "mfma_util.h"
#include "pack_weights.h"
#include <type_traits>
#include "vgg_split_lds.h"
#include "vgg_tiles.h"

namespace {

constexpr int CH = 48;
constexpr int CHP = 24, NPASS = CH / CHP;
constexpr int LO_OFF = CHP * 2;            // lo halves of a pass sit 48 B behind the hi halves
constexpr int PASS_BYTES = 2 * LO_OFF;     // 96
constexpr int VOX = NPASS * PASS_BYTES;    // 192 B per voxel in HBM, as NPASS planes [pass][z][y][x][96 B]:
                                           // a pass's tile rows are then contiguous (interleaved per voxel,
                                           // a row fill touched twice the cache lines it used)
constexpr int KS = (27 * CHP + 31) / 32;   // 21 K-steps per pass (27 * 24 = 648 = 20.25)
constexpr int KTAB = (KS + 4) / 4 * 4;     // table entries per lane group
constexpr int KTAB_BYTES = 4 * KTAB * 4;
static_assert(CHP % 8 == 0, "a lane's 8 k-slots stay inside one tap");

// byte offset of the hi halves of channels ch .. ch+3 (ch % 4 == 0) from a voxel's position
// in plane 0; `plane` = bytes of one pass plane of the tensor
__host__ __device__ constexpr int64_t chan_off(int ch, int64_t plane) { return (ch / CHP) * plane + (ch % CHP) * 2; }

// The third M-block of a 48-channel layer fills only half a K-step: its hi and lo halves
// share ONE fragment [hi | lo] (k-slots j < 4: hi, j >= 4: lo).  Against the weight step
// [w_lo | w_hi] it yields both cross products in one MFMA, against [w_hi | w_lo] the
// hi x hi product (and the lo x lo one, for free): 2 MFMAs where separate hi and lo
// fragments need 3 half-empty ones (pack_weights.h::fpl_pack_chain_step).
__device__ __forceinline__ h16x8 pack_relu_split_x(const f32x4 &blk) {
  u32x4 v;
  Pair2 p;
  p = split_pk_relu(blk[0], blk[1]); v[0] = p.hi; v[2] = p.lo;
  p = split_pk_relu(blk[2], blk[3]); v[1] = p.hi; v[3] = p.lo;
  return __builtin_bit_cast(h16x8, v);
}
__device__ __forceinline__ h16x8 pack_relu_split_x(const f32x4 &blk, unsigned &ovf) {   // half-range guard
  const h16x8 r = pack_relu_split_x(blk);
  const u32x4 v = __builtin_bit_cast(u32x4, r);
  ovf_note(ovf, pk_max_i16(v[0], v[1]));
  return r;
}

// A 1x1 conv on a 48-channel register-chained input, one M-block: w = the block's four
// weight steps [blocks 0,1 hi], [blocks 0,1 lo], [blk 2: w_lo | w_hi], [blk 2: w_hi | w_lo]
__device__ __forceinline__ f32x4 chain48(const h16x8 (&w)[4], const Frag2 &h01, h16x8 hx, f32x4 acc) {
  acc = mfma16(w[1], h01.hi, acc);
  acc = mfma16(w[0], h01.lo, acc);
  acc = mfma16(w[2], hx, acc);
  acc = mfma16(w[3], hx, acc);
  return mfma16(w[0], h01.hi, acc);
}

// The 1x1x1 convolutions that write P1 / P2 use interleaved rows (pack_weights.h,
// fpl_out_channel): lane (c, g) then holds the 12 CONTIGUOUS channels [12 g, 12 g + 12) of
// its voxel - half a pass - and writes 24 B of hi halves and 24 B of lo halves instead of
// three 8-B pieces each, 32 B apart (the stem's P1 stores cost 3.9 of its 21.5 ms).
__device__ __forceinline__ void store_split12(unsigned char *vox, int64_t plane, int g, const f32x4 (&v)[3],
                                              unsigned &ovf) {
  unsigned hi[6], lo[6];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const Pair2 p0 = split_pk(v[b][0], v[b][1], ovf), p1 = split_pk(v[b][2], v[b][3], ovf);
    hi[2 * b] = p0.hi; hi[2 * b + 1] = p1.hi;
    lo[2 * b] = p0.lo; lo[2 * b + 1] = p1.lo;
  }
  unsigned char *d = vox + (g >> 1) * plane + 24 * (g & 1);
  *reinterpret_cast<u32x4_a8 *>(d) = u32x4_a8{hi[0], hi[1], hi[2], hi[3]};
  *reinterpret_cast<u32x2 *>(d + 16) = u32x2{hi[4], hi[5]};
  *reinterpret_cast<u32x4_a8 *>(d + LO_OFF) = u32x4_a8{lo[0], lo[1], lo[2], lo[3]};
  *reinterpret_cast<u32x2 *>(d + LO_OFF + 16) = u32x2{lo[4], lo[5]};
}

// -------------------------------------------------------------------------------
// K1: stem.  ONE workgroup of 8 waves per CU (the hi and lo input tiles, double-buffered,
// are 113 KiB); pooled block 4 x 8 x 32 as in vgg_stem_pool; a wave takes 8 of the
// block's 64 tasks (16 pooled x of one (pz,py) row), 8 sub-steps per task walk the
// 2x2x2 pooling window.  Persistent and software-pipelined like the 16-bit stem: the
// next block's tile rows are loaded, split and stored inside the task loop.
//
// uint8 volumes (round 3): conv(w, (u - mean) / sd) = conv(w / sd, u - c0) + (c0 - mean) / sd
// * sum(w) with c0 = round(mean) clamped to [0, 255].  u - c0 is an integer of at most 8
// bits - EXACT in a half, no lo half - so conv3 1->48 is two MFMAs per M-block (w_hi a,
// w_lo a) on ONE gathered fragment, and the weights are re-split per (mean, sd) on the host
// (12 KiB).  The constant rides in the accumulators' initial value, which also takes care of
// the zero padding past the volume's end (normalised 0 - not an integer in this operand):
// padding voxels are stored as 0 and the initial value of an output whose window has
// (pz, py, px) planes of padding leaves those taps' share of the constant out - a 64-entry
// table of 48-channel vectors in LDS, indexed per lane and sub-step (entry 0 everywhere
// inside the volume), three 16-B LDS reads per sub-step and no branch.
// -------------------------------------------------------------------------------
constexpr int S_PZ = 4, S_PY = 8, S_PX = 32;
constexpr int S_TZ = 2 * S_PZ + 2, S_TY = 2 * S_PY + 2, S_TX = 2 * S_PX + 2;
constexpr int S_TP = 80;                       // row pitch (elements): bank spread of the gather
constexpr int S_TILE = S_TZ * S_TY * S_TP;     // elements of one (hi or lo) tile
constexpr int S_WAVES = 8;
constexpr int S_ROWS = S_TZ * S_TY;            // 180 tile rows of 66 voxels
constexpr int S_WROWS = (S_ROWS + S_WAVES - 1) / S_WAVES;   // 23 per wave (the last takes 19)
constexpr int S_TASKS = S_PZ * S_PY * 2 / S_WAVES;          // 8 tasks per wave and block
constexpr int S_RPT = 3;                       // rows per task iteration
static_assert(S_RPT * S_TASKS >= S_WROWS && S_WROWS <= 64, "stem fill schedule");
constexpr int S_SHTAB = 64 * 48;                // u8 path: floats of the initial-value table
constexpr int S_SH2 = 4 * 12;                   // conv1's shift as [g][b][r] floats (one 16-B read per M-block)
constexpr int S_SMEM = 4 * S_TILE * 2 + 256 * 4 + S_SHTAB * 4 + 2 * S_SH2 * 4;   // shift2 and -shift2

struct StemSArgs {
  const void *src;
  int64_t SZ, SY, SX;      // volume dims
  int64_t z_hi;            // rows >= z_hi are not needed (and may not be resident)
  float mean, sd;
  float c0;                // u8 path: the integer the operand is centred on
  const float *shtab;      // u8 path: [pz 4][py 4][px 4][48] initial accumulator values
  int64_t p1z0;            // global P1 row of chunk-local row 0
  const h16x8 *w1, *w2;    // fragments [part][e][b][lane]; chain48 steps [4][b][lane]
  const float *shift1, *shift2;
  x8::Tensor p1;           // chunk-local pool-1 tensor (planes of 8-channel passes, vgg_split_lds.h)
  int nbx, nby, nbz;       // blocks of S_PX x S_PY x S_PZ pooled voxels
  // half-range guard (mfma_util.h): conv3 1->48's outputs are bounded on the host from
  // sum |w| and the input limit (uint8: |u - c0| <= 255; float volumes: |x| <= xlim, checked
  // per loaded voxel), conv1's pooled outputs are checked where they are split for the store
  unsigned *flag;
  float xlim;
  unsigned char *dump;     // 64 B nobody reads: where lanes outside P1 store (x8::store12_sel)
};

__device__ __forceinline__ int stem_row_off(int row) {
  return ((row / 3) * S_TY + row % 3) * S_TP;
}

struct StemBlock {
  int px0, py0, pz0;       // pooled origin
  int64_t gz0, gy0, gx0;   // origin of its input tile in the volume
  unsigned xc;             // this lane's column 0..63 of the tile, clamped into the volume
  bool x_ok;
};

__device__ __forceinline__ StemBlock stem_block(const StemSArgs &a, int q, int lane) {
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

// tile row handled as entry `l` of wave-local row list starting at wrow0 (clamped: the
// last wave's list runs past the tile, the repeats rewrite row 179 with the same values)
__device__ __forceinline__ int stem_row_of(int wrow0, int l) {
  const int r = wrow0 + (l < S_WROWS ? l : S_WROWS - 1);
  return r < S_ROWS ? r : S_ROWS - 1;
}

// Row addresses inside the task loop: lane l holds, for the wave's l-th tile row, the
// element offset of the (clamped) row from the block's (clamped) origin row, bit 31 =
// the row exists.  A task then needs one v_readlane per row.
template <typename SRC>
__device__ __forceinline__ unsigned stem_row_tab(const StemSArgs &a, const StemBlock &b, int wrow0,
                                                 int lane, const SRC *&base) {
  const int row = stem_row_of(wrow0, lane);
  const int64_t z = b.gz0 + row / S_TY, y = b.gy0 + row % S_TY;
  const bool ok = z < a.z_hi && y < a.SY;
  const int64_t zc = z < a.z_hi ? z : a.z_hi - 1, yc = y < a.SY ? y : a.SY - 1;
  const int64_t zb = b.gz0 < a.z_hi ? b.gz0 : a.z_hi - 1, yb = b.gy0 < a.SY ? b.gy0 : a.SY - 1;
  base = (const SRC *)a.src + (zb * a.SY + yb) * a.SX;
  const unsigned rel = (unsigned)(((zc - zb) * a.SY + (yc - yb)) * a.SX);   // < 2^31: checked at launch
  return rel | (ok ? 0x80000000u : 0u);
}

template <typename SRC>
struct StemRows {
  SRC v[S_RPT];
  bool ok[S_RPT];          // row inside the volume (uniform)
};

// the loads of the wave's rows idx0 .. idx0 + S_RPT - 1 (list indices) through the table
template <typename SRC>
__device__ __forceinline__ void stem_load_rows(const SRC *base, unsigned tab, unsigned xc,
                                               int idx0, StemRows<SRC> &r) {
#pragma unroll
  for (int k = 0; k < S_RPT; ++k) {
    const int idx = idx0 + k < S_WROWS ? idx0 + k : S_WROWS - 1;
    const unsigned t = (unsigned)__builtin_amdgcn_readlane((int)tab, idx);
    r.ok[k] = (t >> 31) != 0u;
    r.v[k] = (base + (t & 0x7FFFFFFFu))[xc];
  }
}

// normalise (v - mean) / sd and split: hi half | lo half << 16; zero past the volume end.
// u8 sources go through a per-workgroup table of the 256 possible results.
template <typename SRC>
__device__ __forceinline__ unsigned stem_norm(const StemSArgs &a, const unsigned *lut, SRC v,
                                              bool ok, unsigned &xmax) {
  unsigned t;
  if (sizeof(SRC) == 1) {
    t = lut[(unsigned)v & 255u];              // u - c0: exact, lo half 0; padding 0
  } else {
    const float x = ((float)v - a.mean) / a.sd;
    // largest |x| seen, as an integer (a NaN is larger than every finite value)
    const unsigned ax = __builtin_bit_cast(unsigned, x) & 0x7FFFFFFFu;
    xmax = ax > xmax ? ax : xmax;
    const h16_t h = (h16_t)x;
    t = (unsigned)h16_bits(x) | ((unsigned)h16_bits(x - (float)h) << 16);
  }
  return ok ? t : 0u;
}

struct StemBits { unsigned b[S_RPT]; };

template <typename SRC>
__device__ __forceinline__ void stem_convert_rows(const StemSArgs &a, const StemBlock &b,
                                                  const unsigned *lut, const StemRows<SRC> &r,
                                                  StemBits &o, unsigned &xmax) {
#pragma unroll
  for (int k = 0; k < S_RPT; ++k) o.b[k] = stem_norm<SRC>(a, lut, r.v[k], r.ok[k] && b.x_ok, xmax);
}

// tile = hi tile, tile + S_TILE = lo tile (LO = false: u8 volumes, no lo tile)
template <bool LO>
__device__ __forceinline__ void stem_write_rows(unsigned short *tile, int wrow0, int idx0, int lane,
                                                const StemBits &o) {
#pragma unroll
  for (int k = 0; k < S_RPT; ++k) {
    const int e = stem_row_of(wrow0, idx0 + k) * S_TP + lane;
    tile[e] = (unsigned short)o.b[k];
    if (LO) tile[S_TILE + e] = (unsigned short)(o.b[k] >> 16);
  }
}

// columns 64 and 65 of the wave's rows: lane l takes list entry l, once per block
template <typename SRC>
struct StemEdge { SRC v[2]; bool ok[2]; };

template <typename SRC>
__device__ __forceinline__ void stem_load_edge(const StemSArgs &a, const StemBlock &b, int wrow0,
                                               int lane, StemEdge<SRC> &e) {
  const int row = stem_row_of(wrow0, lane);
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
__device__ __forceinline__ void stem_store_edge(const StemSArgs &a, unsigned short *tile,
                                                const unsigned *lut, int wrow0, int lane,
                                                const StemEdge<SRC> &e, unsigned &xmax) {
  const int row = stem_row_of(wrow0, lane);
  const unsigned v0 = stem_norm<SRC>(a, lut, e.v[0], e.ok[0], xmax);
  const unsigned v1 = stem_norm<SRC>(a, lut, e.v[1], e.ok[1], xmax);
  *reinterpret_cast<unsigned *>(&tile[row * S_TP + 64]) = (v0 & 0xFFFFu) | (v1 << 16);
  if (sizeof(SRC) != 1)
    *reinterpret_cast<unsigned *>(&tile[S_TILE + row * S_TP + 64]) = (v0 >> 16) | (v1 & 0xFFFF0000u);
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <typename SRC>
__global__ __launch_bounds__(64 * S_WAVES, 2) void vggs_stem_pool(StemSArgs a) {
  // tiles[buf][part]: buf 0 / 1, part hi / lo
  unsigned short *tiles = reinterpret_cast<unsigned short *>(smem);
  unsigned *lut = reinterpret_cast<unsigned *>(smem + 4 * S_TILE * 2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: task counters in SGPRs
  const int c = lane & 15, g = lane >> 4;
  const int nblocks = a.nbx * a.nby * a.nbz;
  int q = blockIdx.x;
  if (q >= nblocks) return;
  constexpr bool INT = sizeof(SRC) == 1;            // integer operand, no lo tile
  if (INT && tid < 256) lut[tid] = (unsigned)h16_bits((float)tid - a.c0);
  const unsigned char *shtab = smem + 4 * S_TILE * 2 + 256 * 4;
  if (INT)
    for (int i = tid; i < S_SHTAB; i += 64 * S_WAVES)
      reinterpret_cast<float *>(smem + 4 * S_TILE * 2 + 256 * 4)[i] = a.shtab[i];

  // per-lane byte offsets of the 3 pair reads and 2 single reads for sub-step parity
  // e = dx (k-slot layout: pack_weights.h::fpl_stem_slot_tap)
  int offP[2][3], offS[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      offP[e][i] = 2 * (g < 3 ? stem_row_off(3 * g + i) + (e == 0 ? 0 : 2)
                              : stem_row_off(6 + i) + (e == 0 ? 2 : 0));
#pragma unroll
    for (int h = 0; h < 2; ++h)
      offS[e][h] = 2 * (stem_row_off(2 * (g < 3 ? g : 2) + h) + (e == 0 ? 2 : 1));
  }
  // weight fragments in registers: [part][e or s][b]
  h16x8 w1[2][2][3], w2[4][3];
  // conv1's shift lives in LDS (12 registers fewer across the task loop: no spills)
  float *sh2t = reinterpret_cast<float *>(smem + 4 * S_TILE * 2 + 256 * 4 + S_SHTAB * 4);
  if (tid < S_SH2) {
    const float v = a.shift2[x8::out_channel((tid % 12) / 4, tid / 12, tid % 4)];
    sh2t[tid] = v;
    sh2t[S_SH2 + tid] = -v;
  }
  // conv1's shift is the same for the eight window positions of a pooled voxel, and
  // max_i(x_i + s) = max_i(x_i) + s with the same rounding: the accumulators start from 0, the
  // running maximum from -s (which is the ReLU: max(max_i x_i, -s) + s), and s is added once per
  // task in front of the store - two table reads per task where every sub-step made three
  const f32x4 *sh2g = reinterpret_cast<const f32x4 *>(sh2t + 12 * g);
  const f32x4 *sh2n = reinterpret_cast<const f32x4 *>(sh2t + S_SH2 + 12 * g);
  f32x4 sh1[3];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        w1[p][s][b] = a.w1[((p * 2 + s) * 3 + b) * 64 + lane];
        w2[2 * p + s][b] = a.w2[((2 * p + s) * 3 + b) * 64 + lane];
      }
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sh1[b][r] = a.shift1[16 * b + 4 * g + r];
    }
  __syncthreads();                      // lut

  unsigned ovf = 0u, xmax = 0u;         // half-range guard: split stores / float inputs
  // ---- the first block's tile: plain fill, once per workgroup
  const int wrow0 = wave * S_WROWS;
  StemBlock blk = stem_block(a, q, lane);
  {
    const SRC *base0;
    const unsigned tab0 = stem_row_tab<SRC>(a, blk, wrow0, lane, base0);
    for (int i0 = 0; i0 < S_WROWS; i0 += S_RPT) {
      StemRows<SRC> rr;
      StemBits hb;
      stem_load_rows<SRC>(base0, tab0, blk.xc, i0, rr);
      stem_convert_rows<SRC>(a, blk, lut, rr, hb, xmax);
      stem_write_rows<!INT>(tiles, wrow0, i0, lane, hb);
    }
    StemEdge<SRC> ee;
    stem_load_edge<SRC>(a, blk, wrow0, lane, ee);
    stem_store_edge<SRC>(a, tiles, lut, wrow0, lane, ee, xmax);
  }
  __syncthreads();

  // fill pipeline state: rr = row group 0 of the next block, ee = its edge columns; a
  // block index past the end is clamped to the last block of this workgroup - its fill
  // then lands, unused, in the idle buffer (no branches in the task loop)
  const int G = (int)gridDim.x;
  auto clampq = [&](int qq, int qlast) { return qq < nblocks ? qq : qlast; };
  StemRows<SRC> rr;
  StemEdge<SRC> ee;
  StemBits hb;                          // the next row group, converted: carried from task to task
  {
    const StemBlock n0 = stem_block(a, clampq(q + G, q), lane);
    const SRC *bn;
    const unsigned tn = stem_row_tab<SRC>(a, n0, wrow0, lane, bn);
    stem_load_rows<SRC>(bn, tn, n0.xc, 0, rr);
    stem_load_edge<SRC>(a, n0, wrow0, lane, ee);
    stem_convert_rows<SRC>(a, n0, lut, rr, hb, xmax);
    stem_load_rows<SRC>(bn, tn, n0.xc, S_RPT, rr);
    // four stores behind those loads, as every task of the loop below leaves them: the loop's
    // wait for `rr` is then vmcnt(4 + ...) on the entry path too (the counter retires in order)
    const f32x4 z3[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    __builtin_amdgcn_sched_barrier(0);
    x8::store12_sel(a.p1, 0, g, z3, ovf, false, a.dump);
  }
  // ---- the task loop, as ONE instruction stream laid out by hand ----------------------------
  // A SIMD issues one VALU-class instruction - MFMAs included - per ~4.4 cycles whatever its number
  // of waves; two ordinary VALU instructions per 16-cycle MFMA are free, every further one costs
  // its issue time, v_fma_mix*_f16 counts twice (tools/micro/mfma_valu_overlap.hip,
  // profiles/r04_micro_mfma_valu_overlap.txt).  A stage (one sub-step: x parity e of window
  // position sp) is 21 MFMAs next to ~58 such slots of ReLU, hi / lo conversion and pooling:
  // at the port's limit, and only if the two kinds alternate.  Left to the compiler (and to two
  // waves per SIMD running the same phases) they came in blocks: 19.8 ms, about the SUM of the
  // matrix pipe's 11.5 ms and the VALU's.  So the layers of a
  // sub-step are skewed by one stage - stage k issues conv3 of sub-step k + 1 and conv1 of
  // sub-step k, whose ReLU / split reads accumulators finished a stage ago - and the stage is
  // written as 15 slots of [one MFMA, one piece of VALU work], pinned by sched_barriers, then
  // a tail of six MFMAs under which the LDS reads of the next stage are issued (and, once a
  // task, the previous task's P1 store or the next tile's row conversion).  The skew runs
  // across tasks; only a block's first task primes it and its last one drains it.
  struct Gin {                          // the taps conv3 of one sub-step reads from the LDS tile
    unsigned p[INT ? 1 : 2][3], s0[INT ? 1 : 2], s1[INT ? 1 : 2];
  };
  struct Geo { int base, xrel, yrel, zrel, pzl, pyl, xh; };
  int cur = 0;
  for (;;) {
    const int qn = q + G;
    const bool has_next = qn < nblocks;                 // uniform
    const StemBlock nxt = stem_block(a, has_next ? qn : q, lane);
    const StemBlock nx2 = stem_block(a, clampq(qn + G, has_next ? qn : q), lane);
    const SRC *base_n, *base_2;
    const unsigned tab_n = stem_row_tab<SRC>(a, nxt, wrow0, lane, base_n);
    const unsigned tab_2 = stem_row_tab<SRC>(a, nx2, wrow0, lane, base_2);
    const unsigned char *tb = reinterpret_cast<const unsigned char *>(tiles + cur * 2 * S_TILE);
    unsigned short *tnext = tiles + (cur ^ 1) * 2 * S_TILE;
    auto geo = [&](int ti) {
      Geo t;
      const int task = wave + S_WAVES * ti;
      const int row = task >> 1;
      t.xh = task & 1; t.pzl = row / S_PY; t.pyl = row % S_PY;
      t.base = 2 * (((2 * t.pzl) * S_TY + 2 * t.pyl) * S_TP + 2 * (16 * t.xh + c));
      // INT: planes of padding in the 3 x 3 x 3 window of this lane's outputs (0 inside)
      t.xrel = (int)(blk.gx0 + 2 * (16 * t.xh + c) + 3 - a.SX);
      t.yrel = (int)(blk.gy0 + 2 * t.pyl + 3 - a.SY);
      t.zrel = (int)(blk.gz0 + 2 * t.pzl + 3 - a.z_hi);
      return t;
    };
    // uint8 volumes: a block whose windows stay clear of the padding past the volume (all but the
    // far faces' blocks) starts every conv3 accumulator from ONE vector per lane - entry 0 of the
    // table - so its pass is compiled without the three 16-B table reads per sub-step
    const bool interior = INT && blk.gx0 + 2 * S_PX + 2 <= a.SX && blk.gy0 + 2 * S_PY + 2 <= a.SY &&
                          blk.gz0 + 2 * S_PZ + 2 <= a.z_hi;
    auto block_pass = [&](auto edge_t) {
      constexpr bool EDGE = decltype(edge_t)::value;
      f32x4 kin[3];
      if (INT && !EDGE) {
#pragma unroll
        for (int b = 0; b < 3; ++b) kin[b] = *reinterpret_cast<const f32x4 *>(shtab + (4 * g) * 4 + 64 * b);
      }
      auto gather = [&](const Geo &t, int sub, Gin &o) {
        const int sp = sub >> 1, e = sub & 1;
        const int so = 2 * ((((sp >> 1) & 1) * S_TY + (sp & 1)) * S_TP);
#pragma unroll
        for (int part = 0; part < (INT ? 1 : 2); ++part) {
          const unsigned char *tp = tb + part * (S_TILE * 2) + t.base + so;
#pragma unroll
          for (int i = 0; i < 3; ++i) o.p[part][i] = *reinterpret_cast<const unsigned *>(tp + offP[e][i]);
          o.s0[part] = *reinterpret_cast<const unsigned short *>(tp + offS[e][0]);
          o.s1[part] = *reinterpret_cast<const unsigned short *>(tp + offS[e][1]);
        }
      };
      // (edge blocks of uint8 volumes) the sub-step's initial accumulator values
      auto gather_init = [&](const Geo &t, int sub, f32x4 (&o)[3]) {
        if (INT && EDGE) {
          const int sp = sub >> 1, e = sub & 1;
          const int pzy = 4 * min(max(t.zrel + (sp >> 1), 0), 3) + min(max(t.yrel + (sp & 1), 0), 3);
          const unsigned char *tp = shtab + ((4 * pzy + min(max(t.xrel + e, 0), 3)) * 48 + 4 * g) * 4;
#pragma unroll
          for (int b = 0; b < 3; ++b) o[b] = *reinterpret_cast<const f32x4 *>(tp + 64 * b);
        }
      };
      auto frag = [&](const Gin &gi, int part) {
        const u32x4 raw = {gi.p[part][0], gi.p[part][1], gi.p[part][2], gi.s0[part] | (gi.s1[part] << 16)};
        return __builtin_bit_cast(h16x8, raw);
      };
      constexpr int NC3 = INT ? 6 : 9;                    // conv3's MFMAs per sub-step
      // MFMA i of conv3, x parity e, on the gathered fragments
      auto c3 = [&](int i, int e, const f32x4 (&gin)[3], h16x8 bh, h16x8 bl, f32x4 (&o)[3]) {
        const int b = i % 3, grp = i / 3;
        if (grp == 0) o[b] = mfma16(w1[1][e][b], bh, INT ? (EDGE ? gin[b] : kin[b]) : sh1[b]);
        else if (!INT && grp == 1) o[b] = mfma16(w1[0][e][b], bl, o[b]);
        else o[b] = mfma16(w1[0][e][b], bh, o[b]);
      };
      // the state the skew carries from stage to stage (and task to task)
      // taps of the sub-steps conv3 does next: position q (8 x task + sub-step) in Gd[q % NGB].
      // uint8 volumes read them THREE positions ahead into two buffers (5 registers each: a
      // full stage between an LDS read and its use; one position less left every stage waiting
      // ~250 cycles on lgkmcnt), float volumes (10 registers a buffer) two ahead into one.
      constexpr int NGB = INT ? 2 : 1, DEPTH = NGB + 1;
      Gin Gd[NGB];
      f32x4 gin[3];                       // (edge blocks) initial values, one position ahead of their use
      f32x4 a1n[3];                       // conv3 of the sub-step whose conv1 comes next
      f32x4 a2[2][3];                     // conv1 of the x pair being pooled
      f32x4 poolf[3];                     // running maximum of the task, started from -shift2
#pragma unroll
      for (int b = 0; b < 3; ++b) poolf[b] = sh2n[b];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int b = 0; b < 3; ++b) a2[e][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      {                                   // prime: conv3 of (task 0, sub-step 0), reads of sub-step 1
        const Geo t0 = geo(0);
        gather(t0, 0, Gd[0]);
        gather_init(t0, 0, gin);
        const h16x8 bh = frag(Gd[0], 0), bl = INT ? bh : frag(Gd[0], INT ? 0 : 1);
#pragma unroll
        for (int i = 0; i < NC3; ++i) c3(i, 0, gin, bh, bl, a1n);
        gather_init(t0, 1, gin);
#pragma unroll
        for (int q = 1; q < DEPTH; ++q) gather(t0, q, Gd[q % NGB]);
      }
#pragma unroll 1
      for (int ti = 0; ti < S_TASKS; ++ti) {
        // The next block's tile arrives in S_TASKS groups of S_RPT rows, three tasks per group:
        // group ti + 2 is LOADED at the end of this task; group ti + 1 (loaded a task ago, `rr`)
        // is CONVERTED late in this task; group ti (`hb`) is WRITTEN to the idle tile buffer in
        // its middle.  Groups 8 and 9 are groups 0 and 1 of the block after next.  The
        // vector-memory counter retires in order: task ti - 1's P1 stores are issued (stage 0
        // of this task) BEHIND the loads of its end, so the conversion's wait for those loads
        // leaves the stores in flight.
        const bool conv2 = ti + 1 >= S_TASKS, load2 = ti + 2 >= S_TASKS;
        const SRC *lbase = load2 ? base_2 : base_n;
        const unsigned ltab = load2 ? tab_2 : tab_n, lxc = load2 ? nx2.xc : nxt.xc;
        const int lidx = S_RPT * ((ti + 2) & (S_TASKS - 1));
        const Geo tg = geo(ti), tn = geo((ti + 1) & (S_TASKS - 1)), tp = geo((ti + S_TASKS - 1) & (S_TASKS - 1));
#pragma unroll
        for (int sub = 0; sub < 8; ++sub) {
          const int sp = sub >> 1, e = sub & 1, en = e ^ 1;
          f32x4 a1[3];
#pragma unroll
          for (int b = 0; b < 3; ++b) a1[b] = a1n[b];
          const Gin &G = Gd[(sub + 1) % NGB];
          const h16x8 bh = frag(G, 0), bl = INT ? bh : frag(G, INT ? 0 : 1);
          float r[12];
          unsigned hi[6], lo[6];
          // VALU pieces: A = ReLU + hi conversion of value pair p, B = its lo halves,
          // PL = two max-pool updates from the finished x pair
          auto A = [&](int p) {
            r[2 * p] = relu_f32(a1[p >> 1][2 * (p & 1)]);
            r[2 * p + 1] = relu_f32(a1[p >> 1][2 * (p & 1) + 1]);
            hi[p] = cvt_pk_h16(r[2 * p], r[2 * p + 1]);
          };
          auto B = [&](int p) {                           // (mfma_util.h::split_pk: tied to r's register)
            lo[p] = split_lo_pk(r[2 * p], hi[p], r[2 * p + 1]);
          };
          auto PL = [&](int k) {
#pragma unroll
            for (int j = 2 * k; j < 2 * k + 2; ++j)
              poolf[j >> 2][j & 3] = __builtin_fmaxf(__builtin_fmaxf(poolf[j >> 2][j & 3], a2[0][j >> 2][j & 3]),
                                                     a2[1][j >> 2][j & 3]);
          };
          // MFMA i of conv1: hi x hi and lo x hi first (they need only the hi halves)
          auto c1 = [&](int i) {
            const int b = i % 3, grp = i / 3;
            const u32x4 h0h = {hi[0], hi[1], hi[2], hi[3]}, h0l = {lo[0], lo[1], lo[2], lo[3]};
            const u32x4 hxv = {hi[4], hi[5], lo[4], lo[5]};
            if (grp == 0) a2[e][b] = mfma16(w2[1][b], __builtin_bit_cast(h16x8, h0h), f32x4{0.f, 0.f, 0.f, 0.f});
            if (grp == 1) a2[e][b] = mfma16(w2[0][b], __builtin_bit_cast(h16x8, h0h), a2[e][b]);
            if (grp == 2) a2[e][b] = mfma16(w2[0][b], __builtin_bit_cast(h16x8, h0l), a2[e][b]);
            if (grp == 3) a2[e][b] = mfma16(w2[2][b], __builtin_bit_cast(h16x8, hxv), a2[e][b]);
            if (grp == 4) a2[e][b] = mfma16(w2[3][b], __builtin_bit_cast(h16x8, hxv), a2[e][b]);
          };
#pragma unroll
          for (int s = 0; s < 15; ++s) {
            if (s < 6) c3(s, en, gin, bh, bl, a1n);
            else c1(s - 6);
            if (s < 6 && e == 0) PL(s);                   // the pair finished a stage ago (sub 0: the last task's)
            if (s < 4) A(s);
            else if (s < 8) B(s - 4);
            else if (s < 10) A(s - 4);
            else if (s < 12) B(s - 6);
            __builtin_amdgcn_sched_barrier(0);
          }
          // ---- tail: LDS reads of the stage after next, the unpaired MFMAs, the once-a-task work
          if (sub + DEPTH < 8) gather(tg, sub + DEPTH, Gd[(sub + DEPTH) % NGB]);
          else gather(tn, sub + DEPTH - 8, Gd[(sub + DEPTH) % NGB]);
          if (sub + 2 < 8) gather_init(tg, sub + 2, gin);
          else gather_init(tn, sub + 2 - 8, gin);
#pragma unroll
          for (int i = 6; i < NC3; ++i) c3(i, en, gin, bh, bl, a1n);
#pragma unroll
          for (int i = 9; i < 15; ++i) c1(i);
          if (sub == 0) {
            // the previous task's pooled voxel (its last pair was pooled in the slots above)
            const int pz = blk.pz0 + tp.pzl, py = blk.py0 + tp.pyl, px = blk.px0 + 16 * tp.xh + c;
            f32x4 pv[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) pv[b] = poolf[b] + sh2g[b];
            x8::store12_sel(a.p1, a.p1.vox(pz, py, px), g, pv, ovf,
                            ti > 0 && pz < a.p1.Z && py < a.p1.Y && px < a.p1.X, a.dump);
#pragma unroll
            for (int b = 0; b < 3; ++b) poolf[b] = sh2n[b];                       // (the ReLU)
          }
          if (sub == 3) stem_write_rows<!INT>(tnext, wrow0, S_RPT * ti, lane, hb);
          if (sub == 5) {
            // (opaque to the optimiser here: otherwise the first instruction of the conversion -
            // and with it the wait for the loads - is hoisted to the top of the task)
#pragma unroll
            for (int k = 0; k < S_RPT; ++k) {
              unsigned t = sizeof(SRC) == 1 ? (unsigned)rr.v[k] : __builtin_bit_cast(unsigned, (float)rr.v[k]);
              asm volatile("" : "+v"(t));
              rr.v[k] = sizeof(SRC) == 1 ? (SRC)t : (SRC)__builtin_bit_cast(float, t);
            }
            stem_convert_rows<SRC>(a, conv2 ? nx2 : nxt, lut, rr, hb, xmax);
          }
          if (sub == 7) stem_load_rows<SRC>(lbase, ltab, lxc, lidx, rr);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      {                                   // drain: the last task's last pair and its store
        const Geo t7 = geo(S_TASKS - 1);
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            poolf[b][r] = __builtin_fmaxf(__builtin_fmaxf(poolf[b][r], a2[0][b][r]), a2[1][b][r]);
        const int pz = blk.pz0 + t7.pzl, py = blk.py0 + t7.pyl, px = blk.px0 + 16 * t7.xh + c;
        f32x4 pv[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) pv[b] = poolf[b] + sh2g[b];
        x8::store12_sel(a.p1, a.p1.vox(pz, py, px), g, pv, ovf,
                        pz < a.p1.Z && py < a.p1.Y && px < a.p1.X, a.dump);
      }
    };
    if (interior) block_pass(std::false_type{});
    else block_pass(std::true_type{});
    if (!has_next) break;
    stem_store_edge<SRC>(a, tnext, lut, wrow0, lane, ee, xmax);
    stem_load_edge<SRC>(a, nx2, wrow0, lane, ee);
    __syncthreads();        // tile[cur] consumed by every wave, tile[cur ^ 1] complete
    q = qn;
    blk = nxt;
    cur ^= 1;
  }
  ovf_commit(ovf, a.flag, FPL_RANGE_STEM);
  if (!INT && xmax > __builtin_bit_cast(unsigned, a.xlim)) atomicOr(a.flag, FPL_RANGE_INPUT);
}

// -------------------------------------------------------------------------------
// Shared 3x3x3 48->48 K loop (K2 and K3): per pass, the 6 x 6 x 18 x 96 B tile of that
// pass's 24 channels (hi and lo halves) is staged by LDS-DMA, then 21 K-steps of
// 4 sub-steps x 3 M-blocks x 3 products.  Weight fragments (6 x 1 KiB per K-step: hi
// and lo, the same for every wave) go from L2 straight into registers, WQ K-steps
// ahead, as in vgg_fused.hip::conv3_kloop.
// -------------------------------------------------------------------------------
template <int TY, int TX>
__device__ __forceinline__ unsigned kslot_entry(int idx) {
  const int g = idx / KTAB;
  int s = idx % KTAB;
  s = s < KS ? s : KS - 1;
  const int f0 = 32 * s + 8 * g;
  const int tap = f0 / CHP, ch0 = f0 % CHP;
  if (tap >= 27) return 0u;           // zero weights; any valid address will do
  return (unsigned)((((tap / 9) * TY + (tap / 3) % 3) * TX + tap % 3) * PASS_BYTES + ch0 * 2);
}

// K-steps of weight fragments in flight: a split K-step is 36 MFMAs (three times the 16-bit
// kernels'), so two steps ahead cover the same time as their six; measured mid 38.7 / 37.7 /
// 38.2 / 38.7 ms with 1 / 2 / 3 / 4 (the U-Net split kernels, 8 - 16 MFMAs per step, keep 3)
constexpr int WQ = 2;
// (Three passes of 16 channels - 41.5 KB tiles - were measured too: mid and tail unchanged at two
// workgroups per CU; three per CU need <= 168 VGPRs, which this loop only reaches by spilling:
// 43 ms.)

// `fill(pass)` puts the pass's 6 x 6 x 18 x 96 B tile into LDS (stage_tile, or vgg_like2's
// stem, which computes it); the barrier after it makes it visible.
template <typename Fill, typename SubOff>
__device__ __forceinline__ void conv3s_kloop_f(Fill fill, unsigned char *tile,
                                               const unsigned *kofftab, const unsigned char *wglobal,
                                               unsigned vbase, SubOff sub_off, f32x4 (&acc)[4][3],
                                               int tid) {
  const int lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const unsigned *ktab = kofftab + g * KTAB;
#pragma unroll 1
  for (int pass = 0; pass < NPASS; ++pass) {
    if (pass) __syncthreads();                       // every wave has left the previous pass's tile
    fill(pass);
    const unsigned char *wl = wglobal + (size_t)pass * KS * 6 * 1024 + lane * 16;
    h16x8 wq[WQ][6];                                 // [..][0..2] hi, [3..5] lo
#pragma unroll
    for (int d = 0; d < WQ; ++d)
#pragma unroll
      for (int f = 0; f < 6; ++f)
        wq[d][f] = *reinterpret_cast<const h16x8 *>(wl + (size_t)(d * 6 + f) * 1024);
    __syncthreads();                                 // tile (LDS-DMA) + table visible
    u32x4 kv = *reinterpret_cast<const u32x4 *>(ktab);
    Frag2 bcur[4], bnxt[4];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const unsigned char *p = tile + vbase + kv[0] + sub_off(sub);
      bcur[sub].hi = *reinterpret_cast<const h16x8 *>(p);
      bcur[sub].lo = *reinterpret_cast<const h16x8 *>(p + LO_OFF);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      // prefetch K-step s+1 (the final prefetch re-reads the last step: harmless)
      if ((s + 1) % 4 == 0) kv = *reinterpret_cast<const u32x4 *>(ktab + s + 1);
      const unsigned koff = kv[(s + 1) % 4];
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const unsigned char *p = tile + vbase + koff + sub_off(sub);
        bnxt[sub].hi = *reinterpret_cast<const h16x8 *>(p);
        bnxt[sub].lo = *reinterpret_cast<const h16x8 *>(p + LO_OFF);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[sub][b] = mfma16(wq[s % WQ][3 + b], bcur[sub].hi, acc[sub][b]);
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[sub][b] = mfma16(wq[s % WQ][b], bcur[sub].lo, acc[sub][b]);
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[sub][b] = mfma16(wq[s % WQ][b], bcur[sub].hi, acc[sub][b]);
      __builtin_amdgcn_s_setprio(0);
      if (s + WQ < KS) {
#pragma unroll
        for (int f = 0; f < 6; ++f)
          wq[s % WQ][f] = *reinterpret_cast<const h16x8 *>(wl + (size_t)((s + WQ) * 6 + f) * 1024);
      }
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) bcur[sub] = bnxt[sub];
    }
  }
}

template <int TZ, int TY, int TX, typename SubOff>
__device__ __forceinline__ void conv3s_kloop(const unsigned char *act, int AZ, int AY, int AX,
                                             int z0, int y0, int x0, unsigned char *tile,
                                             const unsigned *kofftab, const unsigned char *wglobal,
                                             unsigned vbase, SubOff sub_off, f32x4 (&acc)[4][3],
                                             int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  auto fill = [&](int pass) {
    stage_tile<TZ, TY, TX, PASS_BYTES, PASS_BYTES>(act + (int64_t)pass * AZ * AY * AX * PASS_BYTES, AZ, AY, AX,
                                                   z0, y0, x0, tile, wave, lane);
  };
  conv3s_kloop_f(fill, tile, kofftab, wglobal, vbase, sub_off, acc, tid);
}

// tile geometry of the 24-channel-pass K loop (vgg_like2's kernels and their tail below)
constexpr int M_TZ = 6, M_TY = 6, M_TX = 18;
constexpr int M_TILE_BYTES = ((M_TZ * M_TY * M_TX * PASS_BYTES + 1023) / 1024) * 1024;
constexpr int M_SMEM = M_TILE_BYTES + KTAB_BYTES;
static_assert(2 * M_SMEM <= 160 * 1024, "two workgroups must fit one CU");

// -------------------------------------------------------------------------------
// K2 (round 4): conv3 48->48 + conv1 48->48 + maxpool2 on the all-LDS K loop of
// vgg_split_lds.h: ONE persistent workgroup of 8 waves per CU, block = 8 x 4 x 16 pre-pool
// outputs (4 x 2 x 8 pooled), wave = pooled (pz, py) row, 4 sub-steps = its (dz, dy) window,
// x pairs pooled across neighbouring lanes.
// -------------------------------------------------------------------------------
struct MidXArgs {
  x8::Tensor p1;                 // pool-1 tensor
  const unsigned char *w3;       // [pass 6][K-step 7][part][b] fragments
  const h16x8 *w4;               // chain48 steps [4][b][lane], rows in pass order (il = 2)
  const float *shift3, *shift4;
  x8::Tensor p2;                 // pool-2 tensor
  x8::Walk walk;                 // blocks of 4 x 2 x 8 pooled voxels
  unsigned *flag;                // half-range guard (mfma_util.h)
};

__global__ __launch_bounds__(64 * x8::WAVES, 2) void vggs_mid_pool(MidXArgs a) {
  unsigned *ktab = reinterpret_cast<unsigned *>(smem + 2 * x8::BUF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  x8::ktab_init(ktab, tid);
  const x8::TileDma td = x8::tile_dma_init(wave, lane, a.p1.Y, a.p1.XP, (unsigned)a.p1.part_bytes());
  const int S = (int)gridDim.x >> 3, nbricks = a.walk.bricks();
  const int group = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  x8::Cursor cur;
  if (!x8::cursor_first(a.walk, nbricks, group, slot, S, cur)) return;
  const int64_t part = a.p1.part_bytes();
  auto origin = [&](const x8::Cursor &q) {
    return a.p1.p + a.p1.vox(8 * q.bz, 4 * q.by, 16 * q.bx) * 16;
  };
  const int pzl = wave >> 1, pyl = wave & 1;
  const unsigned vb = (unsigned)(((2 * pzl) * x8::ZS + (2 * pyl) * x8::TX + c) * 16);
  auto sub_off = [](int sub) -> unsigned { return (unsigned)((((sub >> 1) & 1) * x8::ZS + (sub & 1) * x8::TX) * 16); };
  unsigned ovf_all = 0u;
  x8::prime(smem, td, origin(cur), part, a.w3, wave, lane);      // also makes the table visible
  for (;;) {
    x8::Cursor nxt = cur;
    const bool has_next = x8::cursor_next(a.walk, slot, S, nxt);
    // (the opaque zero keeps block-invariant loads - shifts, conv1's weight fragments - INSIDE
    // the block loop: hoisted out of it their registers live through the K loop and spill)
    int zero = 0;
    asm volatile("" : "+s"(zero));
    f32x4 acc[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const f32x4 sh = *reinterpret_cast<const f32x4 *>(a.shift3 + zero + 16 * b + 4 * g);
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) acc[sub][b] = sh;
    }
    x8::conv_block(smem, ktab + 8 * g, td, origin(cur), origin(has_next ? nxt : cur), part, a.w3, vb,
                   sub_off, wave, lane, acc);
    // conv1 48->48 chained in registers, pooled over the 4 (dz, dy) window positions
    asm volatile("" : "+s"(zero));
    const h16x8 *w4p = a.w4 + zero;
    f32x4 sh4[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) sh4[b][r] = a.shift4[zero + x8::out_channel(b, g, r)];
    h16x8 w4[3][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int b = 0; b < 3; ++b) w4[b][s] = w4p[(s * 3 + b) * 64 + lane];
    f32x4 pooled[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    unsigned ovf = 0u;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const Frag2 h0 = pack_relu_split(acc[sub][0], acc[sub][1], ovf);
      const h16x8 hx = pack_relu_split_x(acc[sub][2], ovf);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const f32x4 a4 = chain48(w4[b], h0, hx, sh4[b]);
#pragma unroll
        for (int r = 0; r < 4; ++r) pooled[b][r] = __builtin_fmaxf(pooled[b][r], a4[r]);
      }
    }
    // pool the x pair: lanes c and c^1 hold neighbouring pre-pool x
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pooled[b][r] = __builtin_fmaxf(pooled[b][r], __shfl_xor(pooled[b][r], 1));
    const int pz = 4 * cur.bz + pzl, py = 2 * cur.by + pyl, px = 8 * cur.bx + (c >> 1);
    const bool inside = pz < a.p2.Z && py < a.p2.Y && px < a.p2.X;
    if ((c & 1) == 0 && inside)
      x8::store12(a.p2, a.p2.vox(pz, py, px), g, pooled, ovf);
    // edge tiles read past the tensor (x8::Tensor): only what is stored counts for the guard
    ovf_all = pk_max_i16(ovf_all, inside ? ovf : 0u);
    if (!has_next) break;
    cur = nxt;
  }
  ovf_commit(ovf_all, a.flag, FPL_RANGE_MID);
}

// -------------------------------------------------------------------------------
// The head both tail kernels run on their conv3 accumulators, in registers: ReLU + split,
// conv1 48->96, conv1 96->96, conv1 96->1 -> the logit of each of the wave's 4 x 16 coarse voxels
// (returned in lanes g = 0 .. 3 alike).  Two sub-steps at a time (all four in lockstep keep 96
// registers of hi / lo fragments per layer alive and spill), one output-channel PAIR at a time
// (= one K-step of the next layer), conv1 96->1 folded into conv1 96->96's loop, so no layer's
// full output is ever live.  w6: chain48 steps [4][b]; w7, w8: [part][s][b].
// -------------------------------------------------------------------------------
constexpr int T_W7 = 3 * 6, T_W8 = 3;   // fragments per part of L7, L8
__device__ __forceinline__ void head_chain(const f32x4 (&acc)[4][3], const unsigned char *w6p,
                                           const unsigned char *w7p, const unsigned char *w8p, const float *sh6p,
                                           const float *sh7p, float bias8, int lane, unsigned &ovf, float (&logit)[4]) {
  const int c = lane & 15, g = lane >> 4;
  auto frag = [&](const unsigned char *w, int part_, int nfrag, int f) {
    return *reinterpret_cast<const h16x8 *>(w + ((size_t)(part_ * nfrag + f) * 64 + lane) * 16);
  };
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) {
    Frag2 h5[2];
    h16x8 h5x[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      h5[q] = pack_relu_split(acc[2 * sp + q][0], acc[2 * sp + q][1], ovf);
      h5x[q] = pack_relu_split_x(acc[2 * sp + q][2], ovf);
    }
    Frag2 h6[2][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      f32x4 a6[2][2];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int b = 2 * s + bb;
        f32x4 sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) sh[r] = sh6p[16 * b + 4 * g + r];
        h16x8 w6[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) w6[t] = frag(w6p, 0, 0, t * 6 + b);
#pragma unroll
        for (int q = 0; q < 2; ++q) a6[q][bb] = chain48(w6, h5[q], h5x[q], sh);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) h6[q][s] = pack_relu_split(a6[q][0], a6[q][1], ovf);
    }
    f32x4 a8[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      f32x4 a7[2][2];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int b = 2 * s + bb;
        f32x4 sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) sh[r] = sh7p[16 * b + 4 * g + r];
#pragma unroll
        for (int q = 0; q < 2; ++q) a7[q][bb] = sh;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const h16x8 wh = frag(w7p, 0, T_W7, t * 6 + b), wl = frag(w7p, 1, T_W7, t * 6 + b);
#pragma unroll
          for (int q = 0; q < 2; ++q) a7[q][bb] = mfma3(wh, wl, h6[q][t], a7[q][bb]);
        }
      }
      const h16x8 w8h = frag(w8p, 0, T_W8, s), w8l = frag(w8p, 1, T_W8, s);
#pragma unroll
      for (int q = 0; q < 2; ++q) a8[q] = mfma3(w8h, w8l, pack_relu_split(a7[q][0], a7[q][1], ovf), a8[q]);
    }
    // lane (c, g = 0) register 0 holds the logit of coarse voxel c
#pragma unroll
    for (int q = 0; q < 2; ++q) logit[2 * sp + q] = __shfl(a8[q][0], c) + bias8;
    __builtin_amdgcn_sched_barrier(0);       // the second pair's weight loads stay behind the first's work
  }
}

// -------------------------------------------------------------------------------
// K3: conv3 48->48 + BN + ReLU on P2, then - in registers - conv1 48->96, conv1 96->96,
// conv1 96->1 + bias, sigmoid and the x4 nearest upsample store into the (Z,Y,X) f32
// prediction volume (vgg_c5_tail's geometry: block 4 x 4 x 16, wave = z, sub-steps = the
// 4 y rows).  The 1x1 chain runs on two sub-steps at a time: with hi and lo fragments all
// four in lockstep do not fit the register file.
// -------------------------------------------------------------------------------
struct TailSArgs {
  const unsigned char *p2;
  int P2Z, P2Y, P2X;
  const unsigned char *w5;       // [pass][KS][part][b]
  const float *shift5;
  int CZ, CY, CX;                // chunk-local coarse dims
  const unsigned char *w6, *w7, *w8;   // L6: chain48 steps [4][b]; L7, L8: [part][s][b]
  const float *shift6, *shift7;
  float bias8;
  float *dst;                    // (Z,Y,X) prediction volume, row 0
  int64_t DY, DX;                // its pitches
  int64_t gz0;                   // global coarse z of chunk-local coarse row 0
  int64_t VZ, VY, VX;            // valid fine extents (dim - 2 * off)
  int off;                       // rf offset of the network: 7
  BlockGrid bg;
  unsigned *flag;                // half-range guard (mfma_util.h)
};

// (this form - 24-channel passes, two 4-wave workgroups per CU - serves vgg_like2, whose
// tensors keep the round-3 layout; vgg_like runs vggs_c5_tail below)
__global__ __launch_bounds__(256, 2) void vggs_c5_tail_p24(TailSArgs a) {
  unsigned char *tile = smem;
  unsigned *kofftab = reinterpret_cast<unsigned *>(smem + M_TILE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  int xb, yb, zb;
  if (!brick_coords(a.bg, xb, yb, zb)) return;
  const int cx0 = xb * 16, cy0 = yb * 4, cz0 = zb * 4;
  if (tid < 4 * KTAB) kofftab[tid] = kslot_entry<M_TY, M_TX>(tid);

  const unsigned vbase = (unsigned)(((wave * M_TY) * M_TX + c) * PASS_BYTES);
  f32x4 acc[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    f32x4 sh;
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[r] = a.shift5[16 * b + 4 * g + r];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) acc[sub][b] = sh;
  }
  auto sub_off = [](int sub) -> unsigned { return (unsigned)(sub * M_TX * PASS_BYTES); };
  conv3s_kloop<M_TZ, M_TY, M_TX>(a.p2, a.P2Z, a.P2Y, a.P2X, cz0, cy0, cx0, tile, kofftab, a.w5,
                                 vbase, sub_off, acc, tid);

  float logit[4];
  unsigned ovf = 0u;
  head_chain(acc, a.w6, a.w7, a.w8, a.shift6, a.shift7, a.bias8, lane, ovf, logit);
  ovf_commit(ovf, a.flag, FPL_RANGE_TAIL);

  const int cz = cz0 + wave, cx = cx0 + c;
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    const int cy = cy0 + sub;
    const float p = 1.f / (1.f + expf(-logit[sub]));
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
// K3 (round 4): conv3 48->48 + BN + ReLU on P2 through the all-LDS K loop (vgg_split_lds.h),
// block 8 x 4 x 16 coarse voxels, wave = z, sub-steps = the 4 y rows; then - in registers -
// conv1 48->96, conv1 96->96, conv1 96->1 + bias, sigmoid and the x4 nearest upsample store.
// The 1x1 chain runs on two sub-steps at a time (all four in lockstep keep 96 registers of
// hi / lo fragments per layer alive and spill), one output-channel PAIR at a time, conv1 96->1
// folded into conv1 96->96's loop, so no layer's full output is ever live.
// -------------------------------------------------------------------------------
struct TailXArgs {
  x8::Tensor p2;
  const unsigned char *w5;       // [pass 6][K-step 7][part][b]
  const float *shift5;
  int CZ, CY, CX;                // chunk-local coarse dims
  const unsigned char *w6, *w7, *w8;   // L6: chain48 steps [4][b]; L7, L8: [part][s][b]
  const float *shift6, *shift7;
  float bias8;
  float *dst;                    // (Z,Y,X) prediction volume, row 0
  int64_t DY, DX;                // its pitches
  int64_t gz0;                   // global coarse z of chunk-local coarse row 0
  int64_t VZ, VY, VX;            // valid fine extents (dim - 2 * off)
  int off;                       // rf offset of the network: 7
  x8::Walk walk;                 // blocks of 8 x 4 x 16 coarse voxels
  unsigned *flag;                // half-range guard (mfma_util.h)
};

__global__ __launch_bounds__(64 * x8::WAVES, 2) void vggs_c5_tail(TailXArgs a) {
  unsigned *ktab = reinterpret_cast<unsigned *>(smem + 2 * x8::BUF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  x8::ktab_init(ktab, tid);
  const x8::TileDma td = x8::tile_dma_init(wave, lane, a.p2.Y, a.p2.XP, (unsigned)a.p2.part_bytes());
  const int S = (int)gridDim.x >> 3, nbricks = a.walk.bricks();
  const int group = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  x8::Cursor cur;
  if (!x8::cursor_first(a.walk, nbricks, group, slot, S, cur)) return;
  const int64_t part = a.p2.part_bytes();
  auto origin = [&](const x8::Cursor &q) {
    return a.p2.p + a.p2.vox(8 * q.bz, 4 * q.by, 16 * q.bx) * 16;
  };
  const unsigned vb = (unsigned)((wave * x8::ZS + c) * 16);
  auto sub_off = [](int sub) -> unsigned { return (unsigned)(sub * x8::TX * 16); };
  unsigned ovf_all = 0u;
  x8::prime(smem, td, origin(cur), part, a.w5, wave, lane);
  for (;;) {
    x8::Cursor nxt = cur;
    const bool has_next = x8::cursor_next(a.walk, slot, S, nxt);
    // (the opaque zero keeps block-invariant loads - shifts, the head's 66 weight fragments -
    // INSIDE the block loop: hoisted out of it they are live registers, i.e. spills)
    int zero = 0;
    asm volatile("" : "+s"(zero));
    f32x4 acc[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const f32x4 sh = *reinterpret_cast<const f32x4 *>(a.shift5 + zero + 16 * b + 4 * g);
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) acc[sub][b] = sh;
    }
    x8::conv_block(smem, ktab + 8 * g, td, origin(cur), origin(has_next ? nxt : cur), part, a.w5, vb,
                   sub_off, wave, lane, acc);

    unsigned ovf = 0u;
    float logit[4];
    asm volatile("" : "+s"(zero));
    head_chain(acc, a.w6 + zero, a.w7 + zero, a.w8 + zero, a.shift6 + zero, a.shift7 + zero, a.bias8, lane, ovf,
               logit);

    const int cz = 8 * cur.bz + wave, cx = 16 * cur.bx + c;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int cy = 4 * cur.by + sub;
      const float p = 1.f / (1.f + expf(-logit[sub]));
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
    // (voxels past the coarse extent are computed from real rows of the tensor or from its
    // zeroed slack: they cannot raise the guard on their own)
    ovf_all = pk_max_i16(ovf_all, ovf);
    if (!has_next) break;
    cur = nxt;
  }
  ovf_commit(ovf_all, a.flag, FPL_RANGE_TAIL);
}

// -------------------------------------------------------------------------------
// vgg_like2 on split halves (flypylib/fplmodels.py:138-172; the model of the reference's
// scripts/fpl_cx1_0_vgg_4ss.py): [conv3 1->48, conv3 48->48, pool] [conv3, conv3, pool] conv3,
// head.  As in vgg_fused.hip::vgg2_conv3 one kernel template covers the three 48->48
// convolutions before the tail (vggs_c5_tail is the fifth and the head): block 4 x 4 x 16
// outputs, the K loop of K2 / K3, and
//   STEM  each pass's 6 x 6 x 18 tile (24 channels, hi | lo) is COMPUTED from the raw 8 x 8 x 20
//         input tile: conv3 1->48 + BN + ReLU as 41 groups of 16 tile voxels x two M-blocks
//         (the pass's 24 channels + 8 zero rows) x three MFMAs on hi / lo operands;
//   POOL  ReLU + 2x2x2 max pool in fp32, then the split (wave = pooled (z,y) row as in K2),
//         otherwise ReLU, split and a plain store (wave = z as in K3).
// The 48->48 weight rows are interleaved (12 contiguous channels per lane: store_split12).
// -------------------------------------------------------------------------------
constexpr int V2_RZ = M_TZ + 2, V2_RY = M_TY + 2, V2_RX = M_TX + 2;     // raw tile 8 x 8 x 20
constexpr int V2_NRAW = V2_RZ * V2_RY * V2_RX;
static_assert(V2_NRAW % 256 == 0, "raw tile pieces per thread");
constexpr int V2_SMEM = M_TILE_BYTES + KTAB_BYTES + V2_NRAW * 4 + 256 * 4;
static_assert(2 * V2_SMEM <= 160 * 1024, "two vgg_like2 workgroups must fit one CU");

struct V2SArgs {
  // STEM: the raw volume
  const void *src;
  int64_t SZ, SY, SX;
  float mean, sd;
  int64_t gz0;                   // raw z of chunk-local conv row 0
  const h16x8 *wstem;            // [pass][part][blk] fragments, k-slot (g,j) = tap 8g + j
  const float *shstem;           // [48]
  // uint8 volumes (as K1): the operand is u - c0, 1 / sd is in `wstem`, and the initial
  // accumulator values come from shtab[(pz, py, px) planes of padding in the window][48]
  float c0;
  const float *shtab;
  // otherwise: a split tensor in pass planes
  const unsigned char *in;
  int IZ, IY, IX;
  const unsigned char *w;        // [pass][KS][part][b], interleaved rows
  const float *shift;
  unsigned char *out;
  int OZ, OY, OX;                // output dims (pooled dims with POOL)
  BlockGrid bg;
  unsigned *flag;                // half-range guard, as K1 / K2 (xlim: STEM on float volumes)
  float xlim;
};

template <bool STEM, bool POOL, typename SRC>
__global__ __launch_bounds__(256, 2) void vggs2_conv3(V2SArgs a) {
  unsigned char *tile = smem;
  unsigned *kofftab = reinterpret_cast<unsigned *>(smem + M_TILE_BYTES);
  unsigned *rawt = reinterpret_cast<unsigned *>(smem + M_TILE_BYTES + KTAB_BYTES);   // hi | lo << 16
  unsigned *lut = rawt + V2_NRAW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  int xb, yb, zb;
  if (!brick_coords(a.bg, xb, yb, zb)) return;
  // origin of the block's 4 x 4 x 16 conv outputs (= of its input tile)
  const int x0 = xb * 16, y0 = yb * 4, z0 = zb * 4;
  if (tid < 4 * KTAB) kofftab[tid] = kslot_entry<M_TY, M_TX>(tid);
  int toff[8];
  unsigned ovf = 0u, xmax = 0u;
  if (STEM) {
    auto norm_bits = [&](float raw) -> unsigned {
      const float x = (raw - a.mean) / a.sd;
      const unsigned ax = __builtin_bit_cast(unsigned, x) & 0x7FFFFFFFu;
      xmax = ax > xmax ? ax : xmax;
      const h16_t h = (h16_t)x;
      return (unsigned)h16_bits(x) | ((unsigned)h16_bits(x - (float)h) << 16);
    };
    const SRC *src = (const SRC *)a.src;
    if (sizeof(SRC) == 1) {
      lut[tid] = (unsigned)h16_bits((float)tid - a.c0);    // exact; lo half 0; padding 0
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < V2_NRAW / 256; ++j) {
      const int p = tid + 256 * j;
      const int64_t z = a.gz0 + z0 + p / (V2_RY * V2_RX), y = y0 + (p / V2_RX) % V2_RY, x = x0 + p % V2_RX;
      unsigned b = 0u;                                   // zero past the volume end
      if (z < a.SZ && y < a.SY && x < a.SX) {
        const SRC v = src[(z * a.SY + y) * a.SX + x];
        b = sizeof(SRC) == 1 ? lut[(int)v] : norm_bits((float)v);
      }
      rawt[p] = b;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 8 * g + j;
      toff[j] = t < 27 ? ((t / 9) * V2_RY + (t / 3) % 3) * V2_RX + t % 3 : 0;
    }
    __syncthreads();                                    // raw tile visible
  }
  auto fill = [&](int pass) {
    if (STEM) {
      h16x8 wst[2][2];                                  // [part][blk]
      f32x4 shs[2];
#pragma unroll
      for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) wst[part][blk] = a.wstem[((pass * 2 + part) * 2 + blk) * 64 + lane];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lc = 16 * blk + 4 * g + r;          // channel inside the pass
          shs[blk][r] = lc < CHP ? a.shstem[CHP * pass + lc] : 0.f;
        }
      constexpr int NVOX = M_TZ * M_TY * M_TX, NGRP = (NVOX + 15) / 16;
      for (int grp = wave; grp < NGRP; grp += 4) {
        const int v = 16 * grp + c, vv = v < NVOX ? v : NVOX - 1;
        const int tz = vv / (M_TY * M_TX), ty = (vv / M_TX) % M_TY, tx = vv % M_TX;
        const int ro = (tz * V2_RY + ty) * V2_RX + tx;
        u16x8 rh, rl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned t = rawt[ro + toff[j]];
          rh[j] = (unsigned short)t; rl[j] = (unsigned short)(t >> 16);
        }
        Frag2 bf;
        bf.hi = __builtin_bit_cast(h16x8, rh);
        bf.lo = __builtin_bit_cast(h16x8, rl);
        // uint8: the window's planes of padding select the initial values (entry 0 inside)
        const float *init = nullptr;
        if (sizeof(SRC) == 1) {
          const int pz = min(max((int)(a.gz0 + z0 + tz + 3 - a.SZ), 0), 3);
          const int py = min(max((int)(y0 + ty + 3 - a.SY), 0), 3), px = min(max((int)(x0 + tx + 3 - a.SX), 0), 3);
          init = a.shtab + ((pz * 4 + py) * 4 + px) * 48 + CHP * pass + 4 * g;
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          f32x4 a1;
          if (sizeof(SRC) == 1) {
            f32x4 i0 = {0.f, 0.f, 0.f, 0.f};
            if (16 * blk + 4 * g < CHP) i0 = *reinterpret_cast<const f32x4 *>(init + 16 * blk);
            a1 = mfma16(wst[1][blk], bf.hi, i0);
            a1 = mfma16(wst[0][blk], bf.hi, a1);
          } else {
            a1 = mfma3(wst[0][blk], wst[1][blk], bf, shs[blk]);
          }
          const Pair2 p0 = split_pk_relu(a1[0], a1[1]), p1 = split_pk_relu(a1[2], a1[3]);
          if (v < NVOX && 16 * blk + 4 * g < CHP) {
            unsigned char *d = tile + v * PASS_BYTES + (16 * blk + 4 * g) * 2;
            *reinterpret_cast<u32x2 *>(d) = u32x2{p0.hi, p1.hi};
            *reinterpret_cast<u32x2 *>(d + LO_OFF) = u32x2{p0.lo, p1.lo};
          }
        }
      }
    } else {
      stage_tile<M_TZ, M_TY, M_TX, PASS_BYTES, PASS_BYTES>(
          a.in + (int64_t)pass * a.IZ * a.IY * a.IX * PASS_BYTES, a.IZ, a.IY, a.IX, z0, y0, x0, tile, wave, lane);
    }
  };

  // POOL: wave = pooled (z,y) row, sub-steps = the (dz,dy) window; else wave = z, subs = y
  const int pzl = wave >> 1, pyl = wave & 1;
  const unsigned vbase = POOL ? (unsigned)((((2 * pzl) * M_TY + 2 * pyl) * M_TX + c) * PASS_BYTES)
                              : (unsigned)(((wave * M_TY) * M_TX + c) * PASS_BYTES);
  f32x4 acc[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    f32x4 sh;
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[r] = a.shift[12 * g + 4 * b + r];      // interleaved rows
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) acc[sub][b] = sh;
  }
  auto sub_off = [](int sub) -> unsigned {
    return POOL ? (unsigned)(((((sub >> 1) & 1) * M_TY + (sub & 1)) * M_TX) * PASS_BYTES)
                : (unsigned)(sub * M_TX * PASS_BYTES);
  };
  conv3s_kloop_f(fill, tile, kofftab, a.w, vbase, sub_off, acc, tid);

  const int64_t plane = (int64_t)a.OZ * a.OY * a.OX * PASS_BYTES;
  if (POOL) {
    f32x4 pooled[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // 0 = the ReLU
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) pooled[b][r] = __builtin_fmaxf(pooled[b][r], acc[sub][b][r]);
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pooled[b][r] = __builtin_fmaxf(pooled[b][r], __shfl_xor(pooled[b][r], 1));     // the x pair
    const int pz = zb * 2 + pzl, py = yb * 2 + pyl, px = xb * 8 + (c >> 1);
    if ((c & 1) == 0 && pz < a.OZ && py < a.OY && px < a.OX)
      store_split12(a.out + (((int64_t)pz * a.OY + py) * a.OX + px) * PASS_BYTES, plane, g, pooled, ovf);
  } else {
    const int oz = z0 + wave, ox = x0 + c;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int oy = y0 + sub;
      f32x4 o[3];
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[b][r] = __builtin_fmaxf(acc[sub][b][r], 0.f);
      if (oz < a.OZ && oy < a.OY && ox < a.OX)
        store_split12(a.out + (((int64_t)oz * a.OY + oy) * a.OX + ox) * PASS_BYTES, plane, g, o, ovf);
    }
  }
  ovf_commit(ovf, a.flag, FPL_RANGE_MID);
  if (STEM && sizeof(SRC) != 1 && xmax > __builtin_bit_cast(unsigned, a.xlim)) atomicOr(a.flag, FPL_RANGE_INPUT);
}

// -------------------------------------------------------------------------------
// host side: weight packing, slab orchestration
// -------------------------------------------------------------------------------
struct SplitState {
  uint64_t version = ~0ull;
  unsigned char *frags = nullptr;
  float *shifts = nullptr;
  size_t off_w[8] = {0};              // byte offsets of L1..L8 fragment sets
  size_t off_s[8] = {0};              // float offsets of shift1..shift8
  float bias8 = 0.f;
  // u8 volumes: conv3 1->48 on the operand u - c0, fragments of w / sd and the shift with the
  // constant folded in, rebuilt when (mean, sd) change
  unsigned char *w1_int = nullptr;    // [part][e][b][lane] as off_w[0]
  float *shift1_int = nullptr;
  float int_mean = 0.f, int_sd = 0.f, int_c0 = 0.f;
  bool int_valid = false;
  std::vector<uint16_t> w1_int_host;
  std::vector<float> shift1_int_host;       // the table [pz][py][px][48]
  // half-range guard: float volumes must satisfy |normalised voxel| <= xlim for conv3 1->48's
  // outputs to stay inside the half range (kernels check every voxel they load)
  float xlim = 0.f;
};

// Outputs of conv3 1->48 (+ BN) are bounded by |shift[c]| + |scale[c]| sum_taps |w[tap][c]| * X
// for inputs |x| <= X.  FPL_RANGE_LIM leaves room for the rounding of the bound itself.
constexpr double FPL_RANGE_LIM = 65000.0;
// largest X for which every channel's bound stays below FPL_RANGE_LIM (<= 0: no such X)
double stem_input_limit(const float *A, const fpl_op &op, const float *scale, const float *shift_tab,
                        int n_tab) {
  double xl = FPL_RANGE_LIM;
  for (int co = 0; co < op.cout; ++co) {
    double sw = 0.0, sh = 0.0;
    for (int tap = 0; tap < 27; ++tap) sw += std::fabs((double)A[op.w_off + (size_t)tap * op.cout + co]);
    sw *= std::fabs((double)scale[co]);
    for (int e = 0; e < n_tab; ++e) sh = std::max(sh, std::fabs((double)shift_tab[(size_t)e * 48 + co]));
    if (!(sh < FPL_RANGE_LIM)) return 0.0;
    if (sw > 0.0) xl = std::min(xl, (FPL_RANGE_LIM - sh) / sw);
  }
  return xl;
}

void split_state_free(fpl_ctx *ctx, void *p) {
  SplitState *s = (SplitState *)p;
  if (s->frags) hipFree(s->frags);
  if (s->shifts) hipFree(s->shifts);
  if (s->w1_int) hipFree(s->w1_int);
  if (s->shift1_int) hipFree(s->shift1_int);
  delete s;
}

int split_prepare(fpl_ctx *ctx, fpl_program *prog, SplitState **out) {
  SplitState *st = (SplitState *)prog->fast_state_h16[2];
  if (!st) {
    st = new SplitState();
    prog->fast_state_h16[2] = st;
    prog->fast_state_h16_free[2] = split_state_free;
  }
  *out = st;
  if (st->version == prog->arena_version) return 0;
  static const int conv_ops[8] = {0, 1, 3, 4, 6, 7, 8, 9};
  static const int mblocks[8] = {3, 3, 3, 3, 3, 6, 6, 1};
  static const int ksteps[8] = {1, 2, KS, 2, KS, 2, 3, 3};
  const bool v2 = fpl_vgg_variant(prog) == 2;        // vgg_like2: L2 and L4 are 3x3x3 as well
  std::vector<uint16_t> all;
  std::vector<float> shifts;
  const float *A = prog->arena_host.data();
  for (int l = 0; l < 8; ++l) {
    const fpl_op &op = prog->ops[conv_ops[l]];
    std::vector<float> scale(A + op.scale_off, A + op.scale_off + op.cout);
    st->off_w[l] = all.size() * sizeof(uint16_t);
    if (l == 0 && v2) {
      // vggs2_conv3<STEM>: per pass the 24 channels of that pass as two M-blocks (8 zero
      // rows), one K-step of 27 taps, k-slot (g, j) = tap 8g + j: [pass][part][blk]
      for (int pass = 0; pass < NPASS; ++pass) {
        std::vector<float> wp((size_t)27 * 32, 0.f), sp(32, 0.f);
        for (int tap = 0; tap < 27; ++tap)
          for (int ch = 0; ch < CHP; ++ch)
            wp[(size_t)tap * 32 + ch] = A[op.w_off + (size_t)tap * op.cout + pass * CHP + ch];
        for (int ch = 0; ch < CHP; ++ch) sp[ch] = scale[pass * CHP + ch];
        for (int part = 0; part < 2; ++part) {
          std::vector<uint16_t> f;
          fpl_pack_frags(wp.data(), sp.data(), 27, 1, 32, 2, 1, SLOT_SPATIAL, &f, false, part);
          all.insert(all.end(), f.begin(), f.end());
        }
      }
    } else if (l == 0) {
      for (int part = 0; part < 2; ++part) {       // [part][e][b]
        std::vector<uint16_t> f;
        fpl_pack_stem(A + op.w_off, scale.data(), op.cout, &f, part);
        all.insert(all.end(), f.begin(), f.end());
      }
    } else if (op.k == 3 && !v2) {
      // vgg_like (vgg_split_lds.h): passes of 8 channels, [pass 6][K-step 7][part][b] - a pass's
      // 42 fragments are one contiguous 42-KiB run for the LDS-DMA
      for (int pass = 0; pass < x8::NQ; ++pass) {
        std::vector<float> wp((size_t)27 * x8::CQ * op.cout);
        for (int tap = 0; tap < 27; ++tap)
          for (int ch = 0; ch < x8::CQ; ++ch)
            memcpy(&wp[((size_t)tap * x8::CQ + ch) * op.cout],
                   A + op.w_off + ((size_t)tap * op.cin + pass * x8::CQ + ch) * op.cout,
                   op.cout * sizeof(float));
        std::vector<uint16_t> f[2];
        for (int part = 0; part < 2; ++part)
          fpl_pack_frags(wp.data(), scale.data(), 27, x8::CQ, op.cout, 3, x8::KQ, SLOT_SPATIAL, &f[part], 0, part);
        for (int s = 0; s < x8::KQ; ++s)
          for (int part = 0; part < 2; ++part)
            all.insert(all.end(), f[part].begin() + (size_t)s * 3 * 512,
                       f[part].begin() + (size_t)(s + 1) * 3 * 512);
      }
    } else if (op.k == 3) {
      // [pass][K-step][part][b]: per pass the (tap, channel-in-pass) sub-matrix
      for (int pass = 0; pass < NPASS; ++pass) {
        std::vector<float> wp((size_t)27 * CHP * op.cout);
        for (int tap = 0; tap < 27; ++tap)
          for (int ch = 0; ch < CHP; ++ch)
            memcpy(&wp[((size_t)tap * CHP + ch) * op.cout],
                   A + op.w_off + ((size_t)tap * op.cin + pass * CHP + ch) * op.cout,
                   op.cout * sizeof(float));
        std::vector<uint16_t> f[2];
        for (int part = 0; part < 2; ++part)
          fpl_pack_frags(wp.data(), scale.data(), 27, CHP, op.cout, 3, KS, SLOT_SPATIAL, &f[part],
                         /*il=*/v2 && l >= 1 && l <= 3, part);     // vggs2_conv3 stores 12-channel runs
        for (int s = 0; s < KS; ++s)
          for (int part = 0; part < 2; ++part)
            all.insert(all.end(), f[part].begin() + (size_t)s * 3 * 512,
                       f[part].begin() + (size_t)(s + 1) * 3 * 512);
      }
    } else if (op.cin == CH) {
      // chain48 steps: [blocks 0,1 hi], [blocks 0,1 lo], [blk 2: lo | hi], [blk 2: hi | lo]
      static const int blk[4][2] = {{0, 1}, {0, 1}, {2, 2}, {2, 2}};
      static const int part[4][2] = {{0, 0}, {1, 1}, {1, 0}, {0, 1}};
      std::vector<uint16_t> f;
      for (int s = 0; s < 4; ++s)
        fpl_pack_chain_step(A + op.w_off, scale.data(), op.cin, op.cout, mblocks[l], blk[s], part[s], &f,
                            /*il=*/op.cout == CH ? 2 : 0);   // L2 / L4 write P1 / P2: rows in pass order
      all.insert(all.end(), f.begin(), f.end());
    } else {
      for (int part = 0; part < 2; ++part) {       // [part][s][b]
        std::vector<uint16_t> f;
        fpl_pack_frags(A + op.w_off, scale.data(), 1, op.cin, op.cout, mblocks[l], ksteps[l],
                       SLOT_CHAIN, &f, false, part);
        all.insert(all.end(), f.begin(), f.end());
      }
    }
    st->off_s[l] = shifts.size();
    shifts.insert(shifts.end(), A + op.shift_off, A + op.shift_off + op.cout);
    while (shifts.size() % 4) shifts.push_back(0.f);
  }
  for (uint16_t h : all)
    if ((h & 0x7C00u) == 0x7C00u)
      return fpl_fail_range(ctx, "a folded weight exceeds the IEEE-half range (65504); use precision "
                                 "f32 (or 'auto') for this network");
  {
    const fpl_op &op0 = prog->ops[0];
    std::vector<float> sh48(48, 0.f);
    memcpy(sh48.data(), A + op0.shift_off, op0.cout * sizeof(float));
    st->xlim = (float)stem_input_limit(A, op0, A + op0.scale_off, sh48.data(), 1);
    if (!(st->xlim > 0.f))
      return fpl_fail_range(ctx, "the first layer's shift exceeds the IEEE-half range; use precision f32 "
                                 "(or 'auto') for this network");
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
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)vggs_stem_pool<uint8_t>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, S_SMEM));
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)vggs_stem_pool<float>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, S_SMEM));
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)vggs_mid_pool,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, x8::SMEM));
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)vggs_c5_tail_p24,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, M_SMEM));
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)vggs_c5_tail,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, x8::SMEM));
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)vggs2_conv3<true, true, uint8_t>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, V2_SMEM));
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)vggs2_conv3<true, true, float>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, V2_SMEM));
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)vggs2_conv3<false, false, uint8_t>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, V2_SMEM));
  FPL_HIP(ctx, hipFuncSetAttribute((const void *)vggs2_conv3<false, true, uint8_t>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, V2_SMEM));
  st->version = prog->arena_version;
  st->int_valid = false;
  return 0;
}

// the stem's conv3 for uint8 volumes normalised with (mean, sd): operand u - c0
int split_prepare_int(fpl_ctx *ctx, fpl_program *prog, SplitState *st, float mean, float sd) {
  if (st->int_valid && st->int_mean == mean && st->int_sd == sd) return 0;
  const fpl_op &op = prog->ops[0];
  const float *A = prog->arena_host.data();
  const double c0 = std::min(255.0, std::max(0.0, std::nearbyint((double)mean)));
  std::vector<float> scale(op.cout);
  st->shift1_int_host.assign(S_SHTAB, 0.f);
  const double k = (c0 - (double)mean) / (double)sd;
  for (int co = 0; co < op.cout; ++co) {
    scale[co] = (float)((double)A[op.scale_off + co] / (double)sd);
    // entry (pz, py, px): the constant over the taps that are NOT padding, i.e. with
    // dz < 3 - pz, dy < 3 - py, dx < 3 - px
    for (int pz = 0; pz < 4; ++pz)
      for (int py = 0; py < 4; ++py)
        for (int px = 0; px < 4; ++px) {
          double sum = 0.0;
          for (int dz = 0; dz < 3 - pz; ++dz)
            for (int dy = 0; dy < 3 - py; ++dy)
              for (int dx = 0; dx < 3 - px; ++dx)
                sum += (double)A[op.w_off + (size_t)((dz * 3 + dy) * 3 + dx) * op.cout + co];
          st->shift1_int_host[(size_t)((pz * 4 + py) * 4 + px) * 48 + co] =
              (float)((double)A[op.shift_off + co] + k * sum * (double)A[op.scale_off + co]);
        }
  }
  st->w1_int_host.clear();
  if (fpl_vgg_variant(prog) == 2) {
    // vggs2_conv3<STEM>: [pass][part][blk] as in split_prepare, with 1 / sd in the scale
    for (int pass = 0; pass < NPASS; ++pass) {
      std::vector<float> wp((size_t)27 * 32, 0.f), sp(32, 0.f);
      for (int tap = 0; tap < 27; ++tap)
        for (int ch = 0; ch < CHP; ++ch)
          wp[(size_t)tap * 32 + ch] = A[op.w_off + (size_t)tap * op.cout + pass * CHP + ch];
      for (int ch = 0; ch < CHP; ++ch) sp[ch] = scale[pass * CHP + ch];
      for (int part = 0; part < 2; ++part) {
        std::vector<uint16_t> f;
        fpl_pack_frags(wp.data(), sp.data(), 27, 1, 32, 2, 1, SLOT_SPATIAL, &f, false, part);
        st->w1_int_host.insert(st->w1_int_host.end(), f.begin(), f.end());
      }
    }
    st->w1_int_host.resize((size_t)2 * 6 * 512, 0);          // the buffer's fixed size
  } else {
    for (int part = 0; part < 2; ++part) {         // [part][e][b]
      std::vector<uint16_t> f;
      fpl_pack_stem(A + op.w_off, scale.data(), op.cout, &f, part);
      st->w1_int_host.insert(st->w1_int_host.end(), f.begin(), f.end());
    }
  }
  for (uint16_t h : st->w1_int_host)
    if ((h & 0x7C00u) == 0x7C00u)
      return fpl_fail_range_call(ctx, "a first-layer weight divided by std %g exceeds the IEEE-half range; use "
                                 "precision f32 (or 'auto') for this normalisation", (double)sd);
  // half-range guard: the operand is |u - c0| <= max(c0, 255 - c0)
  if (!(stem_input_limit(A, op, scale.data(), st->shift1_int_host.data(), 64) >= std::max(c0, 255.0 - c0)))
    return fpl_fail_range_call(ctx, "the first layer's outputs may exceed the IEEE-half range at mean %g, std %g; "
                               "use precision f32 (or 'auto')", (double)mean, (double)sd);
  const size_t wb = st->w1_int_host.size() * sizeof(uint16_t);
  if (!st->w1_int) FPL_HIP(ctx, hipMalloc((void **)&st->w1_int, wb));
  if (!st->shift1_int) FPL_HIP(ctx, hipMalloc((void **)&st->shift1_int, S_SHTAB * sizeof(float)));
  // stream-ordered behind any launch that still reads the previous set
  FPL_HIP(ctx, hipMemcpyAsync(st->w1_int, st->w1_int_host.data(), wb, hipMemcpyHostToDevice, ctx->stream));
  FPL_HIP(ctx, hipMemcpyAsync(st->shift1_int, st->shift1_int_host.data(), S_SHTAB * sizeof(float),
                              hipMemcpyHostToDevice, ctx->stream));
  st->int_c0 = (float)c0;
  st->int_mean = mean; st->int_sd = sd;
  st->int_valid = true;
  return 0;
}

}  // namespace

bool fpl_split_path_available(const fpl_program *prog, int precision, const int32_t offset[3],
                              const int32_t out_sz[3]) {
  const int variant = fpl_vgg_variant(prog);           // 1: vgg_like (offset 7), 2: vgg_like2 (10)
  if (precision != FPL_PREC_F16S || variant == 0) return false;
  for (int a = 0; a < 3; ++a)
    if (offset[a] != (variant == 1 ? 7 : 10) || out_sz[a] % 4 != 0) return false;
  return true;
}

namespace {

// vgg_like2 over the coarse rows of a slab, as vgg_fused.hip::vgg2_infer (where the lattice
// argument is made): four launches per chunk, the three intermediate tensors in pass planes
//   H1 = pool(conv3(conv3(volume)))   vggs2_conv3<STEM, POOL>   (half resolution)
//   L3 = conv3(H1)                    vggs2_conv3<>
//   Q  = pool(conv3(L3))              vggs2_conv3<POOL>         (quarter resolution)
//   prediction = head(conv3(Q))       vggs_c5_tail
int split2_infer(fpl_ctx *ctx, SplitState *st, const void *src, int src_dtype, float mean, float sd,
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
  const int64_t h_row = (int64_t)HY * HX * VOX, t_row = (int64_t)T3Y * T3X * VOX;
  const char *budget_env = getenv("FPL_VGG_SCRATCH_MB");
  const int64_t budget = budget_env ? (int64_t)atoll(budget_env) << 20 : (int64_t)64 << 30;
  // H1 has 2 (cz + 2) + 4 rows, L3 two fewer
  int64_t cz_chunk = std::max<int64_t>(4, (budget / (h_row + t_row) - 8) / 2);
  cz_chunk = std::min<int64_t>(cz_chunk, cz_hi - cz_lo);
  cz_chunk = (cz_chunk + 3) / 4 * 4;
  FPL_REQUIRE(ctx, (int64_t)HY * HX * PASS_BYTES * 8 < ((int64_t)1 << 32),
              "vgg_like2 split path: a %lld x %lld plane is too large for the tile loader's 32-bit "
              "offsets", (long long)SY, (long long)SX);
  DevTemp tmp(ctx);
  unsigned *flag;
  FPL_TRY(fpl_range_flag(ctx, &flag));
  void *h1v, *l3v, *qv;
  FPL_TRY(tmp.alloc((size_t)(2 * cz_chunk + 8) * h_row, &h1v));
  FPL_TRY(tmp.alloc((size_t)(2 * cz_chunk + 6) * t_row, &l3v));
  FPL_TRY(tmp.alloc((size_t)(cz_chunk + 2) * QY * QX * VOX, &qv));
  const unsigned char *F = st->frags;
  const float *S = st->shifts;
  for (int64_t c0 = cz_lo; c0 < cz_hi; c0 += cz_chunk) {
    const int CZ = (int)std::min<int64_t>(cz_chunk, cz_hi - c0);
    const int QZ = CZ + 2, T3Z = 2 * QZ + 2, HZ = T3Z + 2;
    {
      V2SArgs a = {};
      a.src = src; a.SZ = SZ; a.SY = SY; a.SX = SX; a.mean = mean; a.sd = sd;
      a.gz0 = 4 * c0;
      a.wstem = (const h16x8 *)(F + st->off_w[0]); a.shstem = S + st->off_s[0];
      if (src_dtype == FPL_U8) {
        a.wstem = (const h16x8 *)st->w1_int; a.shtab = st->shift1_int; a.c0 = st->int_c0;
      }
      a.w = F + st->off_w[1]; a.shift = S + st->off_s[1];
      a.out = (unsigned char *)h1v; a.OZ = HZ; a.OY = HY; a.OX = HX;
      a.flag = flag; a.xlim = st->xlim;
      a.bg = BlockGrid{(int)ceil_div64(HX, 8), (int)ceil_div64(HY, 2), (int)ceil_div64(HZ, 2)};
      TimedLaunch tl(ctx, "vggs2_stem_conv3_pool");
      if (src_dtype == FPL_U8)
        vggs2_conv3<true, true, uint8_t><<<brick_grid_size(a.bg), 256, V2_SMEM, stream>>>(a);
      else
        vggs2_conv3<true, true, float><<<brick_grid_size(a.bg), 256, V2_SMEM, stream>>>(a);
    }
    {
      V2SArgs a = {};
      a.in = (const unsigned char *)h1v; a.IZ = HZ; a.IY = HY; a.IX = HX;
      a.w = F + st->off_w[2]; a.shift = S + st->off_s[2];
      a.out = (unsigned char *)l3v; a.OZ = T3Z; a.OY = T3Y; a.OX = T3X;
      a.flag = flag;
      a.bg = BlockGrid{(int)ceil_div64(T3X, 16), (int)ceil_div64(T3Y, 4), (int)ceil_div64(T3Z, 4)};
      TimedLaunch tl(ctx, "vggs2_conv3");
      vggs2_conv3<false, false, uint8_t><<<brick_grid_size(a.bg), 256, V2_SMEM, stream>>>(a);
    }
    {
      V2SArgs a = {};
      a.in = (const unsigned char *)l3v; a.IZ = T3Z; a.IY = T3Y; a.IX = T3X;
      a.w = F + st->off_w[3]; a.shift = S + st->off_s[3];
      a.out = (unsigned char *)qv; a.OZ = QZ; a.OY = QY; a.OX = QX;
      a.flag = flag;
      a.bg = BlockGrid{(int)ceil_div64(QX, 8), (int)ceil_div64(QY, 2), (int)ceil_div64(QZ, 2)};
      TimedLaunch tl(ctx, "vggs2_conv3_pool");
      vggs2_conv3<false, true, uint8_t><<<brick_grid_size(a.bg), 256, V2_SMEM, stream>>>(a);
    }
    {
      TailSArgs a;
      a.p2 = (const unsigned char *)qv; a.P2Z = QZ; a.P2Y = QY; a.P2X = QX;
      a.w5 = F + st->off_w[4]; a.shift5 = S + st->off_s[4];
      a.CZ = CZ; a.CY = CY; a.CX = CX;
      a.w6 = F + st->off_w[5]; a.w7 = F + st->off_w[6]; a.w8 = F + st->off_w[7];
      a.shift6 = S + st->off_s[5]; a.shift7 = S + st->off_s[6]; a.bias8 = st->bias8;
      a.dst = dst; a.DY = SY; a.DX = SX; a.gz0 = c0;
      a.VZ = std::min<int64_t>(fz_hi, VZ); a.VY = VY; a.VX = VX; a.off = OFF;
      a.flag = flag;
      a.bg = BlockGrid{(int)ceil_div64(CX, 16), (int)ceil_div64(CY, 4), (int)ceil_div64(CZ, 4)};
      TimedLaunch tl(ctx, "vggs_c5_tail");
      vggs_c5_tail_p24<<<brick_grid_size(a.bg), 256, M_SMEM, stream>>>(a);
    }
    FPL_HIP(ctx, hipGetLastError());
  }
  return 0;
}

}  // namespace

// Slab orchestration: as the vgg_like branch of fpl_fast_infer_volume_* (vgg_fused.hip;
// the lattice equivalence with FplNetwork.infer, flypylib/fplnetwork.py:146-187, is
// argued there), with P1 / P2 in the split layout.
// Row pitch (voxels of 16 B) of an x8 tensor of X voxels per row.  (Pitches padded by 14 - 126
// voxels were measured on the 520^3 volume - 4128-B rows - and changed nothing: what made that
// size slow was the walk order, vgg_split_lds.h::Cursor.)
static int x8_pitch(int X) { return X; }

int fpl_split_infer_volume(fpl_ctx *ctx, fpl_program *prog, const void *src, int src_dtype,
                           float mean, float sd, const int64_t dims[3],
                           const std::vector<int32_t> origins[3], const int32_t out_sz[3],
                           int32_t zb, int32_t ze, float *dst) {
  SplitState *st;
  FPL_TRY(split_prepare(ctx, prog, &st));
  if (src_dtype == FPL_U8) FPL_TRY(split_prepare_int(ctx, prog, st, mean, sd));
  if (fpl_vgg_variant(prog) == 2)
    return split2_infer(ctx, st, src, src_dtype, mean, sd, dims, origins, out_sz, zb, ze, dst);
  hipStream_t stream = ctx->stream;
  const int64_t SZ = dims[0], SY = dims[1], SX = dims[2];
  const int64_t VZ = SZ - 14, VY = SY - 14, VX = SX - 14;
  if (VZ <= 0 || VY <= 0 || VX <= 0 || zb >= ze) return 0;   // no valid output voxel
  // coarse rows this slab owns (tile rows zb..ze-1 of the reference lattice)
  const int64_t fz_lo = (int64_t)origins[0][zb] - 7;
  const int64_t fz_hi = std::min<int64_t>((int64_t)origins[0][ze - 1] - 7 + out_sz[0], VZ);
  const int64_t cz_lo = fz_lo / 4, cz_hi = ceil_div64(fz_hi, 4);
  const int CY = (int)ceil_div64(VY, 4), CX = (int)ceil_div64(VX, 4);
  const int P2Y = CY + 2, P2X = CX + 2, P1Y = 2 * P2Y + 2, P1X = 2 * P2X + 2;
  const int P1XP = x8_pitch(P1X), P2XP = x8_pitch(P2X);       // row pitches (voxels)
  // chunk of coarse rows bounded by a scratch budget (P1 dominates;
  // FPL_VGG_SCRATCH_MB shrinks it so that tests can force several chunks)
  const int64_t p1_row_bytes = (int64_t)P1Y * P1XP * VOX;
  const char *budget_env = getenv("FPL_VGG_SCRATCH_MB");
  const int64_t budget = budget_env ? (int64_t)atoll(budget_env) << 20 : (int64_t)64 << 30;
  int64_t cz_chunk = std::max<int64_t>(4, (budget / p1_row_bytes - 6) / 2);
  cz_chunk = std::min<int64_t>(cz_chunk, cz_hi - cz_lo);
  cz_chunk = (cz_chunk + 3) / 4 * 4;
  FPL_REQUIRE(ctx, (int64_t)S_TZ * SY * SX < ((int64_t)1 << 31),
              "vgg split path: a %lld x %lld plane is too large for the stem's 31-bit row "
              "offsets", (long long)SY, (long long)SX);
  // the tile loads address a pass's hi AND lo plane from one scalar base with 32-bit lane
  // offsets: two part planes of the chunk plus the tile's reach stay below 4 GiB
  {
    const int64_t max_rows = (((int64_t)1 << 32) / 16 / ((int64_t)P1Y * P1XP) - (x8::TZ + 2));
    FPL_REQUIRE(ctx, max_rows >= 14,
                "vgg split path: a %lld x %lld plane is too large for the tile loader's 32-bit "
                "offsets", (long long)SY, (long long)SX);
    cz_chunk = std::min<int64_t>(cz_chunk, std::max<int64_t>(4, ((max_rows - 6) / 2) / 4 * 4));
  }
  DevTemp tmp(ctx);
  unsigned *flag;
  FPL_TRY(fpl_range_flag(ctx, &flag));
  // P1 / P2 as planes of 8-channel passes (vgg_split_lds.h) with read slack behind them
  x8::Tensor p1 = {nullptr, (int)(2 * cz_chunk + 6), P1Y, P1X, P1XP}, p2 = {nullptr, (int)(cz_chunk + 2), P2Y, P2X, P2XP};
  void *p1v, *p2v;
  FPL_TRY(tmp.alloc((size_t)(p1.bytes() + p1.slack_bytes()), &p1v));
  FPL_TRY(tmp.alloc((size_t)(p2.bytes() + p2.slack_bytes()), &p2v));
  void *dumpv;
  FPL_TRY(tmp.alloc(256, &dumpv));
  // workgroups of the persistent kernels: one per CU, a multiple of the 8 XCDs
  const unsigned pgrid = (unsigned)std::max(8, ctx->n_cu / 8 * 8);
  const unsigned char *F = st->frags;
  const float *S = st->shifts;
  for (int64_t c0 = cz_lo; c0 < cz_hi; c0 += cz_chunk) {
    const int CZ = (int)std::min<int64_t>(cz_chunk, cz_hi - c0);
    const int P2Z = CZ + 2, P1Z = 2 * P2Z + 2;
    p1 = x8::Tensor{(unsigned char *)p1v, P1Z, P1Y, P1X, P1XP};
    p2 = x8::Tensor{(unsigned char *)p2v, P2Z, P2Y, P2X, P2XP};
    // edge tiles read up to TZ planes past the last pass plane: zeros, not stale scratch
    FPL_HIP(ctx, hipMemsetAsync(p1.p + p1.bytes(), 0, (size_t)p1.slack_bytes(), stream));
    FPL_HIP(ctx, hipMemsetAsync(p2.p + p2.bytes(), 0, (size_t)p2.slack_bytes(), stream));
    {
      StemSArgs a;
      a.src = src; a.SZ = SZ; a.SY = SY; a.SX = SX; a.mean = mean; a.sd = sd;
      a.p1z0 = 2 * c0;
      // block rounding may reach past the rows this slab stages: they only feed masked
      // outputs, so they read as zero
      a.z_hi = std::min<int64_t>(SZ, 4 * (c0 + CZ) + 14);
      a.w1 = (const h16x8 *)(F + st->off_w[0]);
      a.w2 = (const h16x8 *)(F + st->off_w[1]);
      a.shift1 = S + st->off_s[0]; a.shift2 = S + st->off_s[1];
      a.c0 = 0.f; a.shtab = nullptr;
      if (src_dtype == FPL_U8) {
        a.w1 = (const h16x8 *)st->w1_int; a.shtab = st->shift1_int;
        a.c0 = st->int_c0;
      }
      a.p1 = p1;
      a.flag = flag; a.xlim = st->xlim; a.dump = (unsigned char *)dumpv;
      a.nbx = (int)ceil_div64(P1X, S_PX); a.nby = (int)ceil_div64(P1Y, S_PY);
      a.nbz = (int)ceil_div64(P1Z, S_PZ);
      // persistent: one workgroup of 8 waves per CU walks the blocks
      const unsigned grid = (unsigned)std::min<int64_t>((int64_t)a.nbx * a.nby * a.nbz,
                                                        (int64_t)ctx->n_cu);
      TimedLaunch tl(ctx, "vggs_stem_pool");
      if (src_dtype == FPL_U8)
        vggs_stem_pool<uint8_t><<<grid, 64 * S_WAVES, S_SMEM, stream>>>(a);
      else
        vggs_stem_pool<float><<<grid, 64 * S_WAVES, S_SMEM, stream>>>(a);
    }
    {
      MidXArgs a;
      a.p1 = p1;
      a.w3 = F + st->off_w[2];
      a.w4 = (const h16x8 *)(F + st->off_w[3]);
      a.shift3 = S + st->off_s[2]; a.shift4 = S + st->off_s[3];
      a.p2 = p2;
      a.flag = flag;
      a.walk = x8::Walk{(int)ceil_div64(P2X, 8), (int)ceil_div64(P2Y, 2), (int)ceil_div64(P2Z, 4)};
      TimedLaunch tl(ctx, "vggs_mid_pool");
      vggs_mid_pool<<<pgrid, 64 * x8::WAVES, x8::SMEM, stream>>>(a);
    }
    {
      TailXArgs a;
      a.p2 = p2;
      a.w5 = F + st->off_w[4]; a.shift5 = S + st->off_s[4];
      a.CZ = CZ; a.CY = CY; a.CX = CX;
      a.w6 = F + st->off_w[5]; a.w7 = F + st->off_w[6]; a.w8 = F + st->off_w[7];
      a.shift6 = S + st->off_s[5]; a.shift7 = S + st->off_s[6]; a.bias8 = st->bias8;
      a.dst = dst; a.DY = SY; a.DX = SX; a.gz0 = c0;
      a.VZ = std::min<int64_t>(fz_hi, VZ); a.VY = VY; a.VX = VX; a.off = 7;
      a.flag = flag;
      a.walk = x8::Walk{(int)ceil_div64(CX, 16), (int)ceil_div64(CY, 4), (int)ceil_div64(CZ, 8)};
      TimedLaunch tl(ctx, "vggs_c5_tail");
      vggs_c5_tail<<<pgrid, 64 * x8::WAVES, x8::SMEM, stream>>>(a);
    }
    FPL_HIP(ctx, hipGetLastError());
  }
  return 0;
}
