// Host-side packing of folded conv weights into MFMA A-fragment order (bf16 or, with
// -DFPL_F16, IEEE half).  One fragment = 64 lanes x 8 x 16 bit = 1 KiB, stored [kstep][mblock][lane][8] so a
// wave fetches a fragment with one coalesced 16-B-per-lane load (or one
// ds_read_b128 per lane from an LDS copy).  Slot maps: see mfma_util.h.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

enum FplSlotMap {
  SLOT_STEM = 0,     // cin = 1: k-slot f = 32s + 8g + j is tap f
  SLOT_SPATIAL = 1,  // k-slot f -> (tap = f / cin, memory channel = f % cin)
  SLOT_CHAIN = 2     // k-slot (s,g,j) -> channel 16(2s + (j>>2)) + 4g + (j&3)
};

// float -> the 16-bit operand type of this build (mfma_util.h), round to nearest even
#ifdef FPL_F16
static inline uint16_t fpl_f32_to_h16(float f) {
  const _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
constexpr float FPL_H16_MAX = 65504.f;
#else
static inline uint16_t fpl_f32_to_h16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
constexpr float FPL_H16_MAX = 3.3e38f;
#endif

// Split operands (vgg_split.hip, IEEE-half build only): a value v is carried as
// hi = half(v) and lo = half(v - hi), ~22 significant bits between them (v - hi is exact
// in fp32; lo may be a subnormal half, which v_mfma_f32_16x16x32_f16 does not flush:
// tools/micro/mfma_denorm.hip).  part 0 = hi (the plain 16-bit operand), 1 = lo.
static inline uint16_t fpl_f32_to_h16_part(float v, int part) {
  const uint16_t hi = fpl_f32_to_h16(v);
  if (part == 0) return hi;
#ifdef FPL_F16
  _Float16 h;
  memcpy(&h, &hi, 2);
  return fpl_f32_to_h16(v - (float)h);
#else
  return 0;                                  // no split form on bfloat16 operands
#endif
}

// Output channel computed by row m of M-block b.  Plain: 16b + m.  Interleaved
// (il): 4*MB*(m/4) + 4b + m%4 - lane (c, g) of the accumulators (rows 4g..4g+3 of
// every block) then owns the 4*MB CONTIGUOUS channels [4*MB*g, 4*MB*(g+1)) of its
// voxel, so an epilogue stores 16-B pieces and a wave writes whole lines.
// il = 2 (48 channels, 3 M-blocks; the x8 tensors of vgg_split_lds.h, passes of 8 channels):
// blocks 0 and 1 of lane group g are the 8 channels of pass g, block 2 the half
// [32 + 4 g, 32 + 4 g + 4) of pass 4 + g / 2 - a lane stores one whole 16-B pass voxel and one
// 8-B half, the same way in every lane.
static inline int fpl_out_channel(int b, int m, int n_mblocks, int il) {
  if (il == 2) return b < 2 ? 8 * (m >> 2) + 4 * b + (m & 3) : 32 + m;
  return il ? 4 * n_mblocks * (m >> 2) + 4 * b + (m & 3) : 16 * b + m;
}

// W: [ntaps * cin][cout] fp32 (Keras memory order), scale[cout] folded in.
// out: n_ksteps * n_mblocks fragments.
static inline void fpl_pack_frags(const float *W, const float *scale, int ntaps,
                                  int cin, int cout, int n_mblocks, int n_ksteps,
                                  FplSlotMap map, std::vector<uint16_t> *out,
                                  int il = 0, int part = 0) {
  out->assign((size_t)n_ksteps * n_mblocks * 512, 0);
  for (int s = 0; s < n_ksteps; ++s)
    for (int b = 0; b < n_mblocks; ++b)
      for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 15, g = lane >> 4;
        const int co = fpl_out_channel(b, m, n_mblocks, il);
        if (co >= cout) continue;
        for (int j = 0; j < 8; ++j) {
          int kidx = -1;
          if (map == SLOT_CHAIN) {
            const int c = 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3);
            if (c < cin) kidx = c;
          } else {
            const int f = 32 * s + 8 * g + j;
            const int tap = f / cin, ch = f % cin;
            if (tap < ntaps) kidx = tap * cin + ch;
          }
          if (kidx < 0) continue;
          const float v = W[(size_t)kidx * cout + co] * scale[co];
          (*out)[(((size_t)s * n_mblocks + b) * 64 + lane) * 8 + j] =
              fpl_f32_to_h16_part(v, part);
        }
      }
}

// One K-step of a register-chained 1x1 conv whose two k-slot halves are chosen freely:
// k-slot (g, j) = channel 16 * blk[j >> 2] + 4g + (j & 3) of the previous layer, weights
// taken as part[j >> 2] (hi / lo, fpl_f32_to_h16_part).  The split kernels use it for the
// last, half-empty K-step of a 48-channel input: [block 2 hi | block 2 lo] against
// [w_lo | w_hi] gives both cross products in ONE MFMA, and against [w_hi | w_lo] the
// hi x hi (and, for free, lo x lo) product.  Appends n_mblocks fragments to *out.
static inline void fpl_pack_chain_step(const float *W, const float *scale, int cin, int cout,
                                       int n_mblocks, const int blk[2], const int part[2],
                                       std::vector<uint16_t> *out, int il = 0) {
  const size_t base = out->size();
  out->resize(base + (size_t)n_mblocks * 512, 0);
  for (int b = 0; b < n_mblocks; ++b)
    for (int lane = 0; lane < 64; ++lane) {
      const int m = lane & 15, g = lane >> 4, co = fpl_out_channel(b, m, n_mblocks, il);
      if (co >= cout) continue;
      for (int j = 0; j < 8; ++j) {
        const int c = 16 * blk[j >> 2] + 4 * g + (j & 3);
        if (c >= cin) continue;
        (*out)[base + (((size_t)b * 64 + lane) * 8 + j)] =
            fpl_f32_to_h16_part(W[(size_t)c * cout + co] * scale[co], part[j >> 2]);
      }
    }
}

// ---- stem (conv3 1->48) k-slot layout ------------------------------------------
// The input tile sits in LDS as bf16; a lane's base x is even, so per tap row
// (tz,ty) one (tx,tx+1) pair is an aligned 32-bit read.  For sub-step parity
// e = dx the three taps of a row sit at element offsets e, e+1, e+2:
//   e = 0: aligned pair (tx0,tx1) at +0, single tx2 at +2
//   e = 1: single tx0 at +1,            aligned pair (tx1,tx2) at +2
// Per lane the 8 k-slots are 3 pair registers + 1 register of two singles:
//   g = 0..2: pairs of rows 3g, 3g+1, 3g+2;  singles of rows 2g, 2g+1
//   g = 3   : singles of rows 6, 7, 8 read as aligned pairs whose other half
//             has zero weight; last register unused
// Returns the tap (0..26) bound to slot j of lane group g, or -1 (zero weight).
static inline int fpl_stem_slot_tap(int e, int g, int j) {
  const int i = j >> 1, h = j & 1;
  if (i < 3) {
    if (g < 3) return (3 * g + i) * 3 + (e == 0 ? h : 1 + h);
    const int row = 6 + i;
    if (e == 0) return h == 0 ? row * 3 + 2 : -1;
    return h == 1 ? row * 3 + 0 : -1;
  }
  if (g == 3) return -1;
  return (2 * g + h) * 3 + (e == 0 ? 2 : 0);
}

// 6 fragments [e][b]: W [27][48] fp32, scale[48]
static inline void fpl_pack_stem(const float *W, const float *scale, int cout,
                                 std::vector<uint16_t> *out, int part = 0) {
  out->assign((size_t)2 * 3 * 512, 0);
  for (int e = 0; e < 2; ++e)
    for (int b = 0; b < 3; ++b)
      for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 15, g = lane >> 4, co = 16 * b + m;
        if (co >= cout) continue;
        for (int j = 0; j < 8; ++j) {
          const int tap = fpl_stem_slot_tap(e, g, j);
          if (tap < 0) continue;
          (*out)[(((size_t)e * 3 + b) * 64 + lane) * 8 + j] =
              fpl_f32_to_h16_part(W[(size_t)tap * cout + co] * scale[co], part);
        }
      }
}
