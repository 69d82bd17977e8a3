// CDNA4 (gfx950) MFMA helpers shared by the fused kernels.
//
// v_mfma_f32_16x16x32_bf16 fragment maps (lane l: c = l & 15, g = l >> 4):
//   A[row c][k = 8g + j]   j = 0..7   (8 bf16 = 4 VGPRs)
//   B[k = 8g + j][col c]
//   D[row = 4g + r][col c] r = 0..3   (4 f32)
// Here A = folded weights^T (rows = output channels), B = activations
// (cols = 16 voxels), so lane (c, g) ends up owning output channels 16b+4g+r of
// voxel c for M-block b.  Feeding D back as the next layer's B operand needs no
// lane movement: k-slot (s, g, j) of the next layer is bound to channel
//   16*(2s + (j>>2)) + 4g + (j&3)
// and the weight packer (pack_weights.h) lays the A fragments out to match.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// 16-bit operand type of the fused kernels.  Each fused translation unit is built
// twice (csrc/build.py): bfloat16 (default; FPL_PREC_BF16) and, with -DFPL_F16, IEEE
// half (FPL_PREC_F16).  Same MFMA rate; half keeps 11 significant bits instead of 8
// (probabilities within ~1e-4 of fp32 instead of ~1e-3) at a 65504 range, which the
// weight packer checks.  FPLK(name) = name_bf16 / name_f16 keeps the two builds'
// kernels and entry points apart.
// A third build of conv_mfma.hip (-DFPL_F16 -DFPL_SPLIT, csrc/build.py) runs the U-Net
// executor on SPLIT operands (FPL_PREC_F16S): v = hi + lo, two IEEE halves; kernels and
// entry points carry the suffix _f16s.
#if defined(FPL_F16) && defined(FPL_SPLIT)
typedef _Float16 h16_t;
#define FPLK(name) name##_f16s
#define FPL_PREC_STR "f16s"
#define FPL_THIS_PREC FPL_PREC_F16S
#define FPL_H16_SLOT 2
#elif defined(FPL_F16)
typedef _Float16 h16_t;
#define FPLK(name) name##_f16
#define FPL_PREC_STR "f16"
#define FPL_THIS_PREC FPL_PREC_F16
#define FPL_H16_SLOT 1
#else
typedef __bf16 h16_t;
#define FPLK(name) name##_bf16
#define FPL_PREC_STR "bf16"
#define FPL_THIS_PREC FPL_PREC_BF16
#define FPL_H16_SLOT 0
#endif
typedef h16_t h16x8 __attribute__((ext_vector_type(8)));
typedef h16_t h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));   // 16-B store at an 8-B aligned address

__device__ __forceinline__ f32x4 mfma16(h16x8 a, h16x8 b, f32x4 c) {
#ifdef FPL_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

typedef h16_t h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// two f32 -> packed 16-bit pair, RNE (one v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
__device__ __forceinline__ unsigned cvt_pk_h16(float a, float b) {
  f32x2 f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, h16x2));
}

// max of packed 16-bit float pairs AS SIGNED INT16 (one v_pk_max_i16).  Against 0 this
// is ReLU on both halves (negative float16/bfloat16 = negative int16, -0.0 included);
// between non-negative values it is the float max - which is all a max-pool of
// post-ReLU activations needs.
__device__ __forceinline__ unsigned pk_max_i16(unsigned a, unsigned b) {
  return __builtin_bit_cast(
      unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a),
                                          __builtin_bit_cast(s16x2, b)));
}

// two accumulator tiles (M-blocks 2s and 2s+1 of the previous layer) -> the B
// fragment of K-step s of the next layer, ReLU applied: 4 cvt_pk + 4 pk_max
__device__ __forceinline__ h16x8 pack_relu(const f32x4 &lo, const f32x4 &hi) {
  u32x4 v;
  v[0] = pk_max_i16(cvt_pk_h16(lo[0], lo[1]), 0u);
  v[1] = pk_max_i16(cvt_pk_h16(lo[2], lo[3]), 0u);
  v[2] = pk_max_i16(cvt_pk_h16(hi[0], hi[1]), 0u);
  v[3] = pk_max_i16(cvt_pk_h16(hi[2], hi[3]), 0u);
  return __builtin_bit_cast(h16x8, v);
}

#ifdef FPL_F16
// ReLU + conversion in ONE instruction for activations known to lie below 1
// (v_cvt_pk_f16_f32 ... clamp: the compiler folds the [0, 1] clamp of the packed halves
// into the conversion's output modifier).  The caller guarantees the upper bound by
// scaling the producing layer by a power of two (exact) and the consuming layer's
// weights by its inverse - see vgg_prepare.  No bfloat16 form: that conversion has no
// clamp modifier.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_relu01(float a, float b) {
  f32x2 f = {a, b};
  f16x2_t h = __builtin_convertvector(f, f16x2_t);
  const f16x2_t z = {(_Float16)0.f, (_Float16)0.f}, o = {(_Float16)1.f, (_Float16)1.f};
  h = __builtin_elementwise_min(__builtin_elementwise_max(h, z), o);
  return __builtin_bit_cast(unsigned, h);
}
#else
__device__ __forceinline__ unsigned cvt_pk_relu01(float a, float b) {
  return pk_max_i16(cvt_pk_h16(a, b), 0u);
}
#endif

// pack_relu / pack_relu_lo with the one-instruction form (inputs < 1 by construction)
template <bool CLAMP01>
__device__ __forceinline__ h16x8 pack_relu_t(const f32x4 &lo, const f32x4 &hi) {
  u32x4 v;
  if (CLAMP01) {
    v[0] = cvt_pk_relu01(lo[0], lo[1]); v[1] = cvt_pk_relu01(lo[2], lo[3]);
    v[2] = cvt_pk_relu01(hi[0], hi[1]); v[3] = cvt_pk_relu01(hi[2], hi[3]);
  } else {
    v[0] = pk_max_i16(cvt_pk_h16(lo[0], lo[1]), 0u); v[1] = pk_max_i16(cvt_pk_h16(lo[2], lo[3]), 0u);
    v[2] = pk_max_i16(cvt_pk_h16(hi[0], hi[1]), 0u); v[3] = pk_max_i16(cvt_pk_h16(hi[2], hi[3]), 0u);
  }
  return __builtin_bit_cast(h16x8, v);
}
template <bool CLAMP01>
__device__ __forceinline__ h16x8 pack_relu_lo_t(const f32x4 &lo) {
  u32x4 v;
  if (CLAMP01) {
    v[0] = cvt_pk_relu01(lo[0], lo[1]); v[1] = cvt_pk_relu01(lo[2], lo[3]);
  } else {
    v[0] = pk_max_i16(cvt_pk_h16(lo[0], lo[1]), 0u); v[1] = pk_max_i16(cvt_pk_h16(lo[2], lo[3]), 0u);
  }
  v[2] = 0u;
  v[3] = 0u;
  return __builtin_bit_cast(h16x8, v);
}

// same with the upper block missing (48 channels = 3 blocks): zeros
__device__ __forceinline__ h16x8 pack_relu_lo(const f32x4 &lo) {
  u32x4 v;
  v[0] = pk_max_i16(cvt_pk_h16(lo[0], lo[1]), 0u);
  v[1] = pk_max_i16(cvt_pk_h16(lo[2], lo[3]), 0u);
  v[2] = 0u;
  v[3] = 0u;
  return __builtin_bit_cast(h16x8, v);
}

// running max-pool of relu(acc) in packed bf16: pooled = max(pooled, bf16(acc))
// with pooled initialised to 0 (rounding is monotonic, so this equals rounding
// the fp32 max)
__device__ __forceinline__ void pool_relu_h16(u32x2 &pooled, const f32x4 &acc) {
  pooled[0] = pk_max_i16(pooled[0], cvt_pk_h16(acc[0], acc[1]));
  pooled[1] = pk_max_i16(pooled[1], cvt_pk_h16(acc[2], acc[3]));
}

__device__ __forceinline__ unsigned short h16_bits(float f) {
  h16_t b = (h16_t)f;
  return __builtin_bit_cast(unsigned short, b);
}

// ---- split operands (FPL_PREC_F16S): v ~ hi + lo, hi = half(v), lo = half(v - hi) -------
// ~22 significant bits between the two halves (v - hi is exact in fp32; lo may be a
// subnormal half, which v_mfma_f32_16x16x32_f16 keeps: tools/micro/mfma_denorm.hip).
struct Pair2 { unsigned hi, lo; };          // two values as packed halves
struct Frag2 { h16x8 hi, lo; };             // a B fragment

#ifdef FPL_F16
// (a, b) -> packed hi halves and packed lo halves.  lo = half(v - hi) is one
// v_fma_mix{lo,hi}_f16 per value: fma(hi as f16, -1, v) is exact in fp32, rounded once to
// the half it writes (hipcc emits two conversions back, a packed subtract and a packed
// conversion for the same arithmetic - 4 instructions instead of 2 in kernels whose VALU
// issue is what the MFMAs wait for).
// The result is written INTO THE REGISTER OF `a` ("+v").  hipcc's hazard recognizer does not
// look inside an asm statement: given a free output register it may pick one that is the dead
// part of the destination tuple of an MFMA still in flight (only element 0 of a 4-register
// result live, say), and that MFMA's late write-back then lands on top of the asm's result -
// seen in round 4 as run-to-run differences of 1e-5 in the head kernel, wherever the
// allocator happened to make that choice.  `a` is always the result of an ordinary VALU
// instruction (a ReLU or a pool maximum) whose own write the compiler did guard, and no MFMA
// can have a live register in its destination, so reusing it is safe by construction.
// THE one place this asm lives (tests/test_host_logic.py checks on the device assembly of every
// split object that no v_fma_mix destination was last written by an MFMA).  "+&v": the first
// instruction writes %0 before the second reads %2, so %0 must not share a register with an
// input even when a and b are the same value.
__device__ __forceinline__ unsigned split_lo_pk(float a, unsigned hi_pair, float b) {
  unsigned lo = __builtin_bit_cast(unsigned, a);
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "+&v"(lo)
      : "v"(hi_pair), "v"(b));
  return lo;
}
__device__ __forceinline__ Pair2 split_pk(float a, float b) {
  Pair2 r;
  r.hi = cvt_pk_h16(a, b);
  r.lo = split_lo_pk(a, r.hi, b);
  return r;
}
#else
__device__ __forceinline__ Pair2 split_pk(float a, float b) {   // no split form on bfloat16
  return Pair2{cvt_pk_h16(a, b), 0u};
}
#endif
// ReLU as an integer max with 0 (negative floats are negative integers; -0.0 included):
// one v_max_i32, where fmaxf costs a canonicalising v_max_f32 more per value
__device__ __forceinline__ float relu_f32(float a) {
  const int i = __builtin_bit_cast(int, a);
  return __builtin_bit_cast(float, i > 0 ? i : 0);
}
__device__ __forceinline__ Pair2 split_pk_relu(float a, float b) {
  return split_pk(relu_f32(a), relu_f32(b));
}

// ---- half-range guard (FPL_PREC_F16S / FPL_PREC_AUTO; the reference predicts in fp32,
// flypylib/fplnetwork.py:175-176, which has no such limit).  A value >= 65520 becomes
// hi = inf, lo = half(v - inf) = -inf, and the next layer's accumulators NaN or, through a
// max-pool or a ReLU, silently wrong finite numbers.  Every value these kernels split is
// non-negative (post-ReLU / post-pool; -0.0 is a negative int16), so "some hi half is inf or
// NaN" is "the running int16 maximum of the packed hi halves reaches 0x7C00": one
// v_pk_max_i16 per split pair into a per-thread word, looked at once at the end of the
// kernel (ovf_commit: an atomic OR into the context's flag word, which the host reads
// after the call; 'auto' then reruns the call on the fp32 executor, 'f16s' fails).
// The layers whose splits sit in VALU-bound inner loops (the stems' conv3 1->C) are covered
// by a host-side bound on their outputs instead (sum |w| * the input limit, checked per
// loaded input value where the input is not uint8).
__device__ __forceinline__ void ovf_note(unsigned &ovf, unsigned hi_pair) { ovf = pk_max_i16(ovf, hi_pair); }
__device__ __forceinline__ bool ovf_hit(unsigned ovf) {
  return (ovf & 0x7C00u) == 0x7C00u || (ovf & 0x7C000000u) == 0x7C000000u;
}
__device__ __forceinline__ void ovf_commit(unsigned ovf, unsigned *flag, unsigned bit) {
  if (ovf_hit(ovf)) atomicOr(flag, bit);
}
// (flag bits FPL_RANGE_*: common.h)

__device__ __forceinline__ Pair2 split_pk(float a, float b, unsigned &ovf) {
  const Pair2 r = split_pk(a, b);
  ovf_note(ovf, r.hi);
  return r;
}
__device__ __forceinline__ Pair2 split_pk_relu(float a, float b, unsigned &ovf) {
  return split_pk(relu_f32(a), relu_f32(b), ovf);
}
// signed values straight from an accumulator (a convolution without ReLU: none of the
// reference's graphs has one in front of a split tensor): plain C, no asm on an MFMA result;
// the guard looks at |hi|
__device__ __forceinline__ Pair2 split_pk_signed(float a, float b, unsigned &ovf) {
  Pair2 r;
  r.hi = cvt_pk_h16(a, b);
  const h16x2 h = __builtin_bit_cast(h16x2, r.hi);
  r.lo = cvt_pk_h16(a - (float)h[0], b - (float)h[1]);
  ovf_note(ovf, r.hi & 0x7FFF7FFFu);
  return r;
}

// two accumulator tiles -> the hi and lo B fragments of a K-step of the next layer, ReLU
// applied (chain / spatial maps as pack_relu)
__device__ __forceinline__ Frag2 pack_relu_split(const f32x4 &lo_blk, const f32x4 &hi_blk) {
  u32x4 h, l;
  Pair2 p;
  p = split_pk_relu(lo_blk[0], lo_blk[1]); h[0] = p.hi; l[0] = p.lo;
  p = split_pk_relu(lo_blk[2], lo_blk[3]); h[1] = p.hi; l[1] = p.lo;
  p = split_pk_relu(hi_blk[0], hi_blk[1]); h[2] = p.hi; l[2] = p.lo;
  p = split_pk_relu(hi_blk[2], hi_blk[3]); h[3] = p.hi; l[3] = p.lo;
  Frag2 f;
  f.hi = __builtin_bit_cast(h16x8, h);
  f.lo = __builtin_bit_cast(h16x8, l);
  return f;
}
// ... with the half-range guard: the four hi words are already packed, so four v_pk_max
__device__ __forceinline__ Frag2 pack_relu_split(const f32x4 &lo_blk, const f32x4 &hi_blk, unsigned &ovf) {
  const Frag2 f = pack_relu_split(lo_blk, hi_blk);
  const u32x4 h = __builtin_bit_cast(u32x4, f.hi);
  ovf_note(ovf, pk_max_i16(pk_max_i16(h[0], h[1]), pk_max_i16(h[2], h[3])));
  return f;
}

// acc += (w_hi + w_lo)(b_hi + b_lo) without the lo x lo product
__device__ __forceinline__ f32x4 mfma3(h16x8 wh, h16x8 wl, const Frag2 &b, f32x4 acc) {
  acc = mfma16(wl, b.hi, acc);
  acc = mfma16(wh, b.lo, acc);
  return mfma16(wh, b.hi, acc);
}
