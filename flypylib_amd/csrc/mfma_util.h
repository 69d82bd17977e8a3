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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// single-instruction ReLU (fmaxf() would add a canonicalising v_max per value)
__device__ __forceinline__ float relu1(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}

__device__ __forceinline__ float max1(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// two accumulator tiles (M-blocks 2s and 2s+1) -> one B fragment of K-step s,
// with ReLU; `hi_valid` = false packs zeros for a missing block
__device__ __forceinline__ bf16x8 pack_relu(const f32x4 &lo, const f32x4 &hi) {
  bf16x8 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r] = (__bf16)relu1(lo[r]);
    v[4 + r] = (__bf16)relu1(hi[r]);
  }
  return v;
}

__device__ __forceinline__ bf16x8 pack_relu_lo(const f32x4 &lo) {
  bf16x8 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r] = (__bf16)relu1(lo[r]);
    v[4 + r] = (__bf16)0.0f;
  }
  return v;
}

__device__ __forceinline__ unsigned short bf16_bits(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
