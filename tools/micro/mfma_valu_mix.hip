// Micro-benchmark: how much VALU work rides for free next to f16 MFMAs of the two shapes
// on gfx950 (two waves per SIMD).  Per "unit" of 32 voxels of the vgg stem: shape A =
// 18 x v_mfma_f32_16x16x32_f16 (9 per 16 voxels), shape B = 10 x v_mfma_f32_32x32x16_f16
// (the 48 -> 64 channel padding included); each with NV independent VALU instructions
// (conversions of accumulators, max3) interleaved by the compiler.
// hipcc --offload-arch=gfx950 -O3 mfma_valu_mix.hip -o mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned cvt2(float a, float b) {
  f2 f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, h2));
}

template <int SHAPE, int NV>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters, float seed) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i * seed); b[i] = (_Float16)(1.0f + i * 0.01f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i;
  unsigned sink = 0;
  if (SHAPE == 0) {
    f4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 18; ++m) {
        acc[m % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m % 6], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < (NV * (m + 1)) / 18 - (NV * m) / 18; ++j) {
          const int q = (m + j) % 8;
          if ((m + j) & 1) sink ^= cvt2(v[q], v[(q + 1) % 8]);
          else v[q] = __builtin_fmaxf(__builtin_fmaxf(v[q], v[(q + 3) % 8]), v[(q + 5) % 8] + 1.f);
        }
      }
    }
    float s = 0;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s + v[0] + v[7] + (float)sink;
  } else {
    f16v acc[3];
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 10; ++m) {
        acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % 3], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < (NV * (m + 1)) / 10 - (NV * m) / 10; ++j) {
          const int q = (m + j) % 8;
          if ((m + j) & 1) sink ^= cvt2(v[q], v[(q + 1) % 8]);
          else v[q] = __builtin_fmaxf(__builtin_fmaxf(v[q], v[(q + 3) % 8]), v[(q + 5) % 8] + 1.f);
        }
      }
    }
    float s = 0;
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s + v[0] + v[7] + (float)sink;
  }
}

template <int SHAPE, int NV>
void run(float *d, const char *name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, blocks = 512;                 // 2 workgroups of 4 waves per CU
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    k<SHAPE, NV><<<blocks, 256>>>(d, iters, 1.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep == 0) continue;
    // units of 32 voxels per SIMD: 2 waves per SIMD
    const double units_per_simd = 2.0 * iters;
    printf("%-10s NV=%2d per 32 voxels: %.3f ms  %.1f ns per unit per SIMD (x2.0 GHz = %.0f cycles)\n",
           name, NV, ms, ms * 1e6 / units_per_simd, ms * 1e6 / units_per_simd * 2.0);
  }
}

int main() {
  float *d; hipMalloc(&d, 512 * 256 * 4);
  run<0, 0>(d, "16x16x32"); run<0, 24>(d, "16x16x32"); run<0, 36>(d, "16x16x32"); run<0, 50>(d, "16x16x32"); run<0, 64>(d, "16x16x32");
  run<1, 0>(d, "32x32x16"); run<1, 24>(d, "32x32x16"); run<1, 36>(d, "32x32x16"); run<1, 50>(d, "32x32x16"); run<1, 64>(d, "32x32x16");
  return 0;
}
