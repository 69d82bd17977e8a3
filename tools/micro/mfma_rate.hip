// Micro-benchmark: issue rate of bf16 MFMA shapes on gfx950 (one wave per SIMD,
// independent accumulators).  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  s16x4 a4 = {1, 2, 3, 4}, b4 = {5, 6, 7, 8};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float *d; hipMalloc(&d, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int blocks = 256; blocks <= 2048; blocks *= 2)
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 1; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) k<0><<<blocks, 256>>>(d, iters); else k<1><<<blocks, 256>>>(d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mfma = (double)blocks * 4 * iters * 8;
      const double flop = mfma * (mode == 0 ? 16384.0 : 8192.0);
      printf("blocks %d mode %d (%s): %.3f ms, %.1f ns per MFMA per wave, %.1f TFLOP/s\n", blocks, mode,
             mode == 0 ? "16x16x32" : "16x16x16", ms, ms * 1e6 / (iters * 8.0), flop / ms / 1e9);
    }
  }
  return 0;
}
