// Micro-benchmark: sustained rate of the bf16 MFMA shapes of gfx950, every SIMD busy,
// operands in registers, independent accumulators, RANDOM operand data (the clock the chip
// holds under MFMA load depends on the data: MI355X_MICROARCH.md, DVFS notes) or constant
// data (argument "const").  The loops are checked in the assembly to hold MFMAs only
// (an earlier version of this file let the compiler shuttle the accumulators through
// AGPRs every iteration - 80 moves per 8 MFMAs - and under-reported the 16x16x32 rate).
//   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate [const]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k16(const bf16x8 *__restrict__ ops, float *out, int iters) {
  const bf16x8 a0 = ops[threadIdx.x], b0 = ops[256 + threadIdx.x];
  const bf16x8 a1 = ops[512 + threadIdx.x], b1 = ops[768 + threadIdx.x];
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  // inline assembly: the accumulators stay in place in VGPRs (through the builtin the
  // compiler moved them through AGPRs at the loop boundary)
#define MF16(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
  for (int it = 0; it < iters; ++it) {
    MF16(c0, a0, b0); MF16(c1, a1, b1); MF16(c2, a0, b1); MF16(c3, a1, b0);
    MF16(c4, a0, b0); MF16(c5, a1, b1); MF16(c6, a0, b1); MF16(c7, a1, b0);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void k32(const bf16x8 *__restrict__ ops, float *out, int iters) {
  const bf16x8 a0 = ops[threadIdx.x], b0 = ops[256 + threadIdx.x];
  const bf16x8 a1 = ops[512 + threadIdx.x], b1 = ops[768 + threadIdx.x];
  f32x16 c0, c1, c2, c3;
  for (int j = 0; j < 16; ++j) c0[j] = 0.f;
  c1 = c0; c2 = c0; c3 = c0;
#define MF32(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
  for (int it = 0; it < iters; ++it) {
    MF32(c0, a0, b0); MF32(c1, a1, b1); MF32(c2, a0, b1); MF32(c3, a1, b0);
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const f32x16 s = c0 + c1 + c2 + c3;
  float t = 0;
  for (int j = 0; j < 16; ++j) t += s[j];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

int main(int argc, char **argv) {
  const bool constant = argc > 1 && !strcmp(argv[1], "const");
  std::vector<unsigned short> h(1024 * 8);
  unsigned x = 12345u;
  for (auto &v : h) {
    x = x * 1664525u + 1013904223u;
    // bf16 in [-1, 1): random sign, exponent 119..126, random mantissa
    v = constant ? 0x3f80 : (unsigned short)(((x >> 31) << 15) | ((119 + ((x >> 8) & 7)) << 7) | ((x >> 16) & 127));
  }
  bf16x8 *ops;
  float *d;
  hipMalloc(&ops, h.size() * 2);
  hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&d, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 40000;
  if (argc > 2) {                       // soak: one shape, 2048 workgroups, for power sampling
    const int shape = atoi(argv[2]) == 32, n = argc > 3 ? atoi(argv[3]) : 250;
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) {
      if (shape == 0) k16<<<2048, 256>>>(ops, d, iters);
      else k32<<<2048, 256>>>(ops, d, iters / 2);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (shape ? 2048.0 * 4 * (iters / 2) * 4 * 32768.0 : 2048.0 * 4 * iters * 8 * 16384.0) * n;
    printf("soak %s x %d: %.1f ms, %.1f TFLOP/s\n", shape ? "32x32x16" : "16x16x32", n, ms,
           flop / ms / 1e9);
    return 0;
  }
  printf("operands: %s\n", constant ? "constant 1.0" : "random bf16 in [-1, 1)");
  for (int rep = 0; rep < 2; ++rep)
    for (int blocks = 512; blocks <= 2048; blocks *= 2) {
      for (int shape = 0; shape < 2; ++shape) {
        hipEventRecord(e0);
        if (shape == 0) k16<<<blocks, 256>>>(ops, d, iters);
        else k32<<<blocks, 256>>>(ops, d, iters / 2);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = shape == 0 ? (double)blocks * 4 * iters * 8 * 16384.0
                                       : (double)blocks * 4 * (iters / 2) * 4 * 32768.0;
        printf("  %4d workgroups (%d waves per SIMD) %s: %8.3f ms  %7.1f TFLOP/s\n", blocks,
               blocks / 256, shape == 0 ? "16x16x32" : "32x32x16", ms, flop / ms / 1e9);
      }
    }
  return 0;
}
