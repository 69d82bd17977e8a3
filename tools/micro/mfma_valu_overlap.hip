// How do v_mfma_f32_16x16x32_f16 and ordinary VALU instructions share a SIMD of gfx950?
// Instruction order pinned with asm volatile.  Per loop iteration NM MFMAs (6 independent
// accumulators) and NV VALU instructions (8 independent registers), either as two phases
// (all MFMAs, then all VALU: what a ReLU / split epilogue between two layers looks like) or
// interleaved; one or two waves per SIMD; the second wave of a SIMD optionally at a higher
// s_setprio.  Prints cycles per iteration and SIMD (at the clock measured by s_memtime...
// here simply wall time x 2.0 GHz; compare rows, not absolutes).
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int OP>
__device__ __forceinline__ void valu(float &x, float y) {
  if (OP == 0) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  if (OP == 1) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  if (OP == 2) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(y));
  if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
}

// MODE 0: NM MFMAs then NV VALU.  MODE 1: after MFMA m, its share of the VALU instructions.
template <int NM, int NV, int MODE, int OP>
__global__ __launch_bounds__(512) void k(float *out, int iters, float seed, int prio) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i * seed); b[i] = (_Float16)(1.0f + i * 0.01f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i;
  f4 acc[6];
  for (int i = 0; i < 6; ++i) acc[i] = f4{0, 0, 0, 0};
  if (prio && threadIdx.x >= 256) __builtin_amdgcn_s_setprio(2);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int m = 0; m < NM; ++m) MFMA(acc[m % 6]);
#pragma unroll
      for (int j = 0; j < NV; ++j) valu<OP>(v[j % 8], seed);
    } else {
      constexpr int N = NM ? NM : 1;
#pragma unroll
      for (int m = 0; m < N; ++m) {
        if (NM) MFMA(acc[m % 6]);
#pragma unroll
        for (int j = 0; j < (NV * (m + 1)) / N - (NV * m) / N; ++j) valu<OP>(v[(m + j) % 8], seed);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int MODE, int OP>
void run(float *d, int waves, int prio, const char *what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 256;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    k<NM, NV, MODE, OP><<<blocks, 64 * waves>>>(d, iters, 1.25f, prio);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double per_wave = ms * 1e6 / iters * 2.0;        // cycles (2.0 GHz) per iteration of one wave
  printf("%-34s NM=%2d NV=%2d op=%d waves/SIMD=%d prio=%d: %7.1f cycles per iteration, %6.1f per SIMD-iteration\n",
         what, NM, NV, OP, waves / 4, prio, per_wave, per_wave / (waves / 4));
}

int main() {
  float *d; hipMalloc(&d, 256 * 512 * 4);
  run<21, 0, 0, 0>(d, 4, 0, "MFMA only");
  run<21, 0, 0, 0>(d, 8, 0, "MFMA only");
  run<0, 54, 0, 0>(d, 4, 0, "VALU only (v_max_f32)");
  run<0, 54, 0, 0>(d, 8, 0, "VALU only (v_max_f32)");
  run<0, 54, 0, 1>(d, 8, 0, "VALU only (v_cvt_pk_f16_f32)");
  run<0, 54, 0, 2>(d, 8, 0, "VALU only (v_fma_mixlo_f16)");
  run<0, 54, 0, 3>(d, 8, 0, "VALU only (v_max3_f32)");
  run<21, 54, 0, 0>(d, 4, 0, "phases");
  run<21, 54, 0, 0>(d, 8, 0, "phases");
  run<21, 54, 0, 0>(d, 8, 1, "phases");
  run<21, 54, 1, 0>(d, 4, 0, "interleaved");
  run<21, 54, 1, 0>(d, 8, 0, "interleaved");
  run<21, 54, 1, 1>(d, 8, 0, "interleaved cvt_pk");
  run<21, 54, 1, 2>(d, 8, 0, "interleaved fma_mix");
  run<21, 54, 1, 3>(d, 8, 0, "interleaved max3");
  run<21, 21, 1, 0>(d, 8, 0, "interleaved 1 per MFMA");
  run<21, 42, 1, 0>(d, 8, 0, "interleaved 2 per MFMA");
  run<21, 63, 1, 0>(d, 8, 0, "interleaved 3 per MFMA");
  run<21, 84, 1, 0>(d, 8, 0, "interleaved 4 per MFMA");
  run<21, 42, 1, 0>(d, 4, 0, "interleaved 2 per MFMA");
  run<21, 63, 1, 0>(d, 4, 0, "interleaved 3 per MFMA");
  run<21, 42, 0, 0>(d, 8, 0, "phases 2 per MFMA");
  run<21, 42, 0, 0>(d, 8, 1, "phases 2 per MFMA");
  return 0;
}
