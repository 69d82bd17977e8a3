// Micro-benchmark: issue rate of fp64 / fp32 VALU instructions on gfx950 (what bounds the
// scipy-exact Gaussian passes of voxel2obj: 31 separate fp64 add/mul per output and
// axis).  hipcc --offload-arch=gfx950 -O3 fp64_rate.hip -o fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: v_add_f64, 1: v_mul_f64, 2: v_fma_f64, 3: v_add_f32, 4: v_cvt_f64_f32
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double seed) {
  double a[8];
  float f[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3 + i; f[i] = (float)a[i]; }
  const double c = seed * 1.0000001;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = __dadd_rn(a[i], c);
      if (MODE == 1) a[i] = __dmul_rn(a[i], c);
      if (MODE == 2) a[i] = __fma_rn(a[i], c, c);
      if (MODE == 3) f[i] = __fadd_rn(f[i], (float)c);
      if (MODE == 4) { a[i] = (double)f[i]; asm volatile("" : "+v"(a[i])); f[i] = __builtin_bit_cast(float, __builtin_bit_cast(int, f[i]) ^ (int)(a[i] > 1e300)); }
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  double *d; hipMalloc(&d, 8192 * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const char *names[5] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_add_f32", "v_cvt_f64_f32(+2 int ops)"};
  for (int blocks = 256; blocks <= 2048; blocks *= 2)     // 1, 2, 4, 8 waves per SIMD
    for (int mode = 0; mode < 5; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        switch (mode) {
          case 0: k<0><<<blocks, 256>>>(d, iters, 1.5); break;
          case 1: k<1><<<blocks, 256>>>(d, iters, 1.5); break;
          case 2: k<2><<<blocks, 256>>>(d, iters, 1.5); break;
          case 3: k<3><<<blocks, 256>>>(d, iters, 1.5); break;
          default: k<4><<<blocks, 256>>>(d, iters, 1.5); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0) continue;
        const double wave_instr_per_simd = (double)blocks / 256.0 * iters * 8;   // per SIMD
        printf("%d waves/SIMD %-28s %.3f ms  %.2f ns per wave-instruction per SIMD (x2.4 GHz = %.1f cycles)\n",
               blocks / 256, names[mode], ms, ms * 1e6 / wave_instr_per_simd,
               ms * 1e6 / wave_instr_per_simd * 2.4);
      }
    }
  return 0;
}
