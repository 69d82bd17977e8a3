// Does v_mfma_f32_16x16x32_f16 keep IEEE-half SUBNORMAL inputs, and does the f32 -> f16
// conversion produce them?  (The split-operand mode stores the low halves of activations
// and weights as halves; values below 2^-14 there must not be flushed.)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_denorm mfma_denorm.hip && ./mfma_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float a_val, float b_val, float *out, unsigned short *bits) {
  h8 a, b;
  const _Float16 ah = (_Float16)a_val, bh = (_Float16)b_val;   // v_cvt_f16_f32
  for (int j = 0; j < 8; ++j) { a[j] = ah; b[j] = bh; }
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; bits[0] = __builtin_bit_cast(unsigned short, ah); }
}
int main() {
  float *d; unsigned short *db;
  hipMalloc(&d, 4); hipMalloc(&db, 2);
  const float vals[] = {ldexpf(1.f, -15), ldexpf(1.f, -20), ldexpf(3.f, -24), ldexpf(1.f, -24)};
  for (float v : vals) {
    for (int side = 0; side < 2; ++side) {
      k<<<1, 64>>>(side ? 1.f : v, side ? v : 1.f, d, db);
      float h; unsigned short hb;
      hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); hipMemcpy(&hb, db, 2, hipMemcpyDeviceToHost);
      printf("subnormal %g on %c: sum of 32 products = %g (expected %g)%s\n", v, side ? 'B' : 'A', h,
             32.0 * v, h == 32.f * v ? "  kept" : "  FLUSHED / rounded");
    }
  }
  return 0;
}
