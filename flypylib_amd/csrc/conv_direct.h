// Direct fp32 conv kernel shared by the per-op inference executor (generic.hip)
// and the training engine (train.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fplhip.h"

static __device__ __forceinline__ float apply_act(float v, int act) {
  if (act == FPL_ACT_RELU) return fmaxf(v, 0.f);
  if (act == FPL_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}

// Direct valid 3-D cross-correlation.  One thread = one output voxel x CT
// consecutive output channels; weight addresses are wave-uniform (scalar loads).
template <int CT>
static __global__ __launch_bounds__(256) void conv3d_direct_f32(
    const float *__restrict__ x, const float *__restrict__ w,
    const float *__restrict__ scale, const float *__restrict__ shift,
    float *__restrict__ y, int64_t n_vox, int D, int H, int W, int cin, int od,
    int oh, int ow, int cout, int k, int act) {
  int64_t vox = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vox >= n_vox) return;
  const int co0 = blockIdx.y * CT;
  int64_t t = vox;
  const int ox = (int)(t % ow); t /= ow;
  const int oy = (int)(t % oh); t /= oh;
  const int oz = (int)(t % od); t /= od;
  const int64_t b = t;
  float acc[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) acc[j] = 0.f;
  for (int dz = 0; dz < k; ++dz)
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        const float *xp =
            x + ((((b * D + oz + dz) * H + oy + dy) * (int64_t)W + ox + dx) * cin);
        const float *wp = w + (int64_t)((dz * k + dy) * k + dx) * cin * cout + co0;
        for (int ci = 0; ci < cin; ++ci) {
          const float xv = xp[ci];
#pragma unroll
          for (int j = 0; j < CT; ++j)
            if (CT == 1 || co0 + j < cout)
              acc[j] = fmaf(xv, wp[(int64_t)ci * cout + j], acc[j]);
        }
      }
  float *yp = y + vox * cout + co0;
#pragma unroll
  for (int j = 0; j < CT; ++j)
    if (co0 + j < cout)
      yp[j] = apply_act(fmaf(acc[j], scale[co0 + j], shift[co0 + j]), act);
}

