// Layer program held on the device + shape inference shared by the executors.
#pragma once
#include "common.h"

struct TensorShape {
  int32_t d = 0, h = 0, w = 0, c = 0;
  int64_t voxels() const { return (int64_t)d * h * w; }
  int64_t elems() const { return voxels() * c; }
};

struct fpl_program {
  fpl_ctx *ctx = nullptr;
  std::vector<fpl_op> ops;
  int32_t n_tensors = 0;
  int32_t out_tensor = 0;
  int32_t stride[3] = {1, 1, 1};
  std::vector<float> arena_host;
  float *arena_dev = nullptr;        // fp32 weights, scale, shift
  int64_t n_arena = 0;
  // packed 16-bit weight fragments etc. of the fused fast paths (vgg_fused.hip,
  // conv_mfma.hip, vgg_split.hip), one slot per operand type: [0] bf16, [1] f16,
  // [2] split f16
  void *fast_state_h16[3] = {nullptr, nullptr, nullptr};
  void (*fast_state_h16_free[3])(fpl_ctx *, void *) = {nullptr, nullptr, nullptr};
  uint64_t arena_version = 0;
  // arena version at which FPL_PREC_AUTO found weights or values beyond the IEEE-half range:
  // that network runs on the fp32 executor until its weights change (infer.hip)
  uint64_t half_range_bad_version = ~0ull;
  // fp32 MFMA executor state (conv_mfma_f32.hip)
  void *fast_state_f32 = nullptr;
  void (*fast_state_f32_free)(fpl_ctx *, void *) = nullptr;
};

// shapes of every tensor for a given input size; returns non-zero + message on
// inconsistent programs
int fpl_infer_shapes(fpl_ctx *ctx, const fpl_program *prog,
                     const int32_t in_dims[3], std::vector<TensorShape> *shapes);

// generic fp32 executor: in (n, D,H,W,1) device f32 -> out (n, d,h,w, c_out)
// device f32 (caller-allocated, shapes from fpl_infer_shapes)
int fpl_forward_generic(fpl_ctx *ctx, fpl_program *prog, const float *in_dev,
                        int32_t n, const int32_t in_dims[3], float *out_dev);
