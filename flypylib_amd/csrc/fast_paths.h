// Fused whole-slab MFMA fast paths (pattern-matched on the layer program).
#pragma once
#include <vector>

#include "program.h"

// The two VGG-style graphs of flypylib/fplmodels.py:102-172 differ only in the kernel
// edge of the second convolution of each block: vgg_like 3,1 | 3,1 | 3,1,1,1 and
// vgg_like2 3,3 | 3,3 | 3,1,1,1 (48 channels, 96 in the two "dense" layers, biased
// sigmoid head, stride 4).  Returns 1 / 2 for those, 0 for anything else.
static inline int fpl_vgg_variant(const fpl_program *prog) {
  static const int kinds[10] = {0, 0, 1, 0, 0, 1, 0, 0, 0, 0};
  static const int cin[10] = {1, 48, 48, 48, 48, 48, 48, 48, 96, 96};
  static const int cout[10] = {48, 48, 48, 48, 48, 48, 48, 96, 96, 1};
  static const int ks_tail[4] = {3, 1, 1, 1};
  if (prog->ops.size() != 10) return 0;
  if (prog->stride[0] != 4 || prog->stride[1] != 4 || prog->stride[2] != 4) return 0;
  const int k2 = prog->ops[1].k;                      // 1: vgg_like, 3: vgg_like2
  if (k2 != 1 && k2 != 3) return 0;
  for (int i = 0; i < 10; ++i) {
    const fpl_op &op = prog->ops[i];
    if (op.kind != kinds[i]) return 0;
    if (op.src0 != (i == 0 ? 0 : prog->ops[i - 1].dst)) return 0;
    if (op.kind == FPL_OP_CONV) {
      const int k = i >= 6 ? ks_tail[i - 6] : ((i == 1 || i == 4) ? k2 : 3);
      if (op.k != k || op.cin != cin[i] || op.cout != cout[i]) return 0;
      if (op.act != (i == 9 ? FPL_ACT_SIGMOID : FPL_ACT_RELU)) return 0;
    } else if (op.p[0] != 2 || op.p[1] != 2 || op.p[2] != 2) {
      return 0;
    }
  }
  if (prog->out_tensor != prog->ops[9].dst) return 0;
  return k2 == 1 ? 1 : 2;
}
// one tile of the reference lattice (flypylib/fplnetwork.py:146-160)
struct FplTileDesc {
  int32_t start[3];   // input window origin in the volume
  int32_t ext[3];     // input window extent (<= tile_in; the rest of the tile is zero)
};

// optional volume-side output of a tile batch: when given, the last kernel writes the
// valid outputs of every tile straight into the (Z,Y,X) prediction volume - no
// per-tile output tensor, no stitch pass.  (Reading the tiles straight from the volume
// in the first kernel was measured too: its scattered byte loads and per-voxel
// normalisation cost more than the 2.6 ms gather pass they replace.)
struct FplTileIO {
  int64_t Y, X;               // volume pitches
  const FplTileDesc *tiles;   // device array, one per tile of the batch
  float *dst;                 // row `dst_z_base` of the prediction volume
  int64_t dst_z_base;
  int32_t off;                // rf_offset
};

// The fused 16-bit paths exist twice, built from the same sources (mfma_util.h):
// *_bf16 (FPL_PREC_BF16) and *_f16 (FPL_PREC_F16).
//
// fpl_fast_infer_volume_*: tries to run the slab [zb, ze) of the tile lattice with
// the fused vgg_like kernels.  Sets *handled = false (and returns 0) when the
// program / precision has no fast path.  `src` / `dst` point at row 0 of the (Z,Y,X)
// volume on the device.
// fpl_fast_path_available_*: true when fpl_fast_infer_volume_* will handle this
// program / precision / lattice.
// fpl_unet_*: unet_like2 MFMA executor (conv_mfma.hip): batch of n equal tiles.
#define FPL_DECLARE_H16_PATHS(sfx)                                                      \
  int fpl_fast_infer_volume_##sfx(fpl_ctx *ctx, fpl_program *prog, const void *src,     \
                                  int src_dtype, float mean, float sd,                  \
                                  const int64_t dims[3], const int32_t tile_in[3],      \
                                  const int32_t offset[3], int precision,               \
                                  const std::vector<int32_t> origins[3],                \
                                  const int32_t out_sz[3], int32_t zb, int32_t ze,      \
                                  float *dst, bool *handled);                           \
  bool fpl_fast_path_available_##sfx(const fpl_program *prog, int precision,            \
                                     const int32_t offset[3], const int32_t out_sz[3]); \
  bool fpl_unet_fast_available_##sfx(const fpl_program *prog, int precision);           \
  int fpl_unet_forward_##sfx(fpl_ctx *ctx, fpl_program *prog, const float *in, int n,   \
                             int T, float *out, const FplTileIO *io);
FPL_DECLARE_H16_PATHS(bf16)
FPL_DECLARE_H16_PATHS(f16)
// split IEEE halves (conv_mfma.hip built with -DFPL_SPLIT): only the fpl_unet_* pair exists
FPL_DECLARE_H16_PATHS(f16s)

// fp32 MFMA executor over any lowered program without ADD (conv_mfma_f32.hip):
// cubic tiles (n, T,T,T) f32 -> network output (n, d,d,d, c) f32
bool fpl_mfma_f32_supported(const fpl_program *prog);
int fpl_forward_mfma_f32(fpl_ctx *ctx, fpl_program *prog, const float *in, int n, int T,
                         float *out);

// fp32 MFMA convolutions for the training engine (conv_mfma_f32.hip)
bool fpl_tm_supported(int k, int cin, int cout);
bool fpl_tm_bwd_supported(int k, int cin, int cout);   // dgrad + wgrad of that conv
// `stats` (optional): per-channel (sum, sum of squares) partials of y for a following
// BatchNorm, fpl_tm_conv_stats_rows(...) rows x 2 x cout doubles (0 rows = not offered)
int64_t fpl_tm_conv_stats_rows(fpl_ctx *ctx, int n, int D, int H, int W_, int cin, int k, int cout);
// `bn` (optional, fpl_tm_bn_view_supported): x is the INPUT of a BatchNorm + ReLU whose
// output is this convolution's real operand; the kernel applies relu(bn(x)) (the fixed
// rounding sequence of train.hip's bn_affine) while loading, so that output tensor never
// exists in HBM.  Per-channel device vectors.
struct FplBnView { const float *mean, *invstd, *gamma, *beta; };
bool fpl_tm_bn_view_supported(int k, int cin, int cout);
int fpl_tm_conv_fwd(fpl_ctx *ctx, const float *x, int n, int D, int H, int W_, int cin, int k,
                    int cout, const float *Wd, const float *bias, int act, float *y,
                    double *stats = nullptr, const FplBnView *bn = nullptr);
// `bstat` (optional, fpl_tm_bn_view_supported(k, cin, cout)): dx is the gradient of a
// BatchNorm + ReLU output whose BN input is bstat->x; the epilogue also accumulates that
// BN's backward sums (sum g, sum g * xhat with g = dx where bn(x) > 0) into
// part[fpl_tm_conv_stats_rows(.., k = 1, ..)][2][cin] - the BN backward's first pass
struct FplBnStat { const float *x; FplBnView bn; double *part; };
// `pg` (optional, fpl_tm_pool_grad_supported(k, cin, cout); weight AND input gradient of the
// same 1x1x1 convolution): dy is not a tensor but the input gradient of a training-mode
// BatchNorm + ReLU + MaxPooling3D(2) layer, made while loading from the POOLED output gradient
// pg->dyp (n, D/2, H/2, W/2, C), the arg-max bytes pg->arg (one per pooled value, four
// channels per word: train.hip::bn_relu_pool4) and the BN input pg->x (n, D, H, W, C) =
// this convolution's output: train.hip::bn_backward_pool4's arithmetic, the same rounding
// sequence.  That pass - a write and a read of the layer's full-resolution tensor - is then not run.
struct FplPoolGrad {
  const float *dyp; const uint32_t *arg; const float *x;
  FplBnView bn; const float *sum_g, *sum_gx; float inv_m;
  int D, H, W;
};
bool fpl_tm_pool_grad_supported(int k, int cin, int cout);
int fpl_tm_conv_dgrad(fpl_ctx *ctx, const float *dy, int n, int od, int oh, int ow, int cout,
                      int k, int cin, const float *Wd, const float *zeros, float *dx,
                      const FplBnStat *bstat = nullptr, const FplPoolGrad *pg = nullptr);
// `bg` (optional, fpl_tm_bn_grad_supported(k, cin, cout)): dy is not a tensor but the input
// gradient of a training-mode BatchNorm (+ ReLU) whose output gradient is bg->g and whose
// input - this convolution's output - is bg->x: dy = gamma * invstd * (g' - sum_g / M - xhat *
// sum_gx / M), g' = g where bn(x) > 0 (train.hip::bn_backward4, the same rounding sequence),
// computed while loading.  That BN's elementwise backward pass - one write and one read
// of the layer's largest tensor - is then not run at all.
struct FplBnGrad { const float *g, *x; FplBnView bn; const float *sum_g, *sum_gx; float inv_m; };
bool fpl_tm_bn_grad_supported(int k, int cin, int cout);
int fpl_tm_conv_wgrad(fpl_ctx *ctx, const float *x, int n, int D, int H, int W_, int cin,
                      const float *dy, int k, int cout, float *dw, const FplBnView *bn = nullptr,
                      const FplBnGrad *bg = nullptr, const FplPoolGrad *pg = nullptr);

// Training: 3x3x3 convolutions of 32 - 192 channels (multiples of 16) - forward: dgrad = 0; input gradient:
// dgrad = 1, x = dy - on split halves (conv_mfma.hip, split build): fp32-grade results at five times the fp32
// matrix rate
bool fpl_tm_conv3_split_supported(int k, int cin, int cout);
int fpl_tm_conv3_split(fpl_ctx *ctx, const float *x, int n, int D, int H, int W_, int cin, int cout, const float *Wd,
                       const float *bias, int dgrad, int relu, float *y);
bool fpl_tm_conv3_wgrad_split_supported(int k, int cin, int cout);
// ... and the weight gradient dw [27][cin][cout] += x (n, D^3, cin) * dy (n, (D - 2)^3, cout) on the planar copies the
// two calls above left in the context (made here when missing); fpl_tm_split_reset drops the copies
// (start and end of a training step: the fp32 tensors they mirror are recycled between steps)
int fpl_tm_conv3_wgrad_split(fpl_ctx *ctx, const float *x, int n, int D, int cin, int cout, const float *dy, float *dw);
void fpl_tm_split_reset(fpl_ctx *ctx);

// Split-operand IEEE-half path for vgg_like (vgg_split.hip, FPL_PREC_F16S): every
// activation and folded weight is carried as hi + lo (two halves, ~22 significant bits)
// and every product as three MFMAs - fp32-grade probabilities at a third of the 16-bit
// MFMA rate.  Same contract as fpl_fast_infer_volume_*.
bool fpl_split_path_available(const fpl_program *prog, int precision,
                              const int32_t offset[3], const int32_t out_sz[3]);
int fpl_split_infer_volume(fpl_ctx *ctx, fpl_program *prog, const void *src, int src_dtype,
                           float mean, float sd, const int64_t dims[3],
                           const std::vector<int32_t> origins[3], const int32_t out_sz[3],
                           int32_t zb, int32_t ze, float *dst);
