/*
 * fplhip.h - C ABI of libfplhip.so: the MI355X (gfx950) engine behind
 * flypylib's T-bar detection hot path.
 *
 * The reference (janelia-flyem/flypylib) is pure Python on Keras/TensorFlow and
 * has no FFI of its own; its boundary for this path is the Python API.  Each
 * entry point below names the reference interface it stands in for
 * (paths relative to the reference repository).  Host bindings: ctypes, see
 * flypylib_amd/_capi.py and INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from fpl_last_error(ctx) (ctx may be NULL for create failures).
 *   - no exceptions / abort across the boundary.
 *   - host buffers are caller-owned, C-contiguous.  Pointers tagged
 *     `mem = FPL_MEM_DEVICE` are device pointers of the same HIP device/primary
 *     context (e.g. a torch tensor's data_ptr()).
 *   - library-owned device memory is freed by the matching *_destroy.
 *   - one context per GPU per process; a context is used by one host thread at
 *     a time; create contexts after fork() (HIP state does not survive fork).
 *   - volumes are (Z, Y, X) with X fastest, as numpy C-order arrays.
 */
#ifndef FPLHIP_H
#define FPLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: these declarations ARE its export list
 * (tests/test_host_logic.py holds `nm -D` of the built library to them) */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define FPL_ABI_VERSION 8

typedef struct fpl_ctx fpl_ctx;
typedef struct fpl_program fpl_program;

enum fpl_mem { FPL_MEM_HOST = 0, FPL_MEM_DEVICE = 1 };
enum fpl_dtype { FPL_U8 = 0, FPL_F32 = 1, FPL_F64 = 2 };
/* arithmetic of the convolutions: fp32 (exact reference arithmetic), or 16-bit MFMA
 * operands with fp32 accumulation - bfloat16 (8 significant bits), IEEE half (11
 * bits, range 65504: probabilities within ~1e-3 of fp32 on trained weights; same speed
 * as bf16), or SPLIT IEEE halves (FPL_PREC_F16S: every operand as hi + lo, ~22 bits,
 * three MFMAs per product: probabilities within ~4e-6 of fp32 - inside the reference's
 * 1e-3 gate, the same detected point set as the fp32 path's, order included up to
 * confidences that tie to 1e-6 - at a third of the 16-bit rate).  Split kernels exist for
 * vgg_like, vgg_like2 (stride-4 lattices) and unet_like / unet_like2 / unet_like3 /
 * unet_like4 (cubic tiles); other graphs are refused at FPL_PREC_F16S.  Its operands are
 * IEEE halves: a folded weight, a normalised input voxel or an activation beyond 65504
 * makes the call FAIL (the kernels check every value they split) - never a wrong result. */
enum fpl_precision {
  /* what FplNetwork.infer uses unless told otherwise: fp32-GRADE results on the fastest
   * executor that delivers them - FPL_PREC_F16S where split kernels exist AND every value
   * stays inside the half range (a call that leaves it is rerun on the fp32 executor:
   * the reference predicts in fp32, flypylib/fplnetwork.py:175-176, which has no such
   * limit), else FPL_PREC_F32 */
  FPL_PREC_AUTO = -1,
  FPL_PREC_F32 = 0, FPL_PREC_BF16 = 1, FPL_PREC_F16 = 2, FPL_PREC_F16S = 3
};

/* fused layer-program ops, produced by LayerGraph.lower_inference()
 * (flypylib_amd/program.py); layer semantics = Keras layers instantiated in
 * flypylib/fplmodels.py:67-526 */
enum fpl_op_kind {
  FPL_OP_CONV = 0,   /* y = act(scale * conv3d_valid(x, W) + shift)          */
  FPL_OP_POOL = 1,   /* MaxPooling3D(p0)                                     */
  FPL_OP_UP = 2,     /* UpSampling3D((p0,p1,p2)) nearest                     */
  FPL_OP_CROP = 3,   /* Cropping3D(((p0,p1),(p2,p3),(p4,p5)))                */
  FPL_OP_CONCAT = 4, /* concatenate([src0, src1]) on channels                */
  FPL_OP_ADD = 5     /* act(src0 + src1)                                     */
};
enum fpl_act { FPL_ACT_NONE = 0, FPL_ACT_RELU = 1, FPL_ACT_SIGMOID = 2 };

typedef struct fpl_op {
  int32_t kind;      /* fpl_op_kind                                          */
  int32_t src0, src1;/* tensor ids (0 = network input); src1 = -1 if unused  */
  int32_t dst;       /* tensor id produced                                   */
  int32_t k;         /* conv kernel edge (1 or 3)                            */
  int32_t cin, cout; /* channels in / out                                    */
  int32_t act;       /* fpl_act                                              */
  int64_t w_off;     /* offsets (floats) into the weight arena:              */
  int64_t scale_off; /*   kernel [k^3*cin][cout] (Keras memory order),       */
  int64_t shift_off; /*   scale[cout], shift[cout]                           */
  int32_t p[6];      /* pool/up factors or crop pairs                        */
} fpl_op;

/* ---- context -------------------------------------------------------------- */
int fpl_abi_version(void);
/* replaces: the implicit TF session/device placement of multi_gpu.make_parallel
 * (flypylib/multi_gpu.py:33-35); one context binds one GPU */
int fpl_ctx_create(int device_id, fpl_ctx **out);
int fpl_ctx_destroy(fpl_ctx *ctx);
const char *fpl_last_error(fpl_ctx *ctx);
/* run all subsequent work of this context on `hip_stream` (hipStream_t; NULL =
 * the context's own stream) */
int fpl_ctx_set_stream(fpl_ctx *ctx, void *hip_stream);
int fpl_ctx_synchronize(fpl_ctx *ctx);
int fpl_device_info(fpl_ctx *ctx, int32_t *n_cu, int64_t *hbm_bytes,
                    char *name, size_t name_cap);
/* PCI bus id ("0000:05:00.0") of the context's GPU: tells ranks that share a GPU
 * from ranks that merely share a device index (per-process HIP_VISIBLE_DEVICES) */
int fpl_device_pci_bus_id(fpl_ctx *ctx, char *out, size_t cap);

/* ---- device buffers (thin; for callers without torch) ---------------------- */
int fpl_malloc(fpl_ctx *ctx, size_t bytes, void **dev_ptr);
int fpl_free(fpl_ctx *ctx, void *dev_ptr);
int fpl_memcpy(fpl_ctx *ctx, void *dst, int dst_mem, const void *src,
               int src_mem, size_t bytes);

/* ---- layer program --------------------------------------------------------- */
/* replaces: FplNetwork._set_infer (flypylib/fplnetwork.py:99-110): build the
 * inference network and load the weights.  `stride[3]` is rf_stride: the final
 * UpSampling3D(rf_stride) is applied on store when != 1. */
int fpl_program_create(fpl_ctx *ctx, const fpl_op *ops, int32_t n_ops,
                       int32_t n_tensors, int32_t out_tensor,
                       const float *arena, int64_t n_arena,
                       const int32_t stride[3], fpl_program **out);
int fpl_program_destroy(fpl_program *prog);
/* replaces: Model.set_weights (fplnetwork.py:109-110) for a lowered arena */
int fpl_program_set_arena(fpl_program *prog, const float *arena,
                          int64_t n_arena);

/* replaces: infer_network.predict(data_batch) (fplnetwork.py:175-176) for a
 * batch of equally-sized tiles: in (n, D,H,W) f32 -> out (n, d,h,w,c) f32, where
 * (d,h,w) is the network output size for input (D,H,W) (times stride) and c its
 * channel count (1 for every reference model).  out_dims receives (d,h,w,c);
 * pass out = NULL to query the shape only.  Generic per-op kernels, fp32: the
 * parity path for every architecture. */
int fpl_program_forward(fpl_ctx *ctx, fpl_program *prog, const float *in,
                        int in_mem, int32_t n, const int32_t in_dims[3],
                        int precision, float *out, int out_mem,
                        int32_t out_dims[4]);

/* replaces: FplNetwork.infer (fplnetwork.py:136-189): tile lattice, zero-padded
 * edge tiles, predict, stitch; the rf_offset border shell of `dst` is zero.
 *   src        (Z,Y,X) volume, dtype u8 or f32; normalised on load as
 *              (v - mean) / std   (pass mean=0,std=1 for pre-normalised input)
 *   tile_in    input tile edge per axis (reference: infer_sz, e.g. 102 / 100);
 *              tiles advance by tile_in - 2*offset exactly as fplnetwork.py:149-159
 *   z_begin/z_end  restrict the work to tile rows [z_begin, z_end) of the tile
 *              lattice along Z (slab sharding across GPUs; 0, -1 = all)
 *   dst        (Z,Y,X) f32, same shape as src */
int fpl_infer_volume(fpl_ctx *ctx, fpl_program *prog, const void *src,
                     int src_dtype, int src_mem, float mean, float std,
                     const int64_t dims[3], const int32_t tile_in[3],
                     const int32_t offset[3], int precision,
                     int32_t z_begin, int32_t z_end, float *dst, int dst_mem);

/* name of the executor the last fpl_infer_volume / fpl_program_forward of this
 * context ran on: "vgg_fused_f16" | "vgg_fused_bf16" | "vgg_split_f16" | "unet_split_f16" | "unet_mfma_f16" |
 * "unet_mfma_bf16" | "graph_split_f16" | "graph_mfma_f16" | "graph_mfma_bf16" (the layer-by-layer
 * executor of the other factories) | "mfma_f32" | "perop_f32" | "none" (empty slab); with the suffix
 * "(range)" - "mfma_f32(range)" - when FPL_PREC_AUTO fell back to fp32 because a weight,
 * an input voxel or an activation left the IEEE-half range of the split kernels.  The fused
 * 16-bit kernels are keyed on the architectures of flypylib/fplmodels.py, the graph executor on
 * the layer kinds and widths it has kernels for; any other graph runs on the fp32 executors -
 * this says which, instead of leaving the caller to infer it from the speed. */
const char *fpl_last_path(fpl_ctx *ctx);

/* ---- post-process ----------------------------------------------------------- */
/* replaces: fplobjdetect.voxel2obj device stages (flypylib/fplobjdetect.py:
 * 158-231): pad by r, Gaussian smooth (weights = scipy's float64 kernel, radius
 * `wr`, passed by the host), zero the r-shell, exact order statistics for the
 * percentile, threshold, radius NMS.  Host epilogue (:233-257) stays in Python.
 *
 * Step 1: smooth; returns the k-th smallest values (0-based ranks, ascending) of
 * the padded smoothed volume for `n_ranks` requested ranks. */
int fpl_v2o_smooth(fpl_ctx *ctx, const float *pred, int pred_mem,
                   const int64_t dims[3], int32_t r, const double *weights,
                   int32_t wr, const int64_t *ranks, int32_t n_ranks,
                   float *rank_values);
/* Optional hint before step 1: the threshold of the following NMS will be >= `floor`
 * (voxel2obj's is max(percentile, thd) >= thd, fplobjdetect.py:183-185).  The smoothing
 * pass then keeps per-cell candidate keys only above it and the NMS starts from them
 * instead of scanning the volume; a smaller threshold than promised is still handled
 * (by the scan).  Order statistics: when every requested rank lies in a first-level
 * radix bin below the floor's (all of them < floor), `rank_values` reports the floor
 * itself for each - max(statistic, floor) is unchanged and the filtered scan + two
 * radix levels are skipped; otherwise the values are exact.  Holds for one
 * fpl_v2o_smooth. */
int fpl_v2o_set_floor(fpl_ctx *ctx, float floor);
/* Step 2: NMS over voxels with (double)value > thresh (and > 0) of the volume prepared
 * by step 1.  out_zyxv: rows (z, y, x in padded coordinates, value) as f64,
 * sorted by (value desc, flat index asc) = the reference's emission order.
 * Returns the number of detections in *n_out (<= cap, else error).  The NMS
 * suppresses in place: it CONSUMES the smoothed volume of step 1 (a second NMS needs a
 * new fpl_v2o_smooth). */
int fpl_v2o_nms(fpl_ctx *ctx, double thresh, double *out_zyxv, int64_t cap,
                int64_t *n_out, int32_t *n_rounds);
/* Segmentation-aware suppression (the seg / seg_dilate / seg_sz_thd / seg_force
 * arguments of voxel2obj, fplobjdetect.py:161-165,177-181,190-224).  Call order:
 * fpl_v2o_smooth (n_ranks = 0) -> fpl_v2o_set_seg -> fpl_v2o_select -> fpl_v2o_nms_seg.
 * set_seg: `seg` = labels (Z,Y,X) of the prediction's dims, 4 or 8 bytes each, zero
 * padded like the prediction; with sz_thd >= 0 the smoothed voxels of segments of
 * fewer than sz_thd voxels are zeroed.  select: exact order statistics of the
 * (possibly edited) smoothed volume.  nms_seg: a pick suppresses only the part of its
 * ball inside its own segment (the segment's mask in the (2r+1)^3 cube grown by
 * seg_dilate iterations of the 6-neighbour dilation) plus the ball of radius
 * seg_force (0 = none); obj_min_dist <= 127 (rows of the (2r+1)^3 cube are bit masks of up
 * to four 64-bit words; <= 31 - one word, every caller of the reference uses 27 - runs from LDS). */
int fpl_v2o_set_seg(fpl_ctx *ctx, const void *seg, int32_t seg_bytes, int seg_mem,
                    const int64_t dims[3], int64_t sz_thd);
int fpl_v2o_select(fpl_ctx *ctx, const int64_t *ranks, int32_t n_ranks, float *rank_values);
int fpl_v2o_nms_seg(fpl_ctx *ctx, double thresh, int32_t seg_dilate, int32_t seg_force,
                    double *out_zyxv, int64_t cap, int64_t *n_out, int32_t *n_rounds);
/* float64 predictions.  The reference pads, smooths (scipy: float64 output for a float64
 * input, no rounding between the axes), takes the percentile and compares in the array's
 * own dtype (fplobjdetect.py:158-231); its results then differ from the float32
 * pipeline's.  Call order: fpl_v2o_smooth_f64 [-> fpl_v2o_set_seg] -> fpl_v2o_select_f64
 * (sorts all padded voxels; exact float64 order statistics) -> host: thresh =
 * max(percentile, thd) -> fpl_v2o_rank_f64 (every voxel > max(thresh, 0) gets its dense
 * rank among those values as a float32 surrogate - exact below 2^24 distinct values, else
 * an error) -> fpl_v2o_nms / fpl_v2o_nms_seg with thresh 0.5 (unchanged: the NMS only
 * needs the order) -> fpl_v2o_values_f64 (the float64 values at the picked voxels' padded
 * flat indices). */
int fpl_v2o_smooth_f64(fpl_ctx *ctx, const double *pred, int pred_mem, const int64_t dims[3],
                       int32_t r, const double *weights, int32_t wr);
/* INTEGER predictions (any integer dtype, handed over as exact doubles): scipy filters an
 * integer array with float64 accumulation and stores every axis pass in the array's own type -
 * a C cast, truncation toward zero (fplobjdetect.py:167-168 on an integer `pred`); numpy's
 * percentile of it is a float64.  fpl_v2o_set_integer(ctx, 1) makes the NEXT
 * fpl_v2o_smooth_f64 truncate after every pass; the rest of the float64 call order is unchanged. */
int fpl_v2o_set_integer(fpl_ctx *ctx, int32_t on);
int fpl_v2o_select_f64(fpl_ctx *ctx, const int64_t *ranks, int32_t n_ranks, double *rank_values);
int fpl_v2o_rank_f64(fpl_ctx *ctx, double thresh, int64_t *n_candidates);
int fpl_v2o_values_f64(fpl_ctx *ctx, const int64_t *flat, int64_t n, double *out);
/* smoothed padded volume of the last fpl_v2o_smooth call (tests) */
int fpl_v2o_copy_smoothed(fpl_ctx *ctx, float *dst, int dst_mem);

/* ---- training -------------------------------------------------------------------- */
/* replaces: train_network.fit_generator's per-step work (flypylib/fplnetwork.py:
 * 112-122, default compile args :74-77): forward in training mode (BatchNorm
 * batch statistics, momentum 0.99, eps 1e-3; inverted Dropout), binary
 * cross-entropy on the sigmoid output (Keras clips p to [1e-7, 1-1e-7]), backward,
 * Adam (Keras defaults).  The layer list is the UNFUSED graph of
 * flypylib_amd/program.py (one entry per Keras layer). */
enum fpl_layer_kind {
  FPL_L_CONV = 0, FPL_L_BN = 1, FPL_L_RELU = 2, FPL_L_POOL = 3, FPL_L_DROPOUT = 4,
  FPL_L_UP = 5, FPL_L_CROP = 6, FPL_L_CONCAT = 7, FPL_L_ADD = 8
};
typedef struct fpl_layer {
  int32_t kind;        /* fpl_layer_kind                                       */
  int32_t src0, src1;  /* tensor ids (0 = network input)                       */
  int32_t dst;
  int32_t k, cin, cout;/* conv                                                 */
  int32_t use_bias;    /* conv                                                 */
  int32_t act;         /* conv activation (fpl_act); sigmoid only on the head  */
  float rate;          /* dropout                                              */
  int32_t p[6];        /* pool/up factors or crop pairs                        */
  int64_t w_off[4];    /* offsets (floats) into the weight arena: conv kernel,
                        * bias | BN gamma, beta, moving_mean, moving_variance   */
} fpl_layer;
typedef struct fpl_trainer fpl_trainer;

/* losses of the reference: Keras 'binary_crossentropy' (fplnetwork.py:74-77) and
 * the custom ones of fplmodels.py:28-50.  Label 2 = "don't care" for the masked
 * kinds.  Every loss is the mean over all output voxels of the batch (Keras
 * means the per-voxel values whatever their mask). */
typedef enum fpl_loss {
  FPL_LOSS_BCE = 0,                 /* clip p to [1e-7, 1-1e-7], -y log p - (1-y) log(1-p) */
  FPL_LOSS_MASKED_BCE = 1,          /* BCE of (p*mask, y*mask), fplmodels.py:41-44 */
  FPL_LOSS_MASKED_WEIGHTED_BCE = 2, /* pos_weight 100 on logits, fplmodels.py:28-39 */
  FPL_LOSS_MASKED_FOCAL = 3         /* gamma 2, alpha 1, fplmodels.py:45-50 */
} fpl_loss;
#define FPL_N_METRIC_SUMS 8

/* `weights`: flat fp32 concatenation of the Keras get_weights() list.  The
 * trainer owns a gradient arena of the same layout (moving-statistics slots hold
 * the pending moving-average delta); fpl_trainer_grad_ptr exposes it so the host
 * can all-reduce it over RCCL between step and apply
 * (replaces the implicit gradient sum over towers of multi_gpu.py:20-61). */
int fpl_trainer_create(fpl_ctx *ctx, const fpl_layer *layers, int32_t n_layers,
                       int32_t n_tensors, int32_t out_tensor, const float *weights,
                       int64_t n_weights, float lr, float beta1, float beta2,
                       float eps, fpl_trainer **out);
int fpl_trainer_destroy(fpl_trainer *t);
/* forward + loss + backward on one batch: data (batch, D,H,W) f32, labels
 * (batch, d,h,w) u8 in {0,1} with (d,h,w) = the network output size.  Fills the
 * gradient arena (mean over the batch).  `seed` drives the dropout masks. */
int fpl_trainer_step(fpl_trainer *t, const float *data, int data_mem,
                     const uint8_t *labels, int labels_mem, int32_t batch,
                     const int32_t patch[3], uint64_t seed, float *loss,
                     float *accuracy);
/* loss used by the following steps (default FPL_LOSS_BCE) */
int fpl_trainer_set_loss(fpl_trainer *t, int loss_kind);
/* raw sums of the last step, from which the host forms the reference's metrics
 * (fplmodels.py:52-65): [0] sum of per-voxel loss, [1] #(round(p) == y),
 * [2] #(round(p*mask) == y*mask)  (masked_accuracy numerator), [3] sum p over
 * y==0, [4] #(y==0), [5] sum (1-p) over y==1, [6] #(y==1), [7] #voxels */
int fpl_trainer_metric_sums(fpl_trainer *t, double out[FPL_N_METRIC_SUMS]);
/* Adam update (and moving-statistics update) from the gradient arena scaled by
 * grad_scale (1/world_size after a sum all-reduce) */
int fpl_trainer_apply(fpl_trainer *t, float grad_scale);
int fpl_trainer_grad_ptr(fpl_trainer *t, void **dev_ptr, int64_t *n_floats);
int fpl_trainer_get_weights(fpl_trainer *t, float *out, int64_t n_weights);
int fpl_trainer_set_weights(fpl_trainer *t, const float *w, int64_t n_weights);
/* optimizer state: Adam's first / second moments (weight-arena layout; the slots of the
 * moving statistics are unused) and the number of updates applied.  replaces: the
 * `optimizer_weights` group of Keras' model.save, through which the reference keeps its
 * optimizer across save_network / load_network and in the per-epoch files
 * (flypylib/fplnetwork.py:9-17,32-44,81-97) */
int fpl_trainer_get_opt_state(fpl_trainer *t, float *m, float *v, int64_t n_weights, int64_t *steps);
int fpl_trainer_set_opt_state(fpl_trainer *t, const float *m, const float *v, int64_t n_weights,
                              int64_t steps);
/* tensors of the last step, for parity tests: grads (host copy of the arena) */
int fpl_trainer_get_grads(fpl_trainer *t, float *out, int64_t n_weights);
/* overwrite the gradient arena from the host (host-staged reductions: towers that
 * share a device, or a process group without device collectives) */
int fpl_trainer_set_grads(fpl_trainer *t, const float *g, int64_t n_weights);

/* ---- data-parallel training: RCCL over xGMI ---------------------------------- */
/* replaces: the implicit gradient sum over the in-graph towers that
 * multi_gpu.make_parallel builds for FplNetwork.make_train_parallel
 * (flypylib/multi_gpu.py:20-61, flypylib/fplnetwork.py:124-128).  One communicator
 * per context (= per GPU); ranks are processes (torchrun, MPI, ...) or host threads
 * of one process, one per GPU.  Rank 0 makes the id, the host carries its 128 bytes
 * to the other ranks, every rank calls fpl_comm_init (collectively: it returns when
 * all nranks have joined).  librccl is dlopen'ed on first use (FPL_RCCL_LIB
 * overrides the search). */
#define FPL_COMM_ID_BYTES 128
int fpl_comm_unique_id(uint8_t out[FPL_COMM_ID_BYTES]);
int fpl_comm_init(fpl_ctx *ctx, int32_t rank, int32_t nranks,
                  const uint8_t unique_id[FPL_COMM_ID_BYTES]);
int fpl_comm_destroy(fpl_ctx *ctx);
/* tear the communicator down WITHOUT waiting for enqueued collectives (ncclCommAbort):
 * for the surviving ranks / towers after a peer failed, whose all-reduce would otherwise
 * block for ever; callable from another host thread than the one stuck in the collective */
int fpl_comm_abort(fpl_ctx *ctx);
/* nranks = 0 when the context has no communicator; lib_path = the librccl in use */
int fpl_comm_info(fpl_ctx *ctx, int32_t *rank, int32_t *nranks, char *lib_path,
                  size_t lib_path_cap);
/* in-place sum over all ranks / copy from `root`, enqueued on the context's stream */
int fpl_comm_allreduce_sum_f32(fpl_ctx *ctx, float *dev_ptr, int64_t n);
int fpl_comm_broadcast_f32(fpl_ctx *ctx, float *dev_ptr, int64_t n, int32_t root);
/* ONE all-reduce (sum) of the trainer's flat gradient arena (0.58 MB for vgg_like,
 * the moving-average deltas ride in it) between fpl_trainer_step and
 * fpl_trainer_apply(t, 1.0f / nranks); stream-ordered, no host synchronisation */
int fpl_allreduce_grads(fpl_trainer *t);
/* weights and Adam moments of every rank := rank `root`'s (start of training) */
int fpl_trainer_broadcast_state(fpl_trainer *t, int32_t root);

/* ---- synthetic data (bench / tests) -------------------- */
/* EM-like uint8 volume from a counter-based hash; bit-identical to
 * flypylib_amd.synth.em_volume_u8 on the host */
int fpl_synth_volume_u8(fpl_ctx *ctx, uint64_t seed, const int64_t dims[3],
                        const int64_t origin[3], uint8_t *dst, int dst_mem);

/* the same volume read as a substack + buffer: dims/origin may reach outside
 * [0, extent); voxels there are 0, as fri_get_image pads (fplobjdetect.py:1044-1070) */
int fpl_synth_substack_u8(fpl_ctx *ctx, uint64_t seed, const int64_t extent[3],
                          const int64_t dims[3], const int64_t origin[3],
                          uint8_t *dst, int dst_mem);

/* a substack + buffer cut out of a uint8 volume RESIDENT in device memory (`src`, extents
 * `extent`, e.g. a whole ROI: 4096^3 is 64 GiB of the 288): the crop + zero padding of
 * fri_get_image (fplobjdetect.py:1044-1070), device to device on the context's stream
 * (stream-ordered, returns without waiting) */
int fpl_crop_substack_u8(fpl_ctx *ctx, const uint8_t *src, const int64_t extent[3],
                         const int64_t dims[3], const int64_t origin[3], uint8_t *dst);

/* ---- substack normalisation (fri_get_image, fplobjdetect.py:1088-1107) -------- */
/* exact 256-bin histogram of a uint8 buffer; the raw / filtered (1 < v < 200) means
 * the reference takes with np.mean are exact functions of it */
int fpl_histogram_u8(fpl_ctx *ctx, const uint8_t *src, int src_mem, int64_t n,
                     uint64_t out[256]);

/* ---- timing ------------------------------------------------------------------ */
/* per-kernel accumulated HIP-event time since the last reset.  Names are
 * NUL-terminated, up to `cap` entries of 64 bytes each. */
int fpl_timing_enable(fpl_ctx *ctx, int on);
int fpl_timing_reset(fpl_ctx *ctx);
int fpl_timing_get(fpl_ctx *ctx, char *names, double *ms, int64_t *launches,
                   int32_t cap, int32_t *n);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FPLHIP_H */
