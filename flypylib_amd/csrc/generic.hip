// Generic fp32 per-op kernels + executor for any lowered layer program.
// This is the parity path for every architecture of flypylib/fplmodels.py and
// the fallback for shapes the fused MFMA kernels do not cover.  Activations are
// channels-last (N, D, H, W, C) fp32, as in the reference's Keras graphs
// (`Input(shape=in_sz+(1,))`, flypylib/fplmodels.py:105-108).
#include "conv_direct.h"
#include "program.h"

namespace {

__global__ void pool_f32(const float *__restrict__ x, float *__restrict__ y,
                         int64_t n_out, int D, int H, int W, int C, int od,
                         int oh, int ow, int f) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  int64_t t = i;
  const int c = (int)(t % C); t /= C;
  const int ox = (int)(t % ow); t /= ow;
  const int oy = (int)(t % oh); t /= oh;
  const int oz = (int)(t % od); t /= od;
  const int64_t b = t;
  float m = -INFINITY;
  for (int dz = 0; dz < f; ++dz)
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx)
        m = fmaxf(m, x[((((b * D + oz * f + dz) * H + oy * f + dy) * (int64_t)W +
                         ox * f + dx) * C) + c]);
  y[i] = m;
}

// crop (lo offsets) and nearest upsample share one index-remap kernel:
// y[b][z][y][x][c] = x[b][(z+lo0)/f0][(y+lo1)/f1][(x+lo2)/f2][c]
__global__ void remap_f32(const float *__restrict__ x, float *__restrict__ y,
                          int64_t n_out, int D, int H, int W, int C, int od,
                          int oh, int ow, int lo0, int lo1, int lo2, int f0,
                          int f1, int f2, int c_off, int c_total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  int64_t t = i;
  const int c = (int)(t % C); t /= C;
  const int ox = (int)(t % ow); t /= ow;
  const int oy = (int)(t % oh); t /= oh;
  const int oz = (int)(t % od); t /= od;
  const int64_t b = t;
  const float v = x[((((b * D + (oz + lo0) / f0) * H + (oy + lo1) / f1) *
                      (int64_t)W + (ox + lo2) / f2) * C) + c];
  y[((((b * od + oz) * oh + oy) * (int64_t)ow + ox) * c_total) + c_off + c] = v;
}

__global__ void add_f32(const float *__restrict__ a, const float *__restrict__ b,
                        float *__restrict__ y, int64_t n, int act) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = apply_act(a[i] + b[i], act);
}

inline dim3 grid1d(int64_t n, int block = 256) {
  return dim3((unsigned)ceil_div64(n, block));
}

}  // namespace

int fpl_infer_shapes(fpl_ctx *ctx, const fpl_program *prog,
                     const int32_t in_dims[3],
                     std::vector<TensorShape> *shapes) {
  shapes->assign(prog->n_tensors, TensorShape());
  (*shapes)[0] = TensorShape{in_dims[0], in_dims[1], in_dims[2], 1};
  for (size_t i = 0; i < prog->ops.size(); ++i) {
    const fpl_op &op = prog->ops[i];
    const TensorShape &a = (*shapes)[op.src0];
    TensorShape o = a;
    FPL_REQUIRE(ctx, a.c > 0, "op %zu reads tensor %d before it is produced", i,
                op.src0);
    switch (op.kind) {
      case FPL_OP_CONV:
        FPL_REQUIRE(ctx, a.c == op.cin, "op %zu: conv cin %d != input channels %d",
                    i, op.cin, a.c);
        o.d = a.d - (op.k - 1); o.h = a.h - (op.k - 1); o.w = a.w - (op.k - 1);
        o.c = op.cout;
        break;
      case FPL_OP_POOL:
        o.d = a.d / op.p[0]; o.h = a.h / op.p[1]; o.w = a.w / op.p[2];
        break;
      case FPL_OP_UP:
        o.d = a.d * op.p[0]; o.h = a.h * op.p[1]; o.w = a.w * op.p[2];
        break;
      case FPL_OP_CROP:
        o.d = a.d - op.p[0] - op.p[1]; o.h = a.h - op.p[2] - op.p[3];
        o.w = a.w - op.p[4] - op.p[5];
        break;
      case FPL_OP_CONCAT: {
        const TensorShape &b = (*shapes)[op.src1];
        FPL_REQUIRE(ctx, a.d == b.d && a.h == b.h && a.w == b.w,
                    "op %zu: concatenate of (%d,%d,%d) with (%d,%d,%d) - input "
                    "size is not compatible with this architecture", i, a.d, a.h,
                    a.w, b.d, b.h, b.w);
        o.c = a.c + b.c;
        break;
      }
      case FPL_OP_ADD: {
        const TensorShape &b = (*shapes)[op.src1];
        FPL_REQUIRE(ctx, a.d == b.d && a.h == b.h && a.w == b.w && a.c == b.c,
                    "op %zu: add of mismatched shapes", i);
        break;
      }
      default:
        return fpl_fail(ctx, "op %zu: unknown kind %d", i, op.kind);
    }
    FPL_REQUIRE(ctx, o.d > 0 && o.h > 0 && o.w > 0,
                "op %zu: input (%d,%d,%d) is too small for this architecture", i,
                in_dims[0], in_dims[1], in_dims[2]);
    (*shapes)[op.dst] = o;
  }
  return 0;
}

int fpl_forward_generic(fpl_ctx *ctx, fpl_program *prog, const float *in_dev,
                        int32_t n, const int32_t in_dims[3], float *out_dev) {
  std::vector<TensorShape> shp;
  FPL_TRY(fpl_infer_shapes(ctx, prog, in_dims, &shp));
  const int nt = prog->n_tensors;
  std::vector<float *> buf(nt, nullptr);
  std::vector<int> last_use(nt, -1);
  for (size_t i = 0; i < prog->ops.size(); ++i) {
    last_use[prog->ops[i].src0] = (int)i;
    if (prog->ops[i].src1 >= 0) last_use[prog->ops[i].src1] = (int)i;
  }
  buf[0] = const_cast<float *>(in_dev);
  DevTemp tmp(ctx);
  hipStream_t st = ctx->stream;
  const float *A = prog->arena_dev;
  for (size_t i = 0; i < prog->ops.size(); ++i) {
    const fpl_op &op = prog->ops[i];
    const TensorShape &a = shp[op.src0];
    const TensorShape &o = shp[op.dst];
    float *dst;
    if (op.dst == prog->out_tensor) {
      dst = out_dev;
    } else {
      void *p;
      FPL_TRY(tmp.alloc((size_t)n * o.elems() * sizeof(float), &p));
      dst = (float *)p;
    }
    buf[op.dst] = dst;
    const int64_t n_out = (int64_t)n * o.elems();
    switch (op.kind) {
      case FPL_OP_CONV: {
        const int64_t n_vox = (int64_t)n * o.voxels();
        TimedLaunch tl(ctx, op.k == 3 ? "generic_conv3_f32" : "generic_conv1_f32");
        if (op.cout % 16 == 0) {
          dim3 g((unsigned)ceil_div64(n_vox, 256), op.cout / 16);
          conv3d_direct_f32<16><<<g, 256, 0, st>>>(
              buf[op.src0], A + op.w_off, A + op.scale_off, A + op.shift_off,
              dst, n_vox, a.d, a.h, a.w, a.c, o.d, o.h, o.w, o.c, op.k, op.act);
        } else {
          dim3 g((unsigned)ceil_div64(n_vox, 256), op.cout);
          conv3d_direct_f32<1><<<g, 256, 0, st>>>(
              buf[op.src0], A + op.w_off, A + op.scale_off, A + op.shift_off,
              dst, n_vox, a.d, a.h, a.w, a.c, o.d, o.h, o.w, o.c, op.k, op.act);
        }
        break;
      }
      case FPL_OP_POOL: {
        FPL_REQUIRE(ctx, op.p[0] == op.p[1] && op.p[1] == op.p[2],
                    "anisotropic pooling is not supported");
        TimedLaunch tl(ctx, "generic_pool_f32");
        pool_f32<<<grid1d(n_out), 256, 0, st>>>(buf[op.src0], dst, n_out, a.d,
                                                a.h, a.w, a.c, o.d, o.h, o.w,
                                                op.p[0]);
        break;
      }
      case FPL_OP_UP: {
        TimedLaunch tl(ctx, "generic_remap_f32");
        remap_f32<<<grid1d(n_out), 256, 0, st>>>(buf[op.src0], dst, n_out, a.d,
                                                 a.h, a.w, a.c, o.d, o.h, o.w, 0,
                                                 0, 0, op.p[0], op.p[1], op.p[2],
                                                 0, o.c);
        break;
      }
      case FPL_OP_CROP: {
        TimedLaunch tl(ctx, "generic_remap_f32");
        remap_f32<<<grid1d(n_out), 256, 0, st>>>(buf[op.src0], dst, n_out, a.d,
                                                 a.h, a.w, a.c, o.d, o.h, o.w,
                                                 op.p[0], op.p[2], op.p[4], 1, 1,
                                                 1, 0, o.c);
        break;
      }
      case FPL_OP_CONCAT: {
        const TensorShape &b = shp[op.src1];
        TimedLaunch tl(ctx, "generic_remap_f32");
        const int64_t na = (int64_t)n * a.elems(), nb = (int64_t)n * b.elems();
        remap_f32<<<grid1d(na), 256, 0, st>>>(buf[op.src0], dst, na, a.d, a.h,
                                              a.w, a.c, o.d, o.h, o.w, 0, 0, 0, 1,
                                              1, 1, 0, o.c);
        remap_f32<<<grid1d(nb), 256, 0, st>>>(buf[op.src1], dst, nb, b.d, b.h,
                                              b.w, b.c, o.d, o.h, o.w, 0, 0, 0, 1,
                                              1, 1, a.c, o.c);
        break;
      }
      case FPL_OP_ADD: {
        TimedLaunch tl(ctx, "generic_add_f32");
        add_f32<<<grid1d(n_out), 256, 0, st>>>(buf[op.src0], buf[op.src1], dst,
                                               n_out, op.act);
        break;
      }
    }
    FPL_HIP(ctx, hipGetLastError());
    // release inputs whose last consumer this was
    for (int s : {op.src0, op.src1}) {
      if (s > 0 && last_use[s] == (int)i && buf[s] && s != prog->out_tensor) {
        tmp.release(buf[s]);
        buf[s] = nullptr;
      }
    }
  }
  return 0;
}

// ---- C ABI: program lifecycle + batch forward -----------------------------------
extern "C" {

int fpl_program_create(fpl_ctx *ctx, const fpl_op *ops, int32_t n_ops,
                       int32_t n_tensors, int32_t out_tensor, const float *arena,
                       int64_t n_arena, const int32_t stride[3],
                       fpl_program **out) {
  if (!ctx || !ops || !out || !arena || !stride)
    return fpl_fail(ctx, "fpl_program_create: NULL argument");
  *out = nullptr;
  FPL_REQUIRE(ctx, n_ops > 0 && n_tensors > 1 && out_tensor > 0 &&
                       out_tensor < n_tensors,
              "fpl_program_create: bad sizes (n_ops=%d n_tensors=%d out=%d)",
              n_ops, n_tensors, out_tensor);
  for (int i = 0; i < n_ops; ++i) {
    const fpl_op &op = ops[i];
    FPL_REQUIRE(ctx, op.src0 >= 0 && op.src0 < n_tensors && op.dst > 0 &&
                         op.dst < n_tensors && op.src1 < n_tensors,
                "fpl_program_create: op %d has tensor ids out of range", i);
    if (op.kind == FPL_OP_CONV) {
      FPL_REQUIRE(ctx, op.k == 1 || op.k == 3,
                  "fpl_program_create: op %d conv kernel %d (only 1 or 3)", i,
                  op.k);
      const int64_t kk = (int64_t)op.k * op.k * op.k * op.cin * op.cout;
      FPL_REQUIRE(ctx, op.w_off >= 0 && op.w_off + kk <= n_arena &&
                           op.scale_off >= 0 &&
                           op.scale_off + op.cout <= n_arena &&
                           op.shift_off >= 0 && op.shift_off + op.cout <= n_arena,
                  "fpl_program_create: op %d weight offsets exceed the arena", i);
    }
    if (op.kind == FPL_OP_POOL || op.kind == FPL_OP_UP)
      FPL_REQUIRE(ctx, op.p[0] > 0 && op.p[1] > 0 && op.p[2] > 0,
                  "fpl_program_create: op %d non-positive factor", i);
  }
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  fpl_program *p = new fpl_program();
  p->ctx = ctx;
  p->ops.assign(ops, ops + n_ops);
  p->n_tensors = n_tensors;
  p->out_tensor = out_tensor;
  for (int a = 0; a < 3; ++a) p->stride[a] = stride[a] > 0 ? stride[a] : 1;
  p->n_arena = n_arena;
  if (hipMalloc((void **)&p->arena_dev, (size_t)n_arena * sizeof(float)) !=
      hipSuccess) {
    delete p;
    return fpl_fail(ctx, "fpl_program_create: arena allocation failed");
  }
  int rc = fpl_program_set_arena(p, arena, n_arena);
  if (rc) {
    hipFree(p->arena_dev);
    delete p;
    return rc;
  }
  *out = p;
  return 0;
}

int fpl_program_set_arena(fpl_program *prog, const float *arena,
                          int64_t n_arena) {
  if (!prog || !arena) return fpl_fail(nullptr, "fpl_program_set_arena: NULL");
  fpl_ctx *ctx = prog->ctx;
  FPL_REQUIRE(ctx, n_arena == prog->n_arena,
              "fpl_program_set_arena: arena has %lld floats, program expects %lld",
              (long long)n_arena, (long long)prog->n_arena);
  prog->arena_host.assign(arena, arena + n_arena);
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  FPL_HIP(ctx, hipMemcpyAsync(prog->arena_dev, prog->arena_host.data(),
                              (size_t)n_arena * sizeof(float),
                              hipMemcpyHostToDevice, ctx->stream));
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  prog->arena_version++;
  return 0;
}

int fpl_program_destroy(fpl_program *prog) {
  if (!prog) return 0;
  fpl_ctx *ctx = prog->ctx;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (int k = 0; k < 3; ++k)
    if (prog->fast_state_h16[k] && prog->fast_state_h16_free[k])
      prog->fast_state_h16_free[k](ctx, prog->fast_state_h16[k]);
  if (prog->fast_state_f32 && prog->fast_state_f32_free)
    prog->fast_state_f32_free(ctx, prog->fast_state_f32);
  hipFree(prog->arena_dev);
  delete prog;
  return 0;
}

}  // extern "C"
