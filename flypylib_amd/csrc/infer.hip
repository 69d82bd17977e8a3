// fpl_program_forward / fpl_infer_volume: the tile -> predict -> stitch lattice of
// FplNetwork.infer (flypylib/fplnetwork.py:136-189) on the device.
//
// Lattice (fplnetwork.py:146-159): out = tile_in - 2*off; tile origins (output
// coordinates) off, off+out, ... while < dim - off; input window
// [origin-off, min(origin+out+off, dim)), zero-padded (in the normalised domain)
// up to tile_in; valid outputs [origin, window_end - off).
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "fast_paths.h"
#include "program.h"

namespace {

typedef FplTileDesc TileDesc;

inline void set_last_path(fpl_ctx *ctx, const char *name) {
  snprintf(ctx->last_path, sizeof(ctx->last_path), "%s", name);
}

template <typename T>
__global__ void gather_tiles(const T *__restrict__ src, int64_t Y, int64_t X,
                             const TileDesc *__restrict__ tiles, int I0, int I1,
                             int I2, float mean, float sd, float *__restrict__ out,
                             int64_t n_total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total) return;
  int64_t t = i;
  const int x = (int)(t % I2); t /= I2;
  const int y = (int)(t % I1); t /= I1;
  const int z = (int)(t % I0); t /= I0;
  const TileDesc td = tiles[t];
  float v = 0.f;
  if (z < td.ext[0] && y < td.ext[1] && x < td.ext[2]) {
    const float raw = (float)src[((int64_t)(td.start[0] + z) * Y + td.start[1] + y) * X +
                                 td.start[2] + x];
    v = (raw - mean) / sd;
  }
  out[i] = v;
}

// dst[origin + p] = tile_out[p / stride] for p < ext - 2*off (per axis)
__global__ void stitch_tiles(const float *__restrict__ tile_out, int o0, int o1,
                             int o2, const TileDesc *__restrict__ tiles, int off0,
                             int off1, int off2, int s0, int s1, int s2, int F0,
                             int F1, int F2, float *__restrict__ dst, int64_t Y,
                             int64_t X, int64_t z_base, int64_t n_total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total) return;
  int64_t t = i;
  const int x = (int)(t % F2); t /= F2;
  const int y = (int)(t % F1); t /= F1;
  const int z = (int)(t % F0); t /= F0;
  const TileDesc td = tiles[t];
  if (z >= td.ext[0] - 2 * off0 || y >= td.ext[1] - 2 * off1 ||
      x >= td.ext[2] - 2 * off2)
    return;
  const float v =
      tile_out[((t * o0 + z / s0) * o1 + y / s1) * (int64_t)o2 + x / s2];
  dst[((int64_t)(td.start[0] + off0 + z - z_base) * Y + td.start[1] + off1 + y) * X +
      td.start[2] + off2 + x] = v;
}

// zero the rf_offset border shell of rows [z_lo, z_hi) of a (Z,Y,X) volume; one
// wave per (z,y) row: full rows inside the z/y shell, else the two x margins
__global__ void clear_shell(float *__restrict__ dst, int64_t Z, int64_t Y, int64_t X,
                            int64_t z_lo, int64_t z_hi, int o0, int o1, int o2) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= (z_hi - z_lo) * Y) return;
  const int64_t z = z_lo + row / Y, y = row % Y;
  float *p = dst + (z * Y + y) * X;
  if (z < o0 || z >= Z - o0 || y < o1 || y >= Y - o1) {
    for (int64_t x = lane; x < X; x += 64) p[x] = 0.f;
  } else {
    if (lane < o2) p[lane] = 0.f;
    if (lane < o2) p[X - o2 + lane] = 0.f;
  }
}

__global__ void upsample_out(const float *__restrict__ in, float *__restrict__ out,
                             int64_t n_total, int d, int h, int w, int c, int s0,
                             int s1, int s2) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total) return;
  int64_t t = i;
  const int cc = (int)(t % c); t /= c;
  const int x = (int)(t % (w * s2)); t /= (w * s2);
  const int y = (int)(t % (h * s1)); t /= (h * s1);
  const int z = (int)(t % (d * s0)); t /= (d * s0);
  out[i] = in[(((t * d + z / s0) * h + y / s1) * (int64_t)w + x / s2) * c + cc];
}

// may neighbouring tiles of the reference lattice be computed as one larger tile with the
// same results?  (no upsampling: then the output's stride is the deepest pooling level;
// every pool sees even extents; the lattice pitch is a multiple of the stride)
bool tiles_mergeable(const fpl_program *prog, const std::vector<TensorShape> &shp,
                     const int32_t out_sz[3]) {
  for (auto &op : prog->ops) {
    if (op.kind == FPL_OP_UP) return false;
    if (op.kind == FPL_OP_POOL) {
      const TensorShape &t = shp[op.src0];
      if (t.d % op.p[0] || t.h % op.p[1] || t.w % op.p[2]) return false;
    }
  }
  for (int a = 0; a < 3; ++a)
    if (out_sz[a] % prog->stride[a]) return false;
  return out_sz[0] == out_sz[1] && out_sz[1] == out_sz[2];
}

}  // namespace

extern "C" {

int fpl_program_forward(fpl_ctx *ctx, fpl_program *prog, const float *in,
                        int in_mem, int32_t n, const int32_t in_dims[3],
                        int precision, float *out, int out_mem,
                        int32_t out_dims[4]) {
  if (!ctx || !prog || !in || !in_dims)
    return fpl_fail(ctx, "fpl_program_forward: NULL argument");
  FPL_REQUIRE(ctx, n > 0, "fpl_program_forward: batch %d", n);
  FPL_REQUIRE(ctx, precision == FPL_PREC_F32,
              "fpl_program_forward: the per-op executor is fp32 only");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<TensorShape> shp;
  FPL_TRY(fpl_infer_shapes(ctx, prog, in_dims, &shp));
  const TensorShape o = shp[prog->out_tensor];
  const int *s = prog->stride;
  if (out_dims) {
    out_dims[0] = o.d * s[0]; out_dims[1] = o.h * s[1]; out_dims[2] = o.w * s[2];
    out_dims[3] = o.c;
  }
  if (!out) return 0;  // shape query
  DevTemp tmp(ctx);
  const int64_t in_elems = (int64_t)n * in_dims[0] * in_dims[1] * in_dims[2];
  const float *in_dev = in;
  if (in_mem == FPL_MEM_HOST) {
    void *p;
    FPL_TRY(tmp.alloc(in_elems * sizeof(float), &p));
    FPL_HIP(ctx, hipMemcpyAsync(p, in, in_elems * sizeof(float),
                                hipMemcpyHostToDevice, ctx->stream));
    in_dev = (const float *)p;
  }
  const int64_t net_elems = (int64_t)n * o.elems();
  const int64_t out_elems = net_elems * s[0] * s[1] * s[2];
  const bool up = s[0] * s[1] * s[2] != 1;
  float *out_dev = out;
  if (out_mem == FPL_MEM_HOST) {
    void *p;
    FPL_TRY(tmp.alloc(out_elems * sizeof(float), &p));
    out_dev = (float *)p;
  }
  float *net_out = out_dev;
  if (up) {
    void *p;
    FPL_TRY(tmp.alloc(net_elems * sizeof(float), &p));
    net_out = (float *)p;
  }
  const bool cubic = in_dims[0] == in_dims[1] && in_dims[1] == in_dims[2];
  if (cubic && fpl_mfma_f32_supported(prog) && !getenv("FPL_FORCE_PEROP")) {
    set_last_path(ctx, "mfma_f32");
    FPL_TRY(fpl_forward_mfma_f32(ctx, prog, in_dev, n, in_dims[0], net_out));
  } else {
    set_last_path(ctx, "perop_f32");
    FPL_TRY(fpl_forward_generic(ctx, prog, in_dev, n, in_dims, net_out));
  }
  if (up) {
    TimedLaunch tl(ctx, "upsample_out");
    upsample_out<<<(unsigned)ceil_div64(out_elems, 256), 256, 0, ctx->stream>>>(
        net_out, out_dev, out_elems, o.d, o.h, o.w, o.c, s[0], s[1], s[2]);
    FPL_HIP(ctx, hipGetLastError());
  }
  if (out_mem == FPL_MEM_HOST)
    FPL_HIP(ctx, hipMemcpyAsync(out, out_dev, out_elems * sizeof(float),
                                hipMemcpyDeviceToHost, ctx->stream));
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

}  // extern "C"

// one pass at a fixed precision (never FPL_PREC_AUTO).  *range_bits: the half-range guard's
// flag word after a FPL_PREC_F16S pass (mfma_util.h; 0 = every split value was finite)
static int infer_volume_impl(fpl_ctx *ctx, fpl_program *prog, const void *src,
                             int src_dtype, int src_mem, float mean, float sd,
                             const int64_t dims[3], const int32_t tile_in[3],
                             const int32_t offset[3], int precision, int32_t z_begin,
                             int32_t z_end, float *dst, int dst_mem, unsigned *range_bits) {
  *range_bits = 0u;
  int32_t out_sz[3];
  std::vector<int32_t> origins[3];
  for (int a = 0; a < 3; ++a) {
    FPL_REQUIRE(ctx, dims[a] > 0 && dims[a] < (int64_t)1 << 31,
                "fpl_infer_volume: bad dims[%d]=%lld", a, (long long)dims[a]);
    out_sz[a] = tile_in[a] - 2 * offset[a];
    FPL_REQUIRE(ctx, out_sz[a] > 0 && offset[a] >= 0,
                "fpl_infer_volume: tile %d / offset %d", tile_in[a], offset[a]);
    for (int64_t o = offset[a]; o < dims[a] - offset[a]; o += out_sz[a])
      origins[a].push_back((int32_t)o);
  }
  // network output for one tile must cover tile_in - 2*off after upsampling
  std::vector<TensorShape> shp;
  FPL_TRY(fpl_infer_shapes(ctx, prog, tile_in, &shp));
  const TensorShape o = shp[prog->out_tensor];
  const int *s = prog->stride;
  FPL_REQUIRE(ctx, o.c == 1, "fpl_infer_volume: network output has %d channels",
              o.c);
  FPL_REQUIRE(ctx, o.d * s[0] == out_sz[0] && o.h * s[1] == out_sz[1] &&
                       o.w * s[2] == out_sz[2],
              "network input shape does not match expected infer_sz: tile "
              "(%d,%d,%d) offset (%d,%d,%d) gives output (%d,%d,%d)*stride, "
              "expected (%d,%d,%d)", tile_in[0], tile_in[1], tile_in[2],
              offset[0], offset[1], offset[2], o.d, o.h, o.w, out_sz[0],
              out_sz[1], out_sz[2]);
  const int32_t nz = (int32_t)origins[0].size();
  const int32_t zb = z_begin < 0 ? 0 : z_begin;
  const int32_t ze = (z_end < 0 || z_end > nz) ? nz : z_end;
  const int64_t Z = dims[0], Y = dims[1], X = dims[2];
  const size_t esz = src_dtype == FPL_U8 ? 1 : 4;
  hipStream_t st = ctx->stream;
  DevTemp tmp(ctx);

  // rows of the volume this call reads / writes
  int64_t rd_lo = 0, rd_hi = 0, wr_lo = 0, wr_hi = 0;
  if (zb < ze) {
    rd_lo = origins[0][zb] - offset[0];
    rd_hi = std::min<int64_t>(origins[0][ze - 1] + out_sz[0] + offset[0], Z);
    wr_lo = origins[0][zb];
    wr_hi = rd_hi - offset[0];
  }
  // the first / last slab also clears the rf_offset border shell
  if (zb == 0) wr_lo = 0;
  if (ze == nz) wr_hi = Z;
  if (zb >= ze && !(zb == 0 && ze == nz)) { wr_lo = wr_hi = 0; }

  // destination rows on the device
  float *dst_dev = dst;        // indexed from row wr_lo when staged
  int64_t dst_base = 0;
  if (dst_mem == FPL_MEM_HOST) {
    void *p;
    FPL_TRY(tmp.alloc((size_t)(wr_hi - wr_lo) * Y * X * sizeof(float), &p));
    dst_dev = (float *)p;
    dst_base = wr_lo;
  }
  // the fused fast path writes every valid voxel itself: only the border shell
  // needs clearing there; the per-op path stitches tiles into a zeroed volume
  const bool cubic = tile_in[0] == tile_in[1] && tile_in[1] == tile_in[2];
  const bool unet_split_ok = cubic && fpl_unet_fast_available_f16s(prog, FPL_PREC_F16S);
  const bool split = fpl_split_path_available(prog, precision, offset, out_sz);
  FPL_REQUIRE(ctx, precision != FPL_PREC_F16S || split || unet_split_ok,
              "fpl_infer_volume: the split-half kernels exist for vgg_like / vgg_like2 (stride-4 lattice), the "
              "U-Net skeletons and layer programs of 3x3x3 / 1x1x1 convolutions with up to 64 (or 128) outputs, "
              "pools, upsamplings, crops, concatenations and adds on cubic tiles; use precision f32 (or 'auto') "
              "for this architecture");
  const bool fast = zb < ze && (split || fpl_fast_path_available_bf16(prog, precision, offset, out_sz) ||
                                fpl_fast_path_available_f16(prog, precision, offset, out_sz));
  if (wr_hi > wr_lo) {
    if (fast && offset[2] <= 64) {
      TimedLaunch tl(ctx, "clear_shell");
      const int64_t rows = (wr_hi - wr_lo) * Y;
      clear_shell<<<(unsigned)ceil_div64(rows, 4), 256, 0, st>>>(
          dst_dev - dst_base * Y * X, Z, Y, X, wr_lo, wr_hi, offset[0], offset[1],
          offset[2]);
      FPL_HIP(ctx, hipGetLastError());
    } else {
      FPL_HIP(ctx, hipMemsetAsync(dst_dev + (wr_lo - dst_base) * Y * X, 0,
                                  (size_t)(wr_hi - wr_lo) * Y * X * sizeof(float), st));
    }
  }
  set_last_path(ctx, "none");
  unsigned *range_flag = nullptr;
  if (precision == FPL_PREC_F16S) {
    FPL_TRY(fpl_range_flag(ctx, &range_flag));
    FPL_HIP(ctx, hipMemsetAsync(range_flag, 0, sizeof(unsigned), st));
  }
  if (zb >= ze) {
    if (dst_mem == FPL_MEM_HOST && wr_hi > wr_lo)
      FPL_HIP(ctx, hipMemcpyAsync(dst + wr_lo * Y * X, dst_dev,
                                  (size_t)(wr_hi - wr_lo) * Y * X * sizeof(float),
                                  hipMemcpyDeviceToHost, st));
    FPL_HIP(ctx, hipStreamSynchronize(st));
    return 0;
  }

  // source rows on the device
  const uint8_t *src_dev = (const uint8_t *)src;   // byte pointer to row 0
  int64_t src_base = 0;
  if (src_mem == FPL_MEM_HOST) {
    void *p;
    FPL_TRY(tmp.alloc((size_t)(rd_hi - rd_lo) * Y * X * esz, &p));
    FPL_HIP(ctx, hipMemcpyAsync(p, (const uint8_t *)src + rd_lo * Y * X * esz,
                                (size_t)(rd_hi - rd_lo) * Y * X * esz,
                                hipMemcpyHostToDevice, st));
    src_dev = (const uint8_t *)p;
    src_base = rd_lo;
  }

  // fused whole-slab fast paths (vgg_like): no tile batch, no stitch
  bool handled = false;
  if (split) {
    FPL_TRY(fpl_split_infer_volume(ctx, prog, src_dev - src_base * Y * X * esz, src_dtype, mean, sd,
                                   dims, origins, out_sz, zb, ze, dst_dev - dst_base * Y * X));
    handled = true;
    set_last_path(ctx, "vgg_split_f16");
  } else {
    FPL_TRY((precision == FPL_PREC_F16 ? fpl_fast_infer_volume_f16 : fpl_fast_infer_volume_bf16)(
        ctx, prog, src_dev - src_base * Y * X * esz, src_dtype, mean, sd, dims, tile_in,
        offset, precision, origins, out_sz, zb, ze, dst_dev - dst_base * Y * X, &handled));
    if (handled) set_last_path(ctx, precision == FPL_PREC_F16 ? "vgg_fused_f16" : "vgg_fused_bf16");
  }
  if (!handled) {
    const bool unet_bf16 = (fpl_unet_fast_available_bf16(prog, precision) ||
                            fpl_unet_fast_available_f16(prog, precision) ||
                            fpl_unet_fast_available_f16s(prog, precision)) &&
                           tile_in[0] == tile_in[1] && tile_in[1] == tile_in[2];
    const bool f32_mfma = precision == FPL_PREC_F32 && fpl_mfma_f32_supported(prog) &&
                          tile_in[0] == tile_in[1] && tile_in[1] == tile_in[2] &&
                          !getenv("FPL_FORCE_PEROP");
    FPL_REQUIRE(ctx, precision == FPL_PREC_F32 || unet_bf16,
                "fpl_infer_volume: no 16-bit MFMA kernels for this architecture yet; "
                "use precision f32");
    set_last_path(ctx, unet_bf16 ? (precision == FPL_PREC_F16S ? "unet_split_f16"
                                    : precision == FPL_PREC_F16 ? "unet_mfma_f16" : "unet_mfma_bf16")
                       : f32_mfma ? "mfma_f32" : "perop_f32");
    // Super-tiles (fp32 MFMA path): a network made of valid convolutions, pools, crops and
    // adds only is translation-equivariant on the lattice of its total stride, and the
    // reference lattice's pitch (tile - 2*off) is a multiple of that stride when every pool
    // sees even extents - so m x m x m neighbouring reference tiles computed as ONE tile of
    // m*pitch + 2*off give, voxel for voxel, the same arithmetic in the same order as the m^3
    // small ones (bit-identical: tests/test_gpu_cnn.py), while the halo that every tile
    // recomputes shrinks from (102/88)^3 = 1.56x to (542/528)^3 = 1.08x of the work
    // (vgg_like, m = 6).  m minimises the voxels computed under a per-tile memory cap.
    int32_t m = 1;
    std::vector<TensorShape> shp_s = shp;
    // (the graph executor of conv_mfma.hip - the 16-bit / split forms of such networks - likewise: its
    // per-voxel arithmetic does not depend on the voxel's place in a tile either; the U-Net skeletons
    // upsample and are never mergeable)
    if ((f32_mfma || unet_bf16) && tiles_mergeable(prog, shp, out_sz) && !getenv("FPL_NO_TILE_MERGE")) {
      const int64_t cnt[3] = {ze - zb, (int64_t)origins[1].size(), (int64_t)origins[2].size()};
      double best = -1;
      for (int32_t c = 1; c <= 8; ++c) {
        const int32_t tin = c * out_sz[0] + 2 * offset[0];
        const int32_t tin3[3] = {tin, tin, tin};
        std::vector<TensorShape> sc;
        if (fpl_infer_shapes(ctx, prog, tin3, &sc)) break;
        const TensorShape oc = sc[prog->out_tensor];
        if (oc.d * s[0] != c * out_sz[0]) break;
        int64_t bytes = 0, biggest = 0;
        for (auto &t : sc) { bytes += t.elems() * 4; biggest = std::max(biggest, t.elems()); }
        if (c > 1 && (bytes > ((int64_t)12 << 30) || biggest >= ((int64_t)1 << 30))) break;
        double cost = (double)tin * tin * tin;
        for (int a = 0; a < 3; ++a) cost *= (double)((cnt[a] + c - 1) / c);
        if (best < 0 || cost < best) { best = cost; m = c; shp_s = sc; }
      }
    }
    const int32_t tile_s[3] = {m * out_sz[0] + 2 * offset[0], m * out_sz[1] + 2 * offset[1],
                               m * out_sz[2] + 2 * offset[2]};
    const int32_t out_s[3] = {m * out_sz[0], m * out_sz[1], m * out_sz[2]};
    const TensorShape o_s = shp_s[prog->out_tensor];
    // tile list in the reference's order (z outer, x inner); a super-tile never reads or
    // writes past the rows of this call's slab
    std::vector<TileDesc> tiles;
    for (int32_t iz = zb; iz < ze; iz += m)
      for (size_t iy = 0; iy < origins[1].size(); iy += m)
        for (size_t ix = 0; ix < origins[2].size(); ix += m) {
          TileDesc td;
          const int32_t org[3] = {origins[0][iz], origins[1][iy], origins[2][ix]};
          const int64_t lim[3] = {rd_hi, dims[1], dims[2]};
          for (int a = 0; a < 3; ++a) {
            td.start[a] = org[a] - offset[a];
            const int64_t end = std::min<int64_t>((int64_t)org[a] + out_s[a] + offset[a], lim[a]);
            td.ext[a] = (int32_t)(end - td.start[a]);
          }
          tiles.push_back(td);
        }
    const int64_t n_tiles = (int64_t)tiles.size();
    void *p;
    FPL_TRY(tmp.alloc(n_tiles * sizeof(TileDesc), &p));
    TileDesc *tiles_dev = (TileDesc *)p;
    FPL_HIP(ctx, hipMemcpyAsync(tiles_dev, tiles.data(), n_tiles * sizeof(TileDesc),
                                hipMemcpyHostToDevice, st));
    // batch size from an activation budget (fp32 per-op path keeps every live
    // tensor of the batch in HBM)
    int64_t per_tile = 0;
    int64_t biggest = 1;
    for (auto &t : shp_s) {
      per_tile += t.elems() * (int64_t)sizeof(float);
      biggest = std::max(biggest, t.elems());
    }
    const int64_t budget = (int64_t)24 << 30;
    int64_t B = std::max<int64_t>(1, std::min<int64_t>(n_tiles, budget / std::max<int64_t>(per_tile, 1)));
    B = std::min<int64_t>(B, 64);
    if (m > 1) B = std::max<int64_t>(1, std::min<int64_t>(B, (((int64_t)1 << 31) - 1) / biggest));
    if (unet_bf16 && m == 1) B = std::min<int64_t>(n_tiles, 48);
    const int64_t tile_elems = (int64_t)tile_s[0] * tile_s[1] * tile_s[2];
    void *in_batch, *out_batch;
    FPL_TRY(tmp.alloc(B * tile_elems * sizeof(float), &in_batch));
    FPL_TRY(tmp.alloc(B * o_s.elems() * sizeof(float), &out_batch));
    const int64_t fine = (int64_t)out_s[0] * out_s[1] * out_s[2];
    for (int64_t t0 = 0; t0 < n_tiles; t0 += B) {
      const int64_t nb = std::min<int64_t>(B, n_tiles - t0);
      {
        TimedLaunch tl(ctx, "gather_tiles");
        const int64_t tot = nb * tile_elems;
        const unsigned g = (unsigned)ceil_div64(tot, 256);
        if (src_dtype == FPL_U8)
          gather_tiles<uint8_t><<<g, 256, 0, st>>>(
              src_dev - src_base * Y * X, Y, X, tiles_dev + t0, tile_s[0],
              tile_s[1], tile_s[2], mean, sd, (float *)in_batch, tot);
        else
          gather_tiles<float><<<g, 256, 0, st>>>(
              (const float *)src_dev - src_base * Y * X, Y, X, tiles_dev + t0,
              tile_s[0], tile_s[1], tile_s[2], mean, sd, (float *)in_batch,
              tot);
        FPL_HIP(ctx, hipGetLastError());
      }
      if (unet_bf16) {
        // the fused unet path writes its outputs straight into the volume
        FplTileIO io;
        io.Y = Y; io.X = X; io.tiles = tiles_dev + t0; io.dst = dst_dev;
        io.dst_z_base = dst_base; io.off = offset[0];
        FPL_TRY((precision == FPL_PREC_F16S ? fpl_unet_forward_f16s
                 : precision == FPL_PREC_F16 ? fpl_unet_forward_f16 : fpl_unet_forward_bf16)(
            ctx, prog, (const float *)in_batch, (int)nb, tile_s[0], nullptr, &io));
        continue;
      }
      if (f32_mfma)
        FPL_TRY(fpl_forward_mfma_f32(ctx, prog, (const float *)in_batch, (int)nb,
                                     tile_s[0], (float *)out_batch));
      else
        FPL_TRY(fpl_forward_generic(ctx, prog, (const float *)in_batch, (int32_t)nb,
                                    tile_in, (float *)out_batch));
      {
        TimedLaunch tl(ctx, "stitch_tiles");
        const int64_t tot = nb * fine;
        stitch_tiles<<<(unsigned)ceil_div64(tot, 256), 256, 0, st>>>(
            (const float *)out_batch, o_s.d, o_s.h, o_s.w, tiles_dev + t0, offset[0],
            offset[1], offset[2], s[0], s[1], s[2], out_s[0], out_s[1],
            out_s[2], dst_dev, Y, X, dst_base, tot);
        FPL_HIP(ctx, hipGetLastError());
      }
    }
  }
  if (range_flag) {
    // the guard's verdict first: a pass that left the half range is not worth copying out
    FPL_HIP(ctx, hipMemcpyAsync(ctx->range_flag_host, range_flag, sizeof(unsigned),
                                hipMemcpyDeviceToHost, st));
    FPL_HIP(ctx, hipStreamSynchronize(st));
    *range_bits = *ctx->range_flag_host;
    if (*range_bits) return 0;
  }
  if (dst_mem == FPL_MEM_HOST)
    FPL_HIP(ctx, hipMemcpyAsync(dst + wr_lo * Y * X, dst_dev,
                                (size_t)(wr_hi - wr_lo) * Y * X * sizeof(float),
                                hipMemcpyDeviceToHost, st));
  FPL_HIP(ctx, hipStreamSynchronize(st));
  return 0;
}

extern "C" {

int fpl_infer_volume(fpl_ctx *ctx, fpl_program *prog, const void *src,
                     int src_dtype, int src_mem, float mean, float sd,
                     const int64_t dims[3], const int32_t tile_in[3],
                     const int32_t offset[3], int precision, int32_t z_begin,
                     int32_t z_end, float *dst, int dst_mem) {
  if (!ctx || !prog || !src || !dims || !tile_in || !offset || !dst)
    return fpl_fail(ctx, "fpl_infer_volume: NULL argument");
  FPL_REQUIRE(ctx, src_dtype == FPL_U8 || src_dtype == FPL_F32,
              "fpl_infer_volume: src dtype must be u8 or f32");
  FPL_REQUIRE(ctx, sd != 0.f, "fpl_infer_volume: std is 0");
  FPL_REQUIRE(ctx, precision == FPL_PREC_F32 || precision == FPL_PREC_BF16 ||
                       precision == FPL_PREC_F16 || precision == FPL_PREC_F16S ||
                       precision == FPL_PREC_AUTO,
              "fpl_infer_volume: unknown precision %d", precision);
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  auto run = [&](int prec, unsigned *bits) -> int {
    // A host destination (FplNetwork.infer: 4 B per voxel back over PCIe, more time than the kernels take):
    // the tile rows go in up to six groups, and a helper thread copies a finished group out - a blocking
    // copy into the caller's pageable array, on a stream of its own - while the GPU computes the next
    // (520^3: 22 -> 16 ms host to host).  Groups of rows are independent (the N-GPU slabs: bit-identical).
    const int64_t Z = dims[0], Y = dims[1], X = dims[2];
    const int64_t out0 = (int64_t)tile_in[0] - 2 * offset[0];
    std::vector<int64_t> org;
    if (out0 > 0 && offset[0] >= 0)
      for (int64_t o = offset[0]; o < Z - offset[0]; o += out0) org.push_back(o);
    const int32_t nz = (int32_t)org.size();
    const int32_t zb = z_begin < 0 ? 0 : z_begin, ze = (z_end < 0 || z_end > nz) ? nz : z_end;
    const bool piped = dst_mem == FPL_MEM_HOST && ze - zb >= 2 && Y > 0 && X > 0 &&
                       (int64_t)(ze - zb) * out0 * Y * X * 4 >= ((int64_t)64 << 20) && !getenv("FPL_NO_D2H_PIPE");
    if (!piped)
      return infer_volume_impl(ctx, prog, src, src_dtype, src_mem, mean, sd, dims, tile_in, offset,
                               prec, z_begin, z_end, dst, dst_mem, bits);
    auto rows = [&](int32_t b, int32_t e, int64_t *lo, int64_t *hi) {      // as infer_volume_impl writes them
      *lo = b == 0 ? 0 : org[b];
      *hi = e == nz ? Z : std::min<int64_t>(org[e - 1] + out0 + offset[0], Z) - offset[0];
    };
    int64_t lo_t, hi_t;
    rows(zb, ze, &lo_t, &hi_t);
    DevTemp tmp(ctx);
    void *pd;
    FPL_TRY(tmp.alloc((size_t)(hi_t - lo_t) * Y * X * sizeof(float), &pd));
    float *vdst = (float *)pd - lo_t * Y * X;                                // "row 0" of the volume on the device
    // ... and a host SOURCE is uploaded by a second helper, a group ahead of the kernels that read it
    const size_t esz = src_dtype == FPL_U8 ? 1 : 4;
    const int32_t G = std::min<int32_t>(6, ze - zb);
    auto group = [&](int32_t g, int32_t *b, int32_t *e) {
      *b = zb + (int32_t)((int64_t)(ze - zb) * g / G);
      *e = zb + (int32_t)((int64_t)(ze - zb) * (g + 1) / G);
    };
    auto read_hi = [&](int32_t e) { return std::min<int64_t>(org[e - 1] + out0 + offset[0], Z); };
    const void *vsrc = src;
    int vsrc_mem = src_mem;
    const int64_t rd_lo_t = org[zb] - offset[0];
    if (src_mem == FPL_MEM_HOST) {
      void *ps;
      FPL_TRY(tmp.alloc((size_t)(read_hi(ze) - rd_lo_t) * Y * X * esz, &ps));
      vsrc = (const uint8_t *)ps - rd_lo_t * Y * X * esz;                   // "row 0" of the source on the device
      vsrc_mem = FPL_MEM_DEVICE;
    }
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<int64_t, int64_t>> todo;
    bool closed = false;
    int32_t uploaded = src_mem == FPL_MEM_HOST ? 0 : G;                     // groups whose source rows are resident
    hipError_t copy_err = hipSuccess, up_err = hipSuccess;
    const int device = ctx->device;
    std::thread uploader([&]() {
      if (src_mem != FPL_MEM_HOST) return;
      hipStream_t us = nullptr;
      hipError_t e = hipSetDevice(device);
      if (e == hipSuccess) e = hipStreamCreateWithFlags(&us, hipStreamNonBlocking);
      int64_t have = rd_lo_t;
      for (int32_t g = 0; g < G; ++g) {
        int32_t b, en;
        group(g, &b, &en);
        const int64_t need = read_hi(en);
        if (e == hipSuccess && need > have) {
          e = hipMemcpyAsync((uint8_t *)const_cast<void *>(vsrc) + have * Y * X * esz,
                             (const uint8_t *)src + have * Y * X * esz, (size_t)(need - have) * Y * X * esz,
                             hipMemcpyHostToDevice, us);
          if (e == hipSuccess) e = hipStreamSynchronize(us);
          have = need;
        }
        std::lock_guard<std::mutex> lk(mu);
        if (e != hipSuccess) up_err = e;
        uploaded = g + 1;                     // (on an error too: the main thread looks at up_err)
        cv.notify_all();
      }
      if (us) hipStreamDestroy(us);
    });
    std::thread copier([&]() {
      hipStream_t cs = nullptr;
      hipError_t e = hipSetDevice(device);
      if (e == hipSuccess) e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
      for (;;) {
        std::pair<int64_t, int64_t> r;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return closed || !todo.empty(); });
          if (todo.empty()) break;
          r = todo.front();
          todo.pop_front();
        }
        if (e != hipSuccess) continue;
        e = hipMemcpyAsync(dst + r.first * Y * X, vdst + r.first * Y * X,
                           (size_t)(r.second - r.first) * Y * X * sizeof(float), hipMemcpyDeviceToHost, cs);
        if (e == hipSuccess) e = hipStreamSynchronize(cs);
      }
      if (cs) hipStreamDestroy(cs);
      copy_err = e;
    });
    int rc = 0;
    *bits = 0u;
    for (int32_t g = 0; g < G && rc == 0 && !*bits; ++g) {
      int32_t b, e;
      group(g, &b, &e);
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return uploaded > g; });
        if (up_err != hipSuccess) break;
      }
      rc = infer_volume_impl(ctx, prog, vsrc, src_dtype, vsrc_mem, mean, sd, dims, tile_in, offset, prec, b, e,
                             vdst, FPL_MEM_DEVICE, bits);
      if (rc == 0 && !*bits) {
        int64_t lo, hi;
        rows(b, e, &lo, &hi);
        std::lock_guard<std::mutex> lk(mu);
        todo.emplace_back(lo, hi);
        cv.notify_all();
      }
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      closed = true;
      cv.notify_all();
    }
    uploader.join();
    copier.join();
    if (rc == 0 && up_err != hipSuccess)
      return fpl_fail(ctx, "fpl_infer_volume: host-to-device copy: %s", hipGetErrorString(up_err));
    if (rc == 0 && copy_err != hipSuccess)
      return fpl_fail(ctx, "fpl_infer_volume: device-to-host copy: %s", hipGetErrorString(copy_err));
    return rc;
  };
  auto where = [](unsigned bits) {
    return bits & FPL_RANGE_INPUT ? "a normalised input voxel (times the first layer's weights)"
           : bits & FPL_RANGE_STEM ? "an activation of the first block"
           : bits & FPL_RANGE_MID ? "an activation of the second block"
           : bits & FPL_RANGE_TAIL ? "an activation of the head" : "an activation";
  };
  unsigned bits = 0u;
  if (precision == FPL_PREC_AUTO) {
    // fp32-GRADE on the fastest executor that delivers it: the split-half kernels where they
    // exist and the values stay inside the IEEE-half range, else the fp32 MFMA executor - the
    // reference predicts in fp32 (flypylib/fplnetwork.py:175-176), which has no range limit
    int32_t out_sz[3];
    for (int a = 0; a < 3; ++a) out_sz[a] = tile_in[a] - 2 * offset[a];
    const bool cubic = tile_in[0] == tile_in[1] && tile_in[1] == tile_in[2];
    const bool have_split = fpl_split_path_available(prog, FPL_PREC_F16S, offset, out_sz) ||
                            (cubic && fpl_unet_fast_available_f16s(prog, FPL_PREC_F16S));
    if (have_split && prog->half_range_bad_version != prog->arena_version) {
      const int rc = run(FPL_PREC_F16S, &bits);
      if (rc == 0 && !bits) return 0;
      if (rc != 0 && rc != FPL_RC_RANGE && rc != FPL_RC_RANGE_CALL) return rc;
      // weights (FPL_RC_RANGE) or activations (bits) beyond the half range: this network runs in
      // fp32 from now on (until its weights change); an out-of-range INPUT voxel, or a
      // normalisation (mean / std of this call) the integer stem cannot take
      // (FPL_RC_RANGE_CALL), only costs this call
      if (rc == FPL_RC_RANGE || (bits & ~FPL_RANGE_INPUT)) prog->half_range_bad_version = prog->arena_version;
    }
    const int rc = run(FPL_PREC_F32, &bits);
    if (rc == 0 && have_split) {
      char name[64];
      snprintf(name, sizeof(name), "%s(range)", ctx->last_path);
      set_last_path(ctx, name);
    }
    return rc;
  }
  const int rc = run(precision, &bits);
  if (rc == FPL_RC_RANGE || rc == FPL_RC_RANGE_CALL) return 1;   // message set by the packer
  if (rc == 0 && bits)
    return fpl_fail(ctx, "fpl_infer_volume: %s exceeds the IEEE-half range (65504) of the split-operand "
                         "kernels (guard bits 0x%x): the result is not valid; use precision f32, or "
                         "'auto', which falls back to it", where(bits), bits);
  return rc;
}

const char *fpl_last_path(fpl_ctx *ctx) { return ctx ? ctx->last_path : ""; }

}  // extern "C"
