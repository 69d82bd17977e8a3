// Synthetic EM-like uint8 volume from a counter-based hash.
// Pure function of (seed, global z, y, x): bit-identical to
// flypylib_amd/synth.py::em_volume_u8, so 4096^3 never has to be stored or shipped.
#include "common.h"

namespace {

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ inline int synth_voxel(uint64_t seed, int64_t z, int64_t y, int64_t x) {
  const uint64_t idx = (((uint64_t)z << 42) | ((uint64_t)y << 21) | (uint64_t)x);
  const uint64_t h = splitmix64(seed ^ splitmix64(idx));
  const int s4 = (int)(h & 255) + (int)((h >> 8) & 255) + (int)((h >> 16) & 255) +
                 (int)((h >> 24) & 255);
  int v = 128 + (((s4 - 510) * 57) >> 8);
  // one dark blob (radius 7) per 64^3 lattice cell, centre jittered by the hash
  const int64_t cz = z >> 6, cy = y >> 6, cx = x >> 6;
  const uint64_t hc = splitmix64((seed + 0x5851F42D4C957F2Dull) ^
                                 splitmix64(((uint64_t)cz << 42) |
                                            ((uint64_t)cy << 21) | (uint64_t)cx));
  const int64_t bz = (cz << 6) + 16 + (int64_t)(hc & 31);
  const int64_t by = (cy << 6) + 16 + (int64_t)((hc >> 8) & 31);
  const int64_t bx = (cx << 6) + 16 + (int64_t)((hc >> 16) & 31);
  const int64_t d2 = (z - bz) * (z - bz) + (y - by) * (y - by) + (x - bx) * (x - bx);
  if (d2 < 49) v -= (int)((60 * (49 - d2)) / 49);
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ void synth_u8(uint64_t seed, int64_t D0, int64_t D1, int64_t D2,
                         int64_t o0, int64_t o1, int64_t o2,
                         uint8_t *__restrict__ dst) {
  const int64_t n = D0 * D1 * D2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride) {
    const int64_t x = i % D2, y = (i / D2) % D1, z = i / (D2 * D1);
    dst[i] = (uint8_t)synth_voxel(seed, z + o0, y + o1, x + o2);
  }
}

// substack read of the synthetic volume clipped to [0, extent): zeros outside, as
// fri_get_image (fplobjdetect.py:1044-1070) pads a substack + buffer at the faces
__global__ void synth_clipped_u8(uint64_t seed, int64_t E0, int64_t E1, int64_t E2,
                                 int64_t D0, int64_t D1, int64_t D2, int64_t o0,
                                 int64_t o1, int64_t o2, uint8_t *__restrict__ dst) {
  const int64_t n = D0 * D1 * D2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride) {
    const int64_t x = i % D2 + o2, y = (i / D2) % D1 + o1, z = i / (D2 * D1) + o0;
    const bool in = z >= 0 && y >= 0 && x >= 0 && z < E0 && y < E1 && x < E2;
    dst[i] = in ? (uint8_t)synth_voxel(seed, z, y, x) : (uint8_t)0;
  }
}

// 256-bin histogram: per-wave private bins in LDS, one atomic per bin per block
__global__ __launch_bounds__(256) void hist_u8(const uint8_t *__restrict__ src, int64_t n,
                                               unsigned long long *__restrict__ out) {
  __shared__ unsigned int bins[4][256];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 256; i += 256) (&bins[0][0])[i] = 0u;
  __syncthreads();
  const int64_t n16 = n / 16;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint4 *s16 = reinterpret_cast<const uint4 *>(src);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    const uint4 v = s16[i];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int b = 0; b < 4; ++b) atomicAdd(&bins[wave][(w[q] >> (8 * b)) & 255u], 1u);
  }
  if (blockIdx.x == 0)
    for (int64_t i = n16 * 16 + threadIdx.x; i < n; i += 256) atomicAdd(&bins[wave][src[i]], 1u);
  __syncthreads();
  const unsigned t = bins[0][threadIdx.x] + bins[1][threadIdx.x] + bins[2][threadIdx.x] +
                     bins[3][threadIdx.x];
  if (t) atomicAdd(&out[threadIdx.x], (unsigned long long)t);
}

}  // namespace

extern "C" int fpl_synth_substack_u8(fpl_ctx *ctx, uint64_t seed, const int64_t extent[3],
                                     const int64_t dims[3], const int64_t origin[3],
                                     uint8_t *dst, int dst_mem) {
  if (!ctx || !extent || !dims || !origin || !dst)
    return fpl_fail(ctx, "fpl_synth_substack_u8: NULL argument");
  for (int a = 0; a < 3; ++a)
    FPL_REQUIRE(ctx, dims[a] > 0 && extent[a] > 0 && extent[a] < ((int64_t)1 << 21),
                "fpl_synth_substack_u8: axis %d out of the 2^21 coordinate range", a);
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t n = dims[0] * dims[1] * dims[2];
  DevTemp tmp(ctx);
  uint8_t *d = dst;
  if (dst_mem == FPL_MEM_HOST) {
    void *p;
    FPL_TRY(tmp.alloc((size_t)n, &p));
    d = (uint8_t *)p;
  }
  const unsigned grid =
      (unsigned)std::min<int64_t>(ceil_div64(n, 256), (int64_t)ctx->n_cu * 32);
  {
    TimedLaunch tl(ctx, "synth_u8");
    synth_clipped_u8<<<grid, 256, 0, ctx->stream>>>(seed, extent[0], extent[1], extent[2],
                                                    dims[0], dims[1], dims[2], origin[0],
                                                    origin[1], origin[2], d);
  }
  FPL_HIP(ctx, hipGetLastError());
  if (dst_mem == FPL_MEM_HOST)
    FPL_HIP(ctx, hipMemcpyAsync(dst, d, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int fpl_histogram_u8(fpl_ctx *ctx, const uint8_t *src, int src_mem, int64_t n,
                                uint64_t out[256]) {
  if (!ctx || !src || !out) return fpl_fail(ctx, "fpl_histogram_u8: NULL argument");
  FPL_REQUIRE(ctx, n >= 0, "fpl_histogram_u8: n %lld", (long long)n);
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  DevTemp tmp(ctx);
  const uint8_t *d = src;
  if (src_mem == FPL_MEM_HOST) {
    void *p;
    FPL_TRY(tmp.alloc((size_t)std::max<int64_t>(n, 16), &p));
    FPL_HIP(ctx, hipMemcpyAsync(p, src, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    d = (const uint8_t *)p;
  }
  FPL_REQUIRE(ctx, ((uintptr_t)d & 15) == 0, "fpl_histogram_u8: source must be 16-byte aligned");
  void *hv;
  FPL_TRY(tmp.alloc(256 * sizeof(unsigned long long), &hv));
  FPL_HIP(ctx, hipMemsetAsync(hv, 0, 256 * sizeof(unsigned long long), ctx->stream));
  const unsigned grid =
      (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div64(n / 16, 256), (int64_t)ctx->n_cu * 8));
  {
    TimedLaunch tl(ctx, "hist_u8");
    hist_u8<<<grid, 256, 0, ctx->stream>>>(d, n, (unsigned long long *)hv);
  }
  FPL_HIP(ctx, hipGetLastError());
  FPL_HIP(ctx, hipMemcpyAsync(out, hv, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int fpl_synth_volume_u8(fpl_ctx *ctx, uint64_t seed,
                                   const int64_t dims[3], const int64_t origin[3],
                                   uint8_t *dst, int dst_mem) {
  if (!ctx || !dims || !origin || !dst)
    return fpl_fail(ctx, "fpl_synth_volume_u8: NULL argument");
  for (int a = 0; a < 3; ++a)
    FPL_REQUIRE(ctx, dims[a] > 0 && origin[a] >= 0 &&
                         origin[a] + dims[a] < ((int64_t)1 << 21),
                "fpl_synth_volume_u8: axis %d out of the 2^21 coordinate range", a);
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t n = dims[0] * dims[1] * dims[2];
  DevTemp tmp(ctx);
  uint8_t *d = dst;
  if (dst_mem == FPL_MEM_HOST) {
    void *p;
    FPL_TRY(tmp.alloc((size_t)n, &p));
    d = (uint8_t *)p;
  }
  const unsigned grid =
      (unsigned)std::min<int64_t>(ceil_div64(n, 256), (int64_t)ctx->n_cu * 32);
  {
    TimedLaunch tl(ctx, "synth_u8");
    synth_u8<<<grid, 256, 0, ctx->stream>>>(seed, dims[0], dims[1], dims[2],
                                            origin[0], origin[1], origin[2], d);
  }
  FPL_HIP(ctx, hipGetLastError());
  if (dst_mem == FPL_MEM_HOST)
    FPL_HIP(ctx, hipMemcpyAsync(dst, d, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}
