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

// blob centre of the 64^3 lattice cell (cz, cy, cx), jittered by the hash
struct SynthBlob {
  int bz, by, bx;
};
__device__ inline SynthBlob synth_blob(uint64_t seed, int cz, int cy, int cx) {
  const uint64_t hc = splitmix64((seed + 0x5851F42D4C957F2Dull) ^
                                 splitmix64(((uint64_t)(int64_t)cz << 42) |
                                            ((uint64_t)(int64_t)cy << 21) |
                                            (uint64_t)(int64_t)cx));
  return SynthBlob{(cz << 6) + 16 + (int)(hc & 31), (cy << 6) + 16 + (int)((hc >> 8) & 31),
                   (cx << 6) + 16 + (int)((hc >> 16) & 31)};
}

// coordinates are < 2^21 (checked by the callers), so they are ints here
__device__ inline int synth_voxel(uint64_t seed, int z, int y, int x, const SynthBlob &b) {
  const uint64_t idx = (((uint64_t)z << 42) | ((uint64_t)y << 21) | (uint64_t)x);
  const uint64_t h = splitmix64(seed ^ splitmix64(idx));
  const uint32_t lo = (uint32_t)h;
  const int s4 = (int)(lo & 255u) + (int)((lo >> 8) & 255u) + (int)((lo >> 16) & 255u) +
                 (int)(lo >> 24);
  int v = 128 + (((s4 - 510) * 57) >> 8);
  // one dark blob (radius 7) per 64^3 lattice cell
  const int dz = z - b.bz, dy = y - b.by, dx = x - b.bx;
  if (abs(dz) < 7 && abs(dy) < 7 && abs(dx) < 7) {
    const int d2 = dz * dz + dy * dy + dx * dx;
    if (d2 < 49) v -= (60 * (49 - d2)) / 49;
  }
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Four consecutive voxels of the flat (D0, D1, D2) box per thread, one 32-bit store.
// A workgroup's chunk starts at a flat index decoded once (64-bit); the threads step
// from there in 32 bits.  CLIPPED: voxels outside [0, E) of the global volume are 0, as
// fri_get_image (fplobjdetect.py:1044-1070) pads a substack + buffer at the faces.
constexpr int SYN_PER_WG = 256 * 4;

template <bool CLIPPED>
__global__ __launch_bounds__(256) void synth_u8(uint64_t seed, int E0, int E1, int E2, int D0,
                                                int D1, int D2, int o0, int o1, int o2,
                                                int64_t n, int64_t n_chunks,
                                                uint8_t *__restrict__ dst) {
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t base = c * SYN_PER_WG;                 // uniform
    const int64_t row0 = base / D2;
    const uint32_t x0 = (uint32_t)(base - row0 * D2);
    const uint32_t z0 = (uint32_t)(row0 / D1), y0 = (uint32_t)(row0 - (int64_t)z0 * D1);
    const uint32_t off = x0 + 4u * threadIdx.x;
    uint32_t dy = off / (uint32_t)D2;
    int x = (int)(off - dy * (uint32_t)D2);
    dy += y0;
    const uint32_t dz = dy / (uint32_t)D1;
    int y = (int)(dy - dz * (uint32_t)D1);
    int z = (int)(z0 + dz);
    const int64_t i = base + 4 * (int64_t)threadIdx.x;
    if (i >= n) continue;
    int cz = (z + o0) >> 6, cy = (y + o1) >> 6, cx = (x + o2) >> 6;
    SynthBlob blob = synth_blob(seed, cz, cy, cx);
    uint32_t word = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int gz = z + o0, gy = y + o1, gx = x + o2;
      if ((gz >> 6) != cz || (gy >> 6) != cy || (gx >> 6) != cx) {
        cz = gz >> 6, cy = gy >> 6, cx = gx >> 6;
        blob = synth_blob(seed, cz, cy, cx);
      }
      int v = 0;
      const bool in = !CLIPPED || (gz >= 0 && gy >= 0 && gx >= 0 && gz < E0 && gy < E1 && gx < E2);
      if (in && i + k < n) v = synth_voxel(seed, gz, gy, gx, blob);
      word |= (uint32_t)v << (8 * k);
      if (++x == D2) {
        x = 0;
        if (++y == D1) y = 0, ++z;
      }
    }
    if (i + 4 <= n) {
      *reinterpret_cast<uint32_t *>(dst + i) = word;
    } else {
      for (int k = 0; k < 4 && i + k < n; ++k) dst[i + k] = (uint8_t)(word >> (8 * k));
    }
  }
}

// A substack + buffer cut out of a volume that is RESIDENT in HBM (a 4096^3 uint8 ROI is 64 GiB of
// the 288): four consecutive voxels of the (D0, D1, D2) box per thread, one 32-bit store; voxels
// outside [0, E) are 0, as fri_get_image pads at the faces (fplobjdetect.py:1044-1070).
__global__ __launch_bounds__(256) void crop_u8(const uint8_t *__restrict__ src, int E0, int E1, int E2,
                                               int D0, int D1, int D2, int o0, int o1, int o2,
                                               int64_t n, int64_t n_chunks, uint8_t *__restrict__ dst) {
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t base = c * SYN_PER_WG;                 // uniform
    const int64_t row0 = base / D2;
    const uint32_t x0 = (uint32_t)(base - row0 * D2);
    const uint32_t z0 = (uint32_t)(row0 / D1), y0 = (uint32_t)(row0 - (int64_t)z0 * D1);
    const uint32_t off = x0 + 4u * threadIdx.x;
    uint32_t dy = off / (uint32_t)D2;
    int x = (int)(off - dy * (uint32_t)D2);
    dy += y0;
    const uint32_t dz = dy / (uint32_t)D1;
    int y = (int)(dy - dz * (uint32_t)D1);
    int z = (int)(z0 + dz);
    const int64_t i = base + 4 * (int64_t)threadIdx.x;
    if (i >= n) continue;
    uint32_t word = 0;
    const int gz0 = z + o0, gy0 = y + o1, gx0 = x + o2;
    if (x + 4 <= D2 && gz0 >= 0 && gy0 >= 0 && gx0 >= 0 && gz0 < E0 && gy0 < E1 && gx0 + 4 <= E2 && i + 4 <= n) {
      // the common case: four voxels of one source row (any alignment: a dword load on this target)
      __builtin_memcpy(&word, src + ((int64_t)gz0 * E1 + gy0) * E2 + gx0, 4);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int gz = z + o0, gy = y + o1, gx = x + o2;
        const bool in = gz >= 0 && gy >= 0 && gx >= 0 && gz < E0 && gy < E1 && gx < E2 && i + k < n;
        const uint32_t v = in ? src[((int64_t)gz * E1 + gy) * E2 + gx] : 0u;
        word |= v << (8 * k);
        if (++x == D2) {
          x = 0;
          if (++y == D1) y = 0, ++z;
        }
      }
    }
    if (i + 4 <= n) {
      *reinterpret_cast<uint32_t *>(dst + i) = word;
    } else {
      for (int k = 0; k < 4 && i + k < n; ++k) dst[i + k] = (uint8_t)(word >> (8 * k));
    }
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
    FPL_REQUIRE(ctx, dims[a] > 0 && dims[a] < ((int64_t)1 << 21) && extent[a] > 0 &&
                         extent[a] < ((int64_t)1 << 21) && origin[a] > -((int64_t)1 << 21) &&
                         origin[a] < ((int64_t)1 << 21),
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
  FPL_REQUIRE(ctx, ((uintptr_t)d & 3) == 0, "fpl_synth_substack_u8: dst must be 4-byte aligned");
  const int64_t n_chunks = ceil_div64(n, SYN_PER_WG);
  const unsigned grid = (unsigned)std::min<int64_t>(n_chunks, (int64_t)ctx->n_cu * 32);
  {
    TimedLaunch tl(ctx, "synth_u8");
    synth_u8<true><<<grid, 256, 0, ctx->stream>>>(
        seed, (int)extent[0], (int)extent[1], (int)extent[2], (int)dims[0], (int)dims[1],
        (int)dims[2], (int)origin[0], (int)origin[1], (int)origin[2], n, n_chunks, d);
  }
  FPL_HIP(ctx, hipGetLastError());
  if (dst_mem == FPL_MEM_HOST)
    FPL_HIP(ctx, hipMemcpyAsync(dst, d, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int fpl_crop_substack_u8(fpl_ctx *ctx, const uint8_t *src, const int64_t extent[3],
                                    const int64_t dims[3], const int64_t origin[3], uint8_t *dst) {
  if (!ctx || !src || !extent || !dims || !origin || !dst)
    return fpl_fail(ctx, "fpl_crop_substack_u8: NULL argument");
  for (int a = 0; a < 3; ++a)
    FPL_REQUIRE(ctx, dims[a] > 0 && dims[a] < ((int64_t)1 << 21) && extent[a] > 0 &&
                         extent[a] < ((int64_t)1 << 21) && origin[a] > -((int64_t)1 << 21) &&
                         origin[a] < ((int64_t)1 << 21),
                "fpl_crop_substack_u8: axis %d out of the 2^21 coordinate range", a);
  FPL_REQUIRE(ctx, ((uintptr_t)dst & 3) == 0, "fpl_crop_substack_u8: dst must be 4-byte aligned");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t n = dims[0] * dims[1] * dims[2];
  const int64_t n_chunks = ceil_div64(n, SYN_PER_WG);
  const unsigned grid = (unsigned)std::min<int64_t>(n_chunks, (int64_t)ctx->n_cu * 32);
  {
    TimedLaunch tl(ctx, "crop_u8");
    crop_u8<<<grid, 256, 0, ctx->stream>>>(src, (int)extent[0], (int)extent[1], (int)extent[2], (int)dims[0],
                                           (int)dims[1], (int)dims[2], (int)origin[0], (int)origin[1],
                                           (int)origin[2], n, n_chunks, dst);
  }
  FPL_HIP(ctx, hipGetLastError());
  return 0;                                     // stream-ordered: the consumers run on ctx->stream too
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
  FPL_REQUIRE(ctx, ((uintptr_t)d & 3) == 0, "fpl_synth_volume_u8: dst must be 4-byte aligned");
  const int64_t n_chunks = ceil_div64(n, SYN_PER_WG);
  const unsigned grid = (unsigned)std::min<int64_t>(n_chunks, (int64_t)ctx->n_cu * 32);
  {
    TimedLaunch tl(ctx, "synth_u8");
    synth_u8<false><<<grid, 256, 0, ctx->stream>>>(seed, 0, 0, 0, (int)dims[0], (int)dims[1],
                                                   (int)dims[2], (int)origin[0], (int)origin[1],
                                                   (int)origin[2], n, n_chunks, d);
  }
  FPL_HIP(ctx, hipGetLastError());
  if (dst_mem == FPL_MEM_HOST)
    FPL_HIP(ctx, hipMemcpyAsync(dst, d, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}
