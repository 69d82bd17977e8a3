// Micro-benchmark: store bandwidth of the voxel2obj z-pass pattern on gfx950.
// A workgroup of 320 threads owns 1280 contiguous bytes of a 636-float row and writes
// that piece in NZ consecutive planes (stride 636*636*4 B), versus the same bytes written
// as one linear stream.   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NZ, bool WAIT>
__global__ __launch_bounds__(512) void planes(float *out, int P0, int P1, int P2, int nxb, int nzb) {
  const unsigned nwork = (unsigned)nxb * nzb * P1;
  const unsigned per_xcd = (nwork + 7) / 8;
  const unsigned slot = blockIdx.x >> 3;
  const unsigned wid = (blockIdx.x & 7) * per_xcd + slot;
  if (slot >= per_xcd || wid >= nwork) return;
  const int zb = (int)(wid % (unsigned)nzb);
  const unsigned rest = wid / (unsigned)nzb;
  const int xb = (int)(rest % (unsigned)nxb), y = (int)(rest / (unsigned)nxb);
  const int x = xb * (int)blockDim.x + (int)threadIdx.x;
  if (x >= P2) return;
  float *dst = out + ((long)zb * NZ * P1 + y) * P2 + x;
  const long pplane = (long)P1 * P2;
  const float v = (float)x;
#pragma unroll
  for (int o = 0; o < NZ; ++o)
    if (zb * NZ + o < P0) dst[o * pplane] = v + o;
}

__global__ __launch_bounds__(256) void linear(float4 *out, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
    out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

__global__ __launch_bounds__(256) void copy4(const float4 *in, float4 *out, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
    out[i] = in[i];
}

int main() {
  const int P = 636;
  const long n = (long)P * P * P;
  float *d, *e;
  hipMalloc(&d, (n + 1024) * 4);
  hipMalloc(&e, (n + 1024) * 4);
  hipMemset(d, 0, n * 4);
  hipMemset(e, 0, n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto time = [&](const char *name, auto launch, double bytes) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.3f ms  %.2f TB/s\n", name, ms / 10, bytes / (ms / 10) / 1e9);
  };
  const int bd = 320, nxb = 2;
  {
    const int nzb = (P + 15) / 16;
    const unsigned grid = ((unsigned)nxb * nzb * P + 7) / 8 * 8;
    time("planes NZ=16 (z_win pattern)", [&] { planes<16, false><<<grid, bd>>>(d, P, P, P, nxb, nzb); }, n * 4.0);
  }
  {
    const int nzb = (P + 3) / 4;
    const unsigned grid = ((unsigned)nxb * nzb * P + 7) / 8 * 8;
    time("planes NZ=4", [&] { planes<4, false><<<grid, bd>>>(d, P, P, P, nxb, nzb); }, n * 4.0);
  }
  {
    const int nzb = P;
    const unsigned grid = ((unsigned)nxb * nzb * P + 7) / 8 * 8;
    time("planes NZ=1 (rows in raster order)", [&] { planes<1, false><<<grid, bd>>>(d, P, P, P, nxb, nzb); }, n * 4.0);
  }
  time("linear float4 stores", [&] { linear<<<256 * 16, 256>>>((float4 *)d, n / 4); }, n * 4.0);
  time("float4 copy (read + write bytes)", [&] { copy4<<<256 * 16, 256>>>((const float4 *)e, (float4 *)d, n / 4); }, n * 8.0);
  return 0;
}
