// What feeds a 16-bit MFMA K loop on gfx950 fastest?  Per step a wave issues NB + NWL ds_read_b128 (B fragments
// of the staged tile, weight fragments kept in LDS), NWG global_load_dwordx4 of weight fragments out of an L2-
// resident stream (every wave the same addresses, two steps ahead), and NM v_mfma_f32_16x16x32_f16 on the
// operands read a step earlier.  One persistent 8-wave workgroup per CU (2 waves per SIMD), as the all-LDS
// kernels run.  Prints the matrix pipe's share of the time: NM x 16 cycles x 2 waves against the measured
// cycles per step and SIMD.  The plain (non-split) vgg mid kernel is NB 12, weights 6, NM 36 per step: all six
// through LDS (round 5's experiment: LDS-read bound), all six per wave from L2 (the shipped kernel: TA-bound),
// or a mix.
//   hipcc --offload-arch=gfx950 -O3 kloop_feed.hip -o kloop_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int NB, int NWL, int NWG, int NM, int ND = 0>
__global__ __launch_bounds__(512, 2) void k(const unsigned char *wglob, float *out, int iters, long long *cyc) {
  constexpr int NACC = NM / 3;                 // every accumulator tile takes three MFMAs a step (R x MB tiles, 3 taps)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // LDS: 96 KiB of "tile" + 32 KiB of "weights", initialised once
  for (int i = tid; i < (128 << 10) / 16; i += 512) reinterpret_cast<u4 *>(smem)[i] = u4{(unsigned)i * 2654435761u, 0x3c003c00u, 0x3c003c00u, (unsigned)i};
  __syncthreads();
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  h8 b[2][NB > 0 ? NB : 1], wl[2][NWL > 0 ? NWL : 1], wg[3][NWG > 0 ? NWG : 1];
  const unsigned char *tile = smem + (wave * 4096 + lane * 16) % (64 << 10);
  const unsigned char *wlds = smem + (96 << 10) + lane * 16;
  const unsigned char *wp = wglob + lane * 16;
  // ND > 0: of the NB fragments only NB - ND are whole reads; per pair of derived ones (the dx = 1, 2 taps of a
  // row) one PATCH read with two of every 16 lanes active, and the derived fragments are made from a whole one
  // by two DPP moves per register (row_shl by the tap, the patched lanes by row_shr:15 / :14)
  auto loadB = [&](int s, int buf) {
#pragma unroll
    for (int i = 0; i < NB - ND; ++i) b[buf][i] = *reinterpret_cast<const h8 *>(tile + ((s * 7 + i * 11) % 32) * 1024);
    if (ND > 0) {
      h8 patch[ND / 2 > 0 ? ND / 2 : 1];
#pragma unroll
      for (int i = 0; i < ND / 2; ++i) {
        patch[i] = b[buf][i % (NB - ND)];
        if ((lane & 15) < 2) patch[i] = *reinterpret_cast<const h8 *>(tile + ((s * 3 + i * 5) % 32) * 1024 + 512);
      }
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const u4 src = __builtin_bit_cast(u4, b[buf][(i / 2) % (NB - ND)]), pt = __builtin_bit_cast(u4, patch[i / 2]);
        u4 d;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int v = (i & 1) ? __builtin_amdgcn_update_dpp((int)src[r], (int)src[r], 0x102, 0xF, 0xF, false)
                          : __builtin_amdgcn_update_dpp((int)src[r], (int)src[r], 0x101, 0xF, 0xF, false);
          v = (i & 1) ? __builtin_amdgcn_update_dpp(v, (int)pt[r], 0x11E, 0xF, 0xF, false)
                      : __builtin_amdgcn_update_dpp(v, (int)pt[r], 0x11F, 0xF, 0xF, false);
          d[r] = (unsigned)v;
        }
        b[buf][NB - ND + i] = __builtin_bit_cast(h8, d);
      }
    }
#pragma unroll
    for (int i = 0; i < NWL; ++i) wl[buf][i] = *reinterpret_cast<const h8 *>(wlds + ((s * 5 + i) % 32) * 1024);
  };
  auto loadG = [&](int s, int buf) {
#pragma unroll
    for (int i = 0; i < NWG; ++i) wg[buf][i] = *reinterpret_cast<const h8 *>(wp + (size_t)((s * NWG + i) % 1024) * 1024);
  };
  loadB(0, 0);
  loadG(0, 0);
  loadG(1, 1);
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; it += 6) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int s = it + u;
      loadB(s + 1, (u + 1) & 1);
      loadG(s + 2, (u + 2) % 3);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const h8 w = (NWL + NWG == 0) ? b[u & 1][0]
                     : (m % (NWL + NWG)) < NWL ? wl[u & 1][NWL ? (m % (NWL + NWG)) % NWL : 0]
                                               : wg[u % 3][NWG ? (m % (NWL + NWG) - NWL) % NWG : 0];
        acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, b[u & 1][NB ? m % NB : 0], acc[m % NACC], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0;
  for (int i = 0; i < NACC; ++i) sum += acc[i][0] + acc[i][2];
  out[blockIdx.x * 512 + tid] = sum;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NB, int NWL, int NWG, int NM, int ND = 0>
void run(const unsigned char *w, float *out, long long *cyc, const char *what) {
  const int iters = 6000, blocks = 256;
  hipFuncSetAttribute((const void *)k<NB, NWL, NWG, NM, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NB, NWL, NWG, NM, ND><<<blocks, 512, 128 << 10>>>(w, out, 600, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NB, NWL, NWG, NM, ND><<<blocks, 512, 128 << 10>>>(w, out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double tick = 0;
  for (long long v : h) tick += (double)v;
  tick /= blocks;
  // s_memtime counts a 100 MHz constant clock: convert through the wall time of the launch
  const double sec = ms * 1e-3, flop = 2.0 * 16 * 16 * 32 * NM * (double)iters * 8 * blocks;
  printf("%-44s B %2d (%d by DPP)  W lds %d  W L2 %d  MFMA %2d per step: %7.1f TFLOP/s = %4.1f %% of 2 500\n", what, NB, ND, NWL, NWG, NM,
         flop / sec / 1e12, 100.0 * flop / sec / 2.5e15);
  (void)tick;
}

int main() {
  unsigned char *w; float *out; long long *cyc;
  hipMalloc(&w, 1 << 20); hipMemset(w, 0x3c, 1 << 20);
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  run<0, 0, 0, 36>(w, out, cyc, "MFMAs alone");
  run<12, 0, 0, 36>(w, out, cyc, "tile reads only");
  run<12, 6, 0, 36>(w, out, cyc, "weights in LDS (round-5 experiment)");
  run<12, 5, 1, 36>(w, out, cyc, "5 of 6 in LDS");
  run<12, 4, 2, 36>(w, out, cyc, "4 of 6 in LDS");
  run<12, 3, 3, 36>(w, out, cyc, "3 of 6 in LDS");
  run<12, 2, 4, 36>(w, out, cyc, "2 of 6 in LDS");
  run<12, 0, 6, 36>(w, out, cyc, "weights per wave from L2 (shipped)");
  run<12, 6, 0, 36, 8>(w, out, cyc, "weights in LDS, dx taps by DPP");
  run<12, 0, 0, 36, 8>(w, out, cyc, "tile reads only, dx taps by DPP");
  run<12, 3, 3, 36, 8>(w, out, cyc, "3 of 6 in LDS, dx taps by DPP");
  return 0;
}
