// voxel2obj device stages (flypylib/fplobjdetect.py:158-231): pad, Gaussian
// smoothing bit-identical to scipy.ndimage.gaussian_filter on float32, margin
// zeroing, exact order statistics (radix select) for np.percentile, and a
// round-based parallel form of the greedy radius NMS.
//
// Smoothing arithmetic (scipy ni_filters.c, symmetric branch; restated and
// checked in oracle/voxel2obj_oracle.py::gaussian_filter_restated): per axis
// 0,1,2: acc = x[0]*w[0]; for j = R..1: acc += (x[-j] + x[+j]) * w[j], all in
// fp64 with separate multiply and add (no FMA), rounded to fp32 per axis,
// 'reflect' boundary.
//
// NMS: the reference repeatedly takes the arg-max of the live candidates (ties:
// lowest flat index) and kills every candidate within distance r.  That set is
// the lexicographically-first maximal independent set in (value desc, index asc)
// order, so any candidate that beats every live candidate of a region covering
// its r-ball can be selected in parallel.  Per round: per-cell (4^3) best live
// key -> separable 15-cell window max (covers >= the ball for r <= 28; window
// scales with r) -> cells whose best equals the window max are winners -> their
// balls are cleared.  The global best always wins, so every round progresses.
#include <algorithm>

#include "common.h"

// The smoothing must round every product and every sum on its own, as scipy's C does on
// x86-64.  hipcc's default for device code fuses a * b + c into v_fma_f64 across
// statements and inlined functions, and `__dmul_rn` / `__dadd_rn` are plain `*` / `+` in
// this toolchain's headers, so they do not stop it.  Two guards: the build passes
// -ffp-contract=on (csrc/build.py), and this file switches contraction off and routes the
// arithmetic through mul_rn / add_rn below, compiled under the pragma, so that the
// property does not hang on a command line.  tests/test_host_logic.py disassembles this
// file and fails on any fp64 FMA; tests/test_gpu_voxel2obj.py holds volumes on which
// fused forms differ from scipy in a float32 voxel (about one per 3e9 voxel-passes).
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ double mul_rn(double a, double b) { return a * b; }
__device__ __forceinline__ double add_rn(double a, double b) { return a + b; }

constexpr int CELL = 4;

__device__ __forceinline__ int64_t reflect_idx(int64_t i, int64_t n) {
  const int64_t p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return i >= n ? p - 1 - i : i;
}

// single-bounce 'reflect' (-1 -> 0, n -> n - 1) for -n <= i < 2n
__device__ __forceinline__ int reflect1(int i, int n) {
  return i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i);
}
// virtual zero-padded view of the unpadded prediction
struct PadView {
  const float *pred;
  int64_t D0, D1, D2;
  int r;
  __device__ __forceinline__ float at(int64_t z, int64_t y, int64_t x) const {
    z -= r; y -= r; x -= r;
    if (z < 0 || y < 0 || x < 0 || z >= D0 || y >= D1 || x >= D2) return 0.f;
    return pred[(z * D1 + y) * D2 + x];
  }
};

// One separable pass.  AXIS 0 reads the virtual padded input; AXIS 2 also
// zeroes the outer r shell on store.
template <int AXIS>
__global__ __launch_bounds__(256) void gauss_pass(
    PadView pv, const float *__restrict__ in, float *__restrict__ out, int64_t P0,
    int64_t P1, int64_t P2, const double *__restrict__ w, int wr, int r) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P0 * P1 * P2) return;
  const int64_t x = i % P2, y = (i / P2) % P1, z = i / (P2 * P1);
  auto load = [&](int64_t d) -> double {
    if (AXIS == 0) return (double)pv.at(reflect_idx(z + d, P0), y, x);
    if (AXIS == 1) return (double)in[(z * P1 + reflect_idx(y + d, P1)) * P2 + x];
    return (double)in[(z * P1 + y) * P2 + reflect_idx(x + d, P2)];
  };
  double acc = mul_rn(load(0), w[0]);
  for (int j = wr; j >= 1; --j)
    acc = add_rn(acc, mul_rn(add_rn(load(-j), load(j)), w[j]));
  float v = (float)acc;
  if (AXIS == 2 && r > 0 &&
      (z < r || y < r || x < r || z >= P0 - r || y >= P1 - r || x >= P2 - r))
    v = 0.f;
  out[i] = v;
}

// Register-window form of one pass: a thread produces OUT consecutive outputs along
// AXIS from OUT + 2*WR loaded values (3.5 loads per output for WR = 10 instead of
// 21; 2.25 with 16 outputs per thread); lanes run along x, so the loads of
// passes 0 and 1 are coalesced.  Same
// arithmetic order as gauss_pass.
template <int AXIS, int WR>
__global__ __launch_bounds__(256) void gauss_pass_win(
    PadView pv, const float *__restrict__ in, float *__restrict__ out, int64_t P0,
    int64_t P1, int64_t P2, const double *__restrict__ w, int r) {
  constexpr int OUT = AXIS == 2 ? 4 : 16;
  constexpr int NW = OUT + 2 * WR;
  const int64_t PA = AXIS == 0 ? P0 : (AXIS == 1 ? P1 : P2);
  const int64_t nblk = (PA + OUT - 1) / OUT;
  // thread grid: (other two axes, x fastest) x blocks along AXIS
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t z, y, x, a0;
  if (AXIS == 0) {
    // x fastest, then the blocks of one (y) row along z, then y: consecutive
    // workgroups share their z halo planes of that row through L2
    if (tid >= nblk * P1 * P2) return;
    x = tid % P2; a0 = ((tid / P2) % nblk) * OUT; y = tid / (P2 * nblk); z = a0;
  } else if (AXIS == 1) {
    if (tid >= P0 * nblk * P2) return;
    x = tid % P2; a0 = ((tid / P2) % nblk) * OUT; z = tid / (P2 * nblk); y = a0;
  } else {
    if (tid >= P0 * P1 * nblk) return;
    a0 = (tid % nblk) * OUT; y = (tid / nblk) % P1; z = tid / (nblk * P1); x = a0;
  }
  double win[NW];
  const bool interior = a0 - WR >= 0 && a0 + OUT + WR <= PA;
  if (AXIS == 0) {
    // the (y, x) column of the unpadded prediction is fixed per thread: one bounds
    // test and one base pointer, then a constant plane stride along z
    const int64_t yy = y - pv.r, xx = x - pv.r;
    const bool col_ok = yy >= 0 && xx >= 0 && yy < pv.D1 && xx < pv.D2;
    const int64_t plane = pv.D1 * pv.D2;
    const float *col = pv.pred + (col_ok ? yy * pv.D2 + xx : 0);
    const int64_t z_lo = a0 - WR - pv.r;                 // unpadded z of win[0]
    if (col_ok && interior && z_lo >= 0 && z_lo + NW <= pv.D0) {
      const float *q = col + z_lo * plane;
#pragma unroll
      for (int i = 0; i < NW; ++i) win[i] = (double)q[i * plane];
    } else {
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        int64_t p = a0 - WR + i;
        if (!interior) p = reflect_idx(p, PA);
        const int64_t zz = p - pv.r;
        win[i] = (col_ok && zz >= 0 && zz < pv.D0) ? (double)col[zz * plane] : 0.0;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      int64_t p = a0 - WR + i;
      if (!interior) p = reflect_idx(p, PA);
      const float v = AXIS == 1 ? in[(z * P1 + p) * P2 + x] : in[(z * P1 + y) * P2 + p];
      win[i] = (double)v;
    }
  }
  double wk[WR + 1];
#pragma unroll
  for (int j = 0; j <= WR; ++j) wk[j] = w[j];
#pragma unroll
  for (int o = 0; o < OUT; ++o) {
    if (a0 + o >= PA) break;
    double acc = mul_rn(win[WR + o], wk[0]);
#pragma unroll
    for (int j = WR; j >= 1; --j)
      acc = add_rn(acc, mul_rn(add_rn(win[WR + o - j], win[WR + o + j]), wk[j]));
    float v = (float)acc;
    const int64_t oz = AXIS == 0 ? a0 + o : z, oy = AXIS == 1 ? a0 + o : y,
                  ox = AXIS == 2 ? a0 + o : x;
    if (AXIS == 2 && r > 0 &&
        (oz < r || oy < r || ox < r || oz >= P0 - r || oy >= P1 - r || ox >= P2 - r))
      v = 0.f;
    out[(oz * P1 + oy) * P2 + ox] = v;
  }
}

// x pass through LDS: a workgroup takes GX_ROWS (z,y) rows x GX_SEG outputs; each
// row segment + 2*WR halo (reflected at the row ends) is loaded coalesced once, a
// thread then reads its 4 + 2*WR inputs as aligned 16-B LDS pieces - one global
// load per input instead of (4 + 2*WR) / 4 strided ones.  Arithmetic as gauss_pass;
// the outer r shell is zeroed on store.
constexpr int GX_ROWS = 8, GX_SEG = 128;
// first radix level of the order statistics: the top L0_BITS key bits, histogrammed by
// the last smoothing pass (10 bits = 4 KiB of LDS: three workgroups of the fused pass
// per CU)
constexpr int L0_BITS = 10, L0_SHIFT = 32 - L0_BITS, L0_BINS = 1 << L0_BITS;

__device__ __forceinline__ uint32_t float_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int WR>
__global__ __launch_bounds__(256) void gauss_x_lds(
    const float *__restrict__ in, float *__restrict__ out, int64_t P0, int64_t P1,
    int64_t P2, const double *__restrict__ w, int r,
    unsigned long long *__restrict__ hist0) {
  constexpr int NW = 4 + 2 * WR;
  constexpr int LROW = (GX_SEG + 2 * WR + 3) / 4 * 4 + 4;     // floats, 16-B multiple
  __shared__ __attribute__((aligned(16))) float tile[GX_ROWS][LROW];
  // first level of the radix select (top L0_BITS key bits of every smoothed voxel),
  // taken while the values are in registers: saves one scan of the volume
  __shared__ unsigned int lh[L0_BINS];
  const int t = threadIdx.x, rl = t >> 5, tx = t & 31;
  for (int b = t; b < L0_BINS; b += 256) lh[b] = 0u;
  double wk[WR + 1];
#pragma unroll
  for (int j = 0; j <= WR; ++j) wk[j] = w[j];
  const int64_t nseg = (P2 + GX_SEG - 1) / GX_SEG;
  const int64_t nrows = P0 * P1;
  const int64_t ntiles = ((nrows + GX_ROWS - 1) / GX_ROWS) * nseg;
  for (int64_t tileid = blockIdx.x; tileid < ntiles; tileid += gridDim.x) {
    const int64_t seg0 = (tileid % nseg) * GX_SEG;
    const int64_t row = (tileid / nseg) * GX_ROWS + rl;
    __syncthreads();                       // previous tile consumed (and lh zeroed)
    if (row < nrows) {
      const float *src = in + row * P2;
      for (int i = tx; i < GX_SEG + 2 * WR; i += 32) {
        int64_t p = seg0 - WR + i;
        if (p < 0 || p >= P2) p = reflect_idx(p, P2);
        tile[rl][i] = src[p];
      }
    }
    __syncthreads();
    const int64_t x0 = seg0 + 4 * tx;
    if (row >= nrows || x0 >= P2) continue;
    double win[NW];
#pragma unroll
    for (int q = 0; q < (NW + 3) / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(&tile[rl][4 * tx + 4 * q]);
      if (4 * q + 0 < NW) win[4 * q + 0] = (double)v.x;
      if (4 * q + 1 < NW) win[4 * q + 1] = (double)v.y;
      if (4 * q + 2 < NW) win[4 * q + 2] = (double)v.z;
      if (4 * q + 3 < NW) win[4 * q + 3] = (double)v.w;
    }
    const int64_t z = row / P1, y = row % P1;
    const bool edge_zy = r > 0 && (z < r || y < r || z >= P0 - r || y >= P1 - r);
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double acc = mul_rn(win[WR + k], wk[0]);
#pragma unroll
      for (int j = WR; j >= 1; --j)
        acc = add_rn(acc, mul_rn(add_rn(win[WR + k - j], win[WR + k + j]), wk[j]));
      const int64_t x = x0 + k;
      o[k] = (edge_zy || (r > 0 && (x < r || x >= P2 - r))) ? 0.f : (float)acc;
    }
    float *dst = out + row * P2 + x0;
    if (x0 + 4 <= P2 && ((row * P2 + x0) & 3) == 0) {
      *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (x0 + k < P2) dst[k] = o[k];
    }
    if (hist0) {
      // runs of equal bins (zeros of the shell, flat regions) cost one atomic
      uint32_t kb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) kb[k] = float_key(o[k]) >> L0_SHIFT;
      const int nv = (int)(P2 - x0 < 4 ? P2 - x0 : 4);
      // a wave whose 256 values share one bin (shell zeros, smooth regions): one atomic
      const uint32_t first = __shfl(kb[0], 0);
      const bool same = nv == 4 && kb[0] == first && kb[1] == first && kb[2] == first &&
                        kb[3] == first;
      if (__ballot(same) == ~0ull) {
        if ((t & 63) == 0) atomicAdd(&lh[first], 256u);
        continue;
      }
      int run = 1;
#pragma unroll
      for (int k = 1; k <= 4; ++k) {
        if (k < nv && kb[k] == kb[k - 1]) { ++run; continue; }
        if (k <= nv) atomicAdd(&lh[kb[k - 1]], (unsigned)run);
        run = 1;
      }
    }
  }
  __syncthreads();
  if (hist0)
    for (int b = t; b < L0_BINS; b += 256)
      if (lh[b]) atomicAdd(&hist0[b], (unsigned long long)lh[b]);
}

// (rounds 2 - 3 ran the z pass as a register-window kernel, gauss_z_win: every thread re-read
// its 2 WR halo, 2.7 x the volume through L2; the ring form below replaced it in round 4 -
// bit-identical, 0.546 -> 0.522 ms at 582^3 - and the window kernel was removed in round 5)
constexpr int GZ_OUT = 12;       // (the size limit of the 32-bit offsets below still counts in these rows)

// z pass, ring form (round 4): a thread walks a SEGMENT of its (y, x) column, 2 WR outputs at a
// time, with the 4 WR inputs they need in registers as two halves; after a step the upper half is
// the next step's lower half and the other is refilled from 2 WR values loaded - as floats, one
// step ahead - while the step was computed.  Every input is read once, a thread's loads are in flight under
// its own arithmetic instead of all in front of it, and the arithmetic per output is the same
// sequence (bit-identical results).  The column is cut into `nseg` segments so that the grid has
// enough waves; a segment's first window is its only re-read.
template <int WR>
__global__ __launch_bounds__(512) void gauss_z_ring(
    PadView pv, float *__restrict__ out, int P0, int P1, int P2,
    const double *__restrict__ w, int nxb, int nseg, int seg_len) {
  constexpr int H = 2 * WR;                        // outputs per step = half a window
  const unsigned wid = blockIdx.x;
  const int seg = (int)(wid % (unsigned)nseg);
  const unsigned rest = wid / (unsigned)nseg;
  const int xb = (int)(rest % (unsigned)nxb), y = (int)(rest / (unsigned)nxb);
  const int x = xb * (int)blockDim.x + (int)threadIdx.x;
  if (x >= P2) return;
  const int a_lo = seg * seg_len, a_hi = min(P0, a_lo + seg_len);     // outputs [a_lo, a_hi)
  if (a_lo >= a_hi) return;
  const int r = pv.r, D0 = (int)pv.D0, D1 = (int)pv.D1, D2 = (int)pv.D2;
  const int yy = y - r, xx = x - r;
  const bool col_ok = yy >= 0 && xx >= 0 && yy < D1 && xx < D2;
  const int pplane = P1 * P2;
  float *dst = out + ((int64_t)a_lo * P1 + y) * P2 + x;
  if (!col_ok) {                                   // a column of the zero padding
    for (int a = a_lo; a < a_hi; ++a, dst += pplane) *dst = 0.f;
    return;
  }
  const int plane = D1 * D2;
  const float *col = pv.pred + (int64_t)yy * D2 + xx;
  // H consecutive padded inputs starting at padded z index `a` (reflected at the padded
  // volume's ends, zero outside the prediction)
  auto load_half = [&](int a, float (&f)[H]) {
    if (a - r >= 0 && a + H - r <= D0) {
      const float *q = col + (int64_t)(a - r) * plane;
#pragma unroll
      for (int i = 0; i < H; ++i) f[i] = q[(int64_t)i * plane];
    } else {
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const int zz = reflect1(a + i, P0) - r;
        f[i] = (zz >= 0 && zz < D0) ? col[(int64_t)zz * plane] : 0.f;
      }
    }
  };
  double wk[WR + 1];
#pragma unroll
  for (int j = 0; j <= WR; ++j) wk[j] = w[j];
  double h0[H], h1[H];
  float nf[H];
  load_half(a_lo - WR, nf);
#pragma unroll
  for (int i = 0; i < H; ++i) h0[i] = (double)nf[i];
  load_half(a_lo - WR + H, nf);
#pragma unroll
  for (int i = 0; i < H; ++i) h1[i] = (double)nf[i];
  // one step: outputs a0 .. a0 + H - 1 from the window [lo | hi] (input a0 - WR + k at k);
  // the next step's new half (inputs a0 + 3 WR ...) is loaded first and converted into `lo` last
  auto step = [&](int a0, double (&lo)[H], double (&hi)[H]) {
    const bool more = a0 + H < a_hi;
    if (more) load_half(a0 + H + WR, nf);
#pragma unroll
    for (int o = 0; o < H; ++o) {
      auto at = [&](int k) -> double { return k < H ? lo[k] : hi[k - H]; };
      double acc = mul_rn(at(WR + o), wk[0]);
#pragma unroll
      for (int j = WR; j >= 1; --j)
        acc = add_rn(acc, mul_rn(add_rn(at(WR + o - j), at(WR + o + j)), wk[j]));
      if (a0 + o < a_hi) dst[(int64_t)o * pplane] = (float)acc;
    }
    dst += (int64_t)H * pplane;
    if (more) {
#pragma unroll
      for (int i = 0; i < H; ++i) lo[i] = (double)nf[i];
    }
  };
  for (int a0 = a_lo; a0 < a_hi; a0 += 2 * H) {
    step(a0, h0, h1);
    if (a0 + H < a_hi) step(a0 + H, h1, h0);
  }
}

// y and x passes fused through one LDS tile: a workgroup takes GYX_TY consecutive y
// rows of one z plane over the WHOLE x extent.  Phase 1 (lanes along x, coalesced):
// a thread slides the register window of gauss_pass_win<1> down its column and stores
// the GYX_TY fp32-rounded y-pass outputs into the tile - with the row's reflected x
// halo, which for a whole row is made of the row's own values.  Phase 2: the x pass of
// gauss_x_lds on the tile rows (aligned 16-B LDS reads), margin zeroing, the 16-B
// stores and the first radix level.  The y-pass volume never exists in HBM (one
// 4 B/voxel write and one read less); arithmetic and rounding points are unchanged.
// The 1-D grid is decoded so that the blocks an XCD receives (b, b + 8, ...) walk the
// y tiles of consecutive planes: y neighbours share their 2*WR halo rows in that L2.
// 12 rows: the tile is 33 KiB at P2 = 636 and four workgroups fit a CU (16 rows, three
// workgroups: 1.07 ms against 0.94; 8 rows: 0.97).
constexpr int GYX_TY = 12;

template <int WR>
__global__ __launch_bounds__(512) void gauss_yx_fused(
    const float *__restrict__ in, float *__restrict__ out, int P0, int P1, int P2,
    const double *__restrict__ w, int r, int lrow,
    unsigned long long *__restrict__ hist0, unsigned long long *__restrict__ cellmax,
    int C1, int C2, float floor_v) {
  constexpr int TY = GYX_TY;
  constexpr int NWY = TY + 2 * WR;
  constexpr int XO = 8;                        // x outputs per thread and iteration
  constexpr int NWX = XO + 2 * WR;
  extern __shared__ __attribute__((aligned(16))) float yx_tile[];   // [TY][lrow], then
  // the keys of the workgroup's (TY / CELL) x C2 cells (8 B each)
  unsigned long long *ck = reinterpret_cast<unsigned long long *>(yx_tile + TY * lrow);
  __shared__ unsigned int lh[L0_BINS];
  const int t = threadIdx.x, bd = blockDim.x;
  for (int b = t; b < L0_BINS; b += bd) lh[b] = 0u;
  for (int b = t; b < (TY / CELL) * C2; b += bd) ck[b] = 0ull;
  double wk[WR + 1];
#pragma unroll
  for (int j = 0; j <= WR; ++j) wk[j] = w[j];
  const int nty = (P1 + TY - 1) / TY;
  const int nwork = ((P0 + CELL - 1) / CELL) * nty;
  // XCD-contiguous work order (speed only): block b -> slot b / 8 of XCD b % 8
  const int per_xcd = (nwork + 7) / 8;
  const int wid = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (wid >= nwork || (int)(blockIdx.x >> 3) >= per_xcd) return;   // whole workgroup
  const int zg = wid / nty, y0 = (wid - zg * nty) * TY;
  // byte offsets of the window's rows inside a plane (uniform; reflected at the ends):
  // soffset operands of the buffer loads - one shared 32-bit voffset per thread, no
  // per-row 64-bit addresses
  int roff[NWY];
#pragma unroll
  for (int i = 0; i < NWY; ++i) roff[i] = reflect1(y0 - WR + i, P1) * P2 * 4;
  // a workgroup walks the CELL planes of its z group: the cell keys stay in LDS.
  // (Running the window loads one column / plane ahead of the arithmetic was measured:
  // 147 VGPRs, two workgroups per CU, 1.9 ms against 1.2 ms - not kept.)
  for (int zi = 0; zi < CELL; ++zi) {
    const int z = zg * CELL + zi;
    if (z >= P0) break;
    const __amdgpu_buffer_rsrc_t plane = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(in + (int64_t)z * P1 * P2), 0, P1 * P2 * 4, 0x00020000);
    __syncthreads();              // the tile of the previous plane is consumed
    // ---- phase 1: y pass, one column per thread and iteration
    for (int x = t; x < P2; x += bd) {
      // all NWY loads in flight before the first conversion (left to itself the
      // scheduler serialises load -> wait -> convert to save registers: 1.6 ms)
      int raw[NWY];
#pragma unroll
      for (int i = 0; i < NWY; ++i)
        raw[i] = __builtin_amdgcn_raw_buffer_load_b32(plane, x * 4, roff[i], 0);
      __builtin_amdgcn_sched_barrier(0);
      double win[NWY];
#pragma unroll
      for (int i = 0; i < NWY; ++i) win[i] = (double)__builtin_bit_cast(float, raw[i]);
      float *colp = yx_tile + x + WR;
#pragma unroll
      for (int o = 0; o < TY; ++o) {
        double acc = mul_rn(win[WR + o], wk[0]);
#pragma unroll
        for (int j = WR; j >= 1; --j)
          acc = add_rn(acc, mul_rn(add_rn(win[WR + o - j], win[WR + o + j]), wk[j]));
        colp[o * lrow] = (float)acc;
      }
    }
    __syncthreads();
    // the rows' reflected x halos ('reflect': -1 -> 0, P2 -> P2 - 1) are the rows' own
    // values: 2 * WR copies per row
    for (int u = t; u < TY * 2 * WR; u += bd) {
      const int row = u / (2 * WR), k = u - row * (2 * WR);
      float *rowp = yx_tile + row * lrow;
      if (k < WR) rowp[WR - 1 - k] = rowp[WR + k];
      else rowp[WR + P2 + (k - WR)] = rowp[WR + P2 - 1 - (k - WR)];
    }
    __syncthreads();
    // ---- phase 2: x pass on the tile rows, XO outputs per thread and iteration; the
    // (row, piece) of unit t + bd * k advances without a division
    const int nq = (P2 + XO - 1) / XO;
    const int step_ty = bd / nq, step_xq = bd - step_ty * nq;
    int ty = t / nq, xq = t - ty * nq;
    for (; ty < TY; ty += step_ty, xq += step_xq) {
      if (xq >= nq) { xq -= nq; ++ty; if (ty >= TY) break; }
      const int y = y0 + ty;
      if (y >= P1) break;
      const int x0 = xq * XO;
      const float *rowp = yx_tile + ty * lrow + x0;
      double win[NWX];
#pragma unroll
      for (int q = 0; q < (NWX + 3) / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(rowp + 4 * q);
        if (4 * q + 0 < NWX) win[4 * q + 0] = (double)v.x;
        if (4 * q + 1 < NWX) win[4 * q + 1] = (double)v.y;
        if (4 * q + 2 < NWX) win[4 * q + 2] = (double)v.z;
        if (4 * q + 3 < NWX) win[4 * q + 3] = (double)v.w;
      }
      const bool edge_zy = r > 0 && (z < r || y < r || z >= P0 - r || y >= P1 - r);
      float o[XO];
#pragma unroll
      for (int k = 0; k < XO; ++k) {
        double acc = mul_rn(win[WR + k], wk[0]);
#pragma unroll
        for (int j = WR; j >= 1; --j)
          acc = add_rn(acc, mul_rn(add_rn(win[WR + k - j], win[WR + k + j]), wk[j]));
        const int x = x0 + k;
        o[k] = (edge_zy || (r > 0 && (x < r || x >= P2 - r))) ? 0.f : (float)acc;
      }
      const int64_t flat = ((int64_t)z * P1 + y) * P2 + x0;
      float *dst = out + flat;
      const int nv = P2 - x0 < XO ? P2 - x0 : XO;
      if (nv == XO && (flat & 3) == 0) {
#pragma unroll
        for (int k = 0; k < XO; k += 4)
          *reinterpret_cast<float4 *>(dst + k) = make_float4(o[k], o[k + 1], o[k + 2], o[k + 3]);
      } else {
#pragma unroll
        for (int k = 0; k < XO; ++k)
          if (k < nv) dst[k] = o[k];
      }
      if (cellmax) {
        // key of the largest value above the floor (>= 0) of each CELL-run:
        // (bits << 32) | ~flat, so that equal values order by ascending flat index
        // (cell_best's key).  With a floor most waves skip the LDS atomics altogether.
        static_assert(XO % CELL == 0, "whole cells per piece");
#pragma unroll
        for (int c = 0; c < XO / CELL; ++c) {
          unsigned long long best = 0ull;
#pragma unroll
          for (int d = 0; d < CELL; ++d) {
            const int k = c * CELL + d;
            if (k < nv && o[k] > floor_v) {
              const unsigned long long key =
                  ((unsigned long long)__float_as_uint(o[k]) << 32) |
                  (0xFFFFFFFFu - (uint32_t)(flat + k));
              best = key > best ? key : best;
            }
          }
          if (best) atomicMax(&ck[(ty / CELL) * C2 + x0 / CELL + c], best);
        }
      }
      if (hist0) {
        // runs of equal bins (the shell's zeros, smooth regions) cost one LDS atomic
        uint32_t kb[XO];
#pragma unroll
        for (int k = 0; k < XO; ++k) kb[k] = float_key(o[k]) >> L0_SHIFT;
        int run = 1;
#pragma unroll
        for (int k = 1; k <= XO; ++k) {
          if (k < nv && kb[k] == kb[k - 1]) { ++run; continue; }
          if (k <= nv) atomicAdd(&lh[kb[k - 1]], (unsigned)run);
          run = 1;
        }
      }
    }
  }
  __syncthreads();
  if (cellmax)
    for (int b = t; b < (TY / CELL) * C2; b += bd) {
      const int yg = b / C2, xc = b - yg * C2;
      const int cy = y0 / CELL + yg;
      if (cy < C1) cellmax[((int64_t)zg * C1 + cy) * C2 + xc] = ck[b];
    }
  if (hist0)
    for (int b = t; b < L0_BINS; b += bd)
      if (lh[b]) atomicAdd(&hist0[b], (unsigned long long)lh[b]);
}

// LDS floats per tile row of gauss_yx_fused: the row, its two halos, rounded up to a
// 16-B multiple plus 4 (the last thread's aligned window may read 3 floats past it)
static inline int gyx_lrow(int64_t P2, int wr) {
  return (int)((ceil_div64(P2, 8) * 8 + 2 * wr + 3) / 4 * 4 + 4);
}
// workgroup size (a multiple of 64 in [256, 512]) that wastes the fewest lanes on `n`
// columns / pieces per pass
static inline int best_block(int64_t n) {
  int best = 256;
  double waste = 1e9;
  for (int bd = 256; bd <= 512; bd += 64) {
    const double wst = (double)(ceil_div64(n, bd) * bd) / (double)n;
    if (wst < waste - 1e-9) { waste = wst; best = bd; }
  }
  return best;
}
static inline size_t gyx_lds_bytes(int64_t P2, int wr) {
  return (size_t)GYX_TY * gyx_lrow(P2, wr) * sizeof(float) +
         (size_t)(GYX_TY / CELL) * ceil_div64(P2, CELL) * sizeof(unsigned long long);
}
static inline bool gyx_fits(const int64_t P[3], int wr) {
  // single-bounce reflections, 32-bit in-plane offsets, tile + cell keys within 60 KiB
  return P[1] * P[2] * 4 < ((int64_t)1 << 31) && P[2] >= 2 * wr && P[1] >= GYX_TY + 2 * wr && P[0] * ceil_div64(P[1], GYX_TY) < (1 << 28) &&
         gyx_lds_bytes(P[2], wr) <= 60 * 1024;
}

// Compaction without global atomics: workgroup b scans elements [b*chunk, (b+1)*chunk)
// and packs those whose key matches (key & mask) == want to the front of the same
// range of `list` (as floats); counts[b] = how many.
__global__ __launch_bounds__(256) void compact_prefix(
    const float *__restrict__ v, int64_t n, int64_t chunk, uint32_t mask, uint32_t want,
    float *__restrict__ list, unsigned int *__restrict__ counts) {
  __shared__ unsigned int cnt;
  if (threadIdx.x == 0) cnt = 0u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t lo = (int64_t)blockIdx.x * chunk;       // chunk: a multiple of 256 floats
  const int64_t hi = lo + chunk < n ? lo + chunk : n;
  float *dst = list + lo;
  // 16-B loads (v is 256-B aligned and lo a multiple of 256 floats); the order of the
  // hits inside the chunk does not matter: the list is only histogrammed
  for (int64_t i0 = lo; i0 < hi; i0 += 1024) {
    const int64_t i = i0 + 4 * (int64_t)threadIdx.x;
    float f[4] = {0.f, 0.f, 0.f, 0.f};
    if (i + 4 <= hi) {
      const float4 q = *reinterpret_cast<const float4 *>(v + i);
      f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (i + c < hi) f[c] = v[i + c];
    }
    bool hit[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) hit[c] = i + c < hi && (float_key(f[c]) & mask) == want;
    if (__ballot(hit[0] | hit[1] | hit[2] | hit[3]) == 0ull) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned long long m = __ballot(hit[c]);
      if (m == 0ull) continue;
      unsigned int base = 0;
      if (lane == 0) base = atomicAdd(&cnt, (unsigned)__popcll(m));
      base = __shfl(base, 0);
      // at most as many hits as elements scanned so far: stays inside the chunk
      if (hit[c]) dst[base + __popcll(m & ((1ull << lane) - 1ull))] = f[c];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = cnt;
}

// key_histogram over the compacted chunks
__global__ __launch_bounds__(256) void key_histogram_chunks(
    const float *__restrict__ list, const unsigned int *__restrict__ counts, int64_t chunk,
    uint32_t mask, uint32_t want, int shift, int bits, unsigned long long *__restrict__ hist) {
  __shared__ unsigned int lh[2048];
  const int nb = 1 << bits;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) lh[b] = 0;
  __syncthreads();
  const float *src = list + (int64_t)blockIdx.x * chunk;
  const unsigned int m = counts[blockIdx.x];
  for (unsigned int i = threadIdx.x; i < m; i += blockDim.x) {
    const uint32_t k = float_key(src[i]);
    if ((k & mask) == want) atomicAdd(&lh[(k >> shift) & (nb - 1)], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += blockDim.x)
    if (lh[b]) atomicAdd(&hist[b], (unsigned long long)lh[b]);
}

template <int WR>
int launch_gauss_win(fpl_ctx *ctx, PadView pv, float *a, float *b, const int64_t P[3],
                     const double *w_dev, int r, unsigned long long *hist0,
                     unsigned long long *cellmax, float floor_v, bool *cellmax_done) {
  *cellmax_done = false;
  hipStream_t st = ctx->stream;
  constexpr int OUT_ZY = 16;                 // outputs per thread of gauss_pass_win<0/1>
  const bool fused = gyx_fits(P, WR) && !getenv("FPL_V2O_UNFUSED");
  // fused: z pass -> b, y+x -> a;  separate passes: z -> a, y -> b, x -> a
  // 32-bit in-column / in-plane offsets of gauss_z_ring
  const bool small = (GZ_OUT + 2 * WR) * pv.D1 * pv.D2 < ((int64_t)1 << 31) &&
                     P[1] * P[2] * GZ_OUT < ((int64_t)1 << 31) && P[0] >= 2 * WR;
  if (small && !getenv("FPL_V2O_UNFUSED")) {
    // ring form: segments of a multiple of 4 WR outputs (two steps), about a quarter of the column
    const int bd = best_block(P[2]);
    const int64_t nxb = ceil_div64(P[2], bd);
    const int64_t seg_len = std::max<int64_t>(4 * WR, ceil_div64(ceil_div64(P[0], 4), 4 * WR) * 4 * WR);
    const int64_t nseg = ceil_div64(P[0], seg_len);
    const int64_t nwork = nxb * nseg * P[1];
    FPL_REQUIRE(ctx, nwork < ((int64_t)1 << 31), "voxel2obj: volume too large");
    TimedLaunch tl(ctx, "v2o_gauss_z");
    gauss_z_ring<WR><<<(unsigned)nwork, bd, 0, st>>>(
        pv, fused ? b : a, (int)P[0], (int)P[1], (int)P[2], w_dev, (int)nxb, (int)nseg, (int)seg_len);
  } else {
    const int64_t n = ceil_div64(P[0], OUT_ZY) * P[1] * P[2];
    TimedLaunch tl(ctx, "v2o_gauss_z");
    gauss_pass_win<0, WR><<<(unsigned)ceil_div64(n, 256), 256, 0, st>>>(
        pv, nullptr, fused ? b : a, P[0], P[1], P[2], w_dev, r);
  }
  if (fused) {
    const int lrow = gyx_lrow(P[2], WR);
    const size_t lds = gyx_lds_bytes(P[2], WR);
    static bool attr_set[FPL_MAX_DEVICES] = {false};
    if (!attr_set[ctx->device % FPL_MAX_DEVICES]) {
      FPL_HIP(ctx, hipFuncSetAttribute((const void *)gauss_yx_fused<WR>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024));
      attr_set[ctx->device % FPL_MAX_DEVICES] = true;
    }
    const int64_t nwork = ceil_div64(P[0], CELL) * ceil_div64(P[1], GYX_TY);
    const int64_t nblk = ceil_div64(nwork, 8) * 8;
    // phase 1 walks P2 columns, phase 2 16 * ceil(P2 / 8) pieces: size the workgroup
    // for the columns (the pieces are ~2x as many and split evenly enough)
    const int bd = best_block(P[2]);
    TimedLaunch tl(ctx, "v2o_gauss_yx");
    gauss_yx_fused<WR><<<(unsigned)nblk, bd, lds, st>>>(
        b, a, (int)P[0], (int)P[1], (int)P[2], w_dev, r, lrow, hist0, cellmax,
        (int)ceil_div64(P[1], CELL), (int)ceil_div64(P[2], CELL), floor_v);
    *cellmax_done = cellmax != nullptr;
    return 0;
  }
  {
    const int64_t n = P[0] * ceil_div64(P[1], OUT_ZY) * P[2];
    TimedLaunch tl(ctx, "v2o_gauss_y");
    gauss_pass_win<1, WR><<<(unsigned)ceil_div64(n, 256), 256, 0, st>>>(pv, a, b, P[0], P[1], P[2], w_dev, r);
  }
  {
    const int64_t ntiles = ceil_div64(P[0] * P[1], GX_ROWS) * ceil_div64(P[2], GX_SEG);
    const int64_t nblk = std::min<int64_t>(ntiles, (int64_t)ctx->n_cu * 8);
    TimedLaunch tl(ctx, "v2o_gauss_x");
    gauss_x_lds<WR><<<(unsigned)nblk, 256, 0, st>>>(b, a, P[0], P[1], P[2], w_dev, r, hist0);
  }
  return 0;
}

// histogram of `bits` key bits at `shift` over the elements whose key matches
// (key & mask) == want
__global__ __launch_bounds__(256) void key_histogram(
    const float *__restrict__ v, int64_t n, uint32_t mask, uint32_t want,
    int shift, int bits, unsigned long long *__restrict__ hist) {
  __shared__ unsigned int lh[2048];
  const int nb = 1 << bits;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) lh[b] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride) {
    const uint32_t k = float_key(v[i]);
    if ((k & mask) == want) atomicAdd(&lh[(k >> shift) & (nb - 1)], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += blockDim.x)
    if (lh[b]) atomicAdd(&hist[b], (unsigned long long)lh[b]);
}

// The NMS works on the smoothed volume IN PLACE: a voxel is a live candidate while
// (double)v > thresh && v > 0, and clear_balls suppresses by storing 0 (a separate
// "live key" volume - one more 4 B/voxel write and read - is not needed; the smoothed
// volume is consumed by the NMS).
//
// best live key per 4x4x4 cell: (value bits << 32) | ~flat_index(padded volume).
// After the first round only the cells a cleared ball touched (marked CELL_DIRTY by
// clear_balls) are re-scanned; the others keep their key.
constexpr unsigned long long CELL_DIRTY = 1ull;    // no live key has value bits 0

// device counters of the NMS (unsigned long long each)
enum { CNT_LIVE = 0,      // cells with a live key
       CNT_ROUND = 1,     // winners of the current round
       CNT_DIRTY = 2,     // cells on the current round's re-scan list
       CNT_TOTAL = 3,     // winners so far
       CNT_ROUNDS = 4,    // rounds that had live cells
       CNT_N = 8 };

// best live key of cell c: its largest voxel that passes the threshold (0: none)
template <bool ALIGNED>
__device__ __forceinline__ unsigned long long scan_cell(const float *__restrict__ s, double thresh,
                                                        int64_t P0, int64_t P1, int64_t P2,
                                                        int64_t C1, int64_t C2, int64_t c) {
  unsigned long long b = 0;
  const int64_t cx = c % C2, cy = (c / C2) % C1, cz = c / (C2 * C1);
#pragma unroll
  for (int dz = 0; dz < CELL; ++dz)
#pragma unroll
    for (int dy = 0; dy < CELL; ++dy) {
      const int64_t z = cz * CELL + dz, y = cy * CELL + dy;
      if (z >= P0 || y >= P1) continue;
      const float *rowp = s + (z * P1 + y) * P2 + cx * CELL;
      float v[4];
      if (ALIGNED) {               // P2 % 4 == 0: the cell row is one aligned 16-B piece
        const float4 q = *reinterpret_cast<const float4 *>(rowp);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int dx = 0; dx < CELL; ++dx) v[dx] = cx * CELL + dx < P2 ? rowp[dx] : 0.f;
      }
#pragma unroll
      for (int dx = 0; dx < CELL; ++dx)
        if ((double)v[dx] > thresh && v[dx] > 0.f) {
          const uint32_t flat = (uint32_t)((z * P1 + y) * P2 + cx * CELL + dx);
          const unsigned long long k =
              ((unsigned long long)__float_as_uint(v[dx]) << 32) | (0xFFFFFFFFu - flat);
          b = k > b ? k : b;
        }
    }
  return b;
}

// adds the workgroup's count of live cells to counters[CNT_LIVE] with ONE atomic (one per
// wave on a single address cost 0.15 ms a round)
__device__ __forceinline__ void add_live(unsigned mine, unsigned long long *counters) {
  __shared__ unsigned wcount[8];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
  if ((threadIdx.x & 63) == 0) wcount[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned tot = 0;
    for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) tot += wcount[w];
    if (tot) atomicAdd(&counters[CNT_LIVE], (unsigned long long)tot);
  }
}

// first round without keys from the smoothing pass: every cell, grid-stride
template <bool ALIGNED>
__global__ __launch_bounds__(256) void cell_best(const float *__restrict__ s, double thresh,
                          int64_t P0, int64_t P1, int64_t P2, int64_t C0,
                          int64_t C1, int64_t C2,
                          unsigned long long *__restrict__ best,
                          unsigned long long *__restrict__ counters) {
  const int64_t n_cells = C0 * C1 * C2;
  unsigned mine = 0;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cells;
       c += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long b = scan_cell<ALIGNED>(s, thresh, P0, P1, P2, C1, C2, c);
    best[c] = b;
    mine += b != 0;
  }
  add_live(mine, counters);
}

// later rounds: only the cells a cleared ball cut through (clear_balls lists them and
// takes them out of the live count; the ones that still hold a live voxel come back)
template <bool ALIGNED>
__global__ __launch_bounds__(256) void cell_rescan(const float *__restrict__ s, double thresh,
                          int64_t P0, int64_t P1, int64_t P2, int64_t C1, int64_t C2,
                          unsigned long long *__restrict__ best,
                          const unsigned int *__restrict__ dirty_list,
                          unsigned long long *__restrict__ counters) {
  const int64_t n = (int64_t)counters[CNT_DIRTY];
  unsigned mine = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = dirty_list[i];
    const unsigned long long b = scan_cell<ALIGNED>(s, thresh, P0, P1, P2, C1, C2, c);
    best[c] = b;
    mine += b != 0;
  }
  add_live(mine, counters);
}

// round 0 from the keys the fused y+x pass took (per cell: its largest positive voxel):
// a cell is live iff that voxel passes the threshold
__global__ __launch_bounds__(256) void cell_threshold(unsigned long long *__restrict__ best,
                                                      int64_t n_cells, double thresh,
                                                      unsigned long long *__restrict__ counters) {
  unsigned mine = 0;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cells;
       c += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long b = best[c];
    if (b && !((double)__uint_as_float((uint32_t)(b >> 32)) > thresh)) {
      b = 0;
      best[c] = 0;
    }
    mine += b != 0;
  }
  add_live(mine, counters);
}

template <int AXIS>
__global__ void window_max(const unsigned long long *__restrict__ in,
                           unsigned long long *__restrict__ out, int64_t C0,
                           int64_t C1, int64_t C2, int hw,
                           const unsigned long long *__restrict__ counters) {
  if (counters[CNT_LIVE] == 0) return;  // no live cell left: the closing round is a no-op
  // XCD-contiguous block order: the blocks an XCD receives (b, b + 8, ...) cover one
  // slab of z, so the 2 hw + 1 rows / planes a cell reads are re-used in THAT XCD's L2
  // (round-robin order made every XCD read the whole array: 296 / 234 MB from HBM per
  // y / z pass over a 32 MB array)
  const unsigned per_xcd = gridDim.x >> 3;           // the grid is a multiple of 8
  const int64_t blk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int64_t c = blk * blockDim.x + threadIdx.x;
  if (c >= C0 * C1 * C2) return;
  const int64_t cx = c % C2, cy = (c / C2) % C1, cz = c / (C2 * C1);
  const int64_t pos = AXIS == 0 ? cz : (AXIS == 1 ? cy : cx);
  const int64_t n = AXIS == 0 ? C0 : (AXIS == 1 ? C1 : C2);
  const int64_t stride = AXIS == 0 ? C1 * C2 : (AXIS == 1 ? C2 : 1);
  const int64_t lo = pos - hw < 0 ? 0 : pos - hw;
  const int64_t hi = pos + hw >= n ? n - 1 : pos + hw;
  unsigned long long m = 0;
  for (int64_t p = lo; p <= hi; ++p) {
    const unsigned long long v = in[c + (p - pos) * stride];
    m = v > m ? v : m;
  }
  out[c] = m;
}

// The same window maxima with fewer loads.  x pass: a workgroup stages 256 consecutive
// cells and hw neighbours on either side in LDS and every thread reads its window there
// (the plain kernel loads 2 hw + 1 values per cell from L1 / L2).  y / z passes: a thread
// owns WM_OUTS consecutive cells along the pass axis at one x (lanes along x: coalesced),
// loads the WM_OUTS + 2 HW values once and derives the WM_OUTS window maxima from a shared
// core plus running suffix / prefix maxima (~4 comparisons per output instead of 2 HW).
// Work items are ordered so that the blocks of an XCD cover a slab ACROSS the pass axis:
// no halo is shared between XCDs.
constexpr int WM_OUTS = 8;

__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) {
  return a > b ? a : b;
}

__global__ __launch_bounds__(256) void window_max_x(
    const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out,
    int64_t n_cells, int C2, int hw, const unsigned long long *__restrict__ counters) {
  if (counters[CNT_LIVE] == 0) return;
  __shared__ unsigned long long sh[256 + 2 * 32];
  const unsigned per_xcd = gridDim.x >> 3;
  const int64_t blk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int64_t c0 = blk * 256;
  if (c0 >= n_cells) return;                       // whole workgroup
  const int t = threadIdx.x;
  for (int i = t; i < 256 + 2 * hw; i += 256) {
    const int64_t c = c0 - hw + i;
    sh[i] = (c >= 0 && c < n_cells) ? in[c] : 0ull;
  }
  __syncthreads();
  const int64_t c = c0 + t;
  if (c >= n_cells) return;
  const int cx = (int)(c % C2);
  const int lo = cx - hw < 0 ? -cx : -hw, hi = cx + hw >= C2 ? C2 - 1 - cx : hw;
  unsigned long long m = 0;
  for (int k = lo; k <= hi; ++k) m = umax64(m, sh[t + hw + k]);
  out[c] = m;
}

// AXIS 1: windows along y (stride C2), AXIS 0: along z (stride C1 * C2)
template <int AXIS, int HW>
__global__ __launch_bounds__(256) void window_max_seg(
    const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out,
    int C0, int C1, int C2, const unsigned long long *__restrict__ counters) {
  if (counters[CNT_LIVE] == 0) return;
  constexpr int W = 2 * HW + 1, NV = WM_OUTS + 2 * HW;
  const int n = AXIS == 0 ? C0 : C1;               // length of the pass axis
  const int other = AXIS == 0 ? C1 : C0;           // the axis the XCD slabs cut
  const int nseg = (n + WM_OUTS - 1) / WM_OUTS;
  const int64_t n_items = (int64_t)other * nseg * C2;
  const unsigned per_xcd = gridDim.x >> 3;
  const int64_t blk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int64_t item = blk * 256 + threadIdx.x;
  if (item >= n_items) return;
  const int cx = (int)(item % C2);
  const int64_t rest = item / C2;
  const int sg = (int)(rest % nseg), o = (int)(rest / nseg);
  const int64_t stride = AXIS == 0 ? (int64_t)C1 * C2 : C2;
  const int64_t base = AXIS == 0 ? (int64_t)o * C2 + cx : (int64_t)o * C1 * C2 + cx;
  const int p0 = sg * WM_OUTS;                     // first output position
  unsigned long long v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = p0 - HW + i;
    const int pc = p < 0 ? 0 : (p >= n ? n - 1 : p);
    const unsigned long long x = in[base + pc * stride];
    v[i] = (p >= 0 && p < n) ? x : 0ull;
  }
  unsigned long long res[WM_OUTS];
  if (W >= WM_OUTS) {
    // every window [i, i + W - 1], i < WM_OUTS, holds [WM_OUTS - 1, W - 1]
    unsigned long long core = v[WM_OUTS - 1];
#pragma unroll
    for (int i = WM_OUTS; i < W; ++i) core = umax64(core, v[i]);
    unsigned long long run = 0;
    unsigned long long left[WM_OUTS];
    left[WM_OUTS - 1] = 0;
#pragma unroll
    for (int i = WM_OUTS - 2; i >= 0; --i) { run = umax64(run, v[i]); left[i] = run; }
    run = 0;
    res[0] = umax64(core, left[0]);
#pragma unroll
    for (int i = 1; i < WM_OUTS; ++i) {
      run = umax64(run, v[W - 1 + i]);
      res[i] = umax64(umax64(core, left[i]), run);
    }
  } else {
#pragma unroll
    for (int i = 0; i < WM_OUTS; ++i) {
      unsigned long long m = v[i];
#pragma unroll
      for (int k = 1; k < W; ++k) m = umax64(m, v[i + k]);
      res[i] = m;
    }
  }
#pragma unroll
  for (int i = 0; i < WM_OUTS; ++i)
    if (p0 + i < n) out[base + (int64_t)(p0 + i) * stride] = res[i];
}

template <int HW>
static void launch_window_max_seg(hipStream_t st, const unsigned long long *best,
                                  unsigned long long *wa, unsigned long long *wb, int C0, int C1,
                                  int C2, int64_t n_cells, const unsigned long long *counters) {
  const unsigned gx = (unsigned)((ceil_div64(n_cells, 256) + 7) / 8 * 8);
  window_max_x<<<gx, 256, 0, st>>>(best, wa, n_cells, C2, HW, counters);
  const int64_t ny = (int64_t)C0 * ceil_div64(C1, WM_OUTS) * C2;
  const unsigned gy = (unsigned)((ceil_div64(ny, 256) + 7) / 8 * 8);
  window_max_seg<1, HW><<<gy, 256, 0, st>>>(wa, wb, C0, C1, C2, counters);
  const int64_t nz = (int64_t)C1 * ceil_div64(C0, WM_OUTS) * C2;
  const unsigned gz = (unsigned)((ceil_div64(nz, 256) + 7) / 8 * 8);
  window_max_seg<0, HW><<<gz, 256, 0, st>>>(wb, wa, C0, C1, C2, counters);
}

__global__ void pick_winners(const unsigned long long *__restrict__ best,
                             const unsigned long long *__restrict__ wmax,
                             int64_t n_cells,
                             unsigned long long *__restrict__ counters,
                             unsigned long long *__restrict__ round_list,
                             unsigned long long *__restrict__ all_list,
                             int64_t cap) {
  if (counters[CNT_LIVE] == 0) return;        // wmax is stale then
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cells) return;
  if (c == 0) atomicAdd(&counters[CNT_ROUNDS], 1ull);
  const unsigned long long b = best[c];
  if (b == 0 || b != wmax[c]) return;
  const unsigned long long slot = atomicAdd(&counters[CNT_ROUND], 1ull);
  round_list[slot] = b;             // one winner per cell at most: fits n_cells
  const unsigned long long g = atomicAdd(&counters[CNT_TOTAL], 1ull);
  if ((int64_t)g < cap) all_list[g] = b;
}

// A cleared ball takes voxels from cell c (`dead`: all of them).  Several balls of a round
// may touch one cell, so the hand-over goes through an exchange: whoever swaps out a LIVE
// key takes the cell out of the live count and - unless the cell is dead - puts it on the
// round's re-scan list (once).  Appends are staged in LDS (`stage`, `n_stage`), one global
// atomic per ball.
constexpr int CLR_STAGE = 6144;               // >= the 17^3 cells of a bounding box at r = 31;
                                              // beyond it appends go straight to the list
__device__ __forceinline__ void touch_cell(unsigned long long *__restrict__ best, int64_t c,
                                           bool dead, unsigned int *stage, unsigned int *n_stage,
                                           unsigned int *n_gone,
                                           unsigned int *__restrict__ dirty_list,
                                           unsigned long long *__restrict__ counters) {
  const unsigned long long old = atomicExch(&best[c], dead ? 0ull : CELL_DIRTY);
  if (old == 0ull) {
    if (!dead) best[c] = 0ull;                // it held no live voxel: nothing to re-scan
    return;
  }
  if (old == CELL_DIRTY) return;              // another ball of this round listed it
  atomicAdd(n_gone, 1u);
  if (dead) return;
  const unsigned int slot = atomicAdd(n_stage, 1u);
  if (slot < (unsigned)CLR_STAGE) stage[slot] = (unsigned int)c;
  else dirty_list[atomicAdd(&counters[CNT_DIRTY], 1ull)] = (unsigned int)c;
}
// the staged cells of one ball -> the global list; live count down by `n_gone`
__device__ __forceinline__ void flush_stage(const unsigned int *stage, unsigned int *n_stage,
                                            unsigned int *n_gone, unsigned int *base,
                                            unsigned int *__restrict__ dirty_list,
                                            unsigned long long *__restrict__ counters) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (*n_stage > (unsigned)CLR_STAGE) *n_stage = (unsigned)CLR_STAGE;
    *base = *n_stage ? (unsigned int)atomicAdd(&counters[CNT_DIRTY], (unsigned long long)*n_stage) : 0u;
    if (*n_gone) atomicAdd(&counters[CNT_LIVE], (unsigned long long)(0ull - *n_gone));
  }
  __syncthreads();
  for (unsigned int i = threadIdx.x; i < *n_stage; i += blockDim.x) dirty_list[*base + i] = stage[i];
  __syncthreads();
  if (threadIdx.x == 0) { *n_stage = 0u; *n_gone = 0u; }
  __syncthreads();
}

// CLR_PARTS workgroups per winner (a round's winners are few and a ball is 82 519 voxels
// and ~2 500 cells at r = 27: one workgroup per ball left most of the chip idle for the
// ~90 us a ball takes): clear the r-ball in the live volume
constexpr int CLR_PARTS = 4;
constexpr int CLR_ROWS = 1024;                // rows of one part of the cube, r <= 31
__global__ __launch_bounds__(256) void clear_balls(
    const unsigned long long *__restrict__ round_list,
    unsigned long long *__restrict__ counters, float *__restrict__ live,
    int64_t P1, int64_t P2, int r,
    unsigned long long *__restrict__ best, int64_t C1, int64_t C2,
    unsigned int *__restrict__ dirty_list) {
  __shared__ unsigned int stage[CLR_STAGE];
  __shared__ unsigned int n_stage, n_gone, stage_base;
  __shared__ int64_t row_off[CLR_ROWS];
  __shared__ short row_hx[CLR_ROWS];
  __shared__ int tab_part;                     // which part's rows the table holds
  if (threadIdx.x == 0) { n_stage = 0u; n_gone = 0u; tab_part = -1; }
  __syncthreads();
  const unsigned long long nwin = counters[CNT_ROUND];
  for (unsigned long long w = blockIdx.x; w < nwin * CLR_PARTS; w += gridDim.x) {
    const unsigned long long wi = w / CLR_PARTS;
    const int part = (int)(w % CLR_PARTS);
    const uint32_t flat = 0xFFFFFFFFu - (uint32_t)(round_list[wi] & 0xFFFFFFFFu);
    const int64_t x = flat % P2, y = (flat / P2) % P1, z = flat / (P2 * P1);
    const int side = 2 * r + 1;
    const int rows = side * side;
    const int row_lo = (int)((int64_t)rows * part / CLR_PARTS),
              row_hi = (int)((int64_t)rows * (part + 1) / CLR_PARTS);
    // eight lanes per (dz, dy) row of the cube, 32 rows per workgroup iteration.  The row
    // geometry (offset from the centre, half width) is the same for every ball: a table
    // per workgroup, one entry per thread instead of once per lane and row (one wave per
    // row - contiguous 220-B stores - was 2.5x slower: the index arithmetic, not the
    // store shape, is what a ball costs)
    if (row_hi - row_lo <= CLR_ROWS) {
      if (tab_part != part) {
        __syncthreads();
        for (int i = threadIdx.x; i < row_hi - row_lo; i += blockDim.x) {
          const int row = row_lo + i;
          const int dz = row / side - r, dy = row % side - r;
          const int rem = r * r - dz * dz - dy * dy;
          int hx = -1;
          if (rem >= 0) {
            hx = (int)sqrtf((float)rem);
            while ((hx + 1) * (hx + 1) <= rem) ++hx;
            while (hx * hx > rem) --hx;
          }
          row_hx[i] = (short)hx;
          row_off[i] = ((int64_t)dz * P1 + dy) * P2;
        }
        __syncthreads();
        if (threadIdx.x == 0) tab_part = part;
        __syncthreads();
      }
      float *centre = live + (z * P1 + y) * P2 + x;
      for (int i = (int)threadIdx.x / 8; i < row_hi - row_lo; i += blockDim.x / 8) {
        const int hx = row_hx[i];
        float *rowp = centre + row_off[i];
        for (int dx = -hx + (int)(threadIdx.x % 8); dx <= hx; dx += 8) rowp[dx] = 0.f;
      }
    } else {
      for (int row = row_lo + (int)threadIdx.x / 8; row < row_hi; row += blockDim.x / 8) {
        const int dz = row / side - r, dy = row % side - r;
        const int rem = r * r - dz * dz - dy * dy;
        if (rem < 0) continue;
        int hx = (int)sqrtf((float)rem);
        while ((hx + 1) * (hx + 1) <= rem) ++hx;
        while (hx * hx > rem) --hx;
        float *rowp = live + ((z + dz) * P1 + (y + dy)) * P2 + x;
        for (int dx = -hx + (int)(threadIdx.x % 8); dx <= hx; dx += 8) rowp[dx] = 0.f;
      }
    }
    // cached keys of the cells in the ball's bounding box (the r shell of the padded
    // volume keeps every ball inside it): a cell wholly inside the ball has no live
    // voxel left (key 0), a cell the ball does not reach keeps its key, only the cells
    // the sphere cuts through are re-scanned next round.  (Every part clears its voxels
    // and hands over its cells in one kernel; the re-scan is the next kernel.)
    const int64_t cz0 = (z - r) / CELL, cy0 = (y - r) / CELL, cx0 = (x - r) / CELL;
    const int nz = (int)((z + r) / CELL - cz0 + 1), ny = (int)((y + r) / CELL - cy0 + 1),
              nx = (int)((x + r) / CELL - cx0 + 1);
    auto span2 = [](int64_t lo, int64_t centre, int &near2, int &far2) {
      // squared distance range from `centre` to the integer interval [lo, lo + CELL - 1]
      const int a = (int)(lo - centre), b = (int)(lo + CELL - 1 - centre);
      const int n = a > 0 ? a : (b < 0 ? -b : 0);
      const int f = -a > b ? -a : b;
      near2 = n * n; far2 = f * f;
    };
    const int ncell = nz * ny * nx;
    const int cell_lo = (int)((int64_t)ncell * part / CLR_PARTS),
              cell_hi = (int)((int64_t)ncell * (part + 1) / CLR_PARTS);
    for (int i = cell_lo + (int)threadIdx.x; i < cell_hi; i += blockDim.x) {
      const int64_t cz = cz0 + i / (ny * nx), cy = cy0 + (i / nx) % ny, cx = cx0 + i % nx;
      int nz2, fz2, ny2, fy2, nx2, fx2;
      span2(cz * CELL, z, nz2, fz2);
      span2(cy * CELL, y, ny2, fy2);
      span2(cx * CELL, x, nx2, fx2);
      if (nz2 + ny2 + nx2 > r * r) continue;                  // untouched
      touch_cell(best, (cz * C1 + cy) * C2 + cx, fz2 + fy2 + fx2 <= r * r, stage, &n_stage,
                 &n_gone, dirty_list, counters);
    }
    flush_stage(stage, &n_stage, &n_gone, &stage_base, dirty_list, counters);
  }
}

// ---- segmentation-aware suppression (reference fplobjdetect.py:161-224) ---------------
// zero-padded u64 copy of the (Z,Y,X) label volume
template <typename T>
__global__ void seg_pad(const T *__restrict__ src, int64_t D0, int64_t D1, int64_t D2, int r,
                        unsigned long long *__restrict__ dst, int64_t P1, int64_t P2,
                        int64_t n_pad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  const int64_t x = i % P2 - r, y = (i / P2) % P1 - r, z = i / (P2 * P1) - r;
  unsigned long long v = 0;
  if (z >= 0 && y >= 0 && x >= 0 && z < D0 && y < D1 && x < D2)
    v = (unsigned long long)src[(z * D1 + y) * D2 + x];
  dst[i] = v;
}

// voxel counts per label: open-addressing hash (slot key = label + 1, 0 = empty); a
// thread folds runs of equal labels among its 16 consecutive voxels into one insert
__device__ __forceinline__ uint64_t seg_hash(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
  return k ^ (k >> 33);
}

__global__ void seg_count(const unsigned long long *__restrict__ seg, int64_t n,
                          unsigned long long *__restrict__ keys,
                          unsigned int *__restrict__ counts, uint64_t mask,
                          unsigned int *__restrict__ overflow) {
  const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i0 >= n) return;
  const int64_t i1 = i0 + 16 < n ? i0 + 16 : n;
  unsigned long long cur = seg[i0];
  unsigned int run = 0;
  for (int64_t i = i0; i <= i1; ++i) {
    const bool end = i == i1;
    const unsigned long long v = end ? 0 : seg[i];
    if (!end && v == cur) { ++run; continue; }
    uint64_t h = seg_hash(cur) & mask;
    bool done = false;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
      const unsigned long long prev = atomicCAS(&keys[h], 0ull, cur + 1ull);
      if (prev == 0ull || prev == cur + 1ull) { atomicAdd(&counts[h], run); done = true; break; }
      h = (h + 1) & mask;
    }
    if (!done) atomicExch(overflow, 1u);
    cur = v; run = 1;
  }
}

// smoothed = 0 where the voxel's segment has fewer than sz_thd voxels
template <typename T>
__global__ void seg_zero_small(const unsigned long long *__restrict__ seg, int64_t n,
                               const unsigned long long *__restrict__ keys,
                               const unsigned int *__restrict__ counts, uint64_t mask,
                               long long sz_thd, T *__restrict__ smoothed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = seg[i];
  uint64_t h = seg_hash(k) & mask;
  while (keys[h] != k + 1ull) h = (h + 1) & mask;      // every label was inserted
  if ((long long)counts[h] < sz_thd) smoothed[i] = (T)0;
}

// one workgroup per winner: the part of its r-ball that lies in its own segment (mask
// of the (2r+1)^3 cube, grown `dilate` times by the 6-neighbour cross, zero beyond the
// cube - scipy.ndimage.binary_dilation(iterations=dilate)) plus the ball of radius
// `force` is cleared in the live volume.  A cube row (z,y) is a bit mask of W 64-bit
// words (2r+1 <= 64 W), built with ballots.  W = 1 (obj_min_dist <= 31, every caller of the
// reference uses 27): both row sets live in LDS; W > 1: in a per-workgroup slice of
// `rows_glob` (the cube of r = 40 is 2 x 105 KB: past the LDS).
__device__ __forceinline__ unsigned long long bit_range(int lo, int hi, int w) {
  // bits lo..hi (inclusive) of a row, as they fall into word w
  const int a = (lo > 64 * w ? lo : 64 * w) - 64 * w;
  const int e = (hi < 64 * w + 63 ? hi : 64 * w + 63) - 64 * w;
  if (a > e) return 0ull;
  const int n = e - a + 1;
  return (n == 64 ? ~0ull : ((1ull << n) - 1ull)) << a;
}

template <int W>
__global__ __launch_bounds__(256) void clear_balls_seg(
    const unsigned long long *__restrict__ round_list,
    unsigned long long *__restrict__ counters, float *__restrict__ live,
    int64_t P1, int64_t P2, int r,
    unsigned long long *__restrict__ best, int64_t C1, int64_t C2,
    const unsigned long long *__restrict__ seg, int dilate, int force,
    unsigned int *__restrict__ dirty_list, unsigned long long *__restrict__ rows_glob) {
  extern __shared__ unsigned long long rows_lds[];       // W == 1: 2 x side*side
  __shared__ unsigned int stage[CLR_STAGE];
  __shared__ unsigned int n_stage, n_gone, stage_base;
  if (threadIdx.x == 0) { n_stage = 0u; n_gone = 0u; }
  __syncthreads();
  const int side = 2 * r + 1, nrows = side * side;
  unsigned long long *ma = W == 1 ? rows_lds : rows_glob + (size_t)blockIdx.x * 2 * nrows * W;
  unsigned long long *mb = ma + (size_t)nrows * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long nwin = counters[CNT_ROUND];
  for (unsigned long long wi = blockIdx.x; wi < nwin; wi += gridDim.x) {
    const uint32_t flat = 0xFFFFFFFFu - (uint32_t)(round_list[wi] & 0xFFFFFFFFu);
    const int64_t x = flat % P2, y = (flat / P2) % P1, z = flat / (P2 * P1);
    const unsigned long long id = seg[flat];
    __syncthreads();                       // previous winner done with the rows
    for (int row = wave; row < nrows; row += 4) {
      const int dz = row / side - r, dy = row % side - r;
      const unsigned long long *srow = seg + ((z + dz) * P1 + (y + dy)) * P2 + x - r;
#pragma unroll
      for (int w = 0; w < W; ++w) {
        const int xi = 64 * w + lane;
        const bool same = xi < side && srow[xi] == id;
        const unsigned long long m = __ballot(same);
        if (lane == 0) ma[(size_t)row * W + w] = m;
      }
    }
    __syncthreads();
    for (int it = 0; it < dilate; ++it) {
      for (int row = threadIdx.x; row < nrows; row += 256) {
        const int rz = row / side, ry = row % side;
        unsigned long long m[W];
#pragma unroll
        for (int w = 0; w < W; ++w) m[w] = ma[(size_t)row * W + w];
#pragma unroll
        for (int w = 0; w < W; ++w) {
          unsigned long long v = m[w] | (m[w] << 1) | (m[w] >> 1);
          if (w > 0) v |= m[w - 1] >> 63;
          if (w + 1 < W) v |= m[w + 1] << 63;
          if (rz > 0) v |= ma[(size_t)(row - side) * W + w];
          if (rz + 1 < side) v |= ma[(size_t)(row + side) * W + w];
          if (ry > 0) v |= ma[(size_t)(row - 1) * W + w];
          if (ry + 1 < side) v |= ma[(size_t)(row + 1) * W + w];
          mb[(size_t)row * W + w] = v & bit_range(0, side - 1, w);
        }
      }
      __syncthreads();
      unsigned long long *t = ma; ma = mb; mb = t;
    }
    for (int row = wave; row < nrows; row += 4) {
      const int dz = row / side - r, dy = row % side - r;
      const int rem = r * r - dz * dz - dy * dy;
      int hx = -1, hf = -1;
      if (rem >= 0) {
        hx = (int)sqrtf((float)rem);
        while ((hx + 1) * (hx + 1) <= rem) ++hx;
        while (hx * hx > rem) --hx;
      }
      const int remf = force * force - dz * dz - dy * dy;
      if (force > 0 && remf >= 0) {
        hf = (int)sqrtf((float)remf);
        while ((hf + 1) * (hf + 1) <= remf) ++hf;
        while (hf * hf > remf) --hf;
      }
      float *lrow = live + ((z + dz) * P1 + (y + dy)) * P2 + x - r;
#pragma unroll
      for (int w = 0; w < W; ++w) {
        unsigned long long bits = 0;
        if (hx >= 0) bits = bit_range(r - hx, r + hx, w) & ma[(size_t)row * W + w];
        if (hf >= 0) bits |= bit_range(r - hf, r + hf, w);
        const int xi = 64 * w + lane;
        if (xi < side && ((bits >> lane) & 1ull)) lrow[xi] = 0.f;
      }
    }
    const int64_t cz0 = (z - r) / CELL, cy0 = (y - r) / CELL, cx0 = (x - r) / CELL;
    const int nz = (int)((z + r) / CELL - cz0 + 1), ny = (int)((y + r) / CELL - cy0 + 1),
              nx = (int)((x + r) / CELL - cx0 + 1);
    for (int i = threadIdx.x; i < nz * ny * nx; i += blockDim.x) {
      const int64_t cz = cz0 + i / (ny * nx), cy = cy0 + (i / nx) % ny, cx = cx0 + i % nx;
      touch_cell(best, (cz * C1 + cy) * C2 + cx, false, stage, &n_stage, &n_gone, dirty_list,
                 counters);
    }
    flush_stage(stage, &n_stage, &n_gone, &stage_base, dirty_list, counters);
  }
}

struct RankQuery {
  int64_t rank;       // remaining rank inside the current prefix
  uint32_t prefix;    // key bits fixed so far
  int64_t count;      // elements that share the prefix (size of the bin last chosen)
};

}  // namespace

// exact order statistics of S.smoothed: 3-level radix select (10 / 11 / 11 bits) on the
// monotone float key.  Level 0 comes out of the x pass when `have_level0` (its
// histogram is already in hist_dev); the elements of the selected level-0 bin are
// then compacted (one filtered scan) and levels 1-2 run on that short list.
// hist_dev: 2049 u64 (level 0 uses the first L0_BINS); scratch: n_pad floats.
static int v2o_select(fpl_ctx *ctx, const V2oState &S, int64_t n_pad, const int64_t *ranks,
                      int32_t n_ranks, float *rank_values, unsigned long long *hist_dev,
                      float *scratch, bool windowed, DevTemp &tmp, float floor_v = 0.f) {
  hipStream_t st = ctx->stream;
  void *p;
    FPL_REQUIRE(ctx, rank_values, "fpl_v2o_smooth: rank_values is NULL");
    std::vector<unsigned long long> hist(2048);
    std::vector<RankQuery> q(n_ranks);
    for (int i = 0; i < n_ranks; ++i) q[i] = RankQuery{ranks[i], 0u, n_pad};
    const int shifts[3] = {L0_SHIFT, 11, 0}, nbits[3] = {L0_BITS, 11, 11};
    auto hgrid_for = [&](int64_t n) {
      return (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div64(n, 256 * 8),
                                                              (int64_t)ctx->n_cu * 16));
    };
    // compaction geometry: one chunk per workgroup
    const int64_t cgrid = std::max<int64_t>(1, std::min<int64_t>(ceil_div64(n_pad, 4096),
                                                                 (int64_t)ctx->n_cu * 16));
    const int64_t chunk = ceil_div64(ceil_div64(n_pad, cgrid), 256) * 256;
    const int64_t nchunks = ceil_div64(n_pad, chunk);
    FPL_TRY(tmp.alloc((size_t)nchunks * sizeof(unsigned int), &p));
    unsigned int *counts_dev = (unsigned int *)p;
    // one level for the queries `idx` that share a prefix: histogram (already in
    // hist_dev if `have`; of the compacted chunks if `chunks`), then rank -> bin
    auto resolve = [&](int lvl, uint32_t mask, const std::vector<int> &idx, bool have,
                       bool chunks) -> int {
      const uint32_t pref = q[idx[0]].prefix;
      if (!have) {
        FPL_HIP(ctx, hipMemsetAsync(hist_dev, 0, 2048 * sizeof(unsigned long long), st));
        TimedLaunch tl(ctx, "v2o_key_histogram");
        if (chunks)
          key_histogram_chunks<<<(unsigned)nchunks, 256, 0, st>>>(
              scratch, counts_dev, chunk, mask, pref, shifts[lvl], nbits[lvl], hist_dev);
        else
          key_histogram<<<hgrid_for(n_pad), 256, 0, st>>>(S.smoothed, n_pad, mask, pref,
                                                          shifts[lvl], nbits[lvl], hist_dev);
      }
      FPL_HIP(ctx, hipMemcpyAsync(hist.data(), hist_dev, 2048 * sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, st));
      FPL_HIP(ctx, hipStreamSynchronize(st));
      const int nb = 1 << nbits[lvl];
      for (int j : idx) {
        int64_t k = q[j].rank;
        int b = 0;
        for (; b < nb; ++b) {
          if (k < (int64_t)hist[b]) break;
          k -= (int64_t)hist[b];
        }
        FPL_REQUIRE(ctx, b < nb, "fpl_v2o_smooth: radix select ran off the histogram "
                                 "(NaN in the volume?)");
        q[j].rank = k;
        q[j].prefix = pref | ((uint32_t)b << shifts[lvl]);
        q[j].count = (int64_t)hist[b];
      }
      return 0;
    };
    auto groups_of = [&](const std::vector<int> &idx) {
      std::vector<std::vector<int>> g;
      for (int j : idx) {
        bool placed = false;
        for (auto &v : g)
          if (q[v[0]].prefix == q[j].prefix) { v.push_back(j); placed = true; break; }
        if (!placed) g.push_back({j});
      }
      return g;
    };
    std::vector<int> all(n_ranks);
    for (int i = 0; i < n_ranks; ++i) all[i] = i;
    FPL_TRY(resolve(0, 0u, all, windowed, false));
    if (floor_v > 0.f) {
      // the caller only uses max(statistic, floor) (fpl_v2o_set_floor): when every rank
      // falls in a first-level bin below the floor's, each statistic is < floor and is
      // reported as the floor - no compaction, no further levels, no further syncs
      uint32_t fb;
      memcpy(&fb, &floor_v, 4);
      const uint32_t floor_bin = (fb | 0x80000000u) >> L0_SHIFT;
      bool below = true;
      for (int i = 0; i < n_ranks; ++i) below = below && (q[i].prefix >> L0_SHIFT) < floor_bin;
      if (below) {
        for (int i = 0; i < n_ranks; ++i) rank_values[i] = floor_v;
        return 0;
      }
    }
    const uint32_t mask0 = (uint32_t)(L0_BINS - 1) << L0_SHIFT, mask1 = mask0 | (0x7FFu << 11);
    for (auto &grp : groups_of(all)) {
      // A first-level bin is a half octave: on dense predictions [0.5, 0.75) alone can
      // hold a third of the volume, and packing that list plus two histograms over it
      // costs more than a second look at the volume.  Fat bin (> 1/8 of the volume):
      // level 1 as a filtered histogram of the VOLUME, then the filtered scan packs the
      // 21-bit prefix's few elements (1.02 -> 0.55 ms on such a 582^3 substack).
      if (q[grp[0]].count * 8 > n_pad) {
        FPL_TRY(resolve(1, mask0, grp, false, false));
        for (auto &g2 : groups_of(grp)) {
          {
            TimedLaunch tl(ctx, "v2o_compact_bin");
            compact_prefix<<<(unsigned)nchunks, 256, 0, st>>>(
                S.smoothed, n_pad, chunk, mask1, q[g2[0]].prefix, scratch, counts_dev);
          }
          FPL_TRY(resolve(2, mask1, g2, false, true));
        }
        continue;
      }
      {
        TimedLaunch tl(ctx, "v2o_compact_bin");
        compact_prefix<<<(unsigned)nchunks, 256, 0, st>>>(S.smoothed, n_pad, chunk, mask0,
                                                          q[grp[0]].prefix, scratch, counts_dev);
      }
      FPL_TRY(resolve(1, mask0, grp, false, true));
      for (auto &g2 : groups_of(grp)) FPL_TRY(resolve(2, mask1, g2, false, true));
    }
    for (int i = 0; i < n_ranks; ++i) {
      const uint32_t k = q[i].prefix;
      const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
      memcpy(&rank_values[i], &u, 4);
    }
  return 0;
}

extern "C" {

int fpl_v2o_smooth(fpl_ctx *ctx, const float *pred, int pred_mem,
                   const int64_t dims[3], int32_t r, const double *weights,
                   int32_t wr, const int64_t *ranks, int32_t n_ranks,
                   float *rank_values) {
  if (!ctx || !pred || !dims || !weights)
    return fpl_fail(ctx, "fpl_v2o_smooth: NULL argument");
  FPL_REQUIRE(ctx, r >= 0 && wr >= 0, "fpl_v2o_smooth: negative radius");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  int64_t P[3];
  for (int a = 0; a < 3; ++a) {
    FPL_REQUIRE(ctx, dims[a] > 0, "fpl_v2o_smooth: dims[%d] = %lld", a,
                (long long)dims[a]);
    P[a] = dims[a] + 2 * (int64_t)r;
  }
  const int64_t n_pad = P[0] * P[1] * P[2];
  FPL_REQUIRE(ctx, n_pad < ((int64_t)1 << 32) - 1,
              "fpl_v2o_smooth: padded volume has %lld voxels; the NMS keys hold "
              "32-bit flat indices - process it as substacks (as "
              "fplobjdetect.full_roi_inference does)", (long long)n_pad);
  for (int i = 0; i < n_ranks; ++i)
    FPL_REQUIRE(ctx, ranks[i] >= 0 && ranks[i] < n_pad,
                "fpl_v2o_smooth: rank %lld out of range", (long long)ranks[i]);
  hipStream_t st = ctx->stream;
  DevTemp tmp(ctx);
  V2oState &S = ctx->v2o;
  S.valid = false;
  S.seg_valid = false;            // a segmentation belongs to one smoothed volume
  S.f64 = false;
  S.sorted = false;
  const size_t vol_bytes = (size_t)n_pad * sizeof(float);
  if (S.cap_bytes < vol_bytes) {
    if (S.smoothed) fpl_dev_release(ctx, S.smoothed);
    S.smoothed = nullptr;
    S.cap_bytes = 0;
    void *p;
    FPL_TRY(fpl_dev_alloc(ctx, vol_bytes, &p));
    S.smoothed = (float *)p;
    S.cap_bytes = vol_bytes;
  }
  const float *pred_dev = pred;
  if (pred_mem == FPL_MEM_HOST) {
    void *p;
    const size_t nb = (size_t)(dims[0] * dims[1] * dims[2]) * sizeof(float);
    FPL_TRY(tmp.alloc(nb, &p));
    FPL_HIP(ctx, hipMemcpyAsync(p, pred, nb, hipMemcpyHostToDevice, st));
    pred_dev = (const float *)p;
  }
  void *p;
  FPL_TRY(tmp.alloc(vol_bytes, &p));
  float *scratch = (float *)p;
  FPL_TRY(tmp.alloc((size_t)(wr + 1) * sizeof(double), &p));
  double *w_dev = (double *)p;
  // weights[0..2wr] symmetric; kernel wants w[j] by distance j
  FPL_HIP(ctx, hipMemcpyAsync(w_dev, weights + wr, (size_t)(wr + 1) * sizeof(double),
                              hipMemcpyHostToDevice, st));
  PadView pv{pred_dev, dims[0], dims[1], dims[2], r};
  const unsigned grid = (unsigned)ceil_div64(n_pad, 256);
  // [0..2047] histogram, [2048] compaction counter
  FPL_TRY(tmp.alloc(2049 * sizeof(unsigned long long), &p));
  unsigned long long *hist_dev = (unsigned long long *)p;
  FPL_HIP(ctx, hipMemsetAsync(hist_dev, 0, 2049 * sizeof(unsigned long long), st));
  // register-window kernels for the kernel radii flypylib's sigmas produce
  // (sigma 1.5, 2, 3, 5 at truncate 2.0); the plain kernel covers the rest
  // per-cell keys for the NMS, filled by the fused pass
  S.cellmax_valid = false;
  {
    const size_t need = (size_t)(ceil_div64(P[0], CELL) * ceil_div64(P[1], CELL) *
                                 ceil_div64(P[2], CELL)) * sizeof(unsigned long long);
    if (S.cellmax_cap_bytes < need) {
      if (S.cellmax) fpl_dev_release(ctx, S.cellmax);
      S.cellmax = nullptr;
      S.cellmax_cap_bytes = 0;
      void *q;
      FPL_TRY(fpl_dev_alloc(ctx, need, &q));
      S.cellmax = (unsigned long long *)q;
      S.cellmax_cap_bytes = need;
    }
  }
  bool cm_done = false;
  const float floor_v = S.floor > 0.f ? S.floor : 0.f;
  S.floor = 0.f;                         // a floor holds for one smoothing
  S.cellmax_floor = floor_v;
  bool windowed = true;
  switch (wr) {
    case 3: FPL_TRY(launch_gauss_win<3>(ctx, pv, S.smoothed, scratch, P, w_dev, r, hist_dev, S.cellmax, floor_v, &cm_done)); break;
    case 4: FPL_TRY(launch_gauss_win<4>(ctx, pv, S.smoothed, scratch, P, w_dev, r, hist_dev, S.cellmax, floor_v, &cm_done)); break;
    case 6: FPL_TRY(launch_gauss_win<6>(ctx, pv, S.smoothed, scratch, P, w_dev, r, hist_dev, S.cellmax, floor_v, &cm_done)); break;
    case 10: FPL_TRY(launch_gauss_win<10>(ctx, pv, S.smoothed, scratch, P, w_dev, r, hist_dev, S.cellmax, floor_v, &cm_done)); break;
    default: windowed = false;
  }
  if (!windowed) {
    {
      TimedLaunch tl(ctx, "v2o_gauss_z");
      gauss_pass<0><<<grid, 256, 0, st>>>(pv, nullptr, S.smoothed, P[0], P[1], P[2],
                                          w_dev, wr, r);
    }
    {
      TimedLaunch tl(ctx, "v2o_gauss_y");
      gauss_pass<1><<<grid, 256, 0, st>>>(pv, S.smoothed, scratch, P[0], P[1], P[2],
                                          w_dev, wr, r);
    }
    {
      TimedLaunch tl(ctx, "v2o_gauss_x");
      gauss_pass<2><<<grid, 256, 0, st>>>(pv, scratch, S.smoothed, P[0], P[1], P[2],
                                          w_dev, wr, r);
    }
  }
  FPL_HIP(ctx, hipGetLastError());
  for (int a = 0; a < 3; ++a) S.pdims[a] = P[a];
  S.r = r;
  S.cellmax_valid = cm_done;

  if (n_ranks > 0)
    FPL_TRY(v2o_select(ctx, S, n_pad, ranks, n_ranks, rank_values, hist_dev, scratch, windowed, tmp,
                       floor_v));
  FPL_HIP(ctx, hipStreamSynchronize(st));
  S.valid = true;
  return 0;
}

int fpl_v2o_set_floor(fpl_ctx *ctx, float floor) {
  if (!ctx) return fpl_fail(nullptr, "fpl_v2o_set_floor: ctx is NULL");
  FPL_REQUIRE(ctx, floor == floor, "fpl_v2o_set_floor: NaN");
  ctx->v2o.floor = floor;
  return 0;
}

int fpl_v2o_copy_smoothed(fpl_ctx *ctx, float *dst, int dst_mem) {
  if (!ctx || !dst) return fpl_fail(ctx, "fpl_v2o_copy_smoothed: NULL argument");
  FPL_REQUIRE(ctx, ctx->v2o.valid, "fpl_v2o_copy_smoothed: no smoothed volume");
  const V2oState &S = ctx->v2o;
  const size_t nb = (size_t)(S.pdims[0] * S.pdims[1] * S.pdims[2]) * sizeof(float);
  FPL_HIP(ctx, hipMemcpyAsync(dst, S.smoothed, nb,
                              dst_mem == FPL_MEM_HOST ? hipMemcpyDeviceToHost
                                                      : hipMemcpyDeviceToDevice,
                              ctx->stream));
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

static int v2o_nms(fpl_ctx *ctx, double thresh, double *out_zyxv, int64_t cap,
                   int64_t *n_out, int32_t *n_rounds, bool use_seg, int seg_dilate,
                   int seg_force) {
  if (!ctx || !out_zyxv || !n_out)
    return fpl_fail(ctx, "fpl_v2o_nms: NULL argument");
  FPL_REQUIRE(ctx, ctx->v2o.valid,
              "fpl_v2o_nms: call fpl_v2o_smooth first");
  FPL_REQUIRE(ctx, cap > 0, "fpl_v2o_nms: cap must be positive");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  V2oState &S = ctx->v2o;
  S.valid = false;                 // the NMS suppresses in place: the volume is consumed
  const int64_t P0 = S.pdims[0], P1 = S.pdims[1], P2 = S.pdims[2];
  const int r = S.r;
  hipStream_t st = ctx->stream;
  DevTemp tmp(ctx);
  const int64_t C0 = ceil_div64(P0, CELL), C1 = ceil_div64(P1, CELL),
                C2 = ceil_div64(P2, CELL);
  const int64_t n_cells = C0 * C1 * C2;
  float *live = S.smoothed;
  // window half-width in cells: cells [c-hw, c+hw] cover [4c-4hw, 4c+4hw+3]
  // which must contain [p-r, p+r] for every p in [4c, 4c+3]
  const int hw = (r + CELL - 1) / CELL;
  void *p;
  // cell keys: the ones the fused smoothing pass left (consumed here), else a scan
  // (keys taken above a floor are complete only for thresholds >= that floor)
  const bool have_keys = S.cellmax_valid && S.cellmax != nullptr &&
                         thresh >= (double)S.cellmax_floor;
  S.cellmax_valid = false;
  unsigned long long *best = S.cellmax;
  if (!have_keys) {
    FPL_TRY(tmp.alloc((size_t)n_cells * 8, &p));
    best = (unsigned long long *)p;
  }
  FPL_TRY(tmp.alloc((size_t)n_cells * 8, &p));
  unsigned long long *wa = (unsigned long long *)p;
  FPL_TRY(tmp.alloc((size_t)n_cells * 8, &p));
  unsigned long long *wb = (unsigned long long *)p;
  FPL_TRY(tmp.alloc((size_t)n_cells * 8, &p));
  unsigned long long *round_list = (unsigned long long *)p;
  FPL_TRY(tmp.alloc((size_t)cap * 8, &p));
  unsigned long long *all_list = (unsigned long long *)p;
  FPL_TRY(tmp.alloc((size_t)n_cells * sizeof(unsigned int), &p));
  unsigned int *dirty_list = (unsigned int *)p;             // cells to re-scan, per round
  // segmentation-aware suppression with rows wider than 64 voxels: the two row sets of
  // every workgroup's cube live in global memory
  unsigned long long *seg_rows = nullptr;
  constexpr int SEG_WGS = 1024;
  if (use_seg && 2 * r + 1 > 64) {
    const size_t W = (2 * r + 1 + 63) / 64 <= 2 ? 2 : 4;
    FPL_TRY(tmp.alloc((size_t)SEG_WGS * 2 * (2 * r + 1) * (2 * r + 1) * W * 8, &p));
    seg_rows = (unsigned long long *)p;
  }
  FPL_TRY(tmp.alloc(CNT_N * 8, &p));
  unsigned long long *counters = (unsigned long long *)p;   // CNT_*
  FPL_HIP(ctx, hipMemsetAsync(counters, 0, CNT_N * 8, st));
  const unsigned cgrid = (unsigned)ceil_div64(n_cells, 256);
  const unsigned wgrid = (cgrid + 7u) / 8u * 8u;       // window_max: whole XCD rounds
  const unsigned bgrid = std::min<unsigned>(cgrid, (unsigned)ctx->n_cu * 8);   // cell_best: grid-stride
  unsigned long long host_cnt[CNT_N];
  const bool aligned = P2 % CELL == 0;
  // live cells: keys per cell and their count in counters[CNT_LIVE]
  {
    TimedLaunch tl(ctx, "v2o_cell_best");
    if (have_keys)
      cell_threshold<<<bgrid, 256, 0, st>>>(best, n_cells, thresh, counters);
    else if (aligned)
      cell_best<true><<<bgrid, 256, 0, st>>>(live, thresh, P0, P1, P2, C0, C1, C2, best, counters);
    else
      cell_best<false><<<bgrid, 256, 0, st>>>(live, thresh, P0, P1, P2, C0, C1, C2, best, counters);
  }
  // A round = winners (cells whose key is the maximum of their window) -> clear their
  // balls (the cells a ball cuts go on a list and out of the live count) -> re-scan the
  // listed cells.  Every kernel of a round is a no-op once no cell is live, so rounds are
  // enqueued NMS_BATCH at a time and the host looks at the counters once per batch
  // (typical substacks finish in two rounds: one sync).
  constexpr int NMS_BATCH = 2;
  int rounds = 0;
  for (;;) {
    for (int b = 0; b < NMS_BATCH; ++b) {
      // winners and re-scan list of this round
      FPL_HIP(ctx, hipMemsetAsync(counters + CNT_ROUND, 0, 2 * 8, st));
      {
        TimedLaunch tl(ctx, "v2o_window_max");
        const bool seg_ok = hw >= 1 && hw <= 8 && n_cells < ((int64_t)1 << 31) &&
                            !getenv("FPL_V2O_WMAX_PLAIN");
        switch (seg_ok ? hw : 0) {
          case 1: launch_window_max_seg<1>(st, best, wa, wb, (int)C0, (int)C1, (int)C2, n_cells, counters); break;
          case 2: launch_window_max_seg<2>(st, best, wa, wb, (int)C0, (int)C1, (int)C2, n_cells, counters); break;
          case 3: launch_window_max_seg<3>(st, best, wa, wb, (int)C0, (int)C1, (int)C2, n_cells, counters); break;
          case 4: launch_window_max_seg<4>(st, best, wa, wb, (int)C0, (int)C1, (int)C2, n_cells, counters); break;
          case 5: launch_window_max_seg<5>(st, best, wa, wb, (int)C0, (int)C1, (int)C2, n_cells, counters); break;
          case 6: launch_window_max_seg<6>(st, best, wa, wb, (int)C0, (int)C1, (int)C2, n_cells, counters); break;
          case 7: launch_window_max_seg<7>(st, best, wa, wb, (int)C0, (int)C1, (int)C2, n_cells, counters); break;
          case 8: launch_window_max_seg<8>(st, best, wa, wb, (int)C0, (int)C1, (int)C2, n_cells, counters); break;
          default:
            window_max<2><<<wgrid, 256, 0, st>>>(best, wa, C0, C1, C2, hw, counters);
            window_max<1><<<wgrid, 256, 0, st>>>(wa, wb, C0, C1, C2, hw, counters);
            window_max<0><<<wgrid, 256, 0, st>>>(wb, wa, C0, C1, C2, hw, counters);
        }
      }
      {
        TimedLaunch tl(ctx, "v2o_pick_winners");
        pick_winners<<<cgrid, 256, 0, st>>>(best, wa, n_cells, counters, round_list,
                                            all_list, cap);
      }
      {
        TimedLaunch tl(ctx, "v2o_clear_balls");
        if (use_seg) {
          const size_t rows = (size_t)2 * (2 * r + 1) * (2 * r + 1) * sizeof(unsigned long long);
          const int W = (2 * r + 1 + 63) / 64;
          if (W == 1)
            clear_balls_seg<1><<<SEG_WGS, 256, rows, st>>>(round_list, counters, live, P1, P2, r, best,
                                                           C1, C2, S.seg, seg_dilate, seg_force,
                                                           dirty_list, nullptr);
          else if (W == 2)
            clear_balls_seg<2><<<SEG_WGS, 256, 0, st>>>(round_list, counters, live, P1, P2, r, best, C1,
                                                        C2, S.seg, seg_dilate, seg_force, dirty_list,
                                                        seg_rows);
          else
            clear_balls_seg<4><<<SEG_WGS, 256, 0, st>>>(round_list, counters, live, P1, P2, r, best, C1,
                                                        C2, S.seg, seg_dilate, seg_force, dirty_list,
                                                        seg_rows);
        } else {
          clear_balls<<<1024, 256, 0, st>>>(round_list, counters, live, P1, P2, r, best, C1, C2,
                                            dirty_list);
        }
      }
      {
        TimedLaunch tl(ctx, "v2o_cell_best");
        if (aligned)
          cell_rescan<true><<<bgrid, 256, 0, st>>>(live, thresh, P0, P1, P2, C1, C2, best,
                                                   dirty_list, counters);
        else
          cell_rescan<false><<<bgrid, 256, 0, st>>>(live, thresh, P0, P1, P2, C1, C2, best,
                                                    dirty_list, counters);
      }
    }
    FPL_HIP(ctx, hipGetLastError());
    FPL_HIP(ctx, hipMemcpyAsync(host_cnt, counters, CNT_N * 8, hipMemcpyDeviceToHost, st));
    FPL_HIP(ctx, hipStreamSynchronize(st));
    FPL_REQUIRE(ctx, (int64_t)host_cnt[CNT_TOTAL] <= cap,
                "fpl_v2o_nms: more than %lld detections; raise cap",
                (long long)cap);
    const int done = (int)host_cnt[CNT_ROUNDS];
    if (getenv("FPL_V2O_DBG"))
      fprintf(stderr, "nms: rounds %d live %llu winners(last round) %llu dirty %llu total %llu of %lld cells\n", done,
              host_cnt[CNT_LIVE], host_cnt[CNT_ROUND], host_cnt[CNT_DIRTY], host_cnt[CNT_TOTAL], (long long)n_cells);
    if (host_cnt[CNT_LIVE] == 0) { rounds = done; break; }
    FPL_REQUIRE(ctx, done == rounds + NMS_BATCH && host_cnt[CNT_ROUND] > 0,
                "fpl_v2o_nms: round %d made no progress (internal error)", done);
    rounds = done;
  }
  const int64_t n = (int64_t)host_cnt[CNT_TOTAL];
  std::vector<unsigned long long> keys((size_t)n);
  if (n)
    FPL_HIP(ctx, hipMemcpy(keys.data(), all_list, (size_t)n * 8,
                           hipMemcpyDeviceToHost));
  // key order descending = (value desc, flat index asc)
  std::sort(keys.begin(), keys.end(),
            [](unsigned long long a, unsigned long long b) { return a > b; });
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t flat = 0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFu);
    const uint32_t vb = (uint32_t)(keys[i] >> 32);
    float v;
    memcpy(&v, &vb, 4);
    out_zyxv[4 * i + 0] = (double)(flat / (P2 * P1));
    out_zyxv[4 * i + 1] = (double)((flat / P2) % P1);
    out_zyxv[4 * i + 2] = (double)(flat % P2);
    out_zyxv[4 * i + 3] = (double)v;
  }
  *n_out = n;
  if (n_rounds) *n_rounds = rounds;
  return 0;
}


int fpl_v2o_nms(fpl_ctx *ctx, double thresh, double *out_zyxv, int64_t cap,
                int64_t *n_out, int32_t *n_rounds) {
  return v2o_nms(ctx, thresh, out_zyxv, cap, n_out, n_rounds, false, 0, 0);
}

int fpl_v2o_nms_seg(fpl_ctx *ctx, double thresh, int32_t seg_dilate, int32_t seg_force,
                    double *out_zyxv, int64_t cap, int64_t *n_out, int32_t *n_rounds) {
  if (!ctx) return fpl_fail(nullptr, "fpl_v2o_nms_seg: ctx is NULL");
  FPL_REQUIRE(ctx, ctx->v2o.valid && ctx->v2o.seg_valid,
              "fpl_v2o_nms_seg: call fpl_v2o_smooth and fpl_v2o_set_seg first");
  FPL_REQUIRE(ctx, 2 * ctx->v2o.r + 1 <= 256,
              "fpl_v2o_nms_seg: obj_min_dist %d > 127 (a cube row is at most four 64-bit masks)",
              ctx->v2o.r);
  FPL_REQUIRE(ctx, seg_dilate >= 0 && seg_force >= 0 && seg_force <= ctx->v2o.r,
              "fpl_v2o_nms_seg: seg_dilate %d / seg_force %d out of range", seg_dilate, seg_force);
  // function attributes belong to the current device: one flag per device (a process may
  // drive several GPUs, one context each; setting it twice is harmless)
  static bool attr_set[FPL_MAX_DEVICES] = {false};
  if (!attr_set[ctx->device % FPL_MAX_DEVICES]) {
    FPL_HIP(ctx, hipFuncSetAttribute((const void *)clear_balls_seg<1>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 63 * 63 * 8));
    attr_set[ctx->device % FPL_MAX_DEVICES] = true;
  }
  return v2o_nms(ctx, thresh, out_zyxv, cap, n_out, n_rounds, true, seg_dilate, seg_force);
}

int fpl_v2o_set_seg(fpl_ctx *ctx, const void *seg, int32_t seg_bytes, int seg_mem,
                    const int64_t dims[3], int64_t sz_thd) {
  if (!ctx || !seg || !dims) return fpl_fail(ctx, "fpl_v2o_set_seg: NULL argument");
  V2oState &S = ctx->v2o;
  FPL_REQUIRE(ctx, S.valid, "fpl_v2o_set_seg: call fpl_v2o_smooth first");
  FPL_REQUIRE(ctx, seg_bytes == 4 || seg_bytes == 8, "fpl_v2o_set_seg: labels must be 4 or 8 bytes");
  const int r = S.r;
  for (int a = 0; a < 3; ++a)
    FPL_REQUIRE(ctx, dims[a] + 2 * r == S.pdims[a],
                "fpl_v2o_set_seg: segmentation dims differ from the prediction's");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DevTemp tmp(ctx);
  const int64_t n = dims[0] * dims[1] * dims[2];
  const int64_t n_pad = S.pdims[0] * S.pdims[1] * S.pdims[2];
  const void *src = seg;
  if (seg_mem == FPL_MEM_HOST) {
    void *p;
    FPL_TRY(tmp.alloc((size_t)n * seg_bytes, &p));
    FPL_HIP(ctx, hipMemcpyAsync(p, seg, (size_t)n * seg_bytes, hipMemcpyHostToDevice, st));
    src = p;
  }
  const size_t need = (size_t)n_pad * sizeof(unsigned long long);
  if (S.seg_cap_bytes < need) {
    if (S.seg) fpl_dev_release(ctx, S.seg);
    S.seg = nullptr; S.seg_cap_bytes = 0;
    void *p;
    FPL_TRY(fpl_dev_alloc(ctx, need, &p));
    S.seg = (unsigned long long *)p; S.seg_cap_bytes = need;
  }
  const unsigned grid = (unsigned)ceil_div64(n_pad, 256);
  {
    TimedLaunch tl(ctx, "v2o_seg_pad");
    if (seg_bytes == 8)
      seg_pad<unsigned long long><<<grid, 256, 0, st>>>((const unsigned long long *)src, dims[0],
          dims[1], dims[2], r, S.seg, S.pdims[1], S.pdims[2], n_pad);
    else
      seg_pad<uint32_t><<<grid, 256, 0, st>>>((const uint32_t *)src, dims[0], dims[1], dims[2],
          r, S.seg, S.pdims[1], S.pdims[2], n_pad);
  }
  FPL_HIP(ctx, hipGetLastError());
  if (sz_thd >= 0) {
    S.cellmax_valid = false;      // the volume is edited below: the fused cell keys are stale
    // voxel count per label (the zero padding counts towards label 0, as in the
    // reference, which pads before np.unique) and the zeroing of small segments
    uint64_t cap = 1ull << 16;
    while (cap < (uint64_t)n_pad / 4 && cap < (1ull << 25)) cap <<= 1;
    void *p;
    FPL_TRY(tmp.alloc(cap * sizeof(unsigned long long), &p));
    unsigned long long *keys = (unsigned long long *)p;
    FPL_TRY(tmp.alloc(cap * sizeof(unsigned int) + 16, &p));
    unsigned int *counts = (unsigned int *)p;
    unsigned int *overflow = counts + cap;
    FPL_HIP(ctx, hipMemsetAsync(keys, 0, cap * sizeof(unsigned long long), st));
    FPL_HIP(ctx, hipMemsetAsync(counts, 0, cap * sizeof(unsigned int) + 16, st));
    {
      TimedLaunch tl(ctx, "v2o_seg_count");
      seg_count<<<(unsigned)ceil_div64(ceil_div64(n_pad, 16), 256), 256, 0, st>>>(
          S.seg, n_pad, keys, counts, cap - 1, overflow);
    }
    unsigned int ovf = 0;
    FPL_HIP(ctx, hipMemcpyAsync(&ovf, overflow, 4, hipMemcpyDeviceToHost, st));
    FPL_HIP(ctx, hipStreamSynchronize(st));
    FPL_REQUIRE(ctx, ovf == 0, "fpl_v2o_set_seg: more than %llu distinct labels",
                (unsigned long long)cap);
    {
      TimedLaunch tl(ctx, "v2o_seg_zero_small");
      if (S.f64) {               // float64 prediction: the volume of record is the double one
        seg_zero_small<double><<<grid, 256, 0, st>>>(S.seg, n_pad, keys, counts, cap - 1,
                                                     (long long)sz_thd, S.smoothed64);
        S.sorted = false;
      } else {
        seg_zero_small<float><<<grid, 256, 0, st>>>(S.seg, n_pad, keys, counts, cap - 1,
                                                    (long long)sz_thd, S.smoothed);
      }
    }
    FPL_HIP(ctx, hipGetLastError());
  }
  FPL_HIP(ctx, hipStreamSynchronize(st));
  S.seg_valid = true;
  return 0;
}

int fpl_v2o_select(fpl_ctx *ctx, const int64_t *ranks, int32_t n_ranks, float *rank_values) {
  if (!ctx || !ranks || !rank_values) return fpl_fail(ctx, "fpl_v2o_select: NULL argument");
  const V2oState &S = ctx->v2o;
  FPL_REQUIRE(ctx, S.valid, "fpl_v2o_select: call fpl_v2o_smooth first");
  FPL_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t n_pad = S.pdims[0] * S.pdims[1] * S.pdims[2];
  for (int i = 0; i < n_ranks; ++i)
    FPL_REQUIRE(ctx, ranks[i] >= 0 && ranks[i] < n_pad, "fpl_v2o_select: rank %lld out of range",
                (long long)ranks[i]);
  DevTemp tmp(ctx);
  void *p;
  FPL_TRY(tmp.alloc(2049 * sizeof(unsigned long long), &p));
  unsigned long long *hist_dev = (unsigned long long *)p;
  FPL_TRY(tmp.alloc((size_t)n_pad * sizeof(float), &p));
  if (n_ranks > 0)
    FPL_TRY(v2o_select(ctx, S, n_pad, ranks, n_ranks, rank_values, hist_dev, (float *)p, false, tmp));
  FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

}  // extern "C"
