// The U-Net's 3x3x3 convolutions on split halves with EVERY operand served from LDS (round 5):
// what vgg_split_lds.h gave vgg_like's mid / tail kernels in round 4, for unet_like2 / 3 / 4
// (flypylib/fplmodels.py:258-407).  Included by conv_mfma.hip in the split build (-DFPL_SPLIT),
// inside its anonymous namespace.
//
// Why.  The round-3 / 4 kernels (conv3_f16s above: two 4-wave workgroups per CU, a 41 KiB tile of
// 16 real channels staged through registers, every wave streaming its own weight fragments from
// L2) sat at 50 - 55 % MFMA-busy with the CU's vector-memory path 70 - 90 % busy
// (profiles/r03_pmc_unet264_rows.json): the same disease vggs_mid_pool had.  Here:
//
//   * tensors are PLANAR: [pass of 8 channels][part hi | lo][tile n][z][y][x][8 halves], 16 B per
//     voxel, pass and part - a tile row is one contiguous run, a B fragment one ds_read_b128;
//   * ONE persistent 8-wave workgroup per CU; a block is 4 (z) x 2R (y) x 16 (x) outputs, wave =
//     (z plane wz, y half wy), R sub-steps per wave, lanes along x;
//   * a pass = 8 input channels x 27 taps = 7 K-steps (lane group g of K-step s holds tap 4 s + g),
//     three MFMAs per product (w_lo a_hi + w_hi a_lo + w_hi a_hi), weight fragments [hi MB | lo MB]
//     per K-step read from LDS;
//   * LDS = two tile buffers (pass p in buffer p & 1) + TWO HALF weight buffers: a pass runs as
//     phase A (K-steps 0 - 3 out of WA) and phase B (4 - 6 out of WB); during A the LDS-DMA of
//     the pass's own B weights and the first half of the NEXT pass's tile goes out, during B the
//     next pass's A weights and the rest of its tile.  Weights are thus single-buffered at pass
//     level (the 64-output layers' 56 KiB per pass would not fit twice) at the price of a second
//     barrier per pass;
//   * sources that are an UpSampling3D(2) (conv3 192->64's first 128 channels, the head's first 64)
//     are staged z-COMPRESSED: the tile holds the 3 low-resolution planes the block's 6 input
//     planes are copies of, the three z taps collapse to two with weights pre-summed per output
//     plane parity (conv_mfma.hip, "parity form"): 18 taps = 5 K-steps (A: 0 - 2, B: 3 - 4), both
//     parities' weights in LDS, a wave reads its own.  y and x stay full resolution (the DMA
//     gathers every source voxel into its 2 x 2 tile positions);
//   * STEM (unet_like2's conv3 1->32 in front of conv3 32->32): the 32-channel tile is not
//     fetched but computed from the raw f32 tile, two passes (16 channels, one M-block x three
//     MFMAs per 16 tile voxels) at a time into the two tile buffers;
//   * epilogues: planar hi / lo stores, MaxPooling3D(2), or the head (conv1 32->32, conv1 32->1,
//     sigmoid, store into the prediction volume), as in conv3_f16s.
//
// PLAIN 16-bit builds (bf16 / f16; SPLIT false: conv_mfma.hip's other two builds include this file
// too): the same kernels with a pass of SIXTEEN channels - plane 0 of a pass = its channels 0 - 7, plane
// 1 = channels 8 - 15 (where the split build keeps hi and lo halves), weight set 0 / 1 per K-step
// likewise - and two MFMAs per K-step, sub-step and M-block, (w0, a0) + (w1, a1), instead of three.
//
// The K loop's tile addressing is conflict-free as in vgg_split_lds.h: TX = 18 and a z-plane
// stride ZS = 6 mod 16 put consecutive taps 1, 16 or ZS - 38 slots apart, 0 or 1 modulo 16.
#pragma once

namespace u8 {

constexpr int WAVES = 8, WZ = 4, WY = 2;
constexpr int TX = 18, TZP = WZ + 2, TZU = WZ / 2 + 1;     // tile planes: plain source / z-compressed
constexpr int zs_for(int n) { return n % 16 <= 6 ? n - n % 16 + 6 : n - n % 16 + 22; }
constexpr int KP = 7, KPA = 4;                            // K-steps of a plain pass, of its phase A
constexpr int MAXPASS = 24;
constexpr int zsu_for(int n) { return n % 16 <= 4 ? n - n % 16 + 4 : n - n % 16 + 20; }
// Upsampled passes, by mode UM:
//   UM_Z  (1): the tile z-compressed, z weights pre-summed per output-plane parity: 18 taps (dz', dy,
//              dx) = 5 K-steps (phase A: 0 - 2), two weight streams (a wave reads its plane's parity);
//   UM_ZY (2): z- AND y-compressed, weights pre-summed per (plane, row) parity: 12 taps (dz', dy', dx)
//              = 3 K-steps (phase A: 0 - 1), four weight streams.  A wave must then work on rows of ONE
//              parity: wave (wz, wy) takes rows y0 + 2 sub + wy (all kernels of this mode, plain passes
//              included, where the row pitch of a sub-step becomes two tile rows).  The 64-output layers
//              stay on UM_Z: four streams of their fragments do not fit beside the tiles.
enum { UM_NONE = 0, UM_Z = 1, UM_ZY = 2 };
template <int UM> struct UK {
  static constexpr int K = UM == UM_ZY ? 3 : 5, KA = UM == UM_ZY ? 2 : 3, NPAR = UM == UM_ZY ? 4 : 2;
  static constexpr int NTAP = UM == UM_ZY ? 12 : 18;
};

template <int R, int UM = UM_Z> struct Geo8 {
  static constexpr int BY = WY * R, TY = BY + 2;
  static constexpr int ZS = zs_for(TY * TX);
  static constexpr int PLANE_P = (TZP - 1) * ZS + TY * TX;            // slots of one part, plain
  // upsampled tile: TZU low-resolution planes; UM_ZY: R + 1 low-resolution rows, plane stride 4 mod 16
  // (the plane wrap of its 2 x 2 x 3 taps is ZSU - TX - 2)
  static constexpr int TYU = UM == UM_ZY ? R + 1 : TY;
  static constexpr int ZSU = UM == UM_ZY ? zsu_for(TYU * TX) : ZS;
  static constexpr int PLANE_U = (TZU - 1) * ZSU + TYU * TX;
  static constexpr int NTP = (2 * PLANE_P + 63) / 64, NTU = (2 * PLANE_U + 63) / 64;   // 1 KiB chunks
  static constexpr int TILE_BYTES = NTP * 1024;
  static constexpr int TCHP = (NTP + WAVES - 1) / WAVES, TCHU = (NTU + WAVES - 1) / WAVES;
  // sub-step s of wave (wz, wy) is row wy * R + s of the block - or 2 s + wy (UM_ZY)
  static constexpr int SUBROW = UM == UM_ZY ? 2 : 1;
  // STEM: the raw f32 tile the 32-channel tile is computed from
  static constexpr int RZ = TZP + 2, RY = TY + 2, RX = TX + 2;
  static constexpr int NRAW = RZ * RY * RX, NRAWT = (NRAW + 64 * WAVES - 1) / (64 * WAVES);
  static constexpr int NGRP = (PLANE_P + 15) / 16;                    // groups of 16 tile slots
  static_assert(ZS % 16 == 6 && (TX - 2) % 16 == 0, "conflict-free tap offsets");
  static_assert(UM != UM_ZY || ZSU % 16 == 4, "conflict-free tap offsets (zy form)");
  static_assert(NTU <= NTP, "the tile buffers are sized for plain passes");
};

template <int MB, int UM = UM_Z> struct Wt {
  static constexpr int FR = 2 * MB;                                   // fragments per K-step: hi set, lo set
  static constexpr int PA = KPA * FR, PB = (KP - KPA) * FR;           // 1 KiB chunks of a plain pass: phase A, B
  static constexpr int UA = UK<UM>::NPAR * UK<UM>::KA * FR;           // upsampled pass: every parity's stream
  static constexpr int UB = UK<UM>::NPAR * (UK<UM>::K - UK<UM>::KA) * FR;
  static constexpr int PBYTES = (PA + PB) * 1024, UBYTES = (UA + UB) * 1024;
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }
template <int MB, int R, int UM, bool STEM, bool FRAGS = true> struct Lds {
  static constexpr int TB = Geo8<R, UM>::TILE_BYTES;
  static constexpr int WA = (UM ? cmax(Wt<MB, UM>::PA, Wt<MB, UM>::UA) : Wt<MB, UM>::PA) * 1024;
  static constexpr int WB = (UM ? cmax(Wt<MB, UM>::PB, Wt<MB, UM>::UB) : Wt<MB, UM>::PB) * 1024;
  static constexpr int OFF_WA = 2 * TB, OFF_WB = OFF_WA + WA, OFF_KTAB = OFF_WB + WB;
  // block-invariant operands (shift vector; stem / head fragments): kept in LDS, because a global
  // load inside the block loop waits - vmcnt retires in order - for every older vector-memory
  // operation of the wave: the epilogue's stores, the next block's raw-tile loads
  // (FRAGS: the stem's / head's fragments; a plain store kernel keeps its shift vector only)
  static constexpr int OFF_CONST = OFF_KTAB + 2 * 32 * 4, CONST_BYTES = FRAGS ? 256 + 6 * 1024 + 256 : 256;
  static constexpr int OFF_RAW = OFF_CONST + CONST_BYTES;             // STEM: the raw tile, [hi | lo << 16] per voxel
  static constexpr int OFF_ROFF = OFF_RAW + 2 * Geo8<R, UM>::NRAW * 2;    // STEM: raw offset of every tile slot (u16)
  static constexpr int BYTES = STEM ? OFF_ROFF + ((Geo8<R, UM>::NGRP * 16 * 2 + 15) / 16) * 16 : OFF_RAW;
  static_assert(BYTES <= 160 * 1024, "one workgroup per CU");
};

struct U3Args {
  // input passes (8 channels each): the hi plane of the pass at voxel (tile 0, crop, crop, crop)
  // of its source tensor; the first `nups` passes read an UpSampling3D(2) of their source
  const unsigned char *src[MAXPASS];
  int npass, nups;
  int PD, PH, PW; unsigned Ppart;          // plain sources: voxels per axis, bytes of one part plane
  int UD, UH, UW; unsigned Upart;          // upsampled sources (their own, low-resolution dims)
  const unsigned char *w;                  // weight stream (pack_u3)
  const float *shift;
  int relu;
  unsigned char *out; int64_t out_part;    // planar output: pass q at out + 2 q out_part
  int OD, OH, OW;
  int keep_lo, keep_hi;                    // full-resolution stores only for voxels in [keep_lo, keep_hi) per axis (the
                                           // tensor's one reader crops it: unet_like2's c1 is read through Cropping3D(6))
  unsigned char *pool; int64_t pool_part;  // EPI_POOL: the pooled tensor (OD/2, OH/2, OW/2)
  int n_tiles, nbx, nby, nbz;
  int xorg;                                // TRANSPOSED: the strip's first column
  // STEM
  const float *raw; int T;
  const h16x8 *wstem; const float *shstem; float xlim;
  // EPI_HEAD
  FplTileIO io;
  const h16x8 *w8, *w9; const float *sh8; float bias9;
  // EPI_F32 (training, conv_split_train): the output as fp32 channels-last (n, OD, OH, OW, opitch floats),
  // every value times *unscale (the exact power of two the operands were scaled by, inverted)
  float *out32; int opitch; const float *unscale;
  unsigned *flag;
  unsigned long long *dbgbuf;              // dbg & 32: per-workgroup cycle stamps [wg][16]
  int dbg;                                 // timing builds only (FPL_U3_DBG): 1 no tile DMA, 2 no weight DMA, 4 no stem fill
};

enum { EPI_STORE = 0, EPI_POOL = 1, EPI_HEAD = 2, EPI_F32 = 3 };

// per-lane byte offsets of a wave's tile chunks (chunk j = wave + 8 i holds slots 64 j .. 64 j + 63
// of [hi plane | lo plane]) from the block's origin voxel in the hi plane of the source
template <int R, int UM, bool UPS, bool TRANSPOSED, int TCH>
__device__ __forceinline__ void tile_off_init(unsigned (&off)[TCH], int wave, int lane, int H, int W, unsigned part_bytes) {
  typedef Geo8<R, UM> GE;
  constexpr int PLANE = UPS ? GE::PLANE_U : GE::PLANE_P;
  constexpr int ZSx = UPS ? GE::ZSU : GE::ZS, TYx = UPS ? GE::TYU : GE::TY;
#pragma unroll
  for (int i = 0; i < TCH; ++i) {
    int slot = 64 * (wave + WAVES * i) + lane;
    slot = slot < 2 * PLANE ? slot : 2 * PLANE - 1;         // the last chunk's tail re-reads the last slot
    const int part = slot >= PLANE;
    const int s = slot - part * PLANE;
    const int tz = s / ZSx;
    int rem = s - tz * ZSx;
    rem = rem < TYx * TX ? rem : TYx * TX - 1;              // padding slots: any valid voxel
    int ty = rem / TX, tx = rem - ty * TX;
    // upsampled source: tz counts low-resolution planes already; so does ty in the zy form
    if (UPS) { tx >>= 1; if (UM != UM_ZY) ty >>= 1; }
    const int sy = TRANSPOSED ? tx : ty, sx = TRANSPOSED ? ty : tx;
    off[i] = (unsigned)(((tz * H + sy) * W + sx) * 16) + (part ? part_bytes : 0u);
  }
}

// tap offset tables: entry [g][s] = byte offset of tap 4 s + g inside a part plane of the tile
// (27 taps (dz, dy, dx); the z-compressed form's 18 taps (dz', dy, dx) through the same formula)
template <int R, int UM>
__device__ __forceinline__ void ktab_init(unsigned *ktabP, unsigned *ktabU, int tid) {
  typedef Geo8<R, UM> GE;
  if (tid < 64) {
    const int u = tid >> 5, g = (tid >> 3) & 3, s = tid & 7;
    const int tap = 4 * s + g;
    unsigned v = 0u;
    if (!u) {
      if (tap < 27) v = (unsigned)(((tap / 9) * GE::ZS + ((tap / 3) % 3) * TX + tap % 3) * 16);
    } else if (UM == UM_ZY) {                         // taps (dz', dy', dx): 2 x 2 x 3
      if (tap < 12) v = (unsigned)(((tap / 6) * GE::ZSU + ((tap / 3) % 2) * TX + tap % 3) * 16);
    } else if (tap < 18) {                            // taps (dz', dy, dx): 2 x 3 x 3
      v = (unsigned)(((tap / 9) * GE::ZSU + ((tap / 3) % 3) * TX + tap % 3) * 16);
    }
    (u ? ktabU : ktabP)[8 * g + s] = v;
  }
}

// One pass out of `tile` (this pass's buffer): phase A = K-steps 0 .. KA - 1 on the weight fragments
// in `wa`, the workgroup barrier, phase B = K-steps KA .. K - 1 on `wb` (this wave's parity already
// applied to both).  issueA() / issueB() are called inside the first K-step of each phase: the
// caller's LDS-DMA goes there.
//
// A K-step is three groups of R * MB MFMAs - (w_lo, a_hi), (w_hi, a_lo), (w_hi, a_hi) - and
// everything else it issues sits between them, in the order it is consumed: behind group 1
// the next step's a_hi and w_lo (w_lo's registers are free then), behind group 2 its a_lo (likewise)
// and w_hi.  Only a_hi and w_hi are held twice; the tile fragments of phase B's first step are
// read BEFORE the barrier (the tile is there for the whole pass), so a phase change exposes the
// latency of 2 MB weight reads and nothing else.
template <int MB, int R, int UM, bool UPS, typename IssueA, typename IssueB>
__device__ __forceinline__ void pass_kloop(const unsigned char *tile, const unsigned char *wa, const unsigned char *wb,
                                           const unsigned *ktab_g, unsigned vb, int lane, f32x4 (&acc)[R][MB],
                                           IssueA issueA, IssueB issueB) {
  typedef Geo8<R, UM> GE;
  constexpr int PLANE = UPS ? GE::PLANE_U : GE::PLANE_P;
  constexpr int K = UPS ? UK<UM>::K : KP, KA = UPS ? UK<UM>::KA : KPA;
  // (zy form: a plain pass's sub-steps are two tile rows apart, a compressed tile's rows ARE the sub-steps)
  constexpr int FR = 2 * MB, ROW = (UPS ? 1 : GE::SUBROW) * TX * 16;
  const unsigned char *wla = wa + lane * 16, *wlb = wb + lane * 16;
  const u32x4 k0 = *reinterpret_cast<const u32x4 *>(ktab_g), k1 = *reinterpret_cast<const u32x4 *>(ktab_g + 4);
  auto koff = [&](int s) -> unsigned { return s < 4 ? k0[s & 3] : k1[s & 3]; };
  auto wfrag = [&](int s, int f) -> h16x8 {
    return *reinterpret_cast<const h16x8 *>(s < KA ? wla + (s * FR + f) * 1024 : wlb + ((s - KA) * FR + f) * 1024);
  };
  h16x8 whi[MB], wlo[MB], whin[MB], bhi[R], blo[R], bhin[R];
  {
    const unsigned char *p = tile + vb + koff(0);
#pragma unroll
    for (int b = 0; b < MB; ++b) wlo[b] = wfrag(0, MB + b);
#pragma unroll
    for (int sub = 0; sub < R; ++sub) (SPLIT ? bhi : blo)[sub] = *reinterpret_cast<const h16x8 *>(p + sub * ROW + (SPLIT ? 0 : PLANE * 16));
#pragma unroll
    for (int b = 0; b < MB; ++b) whi[b] = wfrag(0, b);
#pragma unroll
    for (int sub = 0; sub < R; ++sub) (SPLIT ? blo : bhi)[sub] = *reinterpret_cast<const h16x8 *>(p + sub * ROW + (SPLIT ? PLANE * 16 : 0));
  }
#pragma unroll
  for (int s = 0; s < K; ++s) {
    const bool more = s + 1 < K, wnext = more && s + 1 != KA;       // (weights of step KA: behind the barrier)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (SPLIT) {
#pragma unroll
      for (int sub = 0; sub < R; ++sub)
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[sub][b] = mfma16(wlo[b], bhi[sub], acc[sub][b]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) {                    // (plain build: in front of its first group)
      const unsigned char *p = tile + vb + koff(s + 1);
#pragma unroll
      for (int sub = 0; sub < R; ++sub) bhin[sub] = *reinterpret_cast<const h16x8 *>(p + sub * ROW);
    }
    if (SPLIT ? wnext : false) {
#pragma unroll
      for (int b = 0; b < MB; ++b) wlo[b] = wfrag(s + 1, MB + b);
    }
    if (!SPLIT && wnext) {
#pragma unroll
      for (int b = 0; b < MB; ++b) whin[b] = wfrag(s + 1, b);
    }
    __builtin_amdgcn_sched_barrier(0);
    // split: (w_hi, a_lo); plain: (w1, a1)
#pragma unroll
    for (int sub = 0; sub < R; ++sub)
#pragma unroll
      for (int b = 0; b < MB; ++b) acc[sub][b] = mfma16(SPLIT ? whi[b] : wlo[b], blo[sub], acc[sub][b]);
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      const unsigned char *p = tile + vb + koff(s + 1) + PLANE * 16;
#pragma unroll
      for (int sub = 0; sub < R; ++sub) blo[sub] = *reinterpret_cast<const h16x8 *>(p + sub * ROW);
    }
    if (wnext) {
      if constexpr (SPLIT) {
#pragma unroll
        for (int b = 0; b < MB; ++b) whin[b] = wfrag(s + 1, b);
      } else {
#pragma unroll
        for (int b = 0; b < MB; ++b) wlo[b] = wfrag(s + 1, MB + b);
      }
    }
    if (s == 0) issueA();
    if (s == KA) issueB();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sub = 0; sub < R; ++sub)
#pragma unroll
      for (int b = 0; b < MB; ++b) acc[sub][b] = mfma16(whi[b], bhi[sub], acc[sub][b]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
#pragma unroll
      for (int sub = 0; sub < R; ++sub) bhi[sub] = bhin[sub];
    }
    if (wnext) {
#pragma unroll
      for (int b = 0; b < MB; ++b) whi[b] = whin[b];
    }
    if (s + 1 == KA) {
      __syncthreads();             // phase B's weights have landed, every wave has left WA
#pragma unroll
      for (int b = 0; b < MB; ++b) wlo[b] = wfrag(KA, MB + b);
#pragma unroll
      for (int b = 0; b < MB; ++b) whi[b] = wfrag(KA, b);
    }
  }
}

// a pointer the compiler may not have proved wave-uniform, as a scalar pair: the LDS-DMA then
// takes the form (scalar base + 32-bit lane offset) instead of a 64-bit address per lane
__device__ __forceinline__ const unsigned char *uniform_ptr(const unsigned char *p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const unsigned char *>(((uint64_t)hi << 32) | lo);
}

// a workgroup barrier for LDS traffic only: waits for this wave's LDS operations, NOT for its
// vector-memory ones.  __syncthreads() drains vmcnt as well - it must, where an LDS-DMA has to have
// landed - and behind an epilogue that means waiting for the block's stores to reach memory
// (the stem kernel writes 192 KiB per block: 3.8 of its 30 ms were that wait).
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// LDS-DMA of 64 x 16 B from (scalar base + 32-bit lane offset): the pointer arithmetic is done in
// the global address space so that the selector sees base + zext(offset)
__device__ __forceinline__ void glds16_so(const unsigned char *sbase, unsigned voff, void *l) {
  typedef const __attribute__((address_space(1))) unsigned char *gptr;
  // (the opaque offset keeps the address sum in the DMA's own basic block: hoisted in front of the
  // wave-uniform branch around a chunk it becomes a 64-bit VGPR pair per lane and the scalar form is lost)
  asm volatile("" : "+v"(voff));
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((gptr)sbase + voff),
                                   (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// `n` weight chunks of 1 KiB, contiguous at `wsrc`, into `dst`: chunk j = wave + 8 i
template <int NMAX>
__device__ __forceinline__ void dma_weights(const unsigned char *wsrc, int n, unsigned char *dst, int wave, int lane) {
  const unsigned char *base = uniform_ptr(wsrc + (size_t)wave * 1024);
  const unsigned loff = (unsigned)lane * 16u;
#pragma unroll
  for (int i = 0; i < (NMAX + WAVES - 1) / WAVES; ++i) {
    const int j = wave + WAVES * i;
    if (j < n) glds16_so(base + (size_t)i * (WAVES * 1024), loff, dst + j * 1024);
  }
}
// tile chunks i = I0 .. I1 - 1 of this wave (chunk j = wave + 8 i, j < NT) from `org`
template <int TCH, int I0, int I1, int NT>
__device__ __forceinline__ void dma_tile(const unsigned (&off)[TCH], const unsigned char *org_, unsigned char *dst, int wave) {
  const unsigned char *org = uniform_ptr(org_);
#pragma unroll
  for (int i = I0; i < I1; ++i) {
    const int j = wave + WAVES * i;
    if (WAVES * i + WAVES - 1 < NT || j < NT) glds16_so(org, off[i], dst + j * 1024);
  }
}

// planar store of a lane's 4 MB contiguous channels [4 MB g, 4 MB (g + 1)) of one voxel (interleaved
// weight rows, pack_weights.h::fpl_out_channel) as MB / 2 pieces of 8 channels.  Split build: piece i
// (counted over the tensor's channels) is pass i, its hi and lo halves 16 B each; plain builds: plane
// i & 1 of pass i >> 1, 16 B.  `vox` = the voxel's index in a part plane; pass q of the tensor at
// p + 2 q part.
template <int MB>
__device__ __forceinline__ void store_planar(unsigned char *p, int64_t part, int64_t vox, int g, const f32x4 (&v)[MB], bool relu,
                                             unsigned &ovf) {
#pragma unroll
  for (int h = 0; h < MB / 2; ++h) {
    const int piece = (MB / 2) * g + h;
    if constexpr (SPLIT) {
      u32x4 hi, lo;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 &x = v[2 * h + q];
        const Pair2 p0 = relu ? split_pk_relu(x[0], x[1], ovf) : split_pk_signed(x[0], x[1], ovf);
        const Pair2 p1 = relu ? split_pk_relu(x[2], x[3], ovf) : split_pk_signed(x[2], x[3], ovf);
        hi[2 * q] = p0.hi; hi[2 * q + 1] = p1.hi;
        lo[2 * q] = p0.lo; lo[2 * q + 1] = p1.lo;
      }
      unsigned char *d = p + (int64_t)piece * 2 * part + vox * 16;
      *reinterpret_cast<u32x4 *>(d) = hi;
      *reinterpret_cast<u32x4 *>(d + part) = lo;
    } else {
      u32x4 o;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        o[2 * q] = cvt_pk_h16(v[2 * h + q][0], v[2 * h + q][1]);
        o[2 * q + 1] = cvt_pk_h16(v[2 * h + q][2], v[2 * h + q][3]);
      }
      if (relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pk_max_i16(o[q], 0u);
      }
      *reinterpret_cast<u32x4 *>(p + (int64_t)piece * part + vox * 16) = o;
    }
  }
}

template <int MB, int R, int UM, int EPI, bool STEM, bool TRANSPOSED>
__global__ __launch_bounds__(64 * WAVES, 2) void FPLK(u3conv)(U3Args a) {
  constexpr bool HAS_UPS = UM != UM_NONE;
  static_assert(!STEM || (MB == 2 && !HAS_UPS && !TRANSPOSED), "the stem variant is conv3 32->32");
  static_assert(UM != UM_ZY || EPI != EPI_POOL, "the zy form deals rows to waves by parity: no y pairs in a lane");
  static_assert(EPI != EPI_HEAD || MB == 2, "the head variant is conv3 ->32");
  static_assert(EPI != EPI_POOL || R % 2 == 0, "pool pairs");
  typedef Geo8<R, UM> GE;
  typedef Wt<MB, UM> WT;
  typedef Lds<MB, R, UM, STEM, STEM || EPI == EPI_HEAD> L;
  typedef UK<UM> UKx;
  constexpr int TB = L::TB, TCHP = GE::TCHP, TCHU = GE::TCHU;
  unsigned char *WAb = smem + L::OFF_WA, *WBb = smem + L::OFF_WB;
  unsigned *ktabP = reinterpret_cast<unsigned *>(smem + L::OFF_KTAB), *ktabU = ktabP + 32;
  unsigned short *rawoff = reinterpret_cast<unsigned short *>(smem + L::OFF_ROFF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wz = wave & 3, wy = wave >> 2;
  const int c = lane & 15, g = lane >> 4;
  ktab_init<R, UM>(ktabP, ktabU, tid);
  // constants: [shift 64 f32][6 fragments: stem [q][part] or head w8 [part][b], w9 [part]][sh 64 f32]
  float *shiftL = reinterpret_cast<float *>(smem + L::OFF_CONST);
  h16x8 *fragL = reinterpret_cast<h16x8 *>(smem + L::OFF_CONST + 256);
  float *sh2L = reinterpret_cast<float *>(smem + L::OFF_CONST + 256 + 6 * 1024);
  if (tid < 16 * MB) shiftL[tid] = a.shift[tid];
  // (fragments per operand: the split build's come as hi and lo parts, PM = 2)
  if constexpr (STEM) {
    if (tid < 128 * PM) fragL[tid] = a.wstem[tid];
    if (tid < 32) sh2L[tid] = a.shstem[tid];
  }
  if constexpr (EPI == EPI_HEAD) {
    if (tid < 128 * PM) fragL[tid] = a.w8[tid];
    else if (tid < 192 * PM) fragL[tid] = a.w9[tid - 128 * PM];
    if (tid < 32) sh2L[tid] = a.sh8[tid];
  }
  unsigned offP[STEM ? 1 : TCHP], offU[HAS_UPS ? TCHU : 1];
  if constexpr (!STEM) tile_off_init<R, UM, false, TRANSPOSED, TCHP>(offP, wave, lane, a.PH, a.PW, a.Ppart);
  if constexpr (HAS_UPS) tile_off_init<R, UM, true, TRANSPOSED, TCHU>(offU, wave, lane, a.UH, a.UW, a.Upart);

  // ---- the walk: blocks numbered x fastest, then y, z, tile; group = blockIdx & 7 (one XCD under
  // round-robin placement) takes a contiguous range, its workgroups every S-th block of it
  const int S = (int)gridDim.x >> 3, group = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int per_tile = a.nbx * a.nby * a.nbz;
  const int total = per_tile * a.n_tiles, per = (total + 7) / 8;
  int bi = group * per + slot;
  const int bend = (group + 1) * per < total ? (group + 1) * per : total;
  if (bi >= bend) return;
  struct Blk { int n, z0, y0, x0; };
  auto decode = [&](int i) {
    Blk b;
    b.n = i / per_tile;
    int r = i - b.n * per_tile;
    const int bz = r / (a.nbx * a.nby);
    r -= bz * a.nbx * a.nby;
    const int by = r / a.nbx, bx = r - by * a.nbx;
    b.z0 = WZ * bz;
    // TRANSPOSED: lanes walk y (blocks of 16 along y), sub-steps walk x from xorg
    b.y0 = TRANSPOSED ? 16 * bx : GE::BY * by;
    b.x0 = TRANSPOSED ? a.xorg + GE::BY * by : 16 * bx;
    return b;
  };
  // byte offset of the block's origin voxel inside a pass plane of the plain / upsampled source
  auto boffP = [&](const Blk &b) -> int64_t { return ((((int64_t)b.n * a.PD + b.z0) * a.PH + b.y0) * a.PW + b.x0) * 16; };
  auto boffU = [&](const Blk &b) -> int64_t {
    return ((((int64_t)b.n * a.UD + (b.z0 >> 1)) * a.UH + (b.y0 >> 1)) * a.UW + (b.x0 >> 1)) * 16;
  };
  auto wpass = [&](int p) -> const unsigned char * {
    return a.w + (p < a.nups ? (size_t)p * WT::UBYTES : (size_t)a.nups * WT::UBYTES + (size_t)(p - a.nups) * WT::PBYTES);
  };
  // the LDS-DMA that goes out during phase `half` of a pass whose successor is pass `pn` of block
  // `bn` (same block: pn = p + 1; else pass 0 of the next block): tile of pn into `tdst`
  auto dma_next_tile = [&](int half, int pn, const Blk &bn, unsigned char *tdst) {
    if (a.dbg & 1) return;
    if constexpr (!STEM) {
      if constexpr (HAS_UPS) {
        if (pn < a.nups) {
          const unsigned char *org = a.src[pn] + boffU(bn);
          if (half == 0) dma_tile<TCHU, 0, TCHU / 2, GE::NTU>(offU, org, tdst, wave);
          else dma_tile<TCHU, TCHU / 2, TCHU, GE::NTU>(offU, org, tdst, wave);
          return;
        }
      }
      const unsigned char *org = a.src[pn] + boffP(bn);
      if (half == 0) dma_tile<TCHP, 0, TCHP / 2, GE::NTP>(offP, org, tdst, wave);
      else dma_tile<TCHP, TCHP / 2, TCHP, GE::NTP>(offP, org, tdst, wave);
    }
  };

  // lane's voxel of sub-step 0 in the hi plane of a tile (plain / compressed); the wave's first row
  // of the block and its weight stream among an upsampled pass's parities
  constexpr bool ZY = UM == UM_ZY;
  const int row0 = ZY ? wy : wy * R;
  const unsigned vbP = (unsigned)(((wz * GE::ZS) + row0 * TX + c) * 16);
  const unsigned vbU = (unsigned)((((wz >> 1) * GE::ZSU) + (ZY ? 0 : row0) * TX + c) * 16);
  const int parity = (wz & 1) + (ZY ? 2 * wy : 0);

  // ---- STEM: conv3 1->32 of the raw tile, 16 channels (two passes) at a time.  The raw tile sits in
  // LDS as ONE word per voxel, [hi half | lo half << 16].  K-slots (stem_slot_tap, conv_mfma.hip):
  // lane groups 0 - 2 hold the (dx = 0, 1) pairs of rows 3 g .. 3 g + 2 (a row = (dz, dy)) and the
  // dx = 2 singles of rows 2 g, 2 g + 1, group 3 the dx = 2 singles of rows 6 - 8: a lane's 8 values
  // are three ds_read2_b32 (x, x + 1: any alignment) and two ds_read_b32 - 5 LDS instructions where
  // 16 ds_read_u16 were (narrow reads run at a fraction of the LDS rate with two waves per SIMD, and
  // the gather loop was bound by their number: 9.9 of the kernel's 30 ms).
  float rawv[STEM ? GE::NRAWT : 1];
  unsigned gp[3], gs[2];                          // element offsets: pair bases, singles
  unsigned xmax = 0u;
  unsigned *rawt = reinterpret_cast<unsigned *>(smem + L::OFF_RAW);
  auto fetch_raw = [&](const Blk &b) {
    const float *base = a.raw + (int64_t)b.n * a.T * a.T * a.T;
#pragma unroll
    for (int j = 0; j < GE::NRAWT; ++j) {
      const int p = min(tid + 64 * WAVES * j, GE::NRAW - 1);
      int z = b.z0 + p / (GE::RY * GE::RX), y = b.y0 + (p / GE::RX) % GE::RY, x = b.x0 + p % GE::RX;
      z = z < a.T ? z : a.T - 1;                  // clamped reads only feed masked outputs
      y = y < a.T ? y : a.T - 1;
      x = x < a.T ? x : a.T - 1;
      rawv[j] = base[((int64_t)z * a.T + y) * a.T + x];
    }
  };
  auto put_raw = [&]() {
#pragma unroll
    for (int j = 0; j < GE::NRAWT; ++j) {
      const float x = rawv[j];
      const unsigned ax = __builtin_bit_cast(unsigned, x) & 0x7FFFFFFFu;
      xmax = ax > xmax ? ax : xmax;
      const h16_t h = (h16_t)x;
      if (tid + 64 * WAVES * j < GE::NRAW)
        rawt[tid + 64 * WAVES * j] = (unsigned)h16_bits(x) | (SPLIT ? (unsigned)h16_bits(x - (float)h) << 16 : 0u);
    }
  };
  // passes 2 q and 2 q + 1 of the 32-channel tile into the two tile buffers: plain weight rows, so
  // lane (c, g) holds channels 16 q + 4 g .. + 3 of its voxel: 8 B of pass 2 q + (g >> 1).
  // Software-pipelined: the gathers of group i + 1 and the table entry of group i + 2 are in
  // flight under the MFMAs and the hi / lo conversion of group i.
  auto gather = [&](unsigned ro, Frag2 &bf) {
    unsigned w[8];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const unsigned *q = rawt + ro + gp[i];
      w[2 * i] = q[0];
      w[2 * i + 1] = q[1];
    }
    w[6] = rawt[ro + gs[0]];
    w[7] = rawt[ro + gs[1]];
    u32x4 hi, lo;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      hi[d] = __builtin_amdgcn_perm(w[2 * d + 1], w[2 * d], 0x05040100u);
      if (SPLIT) lo[d] = __builtin_amdgcn_perm(w[2 * d + 1], w[2 * d], 0x07060302u);
    }
    bf.hi = __builtin_bit_cast(h16x8, hi);
    if (SPLIT) bf.lo = __builtin_bit_cast(h16x8, lo);
  };
  auto tab = [&](int grp) -> unsigned { return rawoff[16 * (grp < GE::NGRP ? grp : GE::NGRP - 1) + c]; };
  // split build: fill(q) = passes 2 q and 2 q + 1 (16 channels: one M-block, three MFMAs per group)
  auto fill = [&](auto q_) {             // (generic: only the build that calls it compiles it)
    const int q = q_;
    const h16x8 wh = fragL[(q * 2 + 0) * 64 + lane], wl = fragL[(q * 2 + 1) * 64 + lane];
    const f32x4 sh = *reinterpret_cast<const f32x4 *>(sh2L + 16 * q + 4 * g);
    unsigned char *dst = smem + (g >> 1) * TB + 8 * (g & 1);
    Frag2 bn;
    gather(tab(wave), bn);
    unsigned ro_nn = tab(wave + WAVES);
    for (int grp = wave; grp < GE::NGRP; grp += WAVES) {
      const Frag2 bc = bn;
      const int v = 16 * grp + c;
      gather(ro_nn, bn);                          // (past the last group: a harmless re-read)
      ro_nn = tab(grp + 2 * WAVES);
      const f32x4 a0 = mfma3(wh, wl, bc, sh);
      const Pair2 p0 = split_pk_relu(a0[0], a0[1]), p1 = split_pk_relu(a0[2], a0[3]);
      if (v < GE::PLANE_P) {
        *reinterpret_cast<u32x2 *>(dst + v * 16) = u32x2{p0.hi, p1.hi};
        *reinterpret_cast<u32x2 *>(dst + v * 16 + GE::PLANE_P * 16) = u32x2{p0.lo, p1.lo};
      }
    }
  };
  // plain builds: both passes (pass q = channels 16 q .. 16 q + 15 -> buffer q) from ONE gather:
  // lane (c, g) holds channels 16 q + 4 g .. + 3: 8 B of plane g >> 1
  auto fill_plain = [&](auto) {
    const h16x8 w0 = fragL[lane], w1 = fragL[64 + lane];
    const f32x4 sh0 = *reinterpret_cast<const f32x4 *>(sh2L + 4 * g), sh1 = *reinterpret_cast<const f32x4 *>(sh2L + 16 + 4 * g);
    unsigned char *dst = smem + (g >> 1) * (GE::PLANE_P * 16) + 8 * (g & 1);
    Frag2 bn;
    gather(tab(wave), bn);
    unsigned ro_nn = tab(wave + WAVES);
    for (int grp = wave; grp < GE::NGRP; grp += WAVES) {
      const h16x8 bc = bn.hi;
      const int v = 16 * grp + c;
      gather(ro_nn, bn);
      ro_nn = tab(grp + 2 * WAVES);
      const f32x4 a0 = mfma16(w0, bc, sh0), a1 = mfma16(w1, bc, sh1);
      if (v < GE::PLANE_P) {
        *reinterpret_cast<u32x2 *>(dst + v * 16) =
            u32x2{pk_max_i16(cvt_pk_h16(a0[0], a0[1]), 0u), pk_max_i16(cvt_pk_h16(a0[2], a0[3]), 0u)};
        *reinterpret_cast<u32x2 *>(dst + TB + v * 16) =
            u32x2{pk_max_i16(cvt_pk_h16(a1[0], a1[1]), 0u), pk_max_i16(cvt_pk_h16(a1[2], a1[3]), 0u)};
      }
    }
  };
  if constexpr (STEM) {
    auto rowoff = [](int r) { return ((r / 3) * GE::RY + r % 3) * GE::RX; };
#pragma unroll
    for (int i = 0; i < 3; ++i) gp[i] = (unsigned)(g < 3 ? rowoff(3 * g + i) : rowoff(6 + i) + 2);
#pragma unroll
    for (int k = 0; k < 2; ++k) gs[k] = (unsigned)(g < 3 ? rowoff(2 * g + k) + 2 : 0);
    for (int i = tid; i < GE::NGRP * 16; i += 64 * WAVES) {
      const int tz = i / GE::ZS;
      int rem = i - tz * GE::ZS;
      rem = rem < GE::TY * TX ? rem : GE::TY * TX - 1;
      const int tzc = tz < TZP ? tz : TZP - 1;
      rawoff[i] = (unsigned short)((tzc * GE::RY + rem / TX) * GE::RX + rem % TX);
    }
  }

  Blk cur = decode(bi);
  // ---- prime: pass 0 of the first block (tile into buffer 0, phase-A weights into WA)
  if constexpr (STEM) fetch_raw(cur);
  dma_next_tile(0, 0, cur, smem);
  dma_next_tile(1, 0, cur, smem);
  if (HAS_UPS && a.nups > 0) dma_weights<WT::UA>(wpass(0), WT::UA, WAb, wave, lane);
  else dma_weights<WT::PA>(wpass(0), WT::PA, WAb, wave, lane);
  __syncthreads();

  unsigned ovf_all = 0u;
  // dbg & 32 (diagnostic run): wave 0 sums the cycles between stamps: [0] block start .. raw tile put,
  // [1] fills, [2] K loops, [3] epilogue, [4] blocks
  unsigned long long tsum[5] = {0, 0, 0, 0, 0}, tlast = 0;
  auto stamp = [&](int k) {
#ifdef FPL_U3_STAMPS              // (a diagnostic build: the sums cost a dozen scalar registers)
    if (a.dbg & 32) {
      const unsigned long long t = __builtin_readcyclecounter();
      if (k >= 0) tsum[k] += t - tlast;
      tlast = t;
    }
#endif
  };
  stamp(-1);
  for (;;) {
    const int bnext = bi + S;
    const bool has_next = bnext < bend;
    const Blk nxt = has_next ? decode(bnext) : cur;
    f32x4 acc[R][MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      // (EPI_F32: the operands are scaled, the bias is not: it is added behind the un-scaling)
      const f32x4 sh = EPI == EPI_F32 ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4 *>(shiftL + 4 * MB * g + 4 * b);
#pragma unroll
      for (int sub = 0; sub < R; ++sub) acc[sub][b] = sh;
    }
    if constexpr (STEM) {
      put_raw();
      lds_barrier();
    }
    stamp(0);
    // one pair of passes (2 pp in buffer 0, 2 pp + 1 in buffer 1), both of type UPS (compile time);
    // the upsampled pairs and the plain pairs run in two loops of their own, so that each loop holds
    // ONE instance of the K loop (both in one loop body cost ~60 registers more than either)
    auto pair = [&](auto ups_tag, int pp) {
      constexpr bool UPS = decltype(ups_tag)::value;
      if constexpr (STEM) {
        stamp(2);
        if (!(a.dbg & 4)) {
          if constexpr (SPLIT) fill(pp);
          else fill_plain(0);
        }
        lds_barrier();
        stamp(1);
        // the next block's raw tile: requested here, behind the block's last fill - a wave waits
        // for these loads (in-order vmcnt) at the next barrier that covers an LDS-DMA, two K-loop
        // phases on; issued in front of a fill they held up its first MFMA by a trip to HBM
        if (pp == a.npass / 2 - 1) fetch_raw(nxt);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int p = 2 * pp + h;
        unsigned char *tile = smem + h * TB, *other = smem + (h ^ 1) * TB;
        const bool more = p + 1 < a.npass;
        const int pn = more ? p + 1 : 0;
        const Blk &bn = more ? cur : nxt;
        const unsigned char *wp = wpass(p), *wn = wpass(pn);
        const bool ups_n = HAS_UPS && pn < a.nups;
        // phase A: this pass's B weights + the first half of the next tile go out
        auto issueA = [&]() {
          __builtin_amdgcn_s_setprio(0);
          if (a.dbg & 2) {}
          else if (UPS) dma_weights<WT::UB>(wp + WT::UA * 1024, WT::UB, WBb, wave, lane);
          else dma_weights<WT::PB>(wp + WT::PA * 1024, WT::PB, WBb, wave, lane);
          dma_next_tile(0, pn, bn, other);
          __builtin_amdgcn_s_setprio(1);
        };
        // phase B: the next pass's A weights + the rest of its tile
        auto issueB = [&]() {
          __builtin_amdgcn_s_setprio(0);
          if (a.dbg & 2) {}
          else if (ups_n) dma_weights<WT::UA>(wn, WT::UA, WAb, wave, lane);
          else dma_weights<WT::PA>(wn, WT::PA, WAb, wave, lane);
          dma_next_tile(1, pn, bn, other);
          __builtin_amdgcn_s_setprio(1);
        };
        if constexpr (UPS)
          pass_kloop<MB, R, UM, true>(tile, WAb + parity * (UKx::KA * WT::FR * 1024),
                                      WBb + parity * ((UKx::K - UKx::KA) * WT::FR * 1024), ktabU + 8 * g, vbU, lane, acc,
                                      issueA, issueB);
        else
          pass_kloop<MB, R, UM, false>(tile, WAb, WBb, ktabP + 8 * g, vbP, lane, acc, issueA, issueB);
        if (!(a.dbg & 16)) __syncthreads();
      }
    };
    int pp = 0;
    if constexpr (HAS_UPS) {
#pragma unroll 1
      for (; pp < a.nups / 2; ++pp) pair(std::true_type{}, pp);
    }
#pragma unroll 1
    for (; pp < a.npass / 2; ++pp) pair(std::false_type{}, pp);
    // ---- epilogue
    stamp(2);
    unsigned ovf = 0u;
    if (a.dbg & 8) {                 // timing build: no epilogue (the accumulators are kept alive)
      if (acc[0][0][0] == 123.456f) a.flag[0] = 1u;
      if (!has_next) break;
      cur = nxt;
      bi = bnext;
      continue;
    }
    const int oz = cur.z0 + wz;
    const bool zin = oz < a.OD;
    if constexpr (EPI == EPI_HEAD) {
      // conv1 32->32 (+shift, ReLU), conv1 32->1 (+bias), sigmoid, store.  With interleaved rows lane
      // (c, g) holds channels 8 g .. 8 g + 7 of voxel c: the packed pair IS the K-step of conv1 32->32
      // in SLOT_SPATIAL order.  The head's fragments are loaded once per block (inside the sub-step
      // loop the store into the volume makes the compiler reload them every time) and the sub-steps go
      // through the chain in STAGES, R independent MFMA chains abreast instead of R dependent ones
      // in a row: the matrix pipe sees 6 R + 3 R MFMAs nearly back to back.
      f32x4 sh8[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) sh8[b] = *reinterpret_cast<const f32x4 *>(sh2L + 16 * b + 4 * g);
      const FplTileDesc td = a.io.tiles[cur.n];
      unsigned ovs[R];
      f32x4 t9[R];
      if constexpr (SPLIT) {
        h16x8 f8[4], f9[2];            // [part][b], [part]
#pragma unroll
        for (int f = 0; f < 4; ++f) f8[f] = fragL[f * 64 + lane];
#pragma unroll
        for (int f = 0; f < 2; ++f) f9[f] = fragL[(4 + f) * 64 + lane];
        Frag2 h7[R];
#pragma unroll
        for (int sub = 0; sub < R; ++sub) {
          ovs[sub] = 0u;
          h7[sub] = pack_relu_split(acc[sub][0], acc[sub][1], ovs[sub]);
        }
        f32x4 a8[R][2];
#pragma unroll
        for (int sub = 0; sub < R; ++sub)
#pragma unroll
          for (int b = 0; b < 2; ++b) a8[sub][b] = mfma16(f8[2 + b], h7[sub].hi, sh8[b]);
#pragma unroll
        for (int sub = 0; sub < R; ++sub)
#pragma unroll
          for (int b = 0; b < 2; ++b) a8[sub][b] = mfma16(f8[b], h7[sub].lo, a8[sub][b]);
#pragma unroll
        for (int sub = 0; sub < R; ++sub)
#pragma unroll
          for (int b = 0; b < 2; ++b) a8[sub][b] = mfma16(f8[b], h7[sub].hi, a8[sub][b]);
#pragma unroll
        for (int sub = 0; sub < R; ++sub) {
          h7[sub] = pack_relu_split(a8[sub][0], a8[sub][1], ovs[sub]);
          t9[sub] = mfma16(f9[1], h7[sub].hi, f32x4{0.f, 0.f, 0.f, 0.f});
        }
#pragma unroll
        for (int sub = 0; sub < R; ++sub) t9[sub] = mfma16(f9[0], h7[sub].lo, t9[sub]);
#pragma unroll
        for (int sub = 0; sub < R; ++sub) t9[sub] = mfma16(f9[0], h7[sub].hi, t9[sub]);
      } else {
        h16x8 f8[2], f9;               // [b]; one fragment
#pragma unroll
        for (int f = 0; f < 2; ++f) f8[f] = fragL[f * 64 + lane];
        f9 = fragL[2 * 64 + lane];
        f32x4 a8[R][2];
#pragma unroll
        for (int sub = 0; sub < R; ++sub) {
          ovs[sub] = 0u;
          const h16x8 h7 = pack_relu(acc[sub][0], acc[sub][1]);
#pragma unroll
          for (int b = 0; b < 2; ++b) a8[sub][b] = mfma16(f8[b], h7, sh8[b]);
        }
#pragma unroll
        for (int sub = 0; sub < R; ++sub) t9[sub] = mfma16(f9, pack_relu(a8[sub][0], a8[sub][1]), f32x4{0.f, 0.f, 0.f, 0.f});
      }
#pragma unroll
      for (int sub = 0; sub < R; ++sub) {
        const int oy = TRANSPOSED ? cur.y0 + c : cur.y0 + row0 + GE::SUBROW * sub;
        const int ox = TRANSPOSED ? cur.x0 + row0 + GE::SUBROW * sub : cur.x0 + c;
        const bool in = zin && oy < a.OH && ox < a.OW;
        // (the guard counts voxels of the layer only: past OD / OH / OW the tile held the source's
        // slack and the result is never stored)
        ovf = pk_max_i16(ovf, in ? ovs[sub] : 0u);
        const float logit = __shfl(t9[sub][0], c) + a.bias9;     // lane (c, g = 0) register 0
        if (g == 0 && in && oz < td.ext[0] - 2 * a.io.off && oy < td.ext[1] - 2 * a.io.off && ox < td.ext[2] - 2 * a.io.off)
          a.io.dst[((int64_t)(td.start[0] + a.io.off + oz - a.io.dst_z_base) * a.io.Y + td.start[1] + a.io.off + oy) * a.io.X +
                   td.start[2] + a.io.off + ox] = 1.f / (1.f + __expf(-logit));
      }
    } else if constexpr (EPI == EPI_F32) {
      // interleaved rows: lane (c, g) holds the 4 MB contiguous channels [4 MB g, 4 MB (g + 1)) of voxel
      // c - MB stores of 16 B, a wave's 16 voxels whole 64 MB-byte rows
      const float us = *a.unscale;
#pragma unroll
      for (int sub = 0; sub < R; ++sub) {
        const int oy = cur.y0 + row0 + GE::SUBROW * sub, ox = cur.x0 + c;
        if (zin && oy < a.OH && ox < a.OW) {
          float *d = a.out32 + ((((int64_t)cur.n * a.OD + oz) * a.OH + oy) * a.OW + ox) * a.opitch + 4 * MB * g;
#pragma unroll
          for (int b = 0; b < MB; ++b) {
            f32x4 v = acc[sub][b];
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(shiftL + 4 * MB * g + 4 * b);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = v[r] * us + bias[r];
              if (a.relu) v[r] = __builtin_fmaxf(v[r], 0.f);
            }
            *reinterpret_cast<f32x4 *>(d + 4 * b) = v;
          }
        }
      }
    } else {
#pragma unroll
      for (int sub = 0; sub < R; ++sub) {
        const int oy = TRANSPOSED ? cur.y0 + c : cur.y0 + row0 + GE::SUBROW * sub;
        const int ox = TRANSPOSED ? cur.x0 + row0 + GE::SUBROW * sub : cur.x0 + c;
        if (zin && oy < a.OH && ox < a.OW && oz >= a.keep_lo && oy >= a.keep_lo && ox >= a.keep_lo && oz < a.keep_hi &&
            oy < a.keep_hi && ox < a.keep_hi)
          store_planar<MB>(a.out, a.out_part, (((int64_t)cur.n * a.OD + oz) * a.OH + oy) * a.OW + ox, g, acc[sub], a.relu != 0, ovf);
      }
    }
    if constexpr (EPI == EPI_POOL) {
      // MaxPooling3D(2) of the block (4 x 2R x 16 -> 2 x R x 8) in fp32: y pairs are sub-steps of a
      // lane, x pairs neighbouring lanes, z pairs neighbouring waves (wz ^ 1) through the tile
      // buffer the last pass did not use... both tile buffers may be in flight (the next block's
      // pass 0 is landing in buffer 0): the exchange sits in buffer 1, which the next DMA - issued
      // in the next block's first phase - only touches after the second barrier below
      f32x4 pm[R / 2][MB];
#pragma unroll
      for (int yh = 0; yh < R / 2; ++yh)
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = __builtin_fmaxf(acc[2 * yh][b][r], acc[2 * yh + 1][b][r]);
            pm[yh][b][r] = __builtin_fmaxf(v, __shfl_xor(v, 1));
          }
      f32x4 *xch = reinterpret_cast<f32x4 *>(smem + TB);      // [wave pair (wz >> 1, wy)][yh][b][lane]
      const int pair = (wz >> 1) + 2 * wy;
      if (wz & 1) {
#pragma unroll
        for (int yh = 0; yh < R / 2; ++yh)
#pragma unroll
          for (int b = 0; b < MB; ++b) xch[((pair * (R / 2) + yh) * MB + b) * 64 + lane] = pm[yh][b];
      }
      lds_barrier();
      f32x4 m[R / 2][MB];
      if (!(wz & 1)) {
#pragma unroll
        for (int yh = 0; yh < R / 2; ++yh)
#pragma unroll
          for (int b = 0; b < MB; ++b) {
            const f32x4 o = xch[((pair * (R / 2) + yh) * MB + b) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) m[yh][b][r] = __builtin_fmaxf(pm[yh][b][r], o[r]);
          }
      }
      lds_barrier();
      if (!(wz & 1) && !(c & 1)) {
        const int PD = a.OD / 2, PH = a.OH / 2, PW = a.OW / 2;
        const int pz = (cur.z0 >> 1) + (wz >> 1), px = (cur.x0 >> 1) + (c >> 1);
#pragma unroll
        for (int yh = 0; yh < R / 2; ++yh) {
          const int py = ((cur.y0 + wy * R) >> 1) + yh;
          if (pz < PD && py < PH && px < PW)
            store_planar<MB>(a.pool, a.pool_part, (((int64_t)cur.n * PD + pz) * PH + py) * PW + px, g, m[yh], true, ovf);
        }
      }
    }
    ovf_all = pk_max_i16(ovf_all, ovf);
    stamp(3);
#ifdef FPL_U3_STAMPS
    tsum[4] += 1;
#endif
    if (!has_next) break;
    cur = nxt;
    bi = bnext;
  }
#ifdef FPL_U3_STAMPS
  if ((a.dbg & 32) && tid == 0)
    for (int k = 0; k < 5; ++k) a.dbgbuf[(size_t)blockIdx.x * 8 + k] = tsum[k];
#endif
  if constexpr (SPLIT) {
    if (!a.dbg) ovf_commit(ovf_all, a.flag, FPL_RANGE_UNET);
    if (STEM && xmax > __builtin_bit_cast(unsigned, a.xlim)) atomicOr(a.flag, FPL_RANGE_INPUT);
  }
}

// ---- 1x1x1 convolution on planar tensors: a K-step = 32 real channels = passes 4 s .. 4 s + 3
// (lane group g reads pass 4 s + g: one 16-B load per part), three MFMAs per product; weight
// fragments [s][hi MB | lo MB] in LDS
struct U1Args {
  const unsigned char *in; int64_t in_part;    // planar input, KS * 4 passes
  int64_t M;                                   // voxels
  const unsigned char *w; const float *shift;
  unsigned char *out; int64_t out_part;        // planar output (16 MB channels from this pointer's pass on)
  unsigned *flag;
};
template <int KS, int MB>
__global__ __launch_bounds__(256) void FPLK(u1conv)(U1Args a) {
  constexpr int NF = KS * PM * MB;                  // split: [s][hi MB | lo MB]; plain: [s][MB]
  unsigned char *wl = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  for (int i = tid; i < NF * 64; i += 256) reinterpret_cast<u32x4 *>(wl)[i] = reinterpret_cast<const u32x4 *>(a.w)[i];
  f32x4 sh[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) sh[b] = *reinterpret_cast<const f32x4 *>(a.shift + 4 * MB * g + 4 * b);
  __syncthreads();
  const int64_t groups = (a.M + 15) / 16;
  unsigned ovf = 0u;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < groups; grp += (int64_t)gridDim.x * 4) {
    int64_t m = grp * 16 + c;
    const bool ok = m < a.M;
    m = ok ? m : a.M - 1;
    Frag2 bf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if constexpr (SPLIT) {
        const unsigned char *p = a.in + (int64_t)(4 * s + g) * 2 * a.in_part + m * 16;
        bf[s].hi = *reinterpret_cast<const h16x8 *>(p);
        bf[s].lo = *reinterpret_cast<const h16x8 *>(p + a.in_part);
      } else {          // 32 channels = 4 planes of 8: plane g & 1 of pass 2 s + (g >> 1)
        bf[s].hi = *reinterpret_cast<const h16x8 *>(a.in + (int64_t)(4 * s + g) * a.in_part + m * 16);
      }
    }
    int zero = 0;
    asm volatile("" : "+s"(zero));                  // (keeps the fragment reads inside the loop)
    const unsigned char *wq = wl + zero + lane * 16;
    f32x4 acc[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      acc[b] = sh[b];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if constexpr (SPLIT)
          acc[b] = mfma3(*reinterpret_cast<const h16x8 *>(wq + ((s * 2 * MB + b) * 1024)),
                         *reinterpret_cast<const h16x8 *>(wq + ((s * 2 * MB + MB + b) * 1024)), bf[s], acc[b]);
        else
          acc[b] = mfma16(*reinterpret_cast<const h16x8 *>(wq + ((s * MB + b) * 1024)), bf[s].hi, acc[b]);
      }
    }
    if (ok) store_planar<MB>(a.out, a.out_part, m, g, acc, true, ovf);
  }
  if constexpr (SPLIT) ovf_commit(ovf, a.flag, FPL_RANGE_UNET);
}

}  // namespace u8
