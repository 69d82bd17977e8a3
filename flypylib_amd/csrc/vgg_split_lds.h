// The 3x3x3 48->48 convolution on split halves with EVERY operand served from LDS: the K
// loop of vgg_like's mid and tail kernels (vgg_split.hip) from round 4 on.
//
// Why.  The round-3 kernels (two independent 4-wave workgroups per CU, a 62 KiB tile of one
// 24-channel pass filled by LDS-DMA, weight fragments streamed per wave from L2) sat at 72 %
// MFMA-busy: each pass starts with an exposed tile fill (8 of mid's 37.5 ms by proxy builds,
// profiles/r03_split_mid_proxies.txt), every wave pulls the same 6 KiB of weight fragments
// per K-step through the CU's vector-memory path (4 x 126 KiB per pass against a 62 KiB
// tile), and in-order vmcnt ties the two streams together, so prefetching the next tile from
// the computing waves only moved the stall (rounds 2 and 3).  Here the computing waves issue
// NO vector-memory load they ever wait for inside the K loop:
//
//   * passes of CQ = 8 channels (six per convolution): a pass is its tile (10 x 6 x 18 voxels
//     x [hi 16 B | lo 16 B], 37 KiB) plus ALL its weight fragments (7 K-steps x 6 KiB = 42
//     KiB), 79 KiB - and two such buffers are the CU's 160 KiB.  ONE workgroup of 8 waves per
//     CU, persistent over blocks of 8 x 4 x 16 outputs; while pass i is multiplied out of
//     buffer i & 1, the LDS-DMA of pass i + 1 (the next pass of this block, or pass 0 of the
//     next block) lands in the other buffer.  One barrier per pass (252 MFMAs per wave).
//   * weight fragments are read from LDS (ds_read_b128, lane-linear: conflict-free), so the
//     vector-memory path carries 79 KiB per pass and workgroup instead of 566 KiB per
//     24-channel pass - 4.7 x fewer bytes per MFMA through it;
//   * the tile is PLANAR: a plane of hi halves and a plane of lo halves, 16 B per voxel.
//     With 8 channels per pass a lane's 8 k-slots are one tap, lane group g of K-step s reads
//     tap 4 s + g; consecutive taps are 1, 16 (row wrap: 18 - 2) or XZS - 38 = 80 (plane wrap,
//     the z planes padded from 108 to XZS = 118 voxels) slots apart - 0 or 1 modulo 16 - so the
//     four 16-lane groups of a ds_read_b128 always cover 16 distinct 16-B slots of a 256-B
//     row or read the same address (round 3's [hi 48 B | lo 48 B] voxels: 25 % of the LDS
//     cycles were bank conflicts);
//   * tensors in HBM as planes [pass 6][part hi / lo][z][y][x][8 halves]: a tile row is one
//     288-B run, an epilogue store of 16 voxels 256 contiguous bytes.
//
// Algorithmic work is unchanged: 42 K-steps of 36 MFMAs per block and wave (six passes of
// seven; 27 taps x 8 channels = 216 of 224 k-slots used).
#pragma once
#include "mfma_util.h"
#include "vgg_tiles.h"

namespace x8 {

constexpr int CQ = 8, NQ = 6;              // channels per pass, passes
constexpr int KQ = 7;                      // K-steps per pass: taps 4 s .. 4 s + 3 (tap 27: zero weights)
constexpr int BZ = 8, BY = 4, BX = 16;     // outputs of a block (8 waves x 4 sub-steps x 16 lanes)
constexpr int TZ = BZ + 2, TY = BY + 2, TX = BX + 2;
constexpr int ZS = 118;                    // slots between z planes of the tile (TY * TX = 108, padded)
constexpr int PLANE = (TZ - 1) * ZS + TY * TX;          // 1170 slots of one part (hi or lo)
constexpr int CHUNKS = (2 * PLANE + 63) / 64;            // 37 LDS-DMA instructions (64 slots each)
constexpr int TILE_BYTES = CHUNKS * 1024;                // 37 888
constexpr int WFRAGS = KQ * 6;                           // [K-step][hi b0..2, lo b0..2], 1 KiB each
constexpr int WBYTES = WFRAGS * 1024;                    // 43 008
constexpr int BUF = TILE_BYTES + WBYTES;                 // 80 896
constexpr int KTAB_BYTES = 4 * 8 * 4;                    // tap offsets [g][s]
constexpr int SMEM = 2 * BUF + KTAB_BYTES;               // 161 920 of the CU's 163 840
constexpr int WAVES = 8;
constexpr int TCH = (CHUNKS + WAVES - 1) / WAVES;        // tile chunks per wave (5)
constexpr int WCH = (WFRAGS + WAVES - 1) / WAVES;        // weight chunks per wave (6)
static_assert((ZS - 2 * TX - 2) % 16 == 0 && (TX - 2) % 16 == 0, "conflict-free tap offsets");
static_assert(SMEM <= 160 * 1024, "one workgroup per CU");
static_assert(PLANE % 2 == 0, "");

// a split tensor in HBM: NQ passes x 2 parts of (Z, Y, X) voxels x 16 B; `slack_voxels()` more
// voxels must be readable behind the last plane (edge tiles read past the end; what they read
// only feeds outputs that are never stored)
struct Tensor {
  unsigned char *p;
  int Z, Y, X;
  int XP;                    // row pitch in voxels (>= X; x_pitch())
  __host__ __device__ int64_t part_bytes() const { return (int64_t)Z * Y * XP * 16; }
  __host__ __device__ int64_t pass_bytes() const { return 2 * part_bytes(); }
  __host__ __device__ int64_t bytes() const { return NQ * pass_bytes(); }
  __host__ __device__ int64_t slack_bytes() const { return ((int64_t)(TZ + 1) * Y * XP + 64) * 16; }
  __host__ __device__ int64_t vox(int z, int y, int x) const { return ((int64_t)z * Y + y) * XP + x; }
};

// The block grid of a persistent kernel (the order the workgroups walk it in: Cursor, below)
struct Walk {
  int nbx, nby, nbz;         // blocks
  __host__ __device__ int bricks() const { return ((nbx + 3) / 4) * ((nby + 3) / 4) * ((nbz + 1) / 2); }
};

// per-lane constants of a wave's share of the tile: chunk j = wave + 8 i holds slots
// 64 j .. 64 j + 63 of [hi plane | lo plane]; off[i] = byte offset of the slot's voxel from the
// block's origin voxel in the hi plane (the lo plane is part_bytes behind it: the host keeps a
// pass's two planes plus the tile's reach below 4 GiB, so the loads are scalar base + 32-bit
// lane offset)
struct TileDma {
  unsigned off[TCH];
};
__device__ __forceinline__ TileDma tile_dma_init(int wave, int lane, int SY, int SX, unsigned part_bytes) {
  TileDma d;
#pragma unroll
  for (int i = 0; i < TCH; ++i) {
    int slot = 64 * (wave + WAVES * i) + lane;
    slot = slot < 2 * PLANE ? slot : 2 * PLANE - 1;          // the last chunk's tail re-reads the last slot
    const int part = slot >= PLANE;
    const int s = slot - part * PLANE;
    const int tz = s / ZS;
    int rem = s - tz * ZS;
    rem = rem < TY * TX ? rem : TY * TX - 1;                 // padding slots: any valid voxel
    const int ty = rem / TX, tx = rem - ty * TX;
    d.off[i] = (unsigned)(((tz * SY + ty) * SX + tx) * 16) + (part ? part_bytes : 0u);
  }
  return d;
}

// Filling a buffer by LDS-DMA: 79 chunks of 1 KiB - the 37 tile chunks, then the 42 weight
// fragments, contiguous in the buffer - dealt round-robin to the 8 waves: chunk j = wave + 8 i,
// i = 0 .. 9.  i <= 3 is always a tile chunk, i >= 5 always a weight fragment, i = 4 a tile chunk
// for waves 0 - 4; the one index past the end (wave 7, i = 9) repeats chunk 78.  No branches.
// `org` = the hi plane's origin voxel of the pass's tile, `wpass` = the pass's weight fragments.
// Nothing is waited for here.
//
// What the fill costs (timing builds, profiles/r04_mid_proxies.txt; mid kernel, 1024^3): 27.0 ms
// without it (the MFMA pipe then 100 % busy), 34.5 with it - 2.5 ms for the 42 weight chunks,
// 4.9 for the 37 tile chunks (gathers of 288-B rows).  NOT the price of waiting (barriers without
// the vmcnt wait: the same time), of barriers (none: the same), of the bytes landing in LDS (79
// ds_write_b128 of registers instead: free) or of where the data comes from (79 DMAs of one
// L1-resident KiB: 33.8 ms): it is the LDS-DMA instruction itself, ~24 CU-cycles of stalled
// matrix pipe each.  The same bytes through registers (global_load_dwordx4, ds_write_b128 three
// or six K-steps later) cost MORE (35.3 / 37.8 ms).  Staggering the issue between SIMD partners,
// non-temporal loads, contiguous brick ranges per XCD: no change.
constexpr int NCH = CHUNKS + WFRAGS;                     // 79
constexpr int DCH = (NCH + WAVES - 1) / WAVES;           // 10 per wave
constexpr int HCH = DCH / 2;                             // issued five at a time
static_assert(TCH == 5 && 8 * 4 + 4 < CHUNKS && 8 * 4 + 5 >= CHUNKS, "chunk deal: i = 4 splits at wave 5");
static_assert(DCH % 2 == 0, "two batches");
__device__ __forceinline__ void dma_batch(const TileDma &d, int half, int wave, int lane, const unsigned char *org,
                                          const unsigned char *wpass, unsigned char *buf) {
#pragma unroll
  for (int k = 0; k < HCH; ++k) {
    const int i = HCH * half + k;
    int j = wave + WAVES * i;
    j = j < NCH ? j : NCH - 1;
    const bool tile = i < TCH - 1 || (i < TCH && j < CHUNKS);
    const unsigned char *base = tile ? org : wpass + (j - CHUNKS) * 1024;
    const unsigned off = tile ? d.off[i < TCH ? i : 0] : (unsigned)lane * 16u;
    glds16(base + off, buf + j * 1024);
  }
}

// tap offset table: entry [g][s] = byte offset of tap 4 s + g inside a part plane of the tile
__device__ __forceinline__ void ktab_init(unsigned *ktab, int tid) {
  if (tid < 32) {
    const int g = tid >> 3, s = tid & 7;
    const int tap = 4 * s + g;
    ktab[tid] = tap < 27 ? (unsigned)(((tap / 9) * ZS + ((tap / 3) % 3) * TX + tap % 3) * 16) : 0u;
  }
}

// One pass out of buffer `buf`: 7 K-steps of 4 sub-steps x 3 M-blocks x 3 products.  `vb` =
// byte offset of this lane's voxel of sub-step 0 in the hi plane; sub_off(sub) the sub-step's
// offset.  `issue(s)` is called once per K-step: the caller's LDS-DMA of the next pass goes
// there.  Everything else a K-step issues sits BETWEEN its three groups of 12 MFMAs - the B
// fragments of step s + 1 behind the first group, its weight fragments and the DMA behind the
// second - so the reads complete in the shadow of the MFMAs and the next step opens with
// operands that have landed (hipcc would otherwise wait for the 14 reads right after issuing
// them: with more than 15 LDS operations outstanding it can only wait for all).
template <typename SubOff, typename Issue>
__device__ __forceinline__ void pass_kloop(const unsigned char *buf, const unsigned *ktab_g, unsigned vb,
                                           SubOff sub_off, Issue issue, int lane, f32x4 (&acc)[4][3]) {
  const unsigned char *wl = buf + TILE_BYTES + lane * 16;
  const u32x4 k0 = *reinterpret_cast<const u32x4 *>(ktab_g), k1 = *reinterpret_cast<const u32x4 *>(ktab_g + 4);
  Frag2 bcur[4], bnxt[4];
  h16x8 wcur[6], wnxt[6];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    const unsigned char *p = buf + vb + k0[0] + sub_off(sub);
    bcur[sub].hi = *reinterpret_cast<const h16x8 *>(p);
    bcur[sub].lo = *reinterpret_cast<const h16x8 *>(p + PLANE * 16);
  }
#pragma unroll
  for (int f = 0; f < 6; ++f) wcur[f] = *reinterpret_cast<const h16x8 *>(wl + f * 1024);
#pragma unroll
  for (int s = 0; s < KQ; ++s) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[sub][b] = mfma16(wcur[3 + b], bcur[sub].hi, acc[sub][b]);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < KQ) {
      const unsigned koff = s + 1 < 4 ? k0[(s + 1) & 3] : k1[(s + 1) & 3];
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const unsigned char *p = buf + vb + koff + sub_off(sub);
        bnxt[sub].hi = *reinterpret_cast<const h16x8 *>(p);
        bnxt[sub].lo = *reinterpret_cast<const h16x8 *>(p + PLANE * 16);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[sub][b] = mfma16(wcur[b], bcur[sub].lo, acc[sub][b]);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < KQ) {
#pragma unroll
      for (int f = 0; f < 6; ++f) wnxt[f] = *reinterpret_cast<const h16x8 *>(wl + ((s + 1) * 6 + f) * 1024);
    }
    issue(s);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[sub][b] = mfma16(wcur[b], bcur[sub].hi, acc[sub][b]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < KQ) {
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) bcur[sub] = bnxt[sub];
#pragma unroll
      for (int f = 0; f < 6; ++f) wcur[f] = wnxt[f];
    }
  }
}

// The whole convolution of one block (six passes) inside the persistent loop.
//   org_cur / org_nxt: the hi-plane origin voxel of pass 0 of this / the next block's tile
//                      (no next block: org_nxt = org_cur)
// On entry buffer 0 holds pass 0 of this block (landed, barrier passed); on exit buffer 0
// holds pass 0 of the next block likewise (if any), and every wave has left buffer 1.
// The pass loop is rolled in pairs (buffer 0, buffer 1): LDS addresses stay immediates.
template <typename SubOff>
__device__ __forceinline__ void conv_block(unsigned char *smem, const unsigned *ktab_g, const TileDma &td,
                                           const unsigned char *org_cur, const unsigned char *org_nxt,
                                           int64_t part_bytes, const unsigned char *wglobal, unsigned vb,
                                           SubOff sub_off, int wave, int lane, f32x4 (&acc)[4][3]) {
#pragma unroll 1
  for (int pp = 0; pp < NQ / 2; ++pp) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pass = 2 * pp + h;
      unsigned char *buf = smem + h * BUF, *other = smem + (h ^ 1) * BUF;
      const bool more = pass + 1 < NQ;
      // (after the last block the last pass re-loads pass 0 of that block: harmless, unread)
      const unsigned char *torg = more ? org_cur + (int64_t)(pass + 1) * 2 * part_bytes : org_nxt;
      const unsigned char *tw = wglobal + (size_t)(more ? pass + 1 : 0) * WBYTES;
      auto issue = [&](int s) {
        // the next pass's DMA goes out in two batches, at K-steps 0 and 3 (at least three K-steps
        // for it to land); the wave steps down from the MFMA cluster's priority while it issues
        if (s == 0 || s == 3) {
          __builtin_amdgcn_s_setprio(0);
          dma_batch(td, s == 3, wave, lane, torg, tw, other);
          __builtin_amdgcn_s_setprio(1);
        }
      };
      pass_kloop(buf, ktab_g, vb, sub_off, issue, lane, acc);
      // every wave has left `buf` and its share of the next pass has landed (the barrier's
      // release waits for this wave's vector-memory operations, LDS-DMA included)
      __syncthreads();
    }
  }
}

// the first block's pass 0 into buffer 0 (every wave its chunks), landed and visible on return
__device__ __forceinline__ void prime(unsigned char *smem, const TileDma &td, const unsigned char *org,
                                      int64_t part_bytes, const unsigned char *wglobal, int wave, int lane) {
#pragma unroll
  for (int half = 0; half < 2; ++half) dma_batch(td, half, wave, lane, org, wglobal, smem);
  __syncthreads();
}

// The persistent workgroup's walk.  The blocks are numbered in BRICK-MAJOR order (bricks of 4 x 4
// x 2 blocks, x fastest; the bricks at the far faces hold fewer blocks) and group = blockIdx.x & 7
// (one XCD under round-robin placement) takes the CONTIGUOUS range [group * per, (group + 1) *
// per) of that numbering, per = ceil(blocks / 8); its S = gridDim.x / 8 workgroups take every
// S-th block of the range.  The S blocks in flight at a time are neighbours of one or two bricks,
// whose halo faces sit in that XCD's L2, and every workgroup gets the same number of blocks +- 1.
// (Until round 4 a workgroup owned ONE slot of every brick: with 17 x 65 x 33 blocks - the
// 520^3 volume - the workgroups of slot 0 had 27 % more blocks than the average, and the kernel
// took as long as they did.)
__device__ __forceinline__ void walk_decode(const Walk &w, int i, int &bx, int &by, int &bz) {
  const int layer = w.nbx * w.nby * 2;
  const int kz = i / layer;
  int r = i - kz * layer;
  const int wz = min(2, w.nbz - 2 * kz);
  const int rowb = w.nbx * 4 * wz;
  const int ky = r / rowb;
  r -= ky * rowb;
  const int wy = min(4, w.nby - 4 * ky);
  const int bb = 4 * wy * wz;
  const int kx = r / bb;
  r -= kx * bb;
  const int wx = min(4, w.nbx - 4 * kx);
  bx = 4 * kx + r % wx;
  r /= wx;
  by = 4 * ky + r % wy;
  bz = 2 * kz + r / wy;
}
struct Cursor {
  int i, iend;               // block number, end of the group's range
  int bx, by, bz;
};
__device__ __forceinline__ bool cursor_first(const Walk &w, int /*nbricks*/, int group, int slot, int S, Cursor &c) {
  const int total = w.nbx * w.nby * w.nbz, per = (total + 7) / 8;
  c.i = group * per + slot;
  c.iend = (group + 1) * per < total ? (group + 1) * per : total;
  if (c.i >= c.iend) return false;
  walk_decode(w, c.i, c.bx, c.by, c.bz);
  return true;
}
__device__ __forceinline__ bool cursor_next(const Walk &w, int /*slot*/, int S, Cursor &c) {
  c.i += S;
  if (c.i >= c.iend) return false;
  walk_decode(w, c.i, c.bx, c.by, c.bz);
  return true;
}

// A lane's three accumulator tiles of a 48-channel layer packed with fpl_out_channel(il = 2):
// v[0], v[1] = the 8 channels of pass g, v[2] = channels [32 + 4 g, 32 + 4 g + 4) = half
// g & 1 of pass 4 + g / 2.  Four stores (16 + 8 B of hi halves, 16 + 8 B of lo halves), the same
// in every lane; a wave's 16 voxels of a row are 256 contiguous bytes per store.  `ovf`: the
// half-range guard.
__host__ __device__ constexpr int out_channel(int b, int g, int r) { return b < 2 ? 8 * g + 4 * b + r : 32 + 4 * g + r; }
__device__ __forceinline__ void store12(const Tensor &t, int64_t vox, int g, const f32x4 (&v)[3], unsigned &ovf) {
  unsigned hi[6], lo[6];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const Pair2 p0 = split_pk(v[b][0], v[b][1], ovf), p1 = split_pk(v[b][2], v[b][3], ovf);
    hi[2 * b] = p0.hi; hi[2 * b + 1] = p1.hi;
    lo[2 * b] = p0.lo; lo[2 * b + 1] = p1.lo;
  }
  const int64_t part = t.part_bytes();
  unsigned char *q = t.p + (int64_t)g * 2 * part + vox * 16;                         // pass g, hi plane
  unsigned char *h = t.p + (int64_t)(4 + (g >> 1)) * 2 * part + vox * 16 + 8 * (g & 1);
  *reinterpret_cast<u32x4 *>(q) = u32x4{hi[0], hi[1], hi[2], hi[3]};
  *reinterpret_cast<u32x4 *>(q + part) = u32x4{lo[0], lo[1], lo[2], lo[3]};
  *reinterpret_cast<u32x2 *>(h) = u32x2{hi[4], hi[5]};
  *reinterpret_cast<u32x2 *>(h + part) = u32x2{lo[4], lo[5]};
}

// The same store without a branch around it: lanes with `valid` false write their four pieces
// to `dump` (64 B of scratch nobody reads; the pieces apart, so that no two merge into one store) and leave the guard alone.  For a kernel whose
// loads in flight are counted against the stores behind them (vggs_stem_pool): with the
// stores in a conditional block the wait-count pass has to assume the path without them.
__device__ __forceinline__ void store12_sel(const Tensor &t, int64_t vox, int g, const f32x4 (&v)[3],
                                            unsigned &ovf, bool valid, unsigned char *dump) {
  unsigned hi[6], lo[6], o2 = 0u;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const Pair2 p0 = split_pk(v[b][0], v[b][1], o2), p1 = split_pk(v[b][2], v[b][3], o2);
    hi[2 * b] = p0.hi; hi[2 * b + 1] = p1.hi;
    lo[2 * b] = p0.lo; lo[2 * b + 1] = p1.lo;
  }
  ovf_note(ovf, valid ? o2 : 0u);
  const int64_t part = t.part_bytes();
  unsigned char *q = t.p + (int64_t)g * 2 * part + vox * 16;
  unsigned char *h = t.p + (int64_t)(4 + (g >> 1)) * 2 * part + vox * 16 + 8 * (g & 1);
  *reinterpret_cast<u32x4 *>(valid ? q : dump) = u32x4{hi[0], hi[1], hi[2], hi[3]};
  *reinterpret_cast<u32x4 *>(valid ? q + part : dump + 16) = u32x4{lo[0], lo[1], lo[2], lo[3]};
  *reinterpret_cast<u32x2 *>(valid ? h : dump + 32) = u32x2{hi[4], hi[5]};
  *reinterpret_cast<u32x2 *>(valid ? h + part : dump + 48) = u32x2{lo[4], lo[5]};
}

}  // namespace x8
