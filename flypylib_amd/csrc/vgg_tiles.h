// Pieces shared by the fused vgg kernels (vgg_fused.hip: bf16 / f16 operands;
// vgg_split.hip: split IEEE-half operands): XCD-aware block order, LDS-DMA tile fill.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// a 16-B piece from global memory straight into LDS (global_load_lds_dwordx4): the
// wave's 64 pieces land at l + 16 * lane
__device__ __forceinline__ void glds16(const void *g, void *l) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void *)g,
      (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// Block order of the 3x3x3 kernels.  Workgroups go round-robin over the 8 XCDs, each with
// its own L2.  The 1-D grid is decoded with the z block's low three bits fastest, so an
// XCD owns whole z slabs: the x and y neighbours of a block - whose input tiles overlap
// and whose output rows share cache lines - meet in one L2.  (z neighbours never do, in
// any order: they are thousands of workgroups apart.)
struct BlockGrid { int nbx, nby, nbz; };
__host__ __device__ inline unsigned block_grid_size(const BlockGrid &g) {
  return (unsigned)((int64_t)g.nbx * g.nby * ((g.nbz + 7) / 8 * 8));
}
__device__ __forceinline__ bool block_coords(const BlockGrid &g, int &xb, int &yb, int &zb) {
  unsigned q = blockIdx.x;
  const int z8 = (int)(q & 7u);
  q >>= 3;
  xb = (int)(q % (unsigned)g.nbx); q /= (unsigned)g.nbx;
  yb = (int)(q % (unsigned)g.nby);
  zb = (int)(q / (unsigned)g.nby) * 8 + z8;
  return zb < g.nbz;
}

// Brick order (the split-operand kernels, whose concurrent tiles are twice the L2s' size):
// the 64 workgroups an XCD runs at a time take 64 consecutive indices of the order
// (z / 4, y / 4, x, y % 4, z % 4) - a 4 x 4 x 4 brick of blocks whose tiles overlap in z, y
// and x inside ONE L2 (input footprint 1.4x its share of the tensor, against 2.5x for 64
// separate tiles).  Ragged edges: the last z / y groups are thinner; closed-form decode.
__host__ __device__ inline unsigned brick_grid_size(const BlockGrid &g) {
  const int64_t n = (int64_t)g.nbx * g.nby * g.nbz;
  return (unsigned)((n + 511) / 512 * 512);
}
__device__ __forceinline__ bool brick_coords(const BlockGrid &g, int &xb, int &yb, int &zb) {
  const unsigned q = blockIdx.x, j = q >> 3;
  const int64_t b = (int64_t)(j >> 6) * 512 + (q & 7u) * 64 + (j & 63u);
  if (b >= (int64_t)g.nbx * g.nby * g.nbz) return false;
  const int64_t slab = (int64_t)g.nbx * g.nby * 4;       // blocks of a full group of 4 z
  const int zh = (int)(b / slab);
  const int nzl = g.nbz - 4 * zh < 4 ? g.nbz - 4 * zh : 4;
  int64_t r = b - zh * slab;
  const int64_t col = (int64_t)g.nbx * 4 * nzl;          // blocks of a full group of 4 y
  const int yh = (int)(r / col);
  const int nyl = g.nby - 4 * yh < 4 ? g.nby - 4 * yh : 4;
  r -= yh * col;
  const int cell = nyl * nzl;
  xb = (int)(r / cell);
  const int r3 = (int)(r - (int64_t)xb * cell);
  yb = 4 * yh + r3 / nzl;
  zb = 4 * zh + r3 % nzl;
  return true;
}

// Fill an activation tile by LDS-DMA: the tile is TZ*TY rows of TX voxels of 96 B (six
// 16-B pieces); the source is a (AZ, AY, AX) channels-last tensor whose voxels are
// SRC_VOX bytes apart, `act` pointing at the 96 B of voxel (0,0,0) the tile takes (the
// whole 48-channel voxel of a 16-bit tensor, or one pass of a split tensor).
// Piece idx = lane + 64 wave + 256 it is (row, 16-B column cw) of the tile; the
// coordinates are divided out once and then advanced by constant steps, and the edge
// clamp is a min against per-block bounds: ~10 VALU per piece (the flat index
// arithmetic this replaces, with its divisions, made the fill VALU-bound).
template <int TZ, int TY, int TX, int SRC_VOX = 96, int VB = 96>
__device__ __forceinline__ void stage_tile(const void *act, int AZ, int AY, int AX,
                                           int z0, int y0, int x0,
                                           unsigned char *tile, int wave, int lane) {
  constexpr int PV = VB / 16;                              // 16-B pieces per tile voxel
  constexpr int RC = TX * PV;                              // 16-B pieces per row
  constexpr int TOTAL = TZ * TY * RC;
  constexpr int PIECES = (TOTAL + 63) / 64;
  constexpr int DR = 256 / RC, DC = 256 % RC;              // advance per iteration
  static_assert(DR + 1 < TY, "a step wraps at most one z row");
  const int idx0 = wave * 64 + lane;
  int row = idx0 / RC, cw = idx0 % RC;
  int rz = row / TY, ry = row % TY;
  // clamp: edge blocks only feed masked outputs
  const int zmax = AZ - 1 - z0, ymax = AY - 1 - y0, xmax = AX - 1 - x0;
  const unsigned SY = (unsigned)AX * SRC_VOX, SZ = (unsigned)AY * SY;
  const unsigned char *base = reinterpret_cast<const unsigned char *>(act) +
                              (((int64_t)z0 * AY + y0) * AX + x0) * SRC_VOX;
  for (int p = wave; p < PIECES; p += 4) {
    const bool past = rz >= TZ;                            // tail lanes re-read the last piece
    const int rzc = past ? TZ - 1 : rz, ryc = past ? TY - 1 : ry, cwc = past ? RC - 1 : cw;
    const int vx = cwc / PV, pc = cwc - PV * vx;
    const int zc = rzc < zmax ? rzc : zmax, yc = ryc < ymax ? ryc : ymax, xc = vx < xmax ? vx : xmax;
    const unsigned off = (unsigned)zc * SZ + (unsigned)yc * SY + (unsigned)(xc * SRC_VOX + pc * 16);
    glds16(base + off, tile + (size_t)p * 1024);
    cw += DC;
    int dr = DR;
    if (cw >= RC) { cw -= RC; ++dr; }
    ry += dr;
    if (ry >= TY) { ry -= TY; ++rz; }
  }
}
