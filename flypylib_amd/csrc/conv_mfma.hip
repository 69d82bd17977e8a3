// Generic bf16 MFMA convolution kernels (channels-last, batched tiles) and the
// fused-op executor for unet_like2 inference (flypylib/fplmodels.py:258-304).
//
//   conv3_bf16<MB, PF, STEM>  conv3 CIN->16*MB, CIN = ncc chunks of 32 channels,
//                             input = concat of sources, each optionally nearest-
//                             upsampled x2 or cropped (UpSampling3D / Cropping3D /
//                             concatenate become an index remap in the tile loader);
//                             STEM: the source is conv3 1->32 of the raw f32 tiles,
//                             computed straight into the LDS tile
//   conv1_bf16<CIN, MB, TAIL> 1x1x1 conv as a voxel GEMM; TAIL chains a second
//                             1x1 conv to one sigmoid channel in registers
//   (MaxPooling3D(2) is an epilogue option of conv3_bf16)
//
// conv3_bf16: 4 waves, output block 4 x 4 x 16, wave = z, sub-steps = y, lanes = x;
// persistent over blocks; planar activation tile (6 x 6 x 18 voxels, one 32-channel
// chunk at a time) in LDS; K order (chunk, dz, dx, dy) with y-row fragment reuse;
// weight fragments go from L2 straight into registers, a few K-steps ahead;
// <= 80 KiB LDS so two workgroups share a CU.  Details at the kernel.
#include <algorithm>
#include <cmath>
#include <type_traits>

#include "fast_paths.h"
#include "mfma_util.h"
#include "pack_weights.h"
#include "vgg_tiles.h"      // glds16 (LDS-DMA), used by unet_split_lds.h

namespace {

// Split build (-DFPL_SPLIT, precision f16s): every tensor carries, per 16 REAL channels,
// [hi 16 | lo 16] halves (v = hi + lo, mfma_util.h) - 2 C "physical" channels per voxel.
// A staged chunk of 32 physical channels is then ONE group of 16 real channels, its
// K-step B fragment [a_hi | a_lo], multiplied by [w_hi | w_hi]: a_hi w_hi + a_lo w_hi in
// one MFMA per 16 channels and tap.  The third product a_hi w_lo needs only the hi halves,
// half a K-step: in conv3 TWO taps share one - lanes g < 2 read the hi planes at row group
// k, lanes g >= 2 the hi planes at row group k' (a per-lane LDS address, no lane movement),
// against [w_lo(k) | w_lo(k')] - so a chunk is 27 + 15 K-steps (9 row groups + 5 paired
// ones) instead of 2 x 27: 3.1 MFMAs per product, not 4.  The 1x1x1 kernels keep the
// [w_lo | w_lo] second set (a_lo w_lo rides along unused).  The kernels below are the
// 16-bit ones; what changes is the K-step sequence and weight set, the epilogues (hi / lo
// stores, pooling in fp32) and the stem.
#ifdef FPL_SPLIT
constexpr bool SPLIT = true;
#else
constexpr bool SPLIT = false;
#endif
constexpr int PM = SPLIT ? 2 : 1;         // physical halves per real channel
constexpr int CC = 32;                    // PHYSICAL channels per staged chunk = one K-step per tap
constexpr int RCH = CC / PM;              // real channels per staged chunk
constexpr int TZ = 6, TY = 6, TX = 18;    // input tile of a 4 x 4 x 16 output block
// LDS tile layout: four planes, plane q = channels 8q..8q+7 of every tile voxel at a
// 16-B pitch.  A lane (c, g) reads plane g, voxel v0 + c: the 16-lane groups of a
// ds_read_b128 ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS) then cover 16
// distinct 16-B slots when the plane size is a multiple of 256 B - conflict-free
// with no padding per voxel (a 64-B voxel pitch would be 4-way conflicted, 80 B
// 2-way on 3 of 16 slots; 96 B is clean but wastes a third of the tile).
constexpr int PITCH = 16;
constexpr int PLANE = (TZ * TY * TX * PITCH + 255) / 256 * 256;
constexpr int TILE_BYTES = 4 * PLANE;
constexpr int NPIECE = TZ * TY * TX * 4;  // 16-B pieces of a tile
constexpr int NT = (NPIECE + 255) / 256;  // pieces per thread
constexpr int MAXTAB = 2;                 // distinct source geometries per launch
constexpr int TABN = NT * 64;              // voxel-offset table entries (>= tile voxels)
constexpr int TAB_BYTES = MAXTAB * TABN * 4;
static_assert(NPIECE % 32 == 0, "tile pieces come in groups of 32");
constexpr int NCH = 9;                    // row groups per channel chunk: (dz, dx)
constexpr int KC = 3;                     // K-steps per row group: dy
// Row-group sequence of a chunk.  16-bit builds: the 9 (dz, dx) groups.  Split build:
// M0 M1 P(0,1) M2 M3 P(2,3) M4 M5 P(4,5) M6 M7 P(6,7) M8 P(8,-): M = [a_hi | a_lo] of one
// group against [w_hi | w_hi], P = [a_hi(k) | a_hi(k')] against [w_lo(k) | w_lo(k')].
constexpr int NG = SPLIT ? 14 : NCH;
constexpr bool grp_pair(int gi) { return SPLIT && (gi == 13 || (gi < 12 && gi % 3 == 2)); }
constexpr int grp_k0(int gi) {
  return !SPLIT ? gi : gi >= 12 ? 8 : gi % 3 == 2 ? 2 * (gi / 3) : 2 * (gi / 3) + gi % 3;
}
constexpr int grp_k1(int gi) { return gi >= 12 ? 8 : 2 * (gi / 3) + 1; }   // pairs only
constexpr unsigned grp_off(int k, int ty) { return (unsigned)(((k / 3) * ty * TX + k % 3) * PITCH); }
// Parity form (sources that are an UpSampling3D(2): conv3 192->64's first 128 input
// channels, the head's first 64).  The tile's z planes come in pairs that are copies of one
// low-resolution plane, so the three z taps collapse to two with pre-summed weights that depend
// on the output plane's parity p = z & 1 (block origins are even, the source is not cropped):
//   p = 0: planes z, z + 1 are the same -> (w0 + w1) at plane z,  w2 at plane z + 2
//   p = 1: planes z + 1, z + 2 are      ->  w0 at plane z,  (w1 + w2) at plane z + 1
// A wave owns one z plane, so its parity is the wave's: the upsampled chunk's row groups (dz',
// dx) are the plain sequence's first six (split build: M0 M1 P(0,1) M2 M3 P(2,3) M4 M5 P(4,5), NGU = 9
// groups) of KC steps - with the "dz = 1" plane one or two planes on, and the dz = 2 groups gone: 27
// weight steps instead of 42 (16-bit builds: 18 instead of 27), MFMAs and weight stream both at
// two thirds.  The pre-summed weights are
// made in fp32 and split afterwards (pack_conv3_parity); a wave reads the stream of its parity.
// (Pre-summing along y as well - sub-step parity, 36 steps feeding half the sub-steps each, 43 %
// of the MFMAs but 86 % of the weight stream - was built too: the same time for the head, a
// slowdown for the 64-output layer, whose fragment stream is what the CU's vector-memory path
// can just deliver; profiles/r04_unet_parity_proxy.txt.)
constexpr int NGU = SPLIT ? 9 : 6;
// second plane of the pair for parity p: z + 2 (p = 0) or z + 1 (p = 1)
constexpr int par_row(int p) { return p ? 1 : 2; }
// Geometry of a block of 4 (z: one plane per wave) x R (y: rows per wave) x 16 (x: lanes)
// outputs.  R = 4 everywhere but the 32-output-channel kernels (unet_like2's stem: R = 8, 96
// rows = 12 blocks; its head: R = 6, 82 rows = 14 blocks): a weight fragment then feeds 8 / 6
// MFMAs instead of 4 - these kernels' vector-memory return path is ~90 % busy with the
// per-wave weight stream (profiles/r03_pmc_unet264.json) - and the tile carries 2.1 / 2.25
// instead of 2.53 input voxels per output.  Measured (27 tiles of 100^3, f16 / split): stem
// 1.50 -> 1.38 / 4.3 -> 3.8 ms, head 2.47 -> 2.32 / 6.65 -> 5.96 ms (R = 8 for the head:
// 2.35 / 6.1, it pads 82 rows to 88 and spills).  The 64-channel kernels stay at R = 4: 8
// rows of 4 M-blocks do not fit 256 registers.
template <int R> struct Geo {
  static constexpr int TY = R + 2;
  static constexpr int PLANE = (TZ * TY * TX * PITCH + 255) / 256 * 256;
  static constexpr int TILE_BYTES = 4 * PLANE;
  static constexpr int NPIECE = TZ * TY * TX * 4;
  static constexpr int NT = (NPIECE + 255) / 256;
  static constexpr int TABN = NT * 64;
  static constexpr int TAB_BYTES = MAXTAB * TABN * 4;
  static constexpr int RY = TY + 2;
  static constexpr int NRAW = (TZ + 2) * RY * (TX + 2);
  static constexpr int NRAWT = (NRAW + 255) / 256;       // raw values per thread
  static_assert(NPIECE % 32 == 0, "tile pieces come in groups of 32");
};

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

struct Src {             // one CC-channel chunk of the (virtual) concatenated input
  const h16_t *p;       // (n, D, H, W, C) bf16, with read slack behind it (tile_slack)
  int D, H, W, C;
  int ch0;               // first channel of this chunk inside the source
  int ups;               // 0, or 1 = UpSampling3D(2) of the source (index >> ups)
  int crop;              // Cropping3D(crop) of the source (only with ups == 0)
  int tab;               // which offset table (geometry H, W, C, ups) this source uses
};

// Activation tensors are stored as planes of CC = 32 physical channels: [chunk][n][z][y][x][32]
// (round 3).  A tile loader reads ONE chunk at a time, and with the chunks interleaved per
// voxel (64 of every 128 / 256 / 512 B) each fill touched two to eight times the cache
// lines it used; a chunk plane makes a tile row one contiguous run.  `plane` = elements of
// one chunk plane of the tensor being written.
//
// Epilogue store for interleaved output channels (pack_weights.h, fpl_out_channel):
// lane (c, g) writes the 4*MB contiguous channels [4*MB*g, ...) of its voxel; `vox0` is the
// voxel's position in chunk plane 0.
// `ovf`: the split build's half-range guard (mfma_util.h)
// `planar` (bf16 / f16 builds): the output is a planar tensor of the all-LDS kernels
// (unet_split_lds.h): piece i of 8 channels at vox0 + i * plane, vox0 = the voxel's 16 B in piece 0
template <int MB, bool RELU_ALWAYS>
__device__ __forceinline__ void store_il(h16_t *vox0, int64_t plane, int g, const f32x4 (&acc)[MB], int relu,
                                         unsigned &ovf, int planar = 0) {
  static_assert(MB % 2 == 0, "16-B pieces");
  if (SPLIT) {
    // the lane's 4*MB real channels start at 4*MB*g; per 8 of them one 16-B piece of hi
    // halves and, 16 halves behind it, one of lo halves, inside their chunk of 16 real
#pragma unroll
    for (int h = 0; h < MB / 2; ++h) {
      const int ch = 4 * MB * g + 8 * h;
      h16_t *d = vox0 + (ch / 16) * plane + ch % 16;
      u32x4 hi, lo;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 &v = acc[2 * h + q];
        const bool rl = RELU_ALWAYS || relu;
        const Pair2 p0 = rl ? split_pk_relu(v[0], v[1], ovf) : split_pk_signed(v[0], v[1], ovf);
        const Pair2 p1 = rl ? split_pk_relu(v[2], v[3], ovf) : split_pk_signed(v[2], v[3], ovf);
        hi[2 * q] = p0.hi; hi[2 * q + 1] = p1.hi;
        lo[2 * q] = p0.lo; lo[2 * q + 1] = p1.lo;
      }
      *reinterpret_cast<u32x4 *>(d) = hi;
      *reinterpret_cast<u32x4 *>(d + 16) = lo;
    }
    return;
  }
#pragma unroll
  for (int h = 0; h < MB / 2; ++h) {
    const int ch = 4 * MB * g + 8 * h;
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      o[2 * q] = cvt_pk_h16(acc[2 * h + q][0], acc[2 * h + q][1]);
      o[2 * q + 1] = cvt_pk_h16(acc[2 * h + q][2], acc[2 * h + q][3]);
    }
    if (RELU_ALWAYS || relu) {
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = pk_max_i16(o[q], 0u);
    }
    *reinterpret_cast<u32x4 *>(planar ? vox0 + (ch / 8) * plane : vox0 + (ch / CC) * plane + ch % CC) = o;
  }
}

struct Conv3Args {
  Src src[12];                   // (split build: chunks of 16 real channels - up to 192 / 16)
  int ncc;                       // channel chunks
  int ntab;                      // offset tables in use
  int tabH[MAXTAB], tabW[MAXTAB], tabC[MAXTAB], tabU[MAXTAB];
  const unsigned char *w;        // fragments [cc][dz][dx][dy][mb], 1 KiB each
  const float *shift;
  int relu;
  h16_t *out;                   // chunk plane 0 of the output's channels [0, 16*MB): (n, OD, OH, OW, 32)
  int64_t oplane;                // elements of one chunk plane of `out` (n * OD * OH * OW * 32)
  int OD, OH, OW, zblocks;       // zblocks = ceil(OD/4)
  int nbx, nby, nbz;             // blocks: ceil(OW/16), ceil(OH/4), n * zblocks
  // STEM variant: the (single) source is conv3 1->32 + shift + ReLU of this raw
  // (n, T, T, T) f32 volume, computed into the tile instead of being read
  const float *raw; int T;
  const h16x8 *wstem;           // 2 fragments (SLOT_STEM, interleaved rows); split: [chunk][part]
  const float *shstem;
  // optional fused MaxPooling3D(2) of the (ReLU) output: (n, OD/2, OH/2, OW/2, 16*MB)
  h16_t *pool_out;
  int64_t pplane;                // elements of one chunk plane of `pool_out`
  // Edge strip (transposed view).  An output width that is not a multiple of 16
  // wastes lanes in the last x block (OW = 82: 14 of 16).  The strip x in [xorg, OW)
  // is then run with the block's axes swapped: lanes walk y, the four sub-steps walk
  // x - the same kernel through a transposed offset table, a transposed block origin
  // and weight fragments packed with the dy / dx taps swapped (`w` points at those).
  int transposed, xorg;
  int main_w;                    // untransposed launch: columns [0, main_w) only (0 = all)
  // volume-side output (FplTileIO, fast_paths.h).  HEAD: the epilogue chains conv1 32->32 (+shift, ReLU) and conv1 32->1 (+bias, sigmoid) in
  // registers and stores the probability of every valid voxel into the prediction
  // volume - no 32-channel tensor, no separate head kernel, no stitch pass.
  FplTileIO io;
  const h16x8 *w8, *w9;          // HEAD: 2 fragments (SLOT_SPATIAL), 1 fragment (SLOT_CHAIN); split: [part][b]
  const float *sh8;
  float bias9;
  // split build: half-range guard (mfma_util.h).  STEM: conv3 1->32's outputs are bounded on
  // the host for raw inputs |x| <= xlim, which the kernel checks per raw voxel
  unsigned *flag;
  float xlim;
  // parity form (UPSP instantiations): `w` = the weight stream of even z planes, the odd planes'
  // `wstream` bytes behind it; total_steps = K-steps of one stream (all launches: set by launch_conv3)
  int parity, total_steps;
  int64_t wstream;
  // bf16 / f16 builds: `out` / `pool_out` are planar tensors (store_il)
  int planar;
};

// K order: channel chunk -> dz -> dx -> dy.  For a fixed (chunk, dz, dx) the four
// output rows y0..y0+3 of a wave and the three dy taps touch only six tile rows, so
// a lane holds those six B fragments and every fragment feeds up to 3*MB MFMAs:
// 6 + 3*MB LDS fragment reads per 12*MB MFMAs (12 + 3*MB without the reuse).
//
// Persistent: a workgroup walks output blocks wg, wg + G, ... and flattens
// (block, channel chunk) into one sequence of tiles; the global loads of the next
// tile are issued before the K loop of the current one and land in registers while
// the MFMAs run (PF), so only the LDS store sits between two K loops.
//
// Tile addressing: block origins are even and the sources carry read slack, so a
// piece's address is (uniform block/source base) + (per-thread offset that depends
// only on the source geometry).  The offsets are tabulated once per workgroup in
// LDS - a fetch is one ds_read_b32 and one saddr+voffset global load per piece, no
// per-piece index arithmetic or clamping.  Reads past a source's edge land in its
// neighbouring rows / the slack and only ever feed masked output voxels.
//
// STEM (unet_like2's first pair conv3 1->32, conv3 32->32): the 32-channel tile is
// not read but COMPUTED from the raw (TZ+2, TY+2, TX+2) f32 tile - 41 groups of 16
// tile voxels, two MFMAs each (27 taps in one K-step); with interleaved weight rows
// a lane's 8 outputs are exactly one 16-B piece of plane g.  The 98^3 x 32 stem
// output (44 GB per 729 tiles) never exists in HBM.
constexpr int RZ = TZ + 2, RY = TY + 2, RX = TX + 2;     // raw tile 8 x 8 x 20
constexpr int NRAW = RZ * RY * RX;                       // 1280 = 5 per thread

template <int MB, bool PF, bool STEM = false, bool POOL = false, bool HEAD = false, int R = 4, bool UPSP = false>
__global__ __launch_bounds__(256, (MB == 2 && !SPLIT && R == 4) ? 3 : 2) void FPLK(conv3)(Conv3Args a) {
  static_assert(!UPSP || (!STEM && !POOL), "the parity form: plain sources");
  static_assert(!STEM || (MB == 2 && PF), "the stem variant is conv3 32->32");
  static_assert(!HEAD || (MB == 2 && !POOL), "the head variant is conv3 ->32");
  static_assert(R % 2 == 0, "pool pairs");
  // this instantiation's geometry (the names shadow the R = 4 constants above)
  constexpr int TY = Geo<R>::TY, PLANE = Geo<R>::PLANE, TILE_BYTES = Geo<R>::TILE_BYTES;
  constexpr int NT = Geo<R>::NT, TABN = Geo<R>::TABN, RY = Geo<R>::RY, NRAW = Geo<R>::NRAW;
  constexpr int NRAWT = Geo<R>::NRAWT;
  // K-steps of weight fragments in flight (27 % WQ == 0)
  constexpr int WQ = 3;
  constexpr int ROW = TX * PITCH;
  unsigned char *tile = smem;
  unsigned *offtab = reinterpret_cast<unsigned *>(smem + TILE_BYTES);
  // STEM: [0, TABN) = raw-tile offset of tile voxel v; then the bf16 raw tile
  unsigned short *rawt = reinterpret_cast<unsigned short *>(offtab + TABN);   // split: hi, then lo
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = UPSP ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;   // (scalar: the parity form's row offsets)
  const int c = lane & 15, g = lane >> 4;
  const int G = (int)gridDim.x;                     // multiple of 8 (host)
  // consecutive logical workgroups share an XCD (and its L2): halo reuse
  const int wg = ((int)blockIdx.x % 8) * (G / 8) + (int)blockIdx.x / 8;
  const int64_t total_blocks = (int64_t)a.nbx * a.nby * a.nbz;
  int64_t blk = wg;
  if (blk >= total_blocks) return;

  // piece p = tid + 256 j of a tile is the 16-B piece pc = (tid / 8) % 4 of voxel
  // vox0 + 64 j, vox0 = (tid / 32) * 8 + tid % 8: 8 consecutive lanes store the same
  // piece of 8 consecutive voxels (a conflict-free 128-B LDS store) and a wave still
  // reads 1 KiB contiguous.  offtab[t][v] = byte offset of tile voxel v in a source
  // of geometry t.
  if (STEM) {
    for (int i = tid; i < TABN; i += 256) {
      const int vox = i < TZ * TY * TX ? i : TZ * TY * TX - 1;
      const int tz = vox / (TY * TX), ty = (vox / TX) % TY, tx = vox % TX;
      offtab[i] = (unsigned)((tz * RY + ty) * RX + tx);
    }
  }
  for (int t = 0; t < (STEM ? 0 : a.ntab); ++t) {
    const int H = a.tabH[t], W = a.tabW[t], C = a.tabC[t], U = a.tabU[t];
    for (int i = tid; i < TABN; i += 256) {
      const int vox = i < TZ * TY * TX ? i : TZ * TY * TX - 1;
      const int tz = vox / (TY * TX), ty = (vox / TX) % TY, tx = vox % TX;
      const int sy = a.transposed ? tx : ty, sx = a.transposed ? ty : tx;   // source y, x
      offtab[t * TABN + i] = (unsigned)(((((tz >> U) * H + (sy >> U)) * W + (sx >> U)) * C) * 2);
    }
  }
  const int vox0 = (tid >> 5) * 8 + (tid & 7);
  const unsigned pc = (unsigned)((tid >> 3) & 3);
  unsigned ovf = 0u, xmax = 0u;                     // split build: half-range guard
  u32x4 nt[STEM ? 1 : NT];
  float rawv[STEM ? NRAWT : 1];
  auto fetch = [&](int64_t fb, int cc) {
    const int bx = (int)(fb % a.nbx), by = (int)((fb / a.nbx) % a.nby);
    const int bz = (int)(fb / ((int64_t)a.nbx * a.nby));
    const int n = bz / a.zblocks;
    if (STEM) {
      const int z0 = (bz % a.zblocks) * 4, y0 = by * R, x0 = bx * 16;
      const float *base = a.raw + (int64_t)n * a.T * a.T * a.T;
#pragma unroll
      for (int j = 0; j < NRAWT; ++j) {
        const int p = min(tid + 256 * j, NRAW - 1);
        int z = z0 + p / (RY * RX), y = y0 + (p / RX) % RY, x = x0 + p % RX;
        z = z < a.T ? z : a.T - 1;                 // clamped reads only feed masked
        y = y < a.T ? y : a.T - 1;                 // outputs
        x = x < a.T ? x : a.T - 1;
        rawv[j] = base[((int64_t)z * a.T + y) * a.T + x];
      }
      return;
    }
    const Src s = a.src[cc];
    const int z0 = ((bz % a.zblocks) * 4 + s.crop) >> s.ups;
    const int y0 = ((a.transposed ? bx * 16 : by * R) + s.crop) >> s.ups;
    const int x0 = ((a.transposed ? a.xorg + by * R : bx * 16) + s.crop) >> s.ups;
    const unsigned char *base = reinterpret_cast<const unsigned char *>(
        s.p + ((((int64_t)n * s.D + z0) * s.H + y0) * s.W + x0) * s.C + s.ch0);
    const unsigned *tab = offtab + s.tab * TABN + vox0;
#pragma unroll
    for (int j = 0; j < (STEM ? 1 : NT); ++j)
      nt[j] = *reinterpret_cast<const u32x4 *>(base + (tab[64 * j] + pc * 16));
  };
  // STEM: lane constants of the gather (tap 8g+j of the 27, k-slots 27..31 unused)
  int toff[8];
  h16x8 wsf[2];
  f32x4 shs[2];
  if (STEM) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 8 * g + j;
      toff[j] = t < 27 ? ((t / 9) * RY + (t / 3) % 3) * RX + t % 3 : 0;
    }
    if (!SPLIT) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        wsf[b] = a.wstem[b * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) shs[b][r] = a.shstem[8 * g + 4 * b + r];
      }
    }
  }
  auto put = [&](int cc) {
    if (STEM && SPLIT) {
      // conv3 1->32 of the raw tile for the chunk's 16 real channels [16 cc, 16 cc + 16):
      // input and weights as hi + lo, three products; plain rows, so lane (c, g) holds
      // channels 4g .. 4g+3 of its voxel: 8 B of plane g / 2 (hi) and of plane 2 + g / 2 (lo)
#pragma unroll
      for (int j = 0; j < NRAWT; ++j) {
        const float x = rawv[j];
        const unsigned ax = __builtin_bit_cast(unsigned, x) & 0x7FFFFFFFu;
        xmax = ax > xmax ? ax : xmax;
        const h16_t h = (h16_t)x;
        if (tid + 256 * j < NRAW) {
          rawt[tid + 256 * j] = h16_bits(x);
          rawt[NRAW + tid + 256 * j] = h16_bits(x - (float)h);
        }
      }
      const h16x8 wh = a.wstem[(cc * 2 + 0) * 64 + lane], wl = a.wstem[(cc * 2 + 1) * 64 + lane];
      f32x4 sh;
#pragma unroll
      for (int r = 0; r < 4; ++r) sh[r] = a.shstem[16 * cc + 4 * g + r];
      __syncthreads();                              // raw tiles visible
      constexpr int NGRP = (TZ * TY * TX + 15) / 16;
      for (int grp = wave; grp < NGRP; grp += 4) {
        const int v = 16 * grp + c;
        const unsigned ro = offtab[v];              // TABN >= 16 * NGRP, tail clamped
        u16x8 rh, rl;
#pragma unroll
        for (int j = 0; j < 8; ++j) { rh[j] = rawt[ro + toff[j]]; rl[j] = rawt[NRAW + ro + toff[j]]; }
        Frag2 bf;
        bf.hi = __builtin_bit_cast(h16x8, rh);
        bf.lo = __builtin_bit_cast(h16x8, rl);
        const f32x4 a0 = mfma3(wh, wl, bf, sh);
        const Pair2 p0 = split_pk_relu(a0[0], a0[1]), p1 = split_pk_relu(a0[2], a0[3]);
        if (v < TZ * TY * TX) {
          unsigned char *d = tile + (g >> 1) * PLANE + v * PITCH + 8 * (g & 1);
          *reinterpret_cast<u32x2 *>(d) = u32x2{p0.hi, p1.hi};
          *reinterpret_cast<u32x2 *>(d + 2 * PLANE) = u32x2{p0.lo, p1.lo};
        }
      }
      return;
    }
    if (STEM) {
#pragma unroll
      for (int j = 0; j < NRAWT; ++j)
        if (tid + 256 * j < NRAW) rawt[tid + 256 * j] = h16_bits(rawv[j]);
      __syncthreads();                              // raw tile visible
      constexpr int NGRP = (TZ * TY * TX + 15) / 16;
      for (int grp = wave; grp < NGRP; grp += 4) {
        const int v = 16 * grp + c;
        const unsigned ro = offtab[v];              // TABN >= 16 * NGRP, tail clamped
        u16x8 rw;
#pragma unroll
        for (int j = 0; j < 8; ++j) rw[j] = rawt[ro + toff[j]];
        const h16x8 bf = __builtin_bit_cast(h16x8, rw);
        const f32x4 a0 = mfma16(wsf[0], bf, shs[0]);
        const f32x4 a1 = mfma16(wsf[1], bf, shs[1]);
        u32x4 o;
        o[0] = pk_max_i16(cvt_pk_h16(a0[0], a0[1]), 0u);
        o[1] = pk_max_i16(cvt_pk_h16(a0[2], a0[3]), 0u);
        o[2] = pk_max_i16(cvt_pk_h16(a1[0], a1[1]), 0u);
        o[3] = pk_max_i16(cvt_pk_h16(a1[2], a1[3]), 0u);
        if (v < TZ * TY * TX) *reinterpret_cast<u32x4 *>(tile + g * PLANE + v * PITCH) = o;
      }
      return;
    }
    unsigned char *dst = tile + pc * PLANE + vox0 * PITCH;
#pragma unroll
    for (int j = 0; j < NT - 1; ++j) *reinterpret_cast<u32x4 *>(dst + 64 * j * PITCH) = nt[j];
    if (vox0 < TZ * TY * TX - 64 * (NT - 1))       // the last round is partial
      *reinterpret_cast<u32x4 *>(dst + 64 * (NT - 1) * PITCH) = nt[NT - 1];
  };

  const unsigned vbase = (unsigned)(((wave * TY) * TX + c) * PITCH + g * PLANE);
  f32x4 shv[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) shv[b][r] = a.shift[4 * MB * g + 4 * b + r];
  f32x4 acc[R][MB];
  const int total_steps = a.total_steps;
  __syncthreads();                                  // offset tables visible
  // The tile loads go out BEFORE the weight loads, as in the steady state of the
  // loop below: vmcnt retires in order, and with this order the waits hipcc derives
  // for put() leave the youngest weight loads in flight instead of draining them.
  if (PF) fetch(blk, 0);
  __builtin_amdgcn_sched_barrier(0);                // keep that issue order
  // Weight fragments (MB x 1 KiB per K-step, the same for every wave) come straight
  // from L2 into registers, WQ K-steps ahead of their use and across tile / block
  // boundaries: no LDS ring and no barrier inside the K loop.
  const unsigned char *wl = a.w + lane * 16 + ((UPSP && (wave & 1)) ? a.wstream : 0);
  static_assert((NG * KC) % WQ == 0 && (NGU * KC) % WQ == 0, "the fragment queue's phase is static inside a chunk");
  h16x8 wq[WQ][MB];
#pragma unroll
  for (int d = 0; d < WQ; ++d)
#pragma unroll
    for (int b = 0; b < MB; ++b)
      wq[d][b] = *reinterpret_cast<const h16x8 *>(wl + (size_t)(d * MB + b) * 1024);
  // split: a paired group's rows - hi planes only, lanes g >= 2 at the second group
  const unsigned pbase = (unsigned)(((wave * TY) * TX + c) * PITCH + (g & 1) * PLANE);
  // UPSP (parity form): in an upsampled chunk the row groups (dz', dx) ARE the plain
  // loop's first six - same main / pair sequence - with the dz = 1 plane replaced by the pair's
  // second low-resolution voxel (one or two planes on, by the wave's parity) and the dz = 2
  // groups dropped: `zplane` = byte offset of the "dz = 1" plane for the current chunk
  unsigned zplane = (unsigned)(TY * TX * PITCH);
  auto goff = [&](int k) -> unsigned {
    if (UPSP && k / 3 == 1) return zplane + (unsigned)((k % 3) * PITCH);
    return grp_off(k, TY);
  };
  auto row_addr = [&](int gi, int r) -> const unsigned char * {
    if (grp_pair(gi))
      return tile + pbase + (g >= 2 ? goff(grp_k1(gi)) : goff(grp_k0(gi))) + r * ROW;
    return tile + vbase + goff(grp_k0(gi)) + r * ROW;
  };

  int chunk_base = 0;               // K-steps of the weight stream in front of the current chunk

  for (;;) {
    for (int cc = 0; cc < a.ncc; ++cc) {
      if (cc == 0) {
        chunk_base = 0;
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
          for (int sub = 0; sub < R; ++sub) acc[sub][b] = shv[b];
      }
      __syncthreads();            // every wave has left the previous tile
      if (!PF) fetch(blk, cc);
      put(cc);
      if (PF) {
        const bool last_cc = cc + 1 == a.ncc;
        int64_t nb = last_cc ? blk + G : blk;
        nb = nb < total_blocks ? nb : blk;          // past the end: a harmless reload
        fetch(nb, last_cc ? 0 : cc + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();            // tile visible
      h16x8 brow[2][R + 2];
      const bool ups2 = UPSP && a.src[cc].ups;
      if (UPSP) zplane = ups2 ? (unsigned)(par_row(wave & 1) * TY * TX * PITCH) : (unsigned)(TY * TX * PITCH);
#pragma unroll
      for (int r = 0; r < R + 2; ++r)
        brow[0][r] = *reinterpret_cast<const h16x8 *>(row_addr(0, r));
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        if (UPSP && gi >= NGU && ups2) continue;       // (dz = 2 does not exist there)
        // next row group (wraps to the first; the wrapped read of the last group is unused)
        const int ng = gi + 1 < NG ? gi + 1 : 0;
#pragma unroll
        for (int dy = 0; dy < KC; ++dy) {
          const int st = gi * KC + dy;              // K-step inside the channel chunk
          // spread the R + 2 prefetch reads over the three K-steps
#pragma unroll
          for (int r = (R + 2) * dy / 3; r < (R + 2) * (dy + 1) / 3; ++r)
            brow[(gi + 1) & 1][r] = *reinterpret_cast<const h16x8 *>(row_addr(ng, r));
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int sub = 0; sub < R; ++sub)
#pragma unroll
            for (int b = 0; b < MB; ++b)
              acc[sub][b] = mfma16(wq[st % WQ][b], brow[gi & 1][sub + dy], acc[sub][b]);
          __builtin_amdgcn_s_setprio(0);
          {
            int nxt = chunk_base + st + WQ;
            nxt = nxt < total_steps ? nxt : nxt - total_steps;   // next block starts over
#pragma unroll
            for (int b = 0; b < MB; ++b)
              wq[st % WQ][b] =
                  *reinterpret_cast<const h16x8 *>(wl + ((size_t)nxt * MB + b) * 1024);
          }
        }
      }
      chunk_base += ups2 ? NGU * KC : NG * KC;
    }
    // ---- epilogue: (ReLU) -> bf16, channels-last store
    {
      const int bx = (int)(blk % a.nbx), by = (int)((blk / a.nbx) % a.nby);
      const int bz = (int)(blk / ((int64_t)a.nbx * a.nby));
      const int n = bz / a.zblocks;
      const int oz = (bz % a.zblocks) * 4 + wave;
#pragma unroll
      for (int sub = 0; sub < R; ++sub) {
        const int oy = a.transposed ? bx * 16 + c : by * R + sub;
        const int ox = a.transposed ? a.xorg + by * R + sub : bx * 16 + c;
        if (HEAD) {
          // with interleaved rows lane (c,g) holds channels 8g..8g+7 of voxel c: the
          // packed pair IS the K-step of conv1 32->32 in SLOT_SPATIAL order
          f32x4 a8[2], t9;
          if (SPLIT) {
            // (the guard counts voxels of the layer only: past OD / OH / OW the tile held the
            // source's slack - stale scratch - and the result is never stored)
            unsigned ovs = 0u;
            const Frag2 h7 = pack_relu_split(acc[sub][0], acc[sub][1], ovs);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              f32x4 sh;
#pragma unroll
              for (int r = 0; r < 4; ++r) sh[r] = a.sh8[16 * b + 4 * g + r];
              a8[b] = mfma3(a.w8[b * 64 + lane], a.w8[(2 + b) * 64 + lane], h7, sh);
            }
            t9 = mfma3(a.w9[lane], a.w9[64 + lane], pack_relu_split(a8[0], a8[1], ovs),
                       f32x4{0.f, 0.f, 0.f, 0.f});
            ovf = pk_max_i16(ovf, (oz < a.OD && oy < a.OH && ox < a.OW) ? ovs : 0u);
          } else {
            const h16x8 h7 = pack_relu(acc[sub][0], acc[sub][1]);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              f32x4 sh;
#pragma unroll
              for (int r = 0; r < 4; ++r) sh[r] = a.sh8[16 * b + 4 * g + r];
              a8[b] = mfma16(a.w8[b * 64 + lane], h7, sh);
            }
            t9 = mfma16(a.w9[lane], pack_relu(a8[0], a8[1]), f32x4{0.f, 0.f, 0.f, 0.f});
          }
          const float logit = __shfl(t9[0], c) + a.bias9;     // lane (c, g=0) register 0
          const FplTileDesc td = a.io.tiles[n];
          if (g == 0 && oz < td.ext[0] - 2 * a.io.off && oy < td.ext[1] - 2 * a.io.off &&
              ox < td.ext[2] - 2 * a.io.off && oz < a.OD && oy < a.OH && ox < a.OW)
            a.io.dst[((int64_t)(td.start[0] + a.io.off + oz - a.io.dst_z_base) * a.io.Y +
                      td.start[1] + a.io.off + oy) * a.io.X + td.start[2] + a.io.off + ox] =
                1.f / (1.f + __expf(-logit));
        } else if (oz < a.OD && oy < a.OH && ox < a.OW)
          store_il<MB, false>(a.out + ((((int64_t)n * a.OD + oz) * a.OH + oy) * a.OW + ox) * (a.planar ? 8 : CC),
                              a.oplane, g, acc[sub], a.relu, ovf, a.planar);
      }
    }
    // ---- fused 2x2x2 max pool of the block (4 x 4 x 16 -> 2 x 2 x 8): y pairs are
    // sub-steps of a lane, x pairs neighbouring lanes, z pairs neighbouring waves
    // (through the tile's LDS, free until the next put).  ReLU output is >= 0, so
    // the bf16 order is the int16 order.
    if (POOL && SPLIT) {
      // the same pool in fp32 (the split representation is monotonic: the split of the
      // maximum is the maximum of the splits); ReLU and the hi / lo store at the end
      f32x4 pm[R / 2][MB];
#pragma unroll
      for (int yh = 0; yh < R / 2; ++yh)
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = __builtin_fmaxf(acc[2 * yh][b][r], acc[2 * yh + 1][b][r]);
            pm[yh][b][r] = __builtin_fmaxf(v, __shfl_xor(v, 1));             // x pair (c ^ 1)
          }
      f32x4 *xch = reinterpret_cast<f32x4 *>(tile);          // [wave pair][yh][b][lane]
      __syncthreads();                              // every wave is done with the tile
      if (wave & 1) {
#pragma unroll
        for (int yh = 0; yh < R / 2; ++yh)
#pragma unroll
          for (int b = 0; b < MB; ++b)
            xch[(((wave >> 1) * (R / 2) + yh) * MB + b) * 64 + lane] = pm[yh][b];
      }
      __syncthreads();
      if (!(wave & 1) && !(c & 1)) {
        const int bx = (int)(blk % a.nbx), by = (int)((blk / a.nbx) % a.nby);
        const int bz = (int)(blk / ((int64_t)a.nbx * a.nby));
        const int n = bz / a.zblocks;
        const int PD = a.OD / 2, PH = a.OH / 2, PW = a.OW / 2;
        const int pz = (bz % a.zblocks) * 2 + (wave >> 1), px = bx * 8 + (c >> 1);
#pragma unroll
        for (int yh = 0; yh < R / 2; ++yh) {
          const int py = by * (R / 2) + yh;
          if (pz < PD && py < PH && px < PW) {
            f32x4 m[MB];
#pragma unroll
            for (int b = 0; b < MB; ++b) {
              const f32x4 o = xch[(((wave >> 1) * (R / 2) + yh) * MB + b) * 64 + lane];
#pragma unroll
              for (int r = 0; r < 4; ++r) m[b][r] = __builtin_fmaxf(pm[yh][b][r], o[r]);
            }
            store_il<MB, true>(a.pool_out + ((((int64_t)n * PD + pz) * PH + py) * PW + px) * CC,
                               a.pplane, g, m, 1, ovf);
          }
        }
      }
    }
    if (POOL && !SPLIT) {
      u32x4 pm[R / 2][MB / 2];                      // [y pair][16-B piece]
#pragma unroll
      for (int yh = 0; yh < R / 2; ++yh)
#pragma unroll
        for (int h = 0; h < MB / 2; ++h) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int b = 2 * h + (q >> 1), r0 = (q & 1) * 2;
            const unsigned lo = pk_max_i16(cvt_pk_h16(acc[2 * yh][b][r0], acc[2 * yh][b][r0 + 1]), 0u);
            const unsigned hi = pk_max_i16(cvt_pk_h16(acc[2 * yh + 1][b][r0], acc[2 * yh + 1][b][r0 + 1]), 0u);
            unsigned m = pk_max_i16(lo, hi);
            m = pk_max_i16(m, (unsigned)__shfl_xor((int)m, 1));      // x pair (c ^ 1)
            pm[yh][h][q] = m;
          }
        }
      u32x4 *xch = reinterpret_cast<u32x4 *>(tile);          // [wave pair][yh][h][lane]
      __syncthreads();                              // every wave is done with the tile
      if (wave & 1) {
#pragma unroll
        for (int yh = 0; yh < R / 2; ++yh)
#pragma unroll
          for (int h = 0; h < MB / 2; ++h)
            xch[(((wave >> 1) * (R / 2) + yh) * (MB / 2) + h) * 64 + lane] = pm[yh][h];
      }
      __syncthreads();
      if (!(wave & 1) && !(c & 1)) {
        const int bx = (int)(blk % a.nbx), by = (int)((blk / a.nbx) % a.nby);
        const int bz = (int)(blk / ((int64_t)a.nbx * a.nby));
        const int n = bz / a.zblocks;
        const int PD = a.OD / 2, PH = a.OH / 2, PW = a.OW / 2;
        const int pz = (bz % a.zblocks) * 2 + (wave >> 1), px = bx * 8 + (c >> 1);
#pragma unroll
        for (int yh = 0; yh < R / 2; ++yh) {
          const int py = by * (R / 2) + yh;
          if (pz < PD && py < PH && px < PW) {
            h16_t *vox0 = a.pool_out + ((((int64_t)n * PD + pz) * PH + py) * PW + px) * (a.planar ? 8 : CC);
#pragma unroll
            for (int h = 0; h < MB / 2; ++h) {
              const u32x4 o = xch[(((wave >> 1) * (R / 2) + yh) * (MB / 2) + h) * 64 + lane];
              u32x4 m;
#pragma unroll
              for (int q = 0; q < 4; ++q) m[q] = pk_max_i16(pm[yh][h][q], o[q]);
              const int ch = 4 * MB * g + 8 * h;
              *reinterpret_cast<u32x4 *>(a.planar ? vox0 + (ch / 8) * a.pplane : vox0 + (ch / CC) * a.pplane + ch % CC) = m;
            }
          }
        }
      }
    }
    blk += G;
    if (blk >= total_blocks) break;
  }
  if (SPLIT) {
    ovf_commit(ovf, a.flag, FPL_RANGE_UNET);
    if (STEM && xmax > __builtin_bit_cast(unsigned, a.xlim)) atomicOr(a.flag, FPL_RANGE_INPUT);
  }
}

// ---- 1x1x1 conv as a voxel GEMM ---------------------------------------------------
struct Conv1Args {
  const h16_t *in; int64_t M;   // voxels (n*D*H*W), CIN channels each, in chunk planes of M * 32
  int64_t plane;                 // = M * 32: elements of one chunk plane of `in` and of `out`
  const unsigned char *w;        // [kstep][mb] fragments (SLOT_SPATIAL, 1 tap)
  const float *shift;
  h16_t *out;                   // (M, 16*MB) bf16            (TAIL == 0)
  const h16x8 *w_tail;          // TAIL: [kstep] fragments of the 16*MB -> 1 conv
  float bias_tail;
  float *out_f32;                // TAIL: (M) sigmoid probabilities
  unsigned *flag;                // split build: half-range guard
  int relu;                      // TAIL == 0: ReLU in front of the store (0: a convolution without activation)
  // residual form (gx_exec.h; resnet_like's shortcuts): out = act(crop_ocrop(conv1(x)) + add[.. + acrop]) -
  // the input voxels are n tiles of din^3, the output tensor n tiles of dout^3 = (din - 2 ocrop)^3 (chunk
  // plane `oplane`), the operand n tiles of da^3 (chunk plane `aplane`) read at output coordinates + acrop.
  // add == nullptr: the plain form (out at the input's voxel index, plane `plane`)
  const h16_t *add = nullptr;
  int64_t aplane = 0, oplane = 0;
  int din = 0, dout = 0, da = 0, ocrop = 0, acrop = 0;
};

template <int CIN, int MB, int TAIL>
__global__ __launch_bounds__(256) void FPLK(conv1)(Conv1Args a) {
  static_assert(!SPLIT || TAIL == 0, "the split build writes through the fused head only");
  constexpr int KS = CIN / RCH;                 // K-steps: chunks of 32 physical channels
  constexpr int WMB = PM * MB;                  // split: [w_hi | w_hi] and [w_lo | w_lo] sets
  constexpr int NF = KS * WMB;
  unsigned char *wl = smem;                         // NF KiB of fragments
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  for (int i = tid; i < NF * 64; i += 256)
    reinterpret_cast<u32x4 *>(wl)[i] = reinterpret_cast<const u32x4 *>(a.w)[i];
  f32x4 sh[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      sh[b][r] = a.shift[TAIL ? 16 * b + 4 * g + r : 4 * MB * g + 4 * b + r];
  __syncthreads();
  const int64_t groups = (a.M + 15) / 16;
  unsigned ovf = 0u;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < groups; grp += (int64_t)gridDim.x * 4) {
    int64_t m = grp * 16 + c;
    const bool ok = m < a.M;
    m = ok ? m : a.M - 1;
    h16x8 bf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
      bf[s] = *reinterpret_cast<const h16x8 *>(a.in + s * a.plane + m * CC + 8 * g);
    // (the fragment reads are the same in every trip: without an offset the optimiser cannot
    // see through, they are all hoisted out of the loop - 512 registers and 96 spilled ones
    // in the 128 -> 128 split instance)
    int zero = 0;
    asm volatile("" : "+s"(zero));
    const unsigned char *wq = wl + zero;
    f32x4 acc[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      acc[b] = sh[b];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        acc[b] = mfma16(*reinterpret_cast<const h16x8 *>(wq + ((s * WMB + b) * 64 + lane) * 16),
                        bf[s], acc[b]);
        if (SPLIT)
          acc[b] = mfma16(*reinterpret_cast<const h16x8 *>(wq + ((s * WMB + MB + b) * 64 + lane) * 16),
                          bf[s], acc[b]);
      }
    }
    if (TAIL == 0 && a.add) {
      // the voxel's place in its tile, the output's and the operand's
      int64_t t = m;
      const int x = (int)(t % a.din); t /= a.din;
      const int y = (int)(t % a.din); t /= a.din;
      const int z = (int)(t % a.din); t /= a.din;
      const int ox = x - a.ocrop, oy = y - a.ocrop, oz = z - a.ocrop;
      const bool in = ok && ox >= 0 && oy >= 0 && oz >= 0 && ox < a.dout && oy < a.dout && oz < a.dout;
      if (in) {
        const int64_t ov = ((t * a.dout + oz) * a.dout + oy) * (int64_t)a.dout + ox;
        const int64_t av = ((t * a.da + oz + a.acrop) * a.da + oy + a.acrop) * (int64_t)a.da + ox + a.acrop;
#pragma unroll
        for (int h = 0; h < MB / 2; ++h) {               // the lane's channels [4 MB g + 8 h, + 8): one 16-B piece (and its lo)
          const int ch = 4 * MB * g + 8 * h;
          const h16_t *pa = a.add + (ch / RCH) * a.aplane + av * CC + ch % RCH;
          const h16x8 hi = *reinterpret_cast<const h16x8 *>(pa);
          h16x8 lo = hi;
          if (SPLIT) lo = *reinterpret_cast<const h16x8 *>(pa + 16);
#pragma unroll
          for (int q = 0; q < 8; ++q)
            acc[2 * h + (q >> 2)][q & 3] += SPLIT ? (float)hi[q] + (float)lo[q] : (float)hi[q];
        }
        store_il<MB, false>(a.out + ov * CC, a.oplane, g, acc, a.relu, ovf);
      }
    } else if (TAIL == 0) {
      if (ok) store_il<MB, false>(a.out + m * CC, a.plane, g, acc, a.relu, ovf);
    } else {
      // chained 16*MB -> 1 conv (k-slots bound to the accumulator layout), sigmoid
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < MB / 2; ++s)
        t = mfma16(a.w_tail[s * 64 + lane], pack_relu(acc[2 * s], acc[2 * s + 1]), t);
      const float logit = __shfl(t[0], c) + a.bias_tail;
      if (ok && g == 0) a.out_f32[m] = 1.f / (1.f + __expf(-logit));
    }
  }
  if (SPLIT) ovf_commit(ovf, a.flag, FPL_RANGE_UNET);
}

// ---- unet_like's first stage (fplmodels.py:210-256): conv3 1->32 +BN+ReLU, conv1 32->32
// +BN+ReLU chained in registers (with interleaved rows the packed output of the first is
// the K-step of the second), stored as c1 and max-pooled to p1.  Block 4 x 4 x 16 voxels,
// wave = z, sub-steps = y, lanes = x; the raw 6 x 6 x 18 tile sits in LDS as 16-bit.
struct StemC1Args {
  const float *raw; int T;       // (n, T, T, T) f32 normalised tiles
  const h16x8 *wstem;            // 2 fragments, k-slot (g,j) = tap 8g + j, interleaved rows
  const float *shstem;
  const h16x8 *w1;               // 2 fragments of conv1 32->32 (SLOT_SPATIAL, interleaved rows)
  const float *sh1;
  h16_t *c1, *p1;                // (n, D, D, D, 32), D = T - 2; (n, D/2, D/2, D/2, 32)
  int64_t c1plane, p1plane;      // split: elements of one chunk plane of c1 / p1
  int D, zblocks, nbx, nby;
  unsigned *flag;                // split build: half-range guard (every split is checked here)
};

__global__ __launch_bounds__(256) void FPLK(unet_stem_c1)(StemC1Args a) {
  constexpr int RZ = 6, RY = 6, RX = 18;
  __shared__ unsigned short rawt[PM * RZ * RY * RX];        // split: hi, then lo
  __shared__ u32x4 xch[(SPLIT ? 2 : 1) * 2 * 2 * 64];      // [wave pair][y half]([b])[lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int bx = blockIdx.x % a.nbx, by = (blockIdx.x / a.nbx) % a.nby, bz = blockIdx.x / (a.nbx * a.nby);
  const int n = bz / a.zblocks, z0 = (bz % a.zblocks) * 4, y0 = by * 4, x0 = bx * 16;
  const float *base = a.raw + (int64_t)n * a.T * a.T * a.T;
  unsigned ovf = 0u;
  for (int p = tid; p < RZ * RY * RX; p += 256) {
    int z = z0 + p / (RY * RX), y = y0 + (p / RX) % RY, x = x0 + p % RX;
    z = z < a.T ? z : a.T - 1;                             // clamped reads only feed masked
    y = y < a.T ? y : a.T - 1;                             // outputs
    x = x < a.T ? x : a.T - 1;
    const float v = base[((int64_t)z * a.T + y) * a.T + x];
    rawt[p] = h16_bits(v);
    if (SPLIT) ovf_note(ovf, (unsigned)h16_bits(v) & 0x7FFFu);
    if (SPLIT) rawt[RZ * RY * RX + p] = h16_bits(v - (float)(h16_t)v);
  }
  int toff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int t = 8 * g + j;
    toff[j] = t < 27 ? ((t / 9) * RY + (t / 3) % 3) * RX + t % 3 : 0;
  }
  // split: [part][b] fragment sets (hi parts, then lo parts)
  h16x8 ws[2], w1[2], wsl[2], w1l[2];
  f32x4 shs[2], sh1[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    ws[b] = a.wstem[b * 64 + lane];
    w1[b] = a.w1[b * 64 + lane];
    if (SPLIT) {
      wsl[b] = a.wstem[(2 + b) * 64 + lane];
      w1l[b] = a.w1[(2 + b) * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      shs[b][r] = a.shstem[8 * g + 4 * b + r];
      sh1[b][r] = a.sh1[8 * g + 4 * b + r];
    }
  }
  __syncthreads();
  f32x4 acc[4][2];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    const int ro = (wave * RY + sub) * RX + c;
    u16x8 rw;
#pragma unroll
    for (int j = 0; j < 8; ++j) rw[j] = rawt[ro + toff[j]];
    const h16x8 bf = __builtin_bit_cast(h16x8, rw);
    if (SPLIT) {
      u16x8 rl;
#pragma unroll
      for (int j = 0; j < 8; ++j) rl[j] = rawt[RZ * RY * RX + ro + toff[j]];
      Frag2 b2;
      b2.hi = bf;
      b2.lo = __builtin_bit_cast(h16x8, rl);
      const Frag2 h0 = pack_relu_split(mfma3(ws[0], wsl[0], b2, shs[0]), mfma3(ws[1], wsl[1], b2, shs[1]), ovf);
      acc[sub][0] = mfma3(w1[0], w1l[0], h0, sh1[0]);
      acc[sub][1] = mfma3(w1[1], w1l[1], h0, sh1[1]);
    } else {
      const h16x8 h0 = pack_relu(mfma16(ws[0], bf, shs[0]), mfma16(ws[1], bf, shs[1]));
      acc[sub][0] = mfma16(w1[0], h0, sh1[0]);
      acc[sub][1] = mfma16(w1[1], h0, sh1[1]);
    }
    const int oz = z0 + wave, oy = y0 + sub, ox = x0 + c;
    if (oz < a.D && oy < a.D && ox < a.D)
      store_il<2, true>(a.c1 + ((((int64_t)n * a.D + oz) * a.D + oy) * a.D + ox) * 32, a.c1plane, g, acc[sub], 1, ovf);
  }
  if (SPLIT) {
    // the pool in fp32 (as the POOL epilogue of the split conv3), then split and store
    f32x4 pmf[2][2];
#pragma unroll
    for (int yh = 0; yh < 2; ++yh)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = __builtin_fmaxf(acc[2 * yh][b][r], acc[2 * yh + 1][b][r]);
          pmf[yh][b][r] = __builtin_fmaxf(v, __shfl_xor(v, 1));
        }
    f32x4 *xf = reinterpret_cast<f32x4 *>(xch);            // [wave pair][yh][b][lane]
    if (wave & 1) {
#pragma unroll
      for (int yh = 0; yh < 2; ++yh)
#pragma unroll
        for (int b = 0; b < 2; ++b) xf[(((wave >> 1) * 2 + yh) * 2 + b) * 64 + lane] = pmf[yh][b];
    }
    __syncthreads();
    if (!(wave & 1) && !(c & 1)) {
      const int PD = a.D / 2;
      const int pz = (bz % a.zblocks) * 2 + (wave >> 1), px = bx * 8 + (c >> 1);
#pragma unroll
      for (int yh = 0; yh < 2; ++yh) {
        const int py = by * 2 + yh;
        if (pz < PD && py < PD && px < PD) {
          f32x4 m[2];
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const f32x4 o = xf[(((wave >> 1) * 2 + yh) * 2 + b) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) m[b][r] = __builtin_fmaxf(pmf[yh][b][r], o[r]);
          }
          store_il<2, true>(a.p1 + ((((int64_t)n * PD + pz) * PD + py) * PD + px) * 32, a.p1plane, g, m, 1, ovf);
        }
      }
    }
    ovf_commit(ovf, a.flag, FPL_RANGE_UNET);
    return;
  }
  // 2x2x2 max pool of the block, as the POOL epilogue of conv3
  u32x4 pm[2];
#pragma unroll
  for (int yh = 0; yh < 2; ++yh)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = q >> 1, r0 = (q & 1) * 2;
      const unsigned lo = pk_max_i16(cvt_pk_h16(acc[2 * yh][b][r0], acc[2 * yh][b][r0 + 1]), 0u);
      const unsigned hi = pk_max_i16(cvt_pk_h16(acc[2 * yh + 1][b][r0], acc[2 * yh + 1][b][r0 + 1]), 0u);
      unsigned m = pk_max_i16(lo, hi);
      m = pk_max_i16(m, (unsigned)__shfl_xor((int)m, 1));
      pm[yh][q] = m;
    }
  if (wave & 1) {
    xch[((wave >> 1) * 2 + 0) * 64 + lane] = pm[0];
    xch[((wave >> 1) * 2 + 1) * 64 + lane] = pm[1];
  }
  __syncthreads();
  if (!(wave & 1) && !(c & 1)) {
    const int PD = a.D / 2;
    const int pz = (bz % a.zblocks) * 2 + (wave >> 1), px = bx * 8 + (c >> 1);
#pragma unroll
    for (int yh = 0; yh < 2; ++yh) {
      const int py = by * 2 + yh;
      if (pz < PD && py < PD && px < PD) {
        const u32x4 o = xch[((wave >> 1) * 2 + yh) * 64 + lane];
        u32x4 m;
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = pk_max_i16(pm[yh][q], o[q]);
        *reinterpret_cast<u32x4 *>(a.p1 + ((((int64_t)n * PD + pz) * PD + py) * PD + px) * 32 + 8 * g) = m;
      }
    }
  }
}

// MaxPooling3D(2) of a non-negative (post-ReLU) 16-bit channels-last tensor: a thread takes
// 8 channels (16 B) of one window; 16-bit order = int16 order for non-negative values
__global__ void FPLK(pool2_h16)(const u32x4 *__restrict__ x, u32x4 *__restrict__ y, int64_t n_out,
                                int D, int C8, int od) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  int64_t t = i;
  const int c = (int)(t % C8); t /= C8;
  const int ox = (int)(t % od); t /= od;
  const int oy = (int)(t % od); t /= od;
  const int oz = (int)(t % od); t /= od;
  if (SPLIT) {
    // a chunk plane holds [hi 16 | lo 16] per voxel (C8 = 4): pieces 0, 1 = hi halves of real
    // channels 0-7 / 8-15, pieces 2, 3 their lo halves.  Threads c = 0, 1 pool 8 channels each
    // in fp32 and write the hi piece c and the lo piece c + 2; c = 2, 3 have nothing to do.
    if (c >= 2) return;
    float best[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) best[q] = 0.f;             // post-ReLU values are >= 0
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int64_t vox = (((t * D + 2 * oz + (p >> 2)) * D + 2 * oy + ((p >> 1) & 1)) * (int64_t)D +
                           2 * ox + (p & 1)) * C8;
      const h16x8 hi = __builtin_bit_cast(h16x8, x[vox + c]), lo = __builtin_bit_cast(h16x8, x[vox + 2 + c]);
#pragma unroll
      for (int q = 0; q < 8; ++q) best[q] = __builtin_fmaxf(best[q], (float)hi[q] + (float)lo[q]);
    }
    u32x4 oh, ol;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const Pair2 pr = split_pk(best[2 * q], best[2 * q + 1]);
      oh[q] = pr.hi; ol[q] = pr.lo;
    }
    y[(i / C8) * C8 + c] = oh;
    y[(i / C8) * C8 + 2 + c] = ol;
    return;
  }
  u32x4 m = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const u32x4 v = x[((((t * D + 2 * oz + (p >> 2)) * D + 2 * oy + ((p >> 1) & 1)) * (int64_t)D +
                       2 * ox + (p & 1)) * C8) + c];
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = pk_max_i16(m[q], v[q]);
  }
  y[i] = m;
}

// ---- host: unet_like2 pattern + packed weights -------------------------------------
struct UnetState {
  uint64_t version = ~0ull;
  unsigned char *frags = nullptr;
  float *shifts = nullptr;
  size_t off_w[12] = {0}, off_s[12] = {0};
  size_t half_bytes[12] = {0};   // conv3 with 128 outputs: bytes of the first 64-channel half
  size_t off_w7t = 0;            // conv 7 with the dy / dx taps swapped (edge strip)
  // split build, parity form of the two convolutions that read an UpSampling3D source:
  // [0] = conv3 192->64 (l_up1), [1] = conv3 96->32 (l_up2): both weight streams, one after the other
  size_t off_wp[2] = {0, 0}, wp_stream[2] = {0, 0};
  int wp_steps[2] = {0, 0};
  float bias_tail = 0.f;
  float xlim = 0.f;              // split build: input limit of the stem's half-range bound
  // split build, all-LDS kernels (unet_split_lds.h): the planar-pass weight streams
  unsigned char *frags8 = nullptr;
  size_t off8[12] = {0}, half8[12] = {0};   // per conv; 128-output layers: bytes of the first 64-channel half
  size_t off8t = 0;                         // the head's conv3 with dy / dx swapped (edge strip)
  bool have8 = false;
};

void unet_state_free(fpl_ctx *, void *p) {
  UnetState *s = (UnetState *)p;
  if (s->frags) hipFree(s->frags);
  if (s->shifts) hipFree(s->shifts);
  if (s->frags8) hipFree(s->frags8);
  delete s;
}

// The U-Net skeleton of fplmodels.py:210-407 (unet_like, unet_like2, unet_like3, unet_like4;
// unet_like's second conv of stages 1 and 2 is 1x1):
//   conv3 1->32, conv3 32->32, pool, conv3 32->64, conv3 64->64, pool, BOTTOM,
//   up, [crop skip2], concat, conv3 192->64, conv1 64->64, up, crop skip1, concat,
//   conv3 96->32, conv1 32->32, conv1 32->1 (sigmoid)
// with BOTTOM = conv1 64->128 | conv3 64->128, conv1 128->128 | conv3 64->128, conv3 128->128
// (the crop may be emitted before or after the up - matched by kind counts, conv order
// and data flow).  Conv roles: 0,1 stage 1; 2,3 stage 2; then the bottom; then
// up1 (conv3, conv1), up2 (conv3, conv1), head.
struct UnetDesc {
  int nconv = 0;
  int conv[12];                  // op index per conv, in order
  int nbottom = 0;               // 1 or 2
  int crop2 = 0, crop1 = 0;      // crop of the stage-2 / stage-1 skip
  bool first1 = false, second1 = false;   // unet_like: the second conv of stage 1 / 2 is 1x1
  int l_up1() const { return 4 + nbottom; }      // conv3 192->64
  int l_up2() const { return 6 + nbottom; }      // conv3 96->32
};

bool match_unet(const fpl_program *prog, UnetDesc *d) {
  int np = 0, nu = 0, ncat = 0, ncrop = 0;
  int crops[2] = {0, 0};
  d->nconv = 0;
  for (size_t i = 0; i < prog->ops.size(); ++i) {
    const fpl_op &op = prog->ops[i];
    switch (op.kind) {
      case FPL_OP_CONV:
        if (d->nconv >= 12) return false;
        d->conv[d->nconv++] = (int)i;
        break;
      case FPL_OP_POOL: if (op.p[0] != 2 || op.p[1] != 2 || op.p[2] != 2) return false; ++np; break;
      case FPL_OP_UP: if (op.p[0] != 2 || op.p[1] != 2 || op.p[2] != 2) return false; ++nu; break;
      case FPL_OP_CONCAT: ++ncat; break;
      case FPL_OP_CROP:
        if (ncrop >= 2) return false;
        for (int q = 1; q < 6; ++q) if (op.p[q] != op.p[0]) return false;
        if (op.p[0] <= 0 || op.p[0] % 2) return false;
        crops[ncrop++] = op.p[0];
        break;
      default: return false;
    }
  }
  if (np != 2 || nu != 2 || ncat != 2 || ncrop < 1) return false;
  if (prog->stride[0] != 1 || prog->stride[1] != 1 || prog->stride[2] != 1) return false;
  d->nbottom = d->nconv - 9;
  if (d->nbottom != 1 && d->nbottom != 2) return false;
  auto C = [&](int l) -> const fpl_op & { return prog->ops[d->conv[l]]; };
  auto is = [&](int l, int k, int cin, int cout) {
    return C(l).k == k && C(l).cin == cin && C(l).cout == cout;
  };
  if (!is(0, 3, 1, 32) || !is(2, 3, 32, 64)) return false;
  d->first1 = is(1, 1, 32, 32);
  d->second1 = is(3, 1, 64, 64);
  if (!d->first1 && !is(1, 3, 32, 32)) return false;
  if (!d->second1 && !is(3, 3, 64, 64)) return false;
  if (d->nbottom == 1) {
    if (!is(4, 1, 64, 128)) return false;
  } else {
    if (!is(4, 3, 64, 128)) return false;
    if (!is(5, 1, 128, 128) && !is(5, 3, 128, 128)) return false;
  }
  const int u1 = d->l_up1(), u2 = d->l_up2();
  if (!is(u1, 3, 192, 64) || !is(u1 + 1, 1, 64, 64) || !is(u2, 3, 96, 32) || !is(u2 + 1, 1, 32, 32) ||
      !is(u2 + 2, 1, 32, 1))
    return false;
  for (int l = 0; l < d->nconv; ++l)
    if (C(l).act != (l == d->nconv - 1 ? FPL_ACT_SIGMOID : FPL_ACT_RELU)) return false;
  // crops: the stage-2 skip's (if any) is created first (fplmodels.py:283-291)
  d->crop2 = ncrop == 2 ? crops[0] : 0;
  d->crop1 = ncrop == 2 ? crops[1] : crops[0];
  // data flow: conv 2 and the first bottom conv read pools, the two up convs read concats
  const auto &o = prog->ops;
  auto src_kind = [&](int tensor) -> int {
    for (auto &op : o) if (op.dst == tensor) return op.kind;
    return -1;
  };
  if (src_kind(C(2).src0) != FPL_OP_POOL || src_kind(C(4).src0) != FPL_OP_POOL) return false;
  if (src_kind(C(u1).src0) != FPL_OP_CONCAT || src_kind(C(u2).src0) != FPL_OP_CONCAT) return false;
  for (int l : {1, 3, u1 + 1, u2 + 1, u2 + 2})
    if (C(l).src0 != C(l - 1).dst) return false;
  if (d->nbottom == 2 && C(5).src0 != C(4).dst) return false;
  return prog->out_tensor == C(d->nconv - 1).dst;
}

// conv3 fragments: per CC-channel chunk, K-step order (dz, dx, dy) (or (dz, dy, dx) for
// the transposed edge strip); output channels [co0, co0 + ncout) as one M-block set
// Split build: a chunk is 16 real channels as 32 physical k-slots [hi | lo]; the K-steps of
// a main row group carry the SAME real channel's w_hi in both halves, those of a paired
// group the w_lo of group k in the lower and of group k' in the upper half (NG above):
// [chunk][K-step of the 42][mb].
void pack_conv3(const float *A, const fpl_op &op, int co0, int ncout, bool transposed,
                std::vector<uint16_t> *f) {
  const int ncc = op.cin / RCH, mb = (ncout + 15) / 16;
  const int nst = NG * KC;
  std::vector<float> sub((size_t)nst * CC * ncout), scale(A + op.scale_off + co0, A + op.scale_off + co0 + ncout);
  f->clear();
  auto tap_of = [&](int k, int dy) {               // row group k = (dz, dx)
    const int dz = k / 3, d1 = k % 3;
    return transposed ? dz * 9 + d1 * 3 + dy : dz * 9 + dy * 3 + d1;
  };
  for (int cc = 0; cc < ncc; ++cc) {
    std::fill(sub.begin(), sub.end(), 0.f);
    for (int gi = 0; gi < NG; ++gi)
      for (int dy = 0; dy < KC; ++dy)
        for (int ch = 0; ch < CC; ++ch) {
          // k-slot ch of this K-step: which tap's weight of which real channel
          int k = grp_k0(gi);
          if (grp_pair(gi) && ch >= RCH) {
            if (gi == NG - 1) continue;            // the unpaired ninth group: upper half 0
            k = grp_k1(gi);
          }
          memcpy(&sub[((size_t)(gi * KC + dy) * CC + ch) * ncout],
                 A + op.w_off + ((size_t)tap_of(k, dy) * op.cin + cc * RCH + ch % RCH) * op.cout + co0,
                 ncout * sizeof(float));
        }
    std::vector<uint16_t> fc[2];
    for (int part = 0; part < PM; ++part)
      fpl_pack_frags(sub.data(), scale.data(), nst, CC, ncout, mb, nst, SLOT_SPATIAL, &fc[part], true, part);
    for (int gi = 0; gi < NG; ++gi)
      for (int dy = 0; dy < KC; ++dy) {
        const int ks = gi * KC + dy, part = grp_pair(gi) ? 1 : 0;     // pairs carry the lo parts
        f->insert(f->end(), fc[part].begin() + (size_t)ks * mb * 512,
                  fc[part].begin() + (size_t)(ks + 1) * mb * 512);
      }
  }
}

// The parity form's two weight streams: stream pz = what a wave of z parity pz
// reads, chunk after chunk - the first `n_ups` chunks (the UpSampling3D source) as NGU x KC steps
// [group][dy], the rest as pack_conv3's NG x KC.  The pre-summed z weights are formed in fp32 from the
// layer's own, BN scale and the hi / lo split applied afterwards by fpl_pack_frags.
// *steps = K-steps of one stream.
void pack_conv3_parity(const float *A, const fpl_op &op, int ncout, int n_ups, std::vector<uint16_t> *f,
                       int *steps) {
  const int ncc = op.cin / RCH, mb = (ncout + 15) / 16;
  std::vector<float> scale(A + op.scale_off, A + op.scale_off + ncout);
  std::vector<uint16_t> plain;
  pack_conv3(A, op, 0, ncout, false, &plain);                    // [chunk][42][mb] fragments
  const size_t plain_chunk = (size_t)NG * KC * mb * 512;
  // z taps that fall on the pair's first / second low-resolution plane, by parity
  auto taps = [](int par, int second, int *t) {                  // returns the count
    if (par == 0) { if (!second) { t[0] = 0; t[1] = 1; return 2; } t[0] = 2; return 1; }
    if (!second) { t[0] = 0; return 1; }
    t[0] = 1; t[1] = 2; return 2;
  };
  const int nstu = NGU * KC;
  std::vector<float> sub((size_t)nstu * CC * ncout);
  f->clear();
  *steps = n_ups * nstu + (ncc - n_ups) * NG * KC;
  for (int pz = 0; pz < 2; ++pz)
    for (int cc = 0; cc < ncc; ++cc) {
      if (cc >= n_ups) {
        f->insert(f->end(), plain.begin() + cc * plain_chunk, plain.begin() + (cc + 1) * plain_chunk);
        continue;
      }
      std::fill(sub.begin(), sub.end(), 0.f);
      for (int gi = 0; gi < NGU; ++gi)
        for (int dy = 0; dy < KC; ++dy)
          for (int ch = 0; ch < CC; ++ch) {
            const int k = (grp_pair(gi) && ch >= RCH) ? grp_k1(gi) : grp_k0(gi);   // (dz', dx)
            int tz[2];
            const int nz = taps(pz, k / 3, tz);
            float *dst = &sub[((size_t)(gi * KC + dy) * CC + ch) * ncout];
            for (int iz = 0; iz < nz; ++iz) {
              const float *w = A + op.w_off +
                               ((size_t)(tz[iz] * 9 + dy * 3 + k % 3) * op.cin + cc * RCH + ch % RCH) * op.cout;
              for (int co = 0; co < ncout; ++co) dst[co] += w[co];
            }
          }
      std::vector<uint16_t> fc[2];
      for (int part = 0; part < PM; ++part)
        fpl_pack_frags(sub.data(), scale.data(), nstu, CC, ncout, mb, nstu, SLOT_SPATIAL, &fc[part], true, part);
      for (int gi = 0; gi < NGU; ++gi)
        for (int dy = 0; dy < KC; ++dy) {
          const int ks = gi * KC + dy, part = grp_pair(gi) ? 1 : 0;
          f->insert(f->end(), fc[part].begin() + (size_t)ks * mb * 512,
                    fc[part].begin() + (size_t)(ks + 1) * mb * 512);
        }
    }
}

// 1x1x1 conv on a (physical) channels-last row: K-steps over chunks of 32 physical
// channels; split: [K-step][set][mb] as pack_conv3
void pack_conv1(const float *A, const fpl_op &op, bool il, std::vector<uint16_t> *f) {
  const int ks = op.cin / RCH, mb = (op.cout + 15) / 16;
  std::vector<float> w((size_t)ks * CC * op.cout), scale(A + op.scale_off, A + op.scale_off + op.cout);
  for (int k = 0; k < ks * CC; ++k)
    memcpy(&w[(size_t)k * op.cout], A + op.w_off + (size_t)((k / CC) * RCH + (k % CC) % RCH) * op.cout,
           op.cout * sizeof(float));
  std::vector<uint16_t> fc[2];
  for (int part = 0; part < PM; ++part)
    fpl_pack_frags(w.data(), scale.data(), 1, ks * CC, op.cout, mb, ks, SLOT_SPATIAL, &fc[part], il, part);
  f->clear();
  for (int s = 0; s < ks; ++s)
    for (int part = 0; part < PM; ++part)
      f->insert(f->end(), fc[part].begin() + (size_t)s * mb * 512, fc[part].begin() + (size_t)(s + 1) * mb * 512);
}

int unet_prepare(fpl_ctx *ctx, fpl_program *prog, const UnetDesc &d, UnetState **out) {
  UnetState *st = (UnetState *)prog->fast_state_h16[FPL_H16_SLOT];
  if (!st) {
    st = new UnetState();
    prog->fast_state_h16[FPL_H16_SLOT] = st;
    prog->fast_state_h16_free[FPL_H16_SLOT] = unet_state_free;
  }
  *out = st;
  if (st->version == prog->arena_version) return 0;
  std::vector<uint16_t> all;
  std::vector<float> shifts;
  const float *A = prog->arena_host.data();
  const int l_up2 = d.l_up2(), l_last = d.nconv - 1;
  st->off_w7t = 0;
  st->wp_steps[0] = st->wp_steps[1] = 0;
  for (int l = 0; l < d.nconv; ++l) {
    const fpl_op &op = prog->ops[d.conv[l]];
    std::vector<float> scale(A + op.scale_off, A + op.scale_off + op.cout);
    std::vector<uint16_t> f;
    const int mb = (op.cout + 15) / 16;
    if (SPLIT && d.first1 && l <= 1) {
      // unet_like's chained stem (unet_stem_c1): conv3 1->32 and conv1 32->32 with interleaved
      // rows and REAL channels as k-slots, hi parts then lo parts: [part][b]
      for (int part = 0; part < 2; ++part) {
        std::vector<uint16_t> fp;
        if (l == 0) fpl_pack_frags(A + op.w_off, scale.data(), 27, 1, op.cout, mb, 1, SLOT_STEM, &fp, true, part);
        else fpl_pack_frags(A + op.w_off, scale.data(), 1, op.cin, op.cout, mb, op.cin / 32, SLOT_SPATIAL, &fp, true, part);
        f.insert(f.end(), fp.begin(), fp.end());
      }
    } else if (l == 0 && SPLIT) {
      // conv3 1->32 per chunk of 16 output channels, plain rows: [chunk][part]
      for (int cc = 0; cc < op.cout / 16; ++cc) {
        std::vector<float> wc((size_t)27 * 16);
        for (int t = 0; t < 27; ++t)
          memcpy(&wc[(size_t)t * 16], A + op.w_off + (size_t)t * op.cout + 16 * cc, 16 * sizeof(float));
        for (int part = 0; part < 2; ++part) {
          std::vector<uint16_t> fp;
          fpl_pack_frags(wc.data(), scale.data() + 16 * cc, 27, 1, 16, 1, 1, SLOT_STEM, &fp, false, part);
          f.insert(f.end(), fp.begin(), fp.end());
        }
      }
    } else if (l == 0) {
      fpl_pack_frags(A + op.w_off, scale.data(), 27, 1, op.cout, mb, 1, SLOT_STEM, &f, true);
    } else if (op.k == 3 && op.cout > 64) {
      // 128 output channels: two 64-channel launches, their fragment sets back to back
      FPL_REQUIRE(ctx, op.cout == 128, "unet: conv3 with %d output channels", op.cout);
      std::vector<uint16_t> h0, h1;
      pack_conv3(A, op, 0, 64, false, &h0);
      pack_conv3(A, op, 64, 64, false, &h1);
      st->half_bytes[l] = h0.size() * sizeof(uint16_t);
      f = h0;
      f.insert(f.end(), h1.begin(), h1.end());
    } else if (op.k == 3) {
      pack_conv3(A, op, 0, op.cout, false, &f);
      if (l == d.l_up1() || l == l_up2) {
        // parity forms: the leading chunks of the concatenated input are the upsampled source
        const int w = l == l_up2, n_ups = (l == l_up2 ? 64 : 128) / RCH;
        if (op.cin / RCH > n_ups) {
          std::vector<uint16_t> fp;
          pack_conv3_parity(A, op, op.cout, n_ups, &fp, &st->wp_steps[w]);
          st->off_wp[w] = all.size() * sizeof(uint16_t);
          st->wp_stream[w] = fp.size() / 2 * sizeof(uint16_t);
          all.insert(all.end(), fp.begin(), fp.end());
        }
      }
      if (l == l_up2) {          // the transposed edge strip swaps the roles of dy and dx
        std::vector<uint16_t> ft;
        pack_conv3(A, op, 0, op.cout, true, &ft);
        st->off_w7t = all.size() * sizeof(uint16_t);
        all.insert(all.end(), ft.begin(), ft.end());
      }
    } else if (l == l_last) {
      for (int part = 0; part < PM; ++part) {          // split: [part]
        std::vector<uint16_t> fp;
        fpl_pack_frags(A + op.w_off, scale.data(), 1, op.cin, op.cout, 1, 1, SLOT_CHAIN, &fp, false, part);
        f.insert(f.end(), fp.begin(), fp.end());
      }
    } else if (SPLIT && l == l_last - 1) {
      // conv1 32->32 in the head epilogue: its B fragments are built in registers from the
      // accumulators (hi and lo), REAL channels as k-slots: [part][b], three products
      for (int part = 0; part < 2; ++part) {
        std::vector<uint16_t> fp;
        fpl_pack_frags(A + op.w_off, scale.data(), 1, op.cin, op.cout, mb, op.cin / 32, SLOT_SPATIAL,
                       &fp, false, part);
        f.insert(f.end(), fp.begin(), fp.end());
      }
    } else if (SPLIT) {
      pack_conv1(A, op, true, &f);
    } else {
      // the conv1 before the head feeds the register-chained tail: plain row order there
      fpl_pack_frags(A + op.w_off, scale.data(), 1, op.cin, op.cout, mb, op.cin / 32, SLOT_SPATIAL, &f,
                     l != l_last - 1);
    }
    st->off_w[l] = all.size() * sizeof(uint16_t);
    all.insert(all.end(), f.begin(), f.end());
    st->off_s[l] = shifts.size();
    shifts.insert(shifts.end(), A + op.shift_off, A + op.shift_off + op.cout);
    while (shifts.size() % 4) shifts.push_back(0.f);
  }
#ifdef FPL_F16
  for (uint16_t h : all)
    if ((h & 0x7C00u) == 0x7C00u) {
      const char *msg = "a folded weight exceeds the IEEE-half range (65504); use precision "
                        "bf16, f32 or 'auto' for this network";
      return SPLIT ? fpl_fail_range(ctx, "%s", msg) : fpl_fail(ctx, "%s", msg);
    }
#endif
  if (SPLIT) {
    // half-range guard of the stem computed in the tile loader: conv3 1->32's outputs stay
    // below the limit for raw inputs |x| <= xlim (checked per raw voxel by the kernel)
    const fpl_op &op = prog->ops[d.conv[0]];
    double xl = 65000.0;
    for (int co = 0; co < op.cout; ++co) {
      double sw = 0.0;
      for (int tap = 0; tap < 27; ++tap) sw += std::fabs((double)A[op.w_off + (size_t)tap * op.cout + co]);
      sw *= std::fabs((double)A[op.scale_off + co]);
      const double sh = std::fabs((double)A[op.shift_off + co]);
      if (!(sh < 65000.0)) return fpl_fail_range(ctx, "the first layer's shift exceeds the IEEE-half range");
      if (sw > 0.0) xl = std::min(xl, (65000.0 - sh) / sw);
    }
    st->xlim = (float)xl;
  }
  st->bias_tail = A[prog->ops[d.conv[l_last]].shift_off];
  if (st->frags) FPL_HIP(ctx, hipFree(st->frags));
  if (st->shifts) FPL_HIP(ctx, hipFree(st->shifts));
  st->frags = nullptr; st->shifts = nullptr;
  FPL_HIP(ctx, hipMalloc((void **)&st->frags, all.size() * sizeof(uint16_t)));
  FPL_HIP(ctx, hipMalloc((void **)&st->shifts, shifts.size() * sizeof(float)));
  FPL_HIP(ctx, hipMemcpy(st->frags, all.data(), all.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  FPL_HIP(ctx, hipMemcpy(st->shifts, shifts.data(), shifts.size() * sizeof(float), hipMemcpyHostToDevice));
  st->version = prog->arena_version;
  st->have8 = false;              // (the all-LDS path repacks its own streams on first use)
  return 0;
}

template <int MB, bool STEM = false, bool POOL = false, bool HEAD = false, int R = 4, bool UPSP = false>
int launch_conv3(fpl_ctx *ctx, Conv3Args &a, int n, const char *name) {
  constexpr bool PF = true;
  if (!UPSP) { a.parity = 0; a.wstream = 0; a.total_steps = a.ncc * NG * KC; }
  FPL_REQUIRE(ctx, UPSP == (a.parity != 0) && (!UPSP || !a.transposed), "conv3: parity form / template mismatch");
  typedef Geo<R> GE;
  // STEM keeps the bf16 raw tile behind the (single) offset table
  constexpr int SMEM = GE::TILE_BYTES + (STEM ? GE::TABN * 4 + GE::NRAW * 2 * PM : GE::TAB_BYTES);
  static_assert(2 * SMEM <= 160 * 1024, "two conv3 workgroups must fit one CU");
  // function attributes belong to the current device: one flag per device (a process may
  // drive several GPUs, one context each; setting it twice is harmless)
  static bool attr_set[FPL_MAX_DEVICES] = {false};
  if (!attr_set[ctx->device % FPL_MAX_DEVICES]) {
    FPL_HIP(ctx, hipFuncSetAttribute((const void *)FPLK(conv3)<MB, PF, STEM, POOL, HEAD, R, UPSP>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_set[ctx->device % FPL_MAX_DEVICES] = true;
  }
  a.oplane = (int64_t)n * a.OD * a.OH * a.OW * (a.planar ? 8 : CC);
  a.pplane = (int64_t)n * (a.OD / 2) * (a.OH / 2) * (a.OW / 2) * (a.planar ? 8 : CC);
  // offset tables: one per distinct source geometry
  a.ntab = 0;
  for (int i = 0; i < (STEM ? 0 : a.ncc); ++i) {
    Src &s = a.src[i];
    FPL_REQUIRE(ctx, !(s.ups && s.crop), "conv3: crop of an upsampled source");
    FPL_REQUIRE(ctx, s.crop % 2 == 0, "conv3: odd crop");
    int t = 0;
    for (; t < a.ntab; ++t)
      if (a.tabH[t] == s.H && a.tabW[t] == s.W && a.tabC[t] == s.C && a.tabU[t] == s.ups) break;
    if (t == a.ntab) {
      FPL_REQUIRE(ctx, a.ntab < MAXTAB, "conv3: more than %d source geometries", MAXTAB);
      a.tabH[t] = s.H; a.tabW[t] = s.W; a.tabC[t] = s.C; a.tabU[t] = s.ups;
      ++a.ntab;
    }
    s.tab = t;
  }
  a.zblocks = (int)ceil_div64(a.OD, 4);
  if (a.transposed) {            // lanes walk y, sub-steps walk x in [xorg, OW)
    a.nbx = (int)ceil_div64(a.OH, 16); a.nby = (int)ceil_div64(a.OW - a.xorg, R);
  } else {
    a.nbx = (int)ceil_div64(a.main_w ? a.main_w : a.OW, 16); a.nby = (int)ceil_div64(a.OH, R);
  }
  a.nbz = n * a.zblocks;
  const int64_t total = (int64_t)a.nbx * a.nby * a.nbz;
  // two workgroups per CU, rounded to a multiple of the 8 XCDs
  int64_t grid = std::min<int64_t>((int64_t)ctx->n_cu * ((MB == 2 && !SPLIT && R == 4) ? 3 : 2), (total + 7) / 8 * 8);
  grid = std::max<int64_t>(8, grid / 8 * 8);
  TimedLaunch tl(ctx, name);
  FPL_REQUIRE(ctx, POOL == (a.pool_out != nullptr) && (!POOL || a.relu),
              "conv3: pool output / template mismatch");
  FPLK(conv3)<MB, PF, STEM, POOL, HEAD, R, UPSP><<<(unsigned)grid, 256, SMEM, ctx->stream>>>(a);
  return 0;
}

Src make_src(const h16_t *p, int dim, int C, int ch0, int up, int crop) {
  Src s;
  s.p = p; s.D = s.H = s.W = dim; s.C = C; s.ch0 = ch0; s.ups = up == 2 ? 1 : 0; s.crop = crop; s.tab = 0;
  return s;
}


// =====================================================================================
// The all-LDS executor for unet_like2 / unet_like3 / unet_like4 (round 5): kernels in
// unet_split_lds.h, tensors as planes of CHP-channel passes (8 channels as hi + lo halves in the
// split build, 16 channels as two planes of 8 in the bf16 / f16 builds).  unet_like (whose second
// convolutions are 1x1x1) stays on the kernels above.
// =====================================================================================
constexpr int CHP = SPLIT ? 8 : 16;      // channels per pass
#include "unet_split_lds.h"

// One pass (input channels [ci0, ci0 + 8)) of a 3x3x3 convolution, output channels [co0, co0 +
// 16 mb), interleaved rows, appended to *f in the order the kernel's LDS-DMA wants it:
//   plain pass (um = 0):  [K-step 0..6][hi b0..mb-1 | lo b0..mb-1]            (phase A = K-steps 0 - 3)
//   upsampled pass:       [phase A, B][parity][K-step][hi | lo]
// K-slot (s, g, j) = tap 4 s + g, channel ci0 + j.  Upsampled passes, um = UM_Z: 18 taps (dz', dy, dx),
// the z weights pre-summed per output-plane parity pz (the parity form above), 5 K-steps (A = 0 - 2),
// parities pz = 0, 1; um = UM_ZY: 12 taps (dz', dy', dx), z AND y weights pre-summed per (plane, row)
// parity, 3 K-steps (A = 0 - 1), parities pz + 2 py.  Sums in fp32, BN scale and the hi / lo split
// applied afterwards.  `transposed`: the dy / dx taps swapped (edge strip: the rows of the tile -
// and with them the row parity - run along x).
void pack_u3_pass(const float *A, const fpl_op &op, int ci0, int co0, int mb, int um, bool transposed,
                  std::vector<uint16_t> *f) {
  const int ncout = 16 * mb;
  std::vector<float> scale(A + op.scale_off + co0, A + op.scale_off + co0 + ncout);
  auto W = [&](int dz, int dy, int dx, int ci, int co) {      // (dy, dx) = tile row / column tap
    const int tap = dz * 9 + (transposed ? dx * 3 + dy : dy * 3 + dx);
    return A[op.w_off + ((size_t)tap * op.cin + ci) * op.cout + co];
  };
  // the two weight sets of a K-step: split build = the hi and the lo parts of the pass's 8 channels,
  // plain builds = its channels 0 - 7 and 8 - 15
  auto pack2 = [&](const std::vector<float> (&sub)[2], int ntap, int K, std::vector<uint16_t> (&fc)[2]) {
    for (int set = 0; set < 2; ++set)
      fpl_pack_frags(sub[SPLIT ? 0 : set].data(), scale.data(), ntap, 8, ncout, mb, K, SLOT_SPATIAL, &fc[set], 1,
                     SPLIT ? set : 0);
  };
  // taps of one axis that fall on the pair's first / second low-resolution voxel, by output parity
  auto fold = [](int par, int second, int *t) {
    if (par == 0) { if (!second) { t[0] = 0; t[1] = 1; return 2; } t[0] = 2; return 1; }
    if (!second) { t[0] = 0; return 1; }
    t[0] = 1; t[1] = 2; return 2;
  };
  const bool zy = um == u8::UM_ZY, ups = um != u8::UM_NONE;
  const int npar = !ups ? 1 : zy ? 4 : 2, ntap = !ups ? 27 : zy ? 12 : 18;
  const int K = !ups ? u8::KP : zy ? 3 : 5, KA = !ups ? u8::KPA : zy ? 2 : 3;
  std::vector<uint16_t> fp[4][2];                              // [parity][set]
  for (int par = 0; par < npar; ++par) {
    const int pz = par & 1, py = par >> 1;
    std::vector<float> sub[2];
    for (int half = 0; half < (SPLIT ? 1 : 2); ++half) {
      sub[half].assign((size_t)ntap * 8 * ncout, 0.f);
      for (int t = 0; t < ntap; ++t) {
        int tz[2], ty[2], nz = 1, ny = 1, dx = t % 3;
        if (!ups) { tz[0] = t / 9; ty[0] = (t / 3) % 3; }
        else {
          const int dzp = zy ? t / 6 : t / 9, dyp = zy ? (t / 3) % 2 : (t / 3) % 3;
          nz = fold(pz, dzp, tz);
          ty[0] = dyp;
          if (zy) ny = fold(py, dyp, ty);
        }
        for (int j = 0; j < 8; ++j)
          for (int co = 0; co < ncout; ++co) {
            float v = 0.f;
            for (int iz = 0; iz < nz; ++iz)
              for (int iy = 0; iy < ny; ++iy) v += W(tz[iz], ty[iy], dx, ci0 + 8 * half + j, co0 + co);
            sub[half][((size_t)t * 8 + j) * ncout + co] = v;
          }
      }
    }
    pack2(sub, ntap, K, fp[par]);
  }
  for (int ph = 0; ph < 2; ++ph)
    for (int par = 0; par < npar; ++par) {
      const int s0 = ph ? KA : 0, s1 = ph ? K : KA;
      for (int s = s0; s < s1; ++s)
        for (int set = 0; set < 2; ++set)
          f->insert(f->end(), fp[par][set].begin() + (size_t)s * mb * 512, fp[par][set].begin() + (size_t)(s + 1) * mb * 512);
    }
}

// the whole stream of a convolution's launch: passes 0 .. cin / 8 - 1, the first n_ups upsampled (mode um)
void pack_u3(const float *A, const fpl_op &op, int co0, int mb, int n_ups, int um, bool transposed, std::vector<uint16_t> *f) {
  f->clear();
  for (int p = 0; p < op.cin / CHP; ++p) pack_u3_pass(A, op, CHP * p, co0, mb, p < n_ups ? um : u8::UM_NONE, transposed, f);
}

// 1x1x1 convolution on planar tensors: [K-step s][hi b0..mb-1 | lo b0..mb-1] (plain builds: [s][b]), K-slot (s, g, j) =
// channel 32 s + 8 g + j, output channels [co0, co0 + 16 mb), interleaved rows
void pack_u1(const float *A, const fpl_op &op, int co0, int mb, std::vector<uint16_t> *f) {
  const int ncout = 16 * mb, ks = op.cin / 32;
  std::vector<float> sub((size_t)op.cin * ncout), scale(A + op.scale_off + co0, A + op.scale_off + co0 + ncout);
  for (int ci = 0; ci < op.cin; ++ci)
    for (int co = 0; co < ncout; ++co) sub[(size_t)ci * ncout + co] = A[op.w_off + (size_t)ci * op.cout + co0 + co];
  std::vector<uint16_t> fc[2];
  for (int part = 0; part < PM; ++part)
    fpl_pack_frags(sub.data(), scale.data(), 1, op.cin, ncout, mb, ks, SLOT_SPATIAL, &fc[part], 1, part);
  f->clear();
  for (int s = 0; s < ks; ++s)
    for (int part = 0; part < PM; ++part)
      f->insert(f->end(), fc[part].begin() + (size_t)s * mb * 512, fc[part].begin() + (size_t)(s + 1) * mb * 512);
}

// K-slot -> tap of the all-LDS stem (unet_split_lds.h): groups 0 - 2: j < 6 = the (dx = 0, 1) pair of
// row 3 g + j / 2, j = 6, 7 = the dx = 2 single of rows 2 g, 2 g + 1; group 3: j = 0, 2, 4 = the dx = 2
// singles of rows 6, 7, 8.  A row r = (dz, dy) = (r / 3, r % 3); tap = 3 r + dx.  -1: zero weight.
int stem_slot_tap(int g, int j) {
  if (g < 3) return j < 6 ? (3 * g + (j >> 1)) * 3 + (j & 1) : (2 * g + (j - 6)) * 3 + 2;
  return (j < 6 && !(j & 1)) ? (6 + (j >> 1)) * 3 + 2 : -1;
}
// conv3 1 -> cout as [chunk of 16 output channels][part] fragments, plain rows
void pack_stem_u8(const float *A, const fpl_op &op, std::vector<uint16_t> *f) {
  f->assign((size_t)(op.cout / 16) * PM * 512, 0);
  for (int cc = 0; cc < op.cout / 16; ++cc)
    for (int part = 0; part < PM; ++part)
      for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 15, g = lane >> 4, co = 16 * cc + m;
        for (int j = 0; j < 8; ++j) {
          const int tap = stem_slot_tap(g, j);
          if (tap < 0) continue;
          (*f)[(((size_t)cc * PM + part) * 64 + lane) * 8 + j] =
              fpl_f32_to_h16_part(A[op.w_off + (size_t)tap * op.cout + co] * A[op.scale_off + co], part);
        }
      }
}

// which skeletons the all-LDS executor takes: every 3x3x3 second convolution (not unet_like)
bool unet_lds_ok(const UnetDesc &d) { return !d.first1 && !d.second1; }

// the planar weight streams, once per weight version (beside unet_prepare's fragments, whose
// stem, head and shift tables this path shares)
int unet_prepare_lds(fpl_ctx *ctx, fpl_program *prog, const UnetDesc &d, UnetState *st) {
  if (st->have8) return 0;
  const float *A = prog->arena_host.data();
  std::vector<uint16_t> all, f;
  const int lu1 = d.l_up1(), lu2 = d.l_up2();
  pack_stem_u8(A, prog->ops[d.conv[0]], &f);
  st->off8[0] = 0;
  all.insert(all.end(), f.begin(), f.end());
  for (int l = 1; l < d.nconv - 2; ++l) {          // (the last two = the head epilogue)
    const fpl_op &op = prog->ops[d.conv[l]];
    st->off8[l] = all.size() * sizeof(uint16_t);
    st->half8[l] = 0;
    FPL_REQUIRE(ctx, op.cin % 32 == 0 && op.cin / CHP <= u8::MAXPASS && (op.cout == 32 || op.cout % 64 == 0),
                "unet (all-LDS): conv %d -> %d", op.cin, op.cout);
    if (op.k == 3) {
      const int n_ups = l == lu1 ? 128 / CHP : l == lu2 ? 64 / CHP : 0;
      // Upsampled passes run the zy form, which exists for 32 output channels at a time (four
      // parity streams of a 64-output layer's fragments do not fit in LDS): conv3 192->64 is two
      // launches of 32 channels - a quarter fewer MFMAs than one 64-channel launch in the z form
      const int half = n_ups ? 32 : 64, mbh = op.cout == 32 ? 2 : half / 16;
      for (int h = 0; h < (op.cout > half ? op.cout / half : 1); ++h) {
        pack_u3(A, op, half * h, mbh, n_ups, u8::UM_ZY, false, &f);
        if (h == 1) st->half8[l] = all.size() * sizeof(uint16_t) - st->off8[l];
        all.insert(all.end(), f.begin(), f.end());
      }
      if (l == lu2) {
        pack_u3(A, op, 0, 2, n_ups, u8::UM_ZY, true, &f);
        st->off8t = all.size() * sizeof(uint16_t);
        all.insert(all.end(), f.begin(), f.end());
      }
    } else {
      FPL_REQUIRE(ctx, op.cin % 32 == 0 && op.cout % 64 == 0, "unet (all-LDS): conv1 %d -> %d", op.cin, op.cout);
      for (int h = 0; h < op.cout / 64; ++h) {
        pack_u1(A, op, 64 * h, 4, &f);
        if (h == 1) st->half8[l] = all.size() * sizeof(uint16_t) - st->off8[l];
        all.insert(all.end(), f.begin(), f.end());
      }
    }
  }
#ifdef FPL_F16
  for (uint16_t h : all)
    if ((h & 0x7C00u) == 0x7C00u) {
      const char *msg = "a folded weight exceeds the IEEE-half range (65504); use precision bf16, f32 or 'auto' for this network";
      return SPLIT ? fpl_fail_range(ctx, "%s", msg) : fpl_fail(ctx, "%s", msg);
    }
#endif
  if (st->frags8) FPL_HIP(ctx, hipFree(st->frags8));
  st->frags8 = nullptr;
  FPL_HIP(ctx, hipMalloc((void **)&st->frags8, all.size() * sizeof(uint16_t)));
  FPL_HIP(ctx, hipMemcpy(st->frags8, all.data(), all.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  st->have8 = true;
  return 0;
}

// a planar tensor: n tiles of d^3 voxels, C channels as C / CHP passes x 2 planes x 16 B
struct PlanarT {
  unsigned char *p = nullptr;
  int d = 0, C = 0;
  int64_t part = 0;
  const unsigned char *pass(int q, int crop = 0) const {
    return p + (int64_t)q * 2 * part + (((int64_t)crop * d + crop) * d + crop) * 16;
  }
};

template <int MB, int R, int UM, int EPI, bool STEM = false, bool TRANSPOSED = false>
int launch_u3(fpl_ctx *ctx, u8::U3Args &a, int main_w, const char *name) {
  typedef u8::Lds<MB, R, UM, STEM, STEM || EPI == u8::EPI_HEAD> L;
  typedef u8::Geo8<R, UM> GE;
  constexpr bool HAS_UPS = UM != u8::UM_NONE;
  static bool attr_set[FPL_MAX_DEVICES] = {false};
  if (!attr_set[ctx->device % FPL_MAX_DEVICES]) {
    FPL_HIP(ctx, hipFuncSetAttribute((const void *)FPLK(u8::u3conv)<MB, R, UM, EPI, STEM, TRANSPOSED>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES));
    attr_set[ctx->device % FPL_MAX_DEVICES] = true;
  }
  FPL_REQUIRE(ctx, a.npass % 2 == 0 && a.nups % 2 == 0 && a.npass <= u8::MAXPASS && (HAS_UPS || a.nups == 0),
              "u3conv: %d passes, %d upsampled", a.npass, a.nups);
  FPL_REQUIRE(ctx, (int64_t)a.Ppart + ((int64_t)6 * a.PH * a.PW + 64) * 16 < ((int64_t)1 << 32) &&
                       (int64_t)a.Upart + ((int64_t)4 * a.UH * a.UW + 64) * 16 < ((int64_t)1 << 32),
              "u3conv: a pass plane beyond the 32-bit tile offsets (fewer tiles per batch)");
  a.nbz = (int)ceil_div64(a.OD, u8::WZ);
  if (TRANSPOSED) {            // lanes walk y, sub-steps walk x in [xorg, OW)
    a.nbx = (int)ceil_div64(a.OH, 16); a.nby = (int)ceil_div64(a.OW - a.xorg, GE::BY);
  } else {
    a.nbx = (int)ceil_div64(main_w ? main_w : a.OW, 16); a.nby = (int)ceil_div64(a.OH, GE::BY);
  }
  const int64_t total = (int64_t)a.nbx * a.nby * a.nbz * a.n_tiles;
  FPL_REQUIRE(ctx, total < ((int64_t)1 << 31), "u3conv: too many blocks");
  a.dbg = getenv("FPL_U3_DBG") ? atoi(getenv("FPL_U3_DBG")) : 0;     // timing experiments (results are garbage)
  // one persistent workgroup per CU, a multiple of the 8 XCDs
  int64_t grid = std::min<int64_t>((int64_t)ctx->n_cu / 8 * 8, (total + 7) / 8 * 8);
  grid = std::max<int64_t>(8, grid);
  if (a.dbg & 32) {              // diagnostic run: per-workgroup cycle sums of wave 0 (unet_split_lds.h::stamp)
    unsigned long long *dev = nullptr;
    std::vector<unsigned long long> h((size_t)grid * 8, 0);
    FPL_HIP(ctx, hipMalloc((void **)&dev, h.size() * 8));
    FPL_HIP(ctx, hipMemset(dev, 0, h.size() * 8));
    a.dbgbuf = dev;
    FPLK(u8::u3conv)<MB, R, UM, EPI, STEM, TRANSPOSED><<<(unsigned)grid, 64 * u8::WAVES, L::BYTES, ctx->stream>>>(a);
    FPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FPL_HIP(ctx, hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
    hipFree(dev);
    double t[5] = {0, 0, 0, 0, 0};
    for (int64_t w = 0; w < grid; ++w)
      for (int k = 0; k < 5; ++k) t[k] += (double)h[w * 8 + k];
    const double nb = t[4] > 0 ? t[4] : 1;
    fprintf(stderr, "[FPL_U3_DBG] %s: %.0f blocks; cycles per block (s_memtime): start %.0f  fill %.0f  kloop %.0f  epilogue %.0f\n",
            name, t[4], t[0] / nb, t[1] / nb, t[2] / nb, t[3] / nb);
    return 0;
  }
  TimedLaunch tl(ctx, name);
  FPLK(u8::u3conv)<MB, R, UM, EPI, STEM, TRANSPOSED><<<(unsigned)grid, 64 * u8::WAVES, L::BYTES, ctx->stream>>>(a);
  return 0;
}

int unet_forward_lds(fpl_ctx *ctx, fpl_program *prog, const UnetDesc &D, UnetState *st, const float *in, int n, int T,
                     const FplTileIO *io) {
  FPL_TRY(unet_prepare_lds(ctx, prog, D, st));
  DevTemp tmp(ctx);
  unsigned *flag = nullptr;
  if (SPLIT) FPL_TRY(fpl_range_flag(ctx, &flag));
  const unsigned char *F = st->frags, *F8 = st->frags8;
  const float *S = st->shifts;
  const bool b3[2] = {prog->ops[D.conv[4]].k == 3, D.nbottom == 2 && prog->ops[D.conv[5]].k == 3};
  const int d1a = T - 2, d1 = T - 4, dp1 = d1 / 2, d2a = dp1 - 2, d2 = dp1 - 4, dp2 = d2 / 2;
  const int db0 = dp2 - (b3[0] ? 2 : 0), db = db0 - (b3[1] ? 2 : 0);
  const int d4a = 2 * db - 2, d5a = 2 * d4a - 2;
  (void)d1a;
  FPL_REQUIRE(ctx, d1 > 0 && d1 % 2 == 0 && d2 > 0 && d2 % 2 == 0 && db > 0 && d2 - 2 * D.crop2 == 2 * db &&
                       d1 - 2 * D.crop1 == 2 * d4a && d5a > 0,
              "U-Net tile edge %d does not fit this architecture (pools need even sizes, skips must meet)", T);
  auto cube = [](int d) { return (int64_t)d * d * d; };
  auto talloc = [&](int d, int C, PlanarT *t) -> int {
    t->d = d; t->C = C; t->part = (int64_t)n * cube(d) * 16;
    // tiles of edge blocks read past the last plane: up to 5 planes + 18 rows + 18 voxels
    const size_t slack = ((size_t)6 * d * d + 40 * d + 64) * 16;
    void *q;
    FPL_TRY(tmp.alloc((size_t)(C / CHP) * 2 * t->part + slack, &q));
    t->p = (unsigned char *)q;
    return 0;
  };
  PlanarT c1, p1, c2a, c2, p2, c3a, c3, c4a, c4;
  FPL_TRY(talloc(d1, 32, &c1));
  FPL_TRY(talloc(dp1, 32, &p1));
  FPL_TRY(talloc(d2a, 64, &c2a));
  FPL_TRY(talloc(d2, 64, &c2));
  FPL_TRY(talloc(dp2, 64, &p2));
  if (D.nbottom == 2) FPL_TRY(talloc(db0, 128, &c3a));
  FPL_TRY(talloc(db, 128, &c3));
  FPL_TRY(talloc(d4a, 64, &c4a));
  FPL_TRY(talloc(d4a, 64, &c4));
  auto args = [&](int l, int co0, const PlanarT *outp, int od) {
    u8::U3Args a;
    memset(&a, 0, sizeof(a));
    a.w = F8 + st->off8[l] + (co0 ? st->half8[l] : 0);
    a.shift = S + st->off_s[l] + co0;
    a.relu = 1;
    if (outp) { a.out = outp->p + (int64_t)(co0 / CHP) * 2 * outp->part; a.out_part = outp->part; }
    a.OD = a.OH = a.OW = od;
    a.keep_lo = 0; a.keep_hi = od;
    a.n_tiles = n;
    a.flag = flag; a.xlim = st->xlim;
    a.PD = a.PH = a.PW = a.UD = a.UH = a.UW = 1;
    return a;
  };
  auto plain_src = [&](u8::U3Args &a, const PlanarT &t, int crop) {
    for (int q = 0; q < t.C / CHP; ++q) a.src[a.npass++] = t.pass(q, crop);
    a.PD = a.PH = a.PW = t.d; a.Ppart = (unsigned)t.part;
  };
  auto ups_src = [&](u8::U3Args &a, const PlanarT &t) {
    for (int q = 0; q < t.C / CHP; ++q) a.src[a.npass++] = t.pass(q, 0);
    a.nups = a.npass;
    a.UD = a.UH = a.UW = t.d; a.Upart = (unsigned)t.part;
  };
  if (!SPLIT) {
    // bf16 / f16 builds: the stem pair on the round-4 kernel (two independent 4-wave workgroups per
    // CU: one computes its 32-channel tile while the other multiplies - with a K loop a third as long
    // as the split build's that overlap is worth more than the all-LDS loop: 10.1 against 12.6 ms),
    // writing c1 and p1 as planar tensors
    Conv3Args a;
    memset(&a, 0, sizeof(a));
    a.w = F + st->off_w[1]; a.shift = S + st->off_s[1]; a.relu = 1;
    a.out = (h16_t *)c1.p; a.OD = a.OH = a.OW = d1;
    a.ncc = 32 / RCH;
    for (int cc = 0; cc < a.ncc; ++cc) a.src[cc] = make_src(nullptr, d1a, CC, 0, 1, 0);
    a.raw = in; a.T = T;
    a.wstem = (const h16x8 *)(F + st->off_w[0]); a.shstem = S + st->off_s[0];
    a.pool_out = (h16_t *)p1.p;
    a.planar = 1;
    FPL_TRY((launch_conv3<2, true, true, false, 8>(ctx, a, n, "unet_stem_conv3_32_32_pool")));
  } else {  // conv3 1->32 computed into the tile of conv3 32->32, MaxPooling3D(2) in the epilogue
    u8::U3Args a = args(1, 0, &c1, d1);
    a.npass = 32 / CHP;
    a.raw = in; a.T = T;
    a.wstem = (const h16x8 *)(F8 + st->off8[0]); a.shstem = S + st->off_s[0];
    a.pool = p1.p; a.pool_part = p1.part;
    // c1's only reader (the head's conv3) sees it through Cropping3D(crop1): the shell outside is
    // pooled into p1 but never stored (a third of this kernel's 128 B-per-voxel stores at crop 6 of 96)
    a.keep_lo = D.crop1; a.keep_hi = d1 - D.crop1;
    FPL_TRY((launch_u3<2, 6, u8::UM_NONE, u8::EPI_POOL, true>(ctx, a, 0, "unet_stem_conv3_32_32_pool")));
  }
  {  // conv3 32->64
    u8::U3Args a = args(2, 0, &c2a, d2a);
    plain_src(a, p1, 0);
    FPL_TRY((launch_u3<4, 4, u8::UM_NONE, u8::EPI_STORE>(ctx, a, 0, "unet_conv3_32_64")));
  }
  {  // conv3 64->64, MaxPooling3D(2)
    u8::U3Args a = args(3, 0, &c2, d2);
    plain_src(a, c2a, 0);
    a.pool = p2.p; a.pool_part = p2.part;
    FPL_TRY((launch_u3<4, 4, u8::UM_NONE, u8::EPI_POOL>(ctx, a, 0, "unet_conv3_64_64_pool")));
  }
  auto conv1 = [&](auto kern, int nfrag, const PlanarT &x, int l, int co0, const PlanarT &y, const char *name) {
    u8::U1Args a;
    a.in = x.p; a.in_part = x.part; a.M = (int64_t)n * cube(x.d);
    a.w = F8 + st->off8[l] + (co0 ? st->half8[l] : 0);
    a.shift = S + st->off_s[l] + co0;
    a.out = y.p + (int64_t)(co0 / CHP) * 2 * y.part; a.out_part = y.part;
    a.flag = flag;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(a.M, 64), (int64_t)ctx->n_cu * 8);
    TimedLaunch tl(ctx, name);
    kern<<<grid, 256, nfrag * PM * 1024 / 2, ctx->stream>>>(a);
  };
  auto conv3_to128 = [&](int l, const PlanarT &x, const PlanarT &y, const char *name) -> int {
    for (int h = 0; h < 2; ++h) {
      u8::U3Args a = args(l, 64 * h, &y, y.d);
      plain_src(a, x, 0);
      FPL_TRY((launch_u3<4, 4, u8::UM_NONE, u8::EPI_STORE>(ctx, a, 0, name)));
    }
    return 0;
  };
  // ---- bottom
  if (!b3[0]) {
    for (int h = 0; h < 2; ++h) conv1(FPLK(u8::u1conv)<2, 4>, 16, p2, 4, 64 * h, c3, "unet_conv1_64_128");
  } else {
    const PlanarT &y0 = D.nbottom == 2 ? c3a : c3;
    FPL_TRY(conv3_to128(4, p2, y0, "unet_conv3_64_128"));
    if (b3[1]) FPL_TRY(conv3_to128(5, c3a, c3, "unet_conv3_128_128"));
    else
      for (int h = 0; h < 2; ++h) conv1(FPLK(u8::u1conv)<4, 4>, 32, c3a, 5, 64 * h, c3, "unet_conv1_128_128");
  }
  const int lu1 = D.l_up1(), lu2 = D.l_up2();
  // conv3 (up2(c3) 128 | crop(c2) 64) -> 64, as two launches of 32 output channels in the zy form;
  // blocks of 4 x 14 x 16 (unet_like2's 42 rows are three of them: 4 x 8 x 16 computed 48)
  for (int h = 0; h < 2; ++h) {
    u8::U3Args a = args(lu1, 32 * h, &c4a, d4a);
    ups_src(a, c3);
    plain_src(a, c2, D.crop2);
    FPL_TRY((launch_u3<2, 7, u8::UM_ZY, u8::EPI_STORE>(ctx, a, 0, "unet_conv3_192_64")));
  }
  conv1(FPLK(u8::u1conv)<2, 4>, 16, c4a, lu1 + 1, 0, c4, "unet_conv1_64_64");
  {  // conv3 (up2(c4) 64 | crop(c1) 32) -> 32, then the head in the epilogue.  An output width that
     // is not a multiple of 16 (unet_like2: 82 = 5 x 16 + 2) sends its last columns through the
     // transposed strip (lanes along y, two columns per block) instead of a block column that
     // would use 2 of its 16 lanes
    u8::U3Args a = args(lu2, 0, nullptr, d5a);
    ups_src(a, c4);
    plain_src(a, c1, D.crop1);
    a.io = *io;
    a.w8 = (const h16x8 *)(F + st->off_w[lu2 + 1]); a.sh8 = S + st->off_s[lu2 + 1];
    a.w9 = (const h16x8 *)(F + st->off_w[lu2 + 2]); a.bias9 = st->bias_tail;
    const int rem = d5a % 16;
    const bool strip = rem > 0 && rem <= 4 && d5a > 16;
    FPL_TRY((launch_u3<2, 6, u8::UM_ZY, u8::EPI_HEAD>(ctx, a, strip ? d5a - rem : 0, "unet_conv3_96_32_head")));
    if (strip) {
      u8::U3Args e = a;
      e.w = F8 + st->off8t;
      e.xorg = d5a - rem;
      FPL_TRY((launch_u3<2, 1, u8::UM_ZY, u8::EPI_HEAD, false, true>(ctx, e, 0, "unet_conv3_96_32_head_edge")));
    }
  }
  FPL_HIP(ctx, hipGetLastError());
  return 0;
}

#include "gx_exec.h"         // the graph executor for the programs match_unet does not know

}  // namespace

#ifdef FPL_SPLIT
// =====================================================================================
// Training (round 5): the 3x3x3 48 -> 48 convolutions of a training step - forward and input
// gradient, 2.6 of the step's 9.5 ms on v_mfma_f32_16x16x4_f32 - on split halves through the
// all-LDS kernel above: fp32-grade products at five times the fp32 matrix rate.
//   x (fp32, channels-last) --maxabs--> s = 2^e with max|x| s in [2^10, 2^11) --> planar hi / lo tensor
//   of x s (zero shell of k - 1 voxels for the input gradient) --> u3conv<3, R, EPI_F32> with the
//   weights (times their own power of two, flipped and transposed for the input gradient) split on
//   the device --> y = acc / (s_x s_w) + bias, fp32 channels-last.
// The scales are exact powers of two, so the only roundings are the two 11-bit halves per operand
// (~22 bits; fp32 accumulation): gradients of 1e-7 are as well resolved as activations of 10.
// =====================================================================================
namespace {

__global__ void ts_maxabs(const float *__restrict__ x, int64_t n, unsigned *out) {
  unsigned m = 0u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = max(m, __builtin_bit_cast(unsigned, x[i]) & 0x7FFFFFFFu);
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
// s = the power of two with max s in [2^(t-1), 2^t).  Every thread of the consumer kernels derives it from the
// maximum's bits itself (a one-thread kernel in between is a launch per tensor and per weight set, and the U-Net's
// small layers are nothing but launches); thread 0 of block 0 publishes sc[0] = s, sc[1] = 1 / s, sc[2] (if
// `other`) = 1 / (s other_s) for the kernels behind them
__device__ __forceinline__ float ts_scale_of(const unsigned *maxbits, int t) {
  const float m = __builtin_bit_cast(float, *maxbits);
  int e = 0;
  float s = 1.f;
  if (m > 0.f && m < 3.0e38f) {
    frexpf(m, &e);                       // m = f 2^e, f in [0.5, 1)
    e = t - e;
    e = e < -100 ? -100 : e > 100 ? 100 : e;
    s = ldexpf(1.f, e);
  }
  return s;
}
__device__ __forceinline__ void ts_publish(float s, float *sc, const float *other) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sc[0] = s;
    sc[1] = 1.f / s;
    if (other) sc[2] = (1.f / s) * other[1];
  }
}
// x (n, D, H, W, C) fp32, C a multiple of 16 -> planar split tensor of n tiles (D + 2 pad)^3, C / 8 passes, values
// x s.  One thread per voxel walks its channels 16 at a time (64 B reads: the voxel's lines stay in L1 across
// the steps; a thread per voxel AND pass fetched every line C / 8 times), four 16-B stores per step, each a
// coalesced stream over the wave's consecutive voxels.
__global__ void ts_to_planar(const float *__restrict__ x, int n, int D, int H, int W, int C, int pad,
                             const unsigned *maxbits, float *sc, unsigned char *out, int64_t part, int64_t slack16) {
  const int Dp = D + 2 * pad, Hp = H + 2 * pad, Wp = W + 2 * pad;
  const int64_t nv = (int64_t)n * Dp * Hp * Wp;
  const float s = ts_scale_of(maxbits, 11);
  ts_publish(s, sc, nullptr);
  // the read slack behind the last plane is ZERO (16-B pieces [0, slack16) behind the C / 8 x 2 planes)
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < slack16; v += (int64_t)gridDim.x * blockDim.x)
    *reinterpret_cast<u32x4 *>(out + (int64_t)(C / 8) * 2 * part + v * 16) = u32x4{0u, 0u, 0u, 0u};
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(v % Wp) - pad;
    int64_t t = v / Wp;
    const int yy = (int)(t % Hp) - pad;
    t /= Hp;
    const int zz = (int)(t % Dp) - pad, b = (int)(t / Dp);
    const bool in = zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W;
    const float *q = x + ((((int64_t)b * D + zz) * H + yy) * W + xx) * C;
    for (int c0 = 0; c0 < C; c0 += 16) {
      f32x4 a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = in ? *reinterpret_cast<const f32x4 *>(q + c0 + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        unsigned dummy = 0u;
        const f32x4 a0 = a[2 * pp], a1 = a[2 * pp + 1];
        const Pair2 p0 = split_pk_signed(a0[0] * s, a0[1] * s, dummy), p1 = split_pk_signed(a0[2] * s, a0[3] * s, dummy);
        const Pair2 p2 = split_pk_signed(a1[0] * s, a1[1] * s, dummy), p3 = split_pk_signed(a1[2] * s, a1[3] * s, dummy);
        unsigned char *d = out + (int64_t)(c0 / 8 + pp) * 2 * part + v * 16;
        *reinterpret_cast<u32x4 *>(d) = u32x4{p0.hi, p1.hi, p2.hi, p3.hi};
        *reinterpret_cast<u32x4 *>(d + part) = u32x4{p0.lo, p1.lo, p2.lo, p3.lo};
      }
    }
  }
}
// W fp32 [27][cin][cout] (Keras order) -> the kernel's stream [pass][K-step][hi MB | lo MB][lane][8] for the MB
// output blocks [co0, co0 + 16 MB) of THIS product (forward: outputs = cout, inputs = cin; input gradient: the
// transposed convolution, outputs = cin, inputs = cout, weights W'[tap][co][ci] = W[26 - tap][ci][co]),
// interleaved rows, values W s_w
__global__ void ts_pack_w(const float *__restrict__ Wd, int cin, int cout, int dgrad, int co0, int MB,
                          const unsigned *maxbits, float *sc, const float *xsc, unsigned short *out, int64_t total) {
  const float sw = ts_scale_of(maxbits, 6);
  ts_publish(sw, sc, xsc);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
  int64_t t = i >> 9;
  const int b = (int)(t % MB); t /= MB;
  const int set = (int)(t & 1); t >>= 1;
  const int s = (int)(t % u8::KP), p = (int)(t / u8::KP);
  const int m = lane & 15, g = lane >> 4, tap = 4 * s + g;
  const int ko = co0 + 4 * MB * (m >> 2) + 4 * b + (m & 3), ki = 8 * p + j;   // output / input channel of THIS product
  float v = 0.f;
  if (tap < 27) v = dgrad ? Wd[((size_t)(26 - tap) * cin + ko) * cout + ki] : Wd[((size_t)tap * cin + ki) * cout + ko];
  v *= sw;
  const h16_t h = (h16_t)v;
  out[i] = set ? h16_bits(v - (float)h) : h16_bits(v);
}

}  // namespace

namespace {

// ---- weight gradient of conv3 48 -> 48: dW[t][ci][co] = sum_v X[v + t][ci] dY[v][co] as MFMAs whose K
// runs over VOXELS: A = X^T (16 input channels x 32 voxels of an output row, shifted by the tap), B = dY
// (32 voxels x 16 output channels), both read out of voxel-major planar tiles in LDS by the hardware
// transpose read (ds_read_b64_tr_b16: a 16-lane group takes 4 voxels x 16 channels and every lane
// gets ONE channel's 4 voxels - 8 B of pass 2 cb and 8 B of pass 2 cb + 1 per voxel, so the planar passes of
// 8 channels serve as they are and a tap's shift is an address offset).  Three MFMAs per product
// (x_hi y_hi + x_hi y_lo + x_lo y_hi).  One persistent 8-wave workgroup per CU walks (patch, z plane, chunk of
// y rows); a step = one output row: the 81 (tap, input-channel block) pairs are dealt to the waves, each
// holding its 3 x (10 or 11) accumulator tiles for the whole kernel, added to dW by float atomics at the end.
// LDS: a ring of X rows [dz 3][y slot 4] and two dY rows, [part 2][pass 6][voxel 36] x 16 B in 7 KiB each (plane pitch
// 576 B = 64 mod 256: the two passes of a transposed read and the two lane groups of a half-wave fall on
// four different bank quarters), filled by LDS-DMA one output row ahead.
namespace wg {
constexpr int WAVES = 8, NV = 36;
constexpr int XSLOTS = 12;
// one launch: CIB blocks of 16 input channels (X rows of 2 CIB passes) against COB blocks of 16 output channels
// (dY rows of 2 COB passes); a staged row is [part 2][pass][voxel NV] x 16 B, in whole 1 KiB DMA chunks (the last
// chunk's tail lands in the slot)
template <int CIB, int COB> struct Cfg {
  static constexpr int XP = 2 * CIB, YP = 2 * COB;
  static constexpr int XCHK = (2 * XP * NV * 16 + 1023) / 1024, YCHK = (2 * YP * NV * 16 + 1023) / 1024;
  static constexpr int XROWB = XCHK * 1024, YROWB = YCHK * 1024;
  static constexpr int SMEM = XSLOTS * XROWB + 2 * YROWB;           // 48 -> 48: 14 x 7 KiB = 100 352 B
  static constexpr int NPAIR = 27 * CIB, PPW = (NPAIR + WAVES - 1) / WAVES;   // (tap, ci block) pairs per wave
  static constexpr int NCHK = 3 * XCHK + YCHK, NI = (NCHK + WAVES - 1) / WAVES;   // DMA chunks of a step, per wave
  static_assert(SMEM <= 160 * 1024, "one workgroup's rows fit the LDS");
};
}

struct WgArgs {
  const unsigned char *xp, *yp;       // planar copies: x (n, D^3; the first pass of this launch's channels), dy (n, (Dy + 4)^3, zero shell of 2)
  unsigned xpart, ypart;              // bytes of one plane
  int n, D, Dy, ychunk, nychunk;      // patches, x edge, dy edge (D - 2), output rows per block
  const float *scx, *scy;             // [s, 1 / s] of either copy
  float *dw;
  int cin, cout, ci0;                 // dw is [27][cin][cout]; this launch's input channels start at ci0
};

template <int CIB, int COB>
__global__ __launch_bounds__(64 * wg::WAVES, 2) void tm_wgrad3_split(WgArgs a) {
  using namespace wg;
  typedef Cfg<CIB, COB> CF;
  constexpr int XP = CF::XP, YP = CF::YP, PPW = CF::PPW, NPAIR = CF::NPAIR, NI = CF::NI;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int G = a.Dy + 4;                                       // dy's grid edge
  // ---- DMA decode: chunk j of a staged row of NP passes, this lane's 16 B = (part, pass, voxel)
  auto slot_off = [&](int j, unsigned part_bytes, int NP) -> unsigned {
    int s = 64 * j + lane;
    s = s < 2 * NP * NV ? s : 2 * NP * NV - 1;
    const int pl = s / NV, v = s - pl * NV;                     // LDS plane = part * NP + pass
    const int part = pl / NP, pass = pl - NP * part;
    return (unsigned)(pass * 2 + part) * part_bytes + (unsigned)v * 16u;
  };
  // the chunks of a step's fill: 3 X rows (dz = 0 .. 2) of XCHK chunks and 1 dY row of YCHK, dealt to the waves:
  // chunk id q = wave + 8 i: q < 3 XCHK: X row q / XCHK, chunk q % XCHK; else dY chunk q - 3 XCHK
  unsigned doff[NI];
  int drow[NI], dj[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int q = wave + WAVES * i;
    if (q < 3 * CF::XCHK) { drow[i] = q / CF::XCHK; dj[i] = q - CF::XCHK * drow[i]; doff[i] = slot_off(dj[i], a.xpart, XP); }
    else { drow[i] = 3; dj[i] = q - 3 * CF::XCHK; doff[i] = slot_off(dj[i] < CF::YCHK ? dj[i] : 0, a.ypart, YP); }
  }
  // per-lane offset of a transposed read inside a staged row: lane 4 q + p of a group supplies voxel
  // row q, channels 4 p .. 4 p + 3 = 8 B of pass (p >> 1) of the pair; + cb * 2 passes, + part, + voxel
  const int q4 = c >> 2, p4 = c & 3;
  const unsigned lane_off = (unsigned)(((p4 >> 1) * NV + 8 * g + q4) * 16 + 8 * (p4 & 1));
  auto tr = [&](const unsigned char *p) -> u32x2 {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                         (__attribute__((address_space(3))) s16x4 *)(p)));
  };
  // an 8-voxel operand fragment (channels of block cb, part `part`) of the row at `row` (NP passes), voxels + vofs
  auto frag = [&](const unsigned char *row, int NP, int cb, int part, int vofs) -> h16x8 {
    const unsigned char *p = row + lane_off + ((part * NP + 2 * cb) * NV + vofs) * 16;
    const u32x2 lo = tr(p), hi = tr(p + 4 * 16);
    return __builtin_bit_cast(h16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
  };
  f32x4 acc[PPW][COB];
#pragma unroll
  for (int i = 0; i < PPW; ++i)
#pragma unroll
    for (int ob = 0; ob < COB; ++ob) acc[i][ob] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 kmask;                                  // this lane's k-slots j = voxels 8 g + j: inside the row?
#pragma unroll
  for (int d = 0; d < 4; ++d)
    kmask[d] = (8 * g + 2 * d < a.Dy ? 0xFFFFu : 0u) | (8 * g + 2 * d + 1 < a.Dy ? 0xFFFF0000u : 0u);

  const int nblk = a.n * a.Dy * a.nychunk;
  const int S = (int)gridDim.x;
  unsigned char *ybuf = smem + XSLOTS * CF::XROWB;
  auto xrow = [&](int n, int z, int y) -> const unsigned char * {       // X row (z, y), voxel 0, plane 0
    return a.xp + ((((int64_t)n * a.D + z) * a.D + y) * a.D) * 16;
  };
  auto yrow = [&](int n, int z, int y) -> const unsigned char * {       // dY row (z, y) at grid (z + 2, y + 2, 2)
    return a.yp + ((((int64_t)n * G + z + 2) * G + y + 2) * G + 2) * 16;
  };
  // fill of the rows step (n, z, y) adds: X rows (z + dz, y + 2) into ring slot (dz, (y + 2) & 3), dY row y
  auto fill = [&](int n, int z, int y, bool first) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = wave + WAVES * i;
      if (q >= CF::NCHK) continue;
      if (drow[i] < 3) {
        glds16(xrow(n, z + drow[i], y + 2) + doff[i], smem + (drow[i] * 4 + ((y + 2) & 3)) * CF::XROWB + dj[i] * 1024);
        if (first) {                     // a block's first step also needs rows y and y + 1
          glds16(xrow(n, z + drow[i], y) + doff[i], smem + (drow[i] * 4 + (y & 3)) * CF::XROWB + dj[i] * 1024);
          glds16(xrow(n, z + drow[i], y + 1) + doff[i], smem + (drow[i] * 4 + ((y + 1) & 3)) * CF::XROWB + dj[i] * 1024);
        }
      } else {
        glds16(yrow(n, z, y) + doff[i], ybuf + (y & 1) * CF::YROWB + dj[i] * 1024);
      }
    }
  };
  for (int blk = blockIdx.x; blk < nblk; blk += S) {
    const int yc = blk % a.nychunk, t1 = blk / a.nychunk;
    const int z = t1 % a.Dy, n = t1 / a.Dy;
    const int y0 = yc * a.ychunk, y1 = min(a.Dy, y0 + a.ychunk);
    __syncthreads();                        // every wave has left the previous block's rows
    fill(n, z, y0, true);
    __syncthreads();
    for (int y = y0; y < y1; ++y) {
      if (y + 1 < y1) fill(n, z, y + 1, false);
      const unsigned char *yr = ybuf + (y & 1) * CF::YROWB;
      {                                            // ONE K-step of 32 voxels: the row (Dy <= 32, host)
        constexpr int x0 = 0;
        // (voxels past the row's Dy outputs - the K-step always takes 32 - are other rows' gradients: masked)
        h16x8 bh[COB], bl[COB];
#pragma unroll
        for (int ob = 0; ob < COB; ++ob) {
          bh[ob] = __builtin_bit_cast(h16x8, __builtin_bit_cast(u32x4, frag(yr, YP, ob, 0, x0)) & kmask);
          bl[ob] = __builtin_bit_cast(h16x8, __builtin_bit_cast(u32x4, frag(yr, YP, ob, 1, x0)) & kmask);
        }
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
          const int pi = wave + WAVES * i;
          if (pi < NPAIR) {
            const int tap = pi / CIB, cb = pi - CIB * tap;
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            const unsigned char *xr = smem + (dz * 4 + ((y + dy) & 3)) * CF::XROWB;
            const h16x8 ah = frag(xr, XP, cb, 0, x0 + dx), al = frag(xr, XP, cb, 1, x0 + dx);
#pragma unroll
            for (int ob = 0; ob < COB; ++ob) {
              acc[i][ob] = mfma16(al, bh[ob], acc[i][ob]);
              acc[i][ob] = mfma16(ah, bl[ob], acc[i][ob]);
              acc[i][ob] = mfma16(ah, bh[ob], acc[i][ob]);
            }
          }
        }
      }
      __syncthreads();                      // the next row's fill has landed; this row's slots are free
    }
  }
  // D[row 4 g + r = input channel][col c = output channel]
  const float us = a.scx[1] * a.scy[1];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pi = wave + WAVES * i;
    if (pi < NPAIR) {
      const int tap = pi / CIB, cb = pi - CIB * tap;
#pragma unroll
      for (int ob = 0; ob < COB; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[i][ob][r] * us;
          if (v != 0.f) atomicAdd(&a.dw[((size_t)tap * a.cin + a.ci0 + 16 * cb + 4 * g + r) * a.cout + 16 * ob + c], v);
        }
    }
  }
}

// a zeroed 64-B record of the context's pool (valid until fpl_tm_split_reset); nullptr when the pool is spent
constexpr int ZERO_RECORDS = 1024;
unsigned char *zero_record(fpl_ctx *ctx) {
  if (!ctx->zero_pool) {
    if (hipMalloc((void **)&ctx->zero_pool, ZERO_RECORDS * 64) != hipSuccess) { ctx->zero_pool = nullptr; return nullptr; }
    if (hipMemsetAsync(ctx->zero_pool, 0, ZERO_RECORDS * 64, ctx->stream) != hipSuccess) return nullptr;
    ctx->zero_next = 0;
  }
  if (ctx->zero_next >= ZERO_RECORDS) return nullptr;
  return ctx->zero_pool + 64 * (size_t)ctx->zero_next++;
}

// the planar split copy of a training tensor (x s, s a power of two from the tensor's maximum), made once per
// step and kept in the context: forward leaves x's, the weight gradient dy's (which the input gradient reuses)
int split_copy(fpl_ctx *ctx, const float *x, int n, int D, int pad, int C, FplSplitCopy *out) {
  for (const FplSplitCopy &e : ctx->split_copies)
    if (e.key == x && e.n == n && e.D == D && e.pad == pad && e.C == C) { *out = e; return 0; }
  hipStream_t st = ctx->stream;
  const int Dp = D + 2 * pad;
  FplSplitCopy e;
  e.key = x; e.n = n; e.D = D; e.pad = pad; e.C = C;
  const int64_t nv = (int64_t)n * Dp * Dp * Dp;
  e.part = nv * 16;
  void *q;
  const size_t slack = ((size_t)6 * Dp * Dp + 40 * Dp + 64) * 16;
  // the read slack behind the last plane is ZERO (ts_to_planar writes it): the weight-gradient kernel sums over
  // voxels, and a row's 32-voxel K-step runs past short rows (times masked-out gradients - but 0 x NaN is NaN).
  // The scale record - [0] s, [1] 1 / s; [4] (as unsigned) the maximum's bits - comes zeroed from the pool
  FPL_TRY(fpl_dev_alloc(ctx, (size_t)(C / 8) * 2 * e.part + slack, &q));
  e.planar = (unsigned char *)q;
  unsigned char *rec = zero_record(ctx);
  e.sc_owned = rec == nullptr;
  if (!rec) {
    FPL_TRY(fpl_dev_alloc(ctx, 64, &q));
    rec = (unsigned char *)q;
    FPL_HIP(ctx, hipMemsetAsync(rec, 0, 64, st));
  }
  e.sc = (float *)rec;
  unsigned *maxbits = (unsigned *)rec + 4;
  const int64_t nx = (int64_t)n * D * D * D * C;
  ts_maxabs<<<(unsigned)std::min<int64_t>(ceil_div64(nx, 1024), (int64_t)ctx->n_cu * 8), 256, 0, st>>>(x, nx, maxbits);
  ts_to_planar<<<(unsigned)std::min<int64_t>(ceil_div64(nv, 256), (int64_t)ctx->n_cu * 16), 256, 0, st>>>(
      x, n, D, D, D, C, pad, maxbits, e.sc, e.planar, e.part, (int64_t)(slack / 16));
  ctx->split_copies.push_back(e);
  *out = e;
  return 0;
}

}  // namespace

void fpl_tm_split_reset(fpl_ctx *ctx) {
  for (FplSplitCopy &e : ctx->split_copies) {
    fpl_dev_release(ctx, e.planar);
    if (e.sc_owned) fpl_dev_release(ctx, e.sc);
  }
  ctx->split_copies.clear();
  for (auto &w : ctx->split_wmax)
    if (!ctx->zero_pool || (unsigned char *)w.second < ctx->zero_pool || (unsigned char *)w.second >= ctx->zero_pool + ZERO_RECORDS * 64)
      fpl_dev_release(ctx, w.second);
  ctx->split_wmax.clear();
  if (ctx->zero_pool && ctx->zero_next > 0) {        // the records go back zeroed, behind the kernels that used them
    hipMemsetAsync(ctx->zero_pool, 0, (size_t)ctx->zero_next * 64, ctx->stream);
    ctx->zero_next = 0;
  }
}

// forward / input gradient on split halves: 3x3x3, both channel counts multiples of 16 from 32 up to 192 (inputs
// are passes of 8 channels, at most u8::MAXPASS; outputs go 64 / 48 / 32 to a launch)
bool fpl_tm_conv3_split_supported(int k, int cin, int cout) {
  auto ok = [](int c) { return c >= 32 && c % 16 == 0 && c <= 8 * u8::MAXPASS; };
  return k == 3 && ok(cin) && ok(cout);
}

// forward: x (n, D, H, W, cin) -> y (n, D - 2, H - 2, W - 2, cout) = conv3(x, Wd) + bias
// dgrad:   x = dy (n, D, H, W, cout) -> y = dx (n, D + 2, H + 2, W + 2, cin) (bias = zeros)
int fpl_tm_conv3_split(fpl_ctx *ctx, const float *x, int n, int D, int H, int W_, int cin, int cout, const float *Wd,
                       const float *bias, int dgrad, int relu, float *y) {
  const int Ci = dgrad ? cout : cin, Co = dgrad ? cin : cout, pad = dgrad ? 2 : 0;
  FPL_REQUIRE(ctx, D == H && H == W_, "conv3 (split training): cubic patches only");
  FPL_REQUIRE(ctx, fpl_tm_conv3_split_supported(3, cin, cout), "conv3 (split training): %d -> %d channels", cin, cout);
  const int Dp = D + 2 * pad;
  DevTemp tmp(ctx);
  hipStream_t st = ctx->stream;
  FplSplitCopy xc;
  {
    TimedLaunch tl(ctx, "train_split_prepare");
    FPL_TRY(split_copy(ctx, x, n, D, pad, Ci, &xc));
  }
  void *q;
  FPL_TRY(tmp.alloc(64, &q));
  float *scw = (float *)q;                       // [0] s_w, [1] 1 / s_w, [2] 1 / (s_w s_x): published by ts_pack_w
  // the maximum of a weight tensor: once per step (the forward's and the input gradient's sets scale alike)
  unsigned *wmax = nullptr;
  for (auto &w : ctx->split_wmax) if (w.first == Wd) wmax = w.second;
  if (!wmax) {
    wmax = (unsigned *)zero_record(ctx);
    if (!wmax) {
      FPL_TRY(fpl_dev_alloc(ctx, 64, &q));
      wmax = (unsigned *)q;
      FPL_HIP(ctx, hipMemsetAsync(wmax, 0, 4, st));
    }
    ctx->split_wmax.emplace_back((const void *)Wd, wmax);
    TimedLaunch tl(ctx, "train_split_prepare");
    ts_maxabs<<<8, 256, 0, st>>>(Wd, (int64_t)27 * cin * cout, wmax);
  }
  for (int co0 = 0; co0 < Co;) {                 // 64, 48 or 32 output channels per launch
    const int rem = Co - co0, MB = rem >= 64 ? 4 : rem / 16;
    FPL_REQUIRE(ctx, MB >= 2 && MB <= 4, "conv3 (split training): %d output channels left", rem);
    const int64_t wtotal = (int64_t)(Ci / 8) * u8::KP * 2 * MB * 512;
    FPL_TRY(tmp.alloc((size_t)wtotal * 2, &q));
    unsigned short *wstream = (unsigned short *)q;
    {
      TimedLaunch tl(ctx, "train_split_prepare");
      ts_pack_w<<<(unsigned)ceil_div64(wtotal, 256), 256, 0, st>>>(Wd, cin, cout, dgrad, co0, MB, wmax, scw, xc.sc, wstream, wtotal);
    }
    u8::U3Args a;
    memset(&a, 0, sizeof(a));
    for (int p = 0; p < Ci / 8; ++p) a.src[p] = xc.planar + (int64_t)p * 2 * xc.part;
    a.npass = Ci / 8; a.nups = 0;
    a.PD = Dp; a.PH = Dp; a.PW = Dp; a.Ppart = (unsigned)xc.part;
    a.UD = a.UH = a.UW = 1;
    a.w = (const unsigned char *)wstream;
    a.shift = bias + co0;
    a.relu = relu;
    a.OD = Dp - 2; a.OH = Dp - 2; a.OW = Dp - 2;
    a.keep_lo = 0; a.keep_hi = a.OD;
    a.n_tiles = n;
    a.out32 = y + co0; a.opitch = Co; a.unscale = scw + 2;
    // rows per block: 10 where they divide the layer better (29 -> 30, 20 -> 20), else 8
    const int w10 = (int)ceil_div64(a.OH, 10) * 10, w8 = (int)ceil_div64(a.OH, 8) * 8;
    const char *name = dgrad ? "split_conv3_dgrad" : "split_conv3_fwd";
    const bool r5 = w10 < w8;
    if (MB == 2) FPL_TRY(r5 ? (launch_u3<2, 5, u8::UM_NONE, u8::EPI_F32>(ctx, a, 0, name)) : (launch_u3<2, 4, u8::UM_NONE, u8::EPI_F32>(ctx, a, 0, name)));
    else if (MB == 3) FPL_TRY(r5 ? (launch_u3<3, 5, u8::UM_NONE, u8::EPI_F32>(ctx, a, 0, name)) : (launch_u3<3, 4, u8::UM_NONE, u8::EPI_F32>(ctx, a, 0, name)));
    else FPL_TRY((launch_u3<4, 4, u8::UM_NONE, u8::EPI_F32>(ctx, a, 0, name)));
    co0 += 16 * MB;
  }
  return 0;
}

// the weight gradient on split halves: 48 -> 48 in one launch; input channels in multiples of 32 (two blocks of
// 16 per launch) against 32 or 64 outputs
bool fpl_tm_conv3_wgrad_split_supported(int k, int cin, int cout) {
  return k == 3 && ((cin == 48 && cout == 48) || (cin % 32 == 0 && cin >= 32 && cin <= 8 * u8::MAXPASS && (cout == 32 || cout == 64)));
}

template <int CIB, int COB>
int launch_wgrad_split(fpl_ctx *ctx, WgArgs &a, int n, const char *name) {
  typedef wg::Cfg<CIB, COB> CF;
  static bool attr_set[FPL_MAX_DEVICES] = {false};
  if (!attr_set[ctx->device % FPL_MAX_DEVICES]) {
    FPL_HIP(ctx, hipFuncSetAttribute((const void *)tm_wgrad3_split<CIB, COB>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM));
    attr_set[ctx->device % FPL_MAX_DEVICES] = true;
  }
  // every workgroup ends with 27 x CIB x COB x 256 float atomics into dw: a small layer (the U-Net's 4 - 8-voxel
  // ones) runs on as few workgroups as give each at least four blocks of rows
  const int nblk = n * a.Dy * a.nychunk;
  const unsigned grid = (unsigned)std::min<int>(std::max(8, nblk / 4), ctx->n_cu);
  TimedLaunch tl(ctx, name);
  tm_wgrad3_split<CIB, COB><<<grid, 64 * wg::WAVES, CF::SMEM, ctx->stream>>>(a);
  return 0;
}

// dw [27][cin][cout] += weight gradient of the valid conv3: x (n, D^3, cin), dy (n, (D - 2)^3, cout)
int fpl_tm_conv3_wgrad_split(fpl_ctx *ctx, const float *x, int n, int D, int cin, int cout, const float *dy, float *dw) {
  FPL_REQUIRE(ctx, fpl_tm_conv3_wgrad_split_supported(3, cin, cout), "conv3 wgrad (split): %d -> %d channels", cin, cout);
  FplSplitCopy xc, yc;
  {
    TimedLaunch tl(ctx, "train_split_prepare");
    FPL_TRY(split_copy(ctx, x, n, D, 0, cin, &xc));
    FPL_TRY(split_copy(ctx, dy, n, D - 2, 2, cout, &yc));
  }
  WgArgs a;
  a.yp = yc.planar;
  FPL_REQUIRE(ctx, (cin / 4) * xc.part < ((int64_t)1 << 32) && (cout / 4) * yc.part < ((int64_t)1 << 32),
              "conv3 wgrad (split): the planar copies exceed the kernel's 32-bit offsets");
  a.xpart = (unsigned)xc.part; a.ypart = (unsigned)yc.part;
  a.n = n; a.D = D; a.Dy = D - 2;
  FPL_REQUIRE(ctx, a.Dy >= 1 && a.Dy <= 32, "conv3 wgrad (split): rows of %d outputs (one 32-voxel K-step)", a.Dy);
  a.ychunk = 10; a.nychunk = (int)ceil_div64(a.Dy, a.ychunk);
  a.scx = xc.sc; a.scy = yc.sc; a.dw = dw;
  a.cin = cin; a.cout = cout;
  if (cin == 48 && cout == 48) {
    a.xp = xc.planar; a.ci0 = 0;
    return launch_wgrad_split<3, 3>(ctx, a, n, "split_wgrad3_48to48");
  }
  char name[48];
  snprintf(name, sizeof(name), "split_wgrad3_%dto%d", cin, cout);
  for (int ci0 = 0; ci0 < cin; ci0 += 32) {        // two blocks of 16 input channels per launch
    a.xp = xc.planar + (int64_t)(ci0 / 8) * 2 * xc.part;
    a.ci0 = ci0;
    if (cout == 32) FPL_TRY((launch_wgrad_split<2, 2>(ctx, a, n, name)));
    else FPL_TRY((launch_wgrad_split<2, 4>(ctx, a, n, name)));
  }
  return 0;
}
#endif

bool FPLK(fpl_unet_fast_available)(const fpl_program *prog, int precision) {
  UnetDesc d;
  if (precision != FPL_THIS_PREC) return false;
  // every skeleton match_unet accepts, in every build; the other layer programs op by op (gx_exec.h)
  return match_unet(prog, &d) || (gx_match(prog) && !getenv("FPL_NO_GX"));
}

// in: (n, T,T,T) f32 normalised tiles on the device; out: (n, O,O,O) f32, O = T - 2 * rf_offset
// (unet_like2: T - 18; unet_like3: T - 26; unet_like4: T - 34)
int FPLK(fpl_unet_forward)(fpl_ctx *ctx, fpl_program *prog, const float *in, int n,
                          int T, float *out, const FplTileIO *io) {
  UnetDesc D;
  FPL_REQUIRE(ctx, in && (io || out), "fpl_unet_forward: no input tiles / no output");
  if (!match_unet(prog, &D)) {
    FPL_REQUIRE(ctx, gx_match(prog), "no 16-bit / split-half executor for this layer program");
    return gx_forward(ctx, prog, in, n, T, out, io);
  }
  FPL_REQUIRE(ctx, !SPLIT || io, "fpl_unet_forward: the split-half build writes into a prediction volume");
  UnetState *st;
  FPL_TRY(unet_prepare(ctx, prog, D, &st));
  // unet_like2 / 3 / 4 into a prediction volume: the all-LDS kernels (FPL_UNET_OLDSPLIT=1: the round-4
  // kernels, A/B; FPL_UNET_NOPARITY=1, the 27-tap cross-check of the parity form, exists on those only)
  if (unet_lds_ok(D) && io && !getenv("FPL_UNET_OLDSPLIT") && !getenv("FPL_UNET_NOPARITY"))
    return unet_forward_lds(ctx, prog, D, st, in, n, T, io);
  DevTemp tmp(ctx);
  unsigned *flag = nullptr;
  if (SPLIT) FPL_TRY(fpl_range_flag(ctx, &flag));
  const unsigned char *F = st->frags;
  const float *S = st->shifts;
  const bool b3[2] = {prog->ops[D.conv[4]].k == 3, D.nbottom == 2 && prog->ops[D.conv[5]].k == 3};
  const int d1a = T - 2, d1 = D.first1 ? d1a : T - 4, dp1 = d1 / 2, d2a = dp1 - 2,
            d2 = D.second1 ? d2a : dp1 - 4, dp2 = d2 / 2;
  const int db0 = dp2 - (b3[0] ? 2 : 0), db = db0 - (b3[1] ? 2 : 0);      // bottom outputs
  const int d4a = 2 * db - 2, d5a = 2 * d4a - 2;
  FPL_REQUIRE(ctx, d1 > 0 && d1 % 2 == 0 && d2 > 0 && d2 % 2 == 0 && db > 0 &&
                       d2 - 2 * D.crop2 == 2 * db && d1 - 2 * D.crop1 == 2 * d4a && d5a > 0,
              "U-Net tile edge %d does not fit this architecture (pools need even sizes, skips must meet)", T);
  auto cube = [](int d) { return (int64_t)d * d * d; };
  // conv3 tiles read up to 5 planes + 5 rows + 17 voxels past a source's last voxel
  // (C real channels = C * PM halves per voxel)
  auto balloc = [&](int64_t elems, int dim, int C, h16_t **p) -> int {
    void *q;
    const size_t slack = ((size_t)5 * dim * dim + 9 * dim + 18) * C * PM * 2;   // rows: up to R + 1 = 7 (R = 6)
    int rc = tmp.alloc((size_t)elems * PM * 2 + slack + 64, &q);
    *p = (h16_t *)q;
    return rc;
  };
  // chunk `cc` of a source: its plane of 32 physical channels
  auto src_of = [&](const h16_t *p, int dim, int C, int cc, int up, int crop) {
    return make_src(p ? p + cc * (n * cube(dim) * CC) : p, dim, CC, 0, up, crop);
  };
  h16_t *c1, *p1, *c2a, *c2, *p2, *c3a = nullptr, *c3, *c4a, *c4, *c5a;
  FPL_TRY(balloc(n * cube(d1) * 32, d1, 32, &c1));
  FPL_TRY(balloc(n * cube(dp1) * 32, dp1, 32, &p1));
  FPL_TRY(balloc(n * cube(d2a) * 64, d2a, 64, &c2a));
  FPL_TRY(balloc(n * cube(d2) * 64, d2, 64, &c2));
  FPL_TRY(balloc(n * cube(dp2) * 64, dp2, 64, &p2));
  if (D.nbottom == 2) FPL_TRY(balloc(n * cube(db0) * 128, db0, 128, &c3a));
  FPL_TRY(balloc(n * cube(db) * 128, db, 128, &c3));
  FPL_TRY(balloc(n * cube(d4a) * 64, d4a, 64, &c4a));
  FPL_TRY(balloc(n * cube(d4a) * 64, d4a, 64, &c4));
  FPL_TRY(balloc(n * cube(d5a) * 32, d5a, 32, &c5a));
  hipStream_t stm = ctx->stream;
  auto conv3_args = [&](int l, h16_t *outp, int od) {
    Conv3Args a;
    a.w = F + st->off_w[l]; a.shift = S + st->off_s[l]; a.relu = 1;
    a.out = outp; a.oplane = a.pplane = 0; a.OD = a.OH = a.OW = od; a.ncc = 0; a.zblocks = 0;
    a.raw = nullptr; a.T = 0; a.wstem = nullptr; a.shstem = nullptr; a.pool_out = nullptr;
    a.transposed = 0; a.xorg = 0; a.main_w = 0;
    memset(&a.io, 0, sizeof(a.io)); a.w8 = a.w9 = nullptr; a.sh8 = nullptr; a.bias9 = 0.f;
    a.flag = flag; a.xlim = st->xlim;
    a.parity = 0; a.total_steps = 0; a.wstream = 0; a.planar = 0;
    return a;
  };
  // the parity form for sources that are an UpSampling3D(2) (FPL_UNET_NOPARITY=1: A/B)
  const bool use_par = !getenv("FPL_UNET_NOPARITY");
  auto set_parity = [&](Conv3Args &a, int w) {
    a.w = F + st->off_wp[w]; a.parity = 1; a.wstream = (int64_t)st->wp_stream[w];
    a.total_steps = st->wp_steps[w];
  };
  if (D.first1) {  // unet_like: conv3 1->32 and conv1 32->32 chained, c1 + pooled p1
    StemC1Args a;
    a.raw = in; a.T = T;
    a.wstem = (const h16x8 *)(F + st->off_w[0]); a.shstem = S + st->off_s[0];
    a.w1 = (const h16x8 *)(F + st->off_w[1]); a.sh1 = S + st->off_s[1];
    a.c1 = c1; a.p1 = p1; a.D = d1;
    a.c1plane = (int64_t)n * cube(d1) * CC; a.p1plane = (int64_t)n * cube(dp1) * CC;
    a.flag = flag;
    a.zblocks = (int)ceil_div64(d1, 4); a.nbx = (int)ceil_div64(d1, 16); a.nby = (int)ceil_div64(d1, 4);
    TimedLaunch tl(ctx, "unet_stem_conv1_32_32_pool");
    FPLK(unet_stem_c1)<<<(unsigned)((int64_t)a.nbx * a.nby * n * a.zblocks), 256, 0, stm>>>(a);
  } else {  // conv3 1->32 computed into the tile of conv3 32->32
    Conv3Args a = conv3_args(1, c1, d1);
    a.ncc = 32 / RCH;
    for (int cc = 0; cc < a.ncc; ++cc) a.src[cc] = src_of(nullptr, d1a, 32, cc, 1, 0);
    a.raw = in; a.T = T;
    a.wstem = (const h16x8 *)(F + st->off_w[0]); a.shstem = S + st->off_s[0];
    a.pool_out = p1;                               // MaxPooling3D(2) in the epilogue
    FPL_TRY((launch_conv3<2, true, true, false, 8>(ctx, a, n, "unet_stem_conv3_32_32_pool")));
  }
  {  // conv3 32->64
    Conv3Args a = conv3_args(2, c2a, d2a);
    a.ncc = 32 / RCH;
    for (int cc = 0; cc < a.ncc; ++cc) a.src[cc] = src_of(p1, dp1, 32, cc, 1, 0);
    FPL_TRY((launch_conv3<4>(ctx, a, n, "unet_conv3_32_64")));
  }
  if (!D.second1) {  // conv3 64->64
    Conv3Args a = conv3_args(3, c2, d2);
    a.ncc = 64 / RCH;
    for (int cc = 0; cc < a.ncc; ++cc) a.src[cc] = src_of(c2a, d2a, 64, cc, 1, 0);
    a.pool_out = p2;
    FPL_TRY((launch_conv3<4, false, true>(ctx, a, n, "unet_conv3_64_64_pool")));
  }
  auto conv1 = [&](auto kern, int smem_frags, const h16_t *x, int64_t M, int l, h16_t *y,
                   const char *name) {
    Conv1Args a;
    a.in = x; a.M = M; a.plane = M * CC; a.w = F + st->off_w[l]; a.shift = S + st->off_s[l];
    a.out = y; a.w_tail = nullptr; a.bias_tail = 0.f; a.out_f32 = nullptr; a.flag = flag; a.relu = 1;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(M, 64), (int64_t)ctx->n_cu * 8);
    TimedLaunch tl(ctx, name);
    // (split: two fragment sets per K-step and twice the K-steps; up to 64 KiB of LDS)
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                        smem_frags * PM * PM * 1024);
    kern<<<grid, 256, smem_frags * PM * PM * 1024, stm>>>(a);
  };
  if (D.second1) {  // unet_like: conv1 64->64, then the pool as its own (HBM-bound) pass
    conv1(FPLK(conv1)<64, 4, 0>, 8, c2a, (int64_t)n * cube(d2a), 3, c2, "unet_conv1_64_64");
    // one launch per chunk plane (32 channels = 4 pieces of 16 B per voxel)
    const int64_t no = (int64_t)n * cube(dp2) * 4;
    TimedLaunch tl(ctx, "unet_pool_64");
    for (int ck = 0; ck < 64 * PM / CC; ++ck)
      FPLK(pool2_h16)<<<(unsigned)ceil_div64(no, 256), 256, 0, stm>>>(
          (const u32x4 *)(c2 + ck * (n * cube(d2) * CC)), (u32x4 *)(p2 + ck * (n * cube(dp2) * CC)), no, d2, 4, dp2);
  }
  // conv3 -> 128 channels: two 64-channel launches into the halves of one tensor
  auto conv3_to128 = [&](int l, const h16_t *x, int xd, int xc, h16_t *y, int yd, const char *name) -> int {
    for (int h = 0; h < 2; ++h) {
      Conv3Args a = conv3_args(l, y + (64 * PM / CC) * h * (n * cube(yd) * CC), yd);   // its chunk planes
      a.w = F + st->off_w[l] + (h ? st->half_bytes[l] : 0);
      a.shift = S + st->off_s[l] + 64 * h;
      a.ncc = xc / RCH;
      for (int cc = 0; cc < a.ncc; ++cc) a.src[cc] = src_of(x, xd, xc, cc, 1, 0);
      FPL_TRY((launch_conv3<4>(ctx, a, n, name)));
    }
    return 0;
  };
  // ---- bottom
  if (!b3[0]) {
    conv1(FPLK(conv1)<64, 8, 0>, 16, p2, (int64_t)n * cube(dp2), 4, c3, "unet_conv1_64_128");
  } else {
    h16_t *y0 = D.nbottom == 2 ? c3a : c3;
    FPL_TRY(conv3_to128(4, p2, dp2, 64, y0, db0, "unet_conv3_64_128"));
    if (b3[1]) FPL_TRY(conv3_to128(5, c3a, db0, 128, c3, db, "unet_conv3_128_128"));
    else conv1(FPLK(conv1)<128, 8, 0>, 32, c3a, (int64_t)n * cube(db0), 5, c3, "unet_conv1_128_128");
  }
  const int lu1 = D.l_up1(), lu2 = D.l_up2();
  {  // conv3 (up2(c3) 128 | crop(c2) 64) -> 64
    Conv3Args a = conv3_args(lu1, c4a, d4a);
    const int n3 = 128 / RCH, n2 = 64 / RCH;
    a.ncc = n3 + n2;
    for (int cc = 0; cc < n3; ++cc) a.src[cc] = src_of(c3, db, 128, cc, 2, 0);
    for (int cc = 0; cc < n2; ++cc) a.src[n3 + cc] = src_of(c2, d2, 64, cc, 1, D.crop2);
    if (use_par && st->wp_steps[0]) {
      set_parity(a, 0);
      FPL_TRY((launch_conv3<4, false, false, false, 4, true>(ctx, a, n, "unet_conv3_192_64")));
    } else
      FPL_TRY((launch_conv3<4>(ctx, a, n, "unet_conv3_192_64")));
  }
  conv1(FPLK(conv1)<64, 4, 0>, 8, c4a, (int64_t)n * cube(d4a), lu1 + 1, c4, "unet_conv1_64_64");
  {  // conv3 (up2(c4) 64 | crop(c1) 32) -> 32.  unet_like2's output width 82 = 5 x 16 + 2:
     // the last two columns go through the transposed edge strip instead of a sixth block
     // column that would use 2 of its 16 lanes
    Conv3Args a = conv3_args(lu2, c5a, d5a);
    const int n4 = 64 / RCH, n1 = 32 / RCH;
    a.ncc = n4 + n1;
    for (int cc = 0; cc < n4; ++cc) a.src[cc] = src_of(c4, d4a, 64, cc, 2, 0);
    for (int cc = 0; cc < n1; ++cc) a.src[n4 + cc] = src_of(c1, d1, 32, cc, 1, D.crop1);
    const int rem = d5a % 16;
    const bool strip = rem > 0 && rem <= 4 && d5a > 16 && st->off_w7t != 0;
    if (strip) a.main_w = d5a - rem;
    if (io) {                    // conv1 32->32, conv1 32->1, sigmoid and the store into
      a.io = *io;                // the prediction volume ride in the epilogue
      a.w8 = (const h16x8 *)(F + st->off_w[lu2 + 1]); a.sh8 = S + st->off_s[lu2 + 1];
      a.w9 = (const h16x8 *)(F + st->off_w[lu2 + 2]); a.bias9 = st->bias_tail;
      a.out = nullptr;
    }
    const unsigned char *w_plain = a.w;
    if (use_par && st->wp_steps[1]) {
      set_parity(a, 1);
      if (io) FPL_TRY((launch_conv3<2, false, false, true, 6, true>(ctx, a, n, "unet_conv3_96_32_head")));
      else FPL_TRY((launch_conv3<2, false, false, false, 6, true>(ctx, a, n, "unet_conv3_96_32")));
    } else if (io) {
      FPL_TRY((launch_conv3<2, false, false, true, 6>(ctx, a, n, "unet_conv3_96_32_head")));
    } else {
      FPL_TRY((launch_conv3<2, false, false, false, 6>(ctx, a, n, "unet_conv3_96_32")));
    }
    if (strip) {
      Conv3Args e = a;
      e.main_w = 0; e.transposed = 1; e.xorg = d5a - rem;
      e.parity = 0; e.wstream = 0; e.total_steps = 0; (void)w_plain;
      e.w = F + st->off_w7t;
      if (io) FPL_TRY((launch_conv3<2, false, false, true>(ctx, e, n, "unet_conv3_96_32_head_edge")));
      else FPL_TRY((launch_conv3<2>(ctx, e, n, "unet_conv3_96_32_edge")));
    }
  }
#ifndef FPL_SPLIT
  if (!io) {  // conv1 32->32 (+ReLU) chained into conv1 32->1, sigmoid
    Conv1Args a;
    a.in = c5a; a.M = (int64_t)n * cube(d5a); a.plane = a.M * CC; a.w = F + st->off_w[lu2 + 1]; a.shift = S + st->off_s[lu2 + 1];
    a.out = nullptr; a.w_tail = (const h16x8 *)(F + st->off_w[lu2 + 2]); a.bias_tail = st->bias_tail;
    a.out_f32 = out; a.flag = nullptr; a.relu = 1;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(a.M, 64), (int64_t)ctx->n_cu * 8);
    TimedLaunch tl(ctx, "unet_head_" FPL_PREC_STR);
    FPLK(conv1)<32, 2, 1><<<grid, 256, 2 * 1024, stm>>>(a);
  }
#endif
  FPL_HIP(ctx, hipGetLastError());
  return 0;
}
