// Graph executor on the chunk-plane kernels of conv_mfma.hip (included there, inside its anonymous
// namespace, once per operand build): the layer programs that are NOT one of the U-Net skeletons
// match_unet knows - baseline_model, resnet_like, unet_like4b, unet_like_vol
// (reference flypylib/fplmodels.py:73-100, 174-208, 410-467, 470-526) - run op by op on the same
// descriptor-driven kernels: conv3 (any mix of plain / upsampled / cropped 16-channel source chunks,
// 32 or 64 outputs per launch, optional fused MaxPooling3D), conv1 (voxel GEMM), pool2_h16, and three
// small kernels of this file: the stand-alone first layer (conv3 1 -> C), Add (+ReLU) of a cropped
// shortcut, and the 1x1x1 -> 1 sigmoid head, which takes the hi + lo halves back to fp32 and writes the
// prediction volume (UpSampling3D(rf_stride) on store).
//
// Tensors: planes of CC physical channels, [chunk][n][z][y][x][32] (conv_mfma.hip), channel counts
// padded to a multiple of 32 - a 16- or 48-channel layer is computed as 32 / 64 outputs whose padding
// has zero weights and zero shift (ReLU(0) = 0, hi = lo = 0), and a consumer's weight rows for the padding
// are zero too.  UpSampling3D, Cropping3D and concatenate never materialise: they are resolved into the
// source descriptors of the convolution that reads them (and the crop of a shortcut into Add's indexing).
#pragma once

struct GxView { int base, ups, crop; };   // a materialised tensor seen through at most one UP or CROP

struct GxConvW {
  size_t off_w = 0, half_bytes = 0;       // fragments; 128 outputs: bytes of the first 64-channel half
  size_t off_s = 0;                       // shifts (floats), padded
  int cin_p = 0, cout_p = 0;
  float bias = 0.f;                       // the head's
  float xlim = 0.f;                       // first layer computed in a tile loader: input limit of its half-range bound
  // parity form (conv_mfma.hip) of a 3x3x3 convolution whose LEADING source chunks are an UpSampling3D(2):
  // both weight streams, one after the other; 0 steps = none
  size_t off_wp = 0, wp_stream = 0;
  int wp_steps = 0;
  // HEAD epilogue (gx_chain_head): the chained fragments of the 32 -> 32 and the 32 -> 1 convolution
  size_t off_w8 = 0, off_w9 = 0;
  bool head_chain = false;
};

struct GxState {
  uint64_t version = ~0ull;
  unsigned char *frags = nullptr;
  float *shifts = nullptr;
  std::vector<GxConvW> conv;              // by op index
};

void gx_state_free(fpl_ctx *, void *p) {
  GxState *s = (GxState *)p;
  if (s->frags) hipFree(s->frags);
  if (s->shifts) hipFree(s->shifts);
  delete s;
}

inline int gx_pad32(int c) { return (c + 31) / 32 * 32; }

struct GxPlan {
  std::vector<int> prod;                  // producing op of every tensor (-1: the network input)
  std::vector<int> chan;                  // real channels of every tensor
  std::vector<std::vector<int>> users;    // ops reading a tensor DIRECTLY
};

void gx_plan(const fpl_program *prog, GxPlan *pl) {
  pl->prod.assign(prog->n_tensors, -1);
  pl->chan.assign(prog->n_tensors, 1);
  pl->users.assign(prog->n_tensors, {});
  for (size_t i = 0; i < prog->ops.size(); ++i) {
    const fpl_op &op = prog->ops[i];
    if (op.dst >= 0 && op.dst < prog->n_tensors) { pl->prod[op.dst] = (int)i; pl->chan[op.dst] = op.cout; }
    if (op.src0 >= 0 && op.src0 < prog->n_tensors) pl->users[op.src0].push_back((int)i);
    if (op.src1 >= 0 && op.src1 < prog->n_tensors) pl->users[op.src1].push_back((int)i);
  }
}

bool gx_symmetric(const fpl_op &op, int v0) {
  for (int q = 0; q < 6; ++q) if (op.p[q] != v0) return false;
  return true;
}

// tensor t as a list of views of materialised tensors (concatenation order)
bool gx_views(const fpl_program *prog, const GxPlan &pl, int t, std::vector<GxView> *out) {
  if (t < 0 || t >= prog->n_tensors) return false;
  const int pi = pl.prod[t];
  if (pi < 0) { out->push_back(GxView{t, 0, 0}); return t == 0; }
  const fpl_op &op = prog->ops[pi];
  switch (op.kind) {
    case FPL_OP_CONV: case FPL_OP_POOL: case FPL_OP_ADD:
      out->push_back(GxView{t, 0, 0});
      return true;
    case FPL_OP_UP: case FPL_OP_CROP: {
      std::vector<GxView> in;
      if (!gx_views(prog, pl, op.src0, &in) || in.size() != 1 || in[0].ups || in[0].crop || in[0].base == 0) return false;
      if (op.kind == FPL_OP_UP) {
        if (op.p[0] != 2 || op.p[1] != 2 || op.p[2] != 2) return false;
        out->push_back(GxView{in[0].base, 1, 0});
      } else {
        if (op.p[0] <= 0 || !gx_symmetric(op, op.p[0])) return false;
        out->push_back(GxView{in[0].base, 0, op.p[0]});
      }
      return true;
    }
    case FPL_OP_CONCAT:
      return gx_views(prog, pl, op.src0, out) && gx_views(prog, pl, op.src1, out);
    default:
      return false;
  }
}

bool gx_plain(const fpl_program *prog, const GxPlan &pl, int t, int *base) {
  std::vector<GxView> v;
  if (!gx_views(prog, pl, t, &v) || v.size() != 1 || v[0].ups || v[0].crop || v[0].base == 0) return false;
  *base = v[0].base;
  return true;
}

// the first layer followed by a 1x1x1 convolution of the same (<= 32) width and a pool - unet_like_vol's first
// stage, unet_like's - runs as ONE kernel (unet_stem_c1: the 1x1x1 chained in registers, c1 and the pooled
// tensor stored): returns the 1x1x1 convolution's op index, or -1
int gx_chain_c1(const fpl_program *prog, const GxPlan &pl, int i) {
  const fpl_op &op = prog->ops[i];
  if (op.kind != FPL_OP_CONV || op.k != 3 || op.cin != 1 || pl.users[op.dst].size() != 1) return -1;
  const int j = pl.users[op.dst][0];
  const fpl_op &c = prog->ops[j];
  if (c.kind != FPL_OP_CONV || c.k != 1 || c.cin != op.cout || c.cout > 32 || c.cout == 1 || c.act != FPL_ACT_RELU ||
      c.src0 != op.dst)
    return -1;
  for (int u : pl.users[c.dst]) if (prog->ops[u].kind == FPL_OP_POOL) return j;
  return -1;
}

// the first layer (1 -> 32) in front of a 3x3x3 convolution 32 -> 32 + ReLU + pool - unet_like4b's first stage,
// as unet_like2's - runs inside that convolution's tile loader (conv3<2, PF, STEM, POOL, ., 8>: the 32-channel
// full-resolution tensor never exists in HBM): returns the convolution's op index, or -1
int gx_chain_c3(const fpl_program *prog, const GxPlan &pl, int i) {
  const fpl_op &op = prog->ops[i];
  if (op.kind != FPL_OP_CONV || op.k != 3 || op.cin != 1 || op.cout != 32 || pl.users[op.dst].size() != 1) return -1;
  const int j = pl.users[op.dst][0];
  const fpl_op &c = prog->ops[j];
  if (c.kind != FPL_OP_CONV || c.k != 3 || c.cin != 32 || c.cout != 32 || c.act != FPL_ACT_RELU || c.src0 != op.dst) return -1;
  for (int u : pl.users[c.dst]) if (prog->ops[u].kind == FPL_OP_POOL) return j;
  return -1;
}

// conv3 -> 32 (+ReLU), conv1 32 -> 32 (+ReLU), conv1 32 -> 1 (sigmoid) at the network's end - unet_like4b's and
// unet_like_vol's, as unet_like2's - with rf_stride 1: the two 1x1x1 convolutions, the sigmoid and the store
// into the prediction volume ride in the conv3's epilogue (HEAD).  Returns the 32 -> 32 op (the 32 -> 1 is its
// only user), or -1
int gx_chain_head(const fpl_program *prog, const GxPlan &pl, int i) {
  const fpl_op &op = prog->ops[i];
  if (op.kind != FPL_OP_CONV || op.k != 3 || op.cin == 1 || op.cout != 32 || op.act != FPL_ACT_RELU ||
      pl.users[op.dst].size() != 1 || prog->stride[0] != 1)
    return -1;
  const int j = pl.users[op.dst][0];
  const fpl_op &c = prog->ops[j];
  if (c.kind != FPL_OP_CONV || c.k != 1 || c.cin != 32 || c.cout != 32 || c.act != FPL_ACT_RELU || c.src0 != op.dst ||
      pl.users[c.dst].size() != 1)
    return -1;
  const fpl_op &h = prog->ops[pl.users[c.dst][0]];
  if (h.kind != FPL_OP_CONV || h.k != 1 || h.cin != 32 || h.cout != 1 || h.act != FPL_ACT_SIGMOID || h.dst != prog->out_tensor)
    return -1;
  return j;
}

// A 1x1x1 convolution without activation whose (possibly cropped) output is read by ONE Add and nothing else,
// the Add's other operand computed earlier: the Add (and its ReLU) ride in the convolution's epilogue
// (resnet_like's two shortcuts).  Returns the Add's op index, or -1; *self = 0 / 1: which operand the conv is
int gx_fuse_add(const fpl_program *prog, const GxPlan &pl, int i, int *self) {
  const fpl_op &op = prog->ops[i];
  if (op.kind != FPL_OP_CONV || op.k != 1 || op.cout == 1 || op.act != FPL_ACT_NONE || pl.users[op.dst].size() != 1) return -1;
  int u = pl.users[op.dst][0];
  if (prog->ops[u].kind == FPL_OP_CROP) {                   // conv -> crop -> add
    if (pl.users[prog->ops[u].dst].size() != 1) return -1;
    u = pl.users[prog->ops[u].dst][0];
  }
  const fpl_op &ad = prog->ops[u];
  if (ad.kind != FPL_OP_ADD) return -1;
  std::vector<GxView> a, b;
  if (!gx_views(prog, pl, ad.src0, &a) || !gx_views(prog, pl, ad.src1, &b) || a.size() != 1 || b.size() != 1) return -1;
  const int me = a[0].base == op.dst ? 0 : b[0].base == op.dst ? 1 : -1;
  if (me < 0 || a[0].base == b[0].base) return -1;
  const GxView &other = me == 0 ? b[0] : a[0];
  if (other.ups || other.base == 0 || pl.prod[other.base] >= i) return -1;      // must exist when this conv runs
  *self = me;
  return u;
}

// the layer programs this executor takes (everything match_unet does not)
bool gx_match(const fpl_program *prog) {
  if (prog->ops.empty() || prog->n_tensors < 2) return false;
  if (prog->stride[0] != prog->stride[1] || prog->stride[1] != prog->stride[2] || prog->stride[0] < 1 ||
      prog->stride[0] > 8)
    return false;
  GxPlan pl;
  gx_plan(prog, &pl);
  bool head = false;
  for (size_t i = 0; i < prog->ops.size(); ++i) {
    const fpl_op &op = prog->ops[i];
    if (head) return false;                              // nothing follows the head
    switch (op.kind) {
      case FPL_OP_CONV: {
        if (op.k == 3 && op.cin == 1) {                  // the first layer
          if (op.src0 != 0 || op.act != FPL_ACT_RELU || op.cout > 32) return false;
        } else if (op.k == 3) {
          std::vector<GxView> v;
          if (!gx_views(prog, pl, op.src0, &v)) return false;
          int chunks = 0, cin = 0;
          std::vector<int> bases;
          for (auto &s : v) {
            if (s.base == 0 || (s.ups && s.crop) || s.crop % 2) return false;
            chunks += (pl.chan[s.base] + RCH - 1) / RCH;
            cin += pl.chan[s.base];
            if (std::find(bases.begin(), bases.end(), s.base * 2 + s.ups) == bases.end()) bases.push_back(s.base * 2 + s.ups);
          }
          if (chunks > 12 || cin != op.cin || (int)bases.size() > MAXTAB) return false;
          if (op.cout > 64 && op.cout != 128) return false;
          if (op.act != FPL_ACT_RELU && op.act != FPL_ACT_NONE) return false;
        } else if (op.k == 1 && op.cout == 1) {          // the head
          int b;
          if (op.act != FPL_ACT_SIGMOID || !gx_plain(prog, pl, op.src0, &b) || op.dst != prog->out_tensor) return false;
          head = true;
        } else if (op.k == 1) {
          int b;
          if (!gx_plain(prog, pl, op.src0, &b)) return false;
          const int ci = gx_pad32(op.cin), co = gx_pad32(op.cout);
          if ((ci != 32 && ci != 64 && ci != 128) || (co != 32 && co != 64)) return false;
          if (op.act != FPL_ACT_RELU && op.act != FPL_ACT_NONE) return false;
        } else {
          return false;
        }
        break;
      }
      case FPL_OP_POOL: {
        int b;
        if (op.p[0] != 2 || op.p[1] != 2 || op.p[2] != 2 || !gx_plain(prog, pl, op.src0, &b)) return false;
        const fpl_op &src = prog->ops[pl.prod[b]];       // pool2_h16 takes values >= 0
        if ((src.kind != FPL_OP_CONV && src.kind != FPL_OP_ADD) || src.act != FPL_ACT_RELU) return false;
        break;
      }
      case FPL_OP_ADD: {
        std::vector<GxView> a, b;
        if (!gx_views(prog, pl, op.src0, &a) || !gx_views(prog, pl, op.src1, &b)) return false;
        if (a.size() != 1 || b.size() != 1 || a[0].ups || b[0].ups || a[0].base == 0 || b[0].base == 0) return false;
        if (pl.chan[a[0].base] != pl.chan[b[0].base]) return false;
        if (op.act != FPL_ACT_RELU && op.act != FPL_ACT_NONE) return false;
        break;
      }
      case FPL_OP_UP: case FPL_OP_CROP: case FPL_OP_CONCAT: {
        std::vector<GxView> v;
        if (!gx_views(prog, pl, op.dst, &v)) return false;
        for (int u : pl.users[op.dst]) {                 // views are read by convolutions (a crop also by Add)
          const fpl_op &c = prog->ops[u];
          const bool ok = (c.kind == FPL_OP_CONV && c.k == 3 && c.cin > 1) || c.kind == FPL_OP_CONCAT ||
                          (op.kind == FPL_OP_CROP && c.kind == FPL_OP_ADD);
          if (!ok) return false;
        }
        break;
      }
      default:
        return false;
    }
  }
  return head;
}

// ---- kernels --------------------------------------------------------------------------------

// The first layer on its own: conv3 1 -> 32 (padded) + shift + ReLU of the raw fp32 tiles, stored as c1
// and / or max-pooled into p1.  Block 4 x 4 x 16 voxels, wave = z, sub-steps = y, lanes = x; 27 taps in
// one K-step (SLOT_STEM, interleaved rows), as unet_stem_c1 without its chained 1x1x1 convolution.
struct GxStemArgs {
  const float *raw; int T;
  const h16x8 *wstem;            // [part][b] fragments
  const float *shstem;
  h16_t *c1, *p1;                // (n, D, D, D, 32), D = T - 2; (n, D/2, D/2, D/2, 32); either may be null
  int64_t c1plane, p1plane;
  int D, zblocks, nbx, nby;
  unsigned *flag;
};

__global__ __launch_bounds__(256) void FPLK(gx_stem)(GxStemArgs a) {
  constexpr int RZ = 6, RY = 6, RX = 18;
  __shared__ unsigned short rawt[PM * RZ * RY * RX];        // split: hi, then lo
  __shared__ f32x4 xch[2 * 2 * 2 * 64];                     // [wave pair][y half][b][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int bx = blockIdx.x % a.nbx, by = (blockIdx.x / a.nbx) % a.nby, bz = blockIdx.x / (a.nbx * a.nby);
  const int n = bz / a.zblocks, z0 = (bz % a.zblocks) * 4, y0 = by * 4, x0 = bx * 16;
  const float *base = a.raw + (int64_t)n * a.T * a.T * a.T;
  unsigned ovf = 0u;
  for (int p = tid; p < RZ * RY * RX; p += 256) {
    int z = z0 + p / (RY * RX), y = y0 + (p / RX) % RY, x = x0 + p % RX;
    z = z < a.T ? z : a.T - 1;                             // clamped reads only feed masked outputs
    y = y < a.T ? y : a.T - 1;
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
  h16x8 ws[2], wsl[2];
  f32x4 shs[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    ws[b] = a.wstem[b * 64 + lane];
    wsl[b] = a.wstem[((SPLIT ? 2 : 0) + b) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) shs[b][r] = a.shstem[8 * g + 4 * b + r];
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
      acc[sub][0] = mfma3(ws[0], wsl[0], b2, shs[0]);
      acc[sub][1] = mfma3(ws[1], wsl[1], b2, shs[1]);
    } else {
      acc[sub][0] = mfma16(ws[0], bf, shs[0]);
      acc[sub][1] = mfma16(ws[1], bf, shs[1]);
    }
    const int oz = z0 + wave, oy = y0 + sub, ox = x0 + c;
    if (a.c1 && oz < a.D && oy < a.D && ox < a.D)
      store_il<2, true>(a.c1 + ((((int64_t)n * a.D + oz) * a.D + oy) * a.D + ox) * CC, a.c1plane, g, acc[sub], 1, ovf);
  }
  if (a.p1) {
    // the pool in fp32 (monotonic in either representation), ReLU and the store at the end
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
    if (wave & 1) {
#pragma unroll
      for (int yh = 0; yh < 2; ++yh)
#pragma unroll
        for (int b = 0; b < 2; ++b) xch[(((wave >> 1) * 2 + yh) * 2 + b) * 64 + lane] = pmf[yh][b];
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
            const f32x4 o = xch[(((wave >> 1) * 2 + yh) * 2 + b) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) m[b][r] = __builtin_fmaxf(pmf[yh][b][r], o[r]);
          }
          store_il<2, true>(a.p1 + ((((int64_t)n * PD + pz) * PD + py) * PD + px) * CC, a.p1plane, g, m, 1, ovf);
        }
      }
    }
  }
  if (SPLIT) ovf_commit(ovf, a.flag, FPL_RANGE_UNET);
}

// out = act(a[crop ..] + b): one thread per voxel and 16-B piece group of a chunk plane.  The sum is
// formed in fp32 from hi + lo (exact) - the reference's own addition - and split again.
struct GxAddArgs {
  const h16_t *a, *b;
  h16_t *out;
  int n, da, db, d, crop_a, crop_b, nplanes, relu;
  int64_t aplane, bplane, oplane;      // elements of one chunk plane
  unsigned *flag;
};

__global__ void FPLK(gx_add)(GxAddArgs p) {
  constexpr int NP = SPLIT ? 2 : 4;    // pieces per thread group: split = the two 8-channel halves of 16 real channels
  const int64_t nvox = (int64_t)p.n * p.d * p.d * p.d;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvox * NP * p.nplanes) return;
  const int piece = (int)(i % NP);
  const int64_t v = (i / NP) % nvox;
  const int pl = (int)(i / (NP * nvox));
  int64_t t = v;
  const int x = (int)(t % p.d); t /= p.d;
  const int y = (int)(t % p.d); t /= p.d;
  const int z = (int)(t % p.d); t /= p.d;
  const int64_t va = ((t * p.da + z + p.crop_a) * p.da + y + p.crop_a) * (int64_t)p.da + x + p.crop_a;
  const int64_t vb = ((t * p.db + z + p.crop_b) * p.db + y + p.crop_b) * (int64_t)p.db + x + p.crop_b;
  const h16_t *pa = p.a + pl * p.aplane + va * CC + 8 * piece;
  const h16_t *pb = p.b + pl * p.bplane + vb * CC + 8 * piece;
  h16_t *po = p.out + pl * p.oplane + v * CC + 8 * piece;
  const h16x8 ah = *reinterpret_cast<const h16x8 *>(pa), bh = *reinterpret_cast<const h16x8 *>(pb);
  float s[8];
  if (SPLIT) {
    const h16x8 al = *reinterpret_cast<const h16x8 *>(pa + 16), bl = *reinterpret_cast<const h16x8 *>(pb + 16);
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] = ((float)ah[q] + (float)al[q]) + ((float)bh[q] + (float)bl[q]);
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] = (float)ah[q] + (float)bh[q];
  }
  unsigned ovf = 0u;
  u32x4 oh, ol;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const Pair2 pr = p.relu ? split_pk_relu(s[2 * q], s[2 * q + 1], ovf) : split_pk_signed(s[2 * q], s[2 * q + 1], ovf);
    oh[q] = pr.hi; ol[q] = pr.lo;
  }
  *reinterpret_cast<u32x4 *>(po) = oh;
  if (SPLIT) {
    *reinterpret_cast<u32x4 *>(po + 16) = ol;
    ovf_commit(ovf, p.flag, FPL_RANGE_UNET);
  }
}

// The head: 1x1x1 convolution to ONE channel + bias + sigmoid, in fp32 on hi + lo (no MFMA: 2 C flops per
// voxel against 4 C bytes read).  A thread per network output voxel writes its rf_stride^3 voxels of the
// prediction volume (or the per-tile output tensor when there is no volume).
struct GxHeadArgs {
  const h16_t *x;
  int64_t plane;                 // elements of one chunk plane
  int n, d, C, s;
  const float *w;                // [C], scale folded in
  float bias;
  FplTileIO io;
  int use_io;
  float *out;                    // (n, d, d, d)
};

__global__ void FPLK(gx_head)(GxHeadArgs a) {
  const int64_t nvox = (int64_t)a.n * a.d * a.d * a.d;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvox) return;
  float acc = 0.f;
  for (int c0 = 0; c0 < a.C; c0 += RCH) {
    const h16_t *px = a.x + (c0 / RCH) * a.plane + i * CC;
#pragma unroll
    for (int h = 0; h < RCH / 8; ++h) {
      if (c0 + 8 * h >= a.C) break;
      const h16x8 hi = *reinterpret_cast<const h16x8 *>(px + 8 * h);
      h16x8 lo = hi;
      if (SPLIT) lo = *reinterpret_cast<const h16x8 *>(px + RCH + 8 * h);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int ch = c0 + 8 * h + q;
        if (ch < a.C) acc = __builtin_fmaf(SPLIT ? (float)hi[q] + (float)lo[q] : (float)hi[q], a.w[ch], acc);
      }
    }
  }
  const float prob = 1.f / (1.f + __expf(-(acc + a.bias)));
  if (!a.use_io) { a.out[i] = prob; return; }
  int64_t t = i;
  const int x = (int)(t % a.d); t /= a.d;
  const int y = (int)(t % a.d); t /= a.d;
  const int z = (int)(t % a.d); t /= a.d;
  const FplTileDesc td = a.io.tiles[t];
  const int off = a.io.off;
  for (int dz = 0; dz < a.s; ++dz)
    for (int dy = 0; dy < a.s; ++dy)
      for (int dx = 0; dx < a.s; ++dx) {
        const int fz = z * a.s + dz, fy = y * a.s + dy, fx = x * a.s + dx;
        if (fz < td.ext[0] - 2 * off && fy < td.ext[1] - 2 * off && fx < td.ext[2] - 2 * off)
          a.io.dst[((int64_t)(td.start[0] + off + fz - a.io.dst_z_base) * a.io.Y + td.start[1] + off + fy) * a.io.X +
                   td.start[2] + off + fx] = prob;
      }
}

// ---- host: weights ---------------------------------------------------------------------------

// a convolution with padded channels as a stand-alone arena + op the packers of conv_mfma.hip take:
// rows[r] = the real input channel of padded input channel r (or -1), cout_p >= op.cout
void gx_padded_op(const float *A, const fpl_op &op, const std::vector<int> &rows, int cout_p,
                  std::vector<float> *Ap, fpl_op *opp) {
  const int k3 = op.k * op.k * op.k, cin_p = (int)rows.size();
  Ap->assign((size_t)k3 * cin_p * cout_p + 2 * (size_t)cout_p, 0.f);
  for (int t = 0; t < k3; ++t)
    for (int r = 0; r < cin_p; ++r) {
      if (rows[r] < 0) continue;
      memcpy(&(*Ap)[((size_t)t * cin_p + r) * cout_p], A + op.w_off + ((size_t)t * op.cin + rows[r]) * op.cout,
             op.cout * sizeof(float));
    }
  const size_t so = (size_t)k3 * cin_p * cout_p;
  for (int co = 0; co < cout_p; ++co) {
    (*Ap)[so + co] = co < op.cout ? A[op.scale_off + co] : 1.f;
    (*Ap)[so + cout_p + co] = co < op.cout ? A[op.shift_off + co] : 0.f;
  }
  *opp = op;
  opp->cin = cin_p; opp->cout = cout_p; opp->w_off = 0; opp->scale_off = (int64_t)so; opp->shift_off = (int64_t)(so + cout_p);
}

int gx_prepare(fpl_ctx *ctx, fpl_program *prog, const GxPlan &pl, GxState **out) {
  GxState *st = (GxState *)prog->fast_state_h16[FPL_H16_SLOT];
  if (!st) {
    st = new GxState();
    prog->fast_state_h16[FPL_H16_SLOT] = st;
    prog->fast_state_h16_free[FPL_H16_SLOT] = gx_state_free;
  }
  *out = st;
  if (st->version == prog->arena_version) return 0;
  const float *A = prog->arena_host.data();
  std::vector<uint16_t> all;
  std::vector<float> shifts;
  st->conv.assign(prog->ops.size(), GxConvW());
  for (size_t i = 0; i < prog->ops.size(); ++i) {
    const fpl_op &op = prog->ops[i];
    if (op.kind != FPL_OP_CONV) continue;
    GxConvW &cw = st->conv[i];
    std::vector<uint16_t> f;
    cw.off_s = shifts.size();
    if (op.k == 1 && op.cout == 1) {                       // head: fp32 weights [C] (scale folded in), then the bias
      for (int ci = 0; ci < op.cin; ++ci) shifts.push_back(A[op.w_off + ci] * A[op.scale_off]);
      shifts.push_back(A[op.shift_off]);
      while (shifts.size() % 4) shifts.push_back(0.f);
      cw.cin_p = op.cin; cw.cout_p = 1; cw.bias = A[op.shift_off];
      continue;
    }
    if (op.k == 3 && op.cin == 1 && gx_chain_c3(prog, pl, (int)i) >= 0) {
      // first layer computed in the tile loader of the next convolution (conv3's STEM variant): split build
      // per chunk of 16 output channels, plain rows, [chunk][part]; 16-bit builds two interleaved fragments
      std::vector<float> scale(A + op.scale_off, A + op.scale_off + op.cout);
      if (SPLIT) {
        for (int cc = 0; cc < 2; ++cc) {
          std::vector<float> wc((size_t)27 * 16);
          for (int t = 0; t < 27; ++t) memcpy(&wc[(size_t)t * 16], A + op.w_off + (size_t)t * op.cout + 16 * cc, 16 * sizeof(float));
          for (int part = 0; part < 2; ++part) {
            std::vector<uint16_t> fp;
            fpl_pack_frags(wc.data(), scale.data() + 16 * cc, 27, 1, 16, 1, 1, SLOT_STEM, &fp, false, part);
            f.insert(f.end(), fp.begin(), fp.end());
          }
        }
        double xl = 65000.0;                               // (as unet_prepare: |conv| <= sum |w| |x| + |shift|)
        for (int co = 0; co < op.cout; ++co) {
          double sw = 0.0;
          for (int tap = 0; tap < 27; ++tap) sw += std::fabs((double)A[op.w_off + (size_t)tap * op.cout + co]);
          sw *= std::fabs((double)A[op.scale_off + co]);
          const double sh = std::fabs((double)A[op.shift_off + co]);
          if (!(sh < 65000.0)) return fpl_fail_range(ctx, "the first layer's shift exceeds the IEEE-half range");
          if (sw > 0.0) xl = std::min(xl, (65000.0 - sh) / sw);
        }
        cw.xlim = (float)xl;
      } else {
        fpl_pack_frags(A + op.w_off, scale.data(), 27, 1, op.cout, 2, 1, SLOT_STEM, &f, true);
      }
      cw.cin_p = 1; cw.cout_p = 32;
      for (int co = 0; co < 32; ++co) shifts.push_back(A[op.shift_off + co]);
    } else if (op.k == 3 && op.cin == 1) {                 // first layer: [part][b], SLOT_STEM, interleaved rows
      std::vector<float> scale(A + op.scale_off, A + op.scale_off + op.cout);
      for (int part = 0; part < PM; ++part) {
        std::vector<uint16_t> fp;
        fpl_pack_frags(A + op.w_off, scale.data(), 27, 1, op.cout, 2, 1, SLOT_STEM, &fp, true, part);
        f.insert(f.end(), fp.begin(), fp.end());
      }
      cw.cin_p = 1; cw.cout_p = 32;
      for (int co = 0; co < 32; ++co) shifts.push_back(co < op.cout ? A[op.shift_off + co] : 0.f);
    } else {
      std::vector<int> rows;
      if (op.k == 3) {
        std::vector<GxView> v;
        FPL_REQUIRE(ctx, gx_views(prog, pl, op.src0, &v), "gx: unresolved convolution input");
        int c0 = 0;
        for (auto &s : v) {
          const int C = pl.chan[s.base], nch = (C + RCH - 1) / RCH;
          for (int r = 0; r < nch * RCH; ++r) rows.push_back(r < C ? c0 + r : -1);
          c0 += C;
        }
      } else {
        for (int r = 0; r < gx_pad32(op.cin); ++r) rows.push_back(r < op.cin ? r : -1);
      }
      const int cout_p = op.cout > 64 ? op.cout : gx_pad32(op.cout);
      std::vector<float> Ap;
      fpl_op opp;
      gx_padded_op(A, op, rows, cout_p, &Ap, &opp);
      cw.cin_p = opp.cin; cw.cout_p = cout_p;
      const int pi = op.k == 1 ? pl.prod[op.src0] : -1;
      if (pi >= 0 && gx_chain_c1(prog, pl, pi) == (int)i) {
        // chained behind the first layer (unet_stem_c1): real channels as k-slots, [part][b]
        std::vector<float> scale(Ap.begin() + opp.scale_off, Ap.begin() + opp.scale_off + cout_p);
        for (int part = 0; part < PM; ++part) {
          std::vector<uint16_t> fp;
          fpl_pack_frags(Ap.data(), scale.data(), 1, 32, cout_p, 2, 1, SLOT_SPATIAL, &fp, true, part);
          f.insert(f.end(), fp.begin(), fp.end());
        }
      } else if (op.k == 3 && cout_p > 64) {
        std::vector<uint16_t> h0, h1;
        pack_conv3(Ap.data(), opp, 0, 64, false, &h0);
        pack_conv3(Ap.data(), opp, 64, 64, false, &h1);
        cw.half_bytes = h0.size() * sizeof(uint16_t);
        f = h0;
        f.insert(f.end(), h1.begin(), h1.end());
      } else if (op.k == 3) {
        pack_conv3(Ap.data(), opp, 0, cout_p, false, &f);
        // leading upsampled chunks, plain ones behind them: the z taps of the upsampled part pre-summed per
        // output-plane parity (two thirds of its MFMAs and weight fragments)
        std::vector<GxView> v;
        gx_views(prog, pl, op.src0, &v);
        int n_ups = 0, total = 0;
        bool lead = true, ok = true;
        for (auto &sv : v) {
          const int nch = (pl.chan[sv.base] + RCH - 1) / RCH;
          if (sv.ups) { if (!lead) ok = false; n_ups += nch; } else lead = false;
          total += nch;
        }
        const int jh = gx_chain_head(prog, pl, (int)i);
        if (jh >= 0) {
          const fpl_op &c8 = prog->ops[jh], &c9 = prog->ops[pl.users[c8.dst][0]];
          std::vector<float> s8(A + c8.scale_off, A + c8.scale_off + 32), s9(A + c9.scale_off, A + c9.scale_off + 1);
          cw.head_chain = true;
          cw.off_w8 = all.size() * sizeof(uint16_t);
          for (int part = 0; part < PM; ++part) {          // B fragments built in registers: REAL channels as k-slots
            std::vector<uint16_t> fp;
            fpl_pack_frags(A + c8.w_off, s8.data(), 1, 32, 32, 2, 1, SLOT_SPATIAL, &fp, false, part);
            all.insert(all.end(), fp.begin(), fp.end());
          }
          cw.off_w9 = all.size() * sizeof(uint16_t);
          for (int part = 0; part < PM; ++part) {
            std::vector<uint16_t> fp;
            fpl_pack_frags(A + c9.w_off, s9.data(), 1, 32, 1, 1, 1, SLOT_CHAIN, &fp, false, part);
            all.insert(all.end(), fp.begin(), fp.end());
          }
        }
        bool pooled = false;
        for (int u : pl.users[op.dst]) pooled |= prog->ops[u].kind == FPL_OP_POOL;
        if (ok && n_ups > 0 && n_ups < total && !pooled) {
          std::vector<uint16_t> fp;
          pack_conv3_parity(Ap.data(), opp, cout_p, n_ups, &fp, &cw.wp_steps);
          cw.off_wp = all.size() * sizeof(uint16_t);
          cw.wp_stream = fp.size() / 2 * sizeof(uint16_t);
          all.insert(all.end(), fp.begin(), fp.end());
        }
      } else if (SPLIT) {
        pack_conv1(Ap.data(), opp, true, &f);
      } else {
        std::vector<float> scale(Ap.begin() + opp.scale_off, Ap.begin() + opp.scale_off + cout_p);
        fpl_pack_frags(Ap.data(), scale.data(), 1, opp.cin, cout_p, cout_p / 16, opp.cin / 32, SLOT_SPATIAL, &f, true);
      }
      shifts.insert(shifts.end(), Ap.begin() + opp.shift_off, Ap.begin() + opp.shift_off + cout_p);
    }
    while (shifts.size() % 4) shifts.push_back(0.f);
    cw.off_w = all.size() * sizeof(uint16_t);
    all.insert(all.end(), f.begin(), f.end());
  }
#ifdef FPL_F16
  for (uint16_t h : all)
    if ((h & 0x7C00u) == 0x7C00u) {
      const char *msg = "a folded weight exceeds the IEEE-half range (65504); use precision "
                        "bf16, f32 or 'auto' for this network";
      return SPLIT ? fpl_fail_range(ctx, "%s", msg) : fpl_fail(ctx, "%s", msg);
    }
#endif
  if (st->frags) FPL_HIP(ctx, hipFree(st->frags));
  if (st->shifts) FPL_HIP(ctx, hipFree(st->shifts));
  st->frags = nullptr; st->shifts = nullptr;
  FPL_HIP(ctx, hipMalloc((void **)&st->frags, std::max<size_t>(all.size(), 8) * sizeof(uint16_t)));
  FPL_HIP(ctx, hipMalloc((void **)&st->shifts, std::max<size_t>(shifts.size(), 4) * sizeof(float)));
  FPL_HIP(ctx, hipMemcpy(st->frags, all.data(), all.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  FPL_HIP(ctx, hipMemcpy(st->shifts, shifts.data(), shifts.size() * sizeof(float), hipMemcpyHostToDevice));
  st->version = prog->arena_version;
  return 0;
}

// ---- host: the forward pass ----------------------------------------------------------------------

template <int CIN, int MB>
void gx_launch_conv1(fpl_ctx *ctx, Conv1Args &a) {
  constexpr int SMEM = (CIN / 32) * MB * PM * PM * 1024;
  static bool attr_set[FPL_MAX_DEVICES] = {false};
  if (!attr_set[ctx->device % FPL_MAX_DEVICES]) {
    hipFuncSetAttribute((const void *)FPLK(conv1)<CIN, MB, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set[ctx->device % FPL_MAX_DEVICES] = true;
  }
  const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(a.M, 64), (int64_t)ctx->n_cu * 8);
  FPLK(conv1)<CIN, MB, 0><<<grid, 256, SMEM, ctx->stream>>>(a);
}

int gx_forward(fpl_ctx *ctx, fpl_program *prog, const float *in, int n, int T, float *out, const FplTileIO *io) {
  GxPlan pl;
  gx_plan(prog, &pl);
  GxState *st;
  FPL_TRY(gx_prepare(ctx, prog, pl, &st));
  snprintf(ctx->last_path, sizeof(ctx->last_path), SPLIT ? "graph_split_f16" : "graph_mfma_" FPL_PREC_STR);
  const int32_t tin[3] = {T, T, T};
  std::vector<TensorShape> shp;
  FPL_TRY(fpl_infer_shapes(ctx, prog, tin, &shp));
  DevTemp tmp(ctx);
  unsigned *flag = nullptr;
  if (SPLIT) FPL_TRY(fpl_range_flag(ctx, &flag));
  const unsigned char *F = st->frags;
  const float *S = st->shifts;
  hipStream_t stm = ctx->stream;
  auto cube = [](int d) { return (int64_t)d * d * d; };
  const int NT = prog->n_tensors;
  std::vector<h16_t *> buf(NT, nullptr);
  std::vector<int> dim(NT, 0), cp(NT, 0);
  std::vector<char> done(prog->ops.size(), 0);
  for (int t = 0; t < NT; ++t) {
    FPL_REQUIRE(ctx, shp[t].d == shp[t].h && shp[t].h == shp[t].w && shp[t].d > 0, "gx: tensor %d is not a cube", t);
    dim[t] = shp[t].d;
  }
  // a materialised tensor: chunk planes of its padded channels + the read slack of the conv3 tile loader
  auto balloc = [&](int t) -> int {
    cp[t] = gx_pad32(pl.chan[t]);
    const int64_t elems = (int64_t)n * cube(dim[t]) * cp[t];
    const size_t slack = ((size_t)5 * dim[t] * dim[t] + 11 * dim[t] + 18) * cp[t] * PM * 2;      // (rows: up to R + 3, R = 8)
    void *q;
    FPL_TRY(tmp.alloc((size_t)elems * PM * 2 + slack + 64, &q));
    buf[t] = (h16_t *)q;
    return 0;
  };
  auto plane = [&](int t) { return (int64_t)n * cube(dim[t]) * CC; };
  // the MaxPooling3D that reads tensor t (fused into t's producer where it can be)
  auto pool_user = [&](int t) -> int {
    for (int u : pl.users[t]) if (prog->ops[u].kind == FPL_OP_POOL) return u;
    return -1;
  };
  for (size_t i = 0; i < prog->ops.size(); ++i) {
    const fpl_op &op = prog->ops[i];
    if (done[i]) continue;
    const GxConvW &cw = st->conv[i];
    switch (op.kind) {
      case FPL_OP_UP: case FPL_OP_CROP: case FPL_OP_CONCAT:
        break;                                             // views
      case FPL_OP_CONV: {
        const int jc = gx_chain_c1(prog, pl, (int)i), j3 = gx_chain_c3(prog, pl, (int)i);
        if (j3 >= 0) {                                     // conv3 1 -> 32 inside the tile loader of conv3 32 -> 32 + pool
          const fpl_op &c = prog->ops[j3];
          const int pu = pool_user(c.dst);
          const GxConvW &cj = st->conv[j3];
          FPL_TRY(balloc(c.dst));
          FPL_TRY(balloc(prog->ops[pu].dst));
          Conv3Args a;
          memset(&a, 0, sizeof(a));
          a.w = F + cj.off_w; a.shift = S + cj.off_s; a.relu = 1;
          a.out = buf[c.dst]; a.OD = a.OH = a.OW = dim[c.dst];
          FPL_REQUIRE(ctx, dim[op.dst] == T - 2 && dim[c.dst] == T - 4, "gx: first pair shapes");
          a.ncc = 32 / RCH;
          for (int cc = 0; cc < a.ncc; ++cc) a.src[cc] = make_src(nullptr, T - 2, CC, 0, 1, 0);
          a.raw = in; a.T = T;
          a.wstem = (const h16x8 *)(F + cw.off_w); a.shstem = S + cw.off_s;
          a.pool_out = buf[prog->ops[pu].dst];
          a.flag = flag; a.xlim = cw.xlim;
          done[j3] = 1; done[pu] = 1;
          FPL_TRY((launch_conv3<2, true, true, false, 8>(ctx, a, n, "gx_stem_conv3_32_32_pool")));
        } else if (jc >= 0) {                                     // conv3 1 -> C, conv1 C -> C', pool: one kernel
          const fpl_op &c1op = prog->ops[jc];
          const int pu = pool_user(c1op.dst);
          StemC1Args a;
          a.raw = in; a.T = T;
          a.wstem = (const h16x8 *)(F + cw.off_w); a.shstem = S + cw.off_s;
          a.w1 = (const h16x8 *)(F + st->conv[jc].off_w); a.sh1 = S + st->conv[jc].off_s;
          a.D = dim[op.dst];
          FPL_REQUIRE(ctx, a.D == T - 2 && dim[c1op.dst] == a.D && dim[prog->ops[pu].dst] == a.D / 2, "gx: chained first stage shapes");
          FPL_TRY(balloc(c1op.dst));
          FPL_TRY(balloc(prog->ops[pu].dst));
          a.c1 = buf[c1op.dst]; a.p1 = buf[prog->ops[pu].dst];
          a.c1plane = plane(c1op.dst); a.p1plane = plane(prog->ops[pu].dst);
          a.flag = flag;
          a.zblocks = (int)ceil_div64(a.D, 4); a.nbx = (int)ceil_div64(a.D, 16); a.nby = (int)ceil_div64(a.D, 4);
          done[jc] = 1; done[pu] = 1;
          TimedLaunch tl(ctx, "gx_stem_conv3_conv1_pool");
          FPLK(unet_stem_c1)<<<(unsigned)((int64_t)a.nbx * a.nby * n * a.zblocks), 256, 0, stm>>>(a);
        } else if (op.k == 3 && op.cin == 1) {
          const int pu = pool_user(op.dst);
          const bool need_c1 = pl.users[op.dst].size() > (pu >= 0 ? 1u : 0u);
          GxStemArgs a;
          a.raw = in; a.T = T; a.wstem = (const h16x8 *)(F + cw.off_w); a.shstem = S + cw.off_s;
          a.c1 = a.p1 = nullptr; a.c1plane = a.p1plane = 0;
          a.D = dim[op.dst];
          FPL_REQUIRE(ctx, a.D == T - 2, "gx: first layer output %d for tile %d", a.D, T);
          if (need_c1) { FPL_TRY(balloc(op.dst)); a.c1 = buf[op.dst]; a.c1plane = plane(op.dst); }
          if (pu >= 0) {
            const int pt = prog->ops[pu].dst;
            FPL_REQUIRE(ctx, dim[pt] == a.D / 2, "gx: pooled size");
            FPL_TRY(balloc(pt)); a.p1 = buf[pt]; a.p1plane = plane(pt);
            done[pu] = 1;
          }
          a.flag = flag;
          a.zblocks = (int)ceil_div64(a.D, 4); a.nbx = (int)ceil_div64(a.D, 16); a.nby = (int)ceil_div64(a.D, 4);
          TimedLaunch tl(ctx, "gx_stem_conv3");
          FPLK(gx_stem)<<<(unsigned)((int64_t)a.nbx * a.nby * n * a.zblocks), 256, 0, stm>>>(a);
        } else if (op.k == 3) {
          std::vector<GxView> v;
          FPL_REQUIRE(ctx, gx_views(prog, pl, op.src0, &v), "gx: unresolved convolution input");
          const int jh = cw.head_chain && io ? gx_chain_head(prog, pl, (int)i) : -1;
          if (jh < 0) FPL_TRY(balloc(op.dst));
          const int od = dim[op.dst];
          int pu = op.act == FPL_ACT_RELU && cw.cout_p <= 64 ? pool_user(op.dst) : -1;
          if (pu >= 0) { FPL_TRY(balloc(prog->ops[pu].dst)); done[pu] = 1; }
          for (int h = 0; h < (cw.cout_p > 64 ? 2 : 1); ++h) {
            Conv3Args a;
            a.w = F + cw.off_w + (h ? cw.half_bytes : 0); a.shift = S + cw.off_s + 64 * h; a.relu = op.act == FPL_ACT_RELU;
            a.out = jh >= 0 ? nullptr : buf[op.dst] + (int64_t)(64 * PM / CC) * h * plane(op.dst);
            a.oplane = a.pplane = 0; a.OD = a.OH = a.OW = od; a.ncc = 0; a.zblocks = 0;
            a.raw = nullptr; a.T = 0; a.wstem = nullptr; a.shstem = nullptr;
            a.pool_out = pu >= 0 ? buf[prog->ops[pu].dst] : nullptr;
            a.transposed = 0; a.xorg = 0; a.main_w = 0;
            memset(&a.io, 0, sizeof(a.io)); a.w8 = a.w9 = nullptr; a.sh8 = nullptr; a.bias9 = 0.f;
            a.flag = flag; a.xlim = 0.f;
            a.parity = 0; a.total_steps = 0; a.wstream = 0; a.planar = 0;
            for (auto &s : v) {
              const int b = s.base, nch = (pl.chan[b] + RCH - 1) / RCH;
              FPL_REQUIRE(ctx, buf[b], "gx: source tensor %d not computed", b);
              const int sd = s.ups ? 2 * dim[b] : dim[b] - 2 * s.crop;
              FPL_REQUIRE(ctx, sd == od + 2, "gx: source %d (edge %d) does not fit output edge %d", b, sd, od);
              for (int cc = 0; cc < nch; ++cc)
                a.src[a.ncc++] = make_src(buf[b] + cc * plane(b), dim[b], CC, 0, s.ups ? 2 : 1, s.crop);
            }
            const char *name = cw.cout_p <= 32 ? "gx_conv3_32" : cw.cout_p <= 64 ? "gx_conv3_64" : "gx_conv3_128";
            if (jh >= 0) {
              // conv1 32 -> 32, conv1 32 -> 1, sigmoid and the store into the volume in the epilogue
              const int j9 = pl.users[prog->ops[jh].dst][0];
              a.io = *io;
              a.w8 = (const h16x8 *)(F + cw.off_w8); a.sh8 = S + st->conv[jh].off_s;
              a.w9 = (const h16x8 *)(F + cw.off_w9); a.bias9 = st->conv[j9].bias;
              done[jh] = 1; done[j9] = 1;
              if (cw.wp_steps) {
                a.w = F + cw.off_wp; a.parity = 1; a.wstream = (int64_t)cw.wp_stream; a.total_steps = cw.wp_steps;
                FPL_TRY((launch_conv3<2, false, false, true, 6, true>(ctx, a, n, "gx_conv3_32_head")));
              } else {
                FPL_TRY((launch_conv3<2, false, false, true, 6>(ctx, a, n, "gx_conv3_32_head")));
              }
            } else if (cw.wp_steps && pu < 0 && cw.cout_p <= 64) {
              a.w = F + cw.off_wp; a.parity = 1; a.wstream = (int64_t)cw.wp_stream; a.total_steps = cw.wp_steps;
              if (cw.cout_p <= 32) FPL_TRY((launch_conv3<2, false, false, false, 6, true>(ctx, a, n, name)));
              else FPL_TRY((launch_conv3<4, false, false, false, 4, true>(ctx, a, n, name)));
            } else if (cw.cout_p <= 32) {
              // 32 outputs: 8 / 6 rows per wave where the layer is tall enough - a weight fragment then
              // feeds 8 / 6 MFMAs instead of 4 (conv_mfma.hip, Geo<R>)
              if (pu >= 0 && od >= 32) FPL_TRY((launch_conv3<2, false, true, false, 8>(ctx, a, n, name)));
              else if (pu >= 0) FPL_TRY((launch_conv3<2, false, true>(ctx, a, n, name)));
              else if (od >= 24) FPL_TRY((launch_conv3<2, false, false, false, 6>(ctx, a, n, name)));
              else FPL_TRY((launch_conv3<2>(ctx, a, n, name)));
            } else {
              if (pu >= 0) FPL_TRY((launch_conv3<4, false, true>(ctx, a, n, name)));
              else FPL_TRY((launch_conv3<4>(ctx, a, n, name)));
            }
          }
        } else if (op.cout == 1) {                         // the head
          int b;
          FPL_REQUIRE(ctx, gx_plain(prog, pl, op.src0, &b) && buf[b], "gx: head input");
          GxHeadArgs a;
          a.x = buf[b]; a.plane = plane(b); a.n = n; a.d = dim[b]; a.C = op.cin; a.s = prog->stride[0];
          a.w = S + cw.off_s;
          a.bias = cw.bias;
          memset(&a.io, 0, sizeof(a.io));
          a.use_io = io != nullptr;
          if (io) a.io = *io;
          a.out = out;
          FPL_REQUIRE(ctx, io || out, "gx: no output");
          const int64_t nv = (int64_t)n * cube(a.d);
          TimedLaunch tl(ctx, "gx_head");
          FPLK(gx_head)<<<(unsigned)ceil_div64(nv, 256), 256, 0, stm>>>(a);
        } else {                                           // 1x1x1 convolution
          int b;
          FPL_REQUIRE(ctx, gx_plain(prog, pl, op.src0, &b) && buf[b], "gx: conv1 input");
          int me = 0;
          const int fa = gx_fuse_add(prog, pl, (int)i, &me);
          Conv1Args a;
          a.in = buf[b]; a.M = (int64_t)n * cube(dim[b]); a.plane = a.M * CC;
          a.w = F + cw.off_w; a.shift = S + cw.off_s;
          a.w_tail = nullptr; a.bias_tail = 0.f; a.out_f32 = nullptr; a.flag = flag;
          a.relu = op.act == FPL_ACT_RELU;
          FPL_REQUIRE(ctx, dim[b] == dim[op.dst] && cp[b] == cw.cin_p, "gx: conv1 shapes");
          if (fa >= 0) {                                    // out = act(crop(conv1(x)) + other): the Add in the epilogue
            const fpl_op &ad = prog->ops[fa];
            std::vector<GxView> va, vb;
            gx_views(prog, pl, ad.src0, &va);
            gx_views(prog, pl, ad.src1, &vb);
            const GxView &mine = me == 0 ? va[0] : vb[0], &other = me == 0 ? vb[0] : va[0];
            FPL_REQUIRE(ctx, buf[other.base], "gx: add operand not computed");
            FPL_TRY(balloc(ad.dst));
            FPL_REQUIRE(ctx, cp[ad.dst] == cw.cout_p && cp[other.base] == cw.cout_p, "gx: fused add widths");
            a.out = buf[ad.dst]; a.oplane = plane(ad.dst);
            a.add = buf[other.base]; a.aplane = plane(other.base);
            a.din = dim[b]; a.dout = dim[ad.dst]; a.da = dim[other.base]; a.ocrop = mine.crop; a.acrop = other.crop;
            FPL_REQUIRE(ctx, a.din - 2 * a.ocrop == a.dout && a.da - 2 * a.acrop == a.dout, "gx: fused add shapes");
            a.relu = ad.act == FPL_ACT_RELU;
            done[fa] = 1;
          } else {
            FPL_TRY(balloc(op.dst));
            a.out = buf[op.dst];
          }
          TimedLaunch tl(ctx, fa >= 0 ? "gx_conv1_add" : "gx_conv1");
          const int key = cw.cin_p * 1000 + cw.cout_p;
          switch (key) {
            case 32032: gx_launch_conv1<32, 2>(ctx, a); break;
            case 32064: gx_launch_conv1<32, 4>(ctx, a); break;
            case 64032: gx_launch_conv1<64, 2>(ctx, a); break;
            case 64064: gx_launch_conv1<64, 4>(ctx, a); break;
            case 128032: gx_launch_conv1<128, 2>(ctx, a); break;
            case 128064: gx_launch_conv1<128, 4>(ctx, a); break;
            default: return fpl_fail(ctx, "gx: no conv1 kernel for %d -> %d channels", cw.cin_p, cw.cout_p);
          }
        }
        break;
      }
      case FPL_OP_POOL: {
        int b;
        FPL_REQUIRE(ctx, gx_plain(prog, pl, op.src0, &b) && buf[b], "gx: pool input");
        FPL_TRY(balloc(op.dst));
        const int dp = dim[op.dst];
        FPL_REQUIRE(ctx, dp == dim[b] / 2, "gx: pooled size");
        const int64_t no = (int64_t)n * cube(dp) * 4;
        TimedLaunch tl(ctx, "gx_pool");
        for (int ck = 0; ck < cp[b] * PM / CC; ++ck)
          FPLK(pool2_h16)<<<(unsigned)ceil_div64(no, 256), 256, 0, stm>>>(
              (const u32x4 *)(buf[b] + ck * plane(b)), (u32x4 *)(buf[op.dst] + ck * plane(op.dst)), no, dim[b], 4, dp);
        break;
      }
      case FPL_OP_ADD: {
        std::vector<GxView> va, vb;
        FPL_REQUIRE(ctx, gx_views(prog, pl, op.src0, &va) && gx_views(prog, pl, op.src1, &vb), "gx: add inputs");
        FPL_TRY(balloc(op.dst));
        GxAddArgs a;
        a.a = buf[va[0].base]; a.b = buf[vb[0].base]; a.out = buf[op.dst];
        FPL_REQUIRE(ctx, a.a && a.b, "gx: add inputs not computed");
        a.n = n; a.da = dim[va[0].base]; a.db = dim[vb[0].base]; a.d = dim[op.dst];
        a.crop_a = va[0].crop; a.crop_b = vb[0].crop;
        FPL_REQUIRE(ctx, a.da - 2 * a.crop_a == a.d && a.db - 2 * a.crop_b == a.d, "gx: add shapes");
        a.nplanes = cp[op.dst] * PM / CC; a.relu = op.act == FPL_ACT_RELU;
        a.aplane = plane(va[0].base); a.bplane = plane(vb[0].base); a.oplane = plane(op.dst);
        a.flag = flag;
        const int64_t tot = (int64_t)n * cube(a.d) * (SPLIT ? 2 : 4) * a.nplanes;
        TimedLaunch tl(ctx, "gx_add");
        FPLK(gx_add)<<<(unsigned)ceil_div64(tot, 256), 256, 0, stm>>>(a);
        break;
      }
      default:
        return fpl_fail(ctx, "gx: op kind %d", op.kind);
    }
  }
  FPL_HIP(ctx, hipGetLastError());
  return 0;
}
