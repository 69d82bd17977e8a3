"""CPU restatement (torch-CPU fp32) of the Keras layer arithmetic used by the
reference's models.  TEST INFRASTRUCTURE - see oracle/__init__.py.

PARITY UNPINNED: Keras/TensorFlow are absent and unpinned
(`/root/reference/conda-recipe/meta.yaml:18-19`); semantics restated from the
Keras 2 documentation of the layers the reference instantiates:

  Conv3D            valid padding, stride 1, cross-correlation, kernel
                    (kd,kh,kw,Cin,Cout), channels-last   (fplmodels.py:110-133)
  BatchNormalization axis -1, eps 1e-3; inference y = g*(x-m)/sqrt(v+eps)+b
                                                          (fplmodels.py:67-71)
  MaxPooling3D(2)   stride 2, valid (floor)              (fplmodels.py:114,120)
  UpSampling3D(n)   nearest repeat                        (fplnetwork.py:100-105)
  Cropping3D, concatenate([up, skip]) on channels        (fplmodels.py:284-291)
  Dropout           identity at inference
  sigmoid head      1/(1+exp(-x))                         (fplmodels.py:133,297)

Weights arrive as the Keras `get_weights()` list (creation order).
Arrays are channels-last `(N, D, H, W, C)`.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def _t(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype)


class _W:
    """cursor over a Keras-ordered weight list"""

    def __init__(self, weights, dtype):
        self.w = list(weights)
        self.i = 0
        self.dtype = dtype

    def take(self, n=1):
        out = [_t(a, self.dtype) for a in self.w[self.i:self.i + n]]
        self.i += n
        return out if n > 1 else out[0]

    def done(self):
        assert self.i == len(self.w), 'unused weights: %d of %d' % (
            self.i, len(self.w))


def conv3d_valid(x, kernel, bias=None):
    """x (N,C,D,H,W) torch; kernel (kd,kh,kw,Cin,Cout) Keras layout"""
    w = kernel.permute(4, 3, 0, 1, 2).contiguous()
    return F.conv3d(x, w, bias)


def bn_infer(x, gamma, beta, mean, var):
    shp = (1, -1, 1, 1, 1)
    return (x - mean.view(shp)) / torch.sqrt(var.view(shp) + BN_EPS) \
        * gamma.view(shp) + beta.view(shp)


def conv_bn_relu(x, w, k_bias=False):
    y = conv3d_valid(x, w.take())
    return torch.relu(bn_infer(y, *w.take(4)))


def maxpool2(x):
    return F.max_pool3d(x, 2, 2)


def upsample(x, n):
    n = (n, n, n) if np.isscalar(n) else n
    for ax, f in zip((2, 3, 4), n):
        x = torch.repeat_interleave(x, int(f), dim=ax)
    return x


def crop(x, c):
    return x[:, :, c:x.shape[2] - c, c:x.shape[3] - c, c:x.shape[4] - c]


def vgg_like_forward(x, weights, upsample_stride=None, dtype=torch.float32):
    """fplmodels.py:110-133 (+ UpSampling3D(rf_stride) of fplnetwork.py:100-105
    when `upsample_stride` is given).  x: (N,D,H,W,1) numpy -> (N,d,h,w,1)"""
    w = _W(weights, dtype)
    h = _t(x, dtype).permute(0, 4, 1, 2, 3)
    for _ in range(2):
        h = conv_bn_relu(h, w)          # 3x3x3, 48
        h = conv_bn_relu(h, w)          # 1x1x1, 48
        h = maxpool2(h)
    h = conv_bn_relu(h, w)              # 3x3x3, 48
    h = conv_bn_relu(h, w)              # 1x1x1, 96  (Dropout = identity)
    h = conv_bn_relu(h, w)              # 1x1x1, 96
    kern, bias = w.take(2)
    h = torch.sigmoid(conv3d_valid(h, kern, bias))
    w.done()
    if upsample_stride is not None:
        h = upsample(h, upsample_stride)
    return h.permute(0, 2, 3, 4, 1).contiguous().numpy()


def unet_like2_forward(x, weights, dtype=torch.float32):
    """fplmodels.py:268-297.  x: (N,D,H,W,1), D,H,W = 0 mod 4"""
    w = _W(weights, dtype)
    h = _t(x, dtype).permute(0, 4, 1, 2, 3)
    c1 = conv_bn_relu(conv_bn_relu(h, w), w)                 # 32, 32 (3x3x3)
    c2 = conv_bn_relu(conv_bn_relu(maxpool2(c1), w), w)      # 64, 64 (3x3x3)
    c3 = conv_bn_relu(maxpool2(c2), w)                       # 128 (1x1x1)
    u4 = torch.cat([upsample(c3, 2), c2], dim=1)
    c4 = conv_bn_relu(conv_bn_relu(u4, w), w)                # 64 (3), 64 (1)
    u5 = torch.cat([upsample(c4, 2), crop(c1, 6)], dim=1)
    c5 = conv_bn_relu(conv_bn_relu(u5, w), w)                # 32 (3), 32 (1)
    out = torch.sigmoid(conv3d_valid(c5, w.take()))          # no bias
    w.done()
    return out.permute(0, 2, 3, 4, 1).contiguous().numpy()


def graph_forward(graph, x, dtype=torch.float32, upsample_stride=None):
    """generic interpreter over a `flypylib_amd.program.LayerGraph` (used for the
    architectures that have no hand-written restatement above)"""
    vals = {}
    W = [_t(a, dtype) for a in graph.weights]
    for n in graph.nodes:
        a = [vals[i] for i in n.inputs]
        if n.kind == 'input':
            v = _t(x, dtype).permute(0, 4, 1, 2, 3)
        elif n.kind == 'conv':
            bias = W[n.weight_slots[1]] if n.attrs['use_bias'] else None
            v = conv3d_valid(a[0], W[n.weight_slots[0]], bias)
            if n.attrs['activation'] == 'relu':
                v = torch.relu(v)
            elif n.attrs['activation'] == 'sigmoid':
                v = torch.sigmoid(v)
        elif n.kind == 'bn':
            v = bn_infer(a[0], *[W[s] for s in n.weight_slots])
        elif n.kind == 'relu':
            v = torch.relu(a[0])
        elif n.kind == 'pool':
            v = maxpool2(a[0])
        elif n.kind == 'up':
            v = upsample(a[0], n.attrs['n'])
        elif n.kind == 'crop':
            c = n.attrs['c']
            v = a[0][:, :, c[0][0]:a[0].shape[2] - c[0][1],
                     c[1][0]:a[0].shape[3] - c[1][1],
                     c[2][0]:a[0].shape[4] - c[2][1]]
        elif n.kind == 'concat':
            v = torch.cat(a, dim=1)
        elif n.kind == 'add':
            v = a[0] + a[1]
        elif n.kind == 'drop':
            v = a[0]
        else:
            raise NotImplementedError(n.kind)
        vals[n.idx] = v
    out = vals[graph.output.idx]
    if upsample_stride is not None:
        out = upsample(out, upsample_stride)
    return out.permute(0, 2, 3, 4, 1).contiguous().numpy()


def conv3d_valid_numpy(x, kernel):
    """loop restatement for known-answer tests: x (D,H,W,Cin) float64,
    kernel (kd,kh,kw,Cin,Cout) -> (D-kd+1, H-kh+1, W-kw+1, Cout)"""
    kd, kh, kw, _, cout = kernel.shape
    D, H, Wd = (x.shape[0] - kd + 1, x.shape[1] - kh + 1, x.shape[2] - kw + 1)
    out = np.zeros((D, H, Wd, cout), np.float64)
    for a in range(kd):
        for b in range(kh):
            for c in range(kw):
                out += np.tensordot(x[a:a + D, b:b + H, c:c + Wd, :],
                                    kernel[a, b, c].astype(np.float64),
                                    axes=([3], [0]))
    return out


# ---- bf16 emulation of the fused MI355X kernels (csrc/vgg_fused.hip) -----------
def _bf16_round(t, kind='bf16'):
    """round-to-nearest-even to the operand type of the fused kernels, kept in float32:
    bfloat16, IEEE half (kind='f16'), or the split representation of csrc/vgg_split.hip
    (kind='split': hi = half(v), lo = half(v - hi), value hi + lo, ~22 significant bits)"""
    if kind == 'split':
        hi = t.to(torch.float16).to(torch.float32)
        return hi + (t - hi).to(torch.float16).to(torch.float32)
    return t.to(torch.float16 if kind == 'f16' else torch.bfloat16).to(torch.float32)


def vgg_like_forward_bf16emu(x, weights, upsample_stride=None, kind='bf16'):
    """vgg_like with the rounding points of the fused bf16 kernels: BN scale is
    folded into the kernel before rounding it to bf16, BN shift stays fp32,
    activations are rounded to bf16 after every ReLU, accumulation is fp32.
    x: (N,D,H,W,1) float32 already normalised."""
    w = _W(weights, torch.float32)

    def _r(t):
        return _bf16_round(t, kind)
    h = _r(_t(x, torch.float32).permute(0, 4, 1, 2, 3))

    def block(h, pool):
        kern = w.take()
        g, b, m, v = w.take(4)
        s = g / torch.sqrt(v + BN_EPS)
        kf = _r(kern * s.view(1, 1, 1, 1, -1))
        y = conv3d_valid(h, kf) + (b - m * s).view(1, -1, 1, 1, 1)
        if pool:
            y = maxpool2(y)
        return _r(torch.relu(y))

    h = block(h, False)
    h = block(h, True)
    h = block(h, False)
    h = block(h, True)
    h = block(h, False)
    h = block(h, False)
    h = block(h, False)
    kern, bias = w.take(2)
    h = torch.sigmoid(conv3d_valid(h, _r(kern)) + bias.view(1, -1, 1, 1, 1))
    w.done()
    if upsample_stride is not None:
        h = upsample(h, upsample_stride)
    return h.permute(0, 2, 3, 4, 1).contiguous().numpy()


def unet_like2_forward_bf16emu(x, weights, kind='bf16'):
    """unet_like2 with the rounding points of the bf16 MFMA kernels
    (csrc/conv_mfma.hip): input and every post-ReLU activation rounded to bf16,
    BN scale folded into bf16 kernels, fp32 accumulation and shift."""
    w = _W(weights, torch.float32)

    def _r(t):
        return _bf16_round(t, kind)
    h = _r(_t(x, torch.float32).permute(0, 4, 1, 2, 3))

    def block(h):
        kern = w.take()
        g, b, m, v = w.take(4)
        s = g / torch.sqrt(v + BN_EPS)
        kf = _r(kern * s.view(1, 1, 1, 1, -1))
        y = conv3d_valid(h, kf) + (b - m * s).view(1, -1, 1, 1, 1)
        return _r(torch.relu(y))

    c1 = block(block(h))
    c2 = block(block(maxpool2(c1)))
    c3 = block(maxpool2(c2))
    c4 = block(block(torch.cat([upsample(c3, 2), c2], dim=1)))
    c5 = block(block(torch.cat([upsample(c4, 2), crop(c1, 6)], dim=1)))
    out = torch.sigmoid(conv3d_valid(c5, _r(w.take())))
    w.done()
    return out.permute(0, 2, 3, 4, 1).contiguous().numpy()
