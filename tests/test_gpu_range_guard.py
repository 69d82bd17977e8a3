"""The default precision ('auto') must be as safe as the reference's fp32 predict
(flypylib/fplnetwork.py:175-176): the split-half kernels it prefers carry every value as two
IEEE halves, range 65504.  A network whose folded weights or activations leave that range -
or a float volume with a voxel beyond it - has to come out of 'auto' exactly as the fp32
executor computes it (bit for bit: 'auto' reruns the call there), and explicit 'f16s' has
to fail instead of returning numbers.

The networks here are ordinary seeded ones in which ONE channel of one layer is pushed out of
the half range (its BatchNorm beta plus 1e5) and the NEXT layer's moving means absorb the
constant that channel adds to its convolution, so the probabilities stay an ordinary field
(the comparison is not between two saturated sigmoids) and no folded WEIGHT leaves the range -
what has to notice is the kernel that splits that activation.  The first layer, whose outputs
are bounded on the host instead, is scaled as a whole."""
import numpy as np
import pytest

from flypylib_amd import _capi, fplmodels, synth

pytestmark = pytest.mark.gpu
EPS = np.float32(1e-3)          # Keras BatchNormalization default (fplmodels.py:67-71)


def _blow_up(g, layer, f):
    """conv-BN-ReLU block `layer` (0-based) scaled by f, block layer + 1 compensated"""
    w = [a.copy() for a in g.get_weights()]
    f = np.float32(f)
    w[5 * layer + 1] *= f                      # gamma
    w[5 * layer + 2] *= f                      # beta
    w[5 * (layer + 1) + 3] *= f                # next moving_mean
    w[5 * (layer + 1) + 4] = (w[5 * (layer + 1) + 4] + EPS) * f * f - EPS
    g.set_weights(w)


def _bump(g, layer, big=1e5, channel=0, consumers=None):
    """channel `channel` of conv-BN-ReLU block `layer` shifted up by `big` (far beyond 65504);
    the moving means of the blocks that read it - `consumers`: (block, index of that channel
    among the block's inputs), default the next block - take the constant
    big * sum_taps W[tap, index, :] out again"""
    w = [a.copy() for a in g.get_weights()]
    w[5 * layer + 2][channel] += np.float32(big)
    for nxt, idx in (consumers or [(layer + 1, channel)]):
        kern = w[5 * nxt]                      # (k, k, k, cin, cout)
        w[5 * nxt + 3] += np.float32(big) * kern[:, :, :, idx, :].sum(axis=(0, 1, 2))
    g.set_weights(w)


def _vgg(seed, tile=30):
    g = fplmodels.vgg_like(tile)[0]
    synth.synthetic_weights(g, seed)
    return g


# which layer leaves the half range, and which kernel has to notice:
#   0  conv3 1->48   - bounded on the host from sum |w| (no split pass is tried at all)
#   1  conv1 48->48  - the stem's pooled store
#   2  conv3 48->48  - the mid kernel's accumulators (register chain into conv1)
#   3  conv1 48->48  - the mid kernel's pooled store
#   4  conv3 48->48  - the tail's accumulators
#   5  conv1 48->96  - the head's register chain
@pytest.mark.parametrize('layer', [0, 1, 2, 3, 4, 5])
def test_vgg_auto_falls_back_to_fp32_when_an_activation_leaves_the_half_range(ctx, layer):
    g = _vgg(51)
    clean = [a.copy() for a in g.get_weights()]
    prog = _capi.Program(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(21, (50, 47, 41))
    kw = dict(mean=128.0, std=33.0)
    base = prog.infer_volume(u8, (30,) * 3, (7,) * 3, precision=_capi.PREC_AUTO, **kw)
    assert ctx.last_path() == 'vgg_split_f16'
    if layer == 0:
        _blow_up(g, 0, 2.0 ** 15)
    else:
        _bump(g, layer)
    prog.set_weights_from(g)
    f32 = prog.infer_volume(u8, (30,) * 3, (7,) * 3, precision=_capi.PREC_F32, **kw)
    assert ctx.last_path() == 'mfma_f32'
    assert f32[7:-7, 7:-7, 7:-7].std() > 1e-3          # still an ordinary probability field
    with pytest.raises(_capi.FplHipError, match='half range'):
        prog.infer_volume(u8, (30,) * 3, (7,) * 3, precision=_capi.PREC_F16S, **kw)
    for _ in range(2):          # the second call goes straight to fp32 (remembered per weight set)
        auto = prog.infer_volume(u8, (30,) * 3, (7,) * 3, precision=_capi.PREC_AUTO, **kw)
        assert ctx.last_path() == 'mfma_f32(range)'
        assert np.array_equal(auto, f32)
    # Z slabs (multi-GPU sharding) take the same route
    out = np.zeros_like(f32)
    for lo, hi in ((0, 1), (1, 3)):
        prog.infer_volume(u8, (30,) * 3, (7,) * 3, precision=_capi.PREC_AUTO, z_range=(lo, hi), dst=out, **kw)
    assert np.array_equal(out, f32)
    # new weights: the split path is tried again
    g.set_weights(clean)
    prog.set_weights_from(g)
    again = prog.infer_volume(u8, (30,) * 3, (7,) * 3, precision=_capi.PREC_AUTO, **kw)
    assert ctx.last_path() == 'vgg_split_f16' and np.array_equal(again, base)


def test_vgg_float_volume_with_a_voxel_beyond_the_half_range(ctx):
    """pre-normalised float volumes (what the reference's infer takes): one voxel at 1e6, one
    NaN-free but huge negative one.  'auto' == fp32 for that call only; the next volume runs
    on split halves again."""
    g = _vgg(52)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(22, (46, 50, 44))
    img = (u8.astype(np.float32) - np.float32(128)) / np.float32(33)
    bad = img.copy()
    bad[20, 21, 22] = 1e6
    bad[30, 11, 40] = -3e5
    f32 = prog.infer_volume(bad, (30,) * 3, (7,) * 3, precision=_capi.PREC_F32)
    auto = prog.infer_volume(bad, (30,) * 3, (7,) * 3, precision=_capi.PREC_AUTO)
    assert ctx.last_path() == 'mfma_f32(range)' and np.array_equal(auto, f32)
    with pytest.raises(_capi.FplHipError, match='input voxel'):
        prog.infer_volume(bad, (30,) * 3, (7,) * 3, precision=_capi.PREC_F16S)
    prog.infer_volume(img, (30,) * 3, (7,) * 3, precision=_capi.PREC_AUTO)
    assert ctx.last_path() == 'vgg_split_f16'


def test_vgg_like2_auto_falls_back(ctx):
    g = fplmodels.vgg_like2(36)[0]
    synth.synthetic_weights(g, 53)
    _bump(g, 2)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(23, (60, 52, 47))
    kw = dict(mean=128.0, std=33.0)
    f32 = prog.infer_volume(u8, (36,) * 3, (10,) * 3, precision=_capi.PREC_F32, **kw)
    auto = prog.infer_volume(u8, (36,) * 3, (10,) * 3, precision=_capi.PREC_AUTO, **kw)
    assert ctx.last_path() == 'mfma_f32(range)' and np.array_equal(auto, f32)
    assert f32[10:-10, 10:-10, 10:-10].std() > 1e-3
    with pytest.raises(_capi.FplHipError, match='half range'):
        prog.infer_volume(u8, (36,) * 3, (10,) * 3, precision=_capi.PREC_F16S, **kw)


# U-Net blocks (fplmodels.py:258-304): 0 conv3 1->32, 1 conv3 32->32 (c1: pooled, and the skip
# into block 7 behind the 64 upsampled channels), 2 conv3 32->64, 3 conv3 64->64 (c2: pooled, and
# the skip into block 5 behind the 128 upsampled channels), 4 conv1 64->128, 5 conv3 192->64,
# 6 conv1 64->64, 7 conv3 96->32, 8 conv1 32->32, 9 the sigmoid head
@pytest.mark.parametrize('model,tile,off,layer,consumers', [
    ('unet_like2', 36, 9, 0, None),                     # the host-side bound of the fused stem
    ('unet_like2', 36, 9, 1, [(2, 0), (7, 64)]),        # stem kernel: conv3 store + pooled store
    ('unet_like2', 36, 9, 2, None),                     # a conv3 epilogue store
    ('unet_like2', 36, 9, 3, [(4, 0), (5, 128)]),       # conv3 + pool
    ('unet_like2', 36, 9, 4, None),                     # conv1 64->128
    ('unet_like2', 36, 9, 6, None),                     # conv1 64->64
    ('unet_like2', 36, 9, 7, None),                     # the head's register chain
    ('unet_like', 30, 6, 0, None),                      # unet_like's chained stem (checked in the kernel)
])
def test_unet_auto_falls_back(ctx, model, tile, off, layer, consumers):
    g = getattr(fplmodels, model)(tile)[0]
    synth.synthetic_weights(g, 54)
    prog = _capi.Program(ctx, g, (1, 1, 1))
    u8 = synth.em_volume_u8(24, (tile + 17, tile + 8, tile + 11))
    kw = dict(mean=128.0, std=33.0)
    prog.infer_volume(u8, (tile,) * 3, (off,) * 3, precision=_capi.PREC_AUTO, **kw)
    assert ctx.last_path() == 'unet_split_f16'
    if layer == 0:
        _blow_up(g, 0, 2.0 ** 17)          # (no weight leaves the range: max |w| gamma / sigma ~ 0.2)
    else:
        _bump(g, layer, consumers=consumers)
    prog.set_weights_from(g)
    f32 = prog.infer_volume(u8, (tile,) * 3, (off,) * 3, precision=_capi.PREC_F32, **kw)
    auto = prog.infer_volume(u8, (tile,) * 3, (off,) * 3, precision=_capi.PREC_AUTO, **kw)
    assert ctx.last_path() == 'mfma_f32(range)' and np.array_equal(auto, f32)
    assert f32[off:-off, off:-off, off:-off].std() > 1e-3
    with pytest.raises(_capi.FplHipError, match='half range'):
        prog.infer_volume(u8, (tile,) * 3, (off,) * 3, precision=_capi.PREC_F16S, **kw)


def test_auto_with_weights_beyond_the_half_range(ctx):
    """a folded weight above 65504: 'auto' used to raise here (round 3)"""
    g = _vgg(55)
    w = g.get_weights()
    w[10] = w[10] * np.float32(1e7)            # the third conv's kernel
    g.set_weights(w)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(25, (46, 46, 46))
    kw = dict(mean=128.0, std=33.0)
    f32 = prog.infer_volume(u8, (30,) * 3, (7,) * 3, precision=_capi.PREC_F32, **kw)
    auto = prog.infer_volume(u8, (30,) * 3, (7,) * 3, precision=_capi.PREC_AUTO, **kw)
    assert ctx.last_path() == 'mfma_f32(range)' and np.array_equal(auto, f32)
    with pytest.raises(_capi.FplHipError, match='half range'):
        prog.infer_volume(u8, (30,) * 3, (7,) * 3, precision=_capi.PREC_F16S, **kw)
