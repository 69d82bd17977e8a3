"""The default precision ('auto') must be as safe as the reference's fp32 predict
(flypylib/fplnetwork.py:175-176): the split-half kernels it prefers carry every value as two
IEEE halves, range 65504.  A network whose folded weights or activations leave that range -
or a float volume with a voxel beyond it - has to come out of 'auto' exactly as the fp32
executor computes it (bit for bit: 'auto' reruns the call there), and explicit 'f16s' has
to fail instead of returning numbers.

The networks here are ordinary seeded ones with ONE layer blown up by a power of two F
(BatchNorm gamma and beta times F: its ReLU output is exactly F times the original) and the
NEXT layer's BatchNorm statistics adjusted to undo it (moving mean times F, variance times
F^2), so the probabilities stay an ordinary field and the comparison is not between two
saturated sigmoids."""
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


def _vgg(seed, tile=30):
    g = fplmodels.vgg_like(tile)[0]
    synth.synthetic_weights(g, seed)
    return g


# which layer leaves the half range, and which kernel has to notice:
#   0  conv3 1->48   - bounded on the host from sum |w| (no split path is tried at all)
#   1  conv1 48->48  - the stem's pooled store
#   2  conv3 48->48  - the mid kernel's accumulators
#   5  conv1 48->96  - the head's register chain
@pytest.mark.parametrize('layer,f', [(0, 2.0 ** 15), (1, 2.0 ** 16), (2, 2.0 ** 18), (3, 2.0 ** 16),
                                     (4, 2.0 ** 18), (5, 2.0 ** 16)])
def test_vgg_auto_falls_back_to_fp32_when_an_activation_leaves_the_half_range(ctx, layer, f):
    g = _vgg(51)
    clean = [a.copy() for a in g.get_weights()]
    prog = _capi.Program(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(21, (50, 47, 41))
    kw = dict(mean=128.0, std=33.0)
    base = prog.infer_volume(u8, (30,) * 3, (7,) * 3, precision=_capi.PREC_AUTO, **kw)
    assert ctx.last_path() == 'vgg_split_f16'
    _blow_up(g, layer, f)
    prog.set_weights_from(g)
    f32 = prog.infer_volume(u8, (30,) * 3, (7,) * 3, precision=_capi.PREC_F32, **kw)
    assert ctx.last_path() == 'mfma_f32'
    # the blown-up network is still the same function, up to fp32 rounding
    assert np.abs(f32 - base).max() < 1e-4 and f32[7:-7, 7:-7, 7:-7].std() > 1e-3
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
    _blow_up(g, 2, 2.0 ** 18)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(23, (60, 52, 47))
    kw = dict(mean=128.0, std=33.0)
    f32 = prog.infer_volume(u8, (36,) * 3, (10,) * 3, precision=_capi.PREC_F32, **kw)
    auto = prog.infer_volume(u8, (36,) * 3, (10,) * 3, precision=_capi.PREC_AUTO, **kw)
    assert ctx.last_path() == 'mfma_f32(range)' and np.array_equal(auto, f32)
    assert f32[10:-10, 10:-10, 10:-10].std() > 1e-3
    with pytest.raises(_capi.FplHipError, match='half range'):
        prog.infer_volume(u8, (36,) * 3, (10,) * 3, precision=_capi.PREC_F16S, **kw)


@pytest.mark.parametrize('model,tile,off,layer,f', [
    ('unet_like2', 36, 9, 2, 2.0 ** 18),      # conv3 32->64: a conv3 epilogue store
    ('unet_like2', 36, 9, 0, 2.0 ** 15),      # conv3 1->32: the host-side bound of the fused stem
    ('unet_like2', 36, 9, 7, 2.0 ** 18),      # conv3 96->32: the head's register chain
    ('unet_like', 30, 6, 0, 2.0 ** 15),       # unet_like's chained stem (checked in the kernel)
])
def test_unet_auto_falls_back(ctx, model, tile, off, layer, f):
    g = getattr(fplmodels, model)(tile)[0]
    synth.synthetic_weights(g, 54)
    prog = _capi.Program(ctx, g, (1, 1, 1))
    u8 = synth.em_volume_u8(24, (tile + 17, tile + 8, tile + 11))
    kw = dict(mean=128.0, std=33.0)
    prog.infer_volume(u8, (tile,) * 3, (off,) * 3, precision=_capi.PREC_AUTO, **kw)
    assert ctx.last_path() == 'unet_split_f16'
    _blow_up(g, layer, f)
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
