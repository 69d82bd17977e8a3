"""GPU parity for the fused 16-bit path of vgg_like2 (csrc/vgg_fused.hip, vgg2_conv3 +
vgg_c5_tail; reference flypylib/fplmodels.py:138-172, the model of
scripts/fpl_cx1_0_vgg_4ss.py): against the fp32 oracle over the reference tile lattice
(rf 24, offset 10, stride 4, tile 100 -> 80), with f16 held to the north star's 1e-3 gate,
and against itself across chunkings and slabs (bit-exact)."""
import numpy as np
import pytest

from flypylib_amd import _capi, fplmodels, multi_gpu, synth
from oracle import cnn_oracle, infer_oracle

pytestmark = pytest.mark.gpu
TOL = {'bf16': 1.5e-2, 'f16': 1e-3}
PREC = {'bf16': _capi.PREC_BF16, 'f16': _capi.PREC_F16}
OFF = 10


def _net(seed, tile):
    g = fplmodels.vgg_like2(tile)[0]
    synth.synthetic_weights(g, seed)
    return g


def _oracle(g, img, tile):
    def f32(batch):
        return cnn_oracle.graph_forward(g, batch.astype(np.float32), upsample_stride=(4, 4, 4))
    return infer_oracle.infer_lattice(img, (tile,) * 3, (OFF,) * 3, f32)


@pytest.mark.parametrize('kind', ['bf16', 'f16'])
@pytest.mark.parametrize('shape,tile', [
    ((52, 47, 61), 36), ((44, 44, 44), 36), ((37, 36, 70), 36), ((110, 64, 90), 100)])
def test_vgg_like2_fused_matches_fp32_oracle(ctx, shape, tile, kind):
    g = _net(31, tile)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(11, shape)
    img = (u8.astype(np.float32) - np.float32(128)) / np.float32(33)
    got = prog.infer_volume(u8, (tile,) * 3, (OFF,) * 3, mean=128.0, std=33.0, precision=PREC[kind])
    exact = prog.infer_volume(u8, (tile,) * 3, (OFF,) * 3, mean=128.0, std=33.0,
                              precision=_capi.PREC_F32)
    ref = _oracle(g, img, tile)
    assert got.shape == shape and got.dtype == np.float32
    assert not got[:OFF].any() and not got[-OFF:].any()
    assert not got[:, :OFF].any() and not got[:, :, -OFF:].any()
    assert np.abs(exact - ref).max() < 1e-5           # the fp32 device path is the oracle
    d = np.abs(got - ref)
    assert d.max() < TOL[kind], '%s vs fp32 oracle: max %g' % (kind, d.max())
    assert ref[OFF:-OFF, OFF:-OFF, OFF:-OFF].std() > 1e-3
    ctx.timing(True)
    ctx.timing_reset()
    prog.infer_volume(u8, (tile,) * 3, (OFF,) * 3, mean=128.0, std=33.0, precision=PREC[kind])
    names = set(ctx.timing_get())
    ctx.timing(False)
    assert any(n.startswith('vgg2_stem_conv3_pool_') for n in names), names   # the fused path ran


def test_vgg_like2_float_input_chunks_and_slabs(ctx, monkeypatch):
    g = _net(32, 36)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    img = synth.hash_uniform_f32(5, (95, 50, 58)) * np.float32(4) - np.float32(2)
    kw = dict(precision=_capi.PREC_F16)
    whole = prog.infer_volume(img, (36,) * 3, (OFF,) * 3, **kw)
    ref = _oracle(g, img, 36)
    assert np.abs(whole - ref).max() < TOL['f16']
    # several Z chunks of the scratch tensors give the same bits
    monkeypatch.setenv('FPL_VGG_SCRATCH_MB', '1')
    assert np.array_equal(prog.infer_volume(img, (36,) * 3, (OFF,) * 3, **kw), whole)
    monkeypatch.delenv('FPL_VGG_SCRATCH_MB')
    # slabs of tile rows (the multi-GPU sharding) stitch to the same bits
    n_rows = multi_gpu.n_tile_rows(95, 36, OFF)
    parts = np.zeros_like(whole)
    for zb, ze in multi_gpu.slab_partition(n_rows, 3):
        prog.infer_volume(img, (36,) * 3, (OFF,) * 3, z_range=(zb, ze), dst=parts, **kw)
    assert np.array_equal(parts, whole)


def test_fplnetwork_vgg_like2_f16(ctx):
    """through the reference's API: FplNetwork(vgg_like2).infer"""
    from flypylib_amd import FplNetwork
    net = FplNetwork(fplmodels.vgg_like2, precision='f16')
    synth.synthetic_weights(net.train_single, 33)
    net.infer_sz = (36, 36, 36)
    net._set_infer()
    u8 = synth.em_volume_u8(12, (60, 41, 48))
    img = (u8.astype(np.float32) - np.float32(128)) / np.float32(33)
    got = net.infer(img)
    ref = _oracle(net.train_single, img, 36)
    assert np.abs(got - ref).max() < TOL['f16']


# ---- the split-half build (precision 'f16s', what 'auto' picks): fp32-grade ---------------
@pytest.mark.parametrize('shape,tile', [
    ((52, 47, 61), 36), ((44, 44, 44), 36), ((37, 36, 70), 36), ((110, 64, 90), 100)])
def test_vgg_like2_split_halves_match_fp32_oracle(ctx, shape, tile):
    g = _net(31, tile)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(11, shape)
    for mean, std in ((128.0, 33.0), (121.4, 29.3)):
        img = (u8.astype(np.float32) - np.float32(mean)) / np.float32(std)
        ctx.timing(True)
        ctx.timing_reset()
        got = prog.infer_volume(u8, (tile,) * 3, (OFF,) * 3, mean=mean, std=std, precision=_capi.PREC_AUTO)
        names = set(ctx.timing_get())
        ctx.timing(False)
        assert ctx.last_path() == 'vgg_split_f16' and 'vggs2_stem_conv3_pool' in names, names
        ref = _oracle(g, img, tile)
        assert not got[:OFF].any() and not got[-OFF:].any() and not got[:, :, -OFF:].any()
        d = np.abs(got - ref)
        assert d.max() < 1e-5, 'split halves vs fp32 oracle: max %g' % d.max()
        assert ref[OFF:-OFF, OFF:-OFF, OFF:-OFF].std() > 1e-3


def test_vgg_like2_split_float_input_chunks_and_slabs(ctx, monkeypatch):
    g = _net(32, 36)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    img = synth.hash_uniform_f32(5, (95, 50, 58)) * np.float32(4) - np.float32(2)
    kw = dict(precision=_capi.PREC_F16S)
    whole = prog.infer_volume(img, (36,) * 3, (OFF,) * 3, **kw)
    assert np.abs(whole - _oracle(g, img, 36)).max() < 1e-5
    monkeypatch.setenv('FPL_VGG_SCRATCH_MB', '2')
    assert np.array_equal(prog.infer_volume(img, (36,) * 3, (OFF,) * 3, **kw), whole)
    monkeypatch.delenv('FPL_VGG_SCRATCH_MB')
    n_rows = multi_gpu.n_tile_rows(95, 36, OFF)
    parts = np.zeros_like(whole)
    for zb, ze in multi_gpu.slab_partition(n_rows, 3):
        prog.infer_volume(img, (36,) * 3, (OFF,) * 3, z_range=(zb, ze), dst=parts, **kw)
    assert np.array_equal(parts, whole)
