"""GPU parity for the split-operand vgg_like path (csrc/vgg_split.hip, precision 'f16s'):
every operand as two IEEE halves (hi + lo), three MFMAs per product.  References: the
emulation oracle with the same rounding points (kind='split') and the plain fp32 oracle -
the path is held to fp32-grade agreement (1e-5; observed ~2e-6), two orders inside the
north star's 1e-3 gate."""
import os

import numpy as np
import pytest

from flypylib_amd import _capi, fplmodels, multi_gpu, synth
from oracle import cnn_oracle, infer_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5          # vs the fp32 oracle and vs the split emulation, max abs probability


def _net(seed, tile=30):
    g = fplmodels.vgg_like(tile)[0]
    synth.synthetic_weights(g, seed)
    return g


def _refs(g, img, tile):
    def emu(batch):
        return cnn_oracle.vgg_like_forward_bf16emu(batch.astype(np.float32), g.weights, 4,
                                                   kind='split')

    def f32(batch):
        return cnn_oracle.vgg_like_forward(batch.astype(np.float32), g.weights, 4)
    return (infer_oracle.infer_lattice(img, (tile,) * 3, (7,) * 3, emu),
            infer_oracle.infer_lattice(img, (tile,) * 3, (7,) * 3, f32))


@pytest.mark.parametrize('shape,tile', [
    ((50, 47, 41), 30), ((46, 46, 46), 30), ((31, 30, 64), 30),
    ((75, 33, 90), 30), ((104, 120, 110), 102), ((40, 135, 52), 46)])
def test_split_matches_emulation_and_fp32(ctx, shape, tile):
    g = _net(21, tile)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(9, shape)
    img = (u8.astype(np.float32) - np.float32(128)) / np.float32(33)
    got = prog.infer_volume(u8, (tile,) * 3, (7,) * 3, mean=128.0, std=33.0,
                            precision=_capi.PREC_F16S)
    assert ctx.last_path() == 'vgg_split_f16'
    emu, f32 = _refs(g, img, tile)
    assert got.shape == shape and got.dtype == np.float32
    assert not got[:7].any() and not got[-7:].any()
    assert not got[:, :7].any() and not got[:, :, -7:].any()
    d_emu, d_f32 = np.abs(got - emu), np.abs(got - f32)
    print('split vs emulation %.2e, vs fp32 %.2e' % (d_emu.max(), d_f32.max()))
    assert d_emu.max() < TOL, 'vs split emulation: max %g' % d_emu.max()
    assert d_f32.max() < TOL, 'vs fp32: max %g' % d_f32.max()


def test_split_float_input_and_other_normalisation(ctx):
    """f32 volumes (already normalised, as the reference's infer takes them) and a
    non-integer mean: the input split happens per voxel instead of through the table"""
    g = _net(22, 46)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(10, (60, 77, 94))
    img = (u8.astype(np.float32) - np.float32(127.3)) / np.float32(31.7)
    a = prog.infer_volume(img, (46,) * 3, (7,) * 3, precision=_capi.PREC_F16S)
    b = prog.infer_volume(u8, (46,) * 3, (7,) * 3, mean=127.3, std=31.7,
                          precision=_capi.PREC_F16S)
    _, f32 = _refs(g, img, 46)
    assert np.abs(a - f32).max() < TOL and np.abs(b - f32).max() < TOL
    assert np.abs(a - b).max() < 1e-6


@pytest.mark.parametrize('mean,std', [(127.37, 31.9), (300.5, 40.0), (-20.25, 17.0), (100.0, -25.0)])
def test_split_u8_operand_is_centred_on_an_integer(ctx, mean, std):
    """uint8 volumes run conv3 1->48 on the exact operand u - round(mean) with 1 / std and the
    constant folded into weights and shift (vgg_split.hip, K1); the zero padding past the
    volume's end is the one non-integer value.  Ragged volume (padding on every axis), means
    inside and outside [0, 255], and the per-(mean, std) weight set rebuilt between calls."""
    g = _net(24, 30)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(12, (50, 47, 41))
    for m, s in ((128.0, 33.0), (mean, std)):
        img = (u8.astype(np.float32) - np.float32(m)) / np.float32(s)
        got = prog.infer_volume(u8, (30,) * 3, (7,) * 3, mean=m, std=s, precision=_capi.PREC_F16S)
        _, f32 = _refs(g, img, 30)
        assert f32[7:-7, 7:-7, 7:-7].std() > 1e-4
        assert np.abs(got - f32).max() < TOL, (m, s, np.abs(got - f32).max())


@pytest.mark.parametrize('mean,std', [(128.0, 33.0), (126.63, 29.7)])
def test_split_slabs_chunks_and_tilings_are_bit_identical(ctx, monkeypatch, mean, std):
    g = _net(23, 46)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(11, (131, 70, 121))
    kw = dict(mean=mean, std=std, precision=_capi.PREC_F16S)
    whole = prog.infer_volume(u8, (46,) * 3, (7,) * 3, **kw)
    # Z slabs of the tile lattice (multi-GPU sharding), each written into its rows
    out = np.zeros_like(whole)
    nz = len(range(7, 131 - 7, 32))
    for lo, hi in multi_gpu.slab_partition(nz, 3):
        prog.infer_volume(u8, (46,) * 3, (7,) * 3, z_range=(lo, hi), dst=out, **kw)
    assert np.array_equal(out, whole)
    # another tile size: the coarse grid is anchored at the volume origin
    g2 = _net(23, 30)
    prog2 = _capi.Program(ctx, g2, (4, 4, 4))
    assert np.array_equal(prog2.infer_volume(u8, (30,) * 3, (7,) * 3, **kw), whole)
    # several scratch chunks
    monkeypatch.setenv('FPL_VGG_SCRATCH_MB', '8')
    assert np.array_equal(prog.infer_volume(u8, (46,) * 3, (7,) * 3, **kw), whole)


def test_split_is_refused_for_graphs_without_split_kernels(ctx):
    """split kernels exist for vgg_like, vgg_like2 (tests/test_gpu_vgg2_fused.py), the U-Net family
    (tests/test_gpu_unet_split.py, test_gpu_unet_family.py) and, op by op, the other factories
    (tests/test_gpu_graph_split.py); a 3x3x3 convolution with 96 outputs has none: 'f16s' says so, 'auto' runs it on
    the fp32 executor"""
    from flypylib_amd.program import LayerGraph
    g = LayerGraph(22)
    x = g.conv_bn_relu(g.input(), 32, 3)
    x = g.conv_bn_relu(x, 96, 3)
    g.finish(g.conv(x, 1, 1, use_bias=True, activation='sigmoid'))
    synth.synthetic_weights(g, 3)
    prog = _capi.Program(ctx, g, (1, 1, 1))
    u8 = synth.em_volume_u8(1, (60, 41, 48))
    with pytest.raises(_capi.FplHipError, match='split-half kernels'):
        prog.infer_volume(u8, (22,) * 3, (2,) * 3, mean=128.0, std=33.0,
                          precision=_capi.PREC_F16S)
    prog.infer_volume(u8, (22,) * 3, (2,) * 3, mean=128.0, std=33.0, precision=_capi.PREC_AUTO)
    assert ctx.last_path() == 'mfma_f32'
