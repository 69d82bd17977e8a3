"""GPU parity for unet_like2 on split IEEE-half operands (csrc/conv_mfma.hip built with
-DFPL_SPLIT, precision 'f16s'): every tensor as [hi 16 | lo 16] halves per 16 channels,
(a_hi + a_lo)(w_hi + w_lo) in two MFMAs per 16 channels and tap.  Held to fp32-grade
agreement (1e-5) with the emulation oracle of the same rounding points and with the fp32
MFMA path, on the reference tile lattice."""
import numpy as np
import pytest

from flypylib_amd import _capi, fplmodels, multi_gpu, synth
from oracle import cnn_oracle, infer_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize('shape,tile', [((45, 38, 31), 28), ((60, 52, 70), 36),
                                        ((110, 100, 104), 100)])
def test_unet_split_matches_emulation_and_fp32(ctx, shape, tile):
    g = fplmodels.unet_like2(tile)[0]
    synth.synthetic_weights(g, 41)
    prog = _capi.Program(ctx, g, (1, 1, 1))
    u8 = synth.em_volume_u8(13, shape)
    img = (u8.astype(np.float32) - np.float32(128)) / np.float32(33)
    kw = dict(mean=128.0, std=33.0)
    got = prog.infer_volume(u8, (tile,) * 3, (9,) * 3, precision=_capi.PREC_F16S, **kw)
    assert ctx.last_path() == 'unet_split_f16'
    f32gpu = prog.infer_volume(u8, (tile,) * 3, (9,) * 3, precision=_capi.PREC_F32, **kw)
    if tile <= 36:
        emu = infer_oracle.infer_lattice(
            img, (tile,) * 3, (9,) * 3,
            lambda b: cnn_oracle.unet_like2_forward_bf16emu(b.astype(np.float32), g.weights,
                                                            kind='split'))
        d = np.abs(got - emu)
        print('split vs emulation %.2e' % d.max())
        assert d.max() < TOL, 'vs split emulation: max %g' % d.max()
    d = np.abs(got - f32gpu)
    print('split vs fp32 %.2e' % d.max())
    assert d.max() < TOL, 'split vs fp32: max %g' % d.max()
    assert not got[:9].any() and not got[:, :, -9:].any()
    assert f32gpu[9:-9, 9:-9, 9:-9].std() > 1e-3
    # the default precision of the API takes this path
    auto = prog.infer_volume(u8, (tile,) * 3, (9,) * 3, precision=_capi.PREC_AUTO, **kw)
    assert ctx.last_path() == 'unet_split_f16' and np.array_equal(auto, got)


def test_unet_parity_form_agrees_with_the_plain_taps(ctx, monkeypatch):
    """conv3 192->64 and the head's conv3 read an UpSampling3D(2) for 128 of 192 / 64 of 96 input
    channels: by default their three z taps on those channels collapse to two with weights
    pre-summed per output-plane parity (fp32 sums, split afterwards; 18 taps instead of 27);
    FPL_UNET_NOPARITY=1 runs the 27 taps.  Same probabilities up to the rounding of the
    pre-summed weights' lo halves, both within the gate of the fp32 path.  The 100^3 tile has the
    82-wide layer: main columns in the parity form, edge strip in the plain one."""
    for shape, tile in (((60, 52, 70), 36), ((110, 100, 104), 100)):
        g = fplmodels.unet_like2(tile)[0]
        synth.synthetic_weights(g, 43)
        prog = _capi.Program(ctx, g, (1, 1, 1))
        u8 = synth.em_volume_u8(15, shape)
        kw = dict(mean=128.0, std=33.0)
        par = prog.infer_volume(u8, (tile,) * 3, (9,) * 3, precision=_capi.PREC_F16S, **kw)
        f32gpu = prog.infer_volume(u8, (tile,) * 3, (9,) * 3, precision=_capi.PREC_F32, **kw)
        monkeypatch.setenv('FPL_UNET_NOPARITY', '1')
        plain = prog.infer_volume(u8, (tile,) * 3, (9,) * 3, precision=_capi.PREC_F16S, **kw)
        monkeypatch.delenv('FPL_UNET_NOPARITY')
        assert not np.array_equal(par, plain), 'the switch changed nothing: is the parity form running?'
        print('parity vs plain %.2e, vs fp32 %.2e / %.2e' % (np.abs(par - plain).max(), np.abs(par - f32gpu).max(),
                                                            np.abs(plain - f32gpu).max()))
        assert np.abs(par - plain).max() < 3e-6
        assert np.abs(par - f32gpu).max() < TOL and np.abs(plain - f32gpu).max() < TOL
        prog.close()


def test_unet_split_slabs_equal_whole(ctx):
    tile = 36
    g = fplmodels.unet_like2(tile)[0]
    synth.synthetic_weights(g, 42)
    prog = _capi.Program(ctx, g, (1, 1, 1))
    u8 = synth.em_volume_u8(14, (98, 60, 47))
    kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_F16S)
    whole = prog.infer_volume(u8, (tile,) * 3, (9,) * 3, **kw)
    out = np.zeros_like(whole)
    nz = multi_gpu.n_tile_rows(98, tile, 9)
    for lo, hi in multi_gpu.slab_partition(nz, 3):
        prog.infer_volume(u8, (tile,) * 3, (9,) * 3, z_range=(lo, hi), dst=out, **kw)
    assert np.array_equal(out, whole)


def test_graphs_outside_the_skeleton_go_to_the_graph_executor(ctx):
    """unet_like4b (48-channel bottlenecks) is not the skeleton the fused kernels know: it runs op by op
    on the same kernels (csrc/gx_exec.h; parity: tests/test_gpu_graph_split.py)"""
    from flypylib_amd import fplutils
    off = fplutils.to3d(fplmodels.unet_like4b()[1][1])[0]
    tile = fplmodels.unet_like4b()[2]
    tile = fplutils.to3d(tile)[0]
    g = fplmodels.unet_like4b(tile)[0]
    synth.synthetic_weights(g, 3)
    prog = _capi.Program(ctx, g, (1, 1, 1))
    u8 = synth.em_volume_u8(1, (tile + 6, tile, tile + 9))
    a = prog.infer_volume(u8, (tile,) * 3, (off,) * 3, mean=128.0, std=33.0, precision=_capi.PREC_AUTO)
    assert ctx.last_path() == 'graph_split_f16'
    b = prog.infer_volume(u8, (tile,) * 3, (off,) * 3, mean=128.0, std=33.0, precision=_capi.PREC_F32)
    assert ctx.last_path() == 'mfma_f32' and np.abs(a - b).max() < 1e-5
