"""GPU parity of the fused 16-bit U-Net path (csrc/conv_mfma.hip) on the siblings of
unet_like2 that share its skeleton (reference flypylib/fplmodels.py:210-407): unet_like
(1x1 second convs in stages 1 and 2: a chained stem kernel and a 16-bit pool), unet_like3
(bottom conv3 64->128, conv1 128->128, stage-2 skip cropped by 2) and unet_like4 (two 3x3x3
bottom convs, crop 4).  Their 128-channel conv3 outputs are produced as two 64-channel
launches into one tensor.  Reference: the fp32 oracle over the reference tile lattice;
f16 is held to the north star's 1e-3 gate."""
import numpy as np
import pytest

from flypylib_amd import _capi, fplmodels, fplutils, synth
from oracle import cnn_oracle, infer_oracle

pytestmark = pytest.mark.gpu
TOL = {'bf16': 1.5e-2, 'f16': 1e-3, 'f16s': 1e-5}
PREC = {'bf16': _capi.PREC_BF16, 'f16': _capi.PREC_F16, 'f16s': _capi.PREC_F16S}


@pytest.mark.parametrize('kind', ['bf16', 'f16'])
@pytest.mark.parametrize('name,tile,shape', [
    ('unet_like', 30, (50, 41, 64)), ('unet_like', 102, (110, 102, 130)),
    ('unet_like3', 44, (60, 50, 75)), ('unet_like3', 100, (110, 100, 124)),
    ('unet_like4', 52, (70, 52, 90)), ('unet_like4', 100, (100, 130, 104))])
def test_unet_siblings_fused_match_fp32_oracle(ctx, name, tile, shape, kind):
    _check_sibling(ctx, name, tile, shape, kind)


@pytest.mark.parametrize('name,tile,shape', [
    ('unet_like', 30, (50, 41, 64)), ('unet_like', 102, (110, 102, 130)),
    ('unet_like3', 44, (60, 50, 75)), ('unet_like3', 100, (110, 100, 124)),
    ('unet_like4', 52, (70, 52, 90)), ('unet_like4', 100, (100, 130, 104))])
def test_unet_siblings_split_halves_are_fp32_grade(ctx, name, tile, shape):
    """the split-half build (precision 'f16s', what 'auto' picks) on unet_like2's siblings -
    unet_like's chained stem and pool have split forms of their own: within 1e-5 of the fp32
    oracle"""
    _check_sibling(ctx, name, tile, shape, 'f16s')


def _check_sibling(ctx, name, tile, shape, kind):
    factory = getattr(fplmodels, name)
    _, rf, _, _ = factory()
    off = fplutils.to3d(rf[1])[0]
    g = factory(tile)[0]
    synth.synthetic_weights(g, 41)
    prog = _capi.Program(ctx, g, (1, 1, 1))
    u8 = synth.em_volume_u8(13, shape)
    img = (u8.astype(np.float32) - np.float32(128)) / np.float32(33)

    def f32(batch):
        return cnn_oracle.graph_forward(g, batch.astype(np.float32))
    ref = infer_oracle.infer_lattice(img, (tile,) * 3, (off,) * 3, f32)
    ctx.timing(True)
    ctx.timing_reset()
    got = prog.infer_volume(u8, (tile,) * 3, (off,) * 3, mean=128.0, std=33.0, precision=PREC[kind])
    names = set(ctx.timing_get())
    ctx.timing(False)
    assert ('unet_conv3_64_128' in names) or ('unet_stem_conv1_32_32_pool' in names), names   # fused path ran
    assert got.shape == shape and not got[:off].any() and not got[:, :, -off:].any()
    d = np.abs(got - ref)
    assert d.max() < TOL[kind], '%s %s vs fp32 oracle: max %g' % (name, kind, d.max())
    assert ref[off:-off, off:-off, off:-off].std() > 1e-4
