"""GPU parity for the bf16 MFMA unet_like2 path (csrc/conv_mfma.hip) on the
reference tile lattice (the U-Net is only 4-voxel shift-equivariant)."""
import numpy as np
import pytest

from flypylib_amd import _capi, fplmodels, synth
from oracle import cnn_oracle, infer_oracle

pytestmark = pytest.mark.gpu
EMU_TOL = {'bf16': 1e-2, 'f16': 2e-3}
F32_TOL = {'bf16': 1.5e-2, 'f16': 1e-3}       # f16 meets the north star's 1e-3 gate
PREC = {'bf16': _capi.PREC_BF16, 'f16': _capi.PREC_F16}


@pytest.mark.parametrize('kind', ['bf16', 'f16'])
@pytest.mark.parametrize('shape,tile', [((45, 38, 31), 28), ((60, 52, 70), 36),
                                        ((110, 100, 104), 100)])
def test_unet_bf16_matches_emulation_and_fp32(ctx, shape, tile, kind):
    g = fplmodels.unet_like2(tile)[0]
    synth.synthetic_weights(g, 41)
    prog = _capi.Program(ctx, g, (1, 1, 1))
    u8 = synth.em_volume_u8(13, shape)
    img = (u8.astype(np.float32) - np.float32(128)) / np.float32(33)
    got = prog.infer_volume(u8, (tile,) * 3, (9,) * 3, mean=128.0, std=33.0,
                            precision=PREC[kind])
    f32gpu = prog.infer_volume(u8, (tile,) * 3, (9,) * 3, mean=128.0, std=33.0,
                               precision=_capi.PREC_F32)
    if tile <= 36:
        emu = infer_oracle.infer_lattice(
            img, (tile,) * 3, (9,) * 3,
            lambda b: cnn_oracle.unet_like2_forward_bf16emu(b.astype(np.float32), g.weights,
                                                            kind=kind))
        d = np.abs(got - emu)
        assert d.max() < EMU_TOL[kind], 'vs %s emulation: max %g' % (kind, d.max())
        assert d.mean() < (1e-4 if kind == 'bf16' else 2e-5)
        assert np.mean(d > 1e-3) < 1e-3
    # the fp32 per-op path is itself oracle-checked (tests/test_gpu_cnn.py)
    d = np.abs(got - f32gpu)
    assert d.max() < F32_TOL[kind], '%s vs fp32: max %g' % (kind, d.max())
    assert not got[:9].any() and not got[:, :, -9:].any()
    assert f32gpu[9:-9, 9:-9, 9:-9].std() > 1e-3
