"""The default ('auto' = split halves) executors against the fp32 MFMA executor on volumes of
awkward shapes: extents that leave partial blocks / bricks / tile rows in every kernel's grid
(the persistent kernels' block walks, the stem's edge and interior passes, the U-Net's parity
weight streams and transposed edge strip all depend on them).  Round 4 changed all of those."""
import numpy as np
import pytest

from flypylib_amd import _capi, fplmodels, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,tile,off,shapes', [
    ('vgg_like', 102, 7, [(102, 102, 102), (103, 201, 142), (333, 102, 257), (148, 137, 260)]),
    ('unet_like2', 100, 9, [(100, 100, 100), (101, 197, 140), (241, 121, 232)]),
])
def test_split_path_on_awkward_shapes(ctx, name, tile, off, shapes):
    g = getattr(fplmodels, name)(tile)[0]
    synth.synthetic_weights(g, 77)
    prog = _capi.Program(ctx, g, (4, 4, 4) if name == 'vgg_like' else (1, 1, 1))
    rng = np.random.default_rng(5)
    for shp in shapes:
        u8 = synth.em_volume_u8(int(rng.integers(1, 1000)), shp)
        kw = dict(mean=float(rng.uniform(110, 140)), std=float(rng.uniform(25, 40)))
        a = prog.infer_volume(u8, (tile,) * 3, (off,) * 3, precision=_capi.PREC_AUTO, **kw)
        assert 'split' in ctx.last_path(), ctx.last_path()
        b = prog.infer_volume(u8, (tile,) * 3, (off,) * 3, precision=_capi.PREC_F32, **kw)
        d = float(np.abs(a - b).max())
        assert d < 1e-5, (name, shp, d)
        assert b[off:-off, off:-off, off:-off].std() > 1e-3
    prog.close()
