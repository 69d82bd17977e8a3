"""split path vs the fp32 executor on volumes of awkward shapes (vgg_like and unet_like2)"""
import sys
import numpy as np
sys.path.insert(0, '.')
from flypylib_amd import _capi, fplmodels, synth, runtime
ctx = runtime.get_context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for name, fac, tile, off in (('vgg_like', fplmodels.vgg_like, 102, 7), ('unet_like2', fplmodels.unet_like2, 100, 9)):
    g = fac(tile)[0]
    synth.synthetic_weights(g, 77)
    prog = _capi.Program(ctx, g, (4, 4, 4) if name == 'vgg_like' else (1, 1, 1))
    shapes = [tuple(int(v) for v in rng.integers(tile, 300, 3)) for _ in range(6)]
    shapes += [(tile, tile, tile), (tile + 1, 2 * tile - 3, tile + 40), (333, tile, 257)]
    for shp in shapes:
        u8 = synth.em_volume_u8(int(rng.integers(1, 1000)), shp)
        kw = dict(mean=float(rng.uniform(110, 140)), std=float(rng.uniform(25, 40)))
        a = prog.infer_volume(u8, (tile,) * 3, (off,) * 3, precision=_capi.PREC_F16S, **kw)
        path = ctx.last_path()
        b = prog.infer_volume(u8, (tile,) * 3, (off,) * 3, precision=_capi.PREC_F32, **kw)
        d = float(np.abs(a - b).max())
        worst = max(worst, d)
        print('%-10s %-16s %-16s max |f16s - f32| %.2e %s' % (name, shp, path, d, 'FAIL' if not d < 1e-5 else ''), flush=True)
    prog.close()
print('worst %.2e' % worst)
assert worst < 1e-5
