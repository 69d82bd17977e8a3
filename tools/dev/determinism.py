"""run-to-run identity of the split executors: N inferences of the same volume, compared bit for bit
(a race between an inline-asm result and a late MFMA write-back showed up this way in round 4)"""
import sys
import numpy as np
sys.path.insert(0, '.')
from flypylib_amd import _capi, fplmodels, synth, runtime
ctx = runtime.get_context(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
from flypylib_amd import fplutils
CASES = [('vgg_like', fplmodels.vgg_like, 102, 7, 768), ('unet_like2', fplmodels.unet_like2, 100, 9, 346),
         # the graph executor (csrc/gx_exec.h)
         ('baseline_model', fplmodels.baseline_model, 102, 7, 454), ('resnet_like', fplmodels.resnet_like, 102, 7, 454),
         ('unet_like4b', fplmodels.unet_like4b, 100, 17, 298), ('unet_like_vol', fplmodels.unet_like_vol, 102, 6, 282)]
for name, fac, tile, off, n in CASES:
    g = fac(tile)[0]
    synth.synthetic_weights(g, 5)
    prog = _capi.Program(ctx, g, fplutils.to3d(fac()[1][2]))
    dims = (n,) * 3
    src = ctx.malloc(dims, np.uint8); ctx.synth_volume_u8(3, dims, out=src)
    dst = ctx.malloc(dims, np.float32)
    kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_AUTO, dims=dims, dst=dst)
    prog.infer_volume(src, (tile,) * 3, (off,) * 3, **kw)
    ref = dst.to_host()
    bad = 0
    for i in range(reps):
        prog.infer_volume(src, (tile,) * 3, (off,) * 3, **kw)
        out = dst.to_host()
        if not np.array_equal(out, ref):
            bad += 1
            print('  run %d differs in %d voxels, max %.2e' % (i, int((out != ref).sum()), float(np.abs(out - ref).max())))
    print('%-10s %s^3 on %s: %d of %d runs differ' % (name, n, ctx.last_path(), bad, reps), flush=True)
    assert bad == 0
    src.free(); dst.free(); prog.close()
