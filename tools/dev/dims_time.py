"""per-kernel times of the split vgg_like path on a (Z, Y, X) volume:  python tools/dev/dims_time.py Z Y X [...]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from flypylib_amd import _capi, fplmodels, synth, runtime

ctx = runtime.get_context(0)
g = fplmodels.vgg_like(102)[0]
synth.synthetic_weights(g, 1234)
prog = _capi.Program(ctx, g, (4, 4, 4))
args = [int(v) for v in sys.argv[1:]]
for i in range(0, len(args), 3):
    dims = tuple(args[i:i + 3])
    src = ctx.malloc(dims, np.uint8)
    ctx.synth_volume_u8(1, dims, out=src)
    dst = ctx.malloc(dims, np.float32)
    kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_F16S, dims=dims, dst=dst)
    prog.infer_volume(src, (102,) * 3, (7,) * 3, **kw)
    ctx.synchronize()
    ctx.timing(True); ctx.timing_reset()
    for _ in range(5):
        prog.infer_volume(src, (102,) * 3, (7,) * 3, **kw)
    ctx.synchronize()
    k = {n: round(v['ms'] / 5, 3) for n, v in ctx.timing_get().items()}
    ctx.timing(False)
    vox = np.prod([d - 14 for d in dims])
    print(dims, k, 'Gvox/s of kernel sum', round(vox / sum(k.values()) / 1e6, 2), flush=True)
    del src, dst
