"""unet_like2 on split halves: the all-LDS kernels (round 5) against the round-4 kernels
(FPL_UNET_OLDSPLIT=1) on one box: per-kernel ms and the difference of the two predictions.
    python tools/dev/unet_ab.py [size=510] [reps=3]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from flypylib_amd import _capi, fplmodels, synth, runtime
ctx = runtime.get_context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 510
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
PREC = {'f16s': _capi.PREC_F16S, 'f16': _capi.PREC_F16, 'bf16': _capi.PREC_BF16}[os.environ.get('FPL_AB_PREC', 'f16s')]
g = fplmodels.unet_like2(100)[0]
synth.synthetic_weights(g, 7)
prog = _capi.Program(ctx, g, (1, 1, 1))
dims = (n,) * 3
src = ctx.malloc(dims, np.uint8); ctx.synth_volume_u8(3, dims, out=src)
dst = ctx.malloc(dims, np.float32)
outs = {}
variants = [('lds', None, None)] + ([] if os.environ.get('FPL_AB_SKIP_OLD') else [('old', 'FPL_UNET_OLDSPLIT', '1')])
for v in sys.argv[3:]:
    variants.append(('dbg' + v, 'FPL_U3_DBG', v))
for label, env, val in variants:
    if env:
        os.environ[env] = val
    kw = dict(mean=128.0, std=33.0, precision=PREC, dims=dims, dst=dst)
    def run():
        try:
            prog.infer_volume(src, (100,) * 3, (9,) * 3, **kw)
        except _capi.FplHipError:
            if not label.startswith('dbg'):     # (timing builds compute garbage: the range guard may fire)
                raise
    run()
    ctx.synchronize()
    ctx.timing(True); ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    k = {kk: round(v['ms'] / reps, 3) for kk, v in ctx.timing_get().items()}
    ctx.timing(False)
    print('%s: %.2f ms per pass, kernels %s' % (label, dt * 1e3, k), flush=True)
    outs[label] = dst.to_host() if n <= 600 else None
    if env:
        del os.environ[env]
if outs['lds'] is not None and 'old' in outs:
    d = np.abs(outs['lds'] - outs['old'])
    print('lds vs old: max %.2e mean %.2e' % (d.max(), d.mean()))
