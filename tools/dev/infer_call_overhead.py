"""wall time of fpl_infer_volume calls on one 582^3 substack against the sum of its kernels"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from flypylib_amd import _capi, fplmodels, synth, runtime
ctx = runtime.get_context(0)
g = fplmodels.vgg_like(102)[0]
synth.synthetic_weights(g, 1234)
prog = _capi.Program(ctx, g, (4, 4, 4))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 582
dims = (n,) * 3
src = ctx.malloc(dims, np.uint8); ctx.synth_volume_u8(1, dims, out=src)
dst = ctx.malloc(dims, np.float32)
for label, means in (('same mean', [128.0] * 12), ('new mean per call', [120.0 + 0.37 * i for i in range(12)])):
    for prec, pname in ((_capi.PREC_AUTO, 'auto'), (_capi.PREC_F16S, 'f16s'), (_capi.PREC_F16, 'f16')):
        prog.infer_volume(src, (102,) * 3, (7,) * 3, mean=means[0], std=33.0, precision=prec, dims=dims, dst=dst)
        ctx.synchronize()
        ctx.timing(True); ctx.timing_reset()
        t0 = time.perf_counter()
        for m in means[1:]:
            prog.infer_volume(src, (102,) * 3, (7,) * 3, mean=m, std=33.0, precision=prec, dims=dims, dst=dst)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / (len(means) - 1)
        k = sum(v['ms'] for v in ctx.timing_get().values()) / (len(means) - 1)
        ctx.timing(False)
        print('%-18s %-5s wall %.2f ms per call, kernels %.2f ms' % (label, pname, dt * 1e3, k), flush=True)
