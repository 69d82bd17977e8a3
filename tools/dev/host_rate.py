"""dev probe: PCIe-inclusive rate of FplNetwork.infer-style calls (host u8 in, host f32 out)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flypylib_amd import _capi, fplmodels, runtime, synth

ctx = runtime.get_context(0)
g = fplmodels.vgg_like(102)[0]
synth.synthetic_weights(g, 1)
prog = _capi.Program(ctx, g, (4, 4, 4))
for n in (512, 1024):
    u8 = ctx.synth_volume_u8(1, (n, n, n))
    out = np.empty((n, n, n), np.float32)
    for prec, name in ((_capi.PREC_BF16, 'bf16'),):
        prog.infer_volume(u8, (102,) * 3, (7,) * 3, mean=128.0, std=33.0, precision=prec, dst=out)
        t0 = time.perf_counter()
        prog.infer_volume(u8, (102,) * 3, (7,) * 3, mean=128.0, std=33.0, precision=prec, dst=out)
        dt = time.perf_counter() - t0
        print('%d^3 %s host->host: %.3f s = %.0f Mvox/s (%.1f GB/s over PCIe)'
              % (n, name, dt, (n - 14) ** 3 / dt / 1e6, 5 * n ** 3 / dt / 1e9))
