"""dev: inference rate of any model factory on a synthetic uint8 cube
usage: infer_rate.py <factory> <size> [precision]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flypylib_amd import _capi, fplmodels, fplutils, runtime, synth

ctx = runtime.get_context(0)
name, n = sys.argv[1], int(sys.argv[2])
prec = {'f32': _capi.PREC_F32, 'f16': _capi.PREC_F16, 'bf16': _capi.PREC_BF16}[sys.argv[3] if len(sys.argv) > 3 else 'f32']
net, rf, infer_sz, _ = getattr(fplmodels, name)()
tile = fplutils.to3d(infer_sz)[0]
off = fplutils.to3d(rf[1])[0]
stride = fplutils.to3d(rf[2])
g = getattr(fplmodels, name)(tile)[0]
synth.synthetic_weights(g, 5)
prog = _capi.Program(ctx, g, tuple(stride))
src = torch.empty((n, n, n), dtype=torch.uint8, device='cuda')
dst = torch.empty((n, n, n), dtype=torch.float32, device='cuda')
ctx.synth_volume_u8(4, (n, n, n), out=src)
kw = dict(mean=128.0, std=33.0, precision=prec, dst=dst, dims=(n, n, n))
prog.infer_volume(src, (tile,) * 3, (off,) * 3, **kw)
ctx.synchronize()
ctx.timing(True); ctx.timing_reset()
t0 = time.perf_counter()
prog.infer_volume(src, (tile,) * 3, (off,) * 3, **kw)
ctx.synchronize()
dt = time.perf_counter() - t0
print(json.dumps(dict(model=name, size=n, mvox_s=round((n - 2 * off) ** 3 / dt / 1e6, 1), seconds=round(dt, 4),
                      kernels={k: round(v['ms'], 2) for k, v in ctx.timing_get().items()})))
