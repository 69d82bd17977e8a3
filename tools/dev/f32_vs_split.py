"""1024^3 check of the fp32 MFMA path (super-tiles) against the split-half path: max |diff|
over the whole volume, and the per-kernel times of the fp32 pass."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from flypylib_amd import _capi, fplmodels, synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = _capi.Context(0)
g = fplmodels.vgg_like(102)[0]
synth.synthetic_weights(g, 7)
prog = _capi.Program(ctx, g, (4, 4, 4))
dims = (size,) * 3
src = torch.empty(dims, dtype=torch.uint8, device='cuda')
ctx.synth_volume_u8(20250101, dims, (0, 0, 0), out=src)
out = {}
for name, prec in (('f16s', _capi.PREC_F16S), ('f32', _capi.PREC_F32)):
    dst = torch.empty(dims, dtype=torch.float32, device='cuda')
    kw = dict(mean=128.0, std=33.0, precision=prec, dst=dst, dims=dims)
    prog.infer_volume(src, (102,) * 3, (7,) * 3, **kw)
    ctx.synchronize(); ctx.timing(True); ctx.timing_reset()
    t0 = time.perf_counter()
    prog.infer_volume(src, (102,) * 3, (7,) * 3, **kw)
    ctx.synchronize()
    print(name, ctx.last_path(), round((time.perf_counter() - t0) * 1e3, 2), 'ms',
          {k: round(v['ms'], 2) for k, v in ctx.timing_get().items()}, flush=True)
    ctx.timing(False)
    out[name] = dst
d = (out['f32'] - out['f16s']).abs().max().item()
print('max |f32 - f16s| =', d, ' std', out['f32'][7:-7, 7:-7, 7:-7].std().item())
assert d < 1e-4
