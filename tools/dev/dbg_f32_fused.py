"""A/B of the fp32 MFMA executor: conv3 -> conv1 (-> pool) fused into one kernel against
one kernel per op (FPL_F32_UNFUSED=1), vgg_like, device-resident volume."""
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from flypylib_amd import _capi, fplmodels, synth, runtime
ctx = runtime.get_context(0)
g = fplmodels.vgg_like(102)[0]; synth.synthetic_weights(g, 1234)
prog = _capi.Program(ctx, g, (4, 4, 4))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
src = torch.empty((n,)*3, dtype=torch.uint8, device="cuda"); dst = torch.empty((n,)*3, dtype=torch.float32, device="cuda")
ctx.synth_volume_u8(1, (n,)*3, out=src)
kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_F32, dst=dst, dims=(n,)*3)
res = {}
for mode in ("unfused", "fused"):
    if mode == "unfused": os.environ["FPL_F32_UNFUSED"] = "1"
    else: os.environ.pop("FPL_F32_UNFUSED", None)
    prog.infer_volume(src, (102,)*3, (7,)*3, **kw); ctx.synchronize()
    ctx.timing(True); ctx.timing_reset(); t0 = time.perf_counter()
    prog.infer_volume(src, (102,)*3, (7,)*3, **kw); ctx.synchronize(); dt = time.perf_counter() - t0
    print(mode, round(dt*1e3, 1), "ms", {k: round(v["ms"], 1) for k, v in ctx.timing_get().items()})
    ctx.timing(False)
    res[mode] = dst.clone(); torch.cuda.synchronize()
print("max |fused - unfused|", float((res["fused"] - res["unfused"]).abs().max()))
