"""dev: one training configuration's step time and kernel split
usage: train_step.py <factory> <batch> <patch> [loss]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flypylib_amd import _capi, fplmodels, runtime, synth
from flypylib_amd.program import LayerGraph

ctx = runtime.get_context(0)
name, B, P = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
loss = sys.argv[4] if len(sys.argv) > 4 else 'binary_crossentropy'
g = getattr(fplmodels, name)(P)[0]
synth.synthetic_weights(g, 3)
tr = _capi.Trainer(ctx, g, loss=loss)
out = g.output_shape if hasattr(g, 'output_shape') else None
rng = np.random.default_rng(0)
data = rng.standard_normal((B, P, P, P, 1)).astype(np.float32)
o = int(sys.argv[5]) if len(sys.argv) > 5 else 1
labels = rng.integers(0, 2, (B, o, o, o, 1)).astype(np.uint8)
tr.step(data, labels, 0); tr.apply(1.0)
ctx.synchronize()
ctx.timing(True); ctx.timing_reset()
t0 = time.perf_counter()
steps = 5
for s in range(steps):
    tr.step(data, labels, s + 1); tr.apply(1.0)
ctx.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps(dict(model=name, batch=B, patch=P, ms_per_step=round(dt * 1e3, 3), steps_per_s=round(1 / dt, 1),
                      kernels={k: round(v['ms'] / steps, 3) for k, v in ctx.timing_get().items()})))
