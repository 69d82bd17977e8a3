"""dev: cost of the radix histogram fused into the x pass (smooth with / without ranks)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flypylib_amd import fplobjdetect, runtime, synth

ctx = runtime.get_context(0)
shape = (582, 582, 582)
pred = synth.blob_prob_volume(7, shape, period=48, radius=7.0)
k = fplobjdetect.gaussian_kernel1d(5.0)
n = int(np.prod([s + 54 for s in shape]))
for ranks in ([], [int(0.97 * (n - 1))]):
    ctx.v2o_smooth(pred, shape, 27, k, ranks)
    ctx.timing(True); ctx.timing_reset()
    for _ in range(3):
        ctx.v2o_smooth(pred, shape, 27, k, ranks)
    ctx.synchronize()
    t = ctx.timing_get(); ctx.timing(False)
    print('ranks', len(ranks), {a: round(b['ms'] / 3, 3) for a, b in t.items()})
