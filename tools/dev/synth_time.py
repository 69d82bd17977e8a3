"""time fpl_synth_substack_u8 on a 582^3 substack (device-resident destination)"""
import numpy as np
from flypylib_amd import _capi
ctx = _capi.Context(0)
buf = ctx.malloc((582,) * 3, np.uint8)
args = (1, (1536,) * 3, (582,) * 3, (477, 477, 477), buf)
for i in range(3):
    ctx.synth_substack_u8(*args)
ctx.timing(True)
ctx.timing_reset()
for i in range(10):
    ctx.synth_substack_u8(*args)
print({k: round(v['ms'] / v['launches'], 4) for k, v in ctx.timing_get().items()})
