"""per-kernel times of one vgg_like training step (configs[3]: 32 patches of 64^3)"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from flypylib_amd import _capi, fplmodels, synth
ctx = _capi.Context(0)
g = fplmodels.vgg_like()[0]
synth.synthetic_weights(g, 8)
tr = _capi.Trainer(ctx, g)
rng = np.random.default_rng(0)
data = rng.standard_normal((32, 64, 64, 64)).astype(np.float32)
labels = (rng.random((32, 12, 12, 12)) > 0.9).astype(np.uint8)
ctx.timing(True)
tr.step(data, labels, 0); tr.apply(1.0); ctx.synchronize(); ctx.timing_reset()
reps = 5
t0 = time.perf_counter()
for i in range(reps):
    tr.step(data, labels, i + 1); tr.apply(1.0)
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
kern = ctx.timing_get(); ctx.timing(False)
tot = 0
for k, v in sorted(kern.items(), key=lambda kv: -kv[1]['ms']):
    print('%-40s %8.3f ms  x%d' % (k, v['ms'] / reps, v.get('n', 0) // reps if isinstance(v.get('n', 0), int) else 0))
    tot += v['ms'] / reps
print('sum %.3f ms, wall %.3f ms, %.1f steps/s' % (tot, dt * 1e3, 1 / dt))
ctx.timing(False)
t0 = time.perf_counter()
for i in range(reps):
    tr.step(data, labels, i + 1); tr.apply(1.0)
ctx.synchronize()
print('untimed wall %.3f ms' % ((time.perf_counter() - t0) / reps * 1e3))
from flypylib_amd import train
def forever():
    while True:
        yield data, labels
b = train._Prefetch(forever(), stage=train._DeviceStager(ctx.device))
tr.step(*next(b), 0); tr.apply(1.0)
t0 = time.perf_counter()
for i in range(20):
    tr.step(*next(b), i + 1); tr.apply(1.0)
ctx.synchronize()
dt = (time.perf_counter() - t0) / 20
print('staged wall %.3f ms, %.1f steps/s' % (dt * 1e3, 1 / dt))
b.close()
