import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from flypylib_amd import _capi, train
ctx = _capi.Context(0)
rng = np.random.default_rng(0)
data = rng.standard_normal((32, 64, 64, 64)).astype(np.float32)
labels = (rng.random((32, 12, 12, 12)) > 0.9).astype(np.uint8)
st = train._DeviceStager(0)
for k in range(3):
    t0 = time.perf_counter()
    for i in range(20):
        x, y = st(data, labels)
    print('stager: %.2f ms per batch' % ((time.perf_counter() - t0) / 20 * 1e3))
# pinned variant
pin = torch.empty(data.shape, dtype=torch.float32).pin_memory()
dev = torch.empty(data.shape, dtype=torch.float32, device='cuda')
s = torch.cuda.Stream()
for k in range(3):
    t0 = time.perf_counter()
    for i in range(20):
        pin.numpy()[...] = data
        with torch.cuda.stream(s):
            dev.copy_(pin, non_blocking=True)
        s.synchronize()
    print('pinned: %.2f ms per batch' % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
for i in range(20):
    pin.numpy()[...] = data
print('host memcpy only: %.2f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
