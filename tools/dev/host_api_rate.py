"""Where the host -> host time of the public API goes (FplNetwork.infer: host uint8 in, host float32 out):
fresh pageable output (np.empty: first touch inside the copy), a reused pageable output, pinned output,
pinned input + output.  520^3 by default."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from flypylib_amd import _capi, fplmodels, runtime, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 520
ctx = runtime.get_context(0)
g, rf, _, _ = fplmodels.vgg_like(102)
synth.synthetic_weights(g, 7)
prog = _capi.Program(ctx, g, (4, 4, 4))
u8 = synth.em_volume_u8(5, (n, n, n))
kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_AUTO)


def run(label, src, dst_fn, reps=5):
    ts = []
    for _ in range(reps):
        dst = dst_fn()
        t0 = time.perf_counter()
        out = prog.infer_volume(src, (102,) * 3, (7,) * 3, dst=dst, **kw)
        ts.append(time.perf_counter() - t0)
    print('%-42s %s ms  (median %.1f)' % (label, ' '.join('%.1f' % (t * 1e3) for t in ts), sorted(ts)[reps // 2] * 1e3),
          flush=True)
    return out


ref = run('fresh pageable output (np.empty)', u8, lambda: None)
reused = np.empty((n, n, n), np.float32)
reused[:] = 0
out = run('reused pageable output', u8, lambda: reused)
assert np.array_equal(out, ref)
pin_out = torch.empty((n, n, n), dtype=torch.float32, pin_memory=True).numpy()
out = run('pinned output, pageable input', u8, lambda: pin_out)
assert np.array_equal(out, ref)
pin_in = torch.empty((n, n, n), dtype=torch.uint8, pin_memory=True).numpy()
pin_in[:] = u8
out = run('pinned input and output', pin_in, lambda: pin_out)
assert np.array_equal(out, ref)
# device-resident, for scale
src_d = torch.from_numpy(u8).cuda()
dst_d = torch.empty((n, n, n), dtype=torch.float32, device='cuda')
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prog.infer_volume(src_d, (102,) * 3, (7,) * 3, dst=dst_d, dims=(n, n, n), **kw)
    ctx.synchronize()
    ts.append(time.perf_counter() - t0)
print('device -> device                            %s ms' % ' '.join('%.1f' % (t * 1e3) for t in ts))
t0 = time.perf_counter(); a = np.empty((n, n, n), np.float32); a[:] = 0; print('first touch of the output array alone: %.1f ms' % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); b = torch.empty((n, n, n), dtype=torch.float32, pin_memory=True); print('pinned allocation of the output: %.1f ms' % ((time.perf_counter() - t0) * 1e3))

# the public object: FplNetwork.infer on host arrays, plain and through n slab threads on this one GPU
from flypylib_amd import FplNetwork
net = FplNetwork(fplmodels.vgg_like)
synth.synthetic_weights(net.train_single, 7)
net._set_infer()
for slabs in (1, 2, 3, 4):
    if slabs > 1:
        net.make_infer_parallel(slabs, devices=[0] * slabs)
    ts = []
    res = None
    for _ in range(7):
        t0 = time.perf_counter()
        res = net.infer(u8, normalize=(128., 33.))
        ts.append(time.perf_counter() - t0)
    print('FplNetwork.infer, %d slab thread(s): %s ms' % (slabs, ' '.join('%.1f' % (t * 1e3) for t in ts)), flush=True)
    assert np.array_equal(res, ref)
