"""dev experiment: does running two Z halves of one volume on two HIP streams (two
contexts of the same GPU) overlap the stem of one half with the mid kernel of the other?"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flypylib_amd import _capi, fplmodels, synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tile, off = 102, 7
torch.cuda.set_device(0)
ctxs = [_capi.Context(0) for _ in range(lanes)]
graph = fplmodels.vgg_like(tile)[0]
synth.synthetic_weights(graph, 1234)
progs = [_capi.Program(c, graph, (4, 4, 4)) for c in ctxs]
dims = (size, size, size)
src = torch.empty(dims, dtype=torch.uint8, device='cuda')
dst = torch.zeros(dims, dtype=torch.float32, device='cuda')
ref = torch.zeros(dims, dtype=torch.float32, device='cuda')
ctxs[0].synth_volume_u8(20250101, dims, (0, 0, 0), out=src)
ctxs[0].synchronize()


def whole(d):
    progs[0].infer_volume(src, (tile,) * 3, (off,) * 3, mean=128.0, std=33.0,
                          precision=_capi.PREC_F16, dst=d, dims=dims)
    ctxs[0].synchronize()


# z ranges are tile rows of the reference lattice (pitch 88): split them evenly
n_rows = -(-(size - 2 * off) // (tile - 2 * off))
cuts = [n_rows * i // lanes for i in range(lanes)] + [-1]


def part(i):
    progs[i].infer_volume(src, (tile,) * 3, (off,) * 3, mean=128.0, std=33.0,
                          precision=_capi.PREC_F16, dst=dst, dims=dims,
                          z_range=(cuts[i], cuts[i + 1]))
    ctxs[i].synchronize()


def split():
    th = [threading.Thread(target=part, args=(i,)) for i in range(lanes)]
    for t in th:
        t.start()
    for t in th:
        t.join()


for f, name, arg in ((whole, 'one stream', (ref,)), (split, '%d streams' % lanes, ())):
    f(*arg)
    t0 = time.perf_counter()
    for _ in range(5):
        f(*arg)
    dt = (time.perf_counter() - t0) / 5
    print('%s: %.2f ms' % (name, dt * 1e3))
torch.cuda.synchronize()
print('identical:', bool(torch.equal(ref, dst)))
