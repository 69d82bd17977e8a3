"""whole volume vs Z slabs on the split path: where do they differ?"""
import sys
import numpy as np
sys.path.insert(0, '.')
from flypylib_amd import _capi, fplmodels, multi_gpu, synth, runtime

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = runtime.get_context(0)
g = fplmodels.vgg_like(102)[0]
synth.synthetic_weights(g, 1234)
prog = _capi.Program(ctx, g, (4, 4, 4))
src = ctx.malloc((n, n, n), np.uint8)
ctx.synth_volume_u8(1, (n, n, n), out=src)
dst = ctx.malloc((n, n, n), np.float32)
kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_F16S, dims=(n, n, n))
prog.infer_volume(src, (102,) * 3, (7,) * 3, dst=dst, **kw)
whole = dst.to_host()
prog.infer_volume(src, (102,) * 3, (7,) * 3, dst=dst, **kw)
again = dst.to_host()
print('run-to-run identical:', np.array_equal(whole, again))
if not np.array_equal(whole, again):
    d = np.argwhere(whole != again)
    print('  differing voxels', len(d), 'min', d.min(0), 'max', d.max(0))
rows = multi_gpu.n_tile_rows(n, 102, 7)
dst2 = ctx.malloc((n, n, n), np.float32)
for zr in multi_gpu.slab_partition(rows, 2):
    prog.infer_volume(src, (102,) * 3, (7,) * 3, dst=dst2, z_range=zr, **kw)
sh = dst2.to_host()
d = np.argwhere(sh[7:n - 7] != whole[7:n - 7])
print('slab vs whole differing voxels', len(d))
if len(d):
    print('  min', d.min(0) + [7, 0, 0], 'max', d.max(0) + [7, 0, 0])
    zs = np.unique(d[:, 0] + 7)
    print('  z planes', zs[:40], '...', len(zs))
    ys = np.unique(d[:, 1]); xs = np.unique(d[:, 2])
    print('  y', ys[:40], len(ys), ' x', xs[:40], len(xs))
    print('  max abs diff', np.abs(sh - whole).max())
f32 = ctx.malloc((n, n, n), np.float32)
prog.infer_volume(src, (102,) * 3, (7,) * 3, dst=f32, **dict(kw, precision=_capi.PREC_F32))
ref = f32.to_host()
print('whole vs f32 max', np.abs(whole - ref).max(), ' slabs vs f32 max', np.abs(sh - ref).max())
e = np.abs(whole - ref)
bad = np.argwhere(e > 1e-5)
print('whole: voxels > 1e-5 off f32:', len(bad))
if len(bad):
    print('  min', bad.min(0), 'max', bad.max(0))
e = np.abs(sh - ref)
bad = np.argwhere(e > 1e-5)
print('slabs: voxels > 1e-5 off f32:', len(bad))
if len(bad):
    print('  min', bad.min(0), 'max', bad.max(0), 'z planes', np.unique(bad[:, 0])[:50])
