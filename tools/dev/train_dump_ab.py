"""Layer-by-layer A/B of a training step: split-half convolutions against the fp32 kernels (FPL_TRAIN_F32CONV=1),
from the raw dumps of every activation and activation gradient (FPL_TRAIN_DUMP).  Prints, per tensor, the largest
difference relative to the tensor's largest entry and how many entries differ by more than 1e-4 of it - a flipped
max-pool window shows as a handful of entries, an indexing bug as a dense difference.
    python tools/dev/train_dump_ab.py [patch edge] [patches]"""
import glob
import os
import subprocess
import sys
import tempfile

import numpy as np

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1
CHILD = '''
import sys, numpy as np
sys.path.insert(0, '.')
from flypylib_amd import _capi, fplmodels, synth, runtime
ctx = runtime.get_context(0)
g = fplmodels.unet_like2()[0]
synth.synthetic_weights(g, 5)
rng = np.random.default_rng(4)
T, N = %d, %d
data = rng.standard_normal((N, T, T, T, 1)).astype(np.float32)
labels = (rng.random((N, T - 18, T - 18, T - 18, 1)) > 0.5).astype(np.uint8)
tr = _capi.Trainer(ctx, g)
print('loss', tr.step(data, labels, seed=5)[0])
np.savez(sys.argv[1] + '/grads.npz', *tr.get_grads())
''' % (T, N)
dirs = {}
for tag, env in (('split', {}), ('f32', {'FPL_TRAIN_F32CONV': '1'})):
    d = tempfile.mkdtemp(prefix='dump_' + tag)
    dirs[tag] = d
    e = dict(os.environ, FPL_TRAIN_DUMP=d, **env)
    r = subprocess.run([sys.executable, '-c', CHILD, d], env=e, capture_output=True, text=True)
    print(tag, [l for l in r.stdout.splitlines() if l.startswith('loss')], r.stderr[-300:] if r.returncode else '')
for f in sorted(glob.glob(dirs['split'] + '/*.f32')):
    a = np.fromfile(f, np.float32)
    b = np.fromfile(os.path.join(dirs['f32'], os.path.basename(f)), np.float32)
    m = float(np.abs(b).max()) + 1e-30
    d = np.abs(a - b)
    print('%-22s n %9d  max rel %.2e  entries > 1e-4: %d' % (os.path.basename(f), a.size, d.max() / m, int((d > 1e-4 * m).sum())))
ga, gb = np.load(dirs['split'] + '/grads.npz'), np.load(dirs['f32'] + '/grads.npz')
for k in ga.files:
    m = float(np.abs(gb[k]).max()) + 1e-30
    print('weight grad %-8s max rel %.2e' % (k, float(np.abs(ga[k] - gb[k]).max()) / m))
