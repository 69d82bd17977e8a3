"""dev probe: dump trainer tensors with MFMA vs direct forward and diff them"""
import glob
import os
import subprocess
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1:
    from flypylib_amd import _capi, fplmodels, synth, runtime
    ctx = runtime.get_context(0)
    g = fplmodels.unet_like2()[0]
    synth.synthetic_weights(g, 6)
    rng = np.random.default_rng(8)
    data = rng.standard_normal((2, 24, 24, 24, 1)).astype(np.float32)
    labels = (rng.integers(0, 3, (2, 6, 6, 6, 1)) > 0).astype(np.uint8)
    tr = _capi.Trainer(ctx, g)
    tr.step(data, labels, seed=5)
    sys.exit(0)

out = os.path.join(ROOT, 'gpurun_out', 'dump')
for mode, env in (('mfma', {}), ('direct', {'FPL_TRAIN_DIRECT': '1'})):
    d = os.path.join(out, mode)
    os.makedirs(d, exist_ok=True)
    subprocess.run([sys.executable, __file__, 'child'], env=dict(os.environ, FPL_TRAIN_DUMP=d, **env), check=True)
for f in sorted(glob.glob(os.path.join(out, 'mfma', '*.f32'))):
    a = np.fromfile(f, np.float32)
    b = np.fromfile(f.replace('/mfma/', '/direct/'), np.float32)
    C = int(f.rsplit('_c', 1)[1].split('.')[0])
    e = np.abs(a - b).reshape(-1, C)
    rel = e.max() / (np.abs(b).max() + 1e-30)
    flag = ' <<<' if rel > 1e-4 else ''
    chunks = [float('%.1e' % e[:, k:k + 16].max()) for k in range(0, C, 16)]
    print('%-22s rel %.1e by 16-ch chunk %s%s' % (os.path.basename(f), rel, chunks[:8], flag))

