"""training step: split-half conv3 kernels against the fp32 ones (FPL_TRAIN_F32CONV=1), per gradient tensor"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
from flypylib_amd import _capi, fplmodels, synth, runtime
ctx = runtime.get_context(0)
for shape, lab in (((4, 18, 18, 18, 1), (4, 1, 1, 1, 1)), ((4, 30, 30, 30, 1), (4, 4, 4, 4, 1)), ((4, 46, 46, 46, 1), (4, 8, 8, 8, 1))):
    g = fplmodels.vgg_like()[0]
    synth.synthetic_weights(g, 7)
    rng = np.random.default_rng(4)
    data = rng.standard_normal(shape).astype(np.float32)
    labels = (rng.random(lab) > 0.7).astype(np.uint8)
    res = {}
    for mode in ('split', 'f32'):
        if mode == 'f32':
            os.environ['FPL_TRAIN_F32CONV'] = '1'
        tr = _capi.Trainer(ctx, g)
        loss, acc = tr.step(data, labels, seed=9)
        res[mode] = (loss, [x.copy() for x in tr.get_grads()])
        os.environ.pop('FPL_TRAIN_F32CONV', None)
    print(shape, 'loss', res['split'][0], res['f32'][0])
    for i, (a, b) in enumerate(zip(res['split'][1], res['f32'][1])):
        r = np.max(np.abs(a.astype(np.float64) - b)) / (np.max(np.abs(b)) + 1e-30)
        if r > 1e-5 or 'kernel' in g.weight_names[i] and a.ndim == 5 and a.shape[0] == 3 and a.shape[3] == 48:
            print('   %-28s %s rel %.2e  max|f32| %.2e' % (g.weight_names[i], a.shape, r, np.max(np.abs(b))))
