import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flypylib_amd import _capi, fplmodels, synth, runtime
from oracle import train_oracle
ctx = runtime.get_context(0)
g = fplmodels.unet_like2()[0]
synth.synthetic_weights(g, 6)
rng = np.random.default_rng(8)
data = rng.standard_normal((2, 24, 24, 24, 1)).astype(np.float32)
labels = rng.integers(0, 3, (2, 6, 6, 6, 1)).astype(np.uint8)
for loss in ['binary_crossentropy','masked_focal_loss']:
    lab = labels if loss!='binary_crossentropy' else (labels>0).astype(np.uint8)
    tr = _capi.Trainer(ctx, g, loss=loss)
    tr.step(data, lab, seed=5)
    rl, rm, rg = train_oracle.train_step(g, g.weights, data, lab, 5, loss=loss, return_metrics=True)
    r32 = train_oracle.train_step(g, g.weights, data, lab, 5, loss=loss, dtype=__import__('torch').float32)[2]
    print(loss)
    for i,(a,r,q) in enumerate(zip(tr.get_grads(), rg, r32)):
        if np.max(np.abs(r))<1e-12: continue
        rel=lambda x,y: np.max(np.abs(np.asarray(x,np.float64)-y))/np.max(np.abs(y))
        print('  %-28s gpu %.2e  torch32 %.2e  max|g| %.2e'%(g.weight_names[i], rel(a,r), rel(q,r), np.max(np.abs(r))))

# error pattern of conv_11/kernel
tr = _capi.Trainer(ctx, g)
lab = (labels > 0).astype(np.uint8)
tr.step(data, lab, seed=5)
rl, ra, rg = train_oracle.train_step(g, g.weights, data, lab, 5)
i = g.weight_names.index('conv_11/kernel')
a = np.asarray(tr.get_grads()[i], np.float64); r = rg[i]
e = np.abs(a - r)
print('shape', a.shape, 'max err', e.max(), 'max ref', np.abs(r).max())
print('err by tap (z,y,x):'); print(e.max(axis=(3, 4)).round(5))
print('err by ci chunk of 16:', [float(e[..., 16*k:16*k+16, :].max().round(5)) for k in range(4)])
print('err by co chunk of 16:', [float(e[..., 16*k:16*k+16].max().round(5)) for k in range(4)])
j = g.weight_names.index('bn_12/gamma')
print('bn_12 gamma err', np.abs(np.asarray(tr.get_grads()[j], np.float64) - rg[j]).max())
