"""dev probe: per-layer gradient error of small conv stacks (GPU trainer vs fp64 oracle)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flypylib_amd import _capi, runtime
from flypylib_amd.program import LayerGraph
from oracle import train_oracle

ctx = runtime.get_context(0)


def rel(x, y):
    return np.max(np.abs(np.asarray(x, np.float64) - y)) / (np.max(np.abs(y)) + 1e-30)


for (c1, c2, patch, batch) in [(64, 64, 8, 2), (64, 64, 20, 2), (32, 64, 10, 2), (64, 64, 8, 1),
                               (16, 16, 8, 2), (64, 32, 8, 2), (48, 48, 8, 2), (64, 64, 6, 2)]:
    g = LayerGraph(None, seed=3)
    x = g.relu(g.conv(g.input(), c1, 1))
    x = g.relu(g.conv(x, c2, 3))
    g.finish(g.conv(x, 1, 1, use_bias=True, activation='sigmoid'))
    rng = np.random.default_rng(3)
    data = rng.standard_normal((batch, patch, patch, patch, 1)).astype(np.float32)
    o = patch - 2
    labels = (rng.random((batch, o, o, o, 1)) > 0.5).astype(np.uint8)
    tr = _capi.Trainer(ctx, g)
    l, a = tr.step(data, labels, seed=1)
    rl, ra, rg = train_oracle.train_step(g, g.weights, data, labels, 1)
    print('c1 %d c2 %d patch %d batch %d: loss %.7f vs %.7f' % (c1, c2, patch, batch, l, rl))
    for i, (gg, r) in enumerate(zip(tr.get_grads(), rg)):
        print('   %-20s %.2e' % (g.weight_names[i], rel(gg, r)))
    tr.close()
