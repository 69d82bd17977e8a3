"""dev probe: bisect the unet_like2 gradient mismatch on small graphs"""
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


def run(name, build, patch, batch=2):
    g = LayerGraph(None, seed=3)
    out = build(g)
    g.finish(out)
    g.randomize_bn(9)
    rng = np.random.default_rng(3)
    data = rng.standard_normal((batch, patch, patch, patch, 1)).astype(np.float32)
    tr = _capi.Trainer(ctx, g)
    # find the output size by a dry oracle forward
    import torch
    lab0 = np.zeros((batch, 1, 1, 1, 1), np.uint8)
    try:
        train_oracle.train_step(g, g.weights, data, lab0, 1)
    except Exception as e:
        msg = str(e)
    # brute force: try sizes
    for o in range(1, patch + 1):
        labels = (rng.random((batch, o, o, o, 1)) > 0.5).astype(np.uint8)
        try:
            rl, ra, rg = train_oracle.train_step(g, g.weights, data, labels, 1)
            break
        except Exception:
            continue
    l, a = tr.step(data, labels, seed=1)
    bad = [(g.weight_names[i], rel(gg, r)) for i, (gg, r) in enumerate(zip(tr.get_grads(), rg))
           if np.max(np.abs(r)) > 1e-12 and rel(gg, r) > 1e-4]
    print('%-40s out %d loss %.6f/%.6f  bad: %s' % (name, o, l, rl, bad[:4]))
    tr.close()


def g1(g):
    x = g.bn_relu(g.conv(g.input(), 64, 3))
    x = g.bn_relu(g.conv(x, 64, 3))
    return g.conv(x, 1, 1, activation='sigmoid')


def g2(g):
    x = g.bn_relu(g.conv(g.input(), 64, 1))
    x = g.bn_relu(g.conv(x, 64, 3))
    return g.conv(x, 1, 1, activation='sigmoid')


def g3(g):
    x = g.bn_relu(g.conv(g.input(), 64, 1))
    c2 = g.bn_relu(g.conv(x, 64, 3))
    p = g.pool(c2)
    c3 = g.bn_relu(g.conv(p, 128, 1))
    u = g.concat(g.up(c3), c2)
    x = g.bn_relu(g.conv(u, 64, 3))
    return g.conv(x, 1, 1, activation='sigmoid')


def g4(g):   # two consumers without pool/up
    x = g.bn_relu(g.conv(g.input(), 64, 1))
    c2 = g.bn_relu(g.conv(x, 64, 3))
    c3 = g.bn_relu(g.conv(c2, 64, 1))
    u = g.concat(c3, c2)
    x = g.bn_relu(g.conv(u, 64, 3))
    return g.conv(x, 1, 1, activation='sigmoid')


def g5(g):   # pool only
    x = g.bn_relu(g.conv(g.input(), 64, 1))
    c2 = g.bn_relu(g.conv(x, 64, 3))
    p = g.pool(c2)
    x = g.bn_relu(g.conv(p, 64, 1))
    return g.conv(x, 1, 1, activation='sigmoid')


for name, b, patch in [('stem+bn+conv3', g1, 10), ('conv1+bn+conv3', g2, 8), ('mini unet', g3, 8),
                       ('two consumers', g4, 8), ('pool', g5, 8)]:
    run(name, b, patch)
