"""Random layer programs through the graph executor (csrc/gx_exec.h) against the fp32 CPU oracle.

    python tools/dev/gx_fuzz.py [n graphs] [seed]

Two families: encoders (first layer, 3x3x3 / 1x1x1 convolutions of 16 - 64 channels with or without
BatchNorm, 0 - 2 max-pools, an optional residual Add) and U shapes (one pool, one UpSampling3D, the skip
cropped and concatenated, convolutions either side).  Each graph runs at precision f16s over a volume of two
tiles along z and ragged edges elsewhere, and must sit within 1e-5 of the oracle over the reference lattice;
graphs the executor declines are counted, not failed."""
import sys

import numpy as np

sys.path.insert(0, '.')
from flypylib_amd import _capi, runtime, synth
from flypylib_amd.program import LayerGraph
from oracle import cnn_oracle, infer_oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = runtime.get_context(0)
WIDTHS = [16, 32, 48, 64]


def layer(g, x, dims, rng, relu=True):
    """a random convolution; returns (tensor, dims)"""
    k = 3 if (rng.random() < 0.6 and dims >= 8) else 1
    w = int(rng.choice(WIDTHS))
    if rng.random() < 0.7:
        y = g.conv_bn_relu(x, w, k)
    else:
        y = g.conv(x, w, k, use_bias=bool(rng.random() < 0.5), activation='relu')
    return y, dims - (k - 1), w


def build(T, rng):
    g = LayerGraph(T)
    dims = T - 2
    c0 = int(rng.choice([16, 32]))
    x = g.conv_bn_relu(g.input(), c0, 3) if rng.random() < 0.7 else g.conv(g.input(), c0, 3, use_bias=True, activation='relu')
    w = c0
    stride = 1
    desc = ['stem%d' % c0]
    if rng.random() < 0.5:                       # ---- encoder family
        for _ in range(int(rng.integers(0, 3))):
            x, dims, w = layer(g, x, dims, rng)
            desc.append('c%d' % w)
        for _ in range(int(rng.integers(0, 3))):
            if dims % 2 or dims < 12:
                break
            x = g.pool(x)
            dims //= 2
            stride *= 2
            desc.append('pool')
            for _ in range(int(rng.integers(1, 3))):
                x, dims, w = layer(g, x, dims, rng)
                desc.append('c%d' % w)
        if rng.random() < 0.5 and dims >= 6:     # residual: conv3 (relu), conv1 without activation, add the cropped input
            y = g.conv_bn_relu(x, w, 3)
            y = g.bn(g.conv(y, w, 1)) if rng.random() < 0.5 else g.conv(y, w, 1, use_bias=True)
            x = g.relu(g.add(g.crop(x, 1), y))
            dims -= 2
            desc.append('res')
    else:                                        # ---- U family
        x, dims, w = layer(g, x, dims, rng)
        desc.append('c%d' % w)
        if dims % 2:
            x = g.conv_bn_relu(x, w, 3) if dims % 2 == 0 else g.conv_bn_relu(x, w, 1)
        skip, sdims, sw = x, dims, w
        if dims % 2:
            return None
        y = g.pool(x)
        d = dims // 2
        for _ in range(int(rng.integers(1, 3))):
            y, d, w = layer(g, y, d, rng)
            desc.append('lo%d' % w)
        up = 2 * d
        crop = sdims - up
        if crop < 0 or crop % 4:
            return None
        x = g.concat(g.up(y, 2), g.crop(skip, crop // 2) if crop else skip)
        desc.append('up+skip(crop %d)' % (crop // 2))
        dims = up
        x = g.conv_bn_relu(x, int(rng.choice([32, 64])), 3)
        dims -= 2
        w = 0
        for _ in range(int(rng.integers(0, 2))):
            x, dims, w = layer(g, x, dims, rng)
            desc.append('c%d' % w)
    if dims < 2:
        return None
    g.finish(g.conv(x, 1, 1, use_bias=bool(rng.random() < 0.5), activation='sigmoid'))
    out = dims * stride
    if (T - out) % 2 or T - out < 0:
        return None
    return g, (T - out) // 2, stride, ' '.join(desc)


ok = declined = 0
worst = 0.0
tried = 0
while ok + declined < N and tried < 40 * N:
    tried += 1
    T = int(rng.choice([24, 28, 32, 36, 40]))
    b = build(T, rng)
    if b is None:
        continue
    g, off, stride, desc = b
    synth.synthetic_weights(g, int(rng.integers(1, 1000)))
    try:
        prog = _capi.Program(ctx, g, (stride,) * 3)
    except Exception as e:                      # shapes the lowering refuses
        continue
    pitch = T - 2 * off
    shape = (2 * pitch + 2 * off, T + int(rng.integers(0, 9)), T + int(rng.integers(0, 20)))
    u8 = synth.em_volume_u8(int(rng.integers(1, 99)), shape)
    try:
        got = prog.infer_volume(u8, (T,) * 3, (off,) * 3, mean=128.0, std=33.0, precision=_capi.PREC_F16S)
    except _capi.FplHipError as e:
        if 'split-half kernels' in str(e):
            declined += 1
            print('declined: T %d  %s' % (T, desc), flush=True)
            continue
        raise
    assert ctx.last_path() == 'graph_split_f16', ctx.last_path()
    img = (u8.astype(np.float32) - np.float32(128)) / np.float32(33)

    def f32(batch):
        y = cnn_oracle.graph_forward(g, batch.astype(np.float32))
        for ax in (1, 2, 3):
            if stride != 1:
                y = np.repeat(y, stride, axis=ax)
        return y
    ref = infer_oracle.infer_lattice(img, (T,) * 3, (off,) * 3, f32)
    d = float(np.abs(got - ref).max())
    worst = max(worst, d)
    flag = 'ok ' if d < 1e-5 else 'BAD'
    print('%s %.2e  T %d off %d stride %d  %s' % (flag, d, T, off, stride, desc), flush=True)
    assert d < 1e-5, desc
    ok += 1
    prog.close()
print('graphs through the executor: %d within 1e-5 (worst %.2e), %d declined' % (ok, worst, declined))
