"""Second, torch-free restatement of the Keras arithmetic behind the reference's models
- TEST INFRASTRUCTURE, like everything under oracle/ (never imported by flypylib_amd).

Why it exists: Keras / TensorFlow are absent here and the reference holds no fixtures
for its networks, so `cnn_oracle.py` / `train_oracle.py` (torch, autograd) are PARITY
UNPINNED.  This file restates the same layers a second time, in plain numpy float64
with explicit loops over the kernel taps and a HAND-DERIVED backward pass, written from
the reference's model definitions and the Keras 2.0 - 2.1 layer semantics alone:

  * vgg_like        /root/reference/flypylib/fplmodels.py:102-136
  * unet_like2      /root/reference/flypylib/fplmodels.py:258-304
  * _bn_relu        /root/reference/flypylib/fplmodels.py:67-71
  * default compile /root/reference/flypylib/fplnetwork.py:74-77
    (binary_crossentropy, adam, accuracy)

A disagreement between the two restatements is a misreading in one of them
(tests/test_oracle_cross.py asserts agreement to 1e-10); agreement does not pin either
to Keras - nothing can without a Keras-generated fixture - but it removes the
single-author, single-library failure mode (operator orientation, BN epsilon placement,
the loss clip, Adam's bias correction and epsilon).

Keras semantics restated (TensorFlow backend, channels_last):
  Conv3D              'valid', stride 1, cross-correlation (no kernel flip), kernel
                      (kd, kh, kw, cin, cout)
  BatchNormalization  axis -1, momentum 0.99, epsilon 1e-3;
                      inference  gamma * (x - moving_mean) / sqrt(moving_var + eps) + beta
                      training   batch mean / BIASED batch variance over (N, D, H, W);
                                 moving <- momentum * moving + (1 - momentum) * batch
                                 (5-D inputs take Keras' non-fused path: the moving variance
                                 follows the biased estimate)
  MaxPooling3D(2)     stride 2, 'valid' (trailing odd row / column dropped)
  UpSampling3D(n)     nearest repeat;  Cropping3D;  concatenate on channels
  Dropout(r)          identity at inference; training: keep * x / (1 - r)
  binary_crossentropy output clipped to [1e-7, 1 - 1e-7], -y log p - (1 - y) log(1 - p),
                      mean over every output element
  Adam                lr 1e-3, beta 0.9 / 0.999, epsilon 1e-8 (Keras <= 2.1.2, the era the
                      reference's pins - python 3.6, numpy 1.13 - belong to; 2.1.3 moved
                      the default to 1e-7), lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t),
                      p -= lr_t * m / (sqrt(v) + epsilon)
"""
import numpy as np

BN_EPS = 1e-3
BN_MOMENTUM = 0.99
BCE_CLIP = 1e-7


# ---- layers, forward ----------------------------------------------------------------
def conv3d(x, kernel, bias=None):
    """x (N, D, H, W, Cin), kernel (k, k, k, Cin, Cout): one term per kernel tap"""
    x = np.asarray(x, np.float64)
    kernel = np.asarray(kernel, np.float64)
    k = kernel.shape[0]
    n, d, h, w, _ = x.shape
    od, oh, ow = d - k + 1, h - k + 1, w - k + 1
    out = np.zeros((n, od, oh, ow, kernel.shape[4]))
    for a in range(k):
        for b in range(k):
            for c in range(k):
                out += x[:, a:a + od, b:b + oh, c:c + ow, :] @ kernel[a, b, c]
    if bias is not None:
        out = out + np.asarray(bias, np.float64)
    return out


def bn_inference(x, gamma, beta, mean, var):
    return gamma * (x - mean) / np.sqrt(var + BN_EPS) + beta


def relu(x):
    return np.maximum(x, 0.0)


def maxpool2(x):
    n, d, h, w, c = x.shape
    x = x[:, :d // 2 * 2, :h // 2 * 2, :w // 2 * 2, :]
    x = x.reshape(n, d // 2, 2, h // 2, 2, w // 2, 2, c)
    return x.max(axis=(2, 4, 6))


def upsample(x, f):
    for ax in (1, 2, 3):
        x = np.repeat(x, f, axis=ax)
    return x


def crop(x, c):
    return x[:, c:x.shape[1] - c, c:x.shape[2] - c, c:x.shape[3] - c, :]


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


class _Weights:
    """weights in Keras get_weights() order: layer creation order; Conv3D [kernel(, bias)],
    BatchNormalization [gamma, beta, moving_mean, moving_variance]"""

    def __init__(self, weights):
        self.w = [np.asarray(a, np.float64) for a in weights]
        self.i = 0

    def take(self, n):
        out = self.w[self.i:self.i + n]
        self.i += n
        return out

    def conv_bn_relu(self, x):
        (kernel,) = self.take(1)
        gamma, beta, mean, var = self.take(4)
        return relu(bn_inference(conv3d(x, kernel), gamma, beta, mean, var))


def vgg_like_forward(x, weights):
    """reference fplmodels.py:102-136 at inference (Dropout = identity); x (N, D, H, W, 1).
    Returns the sigmoid probabilities at the network's own (coarse) resolution."""
    w = _Weights(weights)
    h = np.asarray(x, np.float64)
    for _ in range(2):                       # conv3 + conv1 + pool, twice (:110-122)
        h = w.conv_bn_relu(h)
        h = w.conv_bn_relu(h)
        h = maxpool2(h)
    h = w.conv_bn_relu(h)                    # conv3 48 (:124-125)
    h = w.conv_bn_relu(h)                    # 1x1 96 + Dropout (:127-129)
    h = w.conv_bn_relu(h)                    # 1x1 96 + Dropout (:131-133)
    kernel, bias = w.take(2)                 # 1x1 -> 1, bias, sigmoid (:135)
    assert w.i == len(w.w)
    return sigmoid(conv3d(h, kernel, bias))


def unet_like2_forward(x, weights):
    """reference fplmodels.py:258-304 at inference; no bias anywhere (:297)"""
    w = _Weights(weights)
    h = np.asarray(x, np.float64)
    c1 = w.conv_bn_relu(w.conv_bn_relu(h))               # :268-271
    c2 = w.conv_bn_relu(w.conv_bn_relu(maxpool2(c1)))    # :272-277
    c3 = w.conv_bn_relu(maxpool2(c2))                    # 1x1 128 (:278-281)
    u4 = np.concatenate([upsample(c3, 2), c2], axis=-1)  # [UpSampling, skip] (:284)
    c4 = w.conv_bn_relu(w.conv_bn_relu(u4))              # conv3 64, conv1 64 (:285-288)
    u5 = np.concatenate([upsample(c4, 2), crop(c1, 6)], axis=-1)   # (:290-291)
    c5 = w.conv_bn_relu(w.conv_bn_relu(u5))              # conv3 32, conv1 32 (:292-295)
    (kernel,) = w.take(1)
    assert w.i == len(w.w)
    return sigmoid(conv3d(c5, kernel))


# ---- one training step of  conv3 -> BN -> ReLU -> pool2 -> conv1 (+bias) -> sigmoid,
#      binary cross-entropy, Adam: forward and HAND-DERIVED backward ---------------------
def small_net_step(x, labels, w1, gamma, beta, moving_mean, moving_var, w2, b2):
    """Returns dict(loss, accuracy, grads = [dW1, dgamma, dbeta, dW2, db2],
    moving = (new moving_mean, new moving_var)).  x (N, D, H, W, 1), labels = the 0/1
    targets of the output voxels (N, d, h, w, 1)."""
    x = np.asarray(x, np.float64)
    y = np.asarray(labels, np.float64)
    w1, gamma, beta, w2, b2 = (np.asarray(a, np.float64) for a in (w1, gamma, beta, w2, b2))
    # forward
    a1 = conv3d(x, w1)
    mu = a1.mean(axis=(0, 1, 2, 3))
    var = ((a1 - mu) ** 2).mean(axis=(0, 1, 2, 3))       # biased
    inv = 1.0 / np.sqrt(var + BN_EPS)
    xhat = (a1 - mu) * inv
    y1 = gamma * xhat + beta
    r = relu(y1)
    n, d, h, w, c = r.shape
    assert d % 2 == 0 and h % 2 == 0 and w % 2 == 0
    win = r.reshape(n, d // 2, 2, h // 2, 2, w // 2, 2, c)
    p = win.max(axis=(2, 4, 6))
    z = conv3d(p, w2, b2)
    s = sigmoid(z)
    sc = np.clip(s, BCE_CLIP, 1.0 - BCE_CLIP)
    per_voxel = -(y * np.log(sc) + (1.0 - y) * np.log(1.0 - sc))
    loss = per_voxel.mean()
    acc = (np.round(s) == y).mean()
    # backward.  dL/ds = (-y / s + (1 - y) / (1 - s)) / M inside the clip range, 0 outside;
    # ds/dz = s (1 - s)
    m_out = y.size
    inside = (s > BCE_CLIP) & (s < 1.0 - BCE_CLIP)
    dz = np.where(inside, (s - y) / m_out, 0.0)
    # conv1 (1x1x1): z[..., o] = sum_c p[..., c] w2[0,0,0,c,o] + b2[o]
    dw2 = np.einsum('ndhwc,ndhwo->co', p, dz).reshape(w2.shape)
    db2 = dz.sum(axis=(0, 1, 2, 3))
    dp = dz @ w2[0, 0, 0].T
    # max-pool: the whole gradient of a window goes to its maximum.  Windows whose maximum
    # is 0 (every input negative before the ReLU) tie, harmlessly: the ReLU mask below
    # kills whatever they are handed; a tie between positive values would be ambiguous
    is_max = win == p[:, :, None, :, None, :, None, :]
    ties = is_max.sum(axis=(2, 4, 6)) > 1
    assert not (ties & (p > 0)).any(), 'tie between positive values inside a pooling window'
    dr = (is_max * dp[:, :, None, :, None, :, None, :]).reshape(r.shape)
    dy1 = dr * (y1 > 0)
    dgamma = (dy1 * xhat).sum(axis=(0, 1, 2, 3))
    dbeta = dy1.sum(axis=(0, 1, 2, 3))
    dxhat = dy1 * gamma
    # batch statistics depend on every element of a1:
    # da1 = inv * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat))
    da1 = inv * (dxhat - dxhat.mean(axis=(0, 1, 2, 3))
                 - xhat * (dxhat * xhat).mean(axis=(0, 1, 2, 3)))
    k = w1.shape[0]
    od, oh, ow = a1.shape[1:4]
    dw1 = np.zeros_like(w1)
    for a in range(k):
        for b in range(k):
            for cc in range(k):
                dw1[a, b, cc] = np.einsum('ndhwi,ndhwo->io',
                                          x[:, a:a + od, b:b + oh, cc:cc + ow, :], da1)
    new_mean = BN_MOMENTUM * np.asarray(moving_mean, np.float64) + (1 - BN_MOMENTUM) * mu
    new_var = BN_MOMENTUM * np.asarray(moving_var, np.float64) + (1 - BN_MOMENTUM) * var
    return dict(loss=loss, accuracy=acc, grads=[dw1, dgamma, dbeta, dw2, db2],
                moving=(new_mean, new_var), probabilities=s)


def adam_step(params, grads, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """Keras 2.0 - 2.1.2 `Adam.get_updates`; t = 1 for the first step.  Returns
    (new params, new m, new v)."""
    lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    out_p, out_m, out_v = [], [], []
    for p, g, mi, vi in zip(params, grads, m, v):
        mi = b1 * mi + (1.0 - b1) * g
        vi = b2 * vi + (1.0 - b2) * g * g
        out_p.append(p - lr_t * mi / (np.sqrt(vi) + eps))
        out_m.append(mi)
        out_v.append(vi)
    return out_p, out_m, out_v
