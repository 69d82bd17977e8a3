"""CPU restatement (torch autograd, float64) of one training step of the
reference: `train_network.fit_generator` per-batch work with the default compile
args (`flypylib/fplnetwork.py:74-77,112-122`) on the Keras layers of
`flypylib/fplmodels.py`.  TEST INFRASTRUCTURE - see oracle/__init__.py.

PARITY UNPINNED (Keras/TensorFlow absent, unpinned): semantics restated from the
Keras 2 sources of the era (python 3.6 / numpy 1.13, conda-recipe/meta.yaml):
  BatchNormalization  training: batch mean / biased variance over (N,D,H,W);
                      moving = 0.99*moving + 0.01*batch (non-fused 5-D path)
  Dropout(rate)       inverted scaling x/(1-rate); the keep mask is an input
                      (flypylib_amd.synth.dropout_keep_mask) so both sides agree
  binary_crossentropy output clipped to [1e-7, 1-1e-7], converted back to
                      logits, sigmoid_cross_entropy_with_logits, mean
  accuracy            mean(round(p) == y)
  Adam                lr 1e-3, beta 0.9/0.999, eps 1e-8:
                      lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)
"""
import numpy as np
import torch
import torch.nn.functional as F

from flypylib_amd import synth

BN_EPS, BN_MOMENTUM = 1e-3, 0.99


def loss_value(p, y, kind):
    """Scalar loss of the reference for sigmoid outputs p and labels y (same shape,
    last axis = 1): Keras means the per-voxel loss over everything.
    binary_crossentropy: fplnetwork.py:74-77 (Keras/TF: clip to [1e-7, 1-1e-7],
    sigmoid cross-entropy on the recovered logit); masked_*: fplmodels.py:28-50,
    label 2 = don't care."""
    eps = 1e-7
    if kind == 'masked_focal_loss':
        pt = torch.where(y == 1, p, 1 - p)
        mask = (y < 2).to(p.dtype)
        return (-(mask * (1 - pt) ** 2 * torch.log(pt + eps))).mean()
    mask = torch.ones_like(p) if kind == 'binary_crossentropy' else (y != 2).to(p.dtype)
    t, pm = y * mask, p * mask
    pc = pm.clamp(eps, 1 - eps)
    z = torch.log(pc / (1 - pc))
    if kind == 'masked_weighted_binary_crossentropy':
        w = 1 + 99 * t                        # pos_weight = 100
        return ((1 - t) * z + w * (torch.log1p(torch.exp(-z.abs())) + torch.relu(-z))).mean()
    if kind in ('binary_crossentropy', 'masked_binary_crossentropy'):
        return (torch.relu(z) - z * t + torch.log1p(torch.exp(-z.abs()))).mean()
    raise NotImplementedError(kind)


def metric_values(p, y):
    """Keras 'accuracy' and the reference's metrics (fplmodels.py:52-65)"""
    mask = (y != 2).to(p.dtype)
    m0, m1 = (y == 0).to(p.dtype), (y == 1).to(p.dtype)
    return {
        'acc': float((torch.round(p) == y).to(p.dtype).mean()),
        'masked_accuracy': float((y * mask == torch.round(p * mask)).to(p.dtype).mean()),
        'lb0l1err': float((p * m0).sum() / torch.clamp(m0.sum(), min=1)),
        'lb1l1err': float(((1 - p) * m1).sum() / torch.clamp(m1.sum(), min=1)),
    }


def train_step(graph, weights, data, labels, seed, dtype=torch.float64,
               loss='binary_crossentropy', return_metrics=False, info=None):
    """-> (loss, accuracy, grads) where `grads` follows the weight list order;
    for BN moving_mean / moving_variance the entry is the pending delta
    (1-momentum)*(batch_stat - moving).

    `info` (a dict) receives 'min_pool_gap': the smallest relative gap between the
    two largest values of any max-pool window with a positive maximum.  Max-pool
    routes the whole gradient of a window to its argmax, so a gap at rounding level
    (~1e-6) lets two correct fp32 implementations disagree on every gradient
    upstream; parity tests pick inputs whose gap is far above that."""
    W = [torch.tensor(np.asarray(w), dtype=dtype) for w in weights]
    trainable = []
    for n in graph.nodes:
        slots = n.weight_slots[:2] if n.kind == 'bn' else n.weight_slots
        for s in slots:
            W[s].requires_grad_(True)
            trainable.append(s)
    deltas = {}
    vals = {}
    x = torch.tensor(np.asarray(data, np.float32), dtype=dtype)
    if x.ndim == 4:
        x = x[..., None]
    li = -1
    for n in graph.nodes:
        if n.kind == 'input':
            vals[n.idx] = x.permute(0, 4, 1, 2, 3)
            continue
        li += 1
        a = [vals[i] for i in n.inputs]
        if n.kind == 'conv':
            kern = W[n.weight_slots[0]].permute(4, 3, 0, 1, 2)
            bias = W[n.weight_slots[1]] if n.attrs['use_bias'] else None
            v = F.conv3d(a[0], kern, bias)
            if n.attrs['activation'] == 'sigmoid':
                v = torch.sigmoid(v)
            elif n.attrs['activation'] == 'relu':
                v = torch.relu(v)
        elif n.kind == 'bn':
            g, b, mm, mv = (W[s] for s in n.weight_slots)
            mean = a[0].mean(dim=(0, 2, 3, 4))
            var = a[0].var(dim=(0, 2, 3, 4), unbiased=False)
            shp = (1, -1, 1, 1, 1)
            v = (a[0] - mean.view(shp)) / torch.sqrt(var.view(shp) + BN_EPS) \
                * g.view(shp) + b.view(shp)
            deltas[n.weight_slots[2]] = ((mean - mm) * (1 - BN_MOMENTUM)).detach()
            deltas[n.weight_slots[3]] = ((var - mv) * (1 - BN_MOMENTUM)).detach()
        elif n.kind == 'relu':
            v = torch.relu(a[0])
        elif n.kind == 'pool':
            v = F.max_pool3d(a[0], 2, 2)
            if info is not None:
                x_ = a[0].detach()
                d_, h_, w_ = (x_.shape[2] // 2 * 2, x_.shape[3] // 2 * 2, x_.shape[4] // 2 * 2)
                win = x_[:, :, :d_, :h_, :w_].unfold(2, 2, 2).unfold(3, 2, 2).unfold(4, 2, 2)
                top = win.reshape(*win.shape[:5], 8).topk(2, dim=-1).values
                pos = top[..., 0] > 0
                if pos.any():
                    gap = ((top[..., 0] - top[..., 1]) / top[..., 0])[pos].min()
                    info['min_pool_gap'] = min(info.get('min_pool_gap', 1.0), float(gap))
        elif n.kind == 'drop':
            rate = n.attrs['rate']
            cl = a[0].permute(0, 2, 3, 4, 1)               # channels-last order
            keep = synth.dropout_keep_mask(seed, li, cl.numel(), rate)
            keep = torch.tensor(keep.reshape(tuple(cl.shape))).permute(0, 4, 1, 2, 3)
            v = a[0] * keep.to(dtype) / (1.0 - rate)
        elif n.kind == 'up':
            v = a[0]
            for ax, f in zip((2, 3, 4), n.attrs['n']):
                v = torch.repeat_interleave(v, int(f), dim=ax)
        elif n.kind == 'crop':
            c = n.attrs['c']
            v = a[0][:, :, c[0][0]:a[0].shape[2] - c[0][1],
                     c[1][0]:a[0].shape[3] - c[1][1],
                     c[2][0]:a[0].shape[4] - c[2][1]]
        elif n.kind == 'concat':
            v = torch.cat(a, dim=1)
        elif n.kind == 'add':
            v = a[0] + a[1]
        else:
            raise NotImplementedError(n.kind)
        vals[n.idx] = v
    p = vals[graph.output.idx].permute(0, 2, 3, 4, 1)
    y = torch.tensor(np.asarray(labels), dtype=dtype).reshape(p.shape)
    loss = loss_value(p, y, loss)
    acc = (torch.round(p.detach()) == y).to(dtype).mean()
    metrics = metric_values(p.detach(), y)
    loss.backward()
    grads = []
    for i, w in enumerate(W):
        if i in deltas:
            grads.append(deltas[i].numpy())
        elif w.grad is not None:
            grads.append(w.grad.numpy())
        else:
            grads.append(np.zeros(tuple(w.shape)))
    if return_metrics:
        return float(loss.detach()), metrics, grads
    return float(loss.detach()), float(acc), grads


class Adam:
    """Keras-2.0-era Adam on a weight list; `moving` slots get additive deltas"""

    def __init__(self, graph, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.moving = set()
        for n in graph.nodes:
            if n.kind == 'bn':
                self.moving.update(n.weight_slots[2:])
        self.m = None

    def apply(self, weights, grads, scale=1.0):
        if self.m is None:
            self.m = [np.zeros(w.shape) for w in weights]
            self.v = [np.zeros(w.shape) for w in weights]
        self.t += 1
        lr_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        out = []
        for i, (w, g) in enumerate(zip(weights, grads)):
            w = np.asarray(w, np.float64)
            g = np.asarray(g, np.float64) * scale
            if i in self.moving:
                out.append(w + g)
                continue
            self.m[i] = self.b1 * self.m[i] + (1 - self.b1) * g
            self.v[i] = self.b2 * self.v[i] + (1 - self.b2) * g * g
            out.append(w - lr_t * self.m[i] / (np.sqrt(self.v[i]) + self.eps))
        return out
