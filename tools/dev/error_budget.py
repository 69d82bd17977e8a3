"""Error budget of the 16-bit vgg_like path on the committed trained weights: the fused
kernels round (a) the normalised input, (b) every folded weight tensor, (c) every
post-ReLU activation to the 16-bit operand type.  Each rounding point is switched on
alone (everything else fp32) and then all together, for IEEE half, bfloat16 and the
split representation hi + lo (two halves, ~22 bits); the figure is max / mean |dp| of
the sigmoid output against the all-fp32 forward on a 78^3 region of blob_region(2, 110).
CPU only (torch), about a minute.

    python tools/dev/error_budget.py [vgg_like]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cnn_oracle as co                     # noqa: E402
from tests.trained_fixture import blob_region, trained_weights   # noqa: E402


def rnd(t, kind):
    if kind is None:
        return t
    if kind == 'f16':
        return t.to(torch.float16).to(torch.float32)
    if kind == 'bf16':
        return t.to(torch.bfloat16).to(torch.float32)
    if kind == 'split':
        hi = t.to(torch.float16).to(torch.float32)
        lo = (t - hi).to(torch.float16).to(torch.float32)
        return hi + lo
    raise ValueError(kind)


def forward(x, weights, r_in, r_w, r_a):
    """r_w[l], r_a[l]: rounding kind of layer l's folded weights / output activations"""
    w = co._W(weights, torch.float32)
    h = rnd(co._t(x, torch.float32).permute(0, 4, 1, 2, 3), r_in)
    pools = (False, True, False, True, False, False, False)
    for l, pool in enumerate(pools):
        kern = w.take()
        g, b, m, v = w.take(4)
        s = g / torch.sqrt(v + co.BN_EPS)
        kf = rnd(kern * s.view(1, 1, 1, 1, -1), r_w[l])
        y = co.conv3d_valid(h, kf) + (b - m * s).view(1, -1, 1, 1, 1)
        if pool:
            y = co.maxpool2(y)
        h = rnd(torch.relu(y), r_a[l])
    kern, bias = w.take(2)
    logit = co.conv3d_valid(h, rnd(kern, r_w[7])) + bias.view(1, -1, 1, 1, 1)
    return torch.sigmoid(logit).numpy(), logit.numpy()


def main():
    torch.set_num_threads(os.cpu_count())
    weights = trained_weights('vgg_like')
    im, _, _ = blob_region(2, 110)
    x = im[10:88, 10:88, 10:88][None, ..., None]
    none8, none7 = [None] * 8, [None] * 7
    ref, ref_logit = forward(x, weights, None, none8, none7)
    print('reference: %d outputs, logits %.1f .. %.1f, p > 0.5: %d'
          % (ref.size, ref_logit.min(), ref_logit.max(), int((ref > 0.5).sum())))

    def report(tag, p):
        d = np.abs(p - ref)
        print('  %-34s max %.2e  mean %.2e' % (tag, d.max(), d.mean()))
        return d.max()

    for kind in ('f16', 'bf16', 'split'):
        print(kind)
        report('input only', forward(x, weights, kind, none8, none7)[0])
        for l in range(8):
            rw = list(none8)
            rw[l] = kind
            report('weights of L%d only' % (l + 1), forward(x, weights, None, rw, none7)[0])
        for l in range(7):
            ra = list(none7)
            ra[l] = kind
            report('activations after L%d only' % (l + 1), forward(x, weights, None, none8, ra)[0])
        report('all weights', forward(x, weights, None, [kind] * 8, none7)[0])
        report('input + all activations', forward(x, weights, kind, none8, [kind] * 7)[0])
        report('everything (the kernels)', forward(x, weights, kind, [kind] * 8, [kind] * 7)[0])
    # mixed schemes: what is the cheapest set of split points that holds 5e-4?
    print('mixed (f16 everywhere except ...)')
    for tag, split_w, split_a, split_in in (
            ('split tail L5-L8 (w + a)', (4, 5, 6, 7), (4, 5, 6), False),
            ('split tail + mid L3-L8', (2, 3, 4, 5, 6, 7), (2, 3, 4, 5, 6), False),
            ('split all activations + input', (), (0, 1, 2, 3, 4, 5, 6), True),
            ('split all weights', (0, 1, 2, 3, 4, 5, 6, 7), (), False)):
        rw = ['split' if l in split_w else 'f16' for l in range(8)]
        ra = ['split' if l in split_a else 'f16' for l in range(7)]
        report(tag, forward(x, weights, 'split' if split_in else 'f16', rw, ra)[0])


if __name__ == '__main__':
    main()
