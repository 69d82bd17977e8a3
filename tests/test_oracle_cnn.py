"""CPU: hand-derived known-answer tests that pin the Keras-semantics restatement
(oracle/cnn_oracle.py) - the reference's Keras/TF arithmetic cannot run here
(parity unpinned at that boundary)."""
import numpy as np
import torch

from flypylib_amd import fplmodels
from oracle import cnn_oracle


def test_conv_is_cross_correlation_valid_channels_last():
    # impulse at the centre of a 5^3 input: output = kernel flipped? No - for
    # cross-correlation out[o] = sum_t in[o+t] w[t], so an impulse at p yields
    # out[o] = w[p-o]: the kernel appears REVERSED in the output.
    k = np.arange(27, dtype=np.float64).reshape(3, 3, 3, 1, 1)
    x = np.zeros((5, 5, 5, 1))
    x[2, 2, 2, 0] = 1.0
    out = cnn_oracle.conv3d_valid_numpy(x, k)
    assert out.shape == (3, 3, 3, 1)
    assert np.array_equal(out[..., 0], k[::-1, ::-1, ::-1, 0, 0])
    t = cnn_oracle.conv3d_valid(
        torch.tensor(x, dtype=torch.float64).permute(3, 0, 1, 2)[None],
        torch.tensor(k, dtype=torch.float64))
    assert np.array_equal(t[0, 0].numpy(), out[..., 0])


def test_conv_torch_matches_loop_restatement_multichannel():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((6, 7, 8, 3))
    k = rng.standard_normal((3, 3, 3, 3, 5))
    ref = cnn_oracle.conv3d_valid_numpy(x, k)
    t = cnn_oracle.conv3d_valid(
        torch.tensor(x).permute(3, 0, 1, 2)[None], torch.tensor(k))
    assert np.allclose(t[0].permute(1, 2, 3, 0).numpy(), ref, atol=1e-12)


def test_bn_inference_algebra():
    x = torch.full((1, 2, 1, 1, 1), 3.0)
    g, b = torch.tensor([2.0, 1.0]), torch.tensor([0.5, -1.0])
    m, v = torch.tensor([1.0, 3.0]), torch.tensor([4.0 - 1e-3, 1.0 - 1e-3])
    y = cnn_oracle.bn_infer(x, g, b, m, v)
    assert np.allclose(y.view(-1).numpy(), [2 * (3 - 1) / 2 + 0.5, -1.0], atol=1e-6)


def test_pool_floor_and_upsample_repeat():
    x = torch.arange(5 * 5 * 5, dtype=torch.float32).view(1, 1, 5, 5, 5)
    p = cnn_oracle.maxpool2(x)
    assert p.shape == (1, 1, 2, 2, 2)            # floor(5/2): last plane dropped
    assert p[0, 0, 0, 0, 0] == x[0, 0, 1, 1, 1] and p[0, 0, 1, 1, 1] == x[0, 0, 3, 3, 3]
    u = cnn_oracle.upsample(p, 4)
    assert u.shape == (1, 1, 8, 8, 8)
    assert torch.equal(u[0, 0, :4, :4, :4], p[0, 0, 0, 0, 0].expand(4, 4, 4))


def test_vgg_receptive_field_offset_stride():
    """coarse output o sees input [4o, 4o+18) per axis and nothing else"""
    g = fplmodels.vgg_like(30)[0]
    g.randomize_bn(5)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((1, 30, 30, 30, 1)).astype(np.float32)
    base = cnn_oracle.vgg_like_forward(x, g.weights)
    assert base.shape == (1, 4, 4, 4, 1)
    for o in (0, 1, 3):
        lo, hi = 4 * o, 4 * o + 18
        y = x.copy()
        y[0, :lo] += 1.0          # outside the window of output (o, *, *)
        y[0, hi:] -= 1.0
        out = cnn_oracle.vgg_like_forward(y, g.weights)
        assert np.array_equal(out[0, o], base[0, o])
        y = x.copy()
        y[0, lo, 5, 5, 0] += 5.0  # first voxel of the window does matter
        assert not np.array_equal(
            cnn_oracle.vgg_like_forward(y, g.weights)[0, o], base[0, o])


def test_vgg_zero_weights_give_sigmoid_of_bias():
    g = fplmodels.vgg_like(22)[0]
    w = [np.zeros_like(a) for a in g.weights]
    for n in g.nodes:
        if n.kind == 'bn':
            w[n.weight_slots[3]] = np.ones_like(w[n.weight_slots[3]])
    w[-1] = np.array([0.7], np.float32)
    out = cnn_oracle.vgg_like_forward(np.ones((1, 22, 22, 22, 1), np.float32), w)
    assert np.allclose(out, 1 / (1 + np.exp(-0.7)), atol=1e-7)


def test_unet_concat_order_is_upsampled_then_skip():
    """weights of conv4 that read channels [0,128) see the upsampled path, the
    rest the skip (reference fplmodels.py:284)"""
    g = fplmodels.unet_like2(24)[0]
    g.randomize_bn(7)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 24, 24, 24, 1)).astype(np.float32)
    w = g.get_weights()
    conv4 = [n for n in g.nodes if n.kind == 'conv'][5]
    assert g.weights[conv4.weight_slots[0]].shape == (3, 3, 3, 192, 64)
    base = cnn_oracle.unet_like2_forward(x, w)
    w2 = [a.copy() for a in w]
    w2[conv4.weight_slots[0]][:, :, :, 128:, :] = 0        # drop the skip half
    g2 = fplmodels.unet_like2(24)[0]
    g2.set_weights(w2)
    a = cnn_oracle.unet_like2_forward(x, w2)
    b = cnn_oracle.graph_forward(g2, x)
    assert np.allclose(a, b, atol=1e-6) and not np.allclose(a, base)


def test_unet_output_is_shift_equivariant_only_mod_4():
    """the U-Net tile-phase trap of SURVEY section 7: a shift by 4 commutes
    with the network, a shift by 2 does not"""
    g = fplmodels.unet_like2(28)[0]
    g.randomize_bn(11)
    rng = np.random.default_rng(4)
    big = rng.standard_normal((1, 36, 28, 28, 1)).astype(np.float32)
    f = lambda z0: cnn_oracle.unet_like2_forward(big[:, z0:z0 + 28], g.weights)
    o0, o2, o4 = f(0), f(2), f(4)
    assert np.allclose(o0[0, 4:], o4[0, :-4], atol=1e-5)
    assert not np.allclose(o0[0, 2:], o2[0, :-2], atol=1e-3)


def test_oracle_losses_known_answers():
    """hand-derived values of the reference's losses (fplnetwork.py:74-77,
    fplmodels.py:28-65) on a 4-voxel example: p = [.9, .2, .6, .3], y = [1, 0, 2, 1]"""
    import math
    import torch
    from oracle import train_oracle
    p = torch.tensor([0.9, 0.2, 0.6, 0.3], dtype=torch.float64).reshape(1, 1, 1, 4, 1)
    y = torch.tensor([1.0, 0.0, 2.0, 1.0], dtype=torch.float64).reshape(1, 1, 1, 4, 1)
    eps = 1e-7
    # focal: -(1-pt)^2 log(pt + eps), masked voxel contributes 0, mean over all 4
    pts = [0.9, 0.8, None, 0.3]
    focal = sum(-(1 - q) ** 2 * math.log(q + eps) for q in pts if q is not None) / 4
    assert abs(float(train_oracle.loss_value(p, y, 'masked_focal_loss')) - focal) < 1e-12
    # masked BCE: masked voxel -> BCE(0 clipped to eps, 0) = -log(1 - eps)
    bce = (-math.log(0.9) - math.log(0.8) - math.log(1 - eps) - math.log(0.3)) / 4
    assert abs(float(train_oracle.loss_value(p, y, 'masked_binary_crossentropy')) - bce) < 1e-9
    # weighted: positives x100
    wbce = (-100 * math.log(0.9) - math.log(0.8) - math.log(1 - eps) - 100 * math.log(0.3)) / 4
    got = float(train_oracle.loss_value(p, y, 'masked_weighted_binary_crossentropy'))
    assert abs(got - wbce) < 1e-6
    # plain BCE on {0,1} labels
    y01 = torch.tensor([1.0, 0.0, 0.0, 1.0], dtype=torch.float64).reshape(p.shape)
    b = (-math.log(0.9) - math.log(0.8) - math.log(0.4) - math.log(0.3)) / 4
    assert abs(float(train_oracle.loss_value(p, y01, 'binary_crossentropy')) - b) < 1e-9
    m = train_oracle.metric_values(p, y)
    assert m['masked_accuracy'] == 0.75          # voxel 3 (p=.3, y=1) is the only miss
    assert abs(m['lb0l1err'] - 0.2) < 1e-12 and abs(m['lb1l1err'] - (0.1 + 0.7) / 2) < 1e-12
    assert m['acc'] == 0.5                        # label 2 never equals round(p)
