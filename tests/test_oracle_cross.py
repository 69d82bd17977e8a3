"""Cross-check of the two independent restatements of the Keras arithmetic (CPU only):
oracle/cnn_oracle.py + oracle/train_oracle.py (torch, autograd) against
oracle/keras_np.py (numpy float64, explicit tap loops, hand-derived backward pass).

Neither is pinned to Keras (absent here, the reference holds no fixtures for its
networks: PARITY UNPINNED); agreement to 1e-10 removes the
single-author / single-library failure mode: cross-correlation orientation, BN epsilon
placement, the pool's floor, concat order and crop, the BCE clip, the batch-statistics
backward, Adam's bias correction and epsilon."""
import numpy as np
import pytest
import torch

from flypylib_amd import fplmodels, synth
from flypylib_amd.program import LayerGraph
from oracle import cnn_oracle, keras_np, train_oracle


def _rand_weights(graph, seed):
    """non-trivial BN statistics, signed kernels, non-zero bias"""
    rng = np.random.default_rng(seed)
    out = []
    for w, name in zip(graph.weights, graph.weight_names):
        if name.endswith('moving_variance') or name.endswith('gamma'):
            out.append(rng.uniform(0.5, 1.5, w.shape))
        elif name.endswith('kernel'):
            out.append(rng.normal(0, 1.0 / np.sqrt(np.prod(w.shape[:4])), w.shape))
        else:
            out.append(rng.normal(0, 0.2, w.shape))
    return out


def test_vgg_like_forward_two_restatements_agree():
    g = fplmodels.vgg_like(26)[0]           # 26 -> 24 -> 12 -> 10 -> 5 -> 3
    weights = _rand_weights(g, 1)
    x = np.random.default_rng(2).normal(0, 1, (2, 26, 26, 26, 1))
    a = cnn_oracle.vgg_like_forward(x, weights, None, dtype=torch.float64)
    b = keras_np.vgg_like_forward(x, weights)
    assert a.shape == b.shape == (2, 3, 3, 3, 1)
    assert 0.005 < b.std() and 0.1 < b.mean() < 0.9    # not saturated: the comparison means something
    assert np.abs(np.asarray(a) - b).max() < 1e-10
    # and through the generic graph walker the GPU tests use
    g64 = _with(g, weights)                  # the graph stores float32 weights
    c = cnn_oracle.graph_forward(g64, x, dtype=torch.float64)
    b32 = keras_np.vgg_like_forward(x, g64.get_weights())
    assert np.abs(np.asarray(c) - b32).max() < 1e-10


def _with(graph, weights):
    graph.set_weights([np.asarray(w, np.float32) for w in weights])
    return graph


def test_unet_like2_forward_two_restatements_agree():
    g = fplmodels.unet_like2(28)[0]          # 28 = 24 + 4: the concats need in = 0 (mod 4)
    weights = _rand_weights(g, 3)
    x = np.random.default_rng(4).normal(0, 1, (1, 28, 28, 28, 1))
    a = cnn_oracle.unet_like2_forward(x, weights, dtype=torch.float64)
    b = keras_np.unet_like2_forward(x, weights)
    assert a.shape == b.shape == (1, 10, 10, 10, 1)
    assert 0.002 < b.std() and 0.05 < b.mean() < 0.95
    assert np.abs(np.asarray(a) - b).max() < 1e-10


def _small_graph():
    g = LayerGraph(None)
    x = g.pool(g.relu(g.bn(g.conv(g.input(), 4, 3))))
    return g.finish(g.conv(x, 1, 1, use_bias=True, activation='sigmoid'))


def test_training_step_hand_derived_backward_agrees_with_autograd():
    """conv3 -> BN (batch statistics) -> ReLU -> pool2 -> conv1 + bias -> sigmoid, binary
    cross-entropy: loss, accuracy, every gradient, the moving-average update and Adam's
    first two steps"""
    g = _small_graph()
    rng = np.random.default_rng(5)
    w1 = rng.normal(0, 0.3, (3, 3, 3, 1, 4))
    gamma, beta = rng.uniform(0.5, 1.5, 4), rng.normal(0, 0.2, 4)
    mm, mv = rng.normal(0, 0.2, 4), rng.uniform(0.5, 1.5, 4)
    w2, b2 = rng.normal(0, 0.5, (1, 1, 1, 4, 1)), rng.normal(0, 0.2, 1)
    weights = [w1, gamma, beta, mm, mv, w2, b2]
    assert [w.shape for w in g.weights] == [np.shape(w) for w in weights]
    # float32-representable input: the torch oracle takes the batch as the engine does
    x = rng.normal(0, 1, (3, 8, 8, 8, 1)).astype(np.float32).astype(np.float64)
    y = (rng.random((3, 3, 3, 3, 1)) > 0.5).astype(np.uint8)

    mine = keras_np.small_net_step(x, y, w1, gamma, beta, mm, mv, w2, b2)
    loss, acc, grads = train_oracle.train_step(g, weights, x, y, seed=0)
    assert abs(loss - mine['loss']) < 1e-12 and abs(acc - mine['accuracy']) < 1e-12
    for got, want in zip([grads[0], grads[1], grads[2], grads[5], grads[6]], mine['grads']):
        assert np.abs(got - want).max() < 1e-12 * max(1.0, np.abs(want).max())
        assert np.abs(want).max() > 1e-6
    # moving statistics: the torch oracle reports the pending delta
    assert np.abs(mm + grads[3] - mine['moving'][0]).max() < 1e-12
    assert np.abs(mv + grads[4] - mine['moving'][1]).max() < 1e-12

    # two Adam steps on the same batch (bias correction differs between t = 1 and 2)
    adam = train_oracle.Adam(g)
    params = [w1, gamma, beta, w2, b2]
    m = [np.zeros_like(p) for p in params]
    v = [np.zeros_like(p) for p in params]
    cur = list(weights)
    for t in (1, 2):
        step = keras_np.small_net_step(x, y, params[0], params[1], params[2], cur[3], cur[4],
                                       params[3], params[4])
        _, _, gr = train_oracle.train_step(g, cur, x, y, seed=t)
        cur = adam.apply(cur, gr)
        params, m, v = keras_np.adam_step(params, step['grads'], m, v, t)
        for got, want in zip([cur[0], cur[1], cur[2], cur[5], cur[6]], params):
            assert np.abs(got - want).max() < 1e-12
        assert np.abs(cur[3] - step['moving'][0]).max() < 1e-12
        assert np.abs(cur[4] - step['moving'][1]).max() < 1e-12
    assert np.abs(params[0] - w1).max() > 1e-3          # the weights did move


def test_adam_epsilon_is_the_keras_2_0_value():
    """Keras <= 2.1.2 (the reference's era): epsilon 1e-8; with 1e-7 the first step of a
    tiny gradient differs measurably - the two restatements and the engine's default
    (flypylib_amd/train.py::_OPTIMIZERS) must all say 1e-8"""
    from flypylib_amd import train
    assert train._OPTIMIZERS['adam']['eps'] == 1e-8
    p, g = [np.array([1.0])], [np.array([1e-7])]
    a, _, _ = keras_np.adam_step(p, g, [np.zeros(1)], [np.zeros(1)], 1)
    b, _, _ = keras_np.adam_step(p, g, [np.zeros(1)], [np.zeros(1)], 1, eps=1e-7)
    assert abs((1.0 - a[0][0]) / (1.0 - b[0][0]) - 1.0) > 0.5
