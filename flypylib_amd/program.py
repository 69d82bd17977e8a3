"""Layer programs: the network description the HIP engine executes.

The reference describes its architectures as Keras graphs (`flypylib/fplmodels.py`);
here an architecture is a small SSA graph (`LayerGraph`) of the layer kinds those
graphs use:

    conv   Conv3D, 'valid', stride 1, cross-correlation, optional bias / activation
    bn     BatchNormalization(axis=-1, momentum=0.99, eps=1e-3)
    relu   Activation('relu')
    pool   MaxPooling3D(2)   (stride 2, floor)
    up     UpSampling3D(n)   (nearest repeat)
    crop   Cropping3D(c)     (symmetric or (lo, hi) per axis)
    concat concatenate([a, b]) on channels,  add  add([a, b])
    drop   Dropout(p)        (identity at inference)

Weights are kept as a flat list in Keras `get_weights()` order (layer creation
order; conv kernel `(kd,kh,kw,Cin,Cout)` then bias; BN `gamma, beta, moving_mean,
moving_var`), so weight lists are interchangeable with the reference's.

`LayerGraph.lower_inference()` folds BN/ReLU/Dropout into the producing conv and
emits the fused op list + flat fp32 weight arena consumed by `fpl_program_create`
(include/fplhip.h).
"""
import numpy as np

from . import fplutils

BN_EPS = 1e-3          # Keras BatchNormalization default epsilon
BN_MOMENTUM = 0.99     # Keras BatchNormalization default momentum

# fused-op kinds shared with include/fplhip.h (enum fpl_op_kind)
OP_CONV, OP_POOL, OP_UP, OP_CROP, OP_CONCAT, OP_ADD = 0, 1, 2, 3, 4, 5
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


class Node:
    __slots__ = ('idx', 'kind', 'inputs', 'attrs', 'channels', 'size',
                 'weight_slots')

    def __init__(self, idx, kind, inputs, attrs, channels, size):
        self.idx = idx
        self.kind = kind
        self.inputs = inputs
        self.attrs = attrs
        self.channels = channels
        self.size = size            # spatial (d,h,w) or None when fully convolutional
        self.weight_slots = []      # indices into LayerGraph.weights


def _crop_pairs(c):
    if np.isscalar(c):
        return ((c, c),) * 3
    out = []
    for cc in c:
        out.append((cc, cc) if np.isscalar(cc) else (int(cc[0]), int(cc[1])))
    return tuple(out)


class LayerGraph:
    """Keras-Model-shaped container around a list of `Node`s."""

    def __init__(self, in_sz=None, seed=0):
        in_sz = fplutils.to3d(in_sz)
        self.in_sz = None if in_sz[0] is None else tuple(int(s) for s in in_sz)
        self.nodes = []
        self.weights = []           # Keras get_weights() order
        self.weight_names = []
        self.output = None
        self._rng = np.random.default_rng(seed)
        self.compile_args = None
        # Adam state of the last training (m, v: lists in get_weights() order - the moving
        # statistics' slots are unused; iterations: updates applied), or None: what Keras keeps
        # in a compiled model's optimizer and `model.save` in `optimizer_weights`
        self.opt_state = None
        self.inputs_node = self._add('input', [], {}, 1, self.in_sz)

    # ---- graph construction -------------------------------------------------
    def _add(self, kind, inputs, attrs, channels, size):
        n = Node(len(self.nodes), kind, [i.idx for i in inputs], attrs,
                 channels, size)
        self.nodes.append(n)
        return n

    def _new_weight(self, node, name, arr):
        node.weight_slots.append(len(self.weights))
        self.weights.append(np.ascontiguousarray(arr, dtype=np.float32))
        self.weight_names.append('%s_%d/%s' % (node.kind, node.idx, name))

    def input(self):
        return self.inputs_node

    def conv(self, x, filters, k, use_bias=False, activation=None):
        size = None if x.size is None else tuple(s - (k - 1) for s in x.size)
        if size is not None and min(size) < 1:
            raise ValueError('conv3d: input %s too small for kernel %d'
                             % (x.size, k))
        n = self._add('conv', [x], dict(k=k, use_bias=use_bias,
                                        activation=activation), filters, size)
        fan_in, fan_out = k ** 3 * x.channels, k ** 3 * filters
        lim = np.sqrt(6.0 / (fan_in + fan_out))          # glorot_uniform
        self._new_weight(n, 'kernel', self._rng.uniform(
            -lim, lim, size=(k, k, k, x.channels, filters)))
        if use_bias:
            self._new_weight(n, 'bias', np.zeros(filters))
        return n

    def bn(self, x):
        n = self._add('bn', [x], {}, x.channels, x.size)
        c = x.channels
        self._new_weight(n, 'gamma', np.ones(c))
        self._new_weight(n, 'beta', np.zeros(c))
        self._new_weight(n, 'moving_mean', np.zeros(c))
        self._new_weight(n, 'moving_variance', np.ones(c))
        return n

    def relu(self, x):
        return self._add('relu', [x], {}, x.channels, x.size)

    def bn_relu(self, x):
        """BN -> ReLU block (reference `fplmodels.py:67-71`)"""
        return self.relu(self.bn(x))

    def conv_bn_relu(self, x, filters, k):
        return self.bn_relu(self.conv(x, filters, k))

    def pool(self, x, n=2):
        size = None if x.size is None else tuple(s // n for s in x.size)
        return self._add('pool', [x], dict(n=n), x.channels, size)

    def up(self, x, n=2):
        n3 = fplutils.to3d(n)
        size = None if x.size is None else tuple(
            s * f for s, f in zip(x.size, n3))
        return self._add('up', [x], dict(n=tuple(int(f) for f in n3)),
                         x.channels, size)

    def crop(self, x, c):
        pairs = _crop_pairs(c)
        size = None if x.size is None else tuple(
            s - lo - hi for s, (lo, hi) in zip(x.size, pairs))
        return self._add('crop', [x], dict(c=pairs), x.channels, size)

    def concat(self, a, b):
        if a.size is not None and a.size != b.size:
            raise ValueError('concatenate: spatial sizes differ %s vs %s'
                             % (a.size, b.size))
        return self._add('concat', [a, b], {}, a.channels + b.channels, a.size)

    def add(self, a, b):
        if a.size is not None and a.size != b.size:
            raise ValueError('add: spatial sizes differ %s vs %s'
                             % (a.size, b.size))
        return self._add('add', [a, b], {}, a.channels, a.size)

    def dropout(self, x, rate):
        return self._add('drop', [x], dict(rate=rate), x.channels, x.size)

    def finish(self, out):
        self.output = out
        return self

    # ---- Keras-Model-like surface -------------------------------------------
    @property
    def input_shape(self):
        s = self.in_sz if self.in_sz is not None else (None, None, None)
        return (None,) + tuple(s) + (1,)

    @property
    def output_shape(self):
        s = self.output.size if self.output.size is not None else (None,) * 3
        return (None,) + tuple(s) + (self.output.channels,)

    def get_weights(self):
        return [w.copy() for w in self.weights]

    def set_weights(self, weights):
        if len(weights) != len(self.weights):
            raise ValueError('set_weights: expected %d arrays, got %d'
                             % (len(self.weights), len(weights)))
        for i, (old, new) in enumerate(zip(self.weights, weights)):
            new = np.asarray(new, dtype=np.float32)
            if new.shape != old.shape:
                raise ValueError('set_weights: array %d (%s) has shape %s, '
                                 'expected %s' % (i, self.weight_names[i],
                                                  new.shape, old.shape))
            self.weights[i] = np.ascontiguousarray(new)

    def count_params(self):
        return int(sum(w.size for w in self.weights))

    def count_trainable(self):
        tot = 0
        for n in self.nodes:
            slots = n.weight_slots[:2] if n.kind == 'bn' else n.weight_slots
            tot += sum(self.weights[s].size for s in slots)
        return int(tot)

    def summary(self, print_fn=print):
        print_fn('%-4s %-8s %-14s %-18s %s' % ('#', 'layer', 'inputs',
                                               'output (d,h,w,c)', 'params'))
        for n in self.nodes:
            shp = (n.size if n.size is not None else (None,) * 3) + (n.channels,)
            prm = sum(self.weights[s].size for s in n.weight_slots)
            print_fn('%-4d %-8s %-14s %-18s %d' % (n.idx, n.kind,
                                                   str(n.inputs), str(shp), prm))
        print_fn('total params: %d (trainable %d)' % (
            self.count_params(), self.count_trainable()))

    def compile(self, **compile_args):
        """Keras builds a fresh optimizer on `compile()`: a recompiled network starts Adam from
        zero moments and step 0 (a loaded one gets its saved state back AFTER its compile:
        fplnetwork.load_network, as `keras.models.load_model` does)"""
        self.compile_args = dict(compile_args)
        self.opt_state = None
        self.compile_generation = getattr(self, 'compile_generation', 0) + 1

    def save(self, path):
        """weights checkpoint: a Keras-layout `.h5` (what the reference's `model.save`
        calls write, fplnetwork.py:16-17,83; see keras_io.py) when `path` ends in .h5,
        else the package's `.npz` (arrays in get_weights() order)"""
        if path.endswith('.h5'):
            from . import keras_io
            keras_io.save_weights(self, path)
        else:
            extra = {}
            if self.opt_state is not None:
                m, v, it = self.opt_state
                extra = {'opt_iterations': np.int64(it)}
                extra.update({'opt_m_%d' % i: a for i, a in enumerate(m)})
                extra.update({'opt_v_%d' % i: a for i, a in enumerate(v)})
            np.savez(path if path.endswith('.npz') else path + '.npz', *self.weights, **extra)

    def load(self, path):
        if path.endswith('.h5'):
            from . import keras_io
            keras_io.load_weights(self, path)
            return
        with np.load(path if path.endswith('.npz') else path + '.npz') as z:
            n = len([k for k in z.files if k.startswith('arr_')])
            self.set_weights([z['arr_%d' % i] for i in range(n)])
            self.opt_state = None
            if 'opt_iterations' in z.files:
                self.opt_state = ([z['opt_m_%d' % i] for i in range(n)],
                                  [z['opt_v_%d' % i] for i in range(n)], int(z['opt_iterations']))

    def randomize_bn(self, seed=1):
        """non-trivial BN statistics for synthetic benchmarks"""
        rng = np.random.default_rng(seed)
        for n in self.nodes:
            if n.kind != 'bn':
                continue
            g, b, m, v = n.weight_slots
            c = n.channels
            self.weights[g] = rng.uniform(0.5, 1.5, c).astype(np.float32)
            self.weights[b] = (0.1 * rng.standard_normal(c)).astype(np.float32)
            self.weights[m] = (0.1 * rng.standard_normal(c)).astype(np.float32)
            self.weights[v] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    # ---- lowering -----------------------------------------------------------
    def consumers(self):
        cons = {n.idx: [] for n in self.nodes}
        for n in self.nodes:
            for i in n.inputs:
                cons[i].append(n.idx)
        return cons

    def lower_inference(self):
        """Fold conv -> [bn] -> [relu] -> [dropout] chains into fused conv ops.

        Returns (ops, arena): `ops` is a list of dicts
            kind, src0, src1, dst, k, cin, cout, act, w_off, scale_off, shift_off,
            p0..p5 (pool/up factor or crop pairs)
        over tensor ids (0 = network input), `arena` the flat fp32 weights:
        conv kernel as [k^3*cin][cout] (Keras memory order), then per-channel
        `scale`, `shift` with  y = act(scale * conv(x) + shift).
        """
        cons = self.consumers()
        arena = []
        off = [0]

        def push(a):
            a = np.asarray(a, dtype=np.float32).reshape(-1)
            o = off[0]
            arena.append(a)
            off[0] += a.size
            return o

        ops = []
        tensor_of = {self.inputs_node.idx: 0}   # node idx -> tensor id
        n_tensors = [1]

        def new_tensor():
            t = n_tensors[0]
            n_tensors[0] += 1
            return t

        absorbed = set()
        for n in self.nodes:
            if n.idx in absorbed or n.kind == 'input':
                continue
            if n.kind == 'conv':
                k = n.attrs['k']
                cin = self.nodes[n.inputs[0]].channels
                cout = n.channels
                kern = self.weights[n.weight_slots[0]]
                scale = np.ones(cout, np.float32)
                shift = np.zeros(cout, np.float32)
                if n.attrs['use_bias']:
                    shift = self.weights[n.weight_slots[1]].astype(np.float32)
                act = {None: ACT_NONE, 'relu': ACT_RELU,
                       'sigmoid': ACT_SIGMOID}[n.attrs['activation']]
                last = n
                # absorb a linear chain while the producer has a single consumer
                while len(cons[last.idx]) == 1:
                    nxt = self.nodes[cons[last.idx][0]]
                    if nxt.kind == 'bn' and act == ACT_NONE:
                        g, b, m, v = (self.weights[s].astype(np.float64)
                                      for s in nxt.weight_slots)
                        s = g / np.sqrt(v + BN_EPS)
                        shift = ((shift.astype(np.float64) - m) * s
                                 + b).astype(np.float32)
                        scale = (scale.astype(np.float64) * s).astype(np.float32)
                    elif nxt.kind == 'relu' and act == ACT_NONE:
                        act = ACT_RELU
                    elif nxt.kind == 'drop':
                        pass
                    else:
                        break
                    absorbed.add(nxt.idx)
                    last = nxt
                dst = new_tensor()
                ops.append(dict(kind=OP_CONV, src0=tensor_of[n.inputs[0]],
                                src1=-1, dst=dst, k=k, cin=cin, cout=cout,
                                act=act, w_off=push(kern),
                                scale_off=push(scale), shift_off=push(shift),
                                p=(0,) * 6))
                tensor_of[n.idx] = dst
                tensor_of[last.idx] = dst
                # intermediate absorbed nodes alias the fused output too
                cur = n
                while cur.idx != last.idx:
                    cur = self.nodes[cons[cur.idx][0]]
                    tensor_of[cur.idx] = dst
                continue
            if n.kind == 'drop':
                tensor_of[n.idx] = tensor_of[n.inputs[0]]
                continue
            if n.kind in ('bn', 'relu'):
                raise NotImplementedError(
                    'stand-alone %s (node %d) is not produced by any in-scope '
                    'architecture' % (n.kind, n.idx))
            dst = new_tensor()
            src0 = tensor_of[n.inputs[0]]
            src1 = tensor_of[n.inputs[1]] if len(n.inputs) > 1 else -1
            c_in = self.nodes[n.inputs[0]].channels
            base = dict(src0=src0, src1=src1, dst=dst, k=0, cin=c_in,
                        cout=n.channels, act=ACT_NONE, w_off=0, scale_off=0,
                        shift_off=0)
            if n.kind == 'pool':
                f = n.attrs['n']
                ops.append(dict(base, kind=OP_POOL, p=(f, f, f, 0, 0, 0)))
            elif n.kind == 'up':
                ops.append(dict(base, kind=OP_UP, p=n.attrs['n'] + (0, 0, 0)))
            elif n.kind == 'crop':
                c = n.attrs['c']
                ops.append(dict(base, kind=OP_CROP,
                                p=(c[0][0], c[0][1], c[1][0], c[1][1],
                                   c[2][0], c[2][1])))
            elif n.kind == 'concat':
                ops.append(dict(base, kind=OP_CONCAT, p=(0,) * 6))
            elif n.kind == 'add':
                act = ACT_NONE
                if len(cons[n.idx]) == 1 and \
                        self.nodes[cons[n.idx][0]].kind == 'relu':
                    nxt = self.nodes[cons[n.idx][0]]   # add -> relu (resnet_like)
                    absorbed.add(nxt.idx)
                    tensor_of[nxt.idx] = dst
                    act = ACT_RELU
                ops.append(dict(base, kind=OP_ADD, act=act, p=(0,) * 6))
            else:
                raise NotImplementedError(n.kind)
            tensor_of[n.idx] = dst
        out_tensor = tensor_of[self.output.idx]
        arena = (np.concatenate(arena) if arena
                 else np.zeros(0, np.float32)).astype(np.float32)
        return ops, arena, out_tensor, n_tensors[0]


# layer kinds shared with include/fplhip.h (enum fpl_layer_kind)
L_CONV, L_BN, L_RELU, L_POOL, L_DROPOUT, L_UP, L_CROP, L_CONCAT, L_ADD = range(9)


def lower_training(graph):
    """Unfused layer list for the training engine (`fpl_trainer_create`).

    Returns (layers, arena, offsets, out_tensor, n_tensors): one dict per Keras
    layer with tensor ids (0 = input) and float offsets of its weights inside
    `arena`, the flat concatenation of `graph.weights` (Keras order);
    `offsets[i]` is where weight array i starts."""
    offsets, tot = [], 0
    for w in graph.weights:
        offsets.append(tot)
        tot += w.size
    arena = (np.concatenate([w.reshape(-1) for w in graph.weights])
             .astype(np.float32) if graph.weights else np.zeros(0, np.float32))
    tensor_of = {graph.inputs_node.idx: 0}
    layers, n_t = [], 1
    kinds = {'conv': L_CONV, 'bn': L_BN, 'relu': L_RELU, 'pool': L_POOL,
             'drop': L_DROPOUT, 'up': L_UP, 'crop': L_CROP, 'concat': L_CONCAT,
             'add': L_ADD}
    for n in graph.nodes:
        if n.kind == 'input':
            continue
        src = [tensor_of[i] for i in n.inputs]
        d = dict(kind=kinds[n.kind], src0=src[0], src1=src[1] if len(src) > 1 else -1,
                 dst=n_t, k=0, cin=graph.nodes[n.inputs[0]].channels,
                 cout=n.channels, use_bias=0, act=ACT_NONE, rate=0.0, p=(0,) * 6,
                 w_off=[0, 0, 0, 0])
        if n.kind == 'conv':
            a = n.attrs['activation']
            if a == 'relu':
                raise NotImplementedError(
                    'conv with a fused relu activation (unet_like_vol) is not '
                    'trainable yet')
            d.update(k=n.attrs['k'], use_bias=int(n.attrs['use_bias']),
                     act={None: ACT_NONE, 'sigmoid': ACT_SIGMOID}[a])
            d['w_off'][0] = offsets[n.weight_slots[0]]
            if n.attrs['use_bias']:
                d['w_off'][1] = offsets[n.weight_slots[1]]
        elif n.kind == 'bn':
            d['w_off'] = [offsets[s] for s in n.weight_slots]
        elif n.kind == 'pool':
            f = n.attrs['n']
            d['p'] = (f, f, f, 0, 0, 0)
        elif n.kind == 'up':
            d['p'] = tuple(n.attrs['n']) + (0, 0, 0)
        elif n.kind == 'crop':
            c = n.attrs['c']
            d['p'] = (c[0][0], c[0][1], c[1][0], c[1][1], c[2][0], c[2][1])
        elif n.kind == 'drop':
            d['rate'] = float(n.attrs['rate'])
        layers.append(d)
        tensor_of[n.idx] = n_t
        n_t += 1
    return layers, arena, offsets, tensor_of[graph.output.idx], n_t
