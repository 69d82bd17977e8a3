"""Keras `.h5` weight interchange (reference `flypylib/fplnetwork.py:9-17,32-44,81-97`):
what `model.save(path + '.keras.h5')`, `model.save('%s_%03d.h5')` and `save_weights`
write, read into / written from a `program.LayerGraph`.

Keras layout (2.0 - 2.2, TensorFlow backend): the weight tree is the file root
(`save_weights`) or its group `model_weights` (`model.save`); attribute `layer_names`
lists every layer of `model.layers`; each layer is a group with attribute `weight_names`
(`conv3d_3/kernel:0`, `batch_normalization_3/gamma:0`, ...) naming datasets below it.

Matching layers to the graph.  `model.layers` of a functional model is ordered by depth,
not by creation, so for branched networks (resnet_like's shortcut convolutions) the
file order need not be the graph's weight order.  Keras' automatic names carry the
creation order per layer class (`conv3d_7` is the seventh Conv3D built in the session), so
layers are classed by their weight names (kernel / bias -> convolution; gamma / beta /
moving_mean / moving_variance -> batch normalisation), ordered by that number within the
class (file order when a name has none) and assigned to the graph's convolution / BN
nodes in creation order; every shape is checked.

Files are read with h5py when it is installed and with the package's own reader
(`h5min`: the HDF5 subset Keras files use) otherwise; they are written with `h5min`.
"""
import re

import numpy as np

from . import h5min

_ROLE_ORDER = {'kernel': 0, 'bias': 1, 'gamma': 0, 'beta': 1, 'moving_mean': 2,
               'moving_variance': 3}


def _text(x):
    return x.decode('utf8') if isinstance(x, (bytes, np.bytes_)) else str(x)


def open_h5(path):
    """read-only handle with the h5py subset used here (attrs, [], in, keys)"""
    try:
        import h5py
        return h5py.File(path, 'r')
    except ImportError:
        return h5min.File(path)


def read_main(path):
    """the '/main' dataset of an .h5 volume file, as the reference stores images,
    predictions, labels and masks (`fplnetwork.py:137-139`, `fplobjdetect.py:154-156`)"""
    f = open_h5(path)
    try:
        return np.asarray(f['main'][...])
    finally:
        if hasattr(f, 'close'):
            f.close()


def write_main(path, array):
    """one contiguous dataset 'main' (what `h5py.File(path).create_dataset('main',
    data=array)` writes)"""
    h5min.write(path, {'datasets': {'main': np.asarray(array)}})


def _role(weight_name):
    """'conv3d_3/kernel:0' -> 'kernel'"""
    return weight_name.split('/')[-1].split(':')[0]


def keras_layers(root):
    """[(layer name, [(role, array), ...])] for the layers that own weights, file order"""
    if 'layer_names' not in root.attrs and 'model_weights' in root:
        root = root['model_weights']
    out = []
    for layer in np.atleast_1d(root.attrs['layer_names']):
        name = _text(layer)
        grp = root[name]
        wn = grp.attrs['weight_names'] if 'weight_names' in grp.attrs else []
        items = [(_role(_text(w)), np.asarray(grp[_text(w)][...])) for w in np.atleast_1d(wn)]
        if items:
            out.append((name, items))
    return out


def _creation_order(layers):
    """stable sort by the trailing number of Keras' automatic names"""
    def key(item):
        m = re.search(r'_(\d+)$', item[1][0])
        return (0, int(m.group(1))) if m else (1, item[0])
    numbered = [re.search(r'_(\d+)$', n) is not None for n, _ in layers]
    if not all(numbered):
        return layers                      # user-named layers: keep the file's order
    return [l for _, l in sorted(enumerate(layers), key=key)]


def graph_weights_from_layers(graph, layers):
    """arrays in `graph.get_weights()` order"""
    convs = [(n, it) for n, it in layers if {r for r, _ in it} <= {'kernel', 'bias'}]
    bns = [(n, it) for n, it in layers if {r for r, _ in it} & {'gamma', 'beta', 'moving_mean'}]
    other = [n for n, it in layers if (n, it) not in convs and (n, it) not in bns]
    if other:
        raise ValueError('layers with weights this package has no counterpart for: %s' % other)
    convs, bns = _creation_order(convs), _creation_order(bns)
    out = [None] * len(graph.weights)
    ci = bi = 0
    for node in graph.nodes:
        if node.kind not in ('conv', 'bn'):
            continue
        pool, idx = (convs, ci) if node.kind == 'conv' else (bns, bi)
        if idx >= len(pool):
            raise ValueError('the file has %d %s layers, the network needs more'
                             % (len(pool), node.kind))
        name, items = pool[idx]
        items = sorted(items, key=lambda ra: _ROLE_ORDER[ra[0]])
        if len(items) != len(node.weight_slots):
            raise ValueError('%s holds %d arrays, %s_%d of the network %d' % (
                name, len(items), node.kind, node.idx, len(node.weight_slots)))
        for slot, (role, arr) in zip(node.weight_slots, items):
            want = graph.weights[slot].shape
            if tuple(arr.shape) != tuple(want):
                raise ValueError('%s/%s has shape %s, the network expects %s'
                                 % (name, role, tuple(arr.shape), tuple(want)))
            out[slot] = np.asarray(arr, np.float32)
        if node.kind == 'conv':
            ci += 1
        else:
            bi += 1
    if ci != len(convs) or bi != len(bns):
        raise ValueError('the file has %d conv / %d BN layers, the network %d / %d'
                         % (len(convs), len(bns), ci, bi))
    return out


def _trainable_slots(graph):
    """weight slots in the order of Keras' `model.trainable_weights`: layer by layer, kernel
    (, bias) / gamma, beta - the moving statistics are not trainable"""
    out = []
    for node in graph.nodes:
        if node.kind == 'conv':
            out.extend(node.weight_slots)
        elif node.kind == 'bn':
            out.extend(node.weight_slots[:2])
    return out


def optimizer_tree(graph):
    """the `optimizer_weights` group `model.save` writes for Adam (Keras 2.0 - 2.1,
    `keras/optimizers.py`: `self.weights = [self.iterations] + ms + vs`, the TensorFlow-1
    variable names of a model compiled once): iterations, then the first and the second
    moments in `trainable_weights` order.  `load_model` hands the arrays to
    `optimizer.set_weights` in the order of the `weight_names` attribute."""
    m, v, it = graph.opt_state
    slots = _trainable_slots(graph)
    names = ['Adam/iterations:0']
    arrays = [np.asarray(it, np.int64)]
    for k, src in enumerate((m, v)):
        for i, slot in enumerate(slots):
            j = k * len(slots) + i
            names.append('training/Adam/Variable%s:0' % ('_%d' % j if j else ''))
            arrays.append(np.asarray(src[slot], np.float32))
    tree = {'attrs': {'weight_names': np.array([n.encode() for n in names], dtype='S')}, 'groups': {}}
    for n, a in zip(names, arrays):
        parts = n.split('/')
        g = tree
        for part in parts[:-1]:
            g = g.setdefault('groups', {}).setdefault(part, {})
        g.setdefault('datasets', {})[parts[-1]] = a
    return tree


def optimizer_state_from_file(graph, root):
    """(m, v, iterations) in get_weights() order from a whole-model file's `optimizer_weights`
    (Adam: iterations + two arrays per trainable weight), or None"""
    if 'optimizer_weights' not in root:
        return None
    grp = root['optimizer_weights']
    if 'weight_names' not in grp.attrs:
        return None
    names = [_text(w) for w in np.atleast_1d(grp.attrs['weight_names'])]
    slots = _trainable_slots(graph)
    if len(names) < 1 + 2 * len(slots):
        return None                         # another optimizer's state
    try:
        arrays = [np.asarray(grp[n][...]) for n in names]
    except (KeyError, NotImplementedError, ValueError):
        return None                         # a layout this reader does not follow: Adam starts afresh
    it = int(np.asarray(arrays[0]).reshape(-1)[0])
    m = [np.zeros_like(w) for w in graph.weights]
    v = [np.zeros_like(w) for w in graph.weights]
    for i, slot in enumerate(slots):
        a, b = arrays[1 + i], arrays[1 + len(slots) + i]
        if a.shape != graph.weights[slot].shape or b.shape != graph.weights[slot].shape:
            return None
        m[slot], v[slot] = a.astype(np.float32), b.astype(np.float32)
    return m, v, it


def load_weights(graph, path):
    """set `graph`'s weights - and, from a whole-model file that carries it, its optimizer
    state - from a Keras .h5 (weights-only or whole-model file)"""
    f = open_h5(path)
    try:
        graph.set_weights(graph_weights_from_layers(graph, keras_layers(f)))
        graph.opt_state = optimizer_state_from_file(graph, f)
    finally:
        if hasattr(f, 'close'):
            f.close()


def weight_tree(graph):
    """the `save_weights` tree of a graph (h5min.write's input): Keras' automatic layer
    names in creation order, one group per layer of the graph (weightless ones included,
    as Keras lists them)"""
    roles = {'conv': ['kernel', 'bias'], 'bn': ['gamma', 'beta', 'moving_mean',
                                                'moving_variance']}
    names = _layer_names(graph)
    layer_names, groups = [], {}
    for node in graph.nodes:
        name = names[node.idx]
        layer_names.append(name.encode())
        wn, ds = [], {}
        for role, slot in zip(roles.get(node.kind, []), node.weight_slots):
            wn.append(('%s/%s:0' % (name, role)).encode())
            ds['%s:0' % role] = np.asarray(graph.weights[slot], np.float32)
        sub = {'attrs': {'weight_names': np.array(wn, dtype='S') if wn else np.zeros(0, 'S1')}}
        if ds:
            sub['groups'] = {name: {'datasets': ds}}
        groups[name] = sub
    return {'attrs': {'layer_names': np.array(layer_names, dtype='S'),
                      'backend': np.bytes_(b'tensorflow'),
                      'keras_version': np.bytes_(b'2.0.8')},
            'groups': groups}


_KERAS_CLASS = {'conv': ('Conv3D', 'conv3d'), 'bn': ('BatchNormalization', 'batch_normalization'),
                'relu': ('Activation', 'activation'), 'pool': ('MaxPooling3D', 'max_pooling3d'),
                'drop': ('Dropout', 'dropout'), 'up': ('UpSampling3D', 'up_sampling3d'),
                'crop': ('Cropping3D', 'cropping3d'), 'concat': ('Concatenate', 'concatenate'),
                'add': ('Add', 'add'), 'input': ('InputLayer', 'input')}


def _layer_names(graph):
    """Keras' automatic layer names, per class in creation order (node index -> name)"""
    counts, names = {}, {}
    for node in graph.nodes:
        base = _KERAS_CLASS[node.kind][1]
        counts[base] = counts.get(base, 0) + 1
        names[node.idx] = '%s_%d' % (base, counts[base])
    return names


def model_config(graph):
    """`{'class_name': 'Model', 'config': model.get_config()}` as Keras 2.0 - 2.1 writes it
    into the `model_config` attribute of `model.save` files (`keras/engine/topology.py`
    `Container.get_config`, the layers' `get_config`): what the reference's
    `load_model(path + '.keras.h5', custom_objects=...)` (`fplnetwork.py:36-40`) rebuilds
    the network from.  Layer arguments are those `flypylib/fplmodels.py:67-526` passes,
    everything else the Keras defaults of that era."""
    names = _layer_names(graph)
    zeros = {'class_name': 'Zeros', 'config': {}}
    ones = {'class_name': 'Ones', 'config': {}}
    glorot = {'class_name': 'VarianceScaling',
              'config': {'scale': 1.0, 'mode': 'fan_avg', 'distribution': 'uniform', 'seed': None}}
    layers = []
    for node in graph.nodes:
        name, a = names[node.idx], node.attrs
        cfg = {'name': name, 'trainable': True}
        if node.kind == 'input':
            spatial = list(graph.in_sz) if graph.in_sz is not None else [None, None, None]
            cfg = {'batch_input_shape': [None] + spatial + [1], 'dtype': 'float32',
                   'sparse': False, 'name': name}
        elif node.kind == 'conv':
            k = int(a['k'])
            cfg.update(filters=int(node.channels), kernel_size=[k, k, k], strides=[1, 1, 1],
                       padding='valid', data_format='channels_last', dilation_rate=[1, 1, 1],
                       activation=a['activation'] or 'linear', use_bias=bool(a['use_bias']),
                       kernel_initializer=glorot, bias_initializer=zeros,
                       kernel_regularizer=None, bias_regularizer=None,
                       activity_regularizer=None, kernel_constraint=None, bias_constraint=None)
        elif node.kind == 'bn':
            cfg.update(axis=-1, momentum=0.99, epsilon=0.001, center=True, scale=True,
                       beta_initializer=zeros, gamma_initializer=ones,
                       moving_mean_initializer=zeros, moving_variance_initializer=ones,
                       beta_regularizer=None, gamma_regularizer=None, beta_constraint=None,
                       gamma_constraint=None)
        elif node.kind == 'relu':
            cfg.update(activation='relu')
        elif node.kind == 'pool':
            n = int(a['n'])
            cfg.update(pool_size=[n, n, n], padding='valid', strides=[n, n, n],
                       data_format='channels_last')
        elif node.kind == 'drop':
            cfg.update(rate=float(a['rate']), noise_shape=None, seed=None)
        elif node.kind == 'up':
            cfg.update(size=[int(f) for f in a['n']], data_format='channels_last')
        elif node.kind == 'crop':
            cfg.update(cropping=[[int(lo), int(hi)] for lo, hi in a['c']],
                       data_format='channels_last')
        elif node.kind == 'concat':
            cfg.update(axis=-1)
        inbound = [[[names[i], 0, 0, {}] for i in node.inputs]] if node.inputs else []
        layers.append({'name': name, 'class_name': _KERAS_CLASS[node.kind][0], 'config': cfg,
                       'inbound_nodes': inbound})
    return {'class_name': 'Model',
            'config': {'name': 'model_1', 'layers': layers,
                       'input_layers': [[names[graph.inputs_node.idx], 0, 0]],
                       'output_layers': [[names[graph.output.idx], 0, 0]]}}


def training_config(compile_args):
    """the `training_config` attribute of a `model.save` file (`keras/models.py`
    `save_model`): optimizer class + hyper-parameters (as float32 values, which is what
    `K.get_value` returns), loss and metrics by name.  The reference compiles with
    `'adam'` and either Keras' `binary_crossentropy` or one of its own masked losses
    (`fplnetwork.py:74-77`, `fplmodels.py:300`); functions are stored by `__name__`."""
    ca = dict(compile_args or {})

    def name(x):
        return x if isinstance(x, str) else getattr(x, '__name__', str(x))
    f32 = lambda v: float(np.float32(v))            # noqa: E731
    opt = ca.get('optimizer', 'adam')
    if name(opt).lower() != 'adam':
        raise ValueError('training_config: optimizer %r (the engine trains with Adam)' % (opt,))
    return {'optimizer_config': {'class_name': 'Adam',
                                 'config': {'lr': f32(1e-3), 'beta_1': f32(0.9),
                                            'beta_2': f32(0.999), 'epsilon': 1e-08, 'decay': 0.0}},
            'loss': name(ca.get('loss', 'binary_crossentropy')),
            'metrics': [name(m) for m in ca.get('metrics', [])],
            'sample_weight_mode': None, 'loss_weights': None}


def save_weights(graph, path, as_model_save=True):
    """write `graph`'s weights as a Keras .h5.  as_model_save: the file `model.save`
    writes (`fplnetwork.py:16-17,83`) - the weights under the group 'model_weights' and the
    root attributes `model_config` / `training_config` / `keras_version` / `backend`, so
    that the reference's `load_model(path, custom_objects)` can rebuild and load the
    network; otherwise the bare `save_weights` layout.  A graph that has been trained also
    writes Adam's state as `optimizer_weights` (optimizer_tree), as `model.save` does."""
    import json
    tree = weight_tree(graph)
    if as_model_save:
        attrs = {k: tree['attrs'].pop(k) for k in ('backend', 'keras_version')}
        tree = {'attrs': dict(attrs), 'groups': {'model_weights': tree}}
        tree['groups']['model_weights']['attrs'].update(attrs)
        tree['attrs']['model_config'] = np.bytes_(json.dumps(model_config(graph)).encode('utf8'))
        tree['attrs']['training_config'] = np.bytes_(
            json.dumps(training_config(graph.compile_args)).encode('utf8'))
        if getattr(graph, 'opt_state', None) is not None:
            tree['groups']['optimizer_weights'] = optimizer_tree(graph)
    h5min.write(path, tree)
