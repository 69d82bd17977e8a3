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


def load_weights(graph, path):
    """set `graph`'s weights from a Keras .h5 (weights-only or whole-model file)"""
    f = open_h5(path)
    try:
        graph.set_weights(graph_weights_from_layers(graph, keras_layers(f)))
    finally:
        if hasattr(f, 'close'):
            f.close()


def weight_tree(graph):
    """the `save_weights` tree of a graph (h5min.write's input): Keras' automatic layer
    names in creation order, one group per layer of the graph (weightless ones included,
    as Keras lists them)"""
    counts = {}
    cls = {'conv': 'conv3d', 'bn': 'batch_normalization', 'relu': 'activation',
           'pool': 'max_pooling3d', 'drop': 'dropout', 'up': 'up_sampling3d',
           'crop': 'cropping3d', 'concat': 'concatenate', 'add': 'add', 'input': 'input'}
    roles = {'conv': ['kernel', 'bias'], 'bn': ['gamma', 'beta', 'moving_mean',
                                                'moving_variance']}
    layer_names, groups = [], {}
    for node in graph.nodes:
        base = cls[node.kind]
        counts[base] = counts.get(base, 0) + 1
        name = '%s_%d' % (base, counts[base])
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


def save_weights(graph, path, as_model_save=True):
    """write `graph`'s weights as a Keras .h5.  as_model_save: under the group
    'model_weights', where `model.save` puts them (Keras' `load_weights` accepts both
    layouts); no `model_config` is written - on the Keras side rebuild the network from
    its factory and call `load_weights`."""
    tree = weight_tree(graph)
    if as_model_save:
        attrs = {k: tree['attrs'].pop(k) for k in ('backend', 'keras_version')}
        tree = {'attrs': attrs, 'groups': {'model_weights': tree}}
        tree['groups']['model_weights']['attrs'].update(attrs)
    h5min.write(path, tree)
