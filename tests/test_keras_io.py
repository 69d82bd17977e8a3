"""Keras `.h5` interchange on the CPU: the package's HDF5 subset reader / writer (h5min),
the layer matching of keras_io and the reference-pickle unpickler.

h5py and Keras are absent from the build container, the HDF5 C library is not
(tests/h5lib.py binds it with ctypes), so both directions are pinned by the real library:
libhdf5 opens and reads what the writer writes (`model.save` layout incl. `model_config`),
and the reader reads tests/golden/keras_libhdf5.h5, which libhdf5 wrote in the form h5py >=
3 gives Keras files (variable-length string attributes, a chunked optimizer dataset next to
the weights).  Plus: the writer's committed bytes (tests/golden/keras_tiny.h5) and
hand-assembled structures neither writer emits (continuation blocks, compact layout,
version-2 dataspace / version-3 attribute messages)."""
import io
import os
import pickle
import struct
import sys
import types

import numpy as np
import pytest

from flypylib_amd import fplmodels, h5min, keras_io, synth
from flypylib_amd.program import LayerGraph
from tests import h5lib

needs_libhdf5 = pytest.mark.skipif(not h5lib.available(), reason='no libhdf5 to load')

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _tiny_graph():
    g = LayerGraph(None, seed=3)
    x = g.relu(g.bn(g.conv(g.input(), 4, 3)))
    return g.finish(g.conv(x, 1, 1, use_bias=True, activation='sigmoid'))


def test_h5min_round_trip_types_groups_attributes():
    rng = np.random.default_rng(0)
    tree = {'attrs': {'names': np.array([b'a', b'longer_name']), 'n': np.int64(-7),
                      'scalar_str': np.bytes_(b'tensorflow'), 'empty': np.zeros(0, 'S1'),
                      'f': np.array([1.5, -2.25], np.float64)},
            'datasets': {'f32': rng.normal(size=(3, 1, 5)).astype(np.float32),
                         'f64': rng.normal(size=(4,)), 'u8': np.arange(7, dtype=np.uint8),
                         'i32': np.array(-5, np.int32), 'f16': np.array([0.5, 2], np.float16)},
            'groups': {'g1': {'groups': {'deep': {'datasets': {'k:0': np.ones((2, 2), np.float32)}}},
                              'attrs': {'weight_names': np.array([b'deep/k:0'])}},
                       'g0': {}}}
    f = h5min.File(h5min.to_bytes(tree))
    assert f.keys() == ['f16', 'f32', 'f64', 'g0', 'g1', 'i32', 'u8']
    for k, v in tree['datasets'].items():
        got = f[k][...]
        assert got.dtype == v.dtype and got.shape == np.shape(v) and np.array_equal(got, v), k
    assert f['i32'][()] == -5 and f['i32'].shape == ()
    assert list(f.attrs['names']) == [b'a', b'longer_name'] and f.attrs['n'] == -7
    assert f.attrs['scalar_str'] == b'tensorflow' and f.attrs['empty'].shape == (0,)
    assert np.array_equal(f.attrs['f'], [1.5, -2.25])
    assert np.array_equal(f['g1/deep/k:0'][...], np.ones((2, 2)))
    assert f['g1'].attrs['weight_names'][0] == b'deep/k:0' and f['g0'].keys() == []
    assert 'g1' in f and 'nope' not in f and 'g1/deep' in f
    with pytest.raises(KeyError):
        f['g1/nope']
    # groups of any size: the writer raises the file's leaf K
    big = {'datasets': {'d%03d' % i: np.float32(i) for i in range(300)}}
    fb = h5min.File(h5min.to_bytes(big))
    assert len(fb.keys()) == 300 and fb['d299'][()] == 299


def test_h5min_reads_structures_its_writer_does_not_emit():
    """object-header continuation block, compact layout, dataspace v2, attribute v3"""
    w = h5min._Writer(4)
    # dataset with a compact layout and a version-2 dataspace; its attribute sits in a
    # continuation block
    data = np.arange(6, dtype=np.float32).reshape(2, 3)
    space_v2 = struct.pack('<BBBB', 2, 2, 0, 1) + struct.pack('<QQ', 2, 3)
    layout = struct.pack('<BBH', 3, 0, data.nbytes) + data.tobytes()
    nm = b'unit\0'
    dt, ds = h5min._dtype_msg('S2'), struct.pack('<BBBB', 2, 0, 0, 0)
    attr_v3 = struct.pack('<BBHHHB', 3, 0, len(nm), len(dt), len(ds), 0) + nm + dt + ds + b'mm'
    cont_block = w.alloc(h5min._msg(0x000C, attr_v3))
    cont_len = len(h5min._msg(0x000C, attr_v3))
    hdr = w.header([h5min._msg(0x0001, space_v2), h5min._msg(0x0003, h5min._dtype_msg(np.float32)),
                    h5min._msg(0x0008, layout),
                    h5min._msg(0x0010, struct.pack('<QQ', cont_block, cont_len)),
                    b''])
    # the header announces one message more than its first block holds (the attribute)
    w.buf[hdr + 2:hdr + 4] = struct.pack('<H', 5)
    # a root group holding it: reuse the writer's group code for the links
    tree_bytes = bytearray(h5min.to_bytes({'datasets': {'x': np.float32(0)}}))
    base = h5min.File(bytes(tree_bytes))
    # splice: append our objects after the small file and repoint the link 'x'
    off = len(tree_bytes)
    blob = bytes(w.buf[96:])
    shift = off - 96
    patched = bytearray(tree_bytes + blob)
    # addresses inside the blob move by `shift`
    new_hdr = hdr + shift
    cm = new_hdr + 16 + len(h5min._msg(0x0001, space_v2)) + len(h5min._msg(0x0003, h5min._dtype_msg(np.float32))) \
        + len(h5min._msg(0x0008, layout)) + 8
    patched[cm:cm + 8] = struct.pack('<Q', cont_block + shift)
    base._load()
    old = base._links['x']
    pos = bytes(patched).find(struct.pack('<Q', old), 96)
    patched[pos:pos + 8] = struct.pack('<Q', new_hdr)
    patched[40:48] = struct.pack('<Q', len(patched))           # end-of-file address
    f = h5min.File(bytes(patched))
    assert np.array_equal(f['x'][...], data) and f['x'].attrs['unit'] == b'mm'


def test_h5min_names_what_it_cannot_read():
    with pytest.raises(h5min.H5Unsupported, match='not an HDF5 file'):
        h5min.File(b'\0' * 200)
    b = bytearray(h5min.to_bytes({'datasets': {'x': np.float32(1)}}))
    b[8] = 2
    with pytest.raises(h5min.H5Unsupported, match='superblock version 2'):
        h5min.File(bytes(b))


def test_committed_fixture_reads_back(tmp_path):
    """tests/golden/keras_tiny.h5: bytes of the writer for a 3-layer network, committed;
    tests/golden/keras_tiny.npz: the arrays it must yield"""
    want = np.load(os.path.join(GOLDEN, 'keras_tiny.npz'))
    g = _tiny_graph()
    keras_io.load_weights(g, os.path.join(GOLDEN, 'keras_tiny.h5'))
    got = g.get_weights()
    assert len(got) == len(want.files) == 7
    for i, a in enumerate(got):
        assert np.array_equal(a, want['arr_%d' % i])
    f = h5min.File(os.path.join(GOLDEN, 'keras_tiny.h5'))
    assert [n.decode() for n in f['model_weights'].attrs['layer_names']] == [
        'input_1', 'conv3d_1', 'batch_normalization_1', 'activation_1', 'conv3d_2']
    # the writer is deterministic: re-writing the same weights gives the committed bytes
    g.save(str(tmp_path / 'again.h5'))
    assert open(str(tmp_path / 'again.h5'), 'rb').read() == \
        open(os.path.join(GOLDEN, 'keras_tiny.h5'), 'rb').read()


def test_reader_on_a_file_libhdf5_wrote():
    """tests/golden/keras_libhdf5.h5 (tests/golden/make_h5_fixture.py): written by the C
    library, h5py-3 style - the weights load, the variable-length strings resolve through
    the global heap, and what the loader does not need (a chunked dataset) is refused only
    when it is itself asked for"""
    import json
    path = os.path.join(GOLDEN, 'keras_libhdf5.h5')
    want = np.load(os.path.join(GOLDEN, 'keras_libhdf5.npz'))
    g = _tiny_graph()
    g.load(path)                                   # through keras_io.load_weights -> h5min
    for i, a in enumerate(g.get_weights()):
        assert np.array_equal(a, want['arr_%d' % i])
    f = h5min.File(path)
    assert f.attrs['keras_version'] == b'2.2.4' and f.attrs['backend'] == b'tensorflow'
    assert f['model_weights'].attrs['backend'] == b'tensorflow'
    cfg = json.loads(f.attrs['model_config'].decode('utf8'))
    assert cfg == keras_io.model_config(g)
    assert json.loads(f.attrs['training_config'].decode('utf8'))['loss'] == 'binary_crossentropy'
    assert sorted(f.attrs.keys()) == ['backend', 'keras_version', 'model_config', 'training_config']
    ow = f['optimizer_weights']
    assert list(ow.attrs['weight_names']) == [b'Adam/iterations:0', b'Adam/m_0:0']
    assert ow['iterations:0'][()] == 1234
    with pytest.raises(h5min.H5Unsupported, match='chunked'):
        ow['m_0:0'][...]


@needs_libhdf5
@pytest.mark.parametrize('name', ['vgg_like', 'unet_like2'])
def test_libhdf5_reads_what_the_writer_writes(tmp_path, name):
    """`save_network`'s `<path>.keras.h5` through the C library - what h5py / Keras'
    `load_model` see: every group, attribute and dataset, bit for bit"""
    import json
    g = getattr(fplmodels, name)()[0]
    synth.synthetic_weights(g, 13)
    g.compile(loss='masked_focal_loss', optimizer='adam', metrics=['masked_accuracy'])
    p = str(tmp_path / (name + '.keras.h5'))
    g.save(p)
    t = h5lib.read_tree(p)
    assert t['attrs']['backend'] == b'tensorflow' and t['attrs']['keras_version'] == b'2.0.8'
    cfg = json.loads(t['attrs']['model_config'].decode('utf8'))
    assert cfg['class_name'] == 'Model' and len(cfg['config']['layers']) == len(g.nodes)
    tc = json.loads(t['attrs']['training_config'].decode('utf8'))
    assert tc['loss'] == 'masked_focal_loss' and tc['optimizer_config']['class_name'] == 'Adam'
    mw = t['groups']['model_weights']
    layer_names = [n.decode() for n in mw['attrs']['layer_names']]
    assert layer_names == [l['name'] for l in cfg['config']['layers']]
    got = []
    for ln in layer_names:
        for wn in mw['groups'][ln]['attrs']['weight_names']:
            outer, ds = wn.decode().split('/')
            got.append((outer, ds, mw['groups'][ln]['groups'][outer]['datasets'][ds]))
    assert len(got) == len(g.weights)
    # every array is there, exact; the graph rebuilt from the file matches
    g2 = getattr(fplmodels, name)()[0]
    g2.load(p)
    for a, b in zip(g.get_weights(), g2.get_weights()):
        assert np.array_equal(a, b)
    flat = {(o, d): a for o, d, a in got}
    names = keras_io._layer_names(g)
    roles = {'conv': ['kernel', 'bias'], 'bn': ['gamma', 'beta', 'moving_mean', 'moving_variance']}
    for node in g.nodes:
        for role, slot in zip(roles.get(node.kind, []), node.weight_slots):
            assert np.array_equal(flat[(names[node.idx], role + ':0')], g.weights[slot])


@needs_libhdf5
def test_libhdf5_reads_volume_files_and_the_committed_writer_bytes(tmp_path):
    vol = (np.arange(5 * 6 * 7) % 251).astype(np.uint8).reshape(5, 6, 7)
    p = str(tmp_path / 'v.h5')
    keras_io.write_main(p, vol)
    assert np.array_equal(h5lib.read_tree(p)['datasets']['main'], vol)
    t = h5lib.read_tree(os.path.join(GOLDEN, 'keras_tiny.h5'))
    assert sorted(t['groups']['model_weights']['groups']) == [
        'activation_1', 'batch_normalization_1', 'conv3d_1', 'conv3d_2', 'input_1']


def test_model_config_names_the_reference_layers():
    """`model_config` of vgg_like: the layer list of `flypylib/fplmodels.py:102-136` with
    Keras' automatic names, each layer fed by its predecessor"""
    g = fplmodels.vgg_like()[0]
    cfg = keras_io.model_config(g)['config']
    kinds = [l['class_name'] for l in cfg['layers']]
    assert kinds[:8] == ['InputLayer', 'Conv3D', 'BatchNormalization', 'Activation', 'Conv3D',
                         'BatchNormalization', 'Activation', 'MaxPooling3D']
    assert kinds.count('Conv3D') == 8 and kinds.count('Dropout') == 2
    convs = [l['config'] for l in cfg['layers'] if l['class_name'] == 'Conv3D']
    assert [c['filters'] for c in convs] == [48, 48, 48, 48, 48, 96, 96, 1]
    assert [c['kernel_size'][0] for c in convs] == [3, 1, 3, 1, 3, 1, 1, 1]
    assert [c['use_bias'] for c in convs] == [False] * 7 + [True]
    assert convs[-1]['activation'] == 'sigmoid' and convs[0]['activation'] == 'linear'
    assert cfg['layers'][0]['config']['batch_input_shape'] == [None, None, None, None, 1]
    for prev, layer in zip(cfg['layers'], cfg['layers'][1:]):
        assert layer['inbound_nodes'] == [[[prev['name'], 0, 0, {}]]]
    assert cfg['input_layers'] == [['input_1', 0, 0]] and cfg['output_layers'] == [['conv3d_8', 0, 0]]
    # a branched graph: the concatenate of unet_like2 names both producers, upsampled first
    u = keras_io.model_config(fplmodels.unet_like2()[0])['config']
    cat = [l for l in u['layers'] if l['class_name'] == 'Concatenate'][0]
    assert [i[0] for i in cat['inbound_nodes'][0]] == ['up_sampling3d_1', 'activation_4']
    crop = [l for l in u['layers'] if l['class_name'] == 'Cropping3D'][0]
    assert crop['config']['cropping'] == [[6, 6]] * 3


def test_attributes_are_parsed_lazily():
    """an attribute of a type the reader does not know must not make the file, the group
    or its other attributes unreadable"""
    b = bytearray(h5min.to_bytes({'attrs': {'good': np.int32(5), 'odd': np.float32(1.5)},
                                  'datasets': {'x': np.arange(3, dtype=np.float32)}}))
    # turn the datatype class of 'odd' (IEEE float, class 1) into an unknown one (class 7)
    pos = bytes(b).find(b'odd\0')
    assert pos > 0
    dt = pos + 8                                    # name padded to 8 bytes
    assert b[dt] == 0x11
    b[dt] = 0x17
    f = h5min.File(bytes(b))
    assert f.attrs['good'] == 5 and 'odd' in f.attrs and np.array_equal(f['x'][...], [0, 1, 2])
    with pytest.raises(h5min.H5Unsupported, match='datatype class 7'):
        f.attrs['odd']


@pytest.mark.parametrize('name', ['vgg_like', 'unet_like2', 'resnet_like'])
def test_graph_round_trip_through_keras_layout(tmp_path, name):
    g = getattr(fplmodels, name)()[0]
    synth.synthetic_weights(g, 11)
    for model_save in (True, False):
        p = str(tmp_path / ('%s_%d.h5' % (name, model_save)))
        keras_io.save_weights(g, p, as_model_save=model_save)
        g2 = getattr(fplmodels, name)()[0]
        g2.load(p)
        for a, b in zip(g.get_weights(), g2.get_weights()):
            assert np.array_equal(a, b)
    f = h5min.File(p)
    assert b'conv3d_1' in list(f.attrs['layer_names'])


def test_layers_are_matched_by_creation_number_not_file_order():
    """`model.layers` of a branched Keras model is depth-ordered, and a second model of
    a session continues the name counters: conv3d_12 ... - the matching must not care"""
    g = fplmodels.resnet_like()[0]
    synth.synthetic_weights(g, 5)
    layers = keras_io.keras_layers(h5min.File(h5min.to_bytes(keras_io.weight_tree(g))))
    # renumber from 12 / 7 and shuffle the file order
    renamed = []
    for name, items in layers:
        base, num = name.rsplit('_', 1)
        renamed.append(('%s_%d' % (base, int(num) + (11 if base == 'conv3d' else 6)), items))
    rng = np.random.default_rng(0)
    shuffled = [renamed[i] for i in rng.permutation(len(renamed))]
    assert [n for n, _ in shuffled] != [n for n, _ in renamed]
    got = keras_io.graph_weights_from_layers(g, shuffled)
    for a, b in zip(got, g.get_weights()):
        assert np.array_equal(a, b)
    # wrong architecture: named error, nothing assigned
    with pytest.raises(ValueError, match='shape|layers'):
        keras_io.graph_weights_from_layers(fplmodels.vgg_like()[0], shuffled)


def test_reference_pickle_classes_map_onto_this_package():
    """a pickle written by the reference names flypylib.fplnetwork.FplNetwork and
    flypylib.fplmodels.<factory> / loss functions; none of those modules exists here"""
    from flypylib_amd import fplnetwork
    mods = {}
    try:
        for m in ('flypylib', 'flypylib.fplnetwork', 'flypylib.fplmodels'):
            mods[m] = types.ModuleType(m)
            sys.modules[m] = mods[m]

        class FplNetwork:                       # stands in for the reference's class
            pass
        FplNetwork.__module__ = 'flypylib.fplnetwork'
        FplNetwork.__qualname__ = 'FplNetwork'
        mods['flypylib.fplnetwork'].FplNetwork = FplNetwork

        def unet_like2(in_sz=24):
            raise AssertionError('the reference factory must not run')

        def masked_focal_loss(y_true, y_pred):
            raise AssertionError
        for fn in (unet_like2, masked_focal_loss):
            fn.__module__ = 'flypylib.fplmodels'
            fn.__qualname__ = fn.__name__
            setattr(mods['flypylib.fplmodels'], fn.__name__, fn)
        ref = FplNetwork()
        # the attributes the reference's __init__ / save_network leave (fplnetwork.py:54-97)
        ref.__dict__.update(model=unet_like2, rf_size=(24, 24, 24), rf_offset=(9, 9, 9),
                            rf_stride=(1, 1, 1), infer_sz=(100, 100, 100), n_gpu=1,
                            compile_args={'loss': masked_focal_loss, 'optimizer': 'adam',
                                          'metrics': ['accuracy']},
                            train_single=None, train_network=None, infer_network=None)
        blob = pickle.dumps(ref)
    finally:
        for m in mods:
            sys.modules.pop(m, None)
    assert b'flypylib.fplnetwork' in blob and b'flypylib_amd' not in blob
    with pytest.raises(ModuleNotFoundError):
        pickle.loads(blob)
    net = fplnetwork._ReferenceUnpickler(io.BytesIO(blob)).load()
    assert type(net) is fplnetwork.FplNetwork
    assert net.model is fplmodels.unet_like2
    assert net.compile_args['loss'] == fplmodels.masked_focal_loss == 'masked_focal_loss'
    assert net.rf_offset == (9, 9, 9) and net.train_single is None


def test_main_dataset_volume_files(tmp_path):
    """'.h5' volumes with one dataset 'main' - how the reference stores images,
    predictions, labels and masks - through _load_main / write_labels_mask"""
    from flypylib_amd import fplobjdetect, fplsynapses
    vol = (np.arange(4 * 5 * 6) % 251).astype(np.uint8).reshape(4, 5, 6)
    p = str(tmp_path / 'v.h5')
    keras_io.write_main(p, vol)
    assert np.array_equal(keras_io.read_main(p), vol)
    assert np.array_equal(fplobjdetect._load_main(p), vol)
    tb = {'locs': np.array([[12, 14, 16]]), 'conf': np.ones(1)}
    labels, mask = fplsynapses.write_labels_mask(tb, np.ones((36, 38, 40), 'uint8'), 3, 6, 4,
                                                 str(tmp_path / 'lm'))
    assert np.array_equal(keras_io.read_main(str(tmp_path / 'lm_labels.h5')), labels)
    assert np.array_equal(keras_io.read_main(str(tmp_path / 'lm_mask.h5')), mask)
    assert labels.sum() == 123            # set_filter(3): 123 voxels


def _graph_with_opt_state(seed=5):
    g = fplmodels.vgg_like(30)[0]
    synth.synthetic_weights(g, seed)
    g.compile(loss='binary_crossentropy', optimizer='adam', metrics=['accuracy'])
    rng = np.random.default_rng(seed)
    m = [rng.standard_normal(w.shape).astype(np.float32) * 1e-3 for w in g.weights]
    v = [rng.random(w.shape).astype(np.float32) * 1e-6 for w in g.weights]
    for node in g.nodes:                      # the moving statistics have no optimizer state
        if node.kind == 'bn':
            for slot in node.weight_slots[2:]:
                m[slot][...] = 0
                v[slot][...] = 0
    g.opt_state = (m, v, 1234)
    return g


def test_optimizer_state_round_trips_through_the_keras_file_and_the_npz(tmp_path):
    """Keras' model.save keeps the optimizer (`optimizer_weights`: iterations, then Adam's
    first and second moments in trainable_weights order) and load_model restores it - the
    reference saves and loads networks through both (fplnetwork.py:9-17,32-44,81-97)"""
    g = _graph_with_opt_state()
    for name in ('w.h5', 'w.npz'):
        p = str(tmp_path / name)
        g.save(p)
        h = fplmodels.vgg_like(30)[0]
        assert h.opt_state is None
        h.load(p)
        assert all(np.array_equal(a, b) for a, b in zip(h.get_weights(), g.get_weights()))
        m, v, it = h.opt_state
        assert it == 1234
        assert all(np.array_equal(a, b) for a, b in zip(m, g.opt_state[0]))
        assert all(np.array_equal(a, b) for a, b in zip(v, g.opt_state[1]))
    # a graph that was never trained writes no optimizer group and loads without one
    g.opt_state = None
    g.save(str(tmp_path / 'plain.h5'))
    h.load(str(tmp_path / 'plain.h5'))
    assert h.opt_state is None


@needs_libhdf5
def test_libhdf5_reads_the_optimizer_group(tmp_path):
    """the group as Keras 2.0 - 2.1 lays it out: `weight_names` lists 'Adam/iterations:0' and
    'training/Adam/Variable[_k]:0' - nested groups - iterations first, then ms, then vs"""
    g = _graph_with_opt_state(6)
    p = str(tmp_path / 'm.h5')
    g.save(p)
    t = h5lib.read_tree(p)
    ow = t['groups']['optimizer_weights']
    names = [n.decode() if isinstance(n, bytes) else str(n) for n in np.atleast_1d(ow['attrs']['weight_names'])]
    n_train = sum(2 if n.kind == 'bn' else len(n.weight_slots) for n in g.nodes if n.kind in ('conv', 'bn'))
    assert names[0] == 'Adam/iterations:0' and len(names) == 1 + 2 * n_train
    assert names[1] == 'training/Adam/Variable:0' and names[2] == 'training/Adam/Variable_1:0'
    assert int(np.asarray(ow['groups']['Adam']['datasets']['iterations:0'])) == 1234
    adam = ow['groups']['training']['groups']['Adam']['datasets']
    first_kernel = g.opt_state[0][g.nodes[1].weight_slots[0]]
    assert np.array_equal(adam['Variable:0'], first_kernel)
    assert np.array_equal(adam['Variable_%d:0' % n_train], g.opt_state[1][g.nodes[1].weight_slots[0]])
