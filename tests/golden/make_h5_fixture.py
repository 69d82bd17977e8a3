"""tests/golden/keras_libhdf5.h5 (+ .npz): a Keras `model.save` file written by the REAL
HDF5 library (libhdf5 1.10 through tests/h5lib.py - h5py itself is not installed), in the
form h5py >= 3 / tf.keras give such files, so that the package's pure-Python reader
(flypylib_amd/h5min.py) is pinned by bytes it did not write:

  * `keras_version`, `backend`, `model_config`, `training_config`: variable-length UTF-8
    strings (global heap), on the root AND on `model_weights`;
  * `layer_names` / `weight_names`: fixed-length string arrays (Keras writes bytes arrays);
  * contiguous float32 weight datasets two groups deep (`conv3d_1/conv3d_1/kernel:0`);
  * an `optimizer_weights` group holding a CHUNKED dataset and a variable-length string
    array attribute - things the weight loader never touches and must not trip over.

The network is the 3-layer graph of tests/test_keras_io.py with seeded weights; the .npz
holds the arrays the file must yield, in `get_weights()` order.

    python tests/golden/make_h5_fixture.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from flypylib_amd import keras_io, synth          # noqa: E402
from flypylib_amd.program import LayerGraph       # noqa: E402
from tests import h5lib                           # noqa: E402


def tiny_graph():
    g = LayerGraph(None, seed=3)
    x = g.relu(g.bn(g.conv(g.input(), 4, 3)))
    g.finish(g.conv(x, 1, 1, use_bias=True, activation='sigmoid'))
    g.compile(loss='binary_crossentropy', optimizer='adam', metrics=['accuracy'])
    return g


def main():
    assert h5lib.available(), 'no libhdf5 to write the fixture with'
    g = tiny_graph()
    synth.synthetic_weights(g, 29)
    tree = keras_io.weight_tree(g)               # fixed-length names, contiguous datasets
    versions = {'keras_version': '2.2.4', 'backend': 'tensorflow'}      # str -> vlen strings
    tree['attrs'].update(versions)
    rng = np.random.default_rng(5)
    root = {'attrs': dict(versions,
                          model_config=json.dumps(keras_io.model_config(g)),
                          training_config=json.dumps(keras_io.training_config(g.compile_args))),
            'groups': {'model_weights': tree,
                       'optimizer_weights': {
                           'attrs': {'weight_names': np.array(['Adam/iterations:0', 'Adam/m_0:0'],
                                                              dtype=object)},
                           'datasets': {'m_0:0': (rng.normal(size=(8, 8)).astype(np.float32), (4, 4)),
                                        'iterations:0': np.int64(1234)}}}}
    h5lib.write_tree(os.path.join(HERE, 'keras_libhdf5.h5'), root)
    np.savez(os.path.join(HERE, 'keras_libhdf5.npz'), *g.get_weights())
    print('wrote keras_libhdf5.h5 (%d bytes)' % os.path.getsize(os.path.join(HERE, 'keras_libhdf5.h5')))


if __name__ == '__main__':
    main()
