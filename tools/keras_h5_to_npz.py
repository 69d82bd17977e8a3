"""Convert the weights of a Keras 2.x `.h5` file (what the reference's
`FplNetwork.save_network` writes next to its pickle, flypylib/fplnetwork.py:81-97) into
the `.npz` checkpoint `flypylib_amd` loads (`LayerGraph.load`, `fplnetwork.load_network`).

Run it where h5py is installed (this image has none, which is why the package itself
never touches `.h5`):

    python tools/keras_h5_to_npz.py model.h5 model.p.weights.npz [--model vgg_like]

The arrays are written in Keras `get_weights()` order - layers in `layer_names` order,
each layer's `weight_names` in order - which is the order `LayerGraph.set_weights`
expects (layer creation order; conv kernel (kd,kh,kw,Cin,Cout); BN gamma, beta,
moving_mean, moving_variance).  With `--model` the shapes are checked against that
factory's graph before anything is written.
"""
import argparse
import os
import sys

import numpy as np


def _text(x):
    return x.decode('utf8') if isinstance(x, bytes) else str(x)


def keras_weight_list(root):
    """[(name, ndarray)] of an open Keras weight tree: `root` is the h5py.File, or its
    'model_weights' group when the file holds a whole model (`model.save`)"""
    if 'model_weights' in root:
        root = root['model_weights']
    out = []
    for layer in root.attrs['layer_names']:
        grp = root[_text(layer)]
        for wname in grp.attrs['weight_names']:
            out.append((_text(wname), np.asarray(grp[_text(wname)])))
    return out


def check_against(weights, factory_name):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from flypylib_amd import fplmodels
    graph = getattr(fplmodels, factory_name)()[0]
    want = [w.shape for w in graph.get_weights()]
    got = [tuple(a.shape) for _, a in weights]
    if [tuple(s) for s in want] != got:
        raise SystemExit('weights do not fit %s:\n  file  %s\n  graph %s' % (factory_name, got, want))


def main():
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('h5')
    ap.add_argument('npz')
    ap.add_argument('--model', default=None, help='fplmodels factory to check the shapes against')
    args = ap.parse_args()
    try:
        import h5py
    except ImportError:
        raise SystemExit('this converter needs h5py (pip install h5py) - run it where Keras files are read')
    with h5py.File(args.h5, 'r') as f:
        weights = keras_weight_list(f)
    if args.model:
        check_against(weights, args.model)
    np.savez(args.npz, *[a for _, a in weights])
    print('%d arrays, %d parameters -> %s' % (len(weights), sum(a.size for _, a in weights), args.npz))


if __name__ == '__main__':
    main()
