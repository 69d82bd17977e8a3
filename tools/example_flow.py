#!/usr/bin/env python3
"""The reference's documented entry point (scripts/fpl_fib25_example.py: prepare a
training and a test region, train unet_like2 on gen_volume2 batches, save, infer,
evaluate_substacks -> precision / recall) on SYNTHETIC data with flypylib_amd - the
public FIB-25 volume and its annotations are not reachable offline.

    python tools/example_flow.py [--size 128] [--steps 150] [--epochs 2] [--out DIR]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flypylib_amd import FplNetwork, fplmodels, fplobjdetect, fplsynapses   # noqa: E402


def make_region(seed, size, buffer_sz, radius):
    """normalised image with dark blobs ("T-bars") at known positions + their list"""
    rs = np.random.RandomState(seed)
    n = size + 2 * buffer_sz
    im = rs.randn(n, n, n).astype(np.float32) * 0.5
    step = 16
    grid = np.arange(buffer_sz + 8, n - buffer_sz - 8, step)
    locs = np.array([(x, y, z) for z in grid for y in grid for x in grid], np.int64)
    locs = locs + rs.randint(-3, 4, locs.shape)
    zz, yy, xx = np.meshgrid(*(np.arange(-radius, radius + 1),) * 3, indexing='ij')
    ball = (zz ** 2 + yy ** 2 + xx ** 2 <= radius ** 2)
    for x, y, z in locs:
        sl = (slice(z - radius, z + radius + 1), slice(y - radius, y + radius + 1),
              slice(x - radius, x + radius + 1))
        im[sl][ball] -= 2.5
    return im, {'locs': locs.astype(np.float64), 'conf': np.ones(len(locs))}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--steps', type=int, default=150)
    ap.add_argument('--epochs', type=int, default=2)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--out', default=None)
    a = ap.parse_args(argv)
    data_dir = a.out or tempfile.mkdtemp(prefix='fpl_example_')
    os.makedirs(data_dir, exist_ok=True)
    buffer_sz, radius = 14, {'use': 3, 'ign': 6}

    evals = {}
    for prefix, seed in ((0, 1), (1, 2)):          # "0" = training region, "1" = test region
        im, tbars = make_region(seed, a.size, buffer_sz, radius['use'])
        np.save('%s/%d_im.npy' % (data_dir, prefix), im)
        fplsynapses.tbars_to_json_format_raveler(tbars, '%s/%d_synapses.json' % (data_dir, prefix))
        fplsynapses.write_labels_mask(tbars, np.ones(im.shape, 'uint8'), radius['use'],
                                      radius['ign'], buffer_sz, '%s/%d' % (data_dir, prefix))
        evals[prefix] = [['%s/%d_im.npy' % (data_dir, prefix),
                          '%s/%d_synapses.json' % (data_dir, prefix)]]

    network = FplNetwork(fplmodels.unet_like2)
    network.infer_sz = (52, 52, 52)
    train_data = [['%s/0_im.npy' % data_dir, '%s/0_labels.npy' % data_dir,
                   '%s/0_mask.npy' % data_dir]]
    generator = fplobjdetect.gen_volume2(train_data, network.rf_size, a.batch, 0.5,
                                         rng=np.random.RandomState(0))
    network.train(generator, a.steps, a.epochs, '%s/log.csv' % data_dir, '%s/epoch' % data_dir)
    network.save_network('%s/net' % data_dir)

    thds = np.arange(0.3, 0.96, 0.05)
    out = {}
    for name, key in (('train', 0), ('test', 1)):
        images = [[np.load(p), j] for p, j in evals[key]]
        agg, _ = fplobjdetect.evaluate_substacks(network, images, thds, obj_min_dist=6,
                                                 smoothing_sigma=1.5, buffer_sz=buffer_sz)
        out[name] = agg
        best = int(np.argmax(agg.pp * agg.rr))
        print('%s: %d ground-truth points; at threshold %.2f precision %.3f recall %.3f'
              % (name, int(agg.tot_gt[0]), thds[best], agg.pp[best], agg.rr[best]))
    print(open('%s/log.csv' % data_dir).read().strip())
    return out


if __name__ == '__main__':
    main()
