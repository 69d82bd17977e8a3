"""Train the two networks of the trained-weights parity fixture ONCE, on a GPU box, and
write their weights (Keras `get_weights()` order, float32) as

    <out>/trained_vgg_like.npz, <out>/trained_unet_like2.npz

which are then committed under tests/golden/.  Training is not bit-reproducible (float
atomics in the weight gradients), which is exactly why the parity tests read a committed
file instead of training inside the test (tests/test_gpu_trained_parity.py).

    gpurun -- python tools/make_trained_fixture.py gpurun_out/fixture

Data: the synthetic blob regions of tests/trained_fixture.py (dark balls of radius 3 on
noise, the T-bar stand-in), sampled by the package's own generators exactly as the
reference's scripts do (gen_batches for vgg_like, gen_volume2 + masked focal loss for
unet_like2).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from flypylib_amd import FplNetwork, fplmodels, fplobjdetect   # noqa: E402
from tests.trained_fixture import blob_region, RECIPES         # noqa: E402


def train(name):
    r = RECIPES[name]
    net = FplNetwork(getattr(fplmodels, name))
    im, labels, _ = blob_region(1, 96)
    mask = np.ones_like(labels)
    if r['dense']:
        gen = fplobjdetect.gen_volume2([[im, labels, mask]], net.rf_size, r['batch'], 0.5,
                                       rng=np.random.RandomState(0))
    else:
        gen = fplobjdetect.gen_batches([[im, labels, mask]], net.rf_size, r['batch'],
                                       rng=np.random.RandomState(0))
    net.train(gen, r['steps'], 1, None, None)
    return net


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/fixture'
    os.makedirs(out, exist_ok=True)
    for name in RECIPES:
        net = train(name)
        w = [np.asarray(a, np.float32) for a in net.train_single.get_weights()]
        path = os.path.join(out, 'trained_%s.npz' % name)
        np.savez_compressed(path, *w)
        # what the trained network does on an unseen region, in fp32 and on the 16-bit paths
        r = RECIPES[name]
        net.infer_sz = (r['tile'],) * 3
        net._set_infer()
        im, _, locs = blob_region(2, 110)
        p32 = net.infer(im, precision='f32')
        off = r['off']
        inner = p32[off:-off, off:-off, off:-off]
        print('%s: %d arrays, %d parameters -> %s' % (name, len(w), sum(a.size for a in w), path))
        print('  fp32 prediction: max %.4f, median %.2e, voxels > 0.5: %d'
              % (inner.max(), np.median(inner), int((inner > 0.5).sum())))
        for prec in ('f16', 'bf16'):
            d = np.abs(net.infer(im, precision=prec) - p32)
            print('  %s vs fp32: max %.3e mean %.3e  share > 5e-4: %.4f %%'
                  % (prec, d.max(), d.mean(), 100 * np.mean(d > 5e-4)))
        det = fplobjdetect.voxel2obj(p32, obj_min_dist=6, smoothing_sigma=1.5,
                                     buffer_sz=off + 2, thd=0.5)
        hit = np.linalg.norm(det['locs'][:, None, :] - locs[None].astype(float), axis=2).min(axis=1)
        print('  detections %d, %.1f %% within 4 voxels of a planted blob (%d planted)'
              % (len(det['conf']), 100 * np.mean(hit <= 4), len(locs)))


if __name__ == '__main__':
    main()
