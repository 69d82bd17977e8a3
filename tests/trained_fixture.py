"""The trained-weights parity fixture: synthetic blob regions, the training recipes that
produced tests/golden/trained_*.npz (tools/make_trained_fixture.py, run once on a GPU
box) and a loader that puts those weights into an `FplNetwork`."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# name -> training recipe and the inference geometry the tests use
RECIPES = {
    'vgg_like': dict(steps=1500, batch=32, dense=False, tile=46, off=7),
    'unet_like2': dict(steps=800, batch=16, dense=True, tile=52, off=9),
}


def blob_region(seed, n, radius=3, step=16):
    """(image float32 (n,n,n), labels uint8, planted centres (x,y,z)): unit-variance-ish
    noise with dark balls of `radius` on a jittered grid of pitch `step` - the T-bar
    stand-in every trained-weights test uses"""
    rs = np.random.RandomState(seed)
    im = rs.randn(n, n, n).astype(np.float32) * 0.5
    grid = np.arange(12, n - 12, step)
    locs = np.array([(x, y, z) for z in grid for y in grid for x in grid], np.int64)
    locs = locs + rs.randint(-3, 4, locs.shape)
    zz, yy, xx = np.meshgrid(*(np.arange(-radius, radius + 1),) * 3, indexing='ij')
    ball = zz ** 2 + yy ** 2 + xx ** 2 <= radius ** 2
    labels = np.zeros((n, n, n), np.uint8)
    for x, y, z in locs:
        sl = (slice(z - radius, z + radius + 1), slice(y - radius, y + radius + 1),
              slice(x - radius, x + radius + 1))
        im[sl][ball] -= 2.5
        labels[sl][ball] = 1
    return im, labels, locs


def blob_region_u8(seed, n, **kw):
    """the same region as the uint8 volume an EM pipeline would hand over, with the
    FIB-25 normalisation constants (`scripts/fpl_fib25_example.py:131-134`):
    u8 = clip(round(128 + 33 * image))"""
    im, labels, locs = blob_region(seed, n, **kw)
    u8 = np.clip(np.rint(128.0 + 33.0 * im), 0, 255).astype(np.uint8)
    return u8, labels, locs


def trained_weights(name):
    with np.load(os.path.join(GOLDEN, 'trained_%s.npz' % name)) as z:
        return [z['arr_%d' % i] for i in range(len(z.files))]


def trained_network(name, tile=None):
    """an FplNetwork carrying the committed trained weights, inference net built for
    `tile` (default: the recipe's)"""
    from flypylib_amd import FplNetwork, fplmodels
    net = FplNetwork(getattr(fplmodels, name))
    net.train_single.set_weights(trained_weights(name))
    t = RECIPES[name]['tile'] if tile is None else tile
    net.infer_sz = (t,) * 3
    net._set_infer()
    return net
