"""The 1e-3 probability gate of the north star on TRAINED weights (not only the
synthetic glorot + random-BN weights of the other tests): vgg_like and unet_like2 are
trained for a few hundred steps on synthetic blobs with the HIP engine, then the fused
16-bit inference of the trained network is held against (a) the fp32 MFMA path of the
same library and (b) the CPU oracle.  Training is not bit-reproducible (the weight
gradients are summed with float atomics), so every run tests a slightly different network:
over a dozen runs the worst f16 voxel of the 110^3 volume was 6.5e-4 ... 1.004e-3 off fp32
(mean 4e-6 ... 1.2e-5) - AT the 1e-3 the north star asks of the fp32 path, not safely
inside it.  The test therefore bounds the maximum at 2e-3, the mean at 3e-5 and the share
of voxels beyond 5e-4 at 1 % (observed 0.03 - 0.2 %); bf16 - 8 significant bits - is bounded at what it delivers;
the detections of the f16 and fp32 predictions are compared."""
import numpy as np
import pytest

from flypylib_amd import FplNetwork, fplmodels, fplobjdetect
from oracle import cnn_oracle, infer_oracle

pytestmark = pytest.mark.gpu


def _blob_region(seed, n, radius=3, step=16):
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


def _train(factory, steps, batch, dense):
    net = FplNetwork(factory)
    im, labels, _ = _blob_region(1, 96)
    mask = np.ones_like(labels)
    if dense:
        gen = fplobjdetect.gen_volume2([[im, labels, mask]], net.rf_size, batch, 0.5,
                                       rng=np.random.RandomState(0))
    else:
        gen = fplobjdetect.gen_batches([[im, labels, mask]], net.rf_size, batch,
                                       rng=np.random.RandomState(0))
    net.train(gen, steps, 1, None, None)
    return net


@pytest.mark.parametrize('name,steps,batch,dense,tile,off', [
    ('vgg_like', 700, 32, False, 46, 7), ('unet_like2', 500, 16, True, 52, 9)])
def test_gate_on_trained_weights(ctx, name, steps, batch, dense, tile, off):
    net = _train(getattr(fplmodels, name), steps, batch, dense)
    net.infer_sz = (tile,) * 3
    net._set_infer()
    im, _, locs = _blob_region(2, 110)
    p32 = net.infer(im, precision='f32')
    p16 = net.infer(im, precision='f16')
    pb16 = net.infer(im, precision='bf16')
    # the trained network does something: confident on blobs, quiet elsewhere
    assert p32.max() > 0.8 and np.median(p32[off:-off, off:-off, off:-off]) < 0.2
    d16, db16 = np.abs(p16 - p32), np.abs(pb16 - p32)
    assert d16.max() < 2e-3 and d16.mean() < 3e-5 and np.mean(d16 > 5e-4) < 1e-2, \
        'f16 vs fp32 on trained %s: max %g mean %g' % (name, d16.max(), d16.mean())
    assert db16.max() < 3e-2 and np.mean(db16 > 1e-3) < 0.02, (db16.max(), np.mean(db16 > 1e-3))
    print('%s trained: f16 max %.2e mean %.2e | bf16 max %.2e mean %.2e, %.3f %% of voxels > 1e-3'
          % (name, d16.max(), d16.mean(), db16.max(), db16.mean(), 100 * np.mean(db16 > 1e-3)))
    # fp32 library path vs the CPU oracle on one tile of the same trained weights
    g = net.infer_network.graph
    x0 = 20
    tile_in = im[x0:x0 + tile, x0:x0 + tile, x0:x0 + tile][None, ..., None]
    want = cnn_oracle.graph_forward(g, tile_in.astype(np.float32),
                                    upsample_stride=net.rf_stride)[0, ..., 0]
    got = net.infer_network.predict(tile_in)[0, ..., 0]
    assert np.abs(got - want).max() < 1e-4
    # the detections of the f16 and of the fp32 prediction: the same objects.  (Point
    # lists are bit-identical for the SAME prediction - test_gpu_voxel2obj.py; two
    # predictions 1e-4 apart may break a tie between neighbouring voxels differently.)
    kw = dict(obj_min_dist=6, smoothing_sigma=1.5, buffer_sz=off + 2, thd=0.5)
    a = fplobjdetect.voxel2obj(p32, **kw)
    b = fplobjdetect.voxel2obj(p16, **kw)
    assert len(a['conf']) > 20 and abs(len(a['conf']) - len(b['conf'])) <= 1
    dist = np.linalg.norm(a['locs'][:, None, :] - b['locs'][None, :, :], axis=2)
    near = dist.min(axis=1)
    assert np.mean(near <= 2.0) >= 0.98, np.sort(near)[-5:]
    matched = dist.argmin(axis=1)[near <= 2.0]
    np.testing.assert_allclose(a['conf'][near <= 2.0], b['conf'][matched], atol=2e-3)
    # and they are the planted blobs
    hit = np.linalg.norm(a['locs'][:, None, :] - locs[None, :, :].astype(float), axis=2).min(axis=1)
    assert np.mean(hit <= 4.0) > 0.8
