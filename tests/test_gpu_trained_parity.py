"""The 1e-3 probability gate of the north star on TRAINED weights - the committed fixture
tests/golden/trained_{vgg_like,unet_like2}.npz (trained once on a GPU box by
tools/make_trained_fixture.py; training is not bit-reproducible, a committed file is),
not only the synthetic glorot + random-BN weights of the other tests.

What is held, per precision, against the fp32 path of the same library (itself held to
1e-4 of the CPU oracle here, 2e-7 typical):

  f16s  split IEEE halves (vgg_like and unet_like2): fp32-grade - max |dp| < 1e-5 - and the detections
        of its prediction are the SAME POINT SET as those of the fp32 prediction (the same
        voxels, confidences within 1e-5, the same order except between confidences that tie
        to 1e-6: tests/helpers.py::same_detections), also on a 582^3 substack at the pipeline's
        voxel2obj parameters.  This is the path that meets "within 1e-3, identical detections".
  f16   plain IEEE half: inside the 1e-3 gate on this fixture (8.3e-4 / 7.0e-4 observed)
        but without margin, and its detections may differ from fp32's in a tie-break.
  bf16  bounded at what 8 significant bits deliver.
"""
import numpy as np

from tests import helpers
import pytest

from flypylib_amd import fplobjdetect
from oracle import cnn_oracle
from tests.trained_fixture import RECIPES, blob_region, blob_region_u8, trained_network

pytestmark = pytest.mark.gpu


def _same_detections(a, b, conf_tol):
    moved = helpers.same_detections(a, b, conf_tol, tie=1e-6)
    print('detections out of order inside a 1e-6 tie:', moved, 'of', len(a['conf']))
    assert moved <= max(2, len(a['conf']) // 100), moved


@pytest.mark.parametrize('name', ['vgg_like', 'unet_like2'])
def test_gate_on_trained_weights(ctx, name):
    r = RECIPES[name]
    tile, off = r['tile'], r['off']
    net = trained_network(name)
    im, _, locs = blob_region(2, 110)
    p32 = net.infer(im, precision='f32')
    p16 = net.infer(im, precision='f16')
    pb16 = net.infer(im, precision='bf16')
    # the trained network does something: confident on blobs, quiet elsewhere
    assert p32.max() > 0.8 and np.median(p32[off:-off, off:-off, off:-off]) < 0.2
    d16, db16 = np.abs(p16 - p32), np.abs(pb16 - p32)
    print('%s trained: f16 max %.2e mean %.2e | bf16 max %.2e mean %.2e, %.3f %% of voxels > 1e-3'
          % (name, d16.max(), d16.mean(), db16.max(), db16.mean(), 100 * np.mean(db16 > 1e-3)))
    assert d16.max() < 1e-3 and d16.mean() < 3e-5, \
        'f16 vs fp32 on trained %s: max %g mean %g' % (name, d16.max(), d16.mean())
    assert db16.max() < 3e-2 and np.mean(db16 > 1e-3) < 0.02, (db16.max(), np.mean(db16 > 1e-3))
    # fp32 library path vs the CPU oracle on one tile of the same trained weights
    g = net.infer_network.graph
    x0 = 20
    tile_in = im[x0:x0 + tile, x0:x0 + tile, x0:x0 + tile][None, ..., None]
    want = cnn_oracle.graph_forward(g, tile_in.astype(np.float32),
                                    upsample_stride=net.rf_stride)[0, ..., 0]
    got = net.infer_network.predict(tile_in)[0, ..., 0]
    assert np.abs(got - want).max() < 1e-4
    kw = dict(obj_min_dist=6, smoothing_sigma=1.5, buffer_sz=off + 2, thd=0.5)
    a = fplobjdetect.voxel2obj(p32, **kw)
    assert len(a['conf']) > 20
    # ... and they are the planted blobs
    hit = np.linalg.norm(a['locs'][:, None, :] - locs[None, :, :].astype(float), axis=2).min(axis=1)
    assert np.mean(hit <= 4.0) > 0.8
    if True:                          # split halves: both networks of the fixture
        ps = net.infer(im, precision='f16s')
        ds = np.abs(ps - p32)
        print('%s trained: f16s max %.2e mean %.2e' % (name, ds.max(), ds.mean()))
        assert ds.max() < 1e-5, 'split halves vs fp32 on trained %s: max %g' % (name, ds.max())
        _same_detections(a, fplobjdetect.voxel2obj(ps, **kw), 1e-5)
    # plain f16: the same objects, up to a tie-break between neighbouring voxels
    b = fplobjdetect.voxel2obj(p16, **kw)
    assert abs(len(a['conf']) - len(b['conf'])) <= 1
    dist = np.linalg.norm(a['locs'][:, None, :] - b['locs'][None, :, :], axis=2)
    near = dist.min(axis=1)
    assert np.mean(near <= 2.0) >= 0.98, np.sort(near)[-5:]
    matched = dist.argmin(axis=1)[near <= 2.0]
    np.testing.assert_allclose(a['conf'][near <= 2.0], b['conf'][matched], atol=2e-3)


def test_split_detections_identical_on_a_substack(ctx):
    """one substack of the pipeline (512 + 2 * 35 = 582 voxels, uint8 in, r = 27, sigma = 5,
    `fplobjdetect.py:845,1031-1034`): voxel2obj of the split-half prediction == voxel2obj of
    the fp32 prediction - the same voxels, confidences within 1e-5, the same order except
    between confidences that tie to 1e-6 - and the probabilities within 1e-5"""
    net = trained_network('vgg_like', tile=102)
    # blobs 48 +- 3 voxels apart: further than obj_min_dist, as T-bars are - every blob is
    # one smoothed peak.  (At the training pitch of 16 the sigma-5 smoothing leaves a
    # nearly flat field whose 3 276 maxima are decided by differences of 1e-7: there even
    # two fp32 implementations disagree on a handful of tie-breaks.)
    u8, _, locs = blob_region_u8(5, 582, step=48)
    norm = (128.0, 33.0)
    p32 = net.infer(u8, normalize=norm, precision='f32')
    ps = net.infer(u8, normalize=norm, precision='f16s')
    d = np.abs(ps - p32)
    print('582^3 substack: f16s vs fp32 max %.2e mean %.2e' % (d.max(), d.mean()))
    assert d.max() < 1e-5
    kw = dict(obj_min_dist=27, smoothing_sigma=5, buffer_sz=35, thd=0.1)
    a = fplobjdetect.voxel2obj(p32, **kw)
    b = fplobjdetect.voxel2obj(ps, **kw)
    print('detections', len(a['conf']))
    assert len(a['conf']) > 500
    _same_detections(a, b, 1e-5)
    hit = np.linalg.norm(a['locs'][:, None, :] - locs[None, :, :].astype(float), axis=2).min(axis=1)
    assert np.mean(hit <= 4.0) > 0.95
