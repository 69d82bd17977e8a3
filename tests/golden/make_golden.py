"""Regenerate tests/golden/*.npz from the REFERENCE's own code.

Dev-time tool for the build container only (needs /root/reference; the GPU box
never runs this).  Fixtures hold data only: case parameters, seeds of the
deterministic input generators in flypylib_amd/synth.py, and the reference's
outputs (point lists, checksums, small arrays).

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _ref_import import import_reference  # noqa: E402
from flypylib_amd import synth  # noqa: E402

fplutils, fplobjdetect, fplnetwork = import_reference()

# (name, generator, seed, shape, r, sigma, thd, buffer, offset)
V2O_CASES = [
    ('uniform_small', 'uniform', 11, (40, 44, 48), 7, 2.0, 0, 0, (0, 0, 0)),
    ('uniform_buf3', 'uniform', 12, (30, 50, 41), 9, 1.5, 0.1, (3, 4, 5), (1, 2, 3)),
    ('uniform_thd', 'uniform', 13, (64, 64, 64), 5, 5.0, 0.5, 5, (10, 20, 30)),
    ('reflect_r3_s5', 'uniform', 14, (33, 20, 25), 3, 5.0, 0, 2, (0, 0, 0)),
    ('blobs_r9', 'blobs', 21, (72, 80, 96), 9, 2.0, 0.1, 5, (100, 200, 300)),
    ('blobs_r27_s5', 'blobs', 22, (96, 96, 96), 27, 5.0, 0, 0, (0, 0, 0)),
    ('blobs_noncubic', 'blobs', 23, (50, 120, 70), 12, 3.0, 0.3, (2, 0, 7), (5, 6, 7)),
    ('plateau_ties', 'plateau', 31, (48, 48, 48), 7, 1.5, 0, 0, (0, 0, 0)),
    ('all_zero', 'zeros', 0, (20, 22, 24), 5, 2.0, 0, 0, (0, 0, 0)),
    ('thd_above_all', 'uniform', 15, (24, 24, 24), 4, 2.0, 2.0, 0, (0, 0, 0)),
]


def make_pred(kind, seed, shape):
    if kind == 'uniform':
        return synth.hash_uniform_f32(seed, shape)
    if kind == 'blobs':
        return synth.blob_prob_volume(seed, shape)
    if kind == 'zeros':
        return np.zeros(shape, np.float32)
    if kind == 'plateau':
        # exact ties: identical flat-topped cubes on a lattice -> argmax ties
        v = np.zeros(shape, np.float32)
        for z in range(8, shape[0] - 8, 16):
            for y in range(8, shape[1] - 8, 16):
                for x in range(8, shape[2] - 8, 16):
                    v[z:z + 4, y:y + 4, x:x + 4] = 0.75
        return v
    raise ValueError(kind)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gen_set_filter():
    out = {}
    for r in (1, 3, 7, 27):
        f = fplutils.set_filter(r)
        out['r%d_count' % r] = np.int64(f.sum())
        out['r%d_sha' % r] = np.array(sha(f.astype(np.uint8)))
        if r <= 3:
            out['r%d_mask' % r] = f
    np.savez_compressed(os.path.join(HERE, 'set_filter.npz'), **out)


def gen_voxel2obj():
    from scipy import ndimage
    out = {'names': np.array([c[0] for c in V2O_CASES])}
    for name, kind, seed, shape, r, sigma, thd, buf, off in V2O_CASES:
        pred = make_pred(kind, seed, shape)
        res = fplobjdetect.voxel2obj(pred.copy(), r, sigma, off, buf, thd)
        out[name + '_kind'] = np.array(kind)
        out[name + '_params'] = np.array(
            [seed, shape[0], shape[1], shape[2], r], np.int64)
        out[name + '_sigma_thd'] = np.array([sigma, thd], np.float64)
        out[name + '_buffer'] = np.array(fplutils.to3d(buf), np.int64)
        out[name + '_offset'] = np.array(off, np.int64)
        out[name + '_locs'] = res['locs']
        out[name + '_conf'] = res['conf']
        out[name + '_pred_sha'] = np.array(sha(pred))
        # intermediate pins: smoothed padded volume checksum + threshold
        sm = ndimage.gaussian_filter(np.pad(pred, r, 'constant'), sigma,
                                     truncate=2.0)
        sm[:r] = 0; sm[-r:] = 0; sm[:, :r] = 0; sm[:, -r:] = 0
        sm[:, :, :r] = 0; sm[:, :, -r:] = 0
        out[name + '_smooth_sha'] = np.array(sha(sm))
        out[name + '_pct97'] = np.array(np.percentile(sm, 97))
        print('%-16s %4d detections' % (name, len(res['conf'])))
    np.savez_compressed(os.path.join(HERE, 'voxel2obj.npz'), **out)


# (name, pred kind, pred seed, shape, r, sigma, thd, buffer, seg seed, n_sites, tiny,
#  seg_dilate, seg_sz_thd, seg_force)
V2O_SEG_CASES = [
    ('seg_plain', 'blobs', 41, (48, 52, 56), 7, 2.0, 0.05, 2, 3, 9, 0, None, None, None),
    ('seg_dilate', 'blobs', 42, (48, 52, 56), 7, 2.0, 0.05, 0, 4, 12, 0, 2, None, None),
    ('seg_force', 'uniform', 43, (40, 44, 48), 6, 1.5, 0.1, 3, 5, 20, 0, None, None, 2),
    ('seg_small', 'blobs', 44, (56, 48, 52), 9, 2.0, 0.05, 0, 6, 8, 30, 1, 60, None),
    ('seg_all', 'blobs', 45, (64, 60, 56), 9, 3.0, 0.05, (2, 3, 4), 7, 15, 20, 3, 40, 4),
    # obj_min_dist beyond 31: a cube row no longer fits one 64-bit mask (two / four words)
    ('seg_r33', 'blobs', 46, (72, 80, 88), 33, 4.0, 0.05, 3, 8, 10, 0, 2, None, 6),
    ('seg_r40', 'blobs', 47, (90, 84, 96), 40, 5.0, 0.02, 0, 9, 14, 25, 1, 50, None),
    ('seg_r70', 'uniform', 48, (150, 40, 44), 70, 2.0, 0.1, 0, 10, 6, 0, None, None, 3),
]


def gen_voxel2obj_seg():
    """the segmentation-aware branch of the reference's voxel2obj (:161-224)"""
    out = {'names': np.array([c[0] for c in V2O_SEG_CASES])}
    for (name, kind, pseed, shape, r, sigma, thd, buf, sseed, n_sites, tiny, dil, szt,
         force) in V2O_SEG_CASES:
        pred = make_pred(kind, pseed, shape)
        seg = synth.voronoi_segmentation(sseed, shape, n_sites, tiny)
        res = fplobjdetect.voxel2obj(pred.copy(), r, sigma, (0, 0, 0), buf, thd, seg=seg.copy(),
                                     seg_dilate=dil, seg_sz_thd=szt, seg_force=force)
        plain = fplobjdetect.voxel2obj(pred.copy(), r, sigma, (0, 0, 0), buf, thd)
        out[name + '_locs'], out[name + '_conf'] = res['locs'], res['conf']
        out[name + '_pred_sha'], out[name + '_seg_sha'] = np.array(sha(pred)), np.array(sha(seg))
        print('%-12s %4d detections (%d without the segmentation)'
              % (name, len(res['conf']), len(plain['conf'])))
    np.savez_compressed(os.path.join(HERE, 'voxel2obj_seg.npz'), **out)


def gen_voxel2obj_f64():
    """float64 predictions through the reference's voxel2obj (it pads, smooths and
    compares in the input's dtype): point lists + the float64 percentile"""
    from tests.helpers import V2O_F64_CASES, make_pred_f64
    out = {'names': np.array([c[0] for c in V2O_F64_CASES])}
    for name, kind, seed, shape, r, sigma, thd, buf, off, segp in V2O_F64_CASES:
        pred = make_pred_f64(kind, seed, shape)
        assert pred.dtype == np.float64 and not np.array_equal(pred, pred.astype(np.float32))
        kw = {}
        if segp is not None:
            sseed, n_sites, tiny, dil, szt, force = segp
            kw = dict(seg=synth.voronoi_segmentation(sseed, shape, n_sites, tiny), seg_dilate=dil,
                      seg_sz_thd=szt, seg_force=force)
        res = fplobjdetect.voxel2obj(pred.copy(), r, sigma, tuple(off), buf, thd, **kw)
        f32 = fplobjdetect.voxel2obj(pred.astype(np.float32), r, sigma, tuple(off), buf, thd, **kw)
        out[name + '_locs'], out[name + '_conf'] = res['locs'], res['conf']
        out[name + '_pred_sha'] = np.array(sha(pred))
        same = (res['locs'].shape == f32['locs'].shape and np.array_equal(res['locs'], f32['locs'])
                and np.array_equal(res['conf'], f32['conf']))
        print('%-12s %4d detections (float32 input: %d, identical: %s)'
              % (name, len(res['conf']), len(f32['conf']), same))
    np.savez_compressed(os.path.join(HERE, 'voxel2obj_f64.npz'), **out)


def gen_voxel2obj_int():
    """integer predictions through the reference's voxel2obj: scipy filters an integer array
    with float64 accumulation and a C cast (truncation) back to the integer type after every
    axis; np.percentile gives a float64; the point list comes back as int64"""
    from tests.helpers import V2O_INT_CASES, make_pred_int
    out = {'names': np.array([c[0] for c in V2O_INT_CASES])}
    for name, kind, seed, shape, dtype, scale, r, sigma, thd, buf, off, segp in V2O_INT_CASES:
        pred = make_pred_int(kind, seed, shape, dtype, scale)
        kw = {}
        if segp is not None:
            sseed, n_sites, tiny, dil, szt, force = segp
            kw = dict(seg=synth.voronoi_segmentation(sseed, shape, n_sites, tiny), seg_dilate=dil,
                      seg_sz_thd=szt, seg_force=force)
        res = fplobjdetect.voxel2obj(pred.copy(), r, sigma, tuple(off), buf, thd, **kw)
        out[name + '_locs'], out[name + '_conf'] = res['locs'], res['conf']
        out[name + '_pred_sha'] = np.array(sha(pred))
        print('%-12s %4d detections, dtypes %s / %s' % (name, len(res['conf']), res['locs'].dtype,
                                                        res['conf'].dtype))
    np.savez_compressed(os.path.join(HERE, 'voxel2obj_int.npz'), **out)


class _FakeNet:
    """crop-identity stand-in for the Keras inference network"""

    def __init__(self, infer_sz, off, fn):
        self.input_shape = (None,) + tuple(infer_sz) + (1,)
        self.off, self.fn = off, fn

    def predict(self, x, batch_size=1):
        return self.fn(x, self.off)


def _crop_identity(x, off):
    # upsampled full-resolution output of the valid region: out = in cropped
    return x[:, off[0]:x.shape[1] - off[0], off[1]:x.shape[2] - off[1],
             off[2]:x.shape[3] - off[2], :].astype(np.float32)


def _coarse4(x, off):
    # stride-4 net stand-in: value of the voxel at the coarse cell's origin
    c = x[:, off[0]:x.shape[1] - off[0], off[1]:x.shape[2] - off[1],
          off[2]:x.shape[3] - off[2], :]
    c = c[:, ::4, ::4, ::4, :]
    for ax in (1, 2, 3):
        c = np.repeat(c, 4, axis=ax)
    return c.astype(np.float32)


def gen_infer():
    cases = [
        ('crop_50_47_41', 41, (50, 47, 41), (30, 30, 30), (7, 7, 7), _crop_identity, 1),
        ('crop_n_gpu3', 42, (64, 33, 40), (30, 30, 30), (7, 7, 7), _crop_identity, 3),
        ('coarse4_61', 43, (61, 58, 47), (30, 30, 30), (7, 7, 7), _coarse4, 1),
        ('unet_lattice', 44, (70, 45, 52), (28, 28, 28), (9, 9, 9), _crop_identity, 2),
        ('exact_multiple', 45, (46, 46, 46), (30, 30, 30), (7, 7, 7), _crop_identity, 1),
    ]
    out = {'names': np.array([c[0] for c in cases])}
    for name, seed, shape, isz, off, fn, n_gpu in cases:
        img = synth.hash_uniform_f32(seed, shape)
        net = fplnetwork.FplNetwork.__new__(fplnetwork.FplNetwork)
        net.infer_network = _FakeNet(isz, off, fn)
        net.infer_sz, net.rf_offset, net.n_gpu = isz, off, n_gpu
        pred = net.infer(img)
        out[name + '_shape'] = np.array(shape, np.int64)
        out[name + '_isz'] = np.array(isz, np.int64)
        out[name + '_off'] = np.array(off, np.int64)
        out[name + '_n_gpu'] = np.int64(n_gpu)
        out[name + '_fn'] = np.array(fn.__name__)
        out[name + '_seed'] = np.int64(seed)
        out[name + '_pred_sha'] = np.array(sha(pred))
        out[name + '_pred_sample'] = pred[::7, ::5, ::3].copy()
        print('%-16s pred sum %.4f' % (name, pred.sum()))
    np.savez_compressed(os.path.join(HERE, 'infer_lattice.npz'), **out)


# ---- fri_get_image (fplobjdetect.py:1023-1124): substack + buffer read, zero fill
# outside the extents, normalisation by the interpolated filtered mean ---------------
FRI_VOLUME = (11, (20, 24, 28))         # em_volume_u8 seed, shape
# (name, size, z, y, x, buffer, image_normalize)
FRI_CASES = [
    ('interior', 8, 6, 8, 10, 4, [128., 33.]),
    ('interior_frac', 8, 6, 8, 10, 4, [128., 33., 0.3]),
    ('clip_low', 8, 0, 0, 0, 4, [120., 30., 0.5]),
    ('clip_high', 8, 16, 16, 24, 4, [128., 33., 0.0]),
    ('outside', 8, 200, 0, 0, 4, [128., 33.]),
]


def fri_volume():
    """the synthetic EM volume with planted 0 / 1 / 200 / 255 voxels, so that the
    1 < v < 200 filter of the normalisation matters"""
    vol = synth.em_volume_u8(FRI_VOLUME[0], FRI_VOLUME[1])
    vol[::5, ::3, ::7] = 0
    vol[1::4, ::5, 2::3] = 255
    vol[2::6, 1::4, ::5] = 1
    vol[::7, 2::5, 1::3] = 200
    return vol


class _FakeNode:
    """what the reference's DICED branch needs of an array: extents + slicing"""

    def __init__(self, arr):
        self.arr = arr

    def get_extents(self):
        return [slice(0, s) for s in self.arr.shape]

    def __getitem__(self, key):
        return self.arr[key]


def gen_fri_get_image():
    import tempfile
    vol = fri_volume()
    out = {'volume_sha': np.array(sha(vol))}
    with tempfile.TemporaryDirectory() as norm_dir:
        for name, size, z, y, x, buf, norm in FRI_CASES:
            ss = fplobjdetect.szyx(size, z, y, x)
            info = [ss, 'gs://unused', 'uuid', norm, buf, None, norm_dir, 'grayscale']
            image, ss_out = fplobjdetect.fri_get_image(info, _FakeNode(vol), True)
            assert ss_out == ss
            if image is None:
                out[name + '/none'] = np.array(1)
                continue
            # numpy >= 2 makes this float64 (np.float64 mean); the reference's numpy
            # 1.13 kept float32 - store the values, compare to float32 rounding
            out[name + '/image'] = np.asarray(image, np.float64)
            line = open('%s/%d_%d_%d_%d.txt' % (norm_dir, size, z, y, x)).read()
            out[name + '/norm_line'] = np.array(line)
    np.savez_compressed(os.path.join(HERE, 'fri_get_image.npz'), **out)


# ---- fplsynapses JSON formats + the pulp-free branches of obj_pr -----------------------
def synapse_points():
    rs = np.random.RandomState(7)
    return {'locs': np.floor(rs.rand(12, 3) * [60, 50, 40]), 'conf': rs.rand(12)}


def gen_synapses():
    import json
    from flypylib import fplsynapses
    tb = synapse_points()
    out = {}
    dvid = fplsynapses.tbars_to_json_format(tb, labels=np.arange(12) * 3)
    rav = fplsynapses.tbars_to_json_format_raveler(tb)
    out['dvid_json'] = np.array(json.dumps(dvid, sort_keys=True))
    out['raveler_json'] = np.array(json.dumps(rav, sort_keys=True))
    for name, text in (('dvid', json.dumps(dvid)), ('raveler', json.dumps(rav)),
                       ('nested', json.dumps([dvid]))):
        back = fplsynapses.load_from_json(text)
        out[name + '/locs'], out[name + '/conf'] = back['locs'], back['conf']
        cut = fplsynapses.load_from_json(text, (60, 50, 40), (10, 5, 8))
        out[name + '/buf_locs'], out[name + '/buf_conf'] = cut['locs'], cut['conf']
    # obj_pr without pulp: the empty-set branches
    e = np.zeros((0, 3))
    for name, (p, g) in (('no_pred', (e, tb['locs'])), ('no_gt', (tb['locs'], e)), ('none', (e, e))):
        r = fplobjdetect.obj_pr(p, g, 5.0)
        out['pr_' + name] = np.array([r.num_tp, r.tot_pred, r.tot_gt, r.pp, r.rr], np.float64)
    # write_labels_mask: the arrays it hands to (the inert stub of) h5py
    import sys
    h5 = sys.modules['h5py']
    h5.File.reset_mock()
    tb_lm = {'locs': np.array([[12, 14, 16], [20, 15, 13], [25, 25, 25]]), 'conf': np.ones(3)}
    fplsynapses.write_labels_mask(tb_lm, np.ones((36, 38, 40), 'uint8'), 3, 6, 4, '/nonexistent/x')
    sets = h5.File.return_value.__getitem__.return_value.__setitem__.call_args_list
    assert len(sets) == 2
    out['lm_labels'], out['lm_mask'] = np.asarray(sets[0][0][1]), np.asarray(sets[1][0][1])
    np.savez_compressed(os.path.join(HERE, 'synapses.npz'), **out)


def gen_keras_tiny():
    """keras_tiny.h5 / .npz: NOT a reference output (h5py and Keras are absent here) - the
    bytes of the package's own HDF5 writer (flypylib_amd/h5min.py, Keras `model.save`
    layout) for a 3-layer network with seeded weights, and the arrays they must read back
    as.  (The reader's fixture from the real library is keras_libhdf5.h5,
    make_h5_fixture.py; tests/test_keras_io.py also opens this file with libhdf5.)"""
    from flypylib_amd import synth
    from flypylib_amd.program import LayerGraph
    g = LayerGraph(None, seed=3)
    x = g.relu(g.bn(g.conv(g.input(), 4, 3)))
    g.finish(g.conv(x, 1, 1, use_bias=True, activation='sigmoid'))
    synth.synthetic_weights(g, 17)
    g.save(os.path.join(HERE, 'keras_tiny.h5'))
    np.savez(os.path.join(HERE, 'keras_tiny.npz'), *g.get_weights())



def training_volumes(seed, shape):
    """seeded (image f32, labels u8, mask u8) for the training-generator fixtures; shared
    with tests/test_training_data.py (same code there: the fixture holds outputs only)"""
    im = synth.em_volume_u8(seed, shape).astype(np.float32)
    ll = (synth.hash_uniform_f32(seed + 100, shape) > np.float32(0.97)).astype(np.uint8)
    mm = np.ones(shape, np.uint8)
    mm[: shape[0] // 3, : shape[1] // 2, :] = 0
    return im, ll, mm


def gen_training_generators():
    """gen_batches / gen_volume / gen_volume2 of the REFERENCE on seeded .h5 volumes
    (reference fplobjdetect.py:27-130, 524-658, 660-822).  The reference reads its inputs
    with h5py, absent here: for this run `h5py.File(path)['/main'][:]` is served by the
    package's own HDF5 reader (flypylib_amd/h5min.py) on files written by its writer -
    the arrays the reference sees are exactly the arrays below, and everything after the
    read (sampling order, np.random call order, augmentation) is the reference's code."""
    import tempfile
    import types
    from flypylib_amd import h5min, keras_io

    class _File:
        def __init__(self, path, mode='r'):
            self._f = h5min.File(path)

        def __getitem__(self, key):
            return self._f[key.lstrip('/')]

    shim = types.ModuleType('h5py')
    shim.File = _File
    sys.modules['h5py'] = shim
    fplobjdetect.h5py = shim
    wd = tempfile.mkdtemp(prefix='gen_')
    shapes = [(40, 44, 48), (36, 52, 40)]
    train = []
    for v, shape in enumerate(shapes):
        im, ll, mm = training_volumes(50 + v, shape)
        keras_io.write_main('%s/im%d.h5' % (wd, v), im)
        keras_io.write_main('%s/v%d_labels.h5' % (wd, v), ll)
        keras_io.write_main('%s/v%d_mask.h5' % (wd, v), mm)
        train.append(('%s/im%d.h5' % (wd, v), '%s/v%d_' % (wd, v)))
    out = {'shapes': np.array(shapes), 'vol_seeds': np.array([50, 51])}
    cases = [
        ('batches', lambda: fplobjdetect.gen_batches(train, (12, 10, 10), 6), 7, 3),
        ('batches_mask', lambda: fplobjdetect.gen_batches(train, (12, 12, 12), 4, True), 8, 3),
        ('volume', lambda: fplobjdetect.gen_volume(train, (24, 24, 24), 3, 0.5), 9, 4),
        ('volume2', lambda: fplobjdetect.gen_volume2(train, (24, 24, 24), 3, 0.5), 10, 3),
        ('volume2_noise', lambda: fplobjdetect.gen_volume2(train, (24, 24, 24), 2, 0.3,
                                                            noise_aug=[0.05, 0.1]), 11, 3),
    ]
    for name, make, seed, n in cases:
        np.random.seed(seed)
        gen = make()
        for i in range(n):
            d, lab = next(gen)
            # whole batches as checksums; the labels (small) and one example in full
            out['%s_data_sha_%d' % (name, i)] = np.array(sha(np.ascontiguousarray(d)))
            out['%s_labels_%d' % (name, i)] = np.array(lab)
            if i == 0:
                out['%s_example0' % name] = np.array(d[0, ..., 0])
        out['%s_seed' % name] = np.array(seed)
        print(name, 'ok', d.shape, lab.shape, int(lab.sum()))
    np.savez_compressed(os.path.join(HERE, 'training_generators.npz'), **out)


if __name__ == '__main__':
    if 'generators' in sys.argv[1:]:
        gen_training_generators()
        sys.exit(0)
    gen_keras_tiny()
    gen_voxel2obj_seg()
    gen_voxel2obj_f64()
    gen_voxel2obj_int()
    gen_synapses()
    gen_fri_get_image()
    gen_set_filter()
    gen_voxel2obj()
    gen_infer()
    print('golden fixtures written to', HERE)
