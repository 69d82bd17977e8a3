"""Training data path of the U-Nets (reference fplobjdetect.py:660-839,
fplnetwork.py:191-220): gen_volume2, voxel_loss, write_sampling_weights - host
logic, checked through constructed cases and - gen_batches, gen_volume, gen_volume2 -
against the reference's own outputs on seeded .h5 volumes
(tests/golden/training_generators.npz; the reference's h5py reads were served by the
package's HDF5 reader when the fixture was made)."""
import os

import numpy as np
import pytest

from flypylib_amd import FplNetwork, fplobjdetect


class _QuietRng:
    """numpy RandomState whose rand()/randn() return zeros: no augmentation, no
    intensity noise - sampling stays random"""

    def __init__(self, seed):
        self.r = np.random.RandomState(seed)

    def choice(self, *a, **k):
        return self.r.choice(*a, **k)

    def permutation(self, n):
        return self.r.permutation(n)

    def rand(self, *shape):
        return np.zeros(shape)

    def randn(self, *shape):
        return np.zeros(shape) if shape else 0.0


def _coded_volume(shape):
    z, y, x = np.meshgrid(*(np.arange(s) for s in shape), indexing='ij')
    return (z * 10000 + y * 100 + x).astype(np.float32)


def test_gen_volume2_samples_patches_and_dense_labels():
    shape = (40, 44, 48)
    im = _coded_volume(shape)
    ll = np.zeros(shape, np.uint8)
    ll[14:26, 14:26, 14:26] = 1
    mm = np.ones(shape, np.uint8)
    mm[:, :, 36:] = 0                      # masked region -> label 2
    B, ratio = 8, 0.75
    gen = fplobjdetect.gen_volume2([(im, ll, mm)], (24, 24, 24), B, ratio, rng=_QuietRng(1))
    n_neg = n_pos = 0
    mm_eff = mm.copy()                     # the generator clears the mask where a
    for ax in range(3):                    # patch would not fit (reference :696-701)
        sl = [slice(None)] * 3
        sl[ax] = slice(0, 12); mm_eff[tuple(sl)] = 0
        sl[ax] = slice(-12, None); mm_eff[tuple(sl)] = 0
    for _ in range(100):                   # one outer round
        data, labels = next(gen)
        assert data.shape == (B, 24, 24, 24, 1) and data.dtype == np.float32
        assert labels.shape == (B, 6, 6, 6, 1) and labels.dtype == np.uint8
        for ex in range(B):
            code = int(data[ex, 0, 0, 0, 0])
            z0, y0, x0 = code // 10000, (code // 100) % 100, code % 100
            cz, cy, cx = z0 + 12, y0 + 12, x0 + 12
            assert np.array_equal(data[ex, ..., 0], im[z0:z0 + 24, y0:y0 + 24, x0:x0 + 24])
            want = ll[cz - 3:cz + 3, cy - 3:cy + 3, cx - 3:cx + 3].copy()
            m = mm_eff[cz - 3:cz + 3, cy - 3:cy + 3, cx - 3:cx + 3]
            assert mm_eff[cz, cy, cx] == 1
            got = labels[ex, ..., 0]
            assert np.array_equal(got[m == 1], want[m == 1]) and np.all(got[m == 0] == 2)
            if ll[cz, cy, cx] == 1:
                n_pos += 1
            else:
                n_neg += 1
    assert n_neg == round(ratio * 100 * B) and n_pos == 100 * B - n_neg


def test_gen_volume2_weighted_sampling_and_augmentation_keep_data_and_labels_aligned():
    shape = (36, 36, 36)
    rs = np.random.RandomState(3)
    im = rs.randn(*shape).astype(np.float32)
    ll = (rs.rand(*shape) > 0.5).astype(np.uint8)
    mm = np.ones(shape, np.uint8)
    ww = np.zeros(shape, np.float32)
    ww[18, 18, 18] = 1.0                   # the only label-ll[18,18,18] voxel with weight
    ww[17, 19, 16] = 3.0
    ll[18, 18, 18], ll[17, 19, 16] = 0, 1
    gen = fplobjdetect.gen_volume2([(im, ll, mm, ww)], (24, 24, 24), 4, 0.5,
                                   noise_aug=[0, 0], rng=np.random.RandomState(5))
    centres = {0: (18, 18, 18), 1: (17, 19, 16)}
    for _ in range(20):
        data, labels = next(gen)
        for ex in range(4):
            # whatever the augmentation, the patch is a rotation/flip of one of the two
            # weighted centres' patches, with the label block transformed the same way
            found = False
            for cc, (z, y, x) in centres.items():
                p = im[z - 12:z + 12, y - 12:y + 12, x - 12:x + 12]
                q = ll[z - 3:z + 3, y - 3:y + 3, x - 3:x + 3]
                for rot in range(4):
                    for ref in (0, 1):
                        for fpz in (0, 1):
                            a, b = np.rot90(p, rot, (1, 2)), np.rot90(q, rot, (1, 2))
                            if ref:
                                a, b = np.fliplr(a), np.fliplr(b)
                            if fpz:
                                a, b = np.flipud(a), np.flipud(b)
                            if np.array_equal(a, data[ex, ..., 0]):
                                assert np.array_equal(b, labels[ex, ..., 0])
                                found = True
            assert found


def test_voxel_loss_formula_and_sampling_weight_files(tmp_path):
    """hand-derived values: masked voxel -> 0; confident negative (loss < 0.005) ->
    0; clamping to the thresholds; border of rf_size/2 excluded"""
    net = FplNetwork.__new__(FplNetwork)
    net.rf_size = (4, 4, 4)
    shape = (10, 10, 10)
    pred = np.full(shape, 0.5, np.float32)
    ll = np.zeros(shape, np.uint8)
    mm = np.ones(shape, np.uint8)
    pred[5, 5, 5], ll[5, 5, 5] = 0.25, 1            # positive: -log(0.25)
    pred[5, 5, 6] = 0.001                          # negative, -log(0.999) = 0.001 < 0.005 -> 0
    pred[5, 6, 5] = 0.9                            # negative: -log(0.1)
    mm[6, 5, 5] = 0                                # masked
    net.infer = lambda image, normalize=None: pred
    got = net.voxel_loss(None, (ll, mm))
    assert got.dtype == np.float32 and got.shape == shape
    assert abs(got[5, 5, 5] - (-np.log(0.25))) < 1e-6
    assert got[5, 5, 6] == 0 and got[6, 5, 5] == 0
    assert abs(got[5, 6, 5] - (-np.log(0.1))) < 1e-6
    assert abs(got[4, 4, 4] - (-np.log(0.5))) < 1e-6
    assert not got[:2].any() and not got[:, :, -2:].any()      # rf border
    clamped = net.voxel_loss(None, (ll, mm), l0_thresh=(0.8, 1.0), l1_thresh=(0.1, 0.5))
    assert abs(clamped[5, 6, 5] - 1.0) < 1e-6 and abs(clamped[4, 4, 4] - 0.8) < 1e-6
    assert abs(clamped[5, 5, 5] - 0.5) < 1e-6
    aug = fplobjdetect.write_sampling_weights([(np.zeros(shape), ll, mm)], net,
                                              str(tmp_path / 'w'), None, None)
    assert aug[0][3].endswith('w00.npy') and np.array_equal(np.load(aug[0][3]), got)
    # and the weights file feeds gen_volume2
    im = np.zeros((30, 30, 30), np.float32)
    l2 = np.zeros((30, 30, 30), np.uint8); l2[15, 15, 15] = 1
    w2 = np.ones((30, 30, 30), np.float32)
    np.save(str(tmp_path / 'w2.npy'), w2)
    d, lab = next(fplobjdetect.gen_volume2([(im, l2, np.ones_like(l2), str(tmp_path / 'w2.npy'))],
                                           (24, 24, 24), 2, 0.5, rng=np.random.RandomState(0)))
    assert d.shape == (2, 24, 24, 24, 1) and (lab == 1).any() and lab.max() <= 2


def test_gen_volume_cycles_volumes_and_respects_ratio():
    shape = (40, 40, 40)
    ims = [np.full(shape, float(v), np.float32) for v in (1, 2, 3)]
    lls = []
    for v in range(3):
        ll = np.zeros(shape, np.uint8)
        if v != 2:                          # volume 2 has no positives -> negatives only
            ll[18:22, 18:22, 18:22] = 1
        lls.append(ll)
    mm = np.ones(shape, np.uint8)
    gen = fplobjdetect.gen_volume([(ims[v], lls[v], mm) for v in range(3)], (24, 24, 24), 6,
                                  0.25, rng=np.random.RandomState(2))
    n_pos = {0: 0, 1: 0, 2: 0}
    for _ in range(50):
        data, labels = next(gen)
        assert data.shape == (6, 24, 24, 24, 1) and labels.shape == (6, 6, 6, 6, 1)
        for ex in range(6):
            vol = int(data[ex, 0, 0, 0, 0]) - 1
            assert vol == ex % 3                       # example i <- volume i mod 3
            centre = labels[ex, 2:4, 2:4, 2:4, 0]      # the centre voxel survives flips here
            n_pos[vol] += int(centre.max() == 1)
    assert n_pos[2] == 0 and 55 < n_pos[0] < 95 and 55 < n_pos[1] < 95     # ~75 % of 100
    assert fplobjdetect.get_out_sz(18) == 6 and fplobjdetect.get_out_sz(24) == 10   # unet_like sizes


# ---- the reference's own generators, pinned (tests/golden/training_generators.npz holds
#      the outputs of flypylib's gen_batches / gen_volume / gen_volume2 on these inputs,
#      generated by tests/golden/make_golden.py::gen_training_generators) ---------------
def _training_volumes(seed, shape):
    """same inputs as make_golden.py::training_volumes"""
    from flypylib_amd import synth
    im = synth.em_volume_u8(seed, shape).astype(np.float32)
    ll = (synth.hash_uniform_f32(seed + 100, shape) > np.float32(0.97)).astype(np.uint8)
    mm = np.ones(shape, np.uint8)
    mm[: shape[0] // 3, : shape[1] // 2, :] = 0
    return im, ll, mm


@pytest.mark.parametrize('name,n', [('batches', 3), ('batches_mask', 3), ('volume', 4),
                                    ('volume2', 3), ('volume2_noise', 3)])
def test_generators_match_the_reference_outputs(tmp_path, name, n):
    """inputs as .h5 files (read back by the package's own HDF5 reader), the global numpy
    RNG seeded as in the fixture run: batches bit-identical to the reference's"""
    import hashlib
    from flypylib_amd import keras_io
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'training_generators.npz'))
    train = []
    for v, (shape, seed) in enumerate(zip(gold['shapes'], gold['vol_seeds'])):
        im, ll, mm = _training_volumes(int(seed), tuple(int(d) for d in shape))
        keras_io.write_main(str(tmp_path / ('im%d.h5' % v)), im)
        keras_io.write_main(str(tmp_path / ('v%d_labels.h5' % v)), ll)
        keras_io.write_main(str(tmp_path / ('v%d_mask.h5' % v)), mm)
        train.append((str(tmp_path / ('im%d.h5' % v)), str(tmp_path / ('v%d_' % v))))
    make = {
        'batches': lambda: fplobjdetect.gen_batches(train, (12, 10, 10), 6),
        'batches_mask': lambda: fplobjdetect.gen_batches(train, (12, 12, 12), 4, True),
        'volume': lambda: fplobjdetect.gen_volume(train, (24, 24, 24), 3, 0.5),
        'volume2': lambda: fplobjdetect.gen_volume2(train, (24, 24, 24), 3, 0.5),
        'volume2_noise': lambda: fplobjdetect.gen_volume2(train, (24, 24, 24), 2, 0.3,
                                                           noise_aug=[0.05, 0.1]),
    }[name]
    np.random.seed(int(gold['%s_seed' % name]))
    gen = make()
    for i in range(n):
        d, lab = next(gen)
        assert d.dtype == np.float32 and lab.dtype == np.uint8
        assert np.array_equal(lab, gold['%s_labels_%d' % (name, i)]), (name, i)
        if i == 0:
            assert np.array_equal(d[0, ..., 0], gold['%s_example0' % name]), name
        got = hashlib.sha256(np.ascontiguousarray(d).tobytes()).hexdigest()
        assert got == str(gold['%s_data_sha_%d' % (name, i)]), (name, i)
