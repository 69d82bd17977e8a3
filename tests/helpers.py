"""shared test helpers: golden-case input generators (same as make_golden.py)"""
import hashlib

import numpy as np

from flypylib_amd import synth


def make_pred(kind, seed, shape):
    if kind == 'uniform':
        return synth.hash_uniform_f32(seed, shape)
    if kind == 'blobs':
        return synth.blob_prob_volume(seed, shape)
    if kind == 'zeros':
        return np.zeros(shape, np.float32)
    if kind == 'plateau':
        v = np.zeros(shape, np.float32)
        for z in range(8, shape[0] - 8, 16):
            for y in range(8, shape[1] - 8, 16):
                for x in range(8, shape[2] - 8, 16):
                    v[z:z + 4, y:y + 4, x:x + 4] = 0.75
        return v
    raise ValueError(kind)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def v2o_cases(g):
    """yield dicts for every case of tests/golden/voxel2obj.npz"""
    for name in [str(n) for n in g['names']]:
        p = g[name + '_params']
        st = g[name + '_sigma_thd']
        thd = float(st[1])
        yield dict(name=name, kind=str(g[name + '_kind']), seed=int(p[0]),
                   shape=tuple(int(v) for v in p[1:4]), r=int(p[4]),
                   sigma=float(st[0]),
                   thd=int(thd) if thd == int(thd) else thd,
                   buffer=tuple(int(v) for v in g[name + '_buffer']),
                   offset=tuple(int(v) for v in g[name + '_offset']),
                   locs=g[name + '_locs'], conf=g[name + '_conf'],
                   pred_sha=str(g[name + '_pred_sha']),
                   smooth_sha=str(g[name + '_smooth_sha']),
                   pct97=g[name + '_pct97'])


def crop_identity(x, off):
    return x[:, off[0]:x.shape[1] - off[0], off[1]:x.shape[2] - off[1],
             off[2]:x.shape[3] - off[2], :].astype(np.float32)


def coarse4(x, off):
    c = x[:, off[0]:x.shape[1] - off[0], off[1]:x.shape[2] - off[1],
          off[2]:x.shape[3] - off[2], :]
    c = c[:, ::4, ::4, ::4, :]
    for ax in (1, 2, 3):
        c = np.repeat(c, 4, axis=ax)
    return c.astype(np.float32)


FAKE_NETS = {'_crop_identity': crop_identity, '_coarse4': coarse4}


# the segmentation-aware voxel2obj cases of tests/golden/voxel2obj_seg.npz
# (name, pred kind, pred seed, shape, r, sigma, thd, buffer, seg seed, n_sites, tiny,
#  seg_dilate, seg_sz_thd, seg_force) - as in tests/golden/make_golden.py
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



# float64 predictions (tests/golden/voxel2obj_f64.npz): the reference smooths, thresholds
# and compares in the array's own dtype.
# (name, kind, seed, shape, r, sigma, thd, buffer, offset, seg: None | (seed, n_sites, tiny,
#  seg_dilate, seg_sz_thd, seg_force))
V2O_F64_CASES = [
    ('f64_blobs', 'blobs', 61, (64, 60, 56), 9, 3.0, 0.05, 2, (0, 0, 0), None),
    ('f64_uniform', 'uniform', 62, (40, 44, 48), 6, 1.5, 0.1, (1, 2, 3), (10, 20, 30), None),
    ('f64_pct', 'uniform', 63, (36, 36, 36), 5, 2.0, 0, 0, (0, 0, 0), None),
    ('f64_seg', 'blobs', 64, (56, 48, 52), 9, 2.0, 0.05, 0, (0, 0, 0), (6, 8, 30, 1, 60, 3)),
]


def make_pred_f64(kind, seed, shape):
    """a float64 volume that is NOT float32-representable: the float32 generator's values
    plus a hashed perturbation of relative size 1e-9"""
    base = make_pred(kind, seed, shape).astype(np.float64)
    return base * (1.0 + 1e-9 * synth.hash_uniform_f32(seed + 7000, shape).astype(np.float64))


# integer predictions (name, generator, seed, shape, dtype, scale, r, sigma, thd, buffer, offset, seg)
V2O_INT_CASES = [
    ('u8_blobs', 'blobs', 71, (60, 56, 64), 'uint8', 255, 9, 2.0, 20, 2, (0, 0, 0), None),
    ('u8_pct', 'uniform', 72, (40, 44, 48), 'uint8', 255, 5, 1.5, 0, 0, (0, 0, 0), None),
    ('i16_blobs', 'blobs', 73, (48, 52, 50), 'int16', 1000, 7, 3.0, 100, (1, 2, 3), (5, 6, 7), None),
    ('i32_seg', 'blobs', 74, (56, 48, 52), 'int32', 100000, 9, 2.0, 5000, 0, (0, 0, 0), (6, 8, 30, 1, 60, 3)),
]


def make_pred_int(kind, seed, shape, dtype, scale):
    """an integer volume: the float32 generator's values scaled and rounded"""
    return np.round(make_pred(kind, seed, shape).astype(np.float64) * scale).astype(dtype)


def same_detections(a, b, conf_tol, tie=2e-6):
    """Two `voxel2obj` / pipeline results name the same point SET (the north star's gate) with
    confidences within conf_tol, in the same descending-confidence order wherever that order
    is decided by more than `tie`: two peaks whose fp32 confidences differ by a few 1e-8 may
    swap places between two correct implementations of the CNN (seen on the 582^3 trained
    substack: 0.11498727 / 0.11498725)."""
    assert len(a['conf']) == len(b['conf']), (len(a['conf']), len(b['conf']))
    la, lb = np.asarray(a['locs']), np.asarray(b['locs'])
    ia, ib = np.lexsort(la.T[::-1]), np.lexsort(lb.T[::-1])
    assert np.array_equal(la[ia], lb[ib]), 'different point sets'
    np.testing.assert_allclose(np.asarray(a['conf'])[ia], np.asarray(b['conf'])[ib], rtol=0, atol=conf_tol)
    moved = np.where((la != lb).any(axis=1))[0]
    for i in moved:
        assert abs(a['conf'][i] - b['conf'][i]) < tie, (i, a['conf'][i], b['conf'][i])
    return len(moved)
