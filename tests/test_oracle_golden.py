"""CPU: the oracle against the golden vectors produced by the reference's own
code (tests/golden/make_golden.py), and the host-side logic of the product."""
import numpy as np
import pytest

from flypylib_amd import fplobjdetect, fplutils, synth
from oracle import infer_oracle, voxel2obj_oracle
from tests import helpers


def test_set_filter_matches_reference(golden):
    g = golden('set_filter.npz')
    for r in (1, 3, 7, 27):
        f = fplutils.set_filter(r)
        assert f.shape == (2 * r + 1,) * 3 and f.dtype == bool
        assert int(f.sum()) == int(g['r%d_count' % r])
        assert helpers.sha(f.astype(np.uint8)) == str(g['r%d_sha' % r])
    assert np.array_equal(fplutils.set_filter(3), g['r3_mask'])
    assert int(g['r3_count']) == 123 and int(g['r27_count']) == 82519
    inside, dist = fplutils.set_filter(2, return_dist=True)
    assert np.array_equal(inside, dist <= 2)


def test_to3d():
    assert fplutils.to3d(5) == (5, 5, 5)
    assert fplutils.to3d((1, 2, 3)) == (1, 2, 3)
    assert fplutils.to3d(None) == (None, None, None)


@pytest.mark.parametrize('use_scipy', [True, False])
def test_voxel2obj_oracle_matches_reference(golden, use_scipy):
    g = golden('voxel2obj.npz')
    n = 0
    for c in helpers.v2o_cases(g):
        pred = helpers.make_pred(c['kind'], c['seed'], c['shape'])
        assert helpers.sha(pred) == c['pred_sha'], 'input generator drifted'
        res = voxel2obj_oracle.voxel2obj(pred, c['r'], c['sigma'], c['offset'],
                                         c['buffer'], c['thd'],
                                         use_scipy=use_scipy)
        assert np.array_equal(res['locs'], c['locs']), c['name']
        assert np.array_equal(res['conf'], c['conf']), c['name']
        assert res['locs'].dtype == np.float64 and res['locs'].shape[1] == 3
        n += 1
    assert n == 10


def test_gaussian_restatement_is_bit_identical_to_scipy(golden):
    g = golden('voxel2obj.npz')
    for c in helpers.v2o_cases(g):
        pred = helpers.make_pred(c['kind'], c['seed'], c['shape'])
        sm = voxel2obj_oracle.smooth_and_clear(pred, c['r'], c['sigma'],
                                               use_scipy=False)
        assert helpers.sha(sm) == c['smooth_sha'], c['name']
        assert np.percentile(sm, 97) == c['pct97']


def test_percentile_restatement_matches_numpy():
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 100, 1001, 65537, 214 ** 3):
        a = rng.random(n).astype(np.float32)
        if n > 50:
            a[:n // 2] = 0
        s = np.sort(a)
        for q in (97, 50, 0, 100, 12.5):
            lo, hi, gamma = fplobjdetect.percentile_plan(n, q)
            got = fplobjdetect.percentile_lerp(s[lo], s[hi], gamma)
            ref = np.percentile(a, q)
            assert got == ref and got.dtype == ref.dtype, (n, q)


def test_gaussian_kernel_matches_oracle():
    for sigma in (1.5, 2.0, 5.0, 0.7):
        w = fplobjdetect.gaussian_kernel1d(sigma)
        wo, radius = voxel2obj_oracle.gaussian_weights(sigma)
        assert np.array_equal(w, wo) and w.size == 2 * radius + 1


def test_infer_lattice_oracle_matches_reference(golden):
    g = golden('infer_lattice.npz')
    for name in [str(n) for n in g['names']]:
        shape = tuple(int(v) for v in g[name + '_shape'])
        isz = tuple(int(v) for v in g[name + '_isz'])
        off = tuple(int(v) for v in g[name + '_off'])
        fn = helpers.FAKE_NETS[str(g[name + '_fn'])]
        img = synth.hash_uniform_f32(int(g[name + '_seed']), shape)
        pred = infer_oracle.infer_lattice(img, isz, off, lambda b: fn(b, off),
                                          n_gpu=int(g[name + '_n_gpu']))
        assert pred.dtype == np.float32 and pred.shape == shape
        assert helpers.sha(pred) == str(g[name + '_pred_sha']), name
        assert np.array_equal(pred[::7, ::5, ::3], g[name + '_pred_sample'])
        # border shell of width off stays zero
        assert not pred[:off[0]].any() and not pred[:, :, -off[2]:].any()


def test_synth_volume_is_origin_consistent():
    a = synth.em_volume_u8(7, (40, 50, 70), (64, 10, 30))
    b = synth.em_volume_u8(7, (130, 70, 110))
    assert np.array_equal(b[64:104, 10:60, 30:100], a)
    assert a.dtype == np.uint8 and 100 < a.mean() < 150


import pytest  # noqa: E402


@pytest.mark.parametrize('case', helpers.V2O_SEG_CASES, ids=[c[0] for c in helpers.V2O_SEG_CASES])
def test_segmentation_aware_voxel2obj_oracle_matches_reference(golden, case):
    """the seg / seg_dilate / seg_sz_thd / seg_force branch (reference :161-224)"""
    from flypylib_amd import synth
    from oracle import voxel2obj_oracle
    g = golden('voxel2obj_seg.npz')
    name, kind, pseed, shape, r, sigma, thd, buf, sseed, n_sites, tiny, dil, szt, force = case
    pred = helpers.make_pred(kind, pseed, shape)
    seg = synth.voronoi_segmentation(sseed, shape, n_sites, tiny)
    assert helpers.sha(pred) == str(g[name + '_pred_sha'])
    assert helpers.sha(seg) == str(g[name + '_seg_sha'])
    res = voxel2obj_oracle.voxel2obj(pred, r, sigma, (0, 0, 0), buf, thd, seg=seg,
                                     seg_dilate=dil, seg_sz_thd=szt, seg_force=force)
    assert np.array_equal(res['locs'], g[name + '_locs'])
    assert np.array_equal(res['conf'], g[name + '_conf'])


@pytest.mark.parametrize('case', helpers.V2O_F64_CASES, ids=[c[0] for c in helpers.V2O_F64_CASES])
def test_float64_voxel2obj_oracle_matches_reference(golden, case):
    """float64 predictions: the reference pads, smooths, thresholds and compares in the
    input's own dtype - the oracle (numpy / scipy on the float64 array) gives the reference's
    point lists, which differ from those of the float32-rounded input"""
    g = golden('voxel2obj_f64.npz')
    name, kind, seed, shape, r, sigma, thd, buf, off, segp = case
    pred = helpers.make_pred_f64(kind, seed, shape)
    assert helpers.sha(pred) == str(g[name + '_pred_sha'])
    kw = {}
    if segp is not None:
        sseed, n_sites, tiny, dil, szt, force = segp
        kw = dict(seg=synth.voronoi_segmentation(sseed, shape, n_sites, tiny), seg_dilate=dil,
                  seg_sz_thd=szt, seg_force=force)
    res = voxel2obj_oracle.voxel2obj(pred, r, sigma, tuple(off), buf, thd, **kw)
    assert np.array_equal(res['locs'], g[name + '_locs'])
    assert np.array_equal(res['conf'], g[name + '_conf'])
    f32 = voxel2obj_oracle.voxel2obj(pred.astype(np.float32), r, sigma, tuple(off), buf, thd, **kw)
    assert not np.array_equal(f32['conf'], res['conf'])


@pytest.mark.parametrize('case', helpers.V2O_INT_CASES, ids=[c[0] for c in helpers.V2O_INT_CASES])
def test_integer_voxel2obj_oracle_matches_reference(golden, case):
    """integer predictions: scipy's filter truncates back to the integer type after every axis,
    the reference's rows come back as int64 - the oracle gives the reference's point lists"""
    g = golden('voxel2obj_int.npz')
    name, kind, seed, shape, dtype, scale, r, sigma, thd, buf, off, segp = case
    pred = helpers.make_pred_int(kind, seed, shape, dtype, scale)
    assert helpers.sha(pred) == str(g[name + '_pred_sha'])
    kw = {}
    if segp is not None:
        sseed, n_sites, tiny, dil, szt, force = segp
        kw = dict(seg=synth.voronoi_segmentation(sseed, shape, n_sites, tiny), seg_dilate=dil,
                  seg_sz_thd=szt, seg_force=force)
    res = voxel2obj_oracle.voxel2obj(pred, r, sigma, tuple(off), buf, thd, **kw)
    assert res['locs'].dtype == g[name + '_locs'].dtype == np.int64
    assert np.array_equal(res['locs'], g[name + '_locs'])
    assert np.array_equal(res['conf'], g[name + '_conf'])
