"""GPU parity for voxel2obj: bit-exact against the reference's golden point lists
and against the CPU oracle on larger seeded volumes."""
import numpy as np
import pytest

from flypylib_amd import fplobjdetect, synth
from oracle import voxel2obj_oracle
from tests import helpers

pytestmark = pytest.mark.gpu


def test_golden_cases_bit_exact(ctx, golden):
    g = golden('voxel2obj.npz')
    n = 0
    for c in helpers.v2o_cases(g):
        pred = helpers.make_pred(c['kind'], c['seed'], c['shape'])
        # the smoothed volume is read between the stages: the NMS consumes it
        pdims = tuple(s + 2 * c['r'] for s in c['shape'])
        ctx.v2o_smooth(pred, c['shape'], c['r'],
                       fplobjdetect.gaussian_kernel1d(c['sigma'], truncate=2.0), [])
        sm = ctx.v2o_smoothed(pdims)
        assert helpers.sha(sm) == c['smooth_sha'], \
            '%s: smoothed volume differs from scipy' % c['name']
        res, info = fplobjdetect.voxel2obj(pred, c['r'], c['sigma'], c['offset'],
                                           c['buffer'], c['thd'],
                                           return_info=True)
        with pytest.raises(Exception, match='no smoothed volume'):
            ctx.v2o_smoothed(pdims)
        assert np.array_equal(res['locs'], c['locs']), c['name']
        assert np.array_equal(res['conf'], c['conf']), c['name']
        assert res['locs'].dtype == np.float64 and res['conf'].dtype == np.float64
        n += 1
    assert n == 10


@pytest.mark.parametrize('shape,r,sigma,thd,buf', [
    ((120, 110, 130), 27, 5.0, 0, 0),
    ((160, 160, 160), 27, 5.0, 0.02, 30),
    ((90, 140, 75), 9, 2.0, 0.2, (3, 4, 5)),
    ((64, 64, 200), 15, 3.0, 0, 0),
    # radii beyond the tabulated fast paths: window of 2 * 10 + 1 cells (plain window-max
    # kernels), 81^2 cube rows per ball (untabulated row geometry), 23^3 cells per ball
    ((120, 100, 130), 40, 3.0, 0.05, 0),
    ((70, 150, 90), 33, 2.0, 0, 2)])
def test_matches_oracle_on_seeded_volumes(ctx, shape, r, sigma, thd, buf):
    pred = synth.blob_prob_volume(77, shape, period=32, radius=7.0)
    ref = voxel2obj_oracle.voxel2obj(pred, r, sigma, (3, 2, 1), buf, thd)
    got, info = fplobjdetect.voxel2obj(pred, r, sigma, (3, 2, 1), buf, thd,
                                       return_info=True)
    assert len(ref['conf']) > 3
    assert np.array_equal(got['locs'], ref['locs'])
    assert np.array_equal(got['conf'], ref['conf'])
    assert info['rounds'] >= 1


def test_dense_noise_many_rounds(ctx):
    """uniform noise, small radius: thousands of detections over many NMS rounds, the
    percentile (not thd) as threshold, a first-level radix bin that holds a large part of
    the volume - the regime the substack pipeline's random-weight predictions are in"""
    pred = synth.hash_uniform_f32(91, (96, 100, 104))
    for r, sigma, thd in ((5, 1.5, 0), (4, 2.0, 0.2)):
        ref = voxel2obj_oracle.voxel2obj(pred, r, sigma, (0, 0, 0), 0, thd)
        got, info = fplobjdetect.voxel2obj(pred, r, sigma, (0, 0, 0), 0, thd, return_info=True)
        assert len(ref['conf']) > 1000 and info['rounds'] >= 5
        assert np.array_equal(got['locs'], ref['locs'])
        assert np.array_equal(got['conf'], ref['conf'])
    # the same volume, half of it lifted into [0.5, 0.75): ONE first-level bin holds more
    # than 1/8 of the voxels (the histogram-first order of the radix select)
    lifted = pred.copy()
    lifted[:48] = np.float32(0.5) + lifted[:48] * np.float32(0.25)
    ref = voxel2obj_oracle.voxel2obj(lifted, 5, 1.5, (0, 0, 0), 0, 0)
    got = fplobjdetect.voxel2obj(lifted, 5, 1.5, (0, 0, 0), 0, 0)
    assert np.array_equal(got['locs'], ref['locs']) and np.array_equal(got['conf'], ref['conf'])


def test_order_statistics_exact(ctx):
    pred = synth.hash_uniform_f32(5, (50, 60, 70))
    r, sigma = 6, 2.0
    pdims = tuple(s + 2 * r for s in pred.shape)
    n = int(np.prod(pdims))
    ranks = [0, 1, n // 3, n // 2, int(0.97 * (n - 1)), n - 2, n - 1]
    vals = ctx.v2o_smooth(pred, pred.shape, r,
                          fplobjdetect.gaussian_kernel1d(sigma), ranks)
    sm = voxel2obj_oracle.smooth_and_clear(pred, r, sigma)
    s = np.sort(sm.reshape(-1))
    assert np.array_equal(vals, s[ranks])


def test_order_statistics_on_adversarial_values(ctx):
    """the radix select itself, fed through an identity smoothing (one tap of weight 1):
    negatives, signed zeros, denormals, every edge k / 1024 of the linear first-level
    bins with its two float neighbours, powers of two down to 2^-60, values >= 1, inf"""
    rng = np.random.default_rng(7)
    k = np.arange(0, 1025, dtype=np.float32) / np.float32(1024)
    special = np.concatenate([
        k, np.nextafter(k, np.float32(-1)), np.nextafter(k, np.float32(2)),
        np.float32(2.0) ** -np.arange(0, 61, dtype=np.float32),
        -np.float32(2.0) ** -np.arange(0, 61, dtype=np.float32),
        np.array([0.0, -0.0, 1e-45, -1e-45, 1e-39, 1e30, -1e30, np.inf, -np.inf, 1.0, 1.5],
                 np.float32)])
    shape = (24, 50, 60)
    n = int(np.prod(shape))
    vals = rng.uniform(-1.0, 2.0, n).astype(np.float32)
    vals[rng.choice(n, n // 3, replace=False)] = rng.uniform(0.5, 0.75, n // 3).astype(np.float32)
    pos = rng.choice(n, special.size, replace=False)
    vals[pos] = special
    pred = vals.reshape(shape)
    s = np.sort(vals)
    ranks = sorted(set([0, 1, n - 2, n - 1] + [int(x) for x in rng.integers(0, n, 40)]
                       + [int(np.searchsorted(s, v)) for v in special[::37]]))
    got = ctx.v2o_smooth(pred, shape, 0, np.array([1.0]), ranks)
    assert np.array_equal(got, s[ranks])


# (seed, shape, r, sigma): volumes on which a smoothing pass that fuses a * b + c into
# one rounding differs from scipy (x86-64: two roundings) in at least one float32 voxel
# - found with tools/dev/find_fma_witness.py for two fused forms (every product fused;
# the form hipcc's default contraction produces).  The build never shipped a fused pass
# (-ffp-contract=on in csrc/build.py, the pragma in v2o.hip; tests/test_host_logic.py
# disassembles and checks); these are the parity cases that would notice one.
FMA_WITNESSES = [
    (5261, (36, 40, 44), 10, 5.0), (5909, (36, 40, 44), 10, 5.0),
    (232, (40, 36, 44), 6, 3.0), (12132, (40, 36, 44), 6, 3.0),
    (6085, (44, 40, 36), 4, 2.0), (13566, (44, 40, 36), 4, 2.0),
    (647, (40, 44, 36), 3, 1.5), (6023, (40, 44, 36), 3, 1.5),
]


@pytest.mark.parametrize('seed,shape,r,sigma', FMA_WITNESSES)
def test_smoothing_rounds_products_and_sums_separately(ctx, seed, shape, r, sigma, monkeypatch):
    pred = synth.hash_uniform_f32(seed, shape)
    want_r = voxel2obj_oracle.smooth_and_clear(pred, r, sigma)
    w = fplobjdetect.gaussian_kernel1d(sigma)
    for env in (None, 'FPL_V2O_UNFUSED'):
        if env:
            monkeypatch.setenv(env, '1')
        ctx.v2o_smooth(pred, pred.shape, r, w, [])
        assert np.array_equal(ctx.v2o_smoothed(want_r.shape), want_r), env
        if env:
            monkeypatch.delenv(env)
    # the witness voxel may sit in the margin a radius r zeroes: run unpadded as well
    padded = np.pad(pred, r, 'constant')
    ctx.v2o_smooth(padded, padded.shape, 0, w, [])
    assert np.array_equal(ctx.v2o_smoothed(padded.shape),
                          voxel2obj_oracle.smooth_and_clear(padded, 0, sigma))


def test_order_statistics_with_a_floor(ctx):
    """fpl_v2o_set_floor: statistics whose first radix bin lies below the floor's are
    reported as the floor (the caller takes max(statistic, floor)); a rank at or above
    that bin makes every value exact again; the threshold voxel2obj derives is the
    oracle's in both regimes"""
    pred = synth.blob_prob_volume(31, (80, 70, 90), period=32, radius=6.0)
    r, sigma = 10, 3.0
    w = fplobjdetect.gaussian_kernel1d(sigma)
    n = int(np.prod([d + 2 * r for d in pred.shape]))
    s = np.sort(voxel2obj_oracle.smooth_and_clear(pred, r, sigma).reshape(-1))
    lo, hi, gamma = fplobjdetect.percentile_plan(n, 97, np.float32)
    p97 = float(s[hi])
    assert p97 > 0
    # floor far above the percentile: shortcut
    big = np.float32(s[-1] / 2)                # between the percentile and the maximum,
    assert p97 * 2 < big                       # a power of two away from both
    ctx.v2o_set_floor(big)
    assert np.array_equal(ctx.v2o_smooth(pred, pred.shape, r, w, [lo, hi]), [big, big])
    # floor below the percentile: exact
    ctx.v2o_set_floor(np.float32(p97 / 64))
    assert np.array_equal(ctx.v2o_smooth(pred, pred.shape, r, w, [lo, hi]), s[[lo, hi]])
    # one rank below the floor's bin, one above: exact for both
    ctx.v2o_set_floor(big)
    assert np.array_equal(ctx.v2o_smooth(pred, pred.shape, r, w, [lo, n - 1]), s[[lo, n - 1]])
    # the floor holds for one smoothing only
    assert np.array_equal(ctx.v2o_smooth(pred, pred.shape, r, w, [lo, hi]), s[[lo, hi]])
    for thd in (float(big), p97 / 64):
        ref = voxel2obj_oracle.voxel2obj(pred, r, sigma, thd=thd)
        got, info = fplobjdetect.voxel2obj(pred, r, sigma, thd=thd, return_info=True)
        assert np.array_equal(got['locs'], ref['locs'])
        assert np.array_equal(got['conf'], ref['conf'])
        assert float(info['thresh']) == max(
            float(fplobjdetect.percentile_lerp(s[lo], s[hi], gamma)), thd)


def test_negative_and_empty_inputs(ctx):
    # all-negative predictions: threshold < 0 but nothing positive -> no points
    pred = -synth.hash_uniform_f32(6, (30, 30, 30)) - np.float32(0.1)
    ref = voxel2obj_oracle.voxel2obj(pred, 5, 2.0)
    got = fplobjdetect.voxel2obj(pred, 5, 2.0)
    assert got['locs'].shape == ref['locs'].shape == (0, 3)
    # a segmentation of all-negative predictions changes nothing: still no points
    got = fplobjdetect.voxel2obj(pred, 5, 2.0, seg=np.zeros((30, 30, 30), np.uint64), seg_dilate=2)
    assert got['locs'].shape == (0, 3)
    # ... and in float64 (no candidate above max(percentile, 0))
    got = fplobjdetect.voxel2obj(pred.astype(np.float64), 5, 2.0)
    assert got['locs'].shape == (0, 3)


@pytest.mark.parametrize('case', helpers.V2O_SEG_CASES, ids=[c[0] for c in helpers.V2O_SEG_CASES])
def test_segmentation_aware_golden_cases_bit_exact(ctx, golden, case):
    """seg / seg_dilate / seg_sz_thd / seg_force (reference :161-224): the reference's
    own point lists"""
    g = golden('voxel2obj_seg.npz')
    name, kind, pseed, shape, r, sigma, thd, buf, sseed, n_sites, tiny, dil, szt, force = case
    pred = helpers.make_pred(kind, pseed, shape)
    seg = synth.voronoi_segmentation(sseed, shape, n_sites, tiny)
    res = fplobjdetect.voxel2obj(pred, r, sigma, (0, 0, 0), buf, thd, seg=seg, seg_dilate=dil,
                                 seg_sz_thd=szt, seg_force=force)
    assert np.array_equal(res['locs'], g[name + '_locs']), name
    assert np.array_equal(res['conf'], g[name + '_conf']), name


def test_segmentation_aware_matches_oracle_at_pipeline_parameters(ctx):
    """fri_postprocess's call (reference :1143-1150): r 27, sigma 5, seg_dilate 8,
    seg_sz_thd 5000, seg_force 10, uint32 labels"""
    shape = (150, 140, 160)
    pred = synth.blob_prob_volume(78, shape, period=32, radius=7.0)
    seg = (synth.voronoi_segmentation(9, shape, 40, 25) % np.uint64(2 ** 31)).astype(np.uint32)
    kw = dict(seg_dilate=8, seg_sz_thd=5000, seg_force=10)
    ref = voxel2obj_oracle.voxel2obj(pred, 27, 5.0, (5, 6, 7), 10, 0.05, seg=seg, **kw)
    got = fplobjdetect.voxel2obj(pred, 27, 5.0, (5, 6, 7), 10, 0.05, seg=seg, **kw)
    plain = fplobjdetect.voxel2obj(pred, 27, 5.0, (5, 6, 7), 10, 0.05)
    assert len(ref['conf']) > len(plain['conf']) > 10
    assert np.array_equal(got['locs'], ref['locs']) and np.array_equal(got['conf'], ref['conf'])
    with pytest.raises(ValueError):
        fplobjdetect.voxel2obj(pred, 27, 5.0, seg_sz_thd=10)


@pytest.mark.parametrize('case', helpers.V2O_F64_CASES, ids=[c[0] for c in helpers.V2O_F64_CASES])
def test_float64_predictions_match_the_reference_bit_for_bit(ctx, golden, case):
    """a float64 `pred` is padded, smoothed (no rounding between the axes), thresholded and
    compared in float64, as the reference does for its input's dtype
    (fplobjdetect.py:158-231): the reference's own point lists and float64 confidences,
    with and without a segmentation"""
    g = golden('voxel2obj_f64.npz')
    name, kind, seed, shape, r, sigma, thd, buf, off, segp = case
    pred = helpers.make_pred_f64(kind, seed, shape)
    kw = {}
    if segp is not None:
        sseed, n_sites, tiny, dil, szt, force = segp
        kw = dict(seg=synth.voronoi_segmentation(sseed, shape, n_sites, tiny), seg_dilate=dil,
                  seg_sz_thd=szt, seg_force=force)
    res = fplobjdetect.voxel2obj(pred, r, sigma, tuple(off), buf, thd, **kw)
    assert res['conf'].dtype == np.float64
    assert np.array_equal(res['locs'], g[name + '_locs']), name
    assert np.array_equal(res['conf'], g[name + '_conf']), name
    # the float32 pipeline still works afterwards (the context leaves float64 mode)
    again = fplobjdetect.voxel2obj(pred.astype(np.float32), r, sigma, tuple(off), buf, thd, **kw)
    assert len(again['conf']) == len(res['conf']) and not np.array_equal(again['conf'], res['conf'])


@pytest.mark.parametrize('case', helpers.V2O_INT_CASES, ids=[c[0] for c in helpers.V2O_INT_CASES])
def test_integer_predictions_match_the_reference_bit_for_bit(ctx, golden, case):
    """an integer `pred` (uint8 / int16 / int32) is smoothed in its own type - scipy
    accumulates in float64 and truncates back to the integer type after every axis pass - then
    thresholded by a float64 percentile; the reference returns int64 rows
    (fplobjdetect.py:158-236): its own point lists, with and without a segmentation"""
    g = golden('voxel2obj_int.npz')
    name, kind, seed, shape, dtype, scale, r, sigma, thd, buf, off, segp = case
    pred = helpers.make_pred_int(kind, seed, shape, dtype, scale)
    assert helpers.sha(pred) == str(g[name + '_pred_sha'])
    kw = {}
    if segp is not None:
        sseed, n_sites, tiny, dil, szt, force = segp
        kw = dict(seg=synth.voronoi_segmentation(sseed, shape, n_sites, tiny), seg_dilate=dil,
                  seg_sz_thd=szt, seg_force=force)
    res = fplobjdetect.voxel2obj(pred, r, sigma, tuple(off), buf, thd, **kw)
    assert len(g[name + '_conf']) > 5
    assert res['locs'].dtype == g[name + '_locs'].dtype and res['conf'].dtype == g[name + '_conf'].dtype
    assert np.array_equal(res['locs'], g[name + '_locs']), name
    assert np.array_equal(res['conf'], g[name + '_conf']), name
    # the float32 pipeline still works afterwards, and rounds differently
    f32 = fplobjdetect.voxel2obj(pred.astype(np.float32), r, sigma, tuple(off), buf, thd, **kw)
    assert f32['conf'].dtype == np.float64 and not np.array_equal(f32['conf'], res['conf'])


def test_other_dtypes_are_refused_as_in_the_reference(ctx):
    """scipy's gaussian_filter raises on float16 (the reference would fail there too)"""
    with pytest.raises(TypeError, match='float32, float64 or an integer'):
        fplobjdetect.voxel2obj(np.zeros((20, 20, 20), np.float16), 5, 2.0)
