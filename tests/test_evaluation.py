"""Output-side data formats (reference fplsynapses.py:11-111) against the reference's
own outputs, and the point-matching evaluation (fplobjdetect.py:259-455): the empty
branches against the reference, the assignment solver (the reference calls pulp,
absent here) against brute force."""
import itertools
import json
import os

import numpy as np

from flypylib_amd import fplobjdetect, fplsynapses

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'synapses.npz'))


def _points():
    rs = np.random.RandomState(7)
    return {'locs': np.floor(rs.rand(12, 3) * [60, 50, 40]), 'conf': rs.rand(12)}


def test_json_formats_match_the_reference(tmp_path):
    tb = _points()
    dvid = fplsynapses.tbars_to_json_format(tb, labels=np.arange(12) * 3)
    rav = fplsynapses.tbars_to_json_format_raveler(tb, str(tmp_path / 'r.json'))
    assert json.dumps(dvid, sort_keys=True) == str(GOLD['dvid_json'])
    assert json.dumps(rav, sort_keys=True) == str(GOLD['raveler_json'])
    texts = {'dvid': json.dumps(dvid), 'raveler': str(tmp_path / 'r.json'),
             'nested': json.dumps([dvid])}
    for name, src in texts.items():
        back = fplsynapses.load_from_json(src)
        assert np.array_equal(back['locs'], GOLD[name + '/locs'])
        assert np.array_equal(back['conf'], GOLD[name + '/conf'])
        cut = fplsynapses.load_from_json(src, (60, 50, 40), (10, 5, 8))
        assert np.array_equal(cut['locs'], GOLD[name + '/buf_locs'])
        assert np.array_equal(cut['conf'], GOLD[name + '/buf_conf'])
        assert 0 < len(cut['conf']) < 12


def test_obj_pr_empty_branches_match_the_reference():
    e, p = np.zeros((0, 3)), _points()['locs']
    for name, (a, b) in (('no_pred', (e, p)), ('no_gt', (p, e)), ('none', (e, e))):
        r = fplobjdetect.obj_pr(a, b, 5.0)
        assert np.array_equal(np.array([r.num_tp, r.tot_pred, r.tot_gt, r.pp, r.rr], np.float64),
                              GOLD['pr_' + name])
        assert r.match is None


def _brute_force(d, allow_mult):
    """optimum of the reference's integer program by enumeration"""
    n, m = d.shape
    best = 0.0
    if allow_mult:
        return sum(min(0.0, d[:, j].min()) for j in range(m))
    for k in range(1, min(n, m) + 1):
        for rows in itertools.combinations(range(n), k):
            for cols in itertools.permutations(range(m), k):
                if all(d[r, c] < 0 for r, c in zip(rows, cols)):
                    best = min(best, sum(d[r, c] for r, c in zip(rows, cols)))
    return best


def test_obj_match_is_the_optimum_of_the_integer_program():
    rs = np.random.RandomState(1)
    for trial in range(40):
        n, m = rs.randint(1, 6), rs.randint(1, 6)
        d = rs.randn(n, m) * 3 + 1.0
        for allow_mult in (False, True):
            mt = fplobjdetect.obj_match(d, allow_mult)
            assert mt.shape == (n, m) and mt.dtype == bool
            assert not mt[d >= 0].any()                       # only admissible pairs
            assert mt.sum(axis=0).max() <= 1                  # each ground truth once
            if not allow_mult:
                assert mt.sum(axis=1).max() <= 1
            assert abs(d[mt].sum() - _brute_force(d, allow_mult)) < 1e-9


def test_obj_pr_curve_and_aggregate():
    gt = {'locs': np.array([[10., 10, 10], [30, 30, 30], [50, 10, 20]])}
    pred = {'locs': np.array([[11., 10, 10], [30, 31, 30], [80, 80, 80], [49, 10, 20]]),
            'conf': np.array([0.9, 0.8, 0.7, 0.3])}
    r = fplobjdetect.obj_pr_curve(pred, gt, 3.0, np.array([0.2, 0.5, 0.95]))
    assert list(r.num_tp) == [3, 2, 0] and list(r.tot_pred) == [4, 3, 0]
    assert np.allclose(r.pp, [0.75, 2 / 3, 1]) and np.allclose(r.rr, [1, 2 / 3, 0])
    lbl = fplobjdetect.obj_pr(pred['locs'], gt['locs'], 3.0, np.array([1, 2, 3, 4]),
                              np.array([1, 9, 4]))
    assert lbl.num_tp == 2                                     # label mismatch blocks one pair
    agg = fplobjdetect.aggregate_pr([r, r])
    assert np.allclose(agg.num_tp, 2 * r.num_tp) and np.allclose(agg.pp[:2], r.pp[:2], atol=1e-6)


def test_write_labels_mask_matches_the_reference(tmp_path):
    tb = {'locs': np.array([[12, 14, 16], [20, 15, 13], [25, 25, 25]]), 'conf': np.ones(3)}
    labels, mask = fplsynapses.write_labels_mask(tb, np.ones((36, 38, 40), 'uint8'), 3, 6, 4,
                                                 str(tmp_path / 'x'))
    assert np.array_equal(labels, GOLD['lm_labels']) and np.array_equal(mask, GOLD['lm_mask'])
    assert labels.dtype == np.uint8 and labels[16, 14, 12] == 1 and mask[16, 14, 12] == 1
    assert mask[16, 14, 17] == 0                      # in the ignore shell
    assert np.array_equal(np.load(str(tmp_path / 'x_labels.npy')), labels)
    assert np.array_equal(np.load(str(tmp_path / 'x_mask.npy')), mask)
