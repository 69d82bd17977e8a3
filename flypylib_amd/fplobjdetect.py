"""Voxel predictions -> point detections, and training-patch generation.

`voxel2obj` keeps the reference's signature and result
(`flypylib/fplobjdetect.py:132-257`): dict with 'locs' (N x 3, x/y/z order,
float64) and 'conf' (N, float64) in descending-confidence order.  The device
stages (pad, Gaussian smoothing, margin zeroing, exact order statistics,
radius NMS) run in libfplhip.so (`fpl_v2o_smooth`, `fpl_v2o_nms`); the O(N)
epilogue (un-pad, buffer crop, offset) is host numpy, as in the reference.
There is no CPU fallback.
"""
import numpy as np

from . import fplutils, runtime


def gaussian_kernel1d(sigma, truncate=2.0):
    """float64 weights of scipy.ndimage.gaussian_filter1d (order 0), which the
    reference calls with truncate=2.0 (`fplobjdetect.py:167-168`)"""
    sigma = float(sigma)
    radius = int(truncate * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def percentile_plan(n, q, dtype=np.float32):
    """ranks and interpolation weight np.percentile(a, q) uses for a float array
    of `n` elements with the default 'linear' method: returns (prev, next, gamma)
    with gamma in the array's dtype, mirroring numpy's own expression order
    (numpy/lib/_function_base_impl.py: percentile -> _quantile)."""
    dt = np.dtype(dtype).type
    quant = np.asanyarray(np.true_divide(q, dt(100)))
    virtual = np.asanyarray((n - 1) * quant)
    prev = np.asanyarray(np.floor(virtual))
    nxt = np.asanyarray(prev + 1)
    if virtual >= n - 1:
        prev, nxt = np.asanyarray(dt(n - 1)), np.asanyarray(dt(n - 1))
    if virtual < 0:
        prev, nxt = np.asanyarray(dt(0)), np.asanyarray(dt(0))
    gamma = np.asanyarray(virtual - prev, dtype=virtual.dtype)
    return int(prev), int(nxt), gamma


def percentile_lerp(lo, hi, gamma):
    """numpy's `_lerp(a, b, t)` on scalars, same operation order and dtypes"""
    a, b, t = np.asanyarray(lo), np.asanyarray(hi), np.asanyarray(gamma)
    diff = np.subtract(b, a)
    out = np.asanyarray(np.add(a, diff * t))
    if t >= 0.5:
        out = np.asanyarray(np.subtract(b, diff * (1 - t)).astype(out.dtype))
    return out[()]


def voxel2obj(pred, obj_min_dist, smoothing_sigma,
              volume_offset=(0, 0, 0), buffer_sz=0, thd=0,
              seg=None, seg_dilate=None, seg_sz_thd=None, seg_force=None,
              device=None, return_info=False, _ctx=None):
    """convert voxel-wise predictions to object (point) predictions.

    pred: (Z,Y,X) float32 numpy array, or a float32 device tensor of that shape; a float64
    numpy array is smoothed, thresholded and compared in float64, as the reference does
    for its input's own dtype (the results differ from the float32 ones: no rounding
    between the smoothing passes); an INTEGER numpy array likewise in its own type - scipy
    filters it with float64 accumulation and truncates back to the integer type after
    every axis, the point list comes back as int64 (reference :167-168, :233-236).
    float16 raises, as scipy's filter does in the reference.
    Smoothing + percentile(97)-or-`thd` threshold + greedy non-maxima suppression
    with minimum distance `obj_min_dist`; detections inside `buffer_sz` of the
    volume faces are dropped; `volume_offset` (x,y,z) is added.

    With `seg` (integer labels of pred's shape; array, device array or .npy path) the
    suppression is segmentation-aware (reference :161-165,177-181,190-224): a pick
    suppresses only the voxels of its ball that lie in its own segment - its mask
    within the (2r+1)^3 cube, grown by `seg_dilate` binary-dilation iterations - plus
    the ball of radius `seg_force`; `seg_sz_thd` first zeroes the smoothed prediction
    inside segments of fewer voxels (obj_min_dist <= 127 in this mode).
    """
    buffer_sz = fplutils.to3d(buffer_sz)
    if isinstance(pred, str):
        pred = _load_main(pred)
    r = int(obj_min_dist)
    f64 = False
    int_pred = False
    if isinstance(pred, np.ndarray):
        if pred.dtype == np.float64:
            f64 = True
        elif np.issubdtype(pred.dtype, np.integer):
            # an integer volume is smoothed in its own type: exact doubles on the float64
            # kernels, truncated after every axis pass (fpl_v2o_set_integer)
            f64 = int_pred = True
            pred = pred.astype(np.float64)
        elif pred.dtype != np.float32:
            # the reference smooths in the array's own dtype; scipy itself refuses float16
            raise TypeError('voxel2obj: pred must be float32, float64 or an integer type, got %s'
                            % pred.dtype)
        pred = np.ascontiguousarray(pred)
    pred_sz = tuple(int(s) for s in pred.shape)
    assert len(pred_sz) == 3, 'pred must be (Z,Y,X)'

    ctx = _ctx or runtime.get_context(runtime.default_device() if device is None
                                      else device)
    n_pad = int(np.prod([s + 2 * r for s in pred_sz]))
    lo_rank, hi_rank, gamma = percentile_plan(n_pad, 97, np.float64 if f64 else np.float32)
    weights = gaussian_kernel1d(smoothing_sigma, truncate=2.0)
    if f64:
        # float64: the volume of record is the double one; the NMS runs on float32 rank
        # surrogates of the candidates (same order, same ties) and the picked voxels'
        # float64 values are fetched afterwards
        if seg is None and seg_sz_thd is not None:
            raise ValueError('seg_sz_thd needs a segmentation')
        if int_pred:
            ctx.v2o_set_integer(True)
        ctx.v2o_smooth_f64(pred, pred_sz, r, weights)
        if seg is not None:
            if isinstance(seg, str):
                seg = _load_main(seg)
            assert tuple(int(v) for v in seg.shape) == pred_sz, 'seg must have pred\'s shape'
            ctx.v2o_set_seg(seg, pred_sz, seg_sz_thd)
        lo_v, hi_v = ctx.v2o_select_f64([lo_rank, hi_rank])
        thresh = np.maximum(percentile_lerp(lo_v, hi_v, gamma), thd)
        ctx.v2o_rank_f64(float(thresh))
        pts, rounds = (ctx.v2o_nms(0.5) if seg is None
                       else ctx.v2o_nms_seg(0.5, seg_dilate, seg_force))
        pdims = [s + 2 * r for s in pred_sz]
        flat = ((pts[:, 0].astype(np.int64) * pdims[1] + pts[:, 1].astype(np.int64)) * pdims[2]
                + pts[:, 2].astype(np.int64))
        pts[:, 3] = ctx.v2o_values_f64(flat)
    elif seg is None:
        if seg_sz_thd is not None:
            raise ValueError('seg_sz_thd needs a segmentation')
        if thd > 0:
            # thresh = max(percentile, thd) >= thd; as float32 rounded DOWN so that the
            # promise holds whatever thd's own precision
            floor = np.float32(thd)
            if float(floor) > float(thd):
                floor = np.nextafter(floor, np.float32(-np.inf))
            ctx.v2o_set_floor(floor)
        lo_v, hi_v = ctx.v2o_smooth(pred, pred_sz, r, weights, [lo_rank, hi_rank])
        thresh = np.maximum(percentile_lerp(lo_v, hi_v, gamma), thd)
        pts, rounds = ctx.v2o_nms(float(thresh))
    else:
        if isinstance(seg, str):
            seg = _load_main(seg)
        assert tuple(int(v) for v in seg.shape) == pred_sz, 'seg must have pred\'s shape'
        ctx.v2o_smooth(pred, pred_sz, r, weights, [])
        ctx.v2o_set_seg(seg, pred_sz, seg_sz_thd)           # pads; zeroes small segments
        lo_v, hi_v = ctx.v2o_select([lo_rank, hi_rank])
        thresh = np.maximum(percentile_lerp(lo_v, hi_v, gamma), thd)
        pts, rounds = ctx.v2o_nms_seg(float(thresh), seg_dilate, seg_force)

    # pts rows: (z, y, x, value) in padded coordinates, emission order
    obj_pred = np.empty((pts.shape[0], 4), np.float64)
    obj_pred[:, 0] = pts[:, 2]
    obj_pred[:, 1] = pts[:, 1]
    obj_pred[:, 2] = pts[:, 0]
    obj_pred[:, 3] = pts[:, 3]
    obj_pred[:, :3] -= r

    min_bound = np.asarray([[buffer_sz[0], buffer_sz[1], buffer_sz[2], -np.inf]])
    obj_pred = obj_pred[~np.any(obj_pred < min_bound, axis=1)]
    # x is checked against shape[2] with buffer_sz[0] etc. - the reference's own
    # pairing (fplobjdetect.py:239-250), kept literally
    max_bound = np.asarray([[pred_sz[2] - buffer_sz[0],
                             pred_sz[1] - buffer_sz[1],
                             pred_sz[0] - buffer_sz[2], np.inf]])
    obj_pred = obj_pred[~np.any(obj_pred >= max_bound, axis=1)]

    obj_pred = obj_pred + np.array([tuple(volume_offset) + (0,)])
    if int_pred and pts.shape[0]:
        # the reference's rows are [xx, yy, zz, max_val] with an integer max_val: np.asarray
        # makes them int64, and its in-place arithmetic keeps that (:233-252)
        obj_pred = obj_pred.astype(np.int64)
    obj_out = {'locs': obj_pred[:, :3], 'conf': obj_pred[:, 3]}
    if return_info:
        return obj_out, dict(thresh=thresh, rounds=rounds,
                             ranks=(lo_rank, hi_rank), gamma=gamma)
    return obj_out


def _load_main(src):
    """an array, a `.npy` path, or - as in the reference - an `.h5` path whose dataset
    'main' holds the volume (h5py when installed, else the package's own reader)"""
    if isinstance(src, np.ndarray):
        return src
    if isinstance(src, str) and src.endswith('.npy'):
        return np.load(src)
    from . import keras_io
    return keras_io.read_main(src)


def gen_batches(train_data, context_sz, batch_sz, is_mask=False, rng=None):
    """generator of balanced training batches (reference fplobjdetect.py:27-130).

    train_data: sequence of (image, labels_prefix) as in the reference (h5 paths;
    read with h5py or the package's own reader) or of (image, labels, mask) arrays.  Yields
    (data (B,s,s,s,1) float32, labels (B,1,1,1,1) uint8 or (B,6,6,6,1) if is_mask).
    Half of each batch is centred on label-0 voxels, half on label-1 voxels
    (interleaved), followed by rot90 / flip augmentation.
    """
    rng = np.random if rng is None else rng
    context_sz = fplutils.to3d(context_sz)
    n_per_class = int(round(batch_sz / 2))
    half = tuple(int(round(cc / 2)) for cc in context_sz)

    vols = []
    for tr in train_data:
        if len(tr) == 3:
            im, ll, mm = (np.array(_load_main(a)) for a in tr)
        else:
            im = _load_main(tr[0])
            ll = np.array(_load_main('%slabels.h5' % tr[1]))
            mm = np.array(_load_main('%smask.h5' % tr[1]))
        for ax in range(3):                      # patches must fit in the volume
            sl = [slice(None)] * 3
            sl[ax] = slice(0, half[ax]); mm[tuple(sl)] = 0
            sl[ax] = slice(-half[ax], None); mm[tuple(sl)] = 0
        centres = [((ll == cc) & (mm == 1)).nonzero() for cc in range(2)]
        if is_mask:
            ll[mm == 0] = 2
        vols.append((im, ll, centres))

    data = np.zeros((batch_sz,) + tuple(context_sz) + (1,), dtype='float32')
    labels = np.zeros((batch_sz, 6, 6, 6, 1) if is_mask
                      else (batch_sz, 1, 1, 1, 1), dtype='uint8')
    vi = 0
    while True:
        im, ll, centres = vols[vi]
        for cc in range(2):
            n_possible = len(centres[cc][0])
            if n_possible == 0:          # keep last iteration's examples
                continue
            pick = rng.choice(n_possible, n_per_class, True)
            for ii in range(n_per_class):
                z, y, x = (int(centres[cc][a][pick[ii]]) for a in range(3))
                ex = ii * 2 + cc
                data[ex, :, :, :, 0] = im[z - half[0]:z + half[0],
                                          y - half[1]:y + half[1],
                                          x - half[2]:x + half[2]]
                if is_mask:
                    labels[ex, :, :, :, 0] = ll[z - 3:z + 3, y - 3:y + 3,
                                                x - 3:x + 3]
                else:
                    labels[ex, 0] = ll[z, y, x]
        rot = np.floor(4 * rng.rand(batch_sz))
        ref = np.floor(2 * rng.rand(batch_sz))
        fpz = np.floor(2 * rng.rand(batch_sz))
        for ii in range(batch_sz):
            arrs = [data] + ([labels] if is_mask else [])
            for a in arrs:
                v = a[ii, :, :, :, 0]
                if rot[ii]:
                    v = np.rot90(v, int(rot[ii]), (1, 2))
                if ref[ii]:
                    v = np.flip(v, 2)
                if fpz[ii]:
                    v = np.flip(v, 0)
                a[ii, :, :, :, 0] = v
        yield data, labels
        vi = (vi + 1) % len(vols)


# ---- evaluation against ground-truth points (reference fplobjdetect.py:259-455) ------
from collections import namedtuple   # noqa: E402

PR_Result = namedtuple('PR_Result', 'num_tp tot_pred tot_gt pp rr match')


def obj_match(dists, allow_mult=False):
    """match predictions to ground truth: `dists` (N x M) = distance minus the match
    threshold, negative entries are admissible pairs.  Minimises the summed (negative)
    cost with every ground-truth point used at most once and, unless `allow_mult`,
    every prediction at most once - the integer program the reference hands to pulp
    (:259-321), solved here as the equivalent assignment problem.  -> N x M bool."""
    from scipy.optimize import linear_sum_assignment
    dists = np.asarray(dists, np.float64)
    n_pred, n_gt = dists.shape
    out = np.zeros((n_pred, n_gt), dtype=bool)
    if n_pred == 0 or n_gt == 0:
        return out
    if allow_mult:                     # only the ground-truth side is constrained
        best = np.argmin(dists, axis=0)
        ok = dists[best, np.arange(n_gt)] < 0
        out[best[ok], np.arange(n_gt)[ok]] = True
        return out
    rows, cols = linear_sum_assignment(np.minimum(dists, 0.0))   # leaving a pair out costs 0
    ok = dists[rows, cols] < 0
    out[rows[ok], cols[ok]] = True
    return out


def obj_pr(predict_locs, groundtruth_locs, dist_thresh, predict_lbls=None,
           groundtruth_lbls=None, allow_mult=False):
    """precision / recall of predicted vs ground-truth locations at a distance
    threshold (reference :323-376): pairs closer than `dist_thresh` (and, if labels are
    given, with equal labels) are admissible, `obj_match` picks the matching"""
    n_pred, n_gt = predict_locs.shape[0], groundtruth_locs.shape[0]
    if n_pred == 0 or n_gt == 0:
        return PR_Result(num_tp=0, tot_pred=n_pred, tot_gt=n_gt,
                         pp=1 if n_pred == 0 else 0, rr=1 if n_gt == 0 else 0, match=None)
    delta = predict_locs.reshape(n_pred, 1, 3) - groundtruth_locs.reshape(1, n_gt, 3)
    cost = np.sqrt((delta ** 2).sum(axis=2)) - dist_thresh
    if predict_lbls is not None:
        differ = predict_lbls.reshape(-1, 1) != groundtruth_lbls.reshape(1, -1)
        cost += (dist_thresh + 1.) * differ.astype('float32')
    match = obj_match(cost, allow_mult=allow_mult)
    num_tp = match.sum()
    extra = np.maximum(match.sum(axis=1) - 1, 0).sum()     # predictions matched twice
    return PR_Result(num_tp=num_tp, tot_pred=n_pred + extra, tot_gt=n_gt,
                     pp=num_tp / n_pred, rr=num_tp / n_gt, match=match)


def obj_pr_curve(predict, groundtruth, dist_thresh, thresholds, predict_lbls=None,
                 groundtruth_lbls=None, allow_mult=False):
    """precision / recall at each confidence threshold (reference :378-436); `predict`
    / `groundtruth` are {'locs','conf'} dicts or json files"""
    from . import fplsynapses
    if isinstance(predict, str):
        predict = fplsynapses.load_from_json(predict)
    if isinstance(groundtruth, str):
        groundtruth = fplsynapses.load_from_json(groundtruth)
    points = []
    for thd in np.asarray(thresholds).reshape(-1):
        sel = predict['conf'] >= thd
        points.append(obj_pr(predict['locs'][sel, :], groundtruth['locs'], dist_thresh,
                             None if predict_lbls is None else predict_lbls[sel],
                             groundtruth_lbls, allow_mult=allow_mult))

    def column(name):
        return np.array([getattr(pt, name) for pt in points], dtype=np.float64)
    return PR_Result(num_tp=column('num_tp'), tot_pred=column('tot_pred'),
                     tot_gt=column('tot_gt'), pp=column('pp'), rr=column('rr'),
                     match=points[0].match if points else None)


def aggregate_pr(results):
    """pool per-substack PR curves (reference :439-455)"""
    num_tp = sum(r.num_tp for r in results) + np.zeros(results[0].num_tp.shape)
    tot_pred = sum(r.tot_pred for r in results) + np.zeros(results[0].num_tp.shape)
    tot_gt = sum(r.tot_gt for r in results) + np.zeros(results[0].num_tp.shape)
    return PR_Result(num_tp=num_tp, tot_pred=tot_pred, tot_gt=tot_gt,
                     pp=num_tp / (tot_pred + 10e-8), rr=num_tp / (tot_gt + 10e-8), match=None)


def _volumes(train_data, half):
    """(image, labels, mask[, weights]) arrays per training volume with the mask
    cleared where a patch would not fit (reference :690-705); entries are
    (image, labels_prefix[, weights]) h5 paths as in the reference or
    (image, labels, mask[, weights]) arrays / .npy paths"""
    vols = []
    for tr in train_data:
        as_prefix = isinstance(tr[1], str) and not tr[1].endswith('.npy')
        if len(tr) >= 3 and not as_prefix:
            im, ll, mm = (np.array(_load_main(a)) for a in tr[:3])
            ww = np.array(_load_main(tr[3])) if len(tr) > 3 else None
        else:
            im = np.array(_load_main(tr[0]))
            ll = np.array(_load_main('%slabels.h5' % tr[1]))
            mm = np.array(_load_main('%smask.h5' % tr[1]))
            ww = np.array(_load_main(tr[2])) if len(tr) > 2 else None
        for ax in range(3):
            sl = [slice(None)] * 3
            sl[ax] = slice(0, half[ax]); mm[tuple(sl)] = 0
            sl[ax] = slice(-half[ax], None); mm[tuple(sl)] = 0
        vols.append((im, ll, mm, ww))
    return vols


def get_out_sz(in_sz):
    """output size of a unet-style net for an input subvolume (reference :525-534)"""
    import math
    bottleneck_sz = int(math.floor(math.floor((in_sz - 2) / 2) - 2) / 2)
    return (bottleneck_sz * 2 - 2) * 2 - 2


def gen_volume(train_data, context_sz, batch_sz, ratio, rng=None):
    """generator of training batches with dense 6^3 labels (reference
    fplobjdetect.py:536-658; what scripts/fpl_cx1_0_unet_4ss_all.py:41-42 trains
    unet_like2 with): example i of a batch comes from volume i mod n_volumes, centred
    on a label-0 voxel with probability `ratio`, else on a label-1 voxel (label 0 if
    the volume has no positives); masked-out voxels are labelled 2; rot90 in axes
    (1,2), flip of axis 2, flip of axis 0."""
    rng = np.random if rng is None else rng
    context_sz = fplutils.to3d(context_sz)
    half = tuple(int(round(cc / 2)) for cc in context_sz)
    vols = []
    for im, ll, mm, _ in _volumes(train_data, half):
        centres = [((ll == cc) & (mm == 1)).nonzero() for cc in range(2)]
        ll = ll.copy()
        ll[mm == 0] = 2
        vols.append((im, ll, centres))
    data = np.zeros((batch_sz,) + tuple(context_sz) + (1,), dtype='float32')
    labels = np.zeros((batch_sz, 6, 6, 6, 1), dtype='uint8')
    train_idx = 0
    while True:
        for ex in range(batch_sz):
            im, ll, centres = vols[train_idx]
            cc = 0 if rng.uniform(0, 1) < ratio else 1
            if len(centres[cc][0]) == 0:
                cc = 0
            k = rng.choice(len(centres[cc][0]), batch_sz, True)[ex]
            z, y, x = (int(centres[cc][a][k]) for a in range(3))
            data[ex, :, :, :, 0] = im[z - half[0]:z + half[0], y - half[1]:y + half[1],
                                      x - half[2]:x + half[2]]
            labels[ex, :, :, :, 0] = ll[z - 3:z + 3, y - 3:y + 3, x - 3:x + 3]
            train_idx = (train_idx + 1) % len(vols)
        rot = np.floor(4 * rng.rand(batch_sz))
        ref = np.floor(2 * rng.rand(batch_sz))
        fpz = np.floor(2 * rng.rand(batch_sz))
        for ii in range(batch_sz):
            for a in (data, labels):
                v = a[ii, :, :, :, 0]
                if rot[ii]:
                    v = np.rot90(v, int(rot[ii]), (1, 2))
                if ref[ii]:
                    v = np.flip(v, 2)
                if fpz[ii]:
                    v = np.flip(v, 0)
                a[ii, :, :, :, 0] = v
        yield data, labels


def evaluate_substacks(network, substacks, thds, obj_min_dist=27, smoothing_sigma=5,
                       volume_offset=(0, 0, 0), buffer_sz=5, allow_mult=False,
                       normalize=None):
    """precision / recall curves of a network on labelled substacks (reference
    :463-523): per substack [image, ground-truth json (, segmentation)] -> infer ->
    voxel2obj -> obj_pr_curve against the json's T-bars (buffer applied to both), then
    the aggregate.  The reference forks a post-processing worker per substack; here
    inference and voxel2obj share the GPU, so substacks run in order."""
    from . import fplsynapses
    thds = np.asarray(thds)
    results = []
    for ss in substacks:
        pred = network.infer(ss[0], normalize=normalize)
        out = voxel2obj(pred, obj_min_dist, smoothing_sigma, volume_offset, buffer_sz)
        gt = fplsynapses.load_from_json(ss[1], pred.shape, buffer_sz)
        lbls_pd = lbls_gt = None
        if len(ss) >= 3 and ss[2] is not None:
            seg = np.asarray(_load_main(ss[2]))

            def labels_at(tt):
                ind = tt['locs'].astype(int)
                return seg[ind[:, 2], ind[:, 1], ind[:, 0]]
            lbls_pd, lbls_gt = labels_at(out), labels_at(gt)
        results.append(obj_pr_curve(out, gt, obj_min_dist, thds, lbls_pd, lbls_gt,
                                    allow_mult=allow_mult))
    return aggregate_pr(results), results


def gen_volume2(train_data, context_sz, batch_sz, ratio, noise_aug=[0, 0], rng=None):
    """generator of training batches with dense 6^3 labels for the U-Nets (reference
    fplobjdetect.py:660-822): `ratio` of each outer round of 100 batches is centred on
    label-0 voxels, the rest on label-1 voxels, drawn from all volumes (optionally
    with the per-voxel sampling weights of `write_sampling_weights` as a 3rd / 4th
    entry), in a random order; labels of masked-out voxels are 2 (the masked losses
    ignore them); intensity noise `noise_aug = [additive sd, multiplicative sd]`;
    rot90 in axes (1,2), flip of axis 1, flip of axis 0.  Yields
    (data (B,s,s,s,1) float32, labels (B,6,6,6,1) uint8)."""
    rng = np.random if rng is None else rng
    context_sz = fplutils.to3d(context_sz)
    half = tuple(int(round(cc / 2)) for cc in context_sz)
    vols = _volumes(train_data, half)
    weighted = vols[0][3] is not None
    # sample positions per class: (volume index, z, y, x[, weight])
    pos = []
    for cc in range(2):
        cols = [[], [], [], [], []]
        for vi, (im, ll, mm, ww) in enumerate(vols):
            sel = (ll == cc) & (mm == 1)
            if weighted:
                sel &= ww > 0
            idx = sel.nonzero()
            cols[0].append(np.full(idx[0].shape, vi, np.int32))
            for a in range(3):
                cols[1 + a].append(idx[a].astype(np.int32))
            if weighted:
                cols[4].append(ww[idx].astype(np.float64))
        cols = [np.concatenate(c) if c else None for c in cols]
        if weighted:
            cols[4] = cols[4] / np.sum(cols[4].astype('float32'))
        pos.append(cols)
    labelled = []
    for im, ll, mm, ww in vols:
        ll = ll.copy()
        ll[mm == 0] = 2                           # ignored by the masked losses
        labelled.append(ll)

    out_rr = (3, 3, 3)                            # out_sz (6, 6, 6), reference :744-746
    data = np.zeros((batch_sz,) + tuple(context_sz) + (1,), dtype='float32')
    labels = np.zeros((batch_sz, 6, 6, 6, 1), dtype='uint8')
    outer_batches = 100
    outer_batch_sz = outer_batches * batch_sz
    n_neg = int(round(ratio * outer_batch_sz))
    n_pos = outer_batch_sz - n_neg
    while True:
        neg_idx = rng.choice(len(pos[0][0]), n_neg, True, pos[0][4] if weighted else None)
        pos_idx = rng.choice(len(pos[1][0]), n_pos, True, pos[1][4] if weighted else None)
        all_idx = rng.permutation(outer_batch_sz)
        sample_idx = 0
        for _ in range(outer_batches):
            for ex in range(batch_sz):
                k = all_idx[sample_idx]
                sample_idx += 1
                cc, k = (0, neg_idx[k]) if k < n_neg else (1, pos_idx[k - n_neg])
                vi, z, y, x = (int(pos[cc][a][k]) for a in range(4))
                im = vols[vi][0]
                data[ex, :, :, :, 0] = (
                    ((noise_aug[1] * rng.randn()) + 1.) *
                    im[z - half[0]:z + half[0], y - half[1]:y + half[1],
                       x - half[2]:x + half[2]]) + noise_aug[0] * rng.randn()
                labels[ex, :, :, :, 0] = labelled[vi][
                    z - out_rr[0]:z + out_rr[0], y - out_rr[1]:y + out_rr[1],
                    x - out_rr[2]:x + out_rr[2]]
            rot = np.floor(4 * rng.rand(batch_sz))
            ref = np.floor(2 * rng.rand(batch_sz))
            fpz = np.floor(2 * rng.rand(batch_sz))
            for ii in range(batch_sz):
                for a in (data, labels):
                    v = a[ii, :, :, :, 0]
                    if rot[ii]:
                        v = np.rot90(v, int(rot[ii]), (1, 2))
                    if ref[ii]:
                        v = np.fliplr(v)
                    if fpz[ii]:
                        v = np.flipud(v)
                    a[ii, :, :, :, 0] = v
            yield data.copy(), labels.copy()


def write_sampling_weights(train_data, network, fn_prefix, l0_thresh, l1_thresh):
    """per-voxel loss of the current network as sampling weights for `gen_volume2`
    (reference :824-839).  Weights are written as '<fn_prefix>%02d.npy' (the
    reference writes .h5; h5py is not available here) and appended to each entry."""
    train_data_aug = []
    for idx, tr in enumerate(train_data):
        loss = network.voxel_loss(tr[0], tr[1] if isinstance(tr[1], str) else tr[1:3],
                                  l0_thresh, l1_thresh)
        ww_fn = '%s%02d.npy' % (fn_prefix, idx)
        np.save(ww_fn, loss)
        train_data_aug.append(list(tr) + [ww_fn, ])
    return train_data_aug


# the substack pipeline of the reference's fplobjdetect (fplobjdetect.py:841-1216)
from .fplpipeline import (szyx, roi_from_txt, gen_full_tab_roi, fri_filename,  # noqa: E402,F401
                          fri_get_image, full_roi_inference)
