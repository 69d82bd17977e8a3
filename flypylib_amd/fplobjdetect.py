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
              device=None, return_info=False):
    """convert voxel-wise predictions to object (point) predictions.

    pred: (Z,Y,X) float32 numpy array, or a float32 device tensor of that shape.
    Smoothing + percentile(97)-or-`thd` threshold + greedy non-maxima suppression
    with minimum distance `obj_min_dist`; detections inside `buffer_sz` of the
    volume faces are dropped; `volume_offset` (x,y,z) is added.
    """
    if seg is not None or seg_dilate is not None or seg_sz_thd is not None \
            or seg_force:
        raise NotImplementedError(
            'segmentation-aware suppression (reference fplobjdetect.py:161-165,'
            '177-181,190-224) is a SURVEY 8f follow-on')
    buffer_sz = fplutils.to3d(buffer_sz)
    if isinstance(pred, str):
        try:
            import h5py
        except ImportError:
            raise ImportError('reading %r needs h5py; pass an array' % pred)
        with h5py.File(pred, 'r') as f:
            pred = f['/main'][:]
    r = int(obj_min_dist)
    if isinstance(pred, np.ndarray):
        if pred.dtype != np.float32:
            # the reference smooths in the array's own dtype; only float32 (what
            # FplNetwork.infer returns) is bit-matched on the device
            raise TypeError('voxel2obj: pred must be float32, got %s' % pred.dtype)
        pred = np.ascontiguousarray(pred)
    pred_sz = tuple(int(s) for s in pred.shape)
    assert len(pred_sz) == 3, 'pred must be (Z,Y,X)'

    ctx = runtime.get_context(runtime.default_device() if device is None
                              else device)
    n_pad = int(np.prod([s + 2 * r for s in pred_sz]))
    lo_rank, hi_rank, gamma = percentile_plan(n_pad, 97, np.float32)
    weights = gaussian_kernel1d(smoothing_sigma, truncate=2.0)
    lo_v, hi_v = ctx.v2o_smooth(pred, pred_sz, r, weights, [lo_rank, hi_rank])
    thresh = np.maximum(percentile_lerp(lo_v, hi_v, gamma), thd)
    pts, rounds = ctx.v2o_nms(float(thresh))

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
    obj_out = {'locs': obj_pred[:, :3], 'conf': obj_pred[:, 3]}
    if return_info:
        return obj_out, dict(thresh=thresh, rounds=rounds,
                             ranks=(lo_rank, hi_rank), gamma=gamma)
    return obj_out


def _load_main(src):
    if isinstance(src, np.ndarray):
        return src
    try:
        import h5py
    except ImportError:
        raise ImportError('reading %r needs h5py; pass arrays instead' % (src,))
    with h5py.File(src, 'r') as f:
        return f['/main'][:]


def gen_batches(train_data, context_sz, batch_sz, is_mask=False, rng=None):
    """generator of balanced training batches (reference fplobjdetect.py:27-130).

    train_data: sequence of (image, labels_prefix) as in the reference (h5 paths;
    needs h5py) or of (image, labels, mask) arrays.  Yields
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


# the substack pipeline of the reference's fplobjdetect (fplobjdetect.py:841-1216)
from .fplpipeline import (szyx, roi_from_txt, gen_full_tab_roi, fri_filename,  # noqa: E402,F401
                          fri_get_image, full_roi_inference)
