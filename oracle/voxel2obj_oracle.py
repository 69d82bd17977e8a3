"""CPU restatement of `fplobjdetect.voxel2obj` including its segmentation-aware
branch, `/root/reference/flypylib/fplobjdetect.py:132-257`.  TEST INFRASTRUCTURE - see
oracle/__init__.py.  Pinned by tests/golden/voxel2obj_*.npz (point lists produced
by the reference's own function in the build container).

Stages:
  V1 zero-pad by r = obj_min_dist                                   (:158-159)
  V2 scipy gaussian_filter(sigma, truncate=2.0)                     (:167-168)
  V3 zero the outer r shell                                         (:170-175)
  V4 thresh = max(percentile_97(whole padded volume), thd)          (:183-185)
  V5 greedy radius NMS, argmax ties -> lowest flat index            (:187-231)
  V6 un-pad, buffer crop, offset shift, {'locs','conf'}             (:233-257)

`gaussian_filter_restated` spells out the arithmetic the HIP smoothing kernel
must reproduce bit-for-bit (scipy `ni_filters.c` symmetric branch): per axis in
order 0,1,2: line -> float64; out = x[0]*w[0]; for j = R..1 (outermost tap
first): out += (x[-j] + x[+j]) * w[j]  (separate multiply and add, no FMA);
store rounded to float32; 'reflect' boundary (d c b a | a b c d | d c b a).
"""
import numpy as np
from scipy import ndimage


def gaussian_weights(sigma, truncate=2.0):
    """scipy `_gaussian_kernel1d` (order 0): float64 weights, radius int(t*s+.5)"""
    sigma = float(sigma)
    radius = int(truncate * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum(), radius


def _reflect_index(i, n):
    """scipy 'reflect' (half-sample symmetric) index map for any integer i"""
    period = 2 * n
    i = np.mod(i, period)
    return np.where(i >= n, period - 1 - i, i)


def gaussian_filter_restated(vol, sigma, truncate=2.0):
    w, radius = gaussian_weights(sigma, truncate)
    out = np.asarray(vol, dtype=np.float32)
    for axis in range(3):
        n = out.shape[axis]
        src = np.moveaxis(out, axis, 0).astype(np.float64)
        idx = np.arange(n)
        acc = src[idx] * w[radius]
        for j in range(radius, 0, -1):
            lo = src[_reflect_index(idx - j, n)]
            hi = src[_reflect_index(idx + j, n)]
            acc = acc + (lo + hi) * w[radius - j]
        out = np.moveaxis(acc.astype(np.float32), 0, axis)
    return np.ascontiguousarray(out)


def smooth_and_clear(pred, r, sigma, use_scipy=True):
    """V1-V3: padded, smoothed, margin-zeroed float32 volume"""
    vol = np.pad(np.asarray(pred), r, 'constant')
    if use_scipy:
        vol = ndimage.gaussian_filter(vol, sigma, truncate=2.0)
    else:
        vol = gaussian_filter_restated(vol, sigma)
    if r > 0:
        vol[:r], vol[-r:] = 0, 0
        vol[:, :r], vol[:, -r:] = 0, 0
        vol[:, :, :r], vol[:, :, -r:] = 0, 0
    return vol


def greedy_nms(vol, thresh, r, seg=None, seg_dilate=None, seg_force=None):
    """V5 on a prepared volume: list of (x, y, z, value) in selection order.
    With a (padded) segmentation (reference :190-224) a pick only suppresses voxels
    of its r-ball that lie in its own segment - the segment's mask inside the
    (2r+1)^3 cube, grown by `seg_dilate` iterations of scipy's binary_dilation -
    plus, unconditionally, the ball of radius `seg_force`."""
    shape = vol.shape
    flat = vol.reshape(-1)
    live = np.flatnonzero(flat > thresh)          # raster order, ascending
    ax = np.arange(-r, r + 1)
    d2 = ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2
    keep_outside = d2 > r * r
    keep_off_centre = d2 > seg_force * seg_force if seg_force else None
    alive = np.ones(shape, dtype=bool)
    alive_flat = alive.reshape(-1)
    picked = []
    while live.size:
        vals = flat[live]
        j = int(np.argmax(vals))                  # first hit -> lowest index
        v = vals[j]
        if v <= 0:
            break
        z, y, x = np.unravel_index(live[j], shape)
        picked.append((x, y, z, v))
        box = (slice(z - r, z + r + 1), slice(y - r, y + r + 1), slice(x - r, x + r + 1))
        keep = keep_outside
        if seg is not None:
            same = seg[box] == seg[z, y, x]
            if seg_dilate is not None:
                same = ndimage.binary_dilation(same, iterations=seg_dilate)
            keep = np.logical_not(same) | keep_outside
            if seg_force:
                keep = keep & keep_off_centre
        alive[box] &= keep
        live = live[alive_flat[live]]
    return picked


def voxel2obj(pred, obj_min_dist, smoothing_sigma, volume_offset=(0, 0, 0),
              buffer_sz=0, thd=0, use_scipy=True, seg=None, seg_dilate=None,
              seg_sz_thd=None, seg_force=None):
    r = int(obj_min_dist)
    buf = (buffer_sz,) * 3 if np.size(buffer_sz) == 1 else tuple(buffer_sz)
    shape = np.asarray(pred).shape
    vol = smooth_and_clear(pred, r, smoothing_sigma, use_scipy)
    if seg is not None:
        seg = np.pad(np.asarray(seg), r, 'constant')
    if seg_sz_thd is not None:                    # reference :177-181 (padding counts as
        ids, counts = np.unique(seg, return_counts=True)      # part of segment 0)
        for sid, cnt in zip(ids, counts):
            if cnt < seg_sz_thd:
                vol[seg == sid] = 0
    thresh = np.maximum(np.percentile(vol, 97), thd)
    picked = greedy_nms(vol, thresh, r, seg, seg_dilate, seg_force)
    # reference :233-236: np.asarray of the [xx, yy, zz, max_val] rows - int64 for an integer
    # volume (its values are integers of the array's dtype), float64 otherwise, and for no rows
    int_rows = bool(picked) and np.issubdtype(vol.dtype, np.integer)
    pts = (np.asarray(picked, dtype=np.int64 if int_rows else np.float64) if picked
           else np.zeros((0, 4)))
    pts[:, :3] -= r
    lo = np.array([buf[0], buf[1], buf[2], -np.inf])
    pts = pts[~np.any(pts < lo, axis=1)]
    # note the literal pairing of the reference: x <-> shape[2] / buffer[0]
    hi = np.array([shape[2] - buf[0], shape[1] - buf[1], shape[0] - buf[2],
                   np.inf])
    pts = pts[~np.any(pts >= hi, axis=1)]
    pts = pts + np.array([tuple(volume_offset) + (0,)])
    if int_rows:
        pts = pts.astype(np.int64)              # (in place `+=` in the reference keeps int64)
    return {'locs': pts[:, :3], 'conf': pts[:, 3]}
