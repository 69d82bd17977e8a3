"""CPU restatement of `FplNetwork.infer`'s tile/predict/stitch lattice
(`/root/reference/flypylib/fplnetwork.py:146-187`).  TEST INFRASTRUCTURE - see
oracle/__init__.py.  Pinned by tests/golden/infer_*.npz (outputs of the
reference's own `infer` with a fake `infer_network`).
"""
import numpy as np


def tile_lattice(image_shape, infer_sz, rf_offset):
    """tile origins (output coordinates) and input windows, reference order
    (`fplnetwork.py:146-159`): origins step by infer_sz - 2*offset from `offset`
    while < dim - offset; input window = [origin-off, min(origin+out+off, dim))"""
    dims = np.asarray(image_shape, dtype=np.int64)
    isz = np.asarray(infer_sz, dtype=np.int64)
    off = np.asarray(rf_offset, dtype=np.int64)
    out = isz - 2 * off
    axes = [np.arange(off[a], dims[a] - off[a], out[a]) for a in range(3)]
    grid = np.stack(np.meshgrid(*axes, indexing='ij'), 0).reshape(3, -1)
    start = grid - off[:, None]
    end = np.minimum(grid + out[:, None] + off[:, None], dims[:, None])
    return grid, start, end


def infer_lattice(image, infer_sz, rf_offset, predict_fn, n_gpu=1):
    """`predict_fn(batch (n, I,I,I, 1) float64) -> (n, I',I',I', 1)` where the
    first I-2*off outputs per axis are the valid ones (the reference's inference
    net upsamples to full resolution, `fplnetwork.py:100-105`)."""
    image = np.asarray(image)
    isz = tuple(int(v) for v in infer_sz)
    off = tuple(int(v) for v in rf_offset)
    locs, start, end = tile_lattice(image.shape, isz, off)
    n = locs.shape[1]
    n_pad = int(np.ceil(n / float(n_gpu)) * n_gpu)        # fplnetwork.py:162-163
    batch = np.zeros((n_pad,) + isz + (1,))               # float64, zero padded
    for t in range(n):
        s, e = start[:, t], end[:, t]
        ext = e - s
        batch[t, :ext[0], :ext[1], :ext[2], 0] = image[
            s[0]:e[0], s[1]:e[1], s[2]:e[2]]
    pred_batch = predict_fn(batch)
    pred = np.zeros(image.shape, dtype=np.float32)        # border shell stays 0
    for t in range(n):
        o, e = locs[:, t], end[:, t]
        ext = e - start[:, t]
        pred[o[0]:e[0] - off[0], o[1]:e[1] - off[1], o[2]:e[2] - off[2]] = \
            pred_batch[t, :ext[0] - 2 * off[0], :ext[1] - 2 * off[1],
                       :ext[2] - 2 * off[2], 0]
    return pred
