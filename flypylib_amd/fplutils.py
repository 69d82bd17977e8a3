"""Small helpers shared by the T-bar detection path.

Mirrors the two hot-path helpers of the reference (`flypylib/fplutils.py:9-22`):
`to3d` (scalar -> 3-tuple) and `set_filter` (boolean ball used by the NMS).
The skimage/h5py helpers of the reference module (`clahe`, `roi_from_txt`) are
out of scope.
"""
from collections import namedtuple

import numpy as np

szyx = namedtuple('szyx', 'size z y x')


def to3d(vv):
    """scalar -> (v, v, v); anything of size 3 is returned as given
    (reference `fplutils.py:9-12`)."""
    if np.size(vv) == 1:
        if isinstance(vv, (tuple, list, np.ndarray)):
            vv = np.asarray(vv).reshape(-1)[0]
            if isinstance(vv, np.generic):
                vv = vv.item()
        return (vv, vv, vv)
    return vv


def ball_sq_dist(radius):
    """integer squared distance from the centre on a (2r+1)^3 grid"""
    ax = np.arange(-radius, radius + 1, dtype=np.int64)
    return (ax[:, None, None] ** 2 + ax[None, :, None] ** 2
            + ax[None, None, :] ** 2)


def set_filter(radius, return_dist=False):
    """boolean ball `dist <= radius` on a (2r+1)^3 grid (reference
    `fplutils.py:14-22`).  For integer radius, `sqrt(d2) <= r` and `d2 <= r*r`
    select the same voxels; the float distance is only materialised on request."""
    d2 = ball_sq_dist(int(radius))
    inside = d2 <= int(radius) * int(radius)
    if return_dist:
        return inside, np.sqrt(d2.astype(np.float64))
    return inside
