"""CPU restatement of the substack pipeline (TEST INFRASTRUCTURE - only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import this).

Follows `flypylib/fplobjdetect.py`:
  fri_get_image          :1023-1124   substack + buffer read, zero fill, normalisation
  full_roi_inference     :841-986     per substack: infer -> fri_postprocess
  fri_postprocess        :1126-1162   voxel2obj with offset (x-b, y-b, z-b), buffer, thd
Pinned by `tests/golden/fri_get_image.npz` (outputs of the reference's own
fri_get_image on a seeded volume, `tests/golden/make_golden.py`); the CNN part is
`cnn_oracle` (parity unpinned, see its header).
"""
import numpy as np

from . import infer_oracle, voxel2obj_oracle


def fri_get_image(volume, size, z, y, x, buffer_sz, image_normalize, float32_math=True):
    """-> (image or None, record dict).  `float32_math`: the arithmetic of the
    reference's numpy era (float32 array op python/np scalar stays float32); numpy
    >= 2 promotes `image - np.float64` to float64, which differs in the last bit."""
    image_sz = size + 2 * buffer_sz
    image_offset = [z - buffer_sz, y - buffer_sz, x - buffer_sz]
    full_size = volume.shape
    image = np.zeros((image_sz, image_sz, image_sz), 'uint8')
    lo = np.maximum(image_offset, 0)
    hi = np.minimum(np.asarray(image_offset) + image_sz, full_size)
    if lo[0] > hi[0] or lo[1] > hi[1] or lo[2] > hi[2]:
        return None, None
    image[lo[0] - image_offset[0]:hi[0] - image_offset[0],
          lo[1] - image_offset[1]:hi[1] - image_offset[1],
          lo[2] - image_offset[2]:hi[2] - image_offset[2]] = volume[
              lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
    im_raw_mn, im_raw_std = np.mean(image), np.std(image)
    idx = (image < 200) & (image > 1)
    if np.sum(idx) > 0:
        im_flt_mn, im_flt_std = np.mean(image[idx]), np.std(image[idx])
    else:
        im_flt_mn, im_flt_std = image_normalize[0], image_normalize[1]
    global_frac = 1. if len(image_normalize) < 3 else image_normalize[2]
    mn_use = global_frac * image_normalize[0] + (1 - global_frac) * im_flt_mn
    if float32_math:
        out = (image.astype('float32') - np.float32(mn_use)) / np.float32(image_normalize[1])
    else:
        out = (image.astype('float32') - np.float64(mn_use)) / np.float64(image_normalize[1])
    rec = dict(mn_use=mn_use, global_frac=global_frac, im_flt_mn=im_flt_mn,
               im_flt_std=im_flt_std, im_raw_mn=im_raw_mn, im_raw_std=im_raw_std)
    return out, rec


def norm_line(size, buffer_sz, z, y, x, image_normalize, rec):
    return '%d,%d,%d,%d,%d,%g,%g,%g,%g,%g,%g,%g,%g\n' % (
        size, buffer_sz, z, y, x, image_normalize[0], image_normalize[1],
        rec['global_frac'], rec['mn_use'], rec['im_flt_mn'], rec['im_flt_std'],
        rec['im_raw_mn'], rec['im_raw_std'])


def _seg_cube(seg, size, z, y, x, buffer_sz):
    """what dvid_node.get_labels3D hands fri_postprocess (:1139-1142): the labels of
    the substack + buffer box, zero outside the volume"""
    image_sz = size + 2 * buffer_sz
    org = [z - buffer_sz, y - buffer_sz, x - buffer_sz]
    lo = np.maximum(org, 0)
    hi = np.minimum(np.asarray(org) + image_sz, seg.shape)
    cube = np.zeros((image_sz,) * 3, seg.dtype)
    if np.all(hi > lo):
        cube[lo[0] - org[0]:hi[0] - org[0], lo[1] - org[1]:hi[1] - org[1],
             lo[2] - org[2]:hi[2] - org[2]] = seg[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
    return cube


def full_roi_inference(volume, substacks, predict_fn, infer_sz, rf_offset, thd,
                       image_normalize, obj_min_dist=27, smoothing_sigma=5, buffer_sz=35,
                       preds=None, seg=None):
    """substacks: iterable of (size, z, y, x).  `predict_fn` as in
    infer_oracle.infer_lattice.  `preds` (optional dict) substitutes the prediction of
    a substack (e.g. the device's) so that the post-processing can be compared bit for
    bit on identical inputs.  -> {'locs', 'conf'} concatenated in substack order, and
    the per-substack results."""
    locs, conf, per = [], [], {}
    for (size, z, y, x) in substacks:
        image, _ = fri_get_image(volume, size, z, y, x, buffer_sz, image_normalize)
        if image is None:
            out = {'locs': np.zeros((0, 3)), 'conf': np.zeros(0)}
        else:
            if preds is not None and (size, z, y, x) in preds:
                pred = preds[(size, z, y, x)]
            else:
                pred = infer_oracle.infer_lattice(image, infer_sz, rf_offset, predict_fn)
            seg_kw = {}
            if seg is not None:
                seg_kw = dict(seg=_seg_cube(seg, size, z, y, x, buffer_sz), seg_dilate=8,
                              seg_sz_thd=5000, seg_force=10)
            out = voxel2obj_oracle.voxel2obj(
                pred, obj_min_dist, smoothing_sigma,
                (x - buffer_sz, y - buffer_sz, z - buffer_sz), buffer_sz, thd, **seg_kw)
        per[(size, z, y, x)] = out
        locs.append(out['locs'])
        conf.append(out['conf'])
    return {'locs': np.concatenate(locs), 'conf': np.concatenate(conf)}, per
