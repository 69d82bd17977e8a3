"""Substack pipeline: `full_roi_inference` and its helpers
(reference `flypylib/fplobjdetect.py:841-1216`).

The reference streams substacks (size + 2*buffer cubes) out of DVID / DICED / n5,
normalises each one on the host (`fri_get_image`), runs `network.infer` and hands
the prediction to a forked `voxel2obj` worker, with one pickle per substack as
the resume point.  Here a substack never leaves the GPU between those steps:

    uint8 cube (HBM)  ->  histogram -> mn_use          (fpl_histogram_u8)
                      ->  fpl_infer_volume(u8, mn_use, std) -> float32 prediction (HBM)
                      ->  fpl_v2o_smooth / fpl_v2o_nms  -> points (host, a few KB)

and only the point list and the normalisation record go back to the host.  Two
lanes share the GPU: the calling thread cuts (or synthesises) and infers substack
i+1 on the network's context while a second thread post-processes substack i on a
context of its own (own HIP stream, double-buffered predictions); a third thread
cuts the next host cube for array sources.
File formats (ROI text, `<size>_<z>_<y>_<x>.p` pickles, `norm/*.txt`, `all.p`)
are the reference's, so an interrupted run resumes and downstream tools read the
results unchanged.

Data sources.  DVID, DICED and n5 readers (`libdvid`, `diced`, `z5py`) are not
available offline; `data_source` is instead
  * an array-like `(Z,Y,X)` uint8 with numpy slicing (ndarray, `np.memmap`, ...),
  * a uint8 volume already RESIDENT on the GPU (a contiguous torch CUDA tensor or a
    `_capi.DeviceBuffer`): substacks are cut device to device,
  * `'npy://<file>'` (opened with `np.load(mmap_mode='r')`), or
  * `'synth://<seed>,<Z>,<Y>,<X>'`, the counter-hash EM volume of `synth.py`,
    generated on the device substack by substack (nothing is stored);
any other URL raises `NotImplementedError` naming the missing reader.
Multi-GPU: one process per GPU (`torch.distributed` if initialised, else
RANK/WORLD_SIZE); substacks go round-robin to ranks, no collective on the data
path; rank 0 merges the per-substack pickles.
"""
import os
import pickle
import sys
import threading
from collections import namedtuple

import numpy as np

from . import fplobjdetect

szyx = namedtuple('szyx', 'size z y x')          # fplobjdetect.py:24-25


# ---- ROI text files (fplobjdetect.py:1172-1216) ------------------------------------
def roi_from_txt(filename):
    with open(filename, 'r') as f_in:
        substacks = f_in.read().splitlines()
    roi = []
    for ss in substacks:
        roi.append(szyx(*[int(nn) for nn in ss.split(',')]))
    return [roi, ]


def gen_full_tab_roi(filename, store_name, repo_uuid=None,
                     crop=(None, None, None), n_splits=1, step_size=512):
    """write '<filename>_%02d.txt' ROI files covering the volume in `step_size`
    substacks (z outer, x inner), split into `n_splits` files of equal length
    (reference :1172-1208; `store_name` is a data source as in `full_roi_inference`,
    whose extents start at 0)"""
    import itertools
    extents = _open_source(store_name).extent
    lo_hi = [tuple(c) if c is not None else (0, int(e)) for c, e in zip(crop, extents)]
    starts = [range(lo, hi, step_size) for lo, hi in lo_hi]
    n_total = int(np.prod([np.ceil((hi - lo) / float(step_size)) for lo, hi in lo_hi]))
    per_file = int(np.ceil(n_total / n_splits))
    lines = ['%d,%d,%d,%d\n' % (step_size, z, y, x) for z, y, x in itertools.product(*starts)]
    # the reference opens the next file as soon as one is full, so an exact split
    # leaves one empty trailing file: kept, downstream globbing may count on it
    n_files = len(lines) // per_file + 1
    for idx in range(n_files):
        with open('%s_%02d.txt' % (filename, idx), 'w') as f_out:
            f_out.writelines(lines[idx * per_file:(idx + 1) * per_file])


def fri_filename(working_dir, substack):
    return '%s/%d_%d_%d_%d.p' % (working_dir, substack.size,
                                 substack.z, substack.y, substack.x)


# ---- data sources ------------------------------------------------------------------
class _ArraySource:
    def __init__(self, arr):
        assert len(arr.shape) == 3, 'volume must be (Z,Y,X)'
        self.arr = arr
        self.extent = tuple(int(s) for s in arr.shape)

    def cube_host(self, origin, size, dtype=np.uint8):
        """(size,)*3 cube at `origin`, zeros outside the extents; None if the box does
        not meet the volume (reference :1053-1070)"""
        lo = np.maximum(origin, 0)
        hi = np.minimum(np.asarray(origin) + size, self.extent)
        if np.any(lo > hi):
            return None
        image = np.zeros((size,) * 3, dtype)
        image[lo[0] - origin[0]:hi[0] - origin[0],
              lo[1] - origin[1]:hi[1] - origin[1],
              lo[2] - origin[2]:hi[2] - origin[2]] = self.arr[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        return image



class _SynthSource:
    def __init__(self, seed, extent):
        self.seed = int(seed)
        self.extent = tuple(int(e) for e in extent)

    def cube_host(self, origin, size):
        from . import synth
        lo = np.maximum(origin, 0)
        hi = np.minimum(np.asarray(origin) + size, self.extent)
        if np.any(lo > hi):
            return None
        image = np.zeros((size,) * 3, np.uint8)
        if np.all(hi > lo):
            image[lo[0] - origin[0]:hi[0] - origin[0],
                  lo[1] - origin[1]:hi[1] - origin[1],
                  lo[2] - origin[2]:hi[2] - origin[2]] = synth.em_volume_u8(
                      self.seed, tuple(int(v) for v in hi - lo), tuple(int(v) for v in lo))
        return image

    def cube_device(self, ctx, origin, size, dst):
        lo = np.maximum(origin, 0)
        hi = np.minimum(np.asarray(origin) + size, self.extent)
        if np.any(lo > hi):
            return False
        ctx.synth_substack_u8(self.seed, self.extent, (size,) * 3, origin, dst)
        return True


class _DeviceSource:
    """a uint8 (Z,Y,X) volume RESIDENT in the GPU's memory (a torch CUDA tensor or a
    `_capi.DeviceBuffer`): substacks are cut device to device (`fpl_crop_substack_u8`), the
    host never sees the grayscale.  One MI355X holds a 4096^3 ROI (64 GiB) whole."""

    def __init__(self, vol):
        assert len(vol.shape) == 3, 'volume must be (Z,Y,X)'
        if hasattr(vol, 'is_contiguous'):
            import torch
            assert vol.dtype == torch.uint8 and vol.is_contiguous(), \
                'a device volume must be contiguous uint8'
            self.device = vol.device.index or 0
        else:
            assert np.dtype(vol.dtype) == np.uint8, 'a device volume must be uint8'
            self.device = vol.ctx.device
        self.vol = vol
        self.extent = tuple(int(s) for s in vol.shape)

    def cube_host(self, origin, size):
        raise NotImplementedError('a device-resident volume is read on the device '
                                  '(full_roi_inference); fri_get_image needs a host source')

    def cube_device(self, ctx, origin, size, dst):
        assert ctx.device == self.device, \
            'the volume lives on GPU %d, the network on GPU %d' % (self.device, ctx.device)
        lo = np.maximum(origin, 0)
        hi = np.minimum(np.asarray(origin) + size, self.extent)
        if np.any(lo > hi):
            return False
        ctx.crop_substack_u8(self.vol, self.extent, (size,) * 3, origin, dst)
        return True


def _is_device_volume(v):
    from . import _capi
    return isinstance(v, _capi.DeviceBuffer) or bool(getattr(v, 'is_cuda', False))


def _open_source(data_source):
    if isinstance(data_source, (_ArraySource, _SynthSource, _DeviceSource)):
        return data_source
    if _is_device_volume(data_source):
        return _DeviceSource(data_source)
    if isinstance(data_source, str):
        if data_source.startswith('npy://'):
            return _ArraySource(np.load(data_source[6:], mmap_mode='r'))
        if data_source.startswith('synth://'):
            seed, z, y, x = (int(v) for v in data_source[8:].split(','))
            return _SynthSource(seed, (z, y, x))
        reader = ('z5py' if data_source.startswith('n5://') else
                  'diced' if data_source.startswith('gs://') else 'libdvid')
        raise NotImplementedError(
            'data source %r needs %s, which is not installed; pass an array, '
            "'npy://file' or 'synth://seed,Z,Y,X'" % (data_source, reader))
    return _ArraySource(data_source)


# ---- substack normalisation (fri_get_image, fplobjdetect.py:1088-1124) ----------------
def normalisation_from_histogram(hist, image_normalize):
    """-> dict(mn_use, global_frac, im_flt_mn, im_flt_std, im_raw_mn, im_raw_std).
    np.mean of a uint8 array (and of its 1 < v < 200 subset) is the exact integer
    sum over the count, so the 256-bin histogram reproduces it bit for bit."""
    hist = np.asarray(hist, np.float64)
    v = np.arange(256, dtype=np.float64)

    def stats(h):
        n = h.sum()
        mn = (h * v).sum() / n
        return mn, np.sqrt((h * (v - mn) ** 2).sum() / n)
    im_raw_mn, im_raw_std = stats(hist)
    flt = hist.copy()
    flt[:2] = 0                       # image > 1
    flt[200:] = 0                     # image < 200
    if flt.sum() > 0:
        im_flt_mn, im_flt_std = stats(flt)
    else:
        im_flt_mn, im_flt_std = image_normalize[0], image_normalize[1]
    global_frac = 1. if len(image_normalize) < 3 else image_normalize[2]
    mn_use = global_frac * image_normalize[0] + (1 - global_frac) * im_flt_mn
    return dict(mn_use=mn_use, global_frac=global_frac, im_flt_mn=im_flt_mn,
                im_flt_std=im_flt_std, im_raw_mn=im_raw_mn, im_raw_std=im_raw_std)


def _write_norm(norm_dir, substack, buffer_sz, image_normalize, st):
    norm_fn = '%s/%d_%d_%d_%d.txt' % (norm_dir, substack.size, substack.z,
                                      substack.y, substack.x)
    with open(norm_fn, 'w') as f_out:
        f_out.write('%d,%d,%d,%d,%d,%g,%g,%g,%g,%g,%g,%g,%g\n' %
                    (substack.size, buffer_sz, substack.z, substack.y, substack.x,
                     image_normalize[0], image_normalize[1], st['global_frac'],
                     st['mn_use'], st['im_flt_mn'], st['im_flt_std'],
                     st['im_raw_mn'], st['im_raw_std']))


def fri_get_image(substack_info, dvid_node, using_diced=True, instance_name='grayscale'):
    """host form of the reference's reader (:1023-1124): -> (float32 normalised image
    or None, substack).  `substack_info` = [substack, data_source, uuid,
    image_normalize, buffer_sz, local_cache_dir, norm_dir, instance_name];
    `dvid_node` is a data source (see module docstring).  `full_roi_inference`
    does the same on the device and never materialises this array."""
    substack, image_normalize, buffer_sz = substack_info[0], substack_info[3], substack_info[4]
    norm_dir = substack_info[6]
    src = _open_source(dvid_node)
    image_sz = substack.size + 2 * buffer_sz
    image_offset = [substack.z - buffer_sz, substack.y - buffer_sz, substack.x - buffer_sz]
    image = src.cube_host(image_offset, image_sz)
    if image is None:
        return (None, substack)
    st = normalisation_from_histogram(np.bincount(image.reshape(-1), minlength=256),
                                      image_normalize)
    image = (image.astype('float32') - np.float32(st['mn_use'])) / np.float32(image_normalize[1])
    if norm_dir is not None:
        _write_norm(norm_dir, substack, buffer_sz, image_normalize, st)
    return (image, substack)


# ---- the pipeline ------------------------------------------------------------------------
def _rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size(), dist
    except ImportError:
        pass
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), None


def full_roi_inference(data_source, dvid_uuid, dvid_roi,
                       network, thd, working_dir,
                       image_normalize,
                       obj_min_dist=27, smoothing_sigma=5,
                       buffer_sz=35, partition_size=16,
                       local_cache_dir=None,
                       roi_force_file=False,
                       instance_name='grayscale',
                       dvid_seg_info=None, precision=None, timings=None):
    """Predictions of a trained network within the substacks of an ROI, cached per
    substack in `working_dir` (reference :841-986; same arguments).  `dvid_roi` is
    an ROI text file (`roi_from_txt`) or a list of `szyx`; `precision` overrides the
    network's ('f32' / 'bf16').  Returns {'locs': (N,3) x/y/z, 'conf': (N,)} (all
    ranks return the merged result once every rank's substacks are on disk)."""
    # the reference reads a 'segmentation' labelmap from a second DVID node
    # (dvid_seg_info = [server, uuid]); here it is a label volume with the image's
    # extents: array-like or 'npy://file'
    seg_src = None
    if dvid_seg_info is not None:
        seg_src = _open_source(dvid_seg_info)
        assert isinstance(seg_src, _ArraySource), 'the segmentation must be an array source'
    for d in (working_dir, '%s/norm' % working_dir):
        os.makedirs(d, exist_ok=True)
    norm_dir = '%s/norm' % working_dir
    src = _open_source(data_source)
    roi = roi_from_txt(dvid_roi) if isinstance(dvid_roi, str) else [[szyx(*r) for r in dvid_roi]]
    rank, world, dist = _rank_world()

    todo = []
    num_processed = 0
    for rr in roi[0]:
        if os.path.isfile(fri_filename(working_dir, rr)):
            num_processed += 1
            continue
        todo.append(szyx(rr.size, rr.z, rr.y, rr.x))
    if dist is not None and world > 1:
        # every rank must split the SAME list: a rank that starts late would otherwise
        # see substacks the others have already written (control plane only)
        box = [todo, num_processed]
        dist.broadcast_object_list(box, src=0)
        todo, num_processed = box
    mine = todo[rank::world]
    if rank == 0:
        print('already processed: %d' % num_processed)
        print('to process: %d' % len(todo))

    assert network.infer_network is not None, 'network has not been trained'
    prog0 = network.infer_network.program
    dev = prog0.ctx.device
    prec = fplobjdetect_precision(network, precision)
    # A host volume (ndarray, np.memmap) that fits beside the work goes to the GPU ONCE - 3.6 GB for 1536^3, a few
    # hundred ms - and the substacks are cut out of it there: cutting 582^3 cubes on the host (a strided 197 MB
    # copy each) and uploading them one by one costs more than the inference.  FPL_PIPE_RESIDENT_GB (default 64;
    # 0: never) bounds the size; larger volumes keep the host path (cut + upload per substack, prefetched).
    uploaded = None
    if isinstance(src, _ArraySource) and mine and np.dtype(src.arr.dtype) == np.uint8:
        cap = float(os.environ.get('FPL_PIPE_RESIDENT_GB', '64')) * 2 ** 30
        nbytes = int(np.prod(src.extent))
        if 0 < nbytes <= cap:
            uploaded = prog0.ctx.malloc(src.extent, np.uint8)
            plane = int(src.extent[1]) * int(src.extent[2])
            step = max(1, (256 << 20) // max(plane, 1))            # 256 MiB slabs: a memmap is read as it goes
            for z0 in range(0, src.extent[0], step):
                z1 = min(src.extent[0], z0 + step)
                part = np.ascontiguousarray(src.arr[z0:z1])
                prog0.ctx.memcpy(uploaded.ptr + z0 * plane, part, part.nbytes)
            prog0.ctx.synchronize()
            src = _DeviceSource(uploaded)
    # Lanes on the GPU.  An inference lane (a host thread + context + copy of the program)
    # prepares and infers its substacks; each has a post-processing lane of its own (a
    # second thread + context: own HIP stream and voxel2obj state) that works on substack
    # i while the inference lane is on i + 1, predictions double-buffered.  Both kinds go
    # through host code between kernels (statistics, order statistics, NMS rounds), so one
    # pair leaves the GPU idle while both sit there (18 % of the span in a kernel trace):
    # FPL_PIPE_LANES pairs (default 3; 1536^3 on one GPU: 0.310 / 0.297 / 0.286 s with 1 / 2 / 3)
    # interleave their substacks.
    from . import _capi, runtime
    import queue
    n_lanes = max(1, min(int(os.environ.get('FPL_PIPE_LANES', '3')), 4, max(len(mine), 1)))
    failure = []
    done_count = [0]

    def write_result(ss, out):
        tmp_fn = fri_filename(working_dir, ss) + '.tmp%d' % rank
        with open(tmp_fn, 'wb') as f_out:
            pickle.dump(out, f_out)
        os.replace(tmp_fn, fri_filename(working_dir, ss))     # a resume never sees half a file

    def run_lane(lane, items):
        ctx = prog0.ctx if lane == 0 else runtime.get_context(dev, lane=2 * lane)
        prog = prog0 if lane == 0 else _capi.Program(ctx, network.infer_network.graph,
                                                     network.rf_stride)
        ctx_post = runtime.get_context(dev, lane=2 * lane + 1)
        bufs = {}

        def buffers(size):
            if size not in bufs:
                bufs[size] = (ctx.malloc((size,) * 3, np.uint8),
                              [ctx.malloc((size,) * 3, np.float32) for _ in range(2)],
                              [threading.Semaphore(1), threading.Semaphore(1)])
            return bufs[size]

        # host-side prefetch of the next cube (array sources); synthetic cubes are made
        # on the device and need none
        staged = {}

        def stage(k):
            if k < len(items) and isinstance(src, _ArraySource):
                ss = items[k]
                sz = ss.size + 2 * buffer_sz
                staged[k] = src.cube_host([ss.z - buffer_sz, ss.y - buffer_sz, ss.x - buffer_sz], sz)

        work = queue.Queue()

        def post_process():
            while True:
                item = work.get()
                if item is None:
                    return
                ss, pred, free, seg_kw = item
                try:
                    if not failure:
                        out = fplobjdetect.voxel2obj(
                            pred, obj_min_dist, smoothing_sigma,
                            (ss.x - buffer_sz, ss.y - buffer_sz, ss.z - buffer_sz),
                            buffer_sz, thd, _ctx=ctx_post, **seg_kw)
                        write_result(ss, out)
                except BaseException as e:           # surfaces in the main thread below
                    failure.append(e)
                finally:
                    free.release()

        poster = threading.Thread(target=post_process)
        poster.start()
        try:
            stage(0)
            for k, ss in enumerate(items):
                if failure:
                    break
                th = threading.Thread(target=stage, args=(k + 1,))
                th.start()
                image_sz = ss.size + 2 * buffer_sz
                origin = [ss.z - buffer_sz, ss.y - buffer_sz, ss.x - buffer_sz]
                cube, preds, sems = buffers(image_sz)
                if isinstance(src, _ArraySource):
                    image = staged.pop(k)
                    have = image is not None
                    if have:
                        cube.from_host(image)
                else:
                    have = src.cube_device(ctx, origin, image_sz, cube)
                if not have:
                    write_result(ss, {'locs': np.zeros((0, 3)), 'conf': np.zeros(0)})
                else:
                    st = normalisation_from_histogram(ctx.histogram_u8(cube), image_normalize)
                    _write_norm(norm_dir, ss, buffer_sz, image_normalize, st)
                    sems[k % 2].acquire()             # its previous prediction is post-processed
                    prog.infer_volume(cube, network.infer_sz, network.rf_offset,
                                      mean=st['mn_use'], std=image_normalize[1],
                                      precision=prec, dst=preds[k % 2], dims=(image_sz,) * 3)
                    seg_kw = {}
                    if seg_src is not None:           # fri_postprocess, reference :1143-1150
                        seg_dt = seg_src.arr.dtype if seg_src.arr.dtype.itemsize in (4, 8) \
                            else np.uint64
                        seg_kw = dict(seg=seg_src.cube_host(origin, image_sz, seg_dt), seg_dilate=8,
                                      seg_sz_thd=5000, seg_force=10)
                    work.put((ss, preds[k % 2], sems[k % 2], seg_kw))
                th.join()
                done_count[0] += 1
                if rank == 0 and sys.stdout.isatty():
                    sys.stdout.write('\r%d' % done_count[0])
                    sys.stdout.flush()
        except BaseException as e:
            failure.append(e)
        finally:
            work.put(None)
            poster.join()
            for cube, preds, _ in bufs.values():
                cube.free()
                for pbuf in preds:
                    pbuf.free()
            if lane != 0:
                prog.close()

    lanes = [threading.Thread(target=run_lane, args=(l, mine[l::n_lanes]))
             for l in range(1, n_lanes)]
    for t in lanes:
        t.start()
    run_lane(0, mine[0::n_lanes])
    for t in lanes:
        t.join()
    if uploaded is not None:
        uploaded.free()
    if failure:
        raise failure[0]
    if timings is not None:
        timings['substacks'] = done_count[0]
        timings['lanes'] = n_lanes

    if dist is not None and world > 1:
        dist.barrier()
    locs, conf = [], []
    missing = 0
    for rr in roi[0]:
        ff = fri_filename(working_dir, rr)
        if not os.path.isfile(ff):
            if dist is None and world > 1:
                # WORLD_SIZE > 1 from the environment but no process group: the other
                # ranks' substacks cannot be waited for - this rank returns its own part
                # and all.p is NOT written (a silently incomplete merge is worse than none)
                missing += 1
                continue
            raise RuntimeError('substack result %s is missing' % ff)
        with open(ff, 'rb') as f_in:
            obj = pickle.load(f_in)
        locs.append(obj['locs'])
        conf.append(obj['conf'])
    obj = {'locs': np.concatenate(locs) if locs else np.zeros((0, 3)),
           'conf': np.concatenate(conf) if conf else np.zeros(0)}
    if missing:
        import warnings
        warnings.warn('full_roi_inference: %d of %d substacks belong to other ranks and there '
                      'is no initialised process group to wait for them: all.p not written; '
                      'run again once every rank has finished (finished substacks are reused)'
                      % (missing, len(roi[0])))
    elif rank == 0:
        with open('%s/all.p' % working_dir, 'wb') as f_out:
            pickle.dump(obj, f_out)
    return obj


def fplobjdetect_precision(network, precision):
    from .fplnetwork import _PRECISIONS
    return _PRECISIONS[precision or network.precision]
