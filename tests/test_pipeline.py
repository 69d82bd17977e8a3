"""Substack pipeline (reference fplobjdetect.py:841-1216): the host pieces and the
oracle against the reference's own fri_get_image outputs (CPU), and the
device-resident full_roi_inference against the oracle (GPU)."""
import os
import pickle

import numpy as np

from tests import helpers
import pytest

from flypylib_amd import fplobjdetect, fplpipeline, synth
from oracle import pipeline_oracle

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'fri_get_image.npz'))
FRI_VOLUME = (11, (20, 24, 28))
FRI_CASES = [
    ('interior', 8, 6, 8, 10, 4, [128., 33.]),
    ('interior_frac', 8, 6, 8, 10, 4, [128., 33., 0.3]),
    ('clip_low', 8, 0, 0, 0, 4, [120., 30., 0.5]),
    ('clip_high', 8, 16, 16, 24, 4, [128., 33., 0.0]),
    ('outside', 8, 200, 0, 0, 4, [128., 33.]),
]


def fri_volume():
    vol = synth.em_volume_u8(FRI_VOLUME[0], FRI_VOLUME[1])
    vol[::5, ::3, ::7] = 0
    vol[1::4, ::5, 2::3] = 255
    vol[2::6, 1::4, ::5] = 1
    vol[::7, 2::5, 1::3] = 200
    return vol


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize('case', FRI_CASES, ids=[c[0] for c in FRI_CASES])
def test_fri_get_image_matches_the_reference(case, tmp_path):
    name, size, z, y, x, buf, norm = case
    vol = fri_volume()
    assert _sha(vol) == str(GOLD['volume_sha'])
    o_img, o_rec = pipeline_oracle.fri_get_image(vol, size, z, y, x, buf, norm, float32_math=False)
    info = [fplobjdetect.szyx(size, z, y, x), 'unused', 'uuid', norm, buf, None,
            str(tmp_path), 'grayscale']
    h_img, ss = fplobjdetect.fri_get_image(info, vol)
    assert ss == fplobjdetect.szyx(size, z, y, x)
    if name + '/none' in GOLD:
        assert o_img is None and h_img is None
        return
    ref = GOLD[name + '/image']
    # oracle in the reference's arithmetic as run here (numpy 2: float64)
    assert np.array_equal(np.asarray(o_img, np.float64), ref)
    line = str(GOLD[name + '/norm_line'])
    assert pipeline_oracle.norm_line(size, buf, z, y, x, norm, o_rec) == line
    # the host function computes in float32 (the reference's numpy-1.13 semantics and
    # what the device does): equal to the float64 result rounded, within one ulp
    assert h_img.dtype == np.float32
    # (mn_use ~ 128 rounds to float32 with up to 3.8e-6 error, / std)
    np.testing.assert_allclose(h_img, ref, rtol=2e-7, atol=3e-7)
    o32, _ = pipeline_oracle.fri_get_image(vol, size, z, y, x, buf, norm)
    assert np.array_equal(h_img, o32)
    got_line = open('%s/%d_%d_%d_%d.txt' % (tmp_path, size, z, y, x)).read()
    assert got_line == line


def test_normalisation_from_histogram_is_exact():
    vol = fri_volume()
    for norm in ([128., 33.], [100., 20., 0.25]):
        st = fplpipeline.normalisation_from_histogram(np.bincount(vol.reshape(-1), minlength=256), norm)
        idx = (vol < 200) & (vol > 1)
        g = 1. if len(norm) < 3 else norm[2]
        assert st['im_raw_mn'] == np.mean(vol) and st['im_flt_mn'] == np.mean(vol[idx])
        assert st['mn_use'] == g * norm[0] + (1 - g) * np.mean(vol[idx])
        assert abs(st['im_flt_std'] - np.std(vol[idx])) < 1e-9
    empty = fplpipeline.normalisation_from_histogram(np.bincount([0, 1, 255], minlength=256), [5., 2., 0.5])
    assert empty['im_flt_mn'] == 5. and empty['mn_use'] == 5.


def test_roi_files_round_trip(tmp_path):
    vol = np.zeros((100, 130, 64), np.uint8)
    base = str(tmp_path / 'roi')
    fplobjdetect.gen_full_tab_roi(base, vol, None, n_splits=2, step_size=48)
    a = fplobjdetect.roi_from_txt(base + '_00.txt')[0]
    b = fplobjdetect.roi_from_txt(base + '_01.txt')[0]
    assert len(a) + len(b) == 3 * 3 * 2 and len(a) == 9
    assert a[0] == fplobjdetect.szyx(48, 0, 0, 0) and a[1] == fplobjdetect.szyx(48, 0, 0, 48)
    assert b[-1] == fplobjdetect.szyx(48, 96, 96, 48)
    assert fplobjdetect.fri_filename('/w', a[1]) == '/w/48_0_0_48.p'
    with pytest.raises(NotImplementedError, match='libdvid'):
        fplobjdetect.gen_full_tab_roi(base, 'http://dvid:8000', 'uuid')
    with pytest.raises(NotImplementedError, match='z5py'):
        fplpipeline._open_source('n5://some/where')


def test_synth_source_matches_the_host_generator():
    src = fplpipeline._open_source('synth://7,40,50,60')
    cube = src.cube_host([-4, 30, 50], 24)
    ref = np.zeros((24, 24, 24), np.uint8)
    ref[4:, :20, :10] = synth.em_volume_u8(7, (20, 20, 10), (0, 30, 50))
    assert np.array_equal(cube, ref)
    assert src.cube_host([41, 0, 0], 8) is None


# ---- GPU --------------------------------------------------------------------------
def _small_setup():
    from flypylib_amd import FplNetwork, fplmodels
    net = FplNetwork(fplmodels.vgg_like)
    synth.synthetic_weights(net.train_single, 21)
    net.infer_sz = (38, 38, 38)           # small tiles: 24^3 outputs
    net._set_infer()
    vol = synth.em_volume_u8(5, (70, 90, 80))
    roi = [(32, z, y, x) for z in (0, 32, 64) for y in (0, 32, 64) for x in (0, 32, 64)]
    roi.append((32, 400, 0, 0))            # outside the volume: an empty result
    return net, vol, roi


@pytest.mark.gpu
def test_full_roi_inference_matches_the_oracle(ctx, tmp_path):
    """device-resident pipeline vs the CPU oracle on a 70x90x80 volume cut into 28
    substacks of 32 + buffer 10 (faces clipped, one substack outside): the fp32
    predictions agree to 1e-5, and on each substack's device prediction the oracle's
    voxel2obj returns the identical point list"""
    from oracle import cnn_oracle
    net, vol, roi = _small_setup()
    wd = str(tmp_path / 'work')
    kw = dict(obj_min_dist=5, smoothing_sigma=1.5, buffer_sz=10)
    norm = [128., 33., 0.7]
    got = fplobjdetect.full_roi_inference(vol, None, roi, net, 0.2, wd, norm, precision='f32', **kw)
    assert os.path.isfile(wd + '/all.p') and os.path.isfile(wd + '/norm/32_0_0_0.txt')
    graph = net.train_single

    def predict(batch):
        return cnn_oracle.graph_forward(graph, batch.astype(np.float32), upsample_stride=net.rf_stride)
    # (a) end to end through the oracle CNN
    want, per = pipeline_oracle.full_roi_inference(vol, roi, predict, net.infer_sz, net.rf_offset,
                                                   0.2, norm, **kw)
    assert got['locs'].shape == want['locs'].shape and got['locs'].shape[0] > 20
    assert np.array_equal(got['locs'], want['locs'])
    np.testing.assert_allclose(got['conf'], want['conf'], rtol=0, atol=2e-6)
    # per-substack files hold the same as the oracle's per-substack results
    for (size, z, y, x), o in per.items():
        with open(fplobjdetect.fri_filename(wd, fplobjdetect.szyx(size, z, y, x)), 'rb') as f:
            g = pickle.load(f)
        assert np.array_equal(g['locs'], o['locs'])
    # the norm record is the reference's line
    _, rec = pipeline_oracle.fri_get_image(vol, 32, 0, 0, 0, 10, norm)
    assert open(wd + '/norm/32_0_0_0.txt').read() == pipeline_oracle.norm_line(32, 10, 0, 0, 0, norm, rec)
    # (b) the DEFAULT precision ('auto': split IEEE halves for vgg_like): the same point
    # set end to end - identical voxels, confidences at fp32 rounding level, the same order
    # except between confidences that tie to 2e-6 (untrained weights: a near-flat field with
    # 2163 peaks; the split path differs from the fp32 one by up to 8e-7 in probability)
    auto = fplobjdetect.full_roi_inference(vol, None, roi, net, 0.2, str(tmp_path / 'auto'), norm, **kw)
    assert ctx.last_path() in ('vgg_split_f16', 'none')
    moved = helpers.same_detections(auto, want, 2e-6)
    assert moved <= len(want['conf']) // 50, moved


@pytest.mark.gpu
def test_full_roi_inference_resumes_and_bf16_synth_source(ctx, tmp_path):
    """(1) finished substacks are not recomputed (their pickles are read back);
    (2) the synthetic device source + bf16 path give the oracle's points when the
    oracle post-processes the device predictions (bit-exact voxel2obj)"""
    net, vol, roi = _small_setup()
    wd = str(tmp_path / 'work')
    kw = dict(obj_min_dist=5, smoothing_sigma=1.5, buffer_sz=10)
    norm = [128., 33.]
    src = 'synth://5,70,90,80'
    first = fplobjdetect.full_roi_inference(src, None, roi[:5], net, 0.2, wd, norm, precision='bf16', **kw)
    # plant a marker in a finished substack: a resumed run must keep it
    ff = fplobjdetect.fri_filename(wd, fplobjdetect.szyx(*roi[0]))
    with open(ff, 'rb') as f:
        obj = pickle.load(f)
    obj['conf'] = obj['conf'] + 100.0
    with open(ff, 'wb') as f:
        pickle.dump(obj, f)
    full = fplobjdetect.full_roi_inference(src, None, roi, net, 0.2, wd, norm, precision='bf16', **kw)
    n0 = len(obj['conf'])
    assert n0 > 0 and np.all(full['conf'][:n0] > 100.0) and np.all(full['conf'][n0:] < 2.0)
    assert len(full['conf']) > len(first['conf'])
    # same volume from the host array == the synthetic source (bit-identical generator)
    wd2 = str(tmp_path / 'work2')
    again = fplobjdetect.full_roi_inference(vol, None, roi, net, 0.2, wd2, norm, precision='bf16', **kw)
    assert np.array_equal(again['locs'][n0:], full['locs'][n0:])
    assert np.array_equal(again['conf'][n0:], full['conf'][n0:])


@pytest.mark.gpu
def test_device_resident_source_cuts_the_cubes_the_host_reader_cuts(ctx, tmp_path):
    """fpl_crop_substack_u8 == fri_get_image's crop + zero padding (reference
    fplobjdetect.py:1044-1070), and full_roi_inference over a volume resident in HBM
    (torch tensor and DeviceBuffer) returns the host-array run's detections"""
    import torch
    net, vol, roi = _small_setup()
    host = fplpipeline._ArraySource(vol)
    dvol = torch.from_numpy(vol).cuda()
    src = fplpipeline._open_source(dvol)
    assert isinstance(src, fplpipeline._DeviceSource) and src.extent == vol.shape
    for origin, size in [((-10, -10, -10), 52), ((0, 0, 0), 33), ((31, 47, 39), 41), ((5, -3, 60), 27),
                         ((-200, 0, 0), 20), ((69, 89, 79), 9), ((3, 2, 1), 4)]:
        ref = host.cube_host(list(origin), size)
        out = ctx.malloc((size,) * 3, np.uint8)
        have = src.cube_device(ctx, list(origin), size, out)
        assert have == (ref is not None)
        if have:
            assert np.array_equal(out.to_host(), ref), (origin, size)
        out.free()
    kw = dict(obj_min_dist=5, smoothing_sigma=1.5, buffer_sz=10)
    norm = [128., 33., 0.5]
    want = fplobjdetect.full_roi_inference(vol, None, roi, net, 0.2, str(tmp_path / 'h'), norm, **kw)
    got = fplobjdetect.full_roi_inference(dvol, None, roi, net, 0.2, str(tmp_path / 'd'), norm, **kw)
    assert len(want['conf']) > 0
    assert np.array_equal(got['locs'], want['locs']) and np.array_equal(got['conf'], want['conf'])
    dbuf = ctx.malloc(vol.shape, np.uint8).from_host(vol)
    got2 = fplobjdetect.full_roi_inference(dbuf, None, roi, net, 0.2, str(tmp_path / 'b'), norm, **kw)
    assert np.array_equal(got2['locs'], want['locs']) and np.array_equal(got2['conf'], want['conf'])
    dbuf.free()
    with pytest.raises(NotImplementedError):
        src.cube_host([0, 0, 0], 8)


@pytest.mark.gpu
def test_evaluate_substacks_usage_contract(ctx, tmp_path):
    """scripts/fpl_cx1_0_unet_4ss_all.py:60-66: evaluate_substacks(network, [[image,
    json], ...], thresholds, obj_min_dist, smoothing_sigma).  Ground truth = the
    network's own detections above a confidence cut, so precision / recall are known."""
    from flypylib_amd import fplsynapses
    net, vol, _ = _small_setup()
    img = (vol.astype(np.float32) - 128) / 33
    pred = net.infer(img)
    det = fplobjdetect.voxel2obj(pred, 5, 1.5, (0, 0, 0), 5)
    assert len(det['conf']) > 20
    cut = float(np.median(det['conf']))
    keep = det['conf'] >= cut
    gt_json = str(tmp_path / 'gt.json')
    fplsynapses.tbars_to_json_format({'locs': det['locs'][keep], 'conf': det['conf'][keep]}, gt_json)
    thds = np.array([0.0, cut, 2.0])
    agg, per = fplobjdetect.evaluate_substacks(net, [[img, gt_json], [img, gt_json]], thds,
                                               obj_min_dist=5, smoothing_sigma=1.5, buffer_sz=5)
    # the reference applies the ground-truth buffer with pred.shape (z,y,x) against
    # (x,y,z) columns (fplobjdetect.py:470) - kept literally, so count what it keeps
    n_gt = len(fplsynapses.load_from_json(gt_json, pred.shape, 5)['conf'])
    n_keep = int(keep.sum())
    assert 0 < n_gt <= n_keep
    assert len(per) == 2 and per[0].tot_gt[0] == n_gt
    # every ground-truth point is one of the detections: recall 1 at both thresholds
    assert per[0].num_tp[0] == n_gt and per[0].tot_pred[0] == len(det['conf'])
    assert per[0].num_tp[1] == n_gt and per[0].tot_pred[1] == n_keep
    assert per[0].tot_pred[2] == 0 and per[0].pp[2] == 1 and per[0].rr[2] == 0
    assert np.allclose(agg.num_tp, 2 * per[0].num_tp) and abs(agg.rr[0] - 1.0) < 1e-6


@pytest.mark.gpu
def test_documented_example_flow_on_synthetic_data(ctx, tmp_path):
    """the README's entry point of the reference (scripts/fpl_fib25_example.py) end to
    end with flypylib_amd on synthetic blobs: write_labels_mask -> gen_volume2 ->
    FplNetwork(unet_like2).train (masked focal loss) -> save -> evaluate_substacks.
    The trained net must find the planted points of an UNSEEN region."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'example_flow', os.path.join(os.path.dirname(os.path.dirname(__file__)), 'tools',
                                     'example_flow.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(['--size', '96', '--steps', '120', '--epochs', '2', '--out', str(tmp_path)])
    for name in ('train', 'test'):
        agg = out[name]
        best = int(np.argmax(agg.pp * agg.rr))
        assert agg.tot_gt[0] == 125
        assert agg.pp[best] > 0.9 and agg.rr[best] > 0.9, (name, agg.pp, agg.rr)
    rows = open(str(tmp_path / 'log.csv')).read().strip().splitlines()
    assert float(rows[2].split(',')[3]) < float(rows[1].split(',')[3])      # loss falls
    assert (tmp_path / 'net.npz').exists() or (tmp_path / 'net').exists() or \
        any(p.name.startswith('net') for p in tmp_path.iterdir())


@pytest.mark.gpu
def test_full_roi_inference_with_a_segmentation(ctx, tmp_path):
    """dvid_seg_info: per substack the label cube goes to voxel2obj with
    seg_dilate 8, seg_sz_thd 5000, seg_force 10 as in fri_postprocess (reference
    :1139-1150); the oracle post-processes the device predictions identically"""
    net, vol, roi = _small_setup()
    seg = synth.voronoi_segmentation(3, vol.shape, 30, 10)
    kw = dict(obj_min_dist=11, smoothing_sigma=2.0, buffer_sz=12)   # seg_force 10 needs r >= 10
    norm = [128., 33.]
    wd = str(tmp_path / 'seg')
    got = fplobjdetect.full_roi_inference(vol, None, roi[:8], net, 0.2, wd, norm, precision='f32',
                                          dvid_seg_info=seg, **kw)
    plain = fplobjdetect.full_roi_inference(vol, None, roi[:8], net, 0.2, str(tmp_path / 'plain'),
                                            norm, precision='f32', **kw)
    from oracle import cnn_oracle
    graph = net.train_single

    def predict(batch):
        return cnn_oracle.graph_forward(graph, batch.astype(np.float32), upsample_stride=net.rf_stride)
    want, _ = pipeline_oracle.full_roi_inference(vol, roi[:8], predict, net.infer_sz, net.rf_offset,
                                                 0.2, norm, seg=seg, **kw)
    assert np.array_equal(got['locs'], want['locs'])
    np.testing.assert_allclose(got['conf'], want['conf'], rtol=0, atol=2e-6)
    assert len(got['conf']) != len(plain['conf'])


def _roi_rank_worker(rank, world, port, wd, out_q):
    """one rank of the substack-sharded pipeline (every rank on the one GPU of the box)"""
    import traceback
    try:
        import torch.distributed as dist
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        os.environ['LOCAL_RANK'] = '0'
        dist.init_process_group('gloo', rank=rank, world_size=world)
        net, vol, roi = _small_setup()
        out = fplobjdetect.full_roi_inference(vol, None, roi, net, 0.2, wd, [128., 33., 0.7],
                                              obj_min_dist=5, smoothing_sigma=1.5, buffer_sz=10)
        out_q.put((rank, 'ok', (np.asarray(out['locs']), np.asarray(out['conf']))))
        dist.destroy_process_group()
    except Exception:                                   # the parent reports it and ends the peers
        out_q.put((rank, 'error', traceback.format_exc()))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_full_roi_inference_two_ranks_equals_one_process(ctx, tmp_path):
    """substacks go round-robin over the ranks (todo[rank::world]), every rank writes its
    substack pickles, rank 0 merges after a barrier: no data-path collective, and the
    merged list is the single-process list (reference: one process walks the ROI,
    fplobjdetect.py:1000-1090)"""
    import multiprocessing as mp
    import socket
    net, vol, roi = _small_setup()
    kw = dict(obj_min_dist=5, smoothing_sigma=1.5, buffer_sz=10)
    one = fplobjdetect.full_roi_inference(vol, None, roi, net, 0.2, str(tmp_path / 'one'),
                                          [128., 33., 0.7], **kw)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mctx = mp.get_context('spawn')
    q = mctx.Queue()
    wd = str(tmp_path / 'two')
    procs = [mctx.Process(target=_roi_rank_worker, args=(r, 2, port, wd, q), daemon=True)
             for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(2):
            rank, status, payload = q.get(timeout=240)
            assert status == 'ok', 'rank %d failed:\n%s' % (rank, payload)
            got[rank] = payload
        for p in procs:
            p.join(60)
    finally:
        for p in procs:                                 # a peer of a failed rank sits in a barrier
            if p.is_alive():
                p.terminate()
    for rank in (0, 1):                                 # every rank returns the merged result
        locs, conf = got[rank]
        assert len(conf) == len(one['conf']) > 0
        assert np.array_equal(locs, np.asarray(one['locs']))
        assert np.array_equal(conf, np.asarray(one['conf']))
    # each substack was written exactly once
    done = [f for f in os.listdir(wd) if f.endswith('.p') and f != 'all.p']
    assert len(done) == len(roi)
