#!/usr/bin/env python3
"""Secondary measurements for the other BASELINE.json configs (run on the GPU box):

    python tools/bench_configs.py --what unet,train,v2o,pipeline [--out file.json]

  unet      configs[2] unet_like2 inference (reference lattice 100^3, pitch 82) on a
            reduced volume (the fp32 per-op path; size via --unet-size)
  train     configs[3] vgg_like training step, batch 32 of 64^3 patches (1 GPU)
  v2o       voxel2obj on a 582^3 substack-sized probability volume (r=27, sigma=5)
  graphs    baseline_model, resnet_like, unet_like4b, unet_like_vol: 'auto' (graph executor) vs fp32
  roi       configs[4] end to end: fplobjdetect.full_roi_inference over a synthetic
            --roi-size^3 volume (512-substacks + 35 buffer), one substack's points
            diffed against the CPU oracle
  pipeline  configs[4] shape at one substack: vgg_like bf16 inference of a 582^3
            substack + voxel2obj, detections diffed against the CPU oracle on the
            same prediction
  c3share   configs[2] at its stated size: rank 0's slab of the 1024 x 2048 x 2048 volume
            (2 of 13 tile rows + halo), bit-compared with the same rows of a whole-volume run
  c5share   configs[4] at its stated size: rank 0's 64 of the 512 substacks of a 4096^3
            synthetic ROI, one substack diffed against the CPU oracle
These are NOT the driver's bench line (bench.py); they document where the other
rows of SURVEY section 8 stand.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--what', default='unet,train,v2o,pipeline')
    ap.add_argument('--unet-size', type=int, default=264)
    ap.add_argument('--roi-precision', default='auto')
    ap.add_argument('--vgg2-size', type=int, default=1024)
    ap.add_argument('--sub', type=int, default=582)
    ap.add_argument('--roi-size', type=int, default=1536)
    ap.add_argument('--roi-source', default='synth', choices=['synth', 'resident'],
                    help="roi: 'synth' generates every substack + buffer on the device inside the timed "
                         "region (the stand-in for the reference's DVID fetch); 'resident' keeps the whole "
                         "volume in HBM (generated before the clock starts) and cuts the substacks out of it")
    ap.add_argument('--skip-oracle', action='store_true',
                    help='roi: do not re-derive one substack on the CPU oracle (for traces)')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    import torch
    from flypylib_amd import _capi, fplmodels, fplobjdetect, runtime, synth
    ctx = runtime.get_context(0)      # the context voxel2obj uses as well
    res = {'device': ctx.device_info()['name']}
    what = a.what.split(',')

    if 'unet' in what:
        g = fplmodels.unet_like2(100)[0]
        synth.synthetic_weights(g, 7)
        prog = _capi.Program(ctx, g, (1, 1, 1))
        n = a.unet_size
        src = torch.empty((n, n, n), dtype=torch.uint8, device='cuda')
        dst = torch.empty((n, n, n), dtype=torch.float32, device='cuda')
        ctx.synth_volume_u8(3, (n, n, n), out=src)
        for pname, prec in (('bf16_mfma', _capi.PREC_BF16), ('f16_mfma', _capi.PREC_F16),
                            ('f16s_split', _capi.PREC_F16S), ('f32_mfma', _capi.PREC_F32)):
            kw = dict(mean=128.0, std=33.0, precision=prec, dst=dst, dims=(n, n, n))
            prog.infer_volume(src, (100,) * 3, (9,) * 3, **kw)
            ctx.synchronize()
            ctx.timing(True); ctx.timing_reset()
            t0 = time.perf_counter()
            prog.infer_volume(src, (100,) * 3, (9,) * 3, **kw)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            vox = (n - 18) ** 3
            res['unet_like2_' + pname] = dict(
                volume=n, mvox_s=vox / dt / 1e6, seconds=dt,
                tflops_algorithmic=vox * 350720 / dt / 1e12,
                kernels={k: round(v['ms'], 2) for k, v in ctx.timing_get().items()})
            ctx.timing(False)
            print(json.dumps(res['unet_like2_' + pname]), flush=True)

    if 'graphs' in what:
        # the four factories of the graph executor (csrc/gx_exec.h) at their own infer_sz, a volume of
        # 5 - 7 tiles per axis: 'auto' (split halves, op by op) against the fp32 MFMA executor
        from flypylib_amd import fplutils
        for name, tiles in (('baseline_model', 6), ('resnet_like', 6), ('unet_like4b', 7), ('unet_like_vol', 5)):
            factory = getattr(fplmodels, name)
            _, rf, infer_sz, _ = factory()
            tile = fplutils.to3d(infer_sz)[0]
            off = fplutils.to3d(rf[1])[0]
            stride = fplutils.to3d(rf[2])
            g = factory(tile)[0]
            synth.synthetic_weights(g, 7)
            prog = _capi.Program(ctx, g, stride)
            n = tiles * (tile - 2 * off) + 2 * off
            src = torch.empty((n, n, n), dtype=torch.uint8, device='cuda')
            dst = torch.empty((n, n, n), dtype=torch.float32, device='cuda')
            ctx.synth_volume_u8(3, (n, n, n), out=src)
            for pname, prec in (('auto', _capi.PREC_AUTO), ('f16', _capi.PREC_F16), ('f32', _capi.PREC_F32)):
                kw = dict(mean=128.0, std=33.0, precision=prec, dst=dst, dims=(n, n, n))
                prog.infer_volume(src, (tile,) * 3, (off,) * 3, **kw)
                ctx.synchronize()
                ctx.timing(True); ctx.timing_reset()
                t0 = time.perf_counter()
                prog.infer_volume(src, (tile,) * 3, (off,) * 3, **kw)
                ctx.synchronize()
                dt = time.perf_counter() - t0
                key = '%s_%s' % (name, pname)
                res[key] = dict(volume=n, tile=tile, executor=ctx.last_path(), seconds=dt,
                                mvox_s=(n - 2 * off) ** 3 / dt / 1e6,
                                kernels={k: round(v['ms'], 2) for k, v in ctx.timing_get().items()})
                ctx.timing(False)
                print(key, json.dumps(res[key]), flush=True)
            del src, dst

    if 'vgg2' in what:
        # vgg_like2 (scripts/fpl_cx1_0_vgg_4ss.py): 5 x conv3, 159 867 FLOP per output voxel
        g = fplmodels.vgg_like2(100)[0]
        synth.synthetic_weights(g, 8)
        prog = _capi.Program(ctx, g, (4, 4, 4))
        n = a.vgg2_size
        src = torch.empty((n, n, n), dtype=torch.uint8, device='cuda')
        dst = torch.empty((n, n, n), dtype=torch.float32, device='cuda')
        ctx.synth_volume_u8(4, (n, n, n), out=src)
        flop = 2 * (27 * 48 + 27 * 48 * 48 + (2 * 27 * 48 * 48) / 8 + (27 * 48 * 48 + 48 * 96 + 96 * 96 + 96) / 64)
        for pname, prec in (('f16s', _capi.PREC_F16S), ('f16', _capi.PREC_F16), ('bf16', _capi.PREC_BF16),
                            ('f32', _capi.PREC_F32)):
            kw = dict(mean=128.0, std=33.0, precision=prec, dst=dst, dims=(n, n, n))
            prog.infer_volume(src, (100,) * 3, (10,) * 3, **kw)
            ctx.synchronize()
            ctx.timing(True); ctx.timing_reset()
            t0 = time.perf_counter()
            prog.infer_volume(src, (100,) * 3, (10,) * 3, **kw)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            vox = (n - 20) ** 3
            res['vgg_like2_' + pname] = dict(
                volume=n, mvox_s=vox / dt / 1e6, seconds=dt, tflops_algorithmic=vox * flop / dt / 1e12,
                kernels={k: round(v['ms'], 2) for k, v in ctx.timing_get().items()})
            ctx.timing(False)
            print(json.dumps(res['vgg_like2_' + pname]), flush=True)

    if 'train' in what:
        g = fplmodels.vgg_like()[0]
        synth.synthetic_weights(g, 8)
        tr = _capi.Trainer(ctx, g)
        rng = np.random.default_rng(0)
        data = rng.standard_normal((32, 64, 64, 64)).astype(np.float32)
        labels = (rng.random((32, 12, 12, 12)) > 0.9).astype(np.uint8)
        # the loop of train.fit_generator (bench.py's leg): the prefetch worker uploads the next
        # batch while the step runs; 20 steps without per-kernel events, then 5 with
        from flypylib_amd import train as fpl_train

        def forever():
            while True:
                yield data, labels
        batches = fpl_train._Prefetch(forever(), stage=fpl_train._DeviceStager(ctx.device))
        tr.step(*next(batches), 0); tr.apply(1.0)
        ctx.synchronize()
        t0 = time.perf_counter()
        steps = 20
        for s in range(steps):
            tr.step(*next(batches), s + 1); tr.apply(1.0)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ctx.timing(True); ctx.timing_reset()
        for s in range(5):
            tr.step(*next(batches), steps + s + 1); tr.apply(1.0)
        ctx.synchronize()
        kern = ctx.timing_get()
        ctx.timing(False)
        batches.close()
        t0 = time.perf_counter()
        for s in range(5):
            tr.step(data, labels, s + 1); tr.apply(1.0)
        ctx.synchronize()
        dt_host = (time.perf_counter() - t0) / 5
        res['vgg_train_b32_64cubed_f32'] = dict(
            seconds_per_step=dt, steps_per_s=1 / dt, steps_per_s_host_arrays=1 / dt_host,
            tflops_algorithmic=492e9 / dt / 1e12,
            note='fit_generator loop (upload of the next batch overlaps the step); '
                 'steps_per_s_host_arrays = the step fed host arrays directly, H2D inside',
            kernels={k: round(v['ms'] / 5, 2) for k, v in kern.items()})
        print(json.dumps(res['vgg_train_b32_64cubed_f32']), flush=True)

    if 'train_unet' in what:
        # unet_like2 as scripts/fpl_cx1_0_unet_4ss_all.py trains it: rf-sized 24^3 patches,
        # batch 64 per GPU, masked focal loss
        g = fplmodels.unet_like2()[0]
        synth.synthetic_weights(g, 3)
        tr = _capi.Trainer(ctx, g, loss='masked_focal_loss')
        rng = np.random.default_rng(0)
        data = rng.standard_normal((64, 24, 24, 24, 1)).astype(np.float32)
        labels = rng.integers(0, 3, (64, 6, 6, 6, 1)).astype(np.uint8)
        tr.step(data, labels, 0); tr.apply(1.0)
        ctx.synchronize()
        # wall clock over 20 steps WITHOUT per-kernel events (this step is ~150 small launches: the events
        # of the table below cost a fifth of it) ...
        t0 = time.perf_counter()
        for s in range(20):
            tr.step(data, labels, s + 1); tr.apply(1.0)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 20
        # ... then the per-kernel table of 5 more
        ctx.timing(True); ctx.timing_reset()
        t0 = time.perf_counter()
        steps = 5
        for s in range(steps):
            tr.step(data, labels, s + 21); tr.apply(1.0)
        ctx.synchronize()
        dt_ev = (time.perf_counter() - t0) / steps
        res['unet_train_b64_24cubed_f32'] = dict(
            seconds_per_step=dt, steps_per_s=1 / dt, seconds_per_step_with_kernel_events=dt_ev,
            note='includes H2D of the batch',
            kernels={k: round(v['ms'] / steps, 3) for k, v in ctx.timing_get().items()})
        ctx.timing(False)
        print(json.dumps(res['unet_train_b64_24cubed_f32']), flush=True)

    if 'v2o' in what or 'pipeline' in what:
        n = a.sub
        g = fplmodels.vgg_like(102)[0]
        synth.synthetic_weights(g, 9)
        prog = _capi.Program(ctx, g, (4, 4, 4))
        src = torch.empty((n, n, n), dtype=torch.uint8, device='cuda')
        pred = torch.empty((n, n, n), dtype=torch.float32, device='cuda')
        ctx.synth_volume_u8(5, (n, n, n), out=src)
        kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_BF16, dst=pred, dims=(n, n, n))
        prog.infer_volume(src, (102,) * 3, (7,) * 3, **kw)
        ctx.synchronize()
        t0 = time.perf_counter()
        prog.infer_volume(src, (102,) * 3, (7,) * 3, **kw)
        ctx.synchronize()
        t_inf = time.perf_counter() - t0
        # blob-like probabilities make the NMS meaningful: add planted blobs
        prob = synth.blob_prob_volume(11, (n, n, n), period=64, radius=9.0)
        prob_dev = torch.from_numpy(prob).cuda()
        fplobjdetect.voxel2obj(prob_dev, 27, 5, (0, 0, 0), 35, 0.1)      # warm-up
        ctx.timing(True); ctx.timing_reset()
        t0 = time.perf_counter()
        out, info = fplobjdetect.voxel2obj(prob_dev, 27, 5, (0, 0, 0), 35, 0.1,
                                           return_info=True)
        t_v2o = time.perf_counter() - t0
        kern = {k: round(v['ms'], 2) for k, v in ctx.timing_get().items()}
        ctx.timing(False)
        padded = (n + 54) ** 3
        res['voxel2obj_sub%d' % n] = dict(
            seconds=t_v2o, mvox_s=n ** 3 / t_v2o / 1e6, detections=len(out['conf']),
            rounds=info['rounds'], gb_s_algorithmic=12 * padded / t_v2o / 1e9,
            kernels=kern)
        res['infer_sub%d_bf16' % n] = dict(seconds=t_inf,
                                           mvox_s=(n - 14) ** 3 / t_inf / 1e6)
        print(json.dumps(res['voxel2obj_sub%d' % n]), flush=True)
        if 'pipeline' in what:
            from oracle import voxel2obj_oracle
            t0 = time.perf_counter()
            ref = voxel2obj_oracle.voxel2obj(prob, 27, 5, (0, 0, 0), 35, 0.1)
            t_cpu = time.perf_counter() - t0
            same = (np.array_equal(ref['locs'], out['locs'])
                    and np.array_equal(ref['conf'], out['conf']))
            res['pipeline_sub%d' % n] = dict(
                infer_s=t_inf, v2o_s=t_v2o, total_mvox_s=n ** 3 / (t_inf + t_v2o) / 1e6,
                detections_identical_to_cpu_oracle=bool(same),
                cpu_oracle_v2o_s=t_cpu, cpu_oracle_mvox_s=n ** 3 / t_cpu / 1e6)
            print(json.dumps(res['pipeline_sub%d' % n]), flush=True)
    if 'roi' in what:
        # configs[4] end to end on one GPU: full_roi_inference over a synthetic
        # volume cut into 512-substacks + 35 buffer (582^3 cubes), bf16, r 27, sigma 5
        import pickle
        import shutil
        import tempfile
        from flypylib_amd import FplNetwork
        from oracle import voxel2obj_oracle
        n = a.roi_size
        # --roi-precision: 'auto' = the package's default (split halves for vgg_like)
        net = FplNetwork(fplmodels.vgg_like, precision=a.roi_precision)
        synth.synthetic_weights(net.train_single, 9)
        net._set_infer()
        wd = tempfile.mkdtemp(prefix='fri_')
        src = 'synth://5,%d,%d,%d' % (n, n, n)
        fplobjdetect.gen_full_tab_roi(wd + '/roi', src, None, step_size=512)
        roi = fplobjdetect.roi_from_txt(wd + '/roi_00.txt')[0]
        norm = [128., 33., 0.5]
        if a.roi_source == 'resident':
            # the same voxels, made once and kept in HBM: inputs resident when the timed region starts
            resident = torch.empty((n, n, n), dtype=torch.uint8, device='cuda')
            ctx.synth_volume_u8(5, (n, n, n), out=resident)
            torch.cuda.synchronize()
            src = resident
        fplobjdetect.full_roi_inference(src, None, roi[:6], net, 0.1, wd + '/warm', norm)
        ctx.timing(True); ctx.timing_reset()
        t0 = time.perf_counter()
        out = fplobjdetect.full_roi_inference(src, None, wd + '/roi_00.txt', net, 0.1,
                                              wd + '/work', norm)
        dt = time.perf_counter() - t0
        # (lane 0's inference kernels only: HIP events on the six concurrent streams of the pipeline time the
        # waits for one another, not execution - the kernel table of ALL lanes is the rocprofv3 kernel trace of
        # this run, profiles/r05_roi1536_kernel_stats.csv)
        kern = {k: round(v['ms'], 2) for k, v in ctx.timing_get().items()}
        ctx.timing(False)
        if a.skip_oracle:
            print(json.dumps(dict(substacks=len(roi), seconds=dt, mvox_s=n ** 3 / dt / 1e6,
                                  detections=int(len(out['conf'])))), flush=True)
            shutil.rmtree(wd, ignore_errors=True)
            return
        # one substack re-derived and post-processed by the CPU oracle
        ss = roi[len(roi) // 2]
        sz = ss.size + 70
        cube = ctx.malloc((sz,) * 3, np.uint8)
        pred = ctx.malloc((sz,) * 3, np.float32)
        ctx.synth_substack_u8(5, (n, n, n), (sz,) * 3, [ss.z - 35, ss.y - 35, ss.x - 35], cube)
        from flypylib_amd import fplpipeline
        st = fplpipeline.normalisation_from_histogram(ctx.histogram_u8(cube), norm)
        net.infer_network.program.infer_volume(cube, net.infer_sz, net.rf_offset, mean=st['mn_use'],
                                               std=norm[1],
                                               precision=fplpipeline.fplobjdetect_precision(net, None),
                                               dst=pred, dims=(sz,) * 3)
        t0 = time.perf_counter()
        ref = voxel2obj_oracle.voxel2obj(pred.to_host(), 27, 5, (ss.x - 35, ss.y - 35, ss.z - 35), 35, 0.1)
        t_cpu = time.perf_counter() - t0
        got = pickle.load(open(fplobjdetect.fri_filename(wd + '/work', ss), 'rb'))
        same = np.array_equal(ref['locs'], got['locs']) and np.array_equal(ref['conf'], got['conf'])
        res['full_roi_inference_%d' % n] = dict(
            precision=a.roi_precision, source=a.roi_source, executor=ctx.last_path(),
            substacks=len(roi), seconds=dt, mvox_s=n ** 3 / dt / 1e6,
            detections=int(len(out['conf'])), checked_substack=list(ss),
            checked_substack_detections=int(len(got['conf'])),
            detections_identical_to_cpu_oracle=bool(same), cpu_oracle_v2o_s=t_cpu,
            kernel_ms_total=kern)
        print(json.dumps(res['full_roi_inference_%d' % n]), flush=True)
        shutil.rmtree(wd, ignore_errors=True)
    if 'c3share' in what:
        # configs[2] at its stated size, one rank's share: unet_like2 over the
        # 1024 x 2048 x 2048 volume is 13 tile rows of pitch 82 along z; 8 ranks take
        # 2,2,2,2,2,1,1,1 of them.  Rank 0's slab (2 rows + halo = 182 z rows) runs as a
        # standalone volume, as bench.py / torchrun ranks do, and is compared bit for bit
        # with the same rows of a run over the WHOLE volume on this one GPU.
        from flypylib_amd import multi_gpu
        Z, Y, X = 1024, 2048, 2048
        tile, off, world = 100, 9, 8
        g = fplmodels.unet_like2(tile)[0]
        synth.synthetic_weights(g, 7)
        prog = _capi.Program(ctx, g, (1, 1, 1))
        n_rows = multi_gpu.n_tile_rows(Z, tile, off)
        parts = multi_gpu.slab_partition(n_rows, world)
        zb, ze = parts[0]
        pitch = tile - 2 * off
        z_hi = min(ze * pitch + 2 * off, Z)
        whole_src = torch.empty((Z, Y, X), dtype=torch.uint8, device='cuda')
        ctx.synth_volume_u8(3, (Z, Y, X), out=whole_src)
        whole_dst = torch.empty((Z, Y, X), dtype=torch.float32, device='cuda')
        kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_F16)
        t0 = time.perf_counter()
        prog.infer_volume(whole_src, (tile,) * 3, (off,) * 3, dst=whole_dst, dims=(Z, Y, X), **kw)
        ctx.synchronize()
        t_whole = time.perf_counter() - t0
        slab_src = whole_src[:z_hi].contiguous()
        slab_dst = torch.empty((z_hi, Y, X), dtype=torch.float32, device='cuda')
        prog.infer_volume(slab_src, (tile,) * 3, (off,) * 3, dst=slab_dst, dims=(z_hi, Y, X), **kw)
        ctx.synchronize()
        t0 = time.perf_counter()
        prog.infer_volume(slab_src, (tile,) * 3, (off,) * 3, dst=slab_dst, dims=(z_hi, Y, X), **kw)
        ctx.synchronize()
        t_slab = time.perf_counter() - t0
        lo, hi = multi_gpu.slab_rows((zb, ze), Z, tile, off)
        same = bool(torch.equal(slab_dst[lo:hi], whole_dst[lo:hi]))
        own = (hi - lo - off) * (Y - 2 * off) * (X - 2 * off)     # valid voxels rank 0 owns
        res['configs2_rank0_share'] = dict(
            volume=[Z, Y, X], tile_rows=n_rows, partition=parts, rank0_rows=[lo, hi],
            slab_dims=[z_hi, Y, X], seconds=t_slab, mvox_s=own / t_slab / 1e6,
            tflops_algorithmic=own * 350720 / t_slab / 1e12,
            whole_volume_seconds_one_gpu_cold=t_whole,
            whole_volume_mvox_s=(Z - 18) * (Y - 18) * (X - 18) / t_whole / 1e6,
            slab_rows_identical_to_whole=same, executor=ctx.last_path())
        print(json.dumps(res['configs2_rank0_share']), flush=True)
        del whole_src, whole_dst, slab_src, slab_dst
    if 'c5share' in what:
        # configs[4] at its stated size, one rank's share: the 4096^3 ROI is 512 substacks
        # of 512^3 (+ 35 buffer: 582^3 cubes); with 8 ranks a rank takes every 8th = 64.
        # Rank 3's share is run (x block 3: interior substacks; rank 0's all touch the x = 0
        # face, where the zero fill outside the volume leaves next to no detections).
        # RANK / WORLD_SIZE come from the environment as under torchrun; there is no process
        # group here, so the merge step warns and returns the rank's own detections.
        import pickle
        import shutil
        import tempfile
        import warnings
        from flypylib_amd import FplNetwork, fplpipeline
        from oracle import voxel2obj_oracle
        n = 4096
        net = FplNetwork(fplmodels.vgg_like, precision='f16')
        synth.synthetic_weights(net.train_single, 9)
        net._set_infer()
        wd = tempfile.mkdtemp(prefix='fri_')
        src = 'synth://5,%d,%d,%d' % (n, n, n)
        fplobjdetect.gen_full_tab_roi(wd + '/roi', src, None, step_size=512)
        roi = fplobjdetect.roi_from_txt(wd + '/roi_00.txt')[0]
        norm = [128., 33., 0.5]
        share_rank = 3
        os.environ['RANK'], os.environ['WORLD_SIZE'] = str(share_rank), '8'
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            fplobjdetect.full_roi_inference(src, None, roi[:8], net, 0.1, wd + '/warm', norm)
            ctx.timing(True); ctx.timing_reset()
            t0 = time.perf_counter()
            out = fplobjdetect.full_roi_inference(src, None, wd + '/roi_00.txt', net, 0.1,
                                                  wd + '/work', norm)
            dt = time.perf_counter() - t0
        del os.environ['RANK'], os.environ['WORLD_SIZE']
        kern = {k: round(v['ms'], 2) for k, v in ctx.timing_get().items()}
        ctx.timing(False)
        mine = roi[share_rank::8]
        done = [ss for ss in roi if os.path.isfile(fplobjdetect.fri_filename(wd + '/work', ss))]
        # the substack of the share with the most detections goes to the CPU oracle
        counts = [len(pickle.load(open(fplobjdetect.fri_filename(wd + '/work', s_), 'rb'))['conf'])
                  for s_ in mine]
        ss = mine[int(np.argmax(counts))]
        sz = ss.size + 70
        cube = ctx.malloc((sz,) * 3, np.uint8)
        pred = ctx.malloc((sz,) * 3, np.float32)
        ctx.synth_substack_u8(5, (n, n, n), (sz,) * 3, [ss.z - 35, ss.y - 35, ss.x - 35], cube)
        st = fplpipeline.normalisation_from_histogram(ctx.histogram_u8(cube), norm)
        net.infer_network.program.infer_volume(cube, net.infer_sz, net.rf_offset, mean=st['mn_use'],
                                               std=norm[1], precision=_capi.PREC_F16, dst=pred,
                                               dims=(sz,) * 3)
        t0 = time.perf_counter()
        ref = voxel2obj_oracle.voxel2obj(pred.to_host(), 27, 5, (ss.x - 35, ss.y - 35, ss.z - 35), 35, 0.1)
        t_cpu = time.perf_counter() - t0
        got = pickle.load(open(fplobjdetect.fri_filename(wd + '/work', ss), 'rb'))
        same = np.array_equal(ref['locs'], got['locs']) and np.array_equal(ref['conf'], got['conf'])
        res['configs4_rank_share'] = dict(
            roi=[n, n, n], rank=share_rank, world=8, substacks_total=len(roi), substacks_rank=len(mine),
            detections_per_substack_max=int(max(counts)),
            substacks_written=len(done), all_p_written=os.path.isfile(wd + '/work/all.p'),
            seconds=dt, mvox_s_rank=len(mine) * 512 ** 3 / dt / 1e6,
            detections_rank=int(len(out['conf'])), checked_substack=list(ss),
            checked_substack_detections=int(len(got['conf'])),
            detections_identical_to_cpu_oracle=bool(same), cpu_oracle_v2o_s=t_cpu,
            kernel_ms_total=kern)
        print(json.dumps(res['configs4_rank_share']), flush=True)
        shutil.rmtree(wd, ignore_errors=True)
    if a.out:
        json.dump(res, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
