#!/usr/bin/env python3
"""Headline benchmark: vgg_like sliding-window inference Mvoxels/s on a synthetic
uint8 EM volume - BASELINE.json's metric on the volume its metric string names: 520^3 per GPU
(`--size 520`, the default; weak scaling: Z = 520 N).  configs[1]'s 1024^3 volume is timed in the
same run as a leg and reported as `value_1024`.

The headline precision is 'f16s' - split IEEE halves, three MFMAs per product: the fastest
executor whose probabilities are fp32-grade (2 - 4e-6 off fp32 on the trained fixture,
the same detected point set: tests/test_gpu_trained_parity.py), i.e. the one that meets the north
star's gate.  Plain f16 / bf16 (3x the rate, ~1e-3 / ~8e-3 off) and fp32 are legs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of FplNetwork.infer's hot path (normalise -> tile lattice ->
3D-CNN -> upsample -> stitch) over the volume, input uint8 and output float32
both resident in HBM.  With N ranks the volume is N x `size` rows along Z and every
rank takes a contiguous slab of tile rows (weak scaling, no collective on the
data path; RCCL is only used for the barrier / max-over-ranks of the clock).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel from HIP
events recorded on the library's stream during the timed steps; `cpu_baseline`
times the CPU oracle (torch-CPU restatement, NOT Keras) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOP per valid output voxel of vgg_like, by layer:
# conv3 1->48, conv1 48->48 (full res); conv3 48->48, conv1 (1/8); conv3 48->48,
# conv1 48->96, conv1 96->96, conv1 96->1 (1/64)
VGG_FLOP = dict(L1=2592.0, L2=4608.0, L3=15552.0, L4=576.0, L5=1944.0, L6=144.0,
                L7=288.0, L8=3.0)
VGG_FLOP_TOTAL = sum(VGG_FLOP.values())          # 25 707
# which layers each kernel name computes (for roofline.achieved)
KERNEL_LAYERS = {
    'generic_conv3_f32': ('L1', 'L3', 'L5'),
    'generic_conv1_f32': ('L2', 'L4', 'L6', 'L7', 'L8'),
    'mfma_conv3_f32': ('L3', 'L5'), 'mfma_stem_f32': ('L1',),
    'mfma_conv1_f32': ('L2', 'L4', 'L6', 'L7', 'L8'),
    'vgg_stem_pool_bf16': ('L1', 'L2'), 'vgg_stem_pool_f32': ('L1', 'L2'),
    'vgg_mid_pool_bf16': ('L3', 'L4'), 'vgg_mid_pool_f32': ('L3', 'L4'),
    'vgg_head_bf16': ('L5', 'L6', 'L7', 'L8'),
    'vgg_c5_tail_bf16': ('L5', 'L6', 'L7', 'L8'),
    'vgg_stem_pool_f16': ('L1', 'L2'), 'vgg_mid_pool_f16': ('L3', 'L4'),
    'vgg_c5_tail_f16': ('L5', 'L6', 'L7', 'L8'),
    'vgg_head_f32': ('L5', 'L6', 'L7', 'L8'),
    'vggs_stem_pool': ('L1', 'L2'), 'vggs_mid_pool': ('L3', 'L4'),
    'vggs_c5_tail': ('L5', 'L6', 'L7', 'L8'),
}
PEAK_TFLOPS = {'bf16': 2500.0, 'f16': 2500.0, 'f16s': 2500.0, 'f32': 157.3}     # MI355X_MICROARCH.md, dense


def cpu_baseline(seconds_budget=30.0, tiles=36):
    """CPU oracle (oracle/cnn_oracle.py, torch-CPU fp32 conv3d) on reference tiles
    102^3 -> 88^3 of the 520^3 case (configs[0]): one full z row of its 6 x 6 x 6 lattice -
    36 tiles - on all host cores of the box's share (or what fits seconds_budget), then ONE
    thread for two tiles (the figure BASELINE.md quotes a single-process Keras-CPU run
    against).  Runs BEFORE the GPU legs."""
    import torch
    from flypylib_amd import fplmodels, synth
    from oracle import cnn_oracle
    g = fplmodels.vgg_like(102)[0]
    synth.synthetic_weights(g, 1234)
    # the 1-GPU box's CPU share is 16 cores (os.cpu_count() reports the whole host)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    u8 = synth.em_volume_u8(1, (102, 102, 102))
    tile = ((u8.astype(np.float32) - 128.0) / 33.0)[None, ..., None]
    cnn_oracle.vgg_like_forward(tile, g.weights, 4)          # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        cnn_oracle.vgg_like_forward(tile, g.weights, 4)
        n += 1
        dt = time.perf_counter() - t0
        if dt > seconds_budget or n >= tiles:
            break
    torch.set_num_threads(1)
    t1 = time.perf_counter()
    n1 = 0
    while n1 < 2 and (n1 == 0 or time.perf_counter() - t1 < 6.0):
        cnn_oracle.vgg_like_forward(tile, g.weights, 4)
        n1 += 1
    dt1 = time.perf_counter() - t1
    torch.set_num_threads(cores)
    return dict(value=n * 88 ** 3 / dt / 1e6, unit='Mvoxels/s',
                cores=cores, kind='port',
                one_thread_value=n1 * 88 ** 3 / dt1 / 1e6,
                sample='%d reference tiles 102^3->88^3 (the lattice of the 520^3 case: 216 '
                       'such tiles) of the same synthetic volume, torch-CPU fp32 restatement '
                       'of vgg_like (oracle/cnn_oracle.py, NOT Keras), %.1f s on %d threads; '
                       '%d tiles in %.1f s on 1 thread' % (n, dt, cores, n1, dt1))


UNET2_FLOP = 350720.0          # unet_like2, per valid output voxel
TRAIN_C4_FLOP = 492e9          # vgg_like training step, 32 x 64^3 patches (3 x forward)
HBM_PEAK_GBS = 8000.0


def _infer_pass(ctx, prog, torch, size, tile, off, prec, steps, warmup, seed=20250101):
    """`steps` timed passes of fpl_infer_volume over a resident size^3 uint8 volume;
    returns (seconds per step, per-kernel HIP-event timings)"""
    dims = (size,) * 3
    src = torch.empty(dims, dtype=torch.uint8, device='cuda')
    dst = torch.empty(dims, dtype=torch.float32, device='cuda')
    ctx.synth_volume_u8(seed, dims, (0, 0, 0), out=src)
    kw = dict(mean=128.0, std=33.0, precision=prec, dst=dst, dims=dims)
    for _ in range(warmup):
        prog.infer_volume(src, (tile,) * 3, (off,) * 3, **kw)
    ctx.synchronize()
    ctx.timing(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        prog.infer_volume(src, (tile,) * 3, (off,) * 3, **kw)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kern = {k: round(v['ms'] / steps, 4) for k, v in ctx.timing_get().items()}
    ctx.timing(False)
    del src, dst
    return dt, kern, ctx.last_path()


def host_leg(ctx, prog, tile, off, n=520, reps=5):
    """The public API end to end (`FplNetwork.infer`, flypylib/fplnetwork.py:136-189: a host
    uint8 array in, a full-resolution float32 host array out - 4 B per voxel over PCIe, which,
    not the GPU, bounds it): wall time of `Program.infer_volume(host array)`
    at the default precision, the call FplNetwork.infer makes.  Pageable numpy buffers, as a
    caller of the reference would hold; the returned array comes from the binding's recycling
    pool (`_capi.host_empty`: memory fresh from the kernel costs 45 ms of first-touch page faults
    per 520^3 result) and the copy out runs on a helper thread beside the next rows' kernels
    (csrc/infer.hip) - the median of `reps` calls is the steady state of a loop over substacks."""
    from flypylib_amd import _capi, synth
    u8 = synth.em_volume_u8(5, (n, n, n))
    prog.infer_volume(u8, (tile,) * 3, (off,) * 3, mean=128.0, std=33.0, precision=_capi.PREC_AUTO)
    calls = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = prog.infer_volume(u8, (tile,) * 3, (off,) * 3, mean=128.0, std=33.0,
                                precision=_capi.PREC_AUTO)
        calls.append(time.perf_counter() - t0)
    dt = sorted(calls)[reps // 2]
    vox = (n - 2 * off) ** 3
    nbytes = n ** 3 * 5
    return dict(workload='FplNetwork.infer equivalent: host uint8 %d^3 -> host float32 (pageable '
                         'numpy arrays), precision auto; H2D 1 B + D2H 4 B per voxel' % n,
                executor=ctx.last_path(), ms=round(dt * 1e3, 3), mvox_s=round(vox / dt / 1e6, 1),
                bound='pcie', pcie_gbs=round(nbytes / dt / 1e9, 2), out_checksum=float(out[off:-off:37, off:-off:41, off:-off:43].sum()),
                ms_calls=[round(c * 1e3, 3) for c in calls])


def secondary_legs(ctx, torch, prog, tile, off, size, steps):
    """Driver-timed twins of the numbers DESIGN.md quotes beside the headline: the same
    step in the other arithmetic types, the metric string's 520^3 size, and short legs
    for configs[2] (unet_like2), configs[4]'s post-process (voxel2obj on one 582^3
    substack) and configs[3] (one training step).  Each: ms, algorithmic work, the
    fraction of the roofline that bounds it."""
    from flypylib_amd import _capi, fplmodels, fplobjdetect, synth
    legs = {}
    k = max(1, min(steps, 5))

    def vgg_leg(name, n, prec, pname, nsteps):
        dt, kern, path = _infer_pass(ctx, prog, torch, n, tile, off, prec, nsteps, 1)
        vox = (n - 2 * off) ** 3
        tf = vox * VGG_FLOP_TOTAL / dt / 1e12
        legs[name] = dict(workload='vgg_like inference %d^3 uint8, %s' % (n, pname),
                          executor=path, ms=round(dt * 1e3, 3),
                          mvox_s=round(vox / dt / 1e6, 1), bound='mfma',
                          achieved_tflops=round(tf, 2), peak_tflops=PEAK_TFLOPS[pname],
                          frac=round(tf / PEAK_TFLOPS[pname], 4), kernel_ms=kern)

    # configs[1]: the 1024^3 volume in every arithmetic type (f16s = the headline's executor)
    vgg_leg('configs1_f16s', 1024, _capi.PREC_F16S, 'f16s', k)
    vgg_leg('configs1_f16', 1024, _capi.PREC_F16, 'f16', k)
    vgg_leg('configs1_bf16', 1024, _capi.PREC_BF16, 'bf16', k)
    vgg_leg('configs1_f32', 1024, _capi.PREC_F32, 'f32', 1)
    # the metric's own 520^3 volume: the headline's executor again (a twin of `value`) and plain f16
    vgg_leg('configs0_520_f16s', 520, _capi.PREC_F16S, 'f16s', k)
    vgg_leg('configs0_520_f16', 520, _capi.PREC_F16, 'f16', k)
    legs['infer_host_520'] = host_leg(ctx, prog, tile, off)

    # configs[2]: unet_like2 on the reference lattice (tile 100, pitch 82), 510^3 sample:
    # split halves (fp32-grade, what 'auto' runs) and plain f16
    g = fplmodels.unet_like2(100)[0]
    synth.synthetic_weights(g, 7)
    uprog = _capi.Program(ctx, g, (1, 1, 1))
    n = 510
    for pname, prec in (('f16s', _capi.PREC_F16S), ('f16', _capi.PREC_F16)):
        dt, kern, path = _infer_pass(ctx, uprog, torch, n, 100, 9, prec, 2, 1, seed=3)
        vox = (n - 18) ** 3
        tf = vox * UNET2_FLOP / dt / 1e12
        legs['configs2_unet_like2_510_%s' % pname] = dict(
            workload='unet_like2 inference %d^3 uint8 (216 reference tiles 100^3), %s' % (n, pname),
            executor=path, ms=round(dt * 1e3, 3), mvox_s=round(vox / dt / 1e6, 1), bound='mfma',
            achieved_tflops=round(tf, 2), peak_tflops=2500.0, frac=round(tf / 2500.0, 4),
            kernel_ms=kern)
    uprog.close()

    # the other four factories (reference fplmodels.py:73, 174, 410, 470) at the default precision: the
    # graph executor (csrc/gx_exec.h) on split halves, op by op; volumes of 6 / 6 / 5 / 4 tiles per axis
    from flypylib_amd import fplutils
    for name, tiles in (('baseline_model', 6), ('resnet_like', 6), ('unet_like4b', 5), ('unet_like_vol', 4)):
        factory = getattr(fplmodels, name)
        _, rf, infer_sz, _ = factory()
        gt, go = fplutils.to3d(infer_sz)[0], fplutils.to3d(rf[1])[0]
        gg = factory(gt)[0]
        synth.synthetic_weights(gg, 7)
        gprog = _capi.Program(ctx, gg, fplutils.to3d(rf[2]))
        n = tiles * (gt - 2 * go) + 2 * go
        dt, kern, path = _infer_pass(ctx, gprog, torch, n, gt, go, _capi.PREC_AUTO, 2, 1, seed=3)
        legs['other_%s_%d_auto' % (name, n)] = dict(
            workload='%s inference %d^3 uint8 (reference tiles %d^3), precision auto' % (name, n, gt),
            executor=path, ms=round(dt * 1e3, 3), mvox_s=round((n - 2 * go) ** 3 / dt / 1e6, 1), kernel_ms=kern)
        gprog.close()

    # configs[4] post-process: voxel2obj of one 512 + 2 x 35 substack (r 27, sigma 5)
    n = 582
    prob = torch.from_numpy(synth.blob_prob_volume(11, (n, n, n), period=64, radius=9.0)).cuda()
    from flypylib_amd import runtime
    vctx = runtime.get_context(ctx.device)           # the context voxel2obj runs on
    fplobjdetect.voxel2obj(prob, 27, 5, (0, 0, 0), 35, 0.1)       # warm-up
    # wall time per call (no per-kernel events in the stream); the MEDIAN of 9 calls, every call
    # listed in ms_calls: one call in a handful takes tens of ms when the allocations the earlier
    # legs dropped are collected under it (r04: 2.3, 2.2, 60.5, 2.5, 2.3 ms)
    import gc
    gc.collect()
    reps = 9
    calls = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fplobjdetect.voxel2obj(prob, 27, 5, (0, 0, 0), 35, 0.1)
        calls.append(time.perf_counter() - t0)
    dt = sorted(calls)[reps // 2]
    vctx.timing(True)                     # the same calls again for the kernel table (the
    fplobjdetect.voxel2obj(prob, 27, 5, (0, 0, 0), 35, 0.1)       # first one creates the events)
    vctx.timing_reset()
    for _ in range(reps):
        fplobjdetect.voxel2obj(prob, 27, 5, (0, 0, 0), 35, 0.1)
    kern = {kk: round(v['ms'] / reps, 4) for kk, v in vctx.timing_get().items()}
    vctx.timing(False)
    gbs = 12.0 * (n + 54) ** 3 / dt / 1e9
    legs['configs4_voxel2obj_582'] = dict(
        workload='voxel2obj on a 582^3 f32 probability volume resident in HBM, r 27, '
                 'sigma 5, thd 0.1, buffer 35', ms=round(dt * 1e3, 3),
        mvox_s=round(n ** 3 / dt / 1e6, 1), detections=int(len(out['conf'])), bound='hbm',
        achieved_gbs=round(gbs, 1), peak_gbs=HBM_PEAK_GBS, frac=round(gbs / HBM_PEAK_GBS, 4),
        algorithmic_bytes=12 * (n + 54) ** 3, kernel_ms=kern,
        ms_calls=[round(c * 1e3, 3) for c in calls])
    del prob

    # configs[3]: one vgg_like training step, batch 32 of 64^3 patches, fp32
    g = fplmodels.vgg_like()[0]
    synth.synthetic_weights(g, 8)
    tr = _capi.Trainer(ctx, g)
    rng = np.random.default_rng(0)
    data = rng.standard_normal((32, 64, 64, 64)).astype(np.float32)
    labels = (rng.random((32, 12, 12, 12)) > 0.9).astype(np.uint8)
    # the loop of train.fit_generator: batches come from a host generator through the
    # prefetch worker, which uploads batch i+1 on a side stream while step i runs
    from flypylib_amd import train as fpl_train

    def batches_forever():
        while True:
            yield data, labels
    batches = fpl_train._Prefetch(batches_forever(), stage=fpl_train._DeviceStager(ctx.device))
    tr.step(*next(batches), 0)
    tr.apply(1.0)
    ctx.synchronize()
    # wall clock over 20 steps WITHOUT per-kernel events (they cost ~0.4 ms per step) ...
    reps = 20
    t0 = time.perf_counter()
    for i in range(reps):
        tr.step(*next(batches), i + 1)
        tr.apply(1.0)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # ... then the per-kernel breakdown of 5 more
    ctx.timing(True)
    ctx.timing_reset()
    kreps = 5
    for i in range(kreps):
        tr.step(*next(batches), reps + i + 1)
        tr.apply(1.0)
    ctx.synchronize()
    kern = ctx.timing_get()
    ctx.timing(False)
    batches.close()
    t0 = time.perf_counter()
    for i in range(5):                      # the same steps fed host arrays directly
        tr.step(data, labels, i + 1)
        tr.apply(1.0)
    ctx.synchronize()
    dt_host = (time.perf_counter() - t0) / 5
    gpu_ms = sum(v['ms'] for v in kern.values()) / kreps
    reps = kreps                            # divisor of the kernel table below
    tf = TRAIN_C4_FLOP / dt / 1e12
    legs['configs3_train_vgg_b32_64'] = dict(
        workload='vgg_like training step (fwd + bwd + Adam), 32 x 64^3 f32 patches from a host '
                 'generator (fit_generator loop: upload of the next batch overlaps the step), 1 GPU',
        ms=round(dt * 1e3, 3), steps_per_s=round(1 / dt, 2),
        ms_host_batches=round(dt_host * 1e3, 3), kernel_ms_sum=round(gpu_ms, 3),
        kernel_ms={k: round(v['ms'] / reps, 3) for k, v in kern.items()}, bound='mfma',
        achieved_tflops=round(tf, 2), peak_tflops=PEAK_TFLOPS['f32'],
        frac=round(tf / PEAK_TFLOPS['f32'], 4))
    tr.close()

    # unet_like2 as scripts/fpl_cx1_0_unet_4ss_all.py trains it: rf-sized 24^3 patches, batch 64, masked focal loss
    g = fplmodels.unet_like2()[0]
    synth.synthetic_weights(g, 3)
    tr = _capi.Trainer(ctx, g, loss='masked_focal_loss')
    data = rng.standard_normal((64, 24, 24, 24, 1)).astype(np.float32)
    labels = rng.integers(0, 3, (64, 6, 6, 6, 1)).astype(np.uint8)
    tr.step(data, labels, 0)
    tr.apply(1.0)
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        tr.step(data, labels, i + 1)
        tr.apply(1.0)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 20
    legs['other_train_unet_like2_b64_24'] = dict(
        workload='unet_like2 training step (fwd + bwd + Adam), 64 x 24^3 f32 patches (host arrays, H2D inside), '
                 'masked focal loss, 1 GPU', ms=round(dt * 1e3, 3), steps_per_s=round(1 / dt, 2))
    tr.close()
    return legs


def spawn_ranks(n, timeout=None):
    """`python bench.py --gpus N` without a launcher: N children of this script, one per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torchrun would), started before this
    process makes any GPU call.  Rank 0's stdout (the JSON line) is relayed; the exit code is
    non-zero if any rank's is.  The children are POLLED: when one exits non-zero its peers -
    which may sit in a rendezvous or a collective waiting for it until the backend's own
    timeout - are terminated (then killed), and the launcher returns 1; likewise after
    `timeout` seconds overall (FPL_BENCH_TIMEOUT, default 1500)."""
    import socket
    import subprocess
    import threading
    if timeout is None:
        timeout = float(os.environ.get('FPL_BENCH_TIMEOUT', '1500'))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()                     # rank 0's pipe is drained while the children are polled
    t0 = time.monotonic()
    why = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes):
            why = 'rank %d exited with code %d' % next((r, c) for r, c in enumerate(codes) if c not in (None, 0))
        elif time.monotonic() - t0 > timeout:
            why = 'no result after %.0f s' % timeout
        if why:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t1 = time.monotonic()
            while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 10.0:
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            break
        time.sleep(0.05)
    reader.join(10.0)
    codes = [p.returncode for p in procs]
    # exactly ONE line on stdout: the JSON line (libraries - gloo, for one - print their own
    # chatter on the rank's stdout; that goes to stderr here)
    lines = b''.join(chunks).decode(errors='replace').splitlines()
    js = [ln for ln in lines if ln.startswith('{')]
    for ln in lines:
        if not ln.startswith('{'):
            sys.stderr.write(ln + '\n')
    if js and not why:
        sys.stdout.write(js[-1] + '\n')
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if why:
        sys.stderr.write('bench.py: %s; the other ranks were stopped\n' % why)
    if bad:
        sys.stderr.write('bench.py: ranks failed (rank, exit code): %s\n' % bad)
    return 1 if (bad or why) else 0


def ranks_seen(dist, backend):
    """how many ranks the process group's collective really joined: a sum all-reduce of ones
    (on the device under nccl = RCCL over xGMI)"""
    import torch
    t = torch.ones(1, dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


def launcher_selftest(rank, world, backend):
    """FPL_BENCH_SELFTEST=1: the launcher and the process group without any GPU work (the CPU
    test of `--gpus N`); FPL_BENCH_SELFTEST=fail<r>: rank r exits non-zero instead"""
    import torch.distributed as dist
    mode = os.environ['FPL_BENCH_SELFTEST']
    if mode == 'fail%d' % rank:
        sys.exit(3)
    if mode == 'hang':                  # every rank just sits there: the launcher's own timeout
        time.sleep(600)
        return
    if mode.startswith('die') and world > 1:
        # die<r>: every rank joins the process group; rank r then exits non-zero while the
        # others wait for it inside a collective (what a GPU fault on one rank looks like)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
        ranks_seen(dist, 'gloo')
        if mode == 'die%d' % rank:
            os._exit(5)
        import torch
        t = torch.ones(1, dtype=torch.float64)
        dist.all_reduce(t)              # never completes: the dead rank is missing
        time.sleep(600)
        return
    n = 1
    if world > 1 and not mode.startswith('fail'):
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
        n = ranks_seen(dist, 'gloo')
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'selftest': True, 'n_gpus': world, 'n_ranks_seen': n}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--size', type=int, default=520,
                    help='volume edge per GPU (Z is size*gpus); 520 = the volume BASELINE.json\'s '
                         'metric names, 1024 = configs[1]')
    ap.add_argument('--precision', default='f16s', choices=['f16s', 'f16', 'bf16', 'f32'],
                    help='MFMA operands (fp32 accumulation).  f16s (default): split IEEE halves, '
                         'three MFMAs per product - fp32-grade (2 - 4e-6 off fp32, detections '
                         'identical), the path FplNetwork.infer takes by default; f16: plain '
                         'IEEE half, 3x the rate, worst voxel 0.7 - 0.8e-3 off fp32 on trained '
                         'weights; bf16: the operand type BASELINE.json configs[1] names, same '
                         'kernels as f16, only 8 significant bits (up to 8e-3 off); f32: exact '
                         'reference arithmetic on fp32 MFMAs')
    ap.add_argument('--tile', type=int, default=102,
                    help='reference infer_sz (tile lattice pitch = tile-14)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true',
                    help='print the cpu_baseline object and exit (what the main run spawns)')
    ap.add_argument('--no-legs', action='store_true',
                    help='skip the secondary legs (other precisions / sizes / configs)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='process-group backend (gloo only to rehearse N>1 on one GPU)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process only launches - one fresh child per
        # GPU, before anything here touches HIP - and relays rank 0's line
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'WORLD_SIZE %d != --gpus %d' % (world, args.gpus)
    if os.environ.get('FPL_BENCH_SELFTEST'):
        return launcher_selftest(rank, world, args.backend)

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return
    # the CPU leg first, in a process of its own (before this one touches the GPU): its
    # 16-thread pool and buffers are gone when the GPU legs run back to back to the end of
    # this process (left in-process, the pool cost the host-bound legs ~0.4 ms per call)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        import subprocess
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only'],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            cpu = json.loads(r.stdout.strip().splitlines()[-1])
        except (IndexError, ValueError):
            sys.stderr.write('cpu_baseline child failed (rc %d), running in-process:\n%s\n'
                             % (r.returncode, r.stderr[-2000:]))
            cpu = cpu_baseline()

    import torch
    import torch.distributed as dist
    # one rank per GPU; ranks beyond the visible devices wrap around (rehearsals of
    # the N>1 path on a one-GPU box with --backend gloo)
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo')

    from flypylib_amd import _capi, fplmodels, multi_gpu, synth
    ctx = _capi.Context(local_rank)
    info = ctx.device_info()

    tile, off = args.tile, 7
    graph = fplmodels.vgg_like(tile)[0]
    synth.synthetic_weights(graph, 1234)
    prog = _capi.Program(ctx, graph, (4, 4, 4))
    prec = {'bf16': _capi.PREC_BF16, 'f16': _capi.PREC_F16, 'f16s': _capi.PREC_F16S,
            'f32': _capi.PREC_F32}[args.precision]

    # global volume (size*N, size, size); rank slab = contiguous tile rows, which
    # is itself a standalone volume whose lattice coincides with the global one
    Z, Y, X = args.size * world, args.size, args.size
    pitch = tile - 2 * off
    n_rows = multi_gpu.n_tile_rows(Z, tile, off)
    zb, ze = multi_gpu.slab_partition(n_rows, world)[rank]
    z_lo = zb * pitch
    z_hi = min(ze * pitch + 2 * off, Z)
    dims = (z_hi - z_lo, Y, X)
    src = torch.empty(dims, dtype=torch.uint8, device='cuda')
    dst = torch.empty(dims, dtype=torch.float32, device='cuda')
    ctx.synth_volume_u8(20250101, dims, (z_lo, 0, 0), out=src)
    valid_global = (Z - 2 * off) * (Y - 2 * off) * (X - 2 * off)
    # valid outputs this rank produces (rows of the global valid region it owns)
    own_rows = min(ze * pitch + off, Z - off) - (zb * pitch + off)
    valid_local = own_rows * (Y - 2 * off) * (X - 2 * off)

    def step():
        prog.infer_volume(src, (tile,) * 3, (off,) * 3, mean=128.0, std=33.0,
                          precision=prec, dst=dst, dims=dims)

    def barrier():
        torch.cuda.synchronize()
        ctx.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timing(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    timings = ctx.timing_get()
    ctx.timing(False)
    executor = ctx.last_path()
    n_seen = 1
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64,
                         device='cuda' if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        n_seen = ranks_seen(dist, args.backend)
    del src, dst

    # HBM bytes per launch of each kernel: NOT measured in this run - read from the
    # latest committed rocprofv3 --pmc pass of the same command (tools/profile_pmc.py;
    # FETCH_SIZE doubled per the gfx950 correction), valid only for the workload size and
    # operand type it was collected on; `traffic_source` names the file
    traffic_by_kernel = {}
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_%d_%s.json'
                                          % (args.size, args.precision))))
    pmc_path = cands[-1] if cands else ''       # the latest round's pass
    if world == 1 and pmc_path:
        for k, v in json.load(open(pmc_path))['kernels'].items():
            if 'hbm_bytes_per_dispatch' in v:
                # rocprofv3 names template instances ("vgg_mid_pool_f16<false>"); the
                # library's timing names do not carry the arguments
                traffic_by_kernel[k.split('<')[0]] = v['hbm_bytes_per_dispatch']

    # dominant kernel roofline (this rank; ranks are symmetric)
    roof = None
    if timings:
        name = max(timings, key=lambda k: timings[k]['ms'])
        tk = timings[name]
        layers = KERNEL_LAYERS.get(name)
        flop_per_vox = sum(VGG_FLOP[l] for l in layers) if layers else None
        avg_ms = tk['ms'] / tk['launches']
        if flop_per_vox is not None:
            flops_per_launch = flop_per_vox * valid_local * args.steps / tk['launches']
            achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
            peak = PEAK_TFLOPS['f32' if ('generic' in name or name.endswith('f32')) else args.precision]
            roof = dict(bound='mfma', kernel=name, achieved=round(achieved, 3),
                        peak=peak, unit='TFLOP/s',
                        frac=round(achieved / peak, 5),
                        traffic=traffic_by_kernel.get(name),
                        traffic_source=(os.path.relpath(pmc_path, ROOT)
                                        if name in traffic_by_kernel else None),
                        avg_launch_ms=round(avg_ms, 4), launches=tk['launches'],
                        algorithmic_flop_per_voxel=flop_per_vox,
                        kernel_ms_total={k: round(v['ms'], 3)
                                         for k, v in timings.items()})
            # the whole step against the same peak
            tf_step = VGG_FLOP_TOTAL * valid_local * args.steps / dt / 1e12
            roof['whole_step'] = dict(achieved=round(tf_step, 2), frac=round(tf_step / peak, 4))
            if args.precision == 'f16s':
                # `achieved` counts the ALGORITHMIC flops of the convolutions; the split
                # path issues three half-precision MFMAs per product (plus K padding), so
                # the matrix pipe's own utilisation is ~3.4x `frac` for this kernel
                # (profiles/r0*_pmc_hbm_1024_f16s.json: SQ_INSTS_MFMA x 16 384 flop per launch)
                roof['mfma_products_per_multiply'] = 3

    legs = None
    if world == 1 and not args.no_legs:
        legs = secondary_legs(ctx, torch, prog, tile, off, args.size, args.steps)

    if rank == 0:
        value = valid_global * args.steps / dt / 1e6
        leg1024 = (legs or {}).get('configs1_%s' % args.precision)
        host520 = (legs or {}).get('infer_host_520')
        line = {
            'metric': 'inference Mvoxels/sec, vgg_like (rf 18, 22^3 coarse -> 88^3 '
                      'per reference tile), synthetic EM uint8 volume',
            'value': round(value, 2), 'unit': 'Mvoxels/s', 'n_gpus': world,
            'n_ranks_seen': n_seen,       # ranks that joined the max-over-ranks collective
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3),
            # configs[1]'s 1024^3 volume on the same executor, timed in this run
            # (legs.configs1_<dtype>), and the public host -> host API on the 520^3 volume
            # (PCIe-bound: legs.infer_host_520) - neither is `value`
            'value_1024': leg1024['mvox_s'] if leg1024 else None,
            'ms_per_step_1024': leg1024['ms'] if leg1024 else None,
            'value_host_to_host_520': host520['mvox_s'] if host520 else None,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': '%s: vgg_like inference, %dx%dx%d '
                                   'synthetic uint8 volume, reference tile '
                                   'lattice %d^3 (pitch %d), u8 in / f32 out '
                                   'resident in HBM' % (
                                       {520: 'the 520^3 volume of BASELINE.json\'s metric (configs[0]\'s '
                                             'substack size) per GPU',
                                        1024: 'configs[1]'}.get(args.size, 'custom size'),
                                       Z, Y, X, tile, pitch),
                       'volume': [Z, Y, X], 'tile_in': tile, 'executor': executor,
                       'operands': {'f16': 'IEEE half MFMA operands, fp32 accumulate: worst '
                                           'voxel 1.5e-4 off fp32 on these weights, 0.7 - 0.8e-3 '
                                           'on the trained fixture; detections may differ from '
                                           'fp32\'s in a tie-break',
                                    'bf16': 'bfloat16 MFMA operands, fp32 accumulate (as '
                                            'configs[1] names; up to 8e-3 off fp32); '
                                            '--precision f16 stays within ~1e-3',
                                    'f16s': 'split IEEE halves (hi + lo per operand, three MFMAs '
                                            'per product, fp32 accumulate): within 4e-6 of fp32 on '
                                            'the trained fixture, the same detected point set as the fp32 '
                                            'path\'s (tests/test_gpu_trained_parity.py); '
                                            'legs.configs1_f16 / _bf16 are the same step on plain '
                                            '16-bit operands (3x the rate, ~1e-3 / ~8e-3 off), '
                                            'legs.configs1_f32 on the reference\'s own fp32',
                                    'f32': 'fp32 MFMA, exact reference arithmetic'}[args.precision],
                       'parallelism': 'z-slab tile sharding x%d, no collective'
                                      % world,
                       'device': info['name']},
            'roofline': roof,
        }
        if cpu is not None:
            line['cpu_baseline'] = cpu
        if legs is not None:
            line['legs'] = legs
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
