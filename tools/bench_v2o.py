#!/usr/bin/env python3
"""voxel2obj on one substack-sized probability volume, repeated (run on the GPU box):

    python tools/bench_v2o.py [--sub 582] [--reps 5] [--out file.json]

The command rocprofv3 profiles for the v2o_* kernels (kernel trace and --pmc passes,
tools/profile_pmc.py --script tools/bench_v2o.py); prints one JSON line with the wall
time per call and the per-kernel HIP-event times.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sub', type=int, default=582)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    import torch
    from flypylib_amd import fplobjdetect, runtime, synth
    ctx = runtime.get_context(0)
    n = a.sub
    prob = torch.from_numpy(synth.blob_prob_volume(11, (n, n, n), period=64, radius=9.0)).cuda()
    fplobjdetect.voxel2obj(prob, 27, 5, (0, 0, 0), 35, 0.1)
    # wall time without the per-kernel HIP events, then the same calls with them
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out, info = fplobjdetect.voxel2obj(prob, 27, 5, (0, 0, 0), 35, 0.1, return_info=True)
    dt = (time.perf_counter() - t0) / a.reps
    ctx.timing(True)
    fplobjdetect.voxel2obj(prob, 27, 5, (0, 0, 0), 35, 0.1)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fplobjdetect.voxel2obj(prob, 27, 5, (0, 0, 0), 35, 0.1)
    dt_timed = (time.perf_counter() - t0) / a.reps
    kern = {k: round(v['ms'] / a.reps, 4) for k, v in ctx.timing_get().items()}
    ctx.timing(False)
    padded = (n + 54) ** 3
    res = dict(sub=n, ms=round(dt * 1e3, 3), ms_with_kernel_events=round(dt_timed * 1e3, 3), kernel_ms_sum=round(sum(kern.values()), 3),
               detections=len(out['conf']), rounds=info['rounds'], thresh=float(info['thresh']),
               algorithmic_bytes=12 * padded, gb_s_algorithmic=round(12 * padded / dt / 1e9, 1),
               kernels=kern)
    print(json.dumps(res), flush=True)
    if a.out:
        json.dump(res, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
