#!/usr/bin/env python3
"""Find small seeded volumes whose smoothed float32 result differs between separately
rounded products / sums (scipy on x86-64, the oracle) and fused multiply-adds:
    python tools/dev/find_fma_witness.py
Prints (seed, shape, r, sigma, number of differing voxels) rows for the test's table.
The unfused C form is checked against scipy on every volume (it must be identical)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flypylib_amd import fplobjdetect, synth          # noqa: E402
from oracle import voxel2obj_oracle                    # noqa: E402

so = '/tmp/fma_witness.so'
subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-o', so,
                       os.path.join(ROOT, 'tools/dev/fma_witness.c'), '-lm'])
lib = C.CDLL(so)


def smooth(vol, w, wr, fused):
    out = np.empty_like(vol)
    scratch = np.empty_like(vol)
    d = (C.c_int64 * 3)(*vol.shape)
    lib.smooth3(vol.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                scratch.ctypes.data_as(C.c_void_p), d,
                np.ascontiguousarray(w[wr:]).ctypes.data_as(C.c_void_p), wr, int(fused))
    return out


found = {}
for sigma, r, shape in ((5.0, 10, (36, 40, 44)), (3.0, 6, (40, 36, 44)), (2.0, 4, (44, 40, 36)),
                        (1.5, 3, (40, 44, 36))):
    w = fplobjdetect.gaussian_kernel1d(sigma)
    wr = (len(w) - 1) // 2
    seed = 0
    want = int(os.environ.get('WITNESSES', '2'))
    while (min(len(found.setdefault((sigma, m), [])) for m in (1, 2)) < want
           and seed < int(os.environ.get('MAX_SEED', '20000'))):
        seed += 1
        pred = synth.hash_uniform_f32(seed, shape)
        vol = np.pad(pred, r, 'constant')
        a = smooth(vol, w, wr, 0)
        if seed <= 2:
            from scipy import ndimage
            assert np.array_equal(a, ndimage.gaussian_filter(vol, sigma, truncate=2.0))
        for mode in (1, 2):
            if len(found[(sigma, mode)]) >= want:
                continue
            nd = int(np.count_nonzero(a != smooth(vol, w, wr, mode)))
            if nd:
                found[(sigma, mode)].append(seed)
                print((seed, shape, r, sigma), 'fused form', mode, 'differing voxels', nd,
                      flush=True)
