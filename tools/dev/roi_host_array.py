"""full_roi_inference over a HOST array (the caller's ndarray / np.memmap): uploaded once and cut on the GPU
(default) against cut on the host and uploaded per substack (FPL_PIPE_RESIDENT_GB=0).  1024^3, 8 substacks."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, '.')
from flypylib_amd import FplNetwork, fplmodels, fplobjdetect, runtime, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = runtime.get_context(0)
net = FplNetwork(fplmodels.vgg_like)
synth.synthetic_weights(net.train_single, 9)
net._set_infer()
dev = ctx.malloc((n, n, n), np.uint8)
ctx.synth_volume_u8(5, (n, n, n), out=dev)
vol = dev.to_host()
dev.free()
wd = tempfile.mkdtemp(prefix='fri_')
fplobjdetect.gen_full_tab_roi(wd + '/roi', 'synth://5,%d,%d,%d' % (n, n, n), None, step_size=512)
roi = wd + '/roi_00.txt'
norm = [128., 33., 0.5]
res = {}
for tag, cap in (('uploaded once', None), ('cut on the host', '0'), ('uploaded once', None), ('cut on the host', '0')):
    if cap is None:
        os.environ.pop('FPL_PIPE_RESIDENT_GB', None)
    else:
        os.environ['FPL_PIPE_RESIDENT_GB'] = cap
    work = tempfile.mkdtemp(prefix='w_', dir=wd)
    t0 = time.perf_counter()
    out = fplobjdetect.full_roi_inference(vol, None, roi, net, 0.1, work, norm)
    dt = time.perf_counter() - t0
    res.setdefault(tag, []).append((dt, len(out['conf'])))
    print('%-16s %.3f s  %d detections' % (tag, dt, len(out['conf'])), flush=True)
assert len({r[1] for v in res.values() for r in v}) == 1
shutil.rmtree(wd, ignore_errors=True)
