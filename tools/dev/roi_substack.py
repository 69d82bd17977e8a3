"""one interior substack of the 1536^3 synthetic ROI, stage by stage (standalone times):
    PYTHONPATH=. python tools/dev/roi_substack.py [origin]"""
import json
import sys
import time
import numpy as np
from flypylib_amd import FplNetwork, _capi, fplmodels, fplobjdetect, fplpipeline, runtime, synth

org = int(sys.argv[1]) if len(sys.argv) > 1 else 477
ctx = runtime.get_context(0)
net = FplNetwork(fplmodels.vgg_like, precision='bf16')
synth.synthetic_weights(net.train_single, 9)
net._set_infer()
prog = net.infer_network.program
sz = 582
cube = ctx.malloc((sz,) * 3, np.uint8)
pred = ctx.malloc((sz,) * 3, np.float32)
norm = [128., 33., 0.5]
for rep in range(3):
    if rep == 2:
        ctx.timing(True)
        ctx.timing_reset()
    t0 = time.perf_counter()
    ctx.synth_substack_u8(5, (1536,) * 3, (sz,) * 3, [org] * 3, cube)
    st = fplpipeline.normalisation_from_histogram(ctx.histogram_u8(cube), norm)
    t1 = time.perf_counter()
    prog.infer_volume(cube, net.infer_sz, net.rf_offset, mean=st['mn_use'], std=norm[1],
                      precision=_capi.PREC_BF16, dst=pred, dims=(sz,) * 3)
    ctx.synchronize()
    t2 = time.perf_counter()
    out, info = fplobjdetect.voxel2obj(pred, 27, 5, (0, 0, 0), 35, 0.1, return_info=True, _ctx=ctx)
    t3 = time.perf_counter()
kern = {k: round(v['ms'], 3) for k, v in ctx.timing_get().items()}
print(json.dumps(dict(prep_ms=(t1 - t0) * 1e3, infer_ms=(t2 - t1) * 1e3, v2o_ms=(t3 - t2) * 1e3,
                      detections=len(out['conf']), rounds=info['rounds'],
                      thresh=float(info['thresh']), kernel_sum=round(sum(kern.values()), 3),
                      kernels=kern)))
