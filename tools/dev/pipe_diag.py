"""where does full_roi_inference spend its wall time?  per-call host durations of its stages"""
import sys, time, threading, collections, tempfile
import numpy as np
sys.path.insert(0, '.')
from flypylib_amd import FplNetwork, fplmodels, fplobjdetect, fplpipeline, synth, _capi, runtime

T = collections.defaultdict(list)
lock = threading.Lock()
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            with lock:
                T[label].append((threading.get_ident(), t0, time.perf_counter()))
    setattr(obj, name, g)

prec = sys.argv[1] if len(sys.argv) > 1 else 'auto'
n = 1536
net = FplNetwork(fplmodels.vgg_like, precision=prec)
synth.synthetic_weights(net.train_single, 9)
net._set_infer()
wd = tempfile.mkdtemp(prefix='fri_')
src = 'synth://5,%d,%d,%d' % (n, n, n)
fplobjdetect.gen_full_tab_roi(wd + '/roi', src, None, step_size=512)
roi = fplobjdetect.roi_from_txt(wd + '/roi_00.txt')[0]
norm = [128., 33., 0.5]
fplobjdetect.full_roi_inference(src, None, roi[:8], net, 0.1, wd + '/warm', norm)
wrap(_capi.Context, 'histogram_u8', 'hist')
wrap(_capi.Program, 'infer_volume', 'infer')
wrap(_capi.Program, '__init__', 'program_create')
wrap(_capi.Program, 'close', 'program_close')
wrap(fplpipeline.fplobjdetect, 'voxel2obj', 'v2o')
wrap(fplpipeline, '_write_norm', 'write_norm')
wrap(fplpipeline.pickle, 'dump', 'pickle')
wrap(_capi.Context, 'malloc', 'malloc')
wrap(_capi.Context, 'synth_substack_u8', 'synth')
t0 = time.perf_counter()
fplobjdetect.full_roi_inference(src, None, wd + '/roi_00.txt', net, 0.1, wd + '/work', norm)
dt = time.perf_counter() - t0
print('wall %.1f ms' % (dt * 1e3))
for k, v in T.items():
    d = [b - a for _, a, b in v]
    print('%-16s n %3d  total %7.1f ms  mean %6.2f  max %6.2f   first start %6.1f  last end %6.1f' %
          (k, len(d), sum(d) * 1e3, np.mean(d) * 1e3, max(d) * 1e3, (min(a for _, a, _ in v) - t0) * 1e3,
           (max(b for _, _, b in v) - t0) * 1e3))
