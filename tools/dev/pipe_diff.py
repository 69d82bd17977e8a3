import sys, tempfile, os
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from flypylib_amd import FplNetwork, fplmodels, fplobjdetect, synth
net = FplNetwork(fplmodels.vgg_like)
synth.synthetic_weights(net.train_single, 21)
net.infer_sz = (38, 38, 38)
net._set_infer()
vol = synth.em_volume_u8(5, (70, 90, 80))
roi = [(32, z, y, x) for z in (0, 32, 64) for y in (0, 32, 64) for x in (0, 32, 64)]
kw = dict(obj_min_dist=5, smoothing_sigma=1.5, buffer_sz=10)
norm = [128., 33., 0.7]
d = tempfile.mkdtemp()
a = fplobjdetect.full_roi_inference(vol, None, roi, net, 0.2, d + '/a', norm, precision='f32', **kw)
b = fplobjdetect.full_roi_inference(vol, None, roi, net, 0.2, d + '/b', norm, **kw)
print(len(a['conf']), len(b['conf']))
sa = {tuple(l): c for l, c in zip(a['locs'], a['conf'])}
sb = {tuple(l): c for l, c in zip(b['locs'], b['conf'])}
print('only f32:', [(k, sa[k]) for k in sorted(set(sa) - set(sb))][:10])
print('only auto:', [(k, sb[k]) for k in sorted(set(sb) - set(sa))][:10])
common = [k for k in sa if k in sb]
print('max conf diff on common', max(abs(sa[k] - sb[k]) for k in common))
# probabilities on one interior and one face substack, with the pipeline's own normalisation
for mean, std in ((128.0, 33.0), (121.73, 30.41)):
    p32 = net.infer(vol, normalize=(mean, std), precision='f32')
    ps = net.infer(vol, normalize=(mean, std), precision='f16s')
    dd = np.abs(p32 - ps)
    print(mean, std, 'prob max diff %.3e at %s; interior max %.3e' % (dd.max(), np.unravel_index(dd.argmax(), dd.shape), dd[7:40, 7:40, 7:40].max()))
