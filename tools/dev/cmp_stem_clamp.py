import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from flypylib_amd import FplNetwork, fplmodels, synth
for seed in (21, 5):
    net = FplNetwork(fplmodels.vgg_like)
    net.train_network.summary = lambda *a, **k: None
    net.infer_sz = (102,) * 3
    synth.synthetic_weights(net.train_single, seed)
    net._set_infer()
    u8 = synth.em_volume_u8(8 + seed, (190, 190, 190))
    ref = net.infer(u8, normalize=(128., 33.), precision='f32')
    a = net.infer(u8, normalize=(128., 33.), precision='f16')
    os.environ['FPL_STEM_NOCLAMP'] = '1'
    b = net.infer(u8, normalize=(128., 33.), precision='f16')
    del os.environ['FPL_STEM_NOCLAMP']
    for nm, x in (('clamp', a), ('plain', b)):
        d = np.abs(x - ref)
        print(seed, nm, 'max %.3e mean %.3e p99.9 %.3e' % (d.max(), d.mean(), np.quantile(d, 0.999)))
    print(seed, 'clamp vs plain max %.3e' % np.abs(a - b).max())
