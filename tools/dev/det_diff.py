"""how do the detections of the f16s and fp32 predictions differ on the 582^3 trained substack"""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from trained_fixture import trained_network, blob_region_u8
from flypylib_amd import fplobjdetect
net = trained_network('vgg_like', tile=102)
u8, _, locs = blob_region_u8(5, 582, step=48)
norm = (128.0, 33.0)
p32 = net.infer(u8, normalize=norm, precision='f32')
ps = net.infer(u8, normalize=norm, precision='f16s')
d = np.abs(ps - p32)
print('max %.3e mean %.3e' % (d.max(), d.mean()))
kw = dict(obj_min_dist=27, smoothing_sigma=5, buffer_sz=35, thd=0.1)
a = fplobjdetect.voxel2obj(p32, **kw); b = fplobjdetect.voxel2obj(ps, **kw)
print(len(a['conf']), len(b['conf']))
n = min(len(a['conf']), len(b['conf']))
bad = np.where((a['locs'][:n] != b['locs'][:n]).any(axis=1))[0]
print('rows that differ:', len(bad), bad[:20])
sa = set(map(tuple, a['locs'])); sb = set(map(tuple, b['locs']))
print('only in a', sorted(sa - sb)[:10], 'only in b', sorted(sb - sa)[:10])
for i in bad[:6]:
    print(i, a['locs'][i], a['conf'][i], b['locs'][i], b['conf'][i])
