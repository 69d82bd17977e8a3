"""dev rehearsal: full_roi_inference under torch.distributed (gloo) with N ranks on one
GPU must give the single-process result"""
import os
import pickle
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.distributed as dist
from flypylib_amd import FplNetwork, fplmodels, fplobjdetect, synth

world = int(os.environ.get('WORLD_SIZE', '1'))
if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo')
os.environ['LOCAL_RANK'] = '0'                      # every rank on the one GPU
net = FplNetwork(fplmodels.vgg_like, precision='f16')
synth.synthetic_weights(net.train_single, 9)
net._set_infer()
wd = sys.argv[1]
src = 'synth://5,768,768,768'
roi = [(256, z, y, x) for z in (0, 256, 512) for y in (0, 256, 512) for x in (0, 256, 512)]
out = fplobjdetect.full_roi_inference(src, None, roi, net, 0.1, wd, [128., 33., 0.5])
rank = dist.get_rank() if world > 1 else 0
if rank == 0:
    pickle.dump(out, open(wd + '/result.p', 'wb'))
    print('rank 0 of %d: %d detections' % (world, len(out['conf'])))
if world > 1:
    dist.destroy_process_group()
