"""where the host time of a voxel2obj call goes (cProfile over 100 calls on a 582^3 volume)"""
import cProfile
import pstats
import torch
from flypylib_amd import fplobjdetect, runtime, synth
ctx = runtime.get_context(0)
n = 582
prob = torch.from_numpy(synth.blob_prob_volume(11, (n, n, n), period=64, radius=9.0)).cuda()
for _ in range(3):
    fplobjdetect.voxel2obj(prob, 27, 5, (0, 0, 0), 35, 0.1)
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    fplobjdetect.voxel2obj(prob, 27, 5, (0, 0, 0), 35, 0.1)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
