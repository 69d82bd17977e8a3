import numpy as np
from flypylib_amd import _capi, fplmodels, synth
from oracle import cnn_oracle, infer_oracle
ctx=_capi.Context(0)
worst=0
for shape,tile in [((50,47,41),30),((46,46,46),30),((31,30,64),30),((75,33,90),30),((104,120,110),102),((40,135,52),46)]:
    for seed in (21,22,23):
        g=fplmodels.vgg_like(tile)[0]; synth.synthetic_weights(g,seed)
        prog=_capi.Program(ctx,g,(4,4,4))
        u8=synth.em_volume_u8(9,shape)
        img=(u8.astype(np.float32)-np.float32(128))/np.float32(33)
        got=prog.infer_volume(u8,(tile,)*3,(7,)*3,mean=128.0,std=33.0,precision=_capi.PREC_BF16)
        f32=infer_oracle.infer_lattice(img,(tile,)*3,(7,)*3,lambda b: cnn_oracle.vgg_like_forward(b.astype(np.float32),g.weights,4))
        d=np.abs(got-f32); worst=max(worst,d.max()); print(shape,seed,d.max(),d.mean())
print('worst',worst)
