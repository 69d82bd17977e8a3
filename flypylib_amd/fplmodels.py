"""Network architectures for T-bar detection, as layer programs.

Same contract as the reference's factories (`flypylib/fplmodels.py:73-526`):

    model(in_sz=None) -> (net, (rf_size, rf_offset, rf_stride), infer_sz, compile_args)

`net` is a `program.LayerGraph` (Keras-Model-shaped: `get_weights/set_weights`,
`summary`, `compile`, `input_shape`) instead of a Keras graph; the HIP engine
executes its lowered op list.  `vgg_like` and `unet_like2` are the two
architectures of the BASELINE configs; the others use the same layer kinds.

Losses/metrics are referred to by name; their arithmetic lives in the training
engine (binary_crossentropy) or is a SURVEY section 8f follow-on (masked/focal).
"""
import math

from . import fplutils
from .program import LayerGraph

# names of the reference's custom losses / metrics (fplmodels.py:28-65)
masked_weighted_binary_crossentropy = 'masked_weighted_binary_crossentropy'
masked_binary_crossentropy = 'masked_binary_crossentropy'
masked_focal_loss = 'masked_focal_loss'
lb0l1err, lb1l1err, masked_accuracy = 'lb0l1err', 'lb1l1err', 'masked_accuracy'

_UNET_METRICS = [masked_accuracy, lb0l1err, lb1l1err]


def _head(g, x, use_bias):
    return g.conv(x, 1, 1, use_bias=use_bias, activation='sigmoid')


def _dense_tail(g, x, widths=(64,)):
    """1x1x1 'fully connected' stack with Dropout(0.5), then the biased sigmoid
    head shared by baseline/vgg-style models"""
    for w in widths:
        x = g.dropout(g.conv_bn_relu(x, w, 1), 0.5)
    return _head(g, x, use_bias=True)


def baseline_model(in_sz=None):
    """three conv3+pool stages, rf 18 (reference fplmodels.py:73-100)"""
    g = LayerGraph(in_sz)
    x = g.pool(g.conv_bn_relu(g.input(), 32, 3))
    x = g.pool(g.conv_bn_relu(x, 32, 3))
    x = g.conv_bn_relu(x, 32, 3)
    return g.finish(_dense_tail(g, x, (64,))), (18, 7, 4), 102, None


def vgg_like(in_sz=None):
    """VGG-style stack: [conv3, conv1, pool] x2, conv3, two 1x1 'dense' layers,
    biased sigmoid head (reference fplmodels.py:102-136).
    Receptive field 18, first output centred at offset 7, output stride 4."""
    g = LayerGraph(in_sz)
    x = g.input()
    for _ in range(2):
        x = g.conv_bn_relu(x, 48, 3)
        x = g.conv_bn_relu(x, 48, 1)
        x = g.pool(x)
    x = g.conv_bn_relu(x, 48, 3)
    return g.finish(_dense_tail(g, x, (96, 96))), (18, 7, 4), 102, None


def vgg_like2(in_sz=None):
    """vgg_like with 3x3x3 second convs (reference fplmodels.py:138-172)"""
    g = LayerGraph(in_sz)
    x = g.input()
    for _ in range(2):
        x = g.conv_bn_relu(x, 48, 3)
        x = g.conv_bn_relu(x, 48, 3)
        x = g.pool(x)
    x = g.conv_bn_relu(x, 48, 3)
    return g.finish(_dense_tail(g, x, (96, 96))), (24, 10, 4), 100, None


def resnet_like(in_sz=None):
    """two residual stages with cropped shortcuts (reference fplmodels.py:174-208)"""
    g = LayerGraph(in_sz)
    pool1 = g.pool(g.conv_bn_relu(g.input(), 32, 3))

    y = g.conv_bn_relu(pool1, 32, 3)
    y = g.bn(g.conv(y, 32, 1))
    y = g.relu(g.add(g.crop(pool1, 1), y))
    pool2 = g.pool(y)

    z = g.conv_bn_relu(pool2, 64, 3)
    z = g.conv(z, 64, 1)
    shortcut = g.crop(g.conv(pool2, 64, 1), 1)
    z = g.relu(g.add(shortcut, g.bn(z)))
    return g.finish(_head(g, z, use_bias=True)), (18, 7, 4), 102, None


def _unet(in_sz, first_k2, stage2, bottom, crop2, crop1, rf_info, infer_sz, loss):
    """shared U-Net skeleton of unet_like .. unet_like4b

    first_k2  kernel of the second conv in stage 1 (1 or 3)
    stage2    list of (filters, k) for stage 2 after pool1
    bottom    list of (filters, k) at the lowest resolution
    crop2     crop applied to the stage-2 skip (0 = none)
    crop1     crop applied to the stage-1 skip
    """
    g = LayerGraph(in_sz)
    c1 = g.conv_bn_relu(g.input(), 32, 3)
    c1 = g.conv_bn_relu(c1, 32, first_k2)
    x = g.pool(c1)
    for f, k in stage2:
        x = g.conv_bn_relu(x, f, k)
    c2 = x
    x = g.pool(c2)
    for f, k in bottom:
        x = g.conv_bn_relu(x, f, k)
    skip2 = g.crop(c2, crop2) if crop2 else c2
    x = g.concat(g.up(x, 2), skip2)            # [UpSampling(x), skip] order
    x = g.conv_bn_relu(x, 64, 3)
    x = g.conv_bn_relu(x, 64, 1)
    x = g.concat(g.up(x, 2), g.crop(c1, crop1))
    x = g.conv_bn_relu(x, 32, 3)
    x = g.conv_bn_relu(x, 32, 1)
    out = _head(g, x, use_bias=False)
    compile_args = {'loss': loss, 'optimizer': 'adam',
                    'metrics': list(_UNET_METRICS)}
    return g.finish(out), rf_info, infer_sz, compile_args


def unet_like(in_sz=18):
    """reference fplmodels.py:210-256"""
    return _unet(in_sz, 1, [(64, 3), (64, 1)], [(128, 1)], 0, 4,
                 (18, 6, 1), 102, masked_binary_crossentropy)


def unet_like2(in_sz=24):
    """U-Net, rf 24, offset 9, stride 1; focal loss (reference fplmodels.py:258-304)"""
    return _unet(in_sz, 3, [(64, 3), (64, 3)], [(128, 1)], 0, 6,
                 (24, 9, 1), 100, masked_focal_loss)


def unet_like3(in_sz=32):
    """reference fplmodels.py:306-355"""
    return _unet(in_sz, 3, [(64, 3), (64, 3)], [(128, 3), (128, 1)], 2, 10,
                 (32, 13, 1), 100, masked_focal_loss)


def unet_like4(in_sz=40):
    """reference fplmodels.py:358-407"""
    return _unet(in_sz, 3, [(64, 3), (64, 3)], [(128, 3), (128, 3)], 4, 14,
                 (40, 17, 1), 100, masked_focal_loss)


def unet_like4b(in_sz=40):
    """bottlenecked unet_like4 (reference fplmodels.py:410-467)"""
    return _unet(in_sz, 3, [(64, 3), (32, 1), (64, 3)],
                 [(48, 1), (128, 3), (48, 1), (128, 3), (48, 1)], 4, 14,
                 (40, 17, 1), 100, masked_focal_loss)


def unet_like_vol(in_sz=62):
    """BN-free U-Net with ReLU convs (reference fplmodels.py:470-526)"""
    g = LayerGraph(in_sz)

    def cr(x, f, k):
        return g.conv(x, f, k, activation='relu')

    c1 = cr(cr(g.input(), 16, 3), 16, 1)
    c2 = cr(cr(g.pool(c1), 32, 3), 32, 1)
    c3 = cr(g.pool(c2), 64, 1)
    x = g.concat(g.up(c3, 2), c2)
    c4 = cr(cr(x, 64, 3), 64, 1)
    sz = fplutils.to3d(in_sz)
    crops = []
    for ss in sz:
        conv2_sz = math.floor((ss - 2) / 2) - 2
        conv3_sz = math.floor(conv2_sz / 2) * 2
        d = (ss - 2) - (conv3_sz - 2) * 2
        crops.append((math.floor(d / 2), math.ceil(d / 2)))
    x = g.concat(g.up(c4, 2), g.crop(c1, crops))
    x = cr(cr(x, 32, 3), 32, 1)
    out = _head(g, x, use_bias=False)
    compile_args = {'loss': masked_weighted_binary_crossentropy,
                    'optimizer': 'adam', 'metrics': [masked_accuracy]}
    return g.finish(out), (62, 6, 1), 102, compile_args
