"""GPU parity of the graph executor (csrc/gx_exec.h) on the four factories that are neither the VGG
lattice nor one of the U-Net skeletons: baseline_model, resnet_like, unet_like4b, unet_like_vol
(reference flypylib/fplmodels.py:73-100, 174-208, 410-467, 470-526).  They run op by op on the
descriptor-driven conv3 / conv1 kernels of csrc/conv_mfma.hip - channel counts padded to multiples of
32, UpSampling3D / Cropping3D / concatenate resolved into source descriptors, resnet's Add as its own
kernel, the sigmoid head in fp32 - in all three operand builds.  Reference: the fp32 CPU oracle over the
reference's tile lattice; split halves ('f16s', what 'auto' picks) are held to 1e-5, plain f16 to the
north star's 1e-3."""
import numpy as np
import pytest

from flypylib_amd import _capi, fplmodels, fplutils, synth
from flypylib_amd.program import LayerGraph
from oracle import cnn_oracle, infer_oracle

pytestmark = pytest.mark.gpu
TOL = {'bf16': 2e-2, 'f16': 1e-3, 'f16s': 1e-5}
PREC = {'bf16': _capi.PREC_BF16, 'f16': _capi.PREC_F16, 'f16s': _capi.PREC_F16S, 'auto': _capi.PREC_AUTO}

# (factory, tile, volume): a small tile with ragged lattice edges, and the factory's own infer_sz
CASES = [('baseline_model', 38, (60, 41, 75)), ('baseline_model', 102, (110, 102, 130)),
         ('resnet_like', 38, (50, 64, 41)), ('resnet_like', 102, (102, 120, 111)),
         ('unet_like4b', 52, (70, 52, 90)), ('unet_like4b', 100, (100, 130, 104)),
         ('unet_like_vol', 30, (50, 41, 64)), ('unet_like_vol', 102, (110, 102, 130))]


def _setup(ctx, name, tile, seed=41):
    factory = getattr(fplmodels, name)
    _, rf, _, _ = factory()
    off = fplutils.to3d(rf[1])[0]
    stride = fplutils.to3d(rf[2])
    g = factory(tile)[0]
    synth.synthetic_weights(g, seed)
    return g, off, stride, _capi.Program(ctx, g, stride)


def _oracle(g, img, tile, off, stride):
    def f32(batch):
        y = cnn_oracle.graph_forward(g, batch.astype(np.float32))
        for ax in (1, 2, 3):
            if stride[ax - 1] != 1:
                y = np.repeat(y, stride[ax - 1], axis=ax)
        return y
    return infer_oracle.infer_lattice(img, (tile,) * 3, (off,) * 3, f32)


def _check(ctx, name, tile, shape, kind):
    g, off, stride, prog = _setup(ctx, name, tile)
    u8 = synth.em_volume_u8(13, shape)
    img = (u8.astype(np.float32) - np.float32(128)) / np.float32(33)
    ref = _oracle(g, img, tile, off, stride)
    ctx.timing(True)
    ctx.timing_reset()
    got = prog.infer_volume(u8, (tile,) * 3, (off,) * 3, mean=128.0, std=33.0, precision=PREC[kind])
    names = set(ctx.timing_get())
    ctx.timing(False)
    want_path = 'graph_split_f16' if kind in ('f16s', 'auto') else 'graph_mfma_' + kind
    if name == 'resnet_like':
        assert 'gx_conv1_add' in names and 'gx_add' not in names, names      # both shortcuts in conv1 epilogues
    assert ctx.last_path() == want_path and any(k.startswith('gx_stem_conv3') for k in names) and ('gx_head' in names or 'gx_conv3_32_head' in names), \
        (ctx.last_path(), names)
    assert got.shape == shape and not got[:off].any() and not got[:, :, -off:].any()
    d = np.abs(got - ref)
    assert d.max() < TOL['f16s' if kind == 'auto' else kind], '%s %s vs fp32 oracle: max %g' % (name, kind, d.max())
    assert ref[off:-off, off:-off, off:-off].std() > 1e-4
    return got


@pytest.mark.parametrize('name,tile,shape', CASES)
def test_remaining_factories_on_split_halves_are_fp32_grade(ctx, name, tile, shape):
    _check(ctx, name, tile, shape, 'f16s')


@pytest.mark.parametrize('name,tile,shape', CASES[::2])
@pytest.mark.parametrize('kind', ['f16', 'bf16'])
def test_remaining_factories_plain_16_bit(ctx, name, tile, shape, kind):
    _check(ctx, name, tile, shape, kind)


@pytest.mark.parametrize('name,tile,shape', CASES[::2])
def test_auto_picks_the_split_executor_for_all_ten_factories(ctx, name, tile, shape):
    """precision 'auto' = fp32-grade on the fastest executor: split halves for these four as well;
    slabs == whole (the N-GPU decomposition), run to run identical"""
    from flypylib_amd import multi_gpu
    whole = _check(ctx, name, tile, shape, 'auto')
    g, off, stride, prog = _setup(ctx, name, tile)
    u8 = synth.em_volume_u8(13, shape)
    kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_AUTO)
    again = prog.infer_volume(u8, (tile,) * 3, (off,) * 3, **kw)
    assert np.array_equal(again, whole)
    nz = multi_gpu.n_tile_rows(shape[0], tile, off)
    if nz >= 2:
        out = np.zeros(shape, np.float32)
        for lo, hi in multi_gpu.slab_partition(nz, 2):
            prog.infer_volume(u8, (tile,) * 3, (off,) * 3, z_range=(lo, hi), dst=out, **kw)
        assert np.array_equal(out, whole)


def test_half_range_guard_of_the_graph_executor(ctx):
    """an activation beyond 65504: 'f16s' fails, 'auto' reruns on fp32 and says so"""
    g, off, stride, prog = _setup(ctx, 'resnet_like', 38)
    w = g.get_weights()
    w[0] = w[0] * np.float32(3e4)              # the first convolution's kernel
    g.set_weights(w)
    prog = _capi.Program(ctx, g, stride)
    u8 = synth.em_volume_u8(5, (50, 38, 41))
    with pytest.raises(_capi.FplHipError, match='IEEE-half range'):
        prog.infer_volume(u8, (38,) * 3, (off,) * 3, mean=128.0, std=33.0, precision=_capi.PREC_F16S)
    got = prog.infer_volume(u8, (38,) * 3, (off,) * 3, mean=128.0, std=33.0, precision=_capi.PREC_AUTO)
    assert ctx.last_path() == 'mfma_f32(range)' and np.isfinite(got).all()


def test_graphs_outside_the_executors_are_refused(ctx):
    """a layer the 16-bit kernels do not have (a 3x3x3 convolution with 96 outputs): 'f16s' says no,
    'auto' gives it the fp32 MFMA executor"""
    g = LayerGraph(22)
    x = g.conv_bn_relu(g.input(), 32, 3)
    x = g.conv_bn_relu(x, 96, 3)
    g.finish(g.conv(x, 1, 1, use_bias=True, activation='sigmoid'))
    synth.synthetic_weights(g, 3)
    prog = _capi.Program(ctx, g, (1, 1, 1))
    u8 = synth.em_volume_u8(1, (40, 30, 33))
    with pytest.raises(_capi.FplHipError, match='split-half kernels'):
        prog.infer_volume(u8, (22,) * 3, (2,) * 3, mean=128.0, std=33.0, precision=_capi.PREC_F16S)
    prog.infer_volume(u8, (22,) * 3, (2,) * 3, mean=128.0, std=33.0, precision=_capi.PREC_AUTO)
    assert ctx.last_path() == 'mfma_f32'


def _custom_graph(in_sz=None):
    """not one of the reference's factories: 16- and 48-channel layers (padded to 32 / 64), a skip that is
    cropped and concatenated behind an UpSampling3D (the parity form's leading chunks), a residual Add of two
    uncropped tensors, a biased head behind the Add (the stand-alone fp32 head kernel)"""
    g = LayerGraph(in_sz)
    x = g.conv_bn_relu(g.input(), 16, 3)
    x = g.conv_bn_relu(x, 48, 3)                 # T - 4
    y = g.conv_bn_relu(g.pool(x), 64, 3)         # (T - 4) / 2 - 2
    y = g.conv_bn_relu(y, 48, 1)
    u = g.concat(g.up(y, 2), g.crop(x, 2))       # T - 8, 48 + 48 channels
    z = g.conv_bn_relu(u, 32, 3)                 # T - 10
    a = g.conv_bn_relu(z, 32, 1)                 # (an activation in front of the Add: the stand-alone Add kernel;
    r = g.relu(g.add(a, z))                      #  resnet_like's shortcuts take the fused conv1 + Add epilogue)
    return g.finish(g.conv(r, 1, 1, use_bias=True, activation='sigmoid')), (11, 5, 1), 44, None


@pytest.mark.parametrize('kind', ['f16s', 'f16', 'auto'])
def test_a_layer_program_of_the_callers_own(ctx, kind):
    """the graph executor is keyed on layer kinds and widths, not on factory names"""
    tile, off, shape = 44, 5, (60, 44, 77)
    g = _custom_graph(tile)[0]
    synth.synthetic_weights(g, 17)
    prog = _capi.Program(ctx, g, (1, 1, 1))
    u8 = synth.em_volume_u8(3, shape)
    img = (u8.astype(np.float32) - np.float32(128)) / np.float32(33)
    ref = _oracle(g, img, tile, off, (1, 1, 1))
    ctx.timing(True)
    ctx.timing_reset()
    got = prog.infer_volume(u8, (tile,) * 3, (off,) * 3, mean=128.0, std=33.0, precision=PREC[kind])
    names = set(ctx.timing_get())
    ctx.timing(False)
    assert ctx.last_path() == ('graph_mfma_f16' if kind == 'f16' else 'graph_split_f16')
    assert {'gx_stem_conv3', 'gx_add', 'gx_head', 'gx_conv1', 'gx_conv3_32', 'gx_conv3_64'} <= names, names
    d = np.abs(got - ref)
    assert d.max() < TOL['f16' if kind == 'f16' else 'f16s'], d.max()
    assert ref[off:-off, off:-off, off:-off].std() > 1e-4


@pytest.mark.parametrize('name,tiles', [('baseline_model', 6), ('unet_like_vol', 4)])
def test_graph_executor_at_full_size_properties(ctx, name, tiles):
    """the bench legs' volumes (542^3 / 372^3) on the default path: (a) two and three Z slabs of tile rows ==
    the whole volume, bit for bit (super-tiles or not); (b) a zero rf_offset shell; (c) three reference tiles
    (corner, interior, far corner) within 1e-5 of the fp32 oracle"""
    from flypylib_amd import multi_gpu
    factory = getattr(fplmodels, name)
    _, rf, infer_sz, _ = factory()
    tile, off, stride = fplutils.to3d(infer_sz)[0], fplutils.to3d(rf[1])[0], fplutils.to3d(rf[2])
    pitch = tile - 2 * off
    n = tiles * pitch + 2 * off
    g = factory(tile)[0]
    synth.synthetic_weights(g, 1234)
    prog = _capi.Program(ctx, g, stride)
    src = ctx.malloc((n, n, n), np.uint8)
    ctx.synth_volume_u8(20250101, (n, n, n), out=src)
    dst = ctx.malloc((n, n, n), np.float32)
    kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_AUTO, dims=(n, n, n))
    prog.infer_volume(src, (tile,) * 3, (off,) * 3, dst=dst, **kw)
    assert ctx.last_path() == 'graph_split_f16'
    whole = dst.to_host()
    for ax in range(3):
        lo = [slice(None)] * 3
        hi = [slice(None)] * 3
        lo[ax], hi[ax] = slice(0, off), slice(n - off, n)
        assert not whole[tuple(lo)].any() and not whole[tuple(hi)].any()
    assert whole[off:-off, off:-off, off:-off].std() > 1e-4
    rows = multi_gpu.n_tile_rows(n, tile, off)
    dst2 = ctx.malloc((n, n, n), np.float32)
    for parts in (2, 3):
        for zr in multi_gpu.slab_partition(rows, parts):
            prog.infer_volume(src, (tile,) * 3, (off,) * 3, dst=dst2, z_range=zr, **kw)
        assert np.array_equal(dst2.to_host()[off:n - off], whole[off:n - off]), parts
    u8 = src.to_host()
    last = (tiles - 1) * pitch
    for org in ((0, 0, 0), (2 * pitch, 3 * pitch, pitch), (last, last, last)):
        sl = tuple(slice(o, o + tile) for o in org)
        img = (u8[sl].astype(np.float32) - np.float32(128)) / np.float32(33)
        ref = _oracle(g, img, tile, off, stride)
        d = np.abs(whole[sl][off:-off, off:-off, off:-off] - ref[off:-off, off:-off, off:-off])
        assert d.max() < 1e-5, (org, d.max())
    for b in (src, dst, dst2):
        b.free()
    prog.close()
