"""GPU parity: the HIP inference path (through the C ABI) against the CPU oracle
on the same seeded inputs.  Tolerance for the fp32 path: 1e-3 on probabilities
(BASELINE north star); observed errors are ~1e-6."""
import numpy as np
import pytest

from flypylib_amd import FplNetwork, _capi, fplmodels, synth
from flypylib_amd.program import LayerGraph
from oracle import cnn_oracle, infer_oracle
from tests import helpers

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _prog(ctx, graph, stride=(1, 1, 1)):
    return _capi.Program(ctx, graph, stride)


def test_device_is_gfx950(ctx):
    info = ctx.device_info()
    assert 'gfx950' in info['name'] and info['n_cu'] >= 200


def test_synth_volume_device_matches_host(ctx):
    dims, org = (37, 70, 129), (5, 60, 1000)
    dev = ctx.synth_volume_u8(99, dims, org)
    assert np.array_equal(dev, synth.em_volume_u8(99, dims, org))


@pytest.mark.parametrize('dims,org', [((3, 5, 7), (0, 0, 0)), ((2, 3, 1), (62, 63, 64)),
                                      ((9, 66, 130), (120, 1, 50)), ((1, 1, 4099), (7, 7, 60))])
def test_synth_volume_ragged_sizes(ctx, dims, org):
    """the kernel writes four voxels per thread: rows that are not multiples of four,
    chunks that cross rows / planes / blob cells"""
    assert np.array_equal(ctx.synth_volume_u8(5, dims, org), synth.em_volume_u8(5, dims, org))


def test_synth_substack_clips_to_the_extent(ctx):
    """`fri_get_image`'s zero padding outside the volume (fplobjdetect.py:1044-1070)"""
    extent, size, origin = (70, 90, 80), 45, (-10, 60, 50)
    dst = np.empty((size,) * 3, np.uint8)
    ctx.synth_substack_u8(11, extent, (size,) * 3, origin, dst)
    want = np.zeros((size,) * 3, np.uint8)
    lo = [max(0, -o) for o in origin]
    hi = [min(size, e - o) for e, o in zip(extent, origin)]
    g0 = [o + l for o, l in zip(origin, lo)]
    want[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = synth.em_volume_u8(
        11, [h - l for h, l in zip(hi, lo)], g0)
    assert np.array_equal(dst, want) and want.any()


@pytest.mark.parametrize('k,cin,cout,act', [
    (3, 1, 48, 'relu'), (1, 48, 48, 'relu'), (3, 48, 48, 'relu'),
    (1, 48, 96, 'relu'), (1, 96, 1, 'sigmoid'), (3, 32, 64, None),
    (3, 5, 7, 'relu')])
def test_single_conv_layer(ctx, k, cin, cout, act):
    rng = np.random.default_rng(k * 100 + cin)
    g = LayerGraph(10 if k == 3 else 6, seed=cin)
    x = g.input()
    if cin != 1:
        x = g.conv(x, cin, 1)                 # lift to cin channels
    y = g.conv(x, cout, k, use_bias=(act == 'sigmoid'), activation=act
               if act == 'sigmoid' else None)
    if act == 'relu':
        y = g.bn_relu(y)
    g.finish(y)
    g.randomize_bn(cout)
    sz = g.in_sz[0]
    inp = rng.standard_normal((3, sz, sz, sz, 1)).astype(np.float32)
    got = _prog(ctx, g).forward(inp)
    ref = cnn_oracle.graph_forward(g, inp)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) < 1e-4


def test_vgg_like_tile_forward(ctx):
    g = fplmodels.vgg_like(30)[0]
    synth.synthetic_weights(g, 5)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 30, 30, 30, 1)).astype(np.float32)
    got = _prog(ctx, g, (4, 4, 4)).forward(x)
    ref = cnn_oracle.vgg_like_forward(x, g.weights, upsample_stride=4)
    assert got.shape == ref.shape == (2, 16, 16, 16, 1)
    assert np.max(np.abs(got - ref)) < TOL
    assert ref.std() > 1e-3, 'degenerate test network'


def test_unet_like2_tile_forward(ctx):
    g = fplmodels.unet_like2(28)[0]
    synth.synthetic_weights(g, 6)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 28, 28, 28, 1)).astype(np.float32)
    got = _prog(ctx, g).forward(x)
    ref = cnn_oracle.unet_like2_forward(x, g.weights)
    assert got.shape == ref.shape == (2, 10, 10, 10, 1)
    assert np.max(np.abs(got - ref)) < TOL
    assert ref.std() > 1e-3


@pytest.mark.parametrize('factory,size', [
    (fplmodels.baseline_model, 22), (fplmodels.vgg_like2, 28),
    (fplmodels.resnet_like, 22), (fplmodels.unet_like, 22),
    (fplmodels.unet_like3, 36), (fplmodels.unet_like4, 44),
    (fplmodels.unet_like4b, 44), (fplmodels.unet_like_vol, 62)])
def test_other_architectures_forward(ctx, factory, size):
    g, rf, _, _ = factory(size)
    synth.synthetic_weights(g, 7)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((1, size, size, size, 1)).astype(np.float32)
    got = _prog(ctx, g).forward(x)
    ref = cnn_oracle.graph_forward(g, x)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) < TOL


def test_unet_incompatible_tile_is_an_error(ctx):
    g = fplmodels.unet_like2()[0]
    p = _prog(ctx, g)
    with pytest.raises(_capi.FplHipError, match='concatenate'):
        p.forward(np.zeros((1, 26, 26, 26, 1), np.float32))


def _crop_identity_program(ctx, off, stride=1):
    g = LayerGraph(None)
    y = g.conv(g.crop(g.input(), off), 1, 1)
    g.finish(y)
    g.set_weights([np.ones((1, 1, 1, 1, 1), np.float32)])
    return _prog(ctx, g, (stride,) * 3)


def test_infer_lattice_matches_reference_golden(ctx, golden):
    """a crop-identity program through fpl_infer_volume reproduces the outputs of
    the reference's own FplNetwork.infer (tests/golden/infer_lattice.npz)"""
    g = golden('infer_lattice.npz')
    done = 0
    for name in [str(n) for n in g['names']]:
        if str(g[name + '_fn']) != '_crop_identity':
            continue
        shape = tuple(int(v) for v in g[name + '_shape'])
        isz = tuple(int(v) for v in g[name + '_isz'])
        off = tuple(int(v) for v in g[name + '_off'])
        img = synth.hash_uniform_f32(int(g[name + '_seed']), shape)
        pred = _crop_identity_program(ctx, off[0]).infer_volume(img, isz, off)
        assert helpers.sha(pred) == str(g[name + '_pred_sha']), name
        done += 1
    assert done == 4


@pytest.mark.parametrize('shape', [(50, 47, 41), (46, 46, 46), (31, 30, 64)])
def test_infer_volume_vgg_matches_oracle_lattice(ctx, shape):
    g = fplmodels.vgg_like(30)[0]
    synth.synthetic_weights(g, 8)
    prog = _prog(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(3, shape)
    img = ((u8.astype(np.float32) - np.float32(128)) / np.float32(33))

    def predict(batch):
        return cnn_oracle.vgg_like_forward(batch.astype(np.float32), g.weights,
                                           upsample_stride=4)
    ref = infer_oracle.infer_lattice(img, (30,) * 3, (7,) * 3, predict)
    got_f = prog.infer_volume(img, (30,) * 3, (7,) * 3)
    got_u = prog.infer_volume(u8, (30,) * 3, (7,) * 3, mean=128.0, std=33.0)
    assert got_f.shape == shape and got_f.dtype == np.float32
    assert np.max(np.abs(got_f - ref)) < TOL
    assert np.array_equal(got_f, got_u)          # on-device normalisation
    assert not got_f[:7].any() and not got_f[:, :, -7:].any()
    assert ref[7:-7, 7:-7, 7:-7].std() > 1e-3


def test_infer_volume_slabs_equal_whole(ctx):
    """Z-slab sharding (the multi-GPU partition) reproduces the whole-volume
    result row for row"""
    from flypylib_amd import multi_gpu
    g = fplmodels.vgg_like(30)[0]
    synth.synthetic_weights(g, 9)
    prog = _prog(ctx, g, (4, 4, 4))
    img = synth.hash_uniform_f32(4, (90, 40, 36)) - np.float32(0.5)
    whole = prog.infer_volume(img, (30,) * 3, (7,) * 3)
    n = multi_gpu.n_tile_rows(90, 30, 7)
    assert n == 5
    out = np.full_like(whole, np.nan)
    for zr in multi_gpu.slab_partition(n, 3):
        lo, hi = multi_gpu.slab_rows(zr, 90, 30, 7)
        part = prog.infer_volume(img, (30,) * 3, (7,) * 3, z_range=zr)
        out[lo:hi] = part[lo:hi]
    assert np.array_equal(out, whole)


@pytest.mark.parametrize('shape', [(90, 40, 36), (131, 120, 97)])
def test_f32_super_tiles_equal_the_reference_tiles_bit_for_bit(ctx, shape, monkeypatch):
    """the fp32 MFMA path computes m^3 neighbouring reference tiles as one larger tile
    (infer.hip, tiles_mergeable); the arithmetic per voxel is the same, so the volume is
    bit-identical to the one-tile-at-a-time lattice - whole volume and rank slabs"""
    from flypylib_amd import multi_gpu
    g = fplmodels.vgg_like(30)[0]
    synth.synthetic_weights(g, 19)
    prog = _prog(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(23, shape)
    kw = dict(mean=128.0, std=33.0)
    merged = prog.infer_volume(u8, (30,) * 3, (7,) * 3, **kw)
    n = multi_gpu.n_tile_rows(shape[0], 30, 7)
    parts = [(zr, prog.infer_volume(u8, (30,) * 3, (7,) * 3, z_range=zr, **kw))
             for zr in multi_gpu.slab_partition(n, 2)]
    monkeypatch.setenv('FPL_NO_TILE_MERGE', '1')
    plain = prog.infer_volume(u8, (30,) * 3, (7,) * 3, **kw)
    assert ctx.last_path() == 'mfma_f32'
    assert plain[7:-7, 7:-7, 7:-7].std() > 1e-3
    assert np.array_equal(merged, plain)
    for zr, part in parts:
        lo, hi = multi_gpu.slab_rows(zr, shape[0], 30, 7)
        assert np.array_equal(part[lo:hi], plain[lo:hi])


def test_conv1_weights_in_registers_equals_the_generic_kernel(ctx, monkeypatch):
    """conv1_f32_wreg (cin 48: fragments loaded once per wave, next voxels prefetched)
    keeps the generic kernel's group and K order: same bits"""
    g = fplmodels.vgg_like(30)[0]
    synth.synthetic_weights(g, 21)
    prog = _prog(ctx, g, (4, 4, 4))
    x = (synth.hash_uniform_f32(6, (3, 30, 30, 30)) - np.float32(0.5))
    fast = prog.forward(x)
    monkeypatch.setenv('FPL_CONV1_GENERIC', '1')
    slow = prog.forward(x)
    assert fast.std() > 1e-3 and np.array_equal(fast, slow)


def test_fplnetwork_infer_api(ctx):
    net = FplNetwork(fplmodels.vgg_like)
    assert net.rf_size == (18, 18, 18) and net.rf_offset == (7, 7, 7)
    assert net.rf_stride == (4, 4, 4) and net.infer_sz == (102, 102, 102)
    with pytest.raises(AssertionError, match='not been trained'):
        net.infer(np.zeros((40, 40, 40), np.float32))
    net.infer_sz = (30, 30, 30)                 # small tiles keep the test fast
    synth.synthetic_weights(net.train_single, 10)
    net._set_infer()
    img = synth.hash_uniform_f32(5, (40, 44, 38)) - np.float32(0.5)
    pred = net.infer(img)

    def predict(batch):
        return cnn_oracle.vgg_like_forward(batch.astype(np.float32),
                                           net.train_single.weights, 4)
    ref = infer_oracle.infer_lattice(img, (30,) * 3, (7,) * 3, predict)
    assert pred.dtype == np.float32 and pred.shape == img.shape
    assert np.max(np.abs(pred - ref)) < TOL
    # Keras-style predict on the inference network
    tile = np.zeros((1, 30, 30, 30, 1), np.float32)
    tile[0, :, :, :, 0] = img[:30, :30, :30]
    assert np.max(np.abs(net.infer_network.predict(tile) - predict(tile))) < TOL


def test_fplnetwork_unet_reference_lattice(ctx):
    """unet_like2 keeps the reference lattice (tile 28 here, stride 10)"""
    net = FplNetwork(fplmodels.unet_like2)
    assert net.rf_offset == (9, 9, 9) and net.infer_sz == (100, 100, 100)
    net.infer_sz = (28, 28, 28)
    synth.synthetic_weights(net.train_single, 12)
    net._set_infer()
    img = synth.hash_uniform_f32(6, (45, 38, 31)) - np.float32(0.5)
    pred = net.infer(img)

    def predict(batch):
        return cnn_oracle.unet_like2_forward(batch.astype(np.float32),
                                             net.train_single.weights)
    ref = infer_oracle.infer_lattice(img, (28,) * 3, (9,) * 3, predict)
    assert np.max(np.abs(pred - ref)) < TOL


@pytest.mark.parametrize('prec', [_capi.PREC_F32, _capi.PREC_BF16])
@pytest.mark.parametrize('shape', [(12, 40, 40), (40, 14, 40), (40, 40, 13),
                                   (15, 15, 15), (16, 40, 40), (14, 14, 14)])
def test_volumes_with_no_or_one_valid_voxel_row(ctx, shape, prec):
    """edge cases of the lattice (fplnetwork.py:151-155): an axis with
    dim <= 2*rf_offset yields no tile at all -> an all-zero prediction; one with
    dim = 2*offset + 1 yields a single (mostly padded) tile"""
    g = fplmodels.vgg_like(30)[0]
    synth.synthetic_weights(g, 13)
    prog = _prog(ctx, g, (4, 4, 4))
    img = synth.hash_uniform_f32(9, shape) - np.float32(0.5)

    def predict(batch):
        return cnn_oracle.vgg_like_forward(batch.astype(np.float32), g.weights, 4)
    ref = infer_oracle.infer_lattice(img, (30,) * 3, (7,) * 3, predict)
    got = prog.infer_volume(img, (30,) * 3, (7,) * 3, precision=prec)
    assert got.shape == shape
    tol = TOL if prec == _capi.PREC_F32 else 5e-2
    assert np.max(np.abs(got - ref)) < tol
    if min(shape) <= 14:
        assert not got.any() and not ref.any()


def test_bad_arguments_raise_library_errors(ctx):
    g = fplmodels.vgg_like(30)[0]
    prog = _prog(ctx, g, (4, 4, 4))
    img = np.zeros((40, 40, 40), np.float32)
    with pytest.raises(_capi.FplHipError, match='infer_sz'):
        prog.infer_volume(img, (31,) * 3, (7,) * 3)          # 31-14 is not 4*n
    with pytest.raises(_capi.FplHipError, match='std'):
        prog.infer_volume(img, (30,) * 3, (7,) * 3, std=0.0)
    with pytest.raises(_capi.FplHipError, match='too small'):
        prog.forward(np.zeros((1, 10, 10, 10, 1), np.float32))
    # unet tile that breaks the concat shapes, on every precision path
    gu = fplmodels.unet_like2()[0]
    pu = _prog(ctx, gu)
    for prec in (_capi.PREC_F32, _capi.PREC_BF16):
        with pytest.raises(_capi.FplHipError):
            pu.infer_volume(img, (26,) * 3, (9,) * 3, precision=prec)


@pytest.mark.parametrize('name,tile', [('vgg_like', 46), ('unet_like', 34), ('unet_like2', 36),
                                       ('baseline_model', 46)])
def test_f32_fused_epilogues_equal_one_kernel_per_op(ctx, name, tile):
    """fp32 MFMA executor: conv3 -> conv1 (-> pool) fused into one kernel (1x1 conv chained
    in registers, pool in the epilogue) is the same k-ordered fmaf chain per output as the
    separate kernels (FPL_F32_UNFUSED=1): bit-identical"""
    import os
    from flypylib_amd import FplNetwork
    net = FplNetwork(getattr(fplmodels, name))
    net.infer_sz = (tile,) * 3
    synth.synthetic_weights(net.train_single, 4)
    net._set_infer()
    u8 = synth.em_volume_u8(12, (tile + 40, tile + 9, tile + 22))
    a = net.infer(u8, normalize=(128., 33.), precision='f32')
    assert ctx.last_path() == 'mfma_f32'
    os.environ['FPL_F32_UNFUSED'] = '1'
    try:
        b = net.infer(u8, normalize=(128., 33.), precision='f32')
    finally:
        del os.environ['FPL_F32_UNFUSED']
    assert a.std() > 0 and np.array_equal(a, b)
