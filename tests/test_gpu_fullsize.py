"""Full-size runs (BASELINE.json configs) checked through size-independent properties:
sharding == whole, tiling independence, zero border shell, spot checks of
reference tiles against the oracle, idempotent resume of the substack pipeline.
Everything stays on the device except what is compared."""
import numpy as np
import pytest

from flypylib_amd import _capi, fplmodels, fplobjdetect, multi_gpu, synth
from oracle import cnn_oracle, infer_oracle, voxel2obj_oracle

pytestmark = pytest.mark.gpu


def _vgg(ctx, seed=1234):
    g = fplmodels.vgg_like(102)[0]
    synth.synthetic_weights(g, seed)
    return g, _capi.Program(ctx, g, (4, 4, 4))


def test_vgg_like_1024_cubed_bf16_properties(ctx):
    """configs[1]: 1024^3 uint8, bf16.  (a) two Z slabs of tile rows == the whole
    volume, bit for bit; (b) the result does not depend on the tile size (102 vs
    142 = 128 + halo); (c) the rf_offset shell is zero; (d) three reference tiles
    (corner, interior, far edge) against the bf16-emulation oracle."""
    n = 1024
    g, prog = _vgg(ctx)
    src = ctx.malloc((n, n, n), np.uint8)
    ctx.synth_volume_u8(1, (n, n, n), out=src)
    dst = ctx.malloc((n, n, n), np.float32)
    kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_BF16, dims=(n, n, n))
    prog.infer_volume(src, (102,) * 3, (7,) * 3, dst=dst, **kw)
    whole = dst.to_host()
    # (c)
    for ax in range(3):
        lo = [slice(None)] * 3
        hi = [slice(None)] * 3
        lo[ax], hi[ax] = slice(0, 7), slice(n - 7, n)
        assert not whole[tuple(lo)].any() and not whole[tuple(hi)].any()
    assert whole[7:-7, 7:-7, 7:-7].std() > 1e-3
    # (a) slabs of tile rows written into one volume
    rows = multi_gpu.n_tile_rows(n, 102, 7)
    parts = multi_gpu.slab_partition(rows, 2)
    dst2 = ctx.malloc((n, n, n), np.float32)
    ctx.memcpy(dst2, np.zeros(16, np.uint8), 16)     # touch
    for zr in parts:
        prog.infer_volume(src, (102,) * 3, (7,) * 3, dst=dst2, z_range=zr, **kw)
    sharded = dst2.to_host()
    lo_z, hi_z = 7, n - 7
    assert np.array_equal(sharded[lo_z:hi_z], whole[lo_z:hi_z])
    del sharded
    # (b)
    prog.infer_volume(src, (142,) * 3, (7,) * 3, dst=dst2, **kw)
    assert np.array_equal(dst2.to_host(), whole)
    # (d) tiles of the reference lattice (origins multiples of 88) vs the oracle
    u8 = src.to_host()

    def emu(batch):
        return cnn_oracle.vgg_like_forward_bf16emu(batch.astype(np.float32), g.weights, 4)
    for org in ((0, 0, 0), (440, 528, 352), (880, 880, 880)):
        sl = tuple(slice(o, o + 102) for o in org)
        img = (u8[sl].astype(np.float32) - np.float32(128)) / np.float32(33)
        ref = infer_oracle.infer_lattice(img, (102,) * 3, (7,) * 3, emu)
        got = whole[sl][7:-7, 7:-7, 7:-7]
        d = np.abs(got - ref[7:-7, 7:-7, 7:-7])
        assert d.max() < 1e-2 and d.mean() < 1e-4 and np.mean(d > 1e-3) < 1e-3, (org, d.max())
    # (e) the IEEE-half build of the same kernels meets the 1e-3 gate against the fp32
    # oracle on those tiles
    kw16 = dict(kw, precision=_capi.PREC_F16)
    prog.infer_volume(src, (102,) * 3, (7,) * 3, dst=dst2, **kw16)
    half = dst2.to_host()

    def f32(batch):
        return cnn_oracle.vgg_like_forward(batch.astype(np.float32), g.weights, 4)
    for org in ((0, 0, 0), (440, 528, 352), (880, 880, 880)):
        sl = tuple(slice(o, o + 102) for o in org)
        img = (u8[sl].astype(np.float32) - np.float32(128)) / np.float32(33)
        ref = infer_oracle.infer_lattice(img, (102,) * 3, (7,) * 3, f32)
        d = np.abs(half[sl][7:-7, 7:-7, 7:-7] - ref[7:-7, 7:-7, 7:-7])
        assert d.max() < 1e-3, (org, d.max())
    for b in (src, dst, dst2):
        b.free()


def test_vgg_like_1024_cubed_split_halves_properties(ctx):
    """configs[1] on the path the benchmark headlines and 'auto' picks (split IEEE halves,
    'f16s'): (a) two Z slabs of tile rows == the whole volume, bit for bit; (b) tile 102 == tile
    142; (c) zero shell; (d) three reference tiles (corner, interior, far edge) within 1e-5 of
    the fp32 oracle - the reference's own arithmetic (flypylib/fplnetwork.py:175-176)."""
    n = 1024
    g, prog = _vgg(ctx)
    src = ctx.malloc((n, n, n), np.uint8)
    ctx.synth_volume_u8(1, (n, n, n), out=src)
    dst = ctx.malloc((n, n, n), np.float32)
    kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_AUTO, dims=(n, n, n))
    prog.infer_volume(src, (102,) * 3, (7,) * 3, dst=dst, **kw)
    assert ctx.last_path() == 'vgg_split_f16'
    whole = dst.to_host()
    for ax in range(3):
        lo = [slice(None)] * 3
        hi = [slice(None)] * 3
        lo[ax], hi[ax] = slice(0, 7), slice(n - 7, n)
        assert not whole[tuple(lo)].any() and not whole[tuple(hi)].any()
    assert whole[7:-7, 7:-7, 7:-7].std() > 1e-3
    rows = multi_gpu.n_tile_rows(n, 102, 7)
    dst2 = ctx.malloc((n, n, n), np.float32)
    ctx.memcpy(dst2, np.zeros(16, np.uint8), 16)     # touch
    for zr in multi_gpu.slab_partition(rows, 2):
        prog.infer_volume(src, (102,) * 3, (7,) * 3, dst=dst2, z_range=zr, **kw)
    assert np.array_equal(dst2.to_host()[7:n - 7], whole[7:n - 7])
    kw['precision'] = _capi.PREC_F16S
    prog.infer_volume(src, (142,) * 3, (7,) * 3, dst=dst2, **kw)
    assert np.array_equal(dst2.to_host(), whole)
    u8 = src.to_host()

    def f32(batch):
        return cnn_oracle.vgg_like_forward(batch.astype(np.float32), g.weights, 4)
    for org in ((0, 0, 0), (440, 528, 352), (880, 880, 880)):
        sl = tuple(slice(o, o + 102) for o in org)
        img = (u8[sl].astype(np.float32) - np.float32(128)) / np.float32(33)
        ref = infer_oracle.infer_lattice(img, (102,) * 3, (7,) * 3, f32)
        d = np.abs(whole[sl][7:-7, 7:-7, 7:-7] - ref[7:-7, 7:-7, 7:-7])
        assert d.max() < 1e-5, (org, d.max())
    for b in (src, dst, dst2):
        b.free()


def test_unet_like2_512_cubed_bf16_properties(ctx):
    """configs[2] shape at 510^3 (6^3 reference tiles 100^3 -> 82^3): Z slabs ==
    whole (bit-exact), zero shell, two reference tiles vs the bf16-emulation oracle"""
    n = 6 * 82 + 18
    g = fplmodels.unet_like2(100)[0]
    synth.synthetic_weights(g, 7)
    prog = _capi.Program(ctx, g, (1, 1, 1))
    src = ctx.malloc((n, n, n), np.uint8)
    ctx.synth_volume_u8(3, (n, n, n), out=src)
    dst = ctx.malloc((n, n, n), np.float32)
    kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_BF16, dims=(n, n, n))
    prog.infer_volume(src, (100,) * 3, (9,) * 3, dst=dst, **kw)
    whole = dst.to_host()
    assert not whole[:9].any() and not whole[:, :, -9:].any()
    dst2 = ctx.malloc((n, n, n), np.float32)
    for zr in multi_gpu.slab_partition(multi_gpu.n_tile_rows(n, 100, 9), 3):
        prog.infer_volume(src, (100,) * 3, (9,) * 3, dst=dst2, z_range=zr, **kw)
    assert np.array_equal(dst2.to_host()[9:-9], whole[9:-9])
    u8 = src.to_host()

    def emu(batch):
        return cnn_oracle.unet_like2_forward_bf16emu(batch.astype(np.float32), g.weights)
    for org in ((0, 0, 0), (164, 328, 410)):
        sl = tuple(slice(o, o + 100) for o in org)
        img = (u8[sl].astype(np.float32) - np.float32(128)) / np.float32(33)
        ref = infer_oracle.infer_lattice(img, (100,) * 3, (9,) * 3, emu)
        d = np.abs(whole[sl][9:-9, 9:-9, 9:-9] - ref[9:-9, 9:-9, 9:-9])
        assert d.max() < 2e-2 and d.mean() < 2e-4, (org, d.max(), d.mean())
    for b in (src, dst, dst2):
        b.free()


def test_voxel2obj_on_a_582_cubed_substack_is_bit_exact(ctx):
    """configs[4] substack size (512 + 2*35), r 27, sigma 5, buffer 35: the same
    points and confidences as the CPU oracle"""
    n = 582
    prob = synth.blob_prob_volume(11, (n, n, n), period=64, radius=9.0)
    got = fplobjdetect.voxel2obj(prob, 27, 5, (100, 200, 300), 35, 0.1)
    ref = voxel2obj_oracle.voxel2obj(prob, 27, 5, (100, 200, 300), 35, 0.1)
    assert len(ref['conf']) > 300
    assert np.array_equal(got['locs'], ref['locs']) and np.array_equal(got['conf'], ref['conf'])


def test_pipeline_1024_cubed_is_idempotent_and_order_independent(ctx, tmp_path):
    """configs[4] flow on a synthetic 1024^3 volume (8 substacks of 512 + 35 buffer):
    a second run over the finished working directory recomputes nothing and returns
    the same points; processing the substacks in reverse order gives the same
    per-substack results (no state leaks between substacks)"""
    import pickle
    from flypylib_amd import FplNetwork
    net = FplNetwork(fplmodels.vgg_like, precision='bf16')
    synth.synthetic_weights(net.train_single, 9)
    net._set_infer()
    src = 'synth://5,1024,1024,1024'
    roi = [(512, z, y, x) for z in (0, 512) for y in (0, 512) for x in (0, 512)]
    norm = [128., 33., 0.5]
    a = fplobjdetect.full_roi_inference(src, None, roi, net, 0.1, str(tmp_path / 'a'), norm)
    again = fplobjdetect.full_roi_inference(src, None, roi, net, 0.1, str(tmp_path / 'a'), norm)
    assert np.array_equal(a['locs'], again['locs']) and np.array_equal(a['conf'], again['conf'])
    fplobjdetect.full_roi_inference(src, None, roi[::-1], net, 0.1, str(tmp_path / 'b'), norm)
    for r in roi:
        ss = fplobjdetect.szyx(*r)
        pa = pickle.load(open(fplobjdetect.fri_filename(str(tmp_path / 'a'), ss), 'rb'))
        pb = pickle.load(open(fplobjdetect.fri_filename(str(tmp_path / 'b'), ss), 'rb'))
        assert np.array_equal(pa['locs'], pb['locs']) and np.array_equal(pa['conf'], pb['conf'])
    assert len(a['conf']) > 0
    # every detection lies inside its substack (buffer detections are dropped)
    assert a['locs'].min() >= 0 and a['locs'].max() < 1024


def test_configs2_rank_share_equals_the_whole_volume_rows(ctx):
    """configs[2] at its stated size: unet_like2 (trained fixture) over the 1024 x 2048 x 2048
    volume is 13 tile rows of pitch 82 along z; 8 ranks take 2,2,2,2,2,1,1,1 of them.  Rank
    0's slab (2 rows + halo = 182 z rows) run as a standalone volume - what a torchrun rank
    of bench.py / multi_gpu does - equals, bit for bit, the same rows of a run over the
    WHOLE volume on this one GPU; so does rank 7's (the ragged last row)."""
    import torch
    from tests.trained_fixture import trained_weights
    Z, Y, X = 1024, 2048, 2048
    tile, off, world = 100, 9, 8
    g = fplmodels.unet_like2(tile)[0]
    g.set_weights(trained_weights('unet_like2'))
    prog = _capi.Program(ctx, g, (1, 1, 1))
    n_rows = multi_gpu.n_tile_rows(Z, tile, off)
    parts = multi_gpu.slab_partition(n_rows, world)
    assert n_rows == 13 and [e - b for b, e in parts] == [2, 2, 2, 2, 2, 1, 1, 1]
    pitch = tile - 2 * off
    whole_src = torch.empty((Z, Y, X), dtype=torch.uint8, device='cuda')
    ctx.synth_volume_u8(3, (Z, Y, X), out=whole_src)
    whole_dst = torch.empty((Z, Y, X), dtype=torch.float32, device='cuda')
    kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_AUTO)     # the default: split halves
    prog.infer_volume(whole_src, (tile,) * 3, (off,) * 3, dst=whole_dst, dims=(Z, Y, X), **kw)
    assert ctx.last_path() == 'unet_split_f16'
    assert float(whole_dst[off:-off, off:-off, off:-off].std()) > 0
    for rank in (0, 7):
        zb, ze = parts[rank]
        z_lo, z_hi = zb * pitch, min(ze * pitch + 2 * off, Z)
        slab_src = whole_src[z_lo:z_hi].contiguous()
        slab_dst = torch.empty((z_hi - z_lo, Y, X), dtype=torch.float32, device='cuda')
        prog.infer_volume(slab_src, (tile,) * 3, (off,) * 3, dst=slab_dst,
                          dims=(z_hi - z_lo, Y, X), **kw)
        lo, hi = multi_gpu.slab_rows((zb, ze), Z, tile, off)       # rows the rank owns
        # its interior rows (the whole-volume run zeroes the outer shell, the slab run its own)
        a, b = max(lo, off), min(hi, Z - off)
        assert torch.equal(slab_dst[a - z_lo:b - z_lo], whole_dst[a:b]), rank
        del slab_src, slab_dst
    del whole_src, whole_dst
    prog.close()


def test_configs4_rank_share_of_a_4096_cubed_roi(ctx, tmp_path):
    """configs[4] at its stated size, a rank's share: the 4096^3 synthetic ROI is 512
    substacks of 512^3 (+ 35 buffer: 582^3 cubes); of 8 ranks rank 3 takes every 8th.
    Its first 8 substacks run through full_roi_inference at the DEFAULT precision
    (split halves: fp32-grade) with the trained vgg_like fixture; the substack with the
    most detections is recomputed stage by stage and its point list compared with the CPU
    oracle's voxel2obj on the same device prediction - identical."""
    import os
    import pickle
    import warnings
    from flypylib_amd import fplpipeline
    from tests.trained_fixture import trained_network
    n = 4096
    net = trained_network('vgg_like', tile=102)
    wd = str(tmp_path)
    src = 'synth://5,%d,%d,%d' % (n, n, n)
    fplobjdetect.gen_full_tab_roi(wd + '/roi', src, None, step_size=512)
    roi = fplobjdetect.roi_from_txt(wd + '/roi_00.txt')[0]
    assert len(roi) == 512
    mine = roi[3::8][:8]
    norm = [128., 33., 0.5]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out = fplobjdetect.full_roi_inference(src, None, mine, net, 0.1, wd + '/work', norm)
    assert ctx.last_path() == 'vgg_split_f16'
    counts = [len(pickle.load(open(fplobjdetect.fri_filename(wd + '/work', s_), 'rb'))['conf'])
              for s_ in mine]
    assert len(out['conf']) == sum(counts)
    ss = mine[int(np.argmax(counts))]
    print('detections per substack', counts)
    sz = ss.size + 70
    cube = ctx.malloc((sz,) * 3, np.uint8)
    pred = ctx.malloc((sz,) * 3, np.float32)
    ctx.synth_substack_u8(5, (n, n, n), (sz,) * 3, [ss.z - 35, ss.y - 35, ss.x - 35], cube)
    st = fplpipeline.normalisation_from_histogram(ctx.histogram_u8(cube), norm)
    net.infer_network.program.infer_volume(cube, net.infer_sz, net.rf_offset, mean=st['mn_use'],
                                           std=norm[1], precision=_capi.PREC_AUTO, dst=pred,
                                           dims=(sz,) * 3)
    ref = voxel2obj_oracle.voxel2obj(pred.to_host(), 27, 5, (ss.x - 35, ss.y - 35, ss.z - 35), 35, 0.1)
    got = pickle.load(open(fplobjdetect.fri_filename(wd + '/work', ss), 'rb'))
    assert np.array_equal(ref['locs'], got['locs']) and np.array_equal(ref['conf'], got['conf'])
    cube.free()
    pred.free()
