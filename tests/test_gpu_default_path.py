"""What the default path ('auto' = split IEEE halves for vgg_like / unet_like2) must hold
beyond a single tile (round 5):

  * run-to-run identity: repeated inferences of one volume are bit-identical (a race between
    an inline-asm result and a late MFMA write-back showed up in round 4 as 1e-5 differences
    between runs: csrc/mfma_util.h::split_pk);
  * BASELINE.json's metric size, 520^3 (17 x 65 x 33 blocks in the mid kernel: partial bricks
    in every direction): slabs == whole, tile independence, zero shell, reference tiles against
    the fp32 CPU oracle at 1e-5;
  * the CPU oracle in the loop end to end on a volume of 270^3 (27 reference tiles, zero-padded
    edge tiles): CNN oracle -> voxel2obj oracle against FplNetwork.infer -> fplobjdetect.voxel2obj
    on the GPU at the pipeline's parameters (r 27, sigma 5) - the same point set, probabilities
    within 1e-5 (`/root/reference/flypylib/fplnetwork.py:136-189`, `fplobjdetect.py:132-257`).
"""
import numpy as np
import pytest

from flypylib_amd import _capi, fplmodels, fplobjdetect, multi_gpu, synth
from oracle import cnn_oracle, infer_oracle, voxel2obj_oracle
from tests import helpers
from tests.trained_fixture import blob_region_u8, trained_network

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,tile,off,n,stride', [('vgg_like', 102, 7, 384, (4, 4, 4)),
                                                    ('unet_like2', 100, 9, 346, (1, 1, 1))])
def test_default_path_is_deterministic_run_to_run(ctx, name, tile, off, n, stride):
    g = getattr(fplmodels, name)(tile)[0]
    synth.synthetic_weights(g, 5)
    prog = _capi.Program(ctx, g, stride)
    dims = (n,) * 3
    src = ctx.malloc(dims, np.uint8)
    ctx.synth_volume_u8(3, dims, out=src)
    dst = ctx.malloc(dims, np.float32)
    kw = dict(mean=128.0, std=33.0, precision=_capi.PREC_AUTO, dims=dims, dst=dst)
    prog.infer_volume(src, (tile,) * 3, (off,) * 3, **kw)
    assert 'split' in ctx.last_path(), ctx.last_path()
    ref = dst.to_host()
    assert ref[off:-off, off:-off, off:-off].std() > 1e-3
    for i in range(4):
        prog.infer_volume(src, (tile,) * 3, (off,) * 3, **kw)
        out = dst.to_host()
        assert np.array_equal(out, ref), \
            'run %d differs in %d voxels, max %.2e' % (i, int((out != ref).sum()), float(np.abs(out - ref).max()))
    src.free()
    dst.free()
    prog.close()


def test_vgg_like_520_cubed_split_properties(ctx):
    """BASELINE.json's metric size on the default path: (a) two and three Z slabs of tile rows ==
    the whole volume, bit for bit; (b) tile 102 == tile 142; (c) zero shell; (d) three reference
    tiles (corner, interior, far corner) within 1e-5 of the fp32 oracle."""
    n = 520
    g = fplmodels.vgg_like(102)[0]
    synth.synthetic_weights(g, 1234)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    src = ctx.malloc((n, n, n), np.uint8)
    ctx.synth_volume_u8(20250101, (n, n, n), out=src)
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
    for parts in (2, 3):
        for zr in multi_gpu.slab_partition(rows, parts):
            prog.infer_volume(src, (102,) * 3, (7,) * 3, dst=dst2, z_range=zr, **kw)
        assert np.array_equal(dst2.to_host()[7:n - 7], whole[7:n - 7]), parts
    prog.infer_volume(src, (142,) * 3, (7,) * 3, dst=dst2, **dict(kw, precision=_capi.PREC_F16S))
    assert np.array_equal(dst2.to_host(), whole)
    u8 = src.to_host()

    def f32(batch):
        return cnn_oracle.vgg_like_forward(batch.astype(np.float32), g.weights, 4)
    for org in ((0, 0, 0), (176, 264, 88), (416, 416, 416)):
        sl = tuple(slice(o, o + 102) for o in org)
        img = (u8[sl].astype(np.float32) - np.float32(128)) / np.float32(33)
        ref = infer_oracle.infer_lattice(img, (102,) * 3, (7,) * 3, f32)
        d = np.abs(whole[sl][7:-7, 7:-7, 7:-7] - ref[7:-7, 7:-7, 7:-7])
        assert d.max() < 1e-5, (org, d.max())
    for b in (src, dst, dst2):
        b.free()
    prog.close()


def test_cpu_oracle_in_the_loop_end_to_end_270_cubed(ctx):
    """The whole path against the CPU oracle on one volume: the oracle's CNN over the reference
    tile lattice (27 tiles of 102^3, the far ones zero-padded) and the oracle's voxel2obj of THAT
    prediction, against FplNetwork.infer (default precision) and fplobjdetect.voxel2obj on the
    GPU.  19.7 million voxels; trained weights, blobs 48 +- 3 apart, r 27, sigma 5, buffer 35."""
    n = 270
    net = trained_network('vgg_like', tile=102)
    u8, _, locs = blob_region_u8(9, n, step=48)
    norm = (128.0, 33.0)
    got = net.infer(u8, normalize=norm)
    assert 'split' in ctx.last_path(), ctx.last_path()
    g = net.infer_network.graph
    img = (u8.astype(np.float32) - np.float32(norm[0])) / np.float32(norm[1])

    def f32(batch):
        return cnn_oracle.graph_forward(g, batch.astype(np.float32), upsample_stride=net.rf_stride)
    want = infer_oracle.infer_lattice(img, (102,) * 3, (7,) * 3, f32)
    d = np.abs(got - want)
    print('270^3: default path vs CPU oracle max %.2e mean %.2e' % (d.max(), d.mean()))
    assert d.max() < 1e-5
    kw = dict(obj_min_dist=27, smoothing_sigma=5, buffer_sz=35, thd=0.1)
    a = voxel2obj_oracle.voxel2obj(want, **kw)
    b = fplobjdetect.voxel2obj(got, **kw)
    print('detections', len(a['conf']))
    assert len(a['conf']) >= 40
    moved = helpers.same_detections(a, b, 1e-5, tie=1e-6)
    assert moved <= 2, moved
    hit = np.linalg.norm(np.asarray(a['locs'])[:, None, :] - locs[None, :, :].astype(float), axis=2).min(axis=1)
    assert np.mean(hit <= 4.0) > 0.9


@pytest.mark.parametrize('dtype', ['uint8', 'float32'])
def test_host_to_host_pipeline_equals_the_plain_copy_path(ctx, monkeypatch, dtype):
    """fpl_infer_volume with host source and destination: groups of tile rows uploaded, computed and copied
    out by helper threads beside each other (csrc/infer.hip) == upload everything, compute, copy everything
    (FPL_NO_D2H_PIPE=1), for uint8 and float32 sources, whole volumes and a slab of rows; the result array
    comes from the binding's recycling pool"""
    n = 330                                            # 144 MB of float32 out: above the pipeline's 64 MB floor
    g = fplmodels.vgg_like(102)[0]
    synth.synthetic_weights(g, 77)
    prog = _capi.Program(ctx, g, (4, 4, 4))
    u8 = synth.em_volume_u8(9, (n, n, n))
    if dtype == 'uint8':
        src, kw = u8, dict(mean=128.0, std=33.0)
    else:
        src, kw = (u8.astype(np.float32) - np.float32(128)) / np.float32(33), dict()
    kw['precision'] = _capi.PREC_AUTO
    piped = prog.infer_volume(src, (102,) * 3, (7,) * 3, **kw)
    assert piped.base is not None                      # pool memory
    rows = multi_gpu.n_tile_rows(n, 102, 7)
    part = np.zeros((n, n, n), np.float32)
    prog.infer_volume(src, (102,) * 3, (7,) * 3, z_range=(1, rows), dst=part, **kw)
    monkeypatch.setenv('FPL_NO_D2H_PIPE', '1')
    plain = prog.infer_volume(src, (102,) * 3, (7,) * 3, **kw)
    part2 = np.zeros((n, n, n), np.float32)
    prog.infer_volume(src, (102,) * 3, (7,) * 3, z_range=(1, rows), dst=part2, **kw)
    assert np.array_equal(piped, plain) and plain[7:-7, 7:-7, 7:-7].std() > 1e-3
    assert np.array_equal(part, part2) and np.array_equal(part[100:], plain[100:]) and not part[:80].any()
    prog.close()
