"""CPU: product host logic (layer programs, lowering, sharding) and the C-ABI
library surface.  No compute calls - there is no GPU in the build container."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from flypylib_amd import _capi, fplmodels, multi_gpu, program, synth
from oracle import cnn_oracle, infer_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# translation units that hold split-half kernels: (source, extra compiler flags)
SPLIT_UNITS = [('vgg_split.hip', []), ('conv_mfma.hip', ['-DFPL_F16=1', '-DFPL_SPLIT=1'])]


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'fplhip.h')).read()
    declared = set(re.findall(r'\b(fpl_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), 'libfplhip.so does not export %s' % name
    assert declared == set(_capi.SIGNATURES), \
        'ctypes binding and header disagree: %s' % (
            declared ^ set(_capi.SIGNATURES))
    assert _capi.load_library().fpl_abi_version() == _capi.ABI_VERSION


def test_library_exports_nothing_but_the_declared_c_abi():
    """built with -fvisibility=hidden and a version script (csrc/build.py): the dynamic symbol
    table of libfplhip.so holds the header's entry points and no C++ internals"""
    import shutil
    import subprocess
    if not shutil.which('nm'):
        pytest.skip('nm not available')
    hdr = open(os.path.join(ROOT, 'include', 'fplhip.h')).read()
    declared = set(re.findall(r'\b(fpl_[a-z0-9_]+)\s*\(', hdr))
    out = subprocess.run(['nm', '-D', '--defined-only', _capi.LIB_PATH], check=True,
                         stdout=subprocess.PIPE, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared, (sorted(exported - declared)[:5], sorted(declared - exported)[:5])


def test_fpl_op_struct_layout_matches_header():
    # int32 x8, int64 x3, int32 x6 -> 80 bytes with natural alignment
    assert ctypes.sizeof(_capi.fpl_op) == 80
    assert _capi.fpl_op.w_off.offset == 32 and _capi.fpl_op.p.offset == 56


def test_ctx_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_capi.FplHipError):
        _capi.Context(0)


def test_model_contracts_match_reference_table():
    # (factory, rf_info, infer_sz, trainable params) - SURVEY 7/8a
    cases = [(fplmodels.vgg_like, (18, 7, 4), 102, 145105),
             (fplmodels.unet_like2, (24, 9, 1), 100, 623488),
             (fplmodels.baseline_model, (18, 7, 4), 102, None),
             (fplmodels.vgg_like2, (24, 10, 4), 100, None),
             (fplmodels.resnet_like, (18, 7, 4), 102, None),
             (fplmodels.unet_like, (18, 6, 1), 102, None),
             (fplmodels.unet_like3, (32, 13, 1), 100, None),
             (fplmodels.unet_like4, (40, 17, 1), 100, None),
             (fplmodels.unet_like4b, (40, 17, 1), 100, None),
             (fplmodels.unet_like_vol, (62, 6, 1), 102, None)]
    for f, rf, isz, ntrain in cases:
        g, rf_info, infer_sz, compile_args = f()
        assert rf_info == rf and infer_sz == isz, f.__name__
        if ntrain is not None:
            assert g.count_trainable() == ntrain
        gi = f(isz)[0]
        out = gi.output.size
        # fully-convolutional identity: out*stride == infer_sz - 2*offset
        assert all(o * rf[2] == isz - 2 * rf[1] for o in out), f.__name__
        # at the receptive-field size the net yields exactly one coarse output
        # (vgg family) or rf-2*off voxels (unets)
        gr = f(rf[0])[0]
        assert all(o * rf[2] == max(rf[0] - 2 * rf[1], rf[2])
                   for o in gr.output.size), f.__name__
    assert fplmodels.vgg_like()[3] is None
    ca = fplmodels.unet_like2()[3]
    assert ca['loss'] == fplmodels.masked_focal_loss and ca['optimizer'] == 'adam'


def test_vgg_layer_shapes_follow_survey_table():
    g = fplmodels.vgg_like(102)[0]
    convs = [n for n in g.nodes if n.kind in ('conv', 'pool')]
    sizes = [n.size[0] for n in convs]
    assert sizes == [100, 100, 50, 48, 48, 24, 22, 22, 22, 22]
    ch = [n.channels for n in convs]
    assert ch == [48, 48, 48, 48, 48, 48, 48, 96, 96, 1]
    g = fplmodels.unet_like2(100)[0]
    sizes = [n.size[0] for n in g.nodes if n.kind == 'conv']
    assert sizes == [98, 96, 46, 44, 22, 42, 42, 82, 82, 82]


def test_unet_rejects_incompatible_input():
    with pytest.raises(ValueError):
        fplmodels.unet_like2(26)        # 26 is not 0 mod 4 -> concat mismatch


def test_lowering_folds_bn_relu_dropout():
    g = fplmodels.vgg_like(30)[0]
    g.randomize_bn(3)
    ops, arena, out_t, n_t = g.lower_inference()
    kinds = [o['kind'] for o in ops]
    assert kinds == [0, 0, 1, 0, 0, 1, 0, 0, 0, 0]
    assert [o['act'] for o in ops if o['kind'] == 0] == [1] * 7 + [2]
    assert out_t == ops[-1]['dst'] and n_t == len(ops) + 1
    # folded scale/shift reproduce BN inference arithmetic
    o = ops[0]
    gamma, beta, mean, var = (g.weights[i] for i in (1, 2, 3, 4))
    sc = arena[o['scale_off']:o['scale_off'] + 48]
    sh = arena[o['shift_off']:o['shift_off'] + 48]
    x = np.linspace(-2, 2, 48).astype(np.float32)
    ref = gamma * (x - mean) / np.sqrt(var + 1e-3) + beta
    assert np.allclose(sc * x + sh, ref, atol=1e-6)
    # kernel is stored [k^3*cin][cout] in Keras memory order
    k = arena[o['w_off']:o['w_off'] + 27 * 48].reshape(3, 3, 3, 1, 48)
    assert np.array_equal(k, g.weights[0])
    # final conv keeps its bias as the shift
    o = ops[-1]
    assert o['cout'] == 1 and o['k'] == 1 and o['cin'] == 96


def test_lowering_unet_and_resnet_graphs():
    ops, _, _, _ = fplmodels.unet_like2(28)[0].lower_inference()
    assert [o['kind'] for o in ops].count(program.OP_CONCAT) == 2
    assert [o['kind'] for o in ops].count(program.OP_CROP) == 1
    crop = [o for o in ops if o['kind'] == program.OP_CROP][0]
    assert tuple(crop['p']) == (6,) * 6
    ops, _, _, _ = fplmodels.resnet_like(30)[0].lower_inference()
    adds = [o for o in ops if o['kind'] == program.OP_ADD]
    assert len(adds) == 2 and all(o['act'] == program.ACT_RELU for o in adds)


def test_set_get_weights_roundtrip_and_validation():
    g = fplmodels.vgg_like()[0]
    w = g.get_weights()
    assert len(w) == 8 + 1 + 7 * 4      # 8 kernels, 1 bias, 7 BN x 4
    w2 = [a + 1 for a in w]
    g.set_weights(w2)
    assert all(np.array_equal(a, b) for a, b in zip(g.get_weights(), w2))
    with pytest.raises(ValueError):
        g.set_weights(w[:-1])
    bad = list(w)
    bad[0] = np.zeros((3, 3, 3, 1, 47), np.float32)
    with pytest.raises(ValueError):
        g.set_weights(bad)


def test_slab_partition_covers_lattice_once():
    for n_rows in (1, 5, 8, 12, 47):
        for parts in (1, 2, 3, 4, 8):
            sl = multi_gpu.slab_partition(n_rows, parts)
            assert len(sl) == parts and sl[0][0] == 0 and sl[-1][1] == n_rows
            assert all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            sizes = [e - b for b, e in sl]
            assert max(sizes) - min(sizes) <= 1
    # tile-row count equals the reference lattice's
    for dim in (50, 102, 103, 190, 520, 1024):
        locs, _, _ = infer_oracle.tile_lattice((dim, 102, 102), (102,) * 3,
                                               (7,) * 3)
        nz = len(np.unique(locs[0]))
        assert multi_gpu.n_tile_rows(dim, 102, 7) == nz
    # slab rows tile the volume exactly
    dim = 520
    n = multi_gpu.n_tile_rows(dim, 102, 7)
    rows = [multi_gpu.slab_rows(z, dim, 102, 7)
            for z in multi_gpu.slab_partition(n, 4)]
    assert rows[0][0] == 0 and rows[-1][1] == dim
    assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))


def test_oracle_graph_interpreter_agrees_with_handwritten_forwards():
    rng = np.random.default_rng(0)
    g = fplmodels.vgg_like(22)[0]
    g.randomize_bn(1)
    x = rng.standard_normal((2, 22, 22, 22, 1)).astype(np.float32)
    a = cnn_oracle.vgg_like_forward(x, g.weights)
    b = cnn_oracle.graph_forward(g, x)
    assert a.shape == (2, 2, 2, 2, 1) and np.allclose(a, b, atol=1e-6)
    g = fplmodels.unet_like2(28)[0]
    g.randomize_bn(2)
    x = rng.standard_normal((1, 28, 28, 28, 1)).astype(np.float32)
    a = cnn_oracle.unet_like2_forward(x, g.weights)
    b = cnn_oracle.graph_forward(g, x)
    assert a.shape == (1, 10, 10, 10, 1) and np.allclose(a, b, atol=1e-6)


def test_keras_h5_converter_walks_layers_in_get_weights_order(tmp_path):
    """tools/keras_h5_to_npz.py on a stand-in for the h5py tree of a Keras file (h5py is
    absent here): arrays come out in `layer_names` x `weight_names` order and load into
    the vgg_like graph"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        'keras_h5_to_npz', os.path.join(os.path.dirname(os.path.dirname(__file__)), 'tools', 'keras_h5_to_npz.py'))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)

    class Node(dict):
        def __init__(self, items=(), attrs=None):
            super().__init__(items)
            self.attrs = attrs or {}

    graph = fplmodels.vgg_like()[0]
    synth.synthetic_weights(graph, 5)
    ws = graph.get_weights()
    # one Keras layer per 1 (conv) / 4 (BN) / 2 (conv + bias) arrays, names as Keras writes them
    layers, names, i, k = {}, [], 0, 0
    while i < len(ws):
        n = 4 if ws[i].ndim == 1 and i + 3 < len(ws) and all(w.ndim == 1 for w in ws[i:i + 4]) else \
            (2 if ws[i].ndim == 5 and i + 1 < len(ws) and ws[i + 1].ndim == 1 and ws[i + 1].shape[0] == ws[i].shape[-1] == 1 else 1)
        lname = 'layer_%d' % k
        wn = ['%s/w%d:0' % (lname, j) for j in range(n)]
        layers[lname] = Node({w: ws[i + j] for j, w in enumerate(wn)},
                             {'weight_names': [w.encode() for w in wn]})
        names.append(lname.encode())
        i += n
        k += 1
    root = Node({'model_weights': Node(layers, {'layer_names': names})})
    got = conv.keras_weight_list(root)
    assert len(got) == len(ws) and all(np.array_equal(a, b) for (_, a), b in zip(got, ws))
    conv.check_against(got, 'vgg_like')
    np.savez(str(tmp_path / 'w.npz'), *[a for _, a in got])
    g2 = fplmodels.vgg_like()[0]
    g2.load(str(tmp_path / 'w.npz'))
    assert all(np.array_equal(a, b) for a, b in zip(g2.get_weights(), ws))
    with pytest.raises(SystemExit):
        conv.check_against(got[:-1], 'vgg_like')


def test_voxel2obj_smoothing_compiles_without_fp64_fma(tmp_path):
    """scipy's Gaussian filter rounds every product and every sum separately (C on x86-64);
    hipcc would fuse them into v_fma_f64 by default.  Compile v2o.hip to device assembly
    with the build's own flags and look: the smoothing kernels must hold v_mul_f64 and
    v_add_f64 and no fp64 FMA anywhere in the file."""
    import shutil
    import subprocess
    from flypylib_amd.csrc import build
    if not shutil.which(build.HIPCC) and not os.path.exists(build.HIPCC):
        pytest.skip('hipcc not available')
    asm = tmp_path / 'v2o.s'
    flags = [f for f in build.CXXFLAGS if f != '-fPIC']
    subprocess.run([build.HIPCC] + flags + ['-I' + os.path.join(build.ROOT, 'include'), '-S',
                                            '--cuda-device-only', '-o', str(asm),
                                            os.path.join(build.HERE, 'v2o.hip')],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    text = asm.read_text()
    assert not re.search(r'v_fma(c|ak|mk)?_f64', text), 'fp64 FMA in v2o.hip device code'
    # per smoothing kernel: the separately rounded product and sum are there
    for kernel in ('gauss_z_ring', 'gauss_yx_fused', 'gauss_pass_win', 'gauss_x_lds', 'gauss_pass'):
        bodies = re.findall(r'^_ZN[^\n]*%sI[^\n]*:[^\n]*\n(.*?)s_endpgm' % kernel, text, re.S | re.M)
        assert bodies, kernel
        for body in bodies:
            assert 'v_mul_f64' in body and 'v_add_f64' in body, kernel


def test_headline_kernels_compile_without_scratch_spills():
    """the split-half kernels of the headline path as hipcc builds them (the build's own flags;
    metadata of the device assembly): the persistent stem (both volume types), mid and tail kernels
    and vgg_like2's tail keep every register in the register file - a block-invariant load
    hoisted out of the persistent loop, or the head chain run four sub-steps abreast, showed up
    here as 100 - 200 spilled registers in round 4.  The U-Net's split 1x1x1 convolutions likewise
    (128 -> 128 was 512 registers and 96 spilled ones with its LDS fragment reads hoisted)."""
    import shutil
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from flypylib_amd.csrc import build
    if not shutil.which(build.HIPCC) and not os.path.exists(build.HIPCC):
        pytest.skip('hipcc not available')
    import kernel_resources
    res = kernel_resources.kernel_resources('vgg_split.hip')
    seen = set()
    for name, v in res.items():
        for key, limit in (('vggs_mid_pool', 0), ('vggs_c5_tail', 0), ('vggs_stem_poolIh', 0),
                           ('vggs_stem_poolIf', 0), ('vggs2_conv3', 0)):
            if key in name:
                seen.add(key)
                assert v['vgpr_spill_count'] <= limit, (name, v)
                assert v['vgpr_count'] <= 256
    assert seen == {'vggs_mid_pool', 'vggs_c5_tail', 'vggs_stem_poolIh', 'vggs_stem_poolIf', 'vggs2_conv3'}
    res = kernel_resources.kernel_resources('conv_mfma.hip', ['-DFPL_F16=1', '-DFPL_SPLIT=1'])
    conv1 = {n: v for n, v in res.items() if 'conv1_f16s' in n}
    assert len(conv1) >= 3
    for name, v in conv1.items():
        assert v['vgpr_spill_count'] == 0 and v['vgpr_count'] <= 256, (name, v)
    # the U-Net's all-LDS kernels (round 5; csrc/unet_split_lds.h): stem + pool, 32->64, 64->64 + pool,
    # 192->64 (zy form, 14 rows), head (zy form) and its edge strip - two waves per SIMD, no scratch
    u3 = {n: v for n, v in res.items() if 'u3conv_f16s' in n}
    assert len(u3) >= 6, sorted(u3)
    for name, v in u3.items():
        # (one instance reserves 36 B of private segment for a stack object its body never touches:
        # no scratch_ instruction in any of them)
        assert v['vgpr_spill_count'] == 0 and v['private_segment_fixed_size'] <= 64 and v['vgpr_count'] <= 256, (name, v)
    text = kernel_resources.device_asm('conv_mfma.hip', ['-DFPL_F16=1', '-DFPL_SPLIT=1'])
    for body in re.findall(r'^_ZN[^\n]*u3conv_f16s[^\n]*:[^\n]*\n(.*?)s_endpgm', text, re.S | re.M):
        assert 'scratch_' not in body


def test_split_lo_halves_never_land_on_an_mfma_destination():
    """csrc/mfma_util.h::split_lo_pk writes the lo halves by inline asm (two v_fma_mix*_f16), which
    hipcc's hazard recognizer does not look into: its destination register must be one an
    ordinary VALU instruction wrote last - never part of the destination tuple of an MFMA that
    may still be in flight (round 4: run-to-run differences of 1e-5).  Checked on the device
    assembly of every translation unit that splits; the checker itself is exercised first."""
    import shutil
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from flypylib_amd.csrc import build
    if not shutil.which(build.HIPCC) and not os.path.exists(build.HIPCC):
        pytest.skip('hipcc not available')
    import kernel_resources as kr
    bad_asm = ('\tv_mfma_f32_16x16x32_f16 v[4:7], v[10:13], v[14:17], v[4:7]\n'
               '\tv_cvt_pk_f16_f32 v20, v4, v8\n'
               '\tv_fma_mixlo_f16 v5, v20, -1.0, v5 op_sel_hi:[1,0,0]\n')
    good_asm = ('\tv_mfma_f32_16x16x32_f16 v[4:7], v[10:13], v[14:17], v[4:7]\n'
                '\tv_max_i32_e32 v5, 0, v5\n'
                '\tds_write_b128 v5, v[30:33]\n'
                '\tv_fma_mixlo_f16 v5, v20, -1.0, v5 op_sel_hi:[1,0,0]\n'
                '\tv_fma_mixhi_f16 v5, v20, -1.0, v9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n')
    assert len(kr.fma_mix_mfma_overlaps(bad_asm)) == 1 and not kr.fma_mix_mfma_overlaps(good_asm)
    for src, extra in SPLIT_UNITS:
        text = kr.device_asm(src, extra)
        assert len(re.findall(r'v_fma_mixlo_f16', text)) > 50, src
        bad = kr.fma_mix_mfma_overlaps(text)
        assert not bad, (src, bad[:3])


def test_host_pool_recycles_blocks_only_when_every_view_is_dead():
    """_capi.host_empty (the arrays FplNetwork.infer returns): a block goes back to the pool when the
    LAST array built on it dies - a caller that keeps a slice keeps the block"""
    import gc
    from flypylib_amd import _capi
    shape = (64, 256, 256)                       # 16 MiB of float32: above the pool's 8 MiB floor
    a = _capi.host_empty(shape)
    assert a.shape == shape and a.dtype == np.float32 and a.flags['C_CONTIGUOUS'] and a.flags['WRITEABLE']
    addr = a.ctypes.data
    a[...] = 3.0
    keep = a[5:7, ::2]
    del a
    gc.collect()
    b = _capi.host_empty(shape)
    assert b.ctypes.data != addr and float(keep.mean()) == 3.0     # the slice still owns the first block
    del keep
    gc.collect()
    c = _capi.host_empty(shape)
    assert c.ctypes.data == addr                                  # ... and now it is handed out again
    small = _capi.host_empty((4, 4, 4))                           # small arrays: plain numpy
    assert small.base is None
