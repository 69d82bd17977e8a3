"""Build libfplhip.so for gfx950 (MI355X) with hipcc, in-tree.

    python -m flypylib_amd.csrc.build [--force] [-j N]

Each .hip translation unit is compiled to build/<name>.o and linked into
flypylib_amd/lib/libfplhip.so.  An object is reused only when the SHA-256 of what it
was built from (its source, every header, the compiler flags) matches the stamp next
to it - modification times say nothing on a tree that ships objects, and the driver's
build check must compile what it ships.  hipcc cross-compiles without a GPU, so this
runs in the CPU-only build container.
"""
import argparse
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OBJ_DIR = os.path.join(HERE, 'build')
LIB_DIR = os.path.join(os.path.dirname(HERE), 'lib')
LIB = os.path.join(LIB_DIR, 'libfplhip.so')
ARCH = 'gfx950'
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
CXXFLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC',
            '-fvisibility=hidden', '-fvisibility-inlines-hidden',
            '-ffp-contract=on', '-Wall', '-Wno-unused-function',
            '-Wno-unused-but-set-variable', '-Wno-unused-variable',
            '-Wno-unused-value', '-Wno-unused-result']
# throw-away timing experiments only (e.g. FPL_EXTRA_CXXFLAGS=-DFPL_EXP_NOMATH=1)
CXXFLAGS += os.environ.get('FPL_EXTRA_CXXFLAGS', '').split()


# translation units built a second time with -DFPL_F16 (IEEE-half operands instead of
# bfloat16; mfma_util.h)
DUAL_PRECISION = ('vgg_fused.hip', 'conv_mfma.hip')
# ... and a third time on SPLIT IEEE-half operands (hi + lo per value; -DFPL_SPLIT):
# the U-Net executor (vgg_like's split kernels are a file of their own, vgg_split.hip)
SPLIT_BUILD = ('conv_mfma.hip',)


def _sources():
    """(source file, object stem, extra flags)"""
    out = []
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.hip'):
            out.append((f, f[:-4], []))
            if f in DUAL_PRECISION:
                out.append((f, f[:-4] + '_f16', ['-DFPL_F16=1']))
            if f in SPLIT_BUILD:
                out.append((f, f[:-4] + '_f16s', ['-DFPL_F16=1', '-DFPL_SPLIT=1']))
    return out


def _headers_digest():
    """one digest over every header a translation unit may include"""
    hs = sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h'))
    hs.append(os.path.join(ROOT, 'include', 'fplhip.h'))
    h = hashlib.sha256()
    for p in hs:
        h.update(os.path.basename(p).encode() + b'\0')
        h.update(open(p, 'rb').read())
    return h.hexdigest()


def _compile(unit, force, hdr_digest):
    src, stem, extra = unit
    obj = os.path.join(OBJ_DIR, stem + '.o')
    stamp = obj + '.sha256'
    sp = os.path.join(HERE, src)
    h = hashlib.sha256()
    h.update(open(sp, 'rb').read())
    h.update(hdr_digest.encode())
    h.update(' '.join(CXXFLAGS + extra).encode())
    want = h.hexdigest()
    if (not force and os.path.exists(obj) and os.path.exists(stamp)
            and open(stamp).read().strip() == want):
        return obj, None
    cmd = [HIPCC] + CXXFLAGS + extra + ['-c', sp, '-o', obj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True)
    if r.returncode != 0:
        if os.path.exists(stamp):
            os.remove(stamp)
        raise RuntimeError('hipcc failed for %s:\n%s' % (stem, r.stdout))
    with open(stamp, 'w') as f:
        f.write(want + '\n')
    return obj, r.stdout


def build(force=False, jobs=4, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    hdr_m = _headers_digest()
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(lambda s: _compile(s, force, hdr_m), srcs))
    objs = [o for o, _ in results]
    rebuilt = [u[1] for u, (_, out) in zip(srcs, results) if out is not None]
    for u, (_, out) in zip(srcs, results):
        if out and verbose and out.strip():
            print('[%s]\n%s' % (u[1], out.strip()))
    # the library carries the digest of the objects it was linked from
    lh = hashlib.sha256()
    for o in objs:
        lh.update(open(o + '.sha256').read().encode())
    lib_want = lh.hexdigest()
    lib_stamp = LIB + '.sha256'
    if (rebuilt or not os.path.exists(LIB) or not os.path.exists(lib_stamp)
            or open(lib_stamp).read().strip() != lib_want):
        # the export list = the C entry points include/fplhip.h declares, nothing else (the
        # C++ internals are hidden by -fvisibility=hidden, libstdc++'s template instances -
        # default visibility by their own headers - by this version script)
        vs = os.path.join(OBJ_DIR, 'exports.map')
        with open(vs, 'w') as f:
            f.write('{ global: fpl_*; local: *; };\n')
        cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-Wl,--version-script=' + vs,
               '-o', LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stdout)
        with open(lib_stamp, 'w') as f:
            f.write(lib_want + '\n')
        if verbose:
            print('linked %s (%d objects, rebuilt: %s)' % (
                os.path.relpath(LIB, ROOT), len(objs), ', '.join(rebuilt) or '-'))
    elif verbose:
        print('%s is up to date' % os.path.relpath(LIB, ROOT))
    return LIB


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--force', action='store_true')
    ap.add_argument('-j', type=int, default=4)
    a = ap.parse_args()
    try:
        build(a.force, a.j)
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
