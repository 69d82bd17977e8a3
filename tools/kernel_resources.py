"""Registers, spills, scratch and LDS of every kernel of a translation unit, read from the
metadata hipcc writes into the device assembly (the build's own flags).

    python tools/kernel_resources.py vgg_split.hip [-DFPL_F16=1 ...] [--keep out.s]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


_ASM = {}          # (source, flags) -> device assembly text (one compile per process)


def device_asm(src, extra=(), keep=None):
    """the gfx950 device assembly of a translation unit, built with the library's own flags"""
    from flypylib_amd.csrc import build
    key = (src, tuple(extra))
    if key in _ASM and not keep:
        return _ASM[key]
    flags = [f for f in build.CXXFLAGS if f != '-fPIC']
    out = keep or os.path.join(tempfile.mkdtemp(), 'k.s')
    subprocess.run([build.HIPCC] + flags + list(extra) +
                   ['-I' + os.path.join(build.ROOT, 'include'), '-S', '--cuda-device-only', '-o', out,
                    os.path.join(build.HERE, src)], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.STDOUT)
    _ASM[key] = open(out).read()
    return _ASM[key]


_NO_VDST = ('ds_write', 'ds_store', 'global_store', 'buffer_store', 'scratch_store', 'flat_store',
            'v_cmp', 'v_cmpx', 'v_readfirstlane', 'v_readlane', 'global_atomic', 'buffer_atomic',
            'v_nop', 'global_load_lds', 'buffer_load')


def _vdst(ins):
    """VGPR numbers an instruction line writes (first operand of VALU / load forms), is-MFMA"""
    m = re.match(r'\s+(\w+)\s+(v\[(\d+):(\d+)\]|v(\d+)(?![\d\[])|a\[(\d+):(\d+)\]|a(\d+)(?![\d\[]))', ins)
    if not m:
        return None, False
    op = m.group(1)
    if op.startswith('s_') or any(op.startswith(p) for p in _NO_VDST):
        return None, False
    if m.group(2).startswith('a'):
        return set(), op.startswith('v_mfma')          # AGPR destination: no VGPR written
    if m.group(3) is not None:
        regs = set(range(int(m.group(3)), int(m.group(4)) + 1))
    else:
        regs = {int(m.group(5))}
    return regs, op.startswith('v_mfma')


def fma_mix_mfma_overlaps(text, window=64):
    """split_pk's rule (csrc/mfma_util.h), checked on the device assembly: the destination of a
    v_fma_mix{lo,hi}_f16 (written by inline asm, which the compiler's hazard recognizer does not
    see) must have been written LAST by an ordinary instruction, never by an MFMA that may still
    be in flight.  Returns the offending (line number, fma_mix line, mfma line) triples: for every
    fma_mix destination the nearest earlier writer within `window` instructions is looked up."""
    lines = text.split('\n')
    ins = [(i, l) for i, l in enumerate(lines) if re.match(r'\s+[a-z]\w+\s', l) and not l.lstrip().startswith('.')]
    bad = []
    for k, (i, l) in enumerate(ins):
        m = re.match(r'\s+v_fma_mix(lo|hi)_f16\s+v(\d+)\b', l)
        if not m:
            continue
        reg = int(m.group(2))
        for j in range(k - 1, max(-1, k - 1 - window), -1):
            regs, is_mfma = _vdst(ins[j][1])
            if regs and reg in regs:
                if is_mfma:
                    bad.append((i + 1, l.strip(), ins[j][1].strip()))
                break
    return bad


def kernel_resources(src, extra=(), keep=None):
    text = device_asm(src, extra, keep)
    res = {}
    for blk in re.split(r'\n  - \.agpr_count:', text)[1:]:
        blk = '.agpr_count:' + blk
        get = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk)
        name = get('name').group(1)
        res[name] = {k: int(get(k).group(1)) for k in
                     ('vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count',
                      'private_segment_fixed_size', 'group_segment_fixed_size')}
    return res


def demangle(names):
    try:
        r = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names), text=True,
                           stdout=subprocess.PIPE)
        return r.stdout.split('\n')
    except OSError:
        return names


if __name__ == '__main__':
    args = sys.argv[1:]
    keep = None
    if '--keep' in args:
        i = args.index('--keep')
        keep = args[i + 1]
        del args[i:i + 2]
    r = kernel_resources(args[0], args[1:], keep)
    names = list(r)
    for n, d in zip(names, demangle(names)):
        v = r[n]
        print('%-90s vgpr %3d agpr %3d sgpr %3d spill %3d scratch %4d lds %6d' % (
            d[:90], v['vgpr_count'], v['agpr_count'], v['sgpr_count'], v['vgpr_spill_count'],
            v['private_segment_fixed_size'], v['group_segment_fixed_size']))
