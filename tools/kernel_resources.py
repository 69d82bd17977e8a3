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


def kernel_resources(src, extra=(), keep=None):
    from flypylib_amd.csrc import build
    flags = [f for f in build.CXXFLAGS if f != '-fPIC']
    out = keep or os.path.join(tempfile.mkdtemp(), 'k.s')
    subprocess.run([build.HIPCC] + flags + list(extra) +
                   ['-I' + os.path.join(build.ROOT, 'include'), '-S', '--cuda-device-only', '-o', out,
                    os.path.join(build.HERE, src)], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.STDOUT)
    text = open(out).read()
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
