"""Instruction mix per kernel of a device assembly file (hipcc --cuda-device-only -S):
MFMA / VALU / LDS / vector-memory / scratch / scalar counts and the most frequent VALU
opcodes; optionally only the lines between two markers (a loop body).

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S x.hip -o x.s
    python tools/dev/isa_mix.py x.s [kernel-substring]
"""
import collections
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split('\n')
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
    for (i, name), (j, _) in zip(starts, starts[1:] + [(len(lines), '')]):
        if want not in name:
            continue
        end = next((k for k in range(i, j) if 's_endpgm' in lines[k]), j)
        cnt = collections.Counter()
        for l in lines[i:end + 1]:
            m = re.match(r'^\s+([a-z_0-9]+)\s', l)
            if m:
                cnt[m.group(1)] += 1
        groups = collections.Counter()
        for k, v in cnt.items():
            if k.startswith('v_mfma'):
                groups['mfma'] += v
            elif k.startswith('ds_'):
                groups['lds'] += v
            elif k.startswith('scratch_'):
                groups['scratch'] += v
            elif k.startswith('global_') or k.startswith('buffer_'):
                groups['vmem'] += v
            elif k.startswith('v_'):
                groups['valu'] += v
            elif k.startswith('s_'):
                groups['salu'] += v
        print(name, sum(cnt.values()), dict(groups))
        print('   VALU:', [(k, v) for k, v in cnt.most_common(40)
                           if k.startswith('v_') and not k.startswith('v_mfma')][:16])
        print('   other:', [(k, v) for k, v in cnt.most_common(60)
                            if not k.startswith('v_')][:14])


if __name__ == '__main__':
    main()
