#!/usr/bin/env python3
"""Collect rocprofv3 PMC counters for the bench kernels (run ON the GPU box):

    python tools/profile_pmc.py --out gpurun_out/pmc_r01 [--size 512] [--precision bf16]

One rocprofv3 run per counter group (SQ has 8 slots, TCC 4; FETCH_SIZE and
WRITE_SIZE cannot share a pass - MI355X_MICROARCH.md, rocprofv3 PMC slots), each
with --kernel-trace only.  Writes <out>/summary.json and summary.md: per kernel,
mean counter value per dispatch.  HBM traffic follows the guide's gfx950
correction: bytes = 2 * FETCH_SIZE * 1024 (wide coalesced reads are tallied at
half) + WRITE_SIZE * 1024.
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GROUPS = {
    'sq_time': ['SQ_WAVES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY',
                'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY',
                'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'],
    'sq_inst': ['SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS',
                'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAIT_INST_LDS',
                'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VALU'],
    'sq_mem': ['SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SALU',
               'SQ_INSTS_SMEM', 'SQ_ACTIVE_INST_VMEM', 'SQ_INST_LEVEL_VMEM',
               'SQ_INST_LEVEL_LDS', 'SQ_LDS_UNALIGNED_STALL'],
    'fetch': ['FETCH_SIZE'],
    'rdreq': ['TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_32B_sum'],
    'l2hit': ['TCC_HIT_sum', 'TCC_MISS_sum'],
    'write': ['WRITE_SIZE', 'TCC_EA0_WRREQ_sum'],
    # vector-memory path of a CU: L1 (TCP) accesses / misses to L2, address unit busy
    'tcp': ['TCP_TOTAL_CACHE_ACCESSES_sum', 'TCP_TCC_READ_REQ_sum', 'TCP_PENDING_STALL_CYCLES_sum',
            'TCP_TCC_READ_REQ_LATENCY_sum'],
    'ta': ['TA_TA_BUSY_sum', 'TA_BUSY_avr', 'TD_TD_BUSY_sum', 'TCP_GATE_EN1_sum'],
}


def run_group(name, counters, out, bench_args, script='bench.py'):
    d = os.path.join(out, name)
    os.makedirs(d, exist_ok=True)
    cmd = ['rocprofv3', '--pmc'] + counters + [
        '--kernel-trace', '--output-format', 'csv', '-d', d, '--',
        sys.executable, os.path.join(ROOT, script)] + bench_args
    env = dict(os.environ, TMPDIR='/tmp')
    print('[pmc] group %s: %s' % (name, ' '.join(counters)), flush=True)
    try:
        r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        print('[pmc] group %s timed out' % name, flush=True)
        return {}
    open(os.path.join(d, 'run.log'), 'w').write(r.stdout)
    if r.returncode != 0:
        print('group %s failed (rc %d); see %s/run.log' % (name, r.returncode, d))
        return {}
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'),
                       recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get('Kernel_Name', '')
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
    return acc


def short(name):
    """kernel name without namespace, return type and argument list (template
    arguments kept: they tell the variants of one kernel apart)"""
    n = name.replace('(anonymous namespace)::', '').replace('void ', '')
    depth, out = 0, []
    for ch in n:
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            break
        out.append(ch)
    return ''.join(out).strip()[:80]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--precision', default='bf16')
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--groups', default=','.join(GROUPS))
    ap.add_argument('--target', default='bench', choices=['bench', 'unet', 'v2o', 'graphs'],
                    help="bench = bench.py (vgg_like); unet = tools/bench_configs.py "
                         "--what unet --unet-size SIZE; v2o = tools/bench_v2o.py --sub SIZE")
    a = ap.parse_args()
    a.out = os.path.abspath(a.out)
    os.makedirs(a.out, exist_ok=True)
    bench_args = ['--size', str(a.size), '--precision', a.precision, '--steps',
                  str(a.steps), '--warmup', '1', '--no-cpu-baseline', '--no-legs']
    script = 'bench.py'
    if a.target == 'unet':
        script = os.path.join('tools', 'bench_configs.py')
        bench_args = ['--what', 'unet', '--unet-size', str(a.size)]
    if a.target == 'graphs':           # the graph executor's four factories ('auto', f16, f32 in turn)
        script = os.path.join('tools', 'bench_configs.py')
        bench_args = ['--what', 'graphs']
    if a.target == 'v2o':
        script = os.path.join('tools', 'bench_v2o.py')
        bench_args = ['--sub', str(a.size), '--reps', str(a.steps)]
    summary = defaultdict(dict)
    for gname in a.groups.split(','):
        acc = run_group(gname, GROUPS[gname], a.out, bench_args, script)
        for k, cs in acc.items():
            for c, vals in cs.items():
                summary[short(k)][c] = sum(vals) / len(vals)
                summary[short(k)]['dispatches'] = len(vals)
    for k, cs in summary.items():
        if 'FETCH_SIZE' in cs or 'WRITE_SIZE' in cs:
            cs['hbm_bytes_per_dispatch'] = (2 * cs.get('FETCH_SIZE', 0) * 1024
                                            + cs.get('WRITE_SIZE', 0) * 1024)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and cs.get('SQ_BUSY_CYCLES'):
            cs['mfma_busy_frac_of_sq_busy'] = (cs['SQ_VALU_MFMA_BUSY_CYCLES']
                                               / cs['SQ_BUSY_CYCLES'])
    json.dump({'bench_args': bench_args, 'kernels': summary},
              open(os.path.join(a.out, 'summary.json'), 'w'), indent=1)
    with open(os.path.join(a.out, 'summary.md'), 'w') as f:
        f.write('# rocprofv3 PMC summary (mean per dispatch)\n\n`%s %s`\n\n'
                % (script, ' '.join(bench_args)))
        for k, cs in sorted(summary.items()):
            f.write('## %s\n\n| counter | value |\n|---|---|\n' % k)
            for c, v in sorted(cs.items()):
                f.write('| %s | %.6g |\n' % (c, v))
            f.write('\n')
    print(open(os.path.join(a.out, 'summary.md')).read())


if __name__ == '__main__':
    main()
