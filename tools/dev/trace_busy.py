#!/usr/bin/env python3
"""GPU busy time (union of kernel intervals) of a rocprofv3 --kernel-trace CSV, overall and
inside the longest dense burst - what a multi-stream pipeline leaves idle between kernels.
    python tools/dev/trace_busy.py <kernel_trace.csv> [gap_us=2000]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 2e6
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# split into bursts separated by > gap ns of idle
bursts, cur = [], [iv[0]]
end = iv[0][1]
for s, e, n in iv[1:]:
    if s - end > gap:
        bursts.append(cur)
        cur = []
    cur.append((s, e, n))
    end = max(end, e)
bursts.append(cur)
best = max(bursts, key=lambda b: sum(e - s for s, e, _ in b))
for name, b in (('longest burst', best),):
    t0, t1 = b[0][0], max(e for _, e, _ in b)
    busy, pe = 0, t0
    for s, e, _ in b:
        if e > pe:
            busy += e - max(s, pe)
            pe = e
    by = {}
    for s, e, n in b:
        k = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0][:40]
        by[k] = by.get(k, 0) + (e - s)
    print('%s: %d kernels, span %.2f ms, busy (union) %.2f ms = %.1f %%, sum of kernel times %.2f ms'
          % (name, len(b), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0),
             sum(e - s for s, e, _ in b) / 1e6))
    for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:14]:
        print('   %-42s %8.2f ms' % (k, v / 1e6))
