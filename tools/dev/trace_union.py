"""GPU busy share of a multi-stream run from a rocprofv3 kernel trace: the union of the kernels' [start, end)
intervals against the span they cover.  (Per-kernel durations of concurrent streams overlap - their sum says
nothing about how busy the GPU was; HIP events on concurrent streams time the waits for one another.)

    rocprofv3 --kernel-trace --output-format csv -d DIR -o roi -- python3 tools/bench_configs.py --what roi --skip-oracle
    python tools/dev/trace_union.py DIR [--last-span-ms MS] > profiles/rNN_roi1536_trace_union.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
last_ms = float(sys.argv[sys.argv.index('--last-span-ms') + 1]) if '--last-span-ms' in sys.argv else None
files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
assert files, 'no kernel trace under %s' % d
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
t_end = max(e for _, e, _ in rows)
if last_ms is not None:           # only the timed run: the last MS milliseconds of the trace
    rows = [r for r in rows if r[0] >= t_end - int(last_ms * 1e6)]
t0, t1 = rows[0][0], max(e for _, e, _ in rows)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = defaultdict(lambda: [0, 0])
for s, e, k in rows:
    k = k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:48]
    tot[k][0] += 1
    tot[k][1] += e - s
span = t1 - t0
print('kernels %d, span %.1f ms, union of kernel intervals %.1f ms = %.1f %% of the span; sum of durations %.1f ms '
      '(%.2f x the span: concurrent streams)' % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span,
                                                sum(v[1] for v in tot.values()) / 1e6,
                                                sum(v[1] for v in tot.values()) / span))
for k, (n, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
    print('  %-50s %5d launches %9.1f ms' % (k, n, ns / 1e6))


# ---- who holds the GPU: classes of kernels, their unions and the overlap between them
def klass(k):
    if 'vggs_' in k or 'vgg_' in k or 'conv3' in k or 'u3conv' in k or 'clear_shell' in k:
        return 'inference'
    if 'synth_u8' in k or 'crop_u8' in k or 'hist_u8' in k:
        return 'source'
    if 'rocclr' in k:
        return 'copy/fill'
    return 'voxel2obj'


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append([s, e])
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


by = defaultdict(list)
for s, e, k in rows:
    by[klass(k)].append((s, e))
un = {c: union(v) for c, v in by.items()}
print('\nclass                launches   sum of durations   union (some kernel of the class on the GPU)')
for c in sorted(un, key=lambda c: -length(un[c])):
    print('  %-18s %8d %12.1f ms %12.1f ms = %5.1f %% of the span' %
          (c, len(by[c]), sum(e - s for s, e in by[c]) / 1e6, length(un[c]) / 1e6, 100.0 * length(un[c]) / span))
inf = un.get('inference', [])
others = union([iv for c, v in by.items() if c != 'inference' for iv in v])
both = intersect(inf, others)
print('inference kernels on the GPU: %.1f ms; of that with another class beside them: %.1f ms; '
      'no inference kernel on the GPU: %.1f ms' %
      (length(inf) / 1e6, length(both) / 1e6, (span - length(inf)) / 1e6))
# depth of the inference queue: how many inference kernels are in flight (dispatched, not finished)
ev = sorted([(s, 1) for s, _ in by.get('inference', [])] + [(e, -1) for _, e in by.get('inference', [])])
depth_t = defaultdict(int)
d, last = 0, None
for t, dv in ev:
    if last is not None:
        depth_t[d] += t - last
    d += dv
    last = t
print('inference kernels in flight: ' + ', '.join('%d: %.1f ms' % (k, v / 1e6) for k, v in sorted(depth_t.items()) if k))
