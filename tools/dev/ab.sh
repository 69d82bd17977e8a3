#!/bin/bash
# A/B of prebuilt library variants (tools/dev/libs/*.so) on bench.py:
#   tools/dev/ab.sh PRECISION SIZE variant...      (prints Mvox/s and per-kernel ms)
prec=$1; size=$2; shift 2
cp flypylib_amd/lib/libfplhip.so /tmp/orig.so
for v in "$@"; do
  cp tools/dev/libs/$v.so flypylib_amd/lib/libfplhip.so
  python bench.py --precision $prec --size $size --steps 5 --warmup 2 --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_total']
print('$v', $size, d['value'], {n: round(t / d['steps'], 3) for n, t in k.items()})"
done
cp /tmp/orig.so flypylib_amd/lib/libfplhip.so
