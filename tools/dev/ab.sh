#!/bin/bash
# A/B of prebuilt library variants (tools/dev/libs/*.so) on bench.py at two sizes
cp flypylib_amd/lib/libfplhip.so /tmp/orig.so
for v in "$@"; do
  cp tools/dev/libs/$v.so flypylib_amd/lib/libfplhip.so
  for size in 582 1024; do
    python bench.py --size $size --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', $size, d['value'], d['roofline']['kernel_ms_total'])"
  done
done
cp /tmp/orig.so flypylib_amd/lib/libfplhip.so
