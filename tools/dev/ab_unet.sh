#!/bin/bash
# A/B of prebuilt library variants (tools/dev/libs/*.so) on the unet_like2 bench
cp flypylib_amd/lib/libfplhip.so /tmp/orig.so
for v in "$@"; do
  cp tools/dev/libs/$v.so flypylib_amd/lib/libfplhip.so
  python tools/bench_configs.py --what unet --out /tmp/u_$v.json > /tmp/u_$v.log 2>&1
  python - <<PY
import json
d = json.load(open('/tmp/u_$v.json'))
for k, v in d.items():
    if isinstance(v, dict) and 'seconds' in v:
        print('$v', k, round(v['seconds'] * 1e3, 2), 'ms', {a: b for a, b in v.get('kernels', {}).items() if 'stem' in a or 'head' in a})
PY
done
cp /tmp/orig.so flypylib_amd/lib/libfplhip.so
