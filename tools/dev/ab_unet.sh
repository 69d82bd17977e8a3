#!/bin/bash
# A/B of prebuilt library variants (tools/dev/libs/*.so) on the unet_like2 leg of tools/bench_configs.py:
#   tools/dev/ab_unet.sh variant...
cp flypylib_amd/lib/libfplhip.so /tmp/orig.so
for v in "$@"; do
  cp tools/dev/libs/$v.so flypylib_amd/lib/libfplhip.so
  python tools/bench_configs.py --what unet --out /tmp/u_$v.json > /tmp/u_$v.log 2>&1
  python - <<PY
import json
d=json.load(open('/tmp/u_$v.json'))
u=d.get('unet', d)
print('$v', json.dumps(u)[:1500])
PY
done
cp /tmp/orig.so flypylib_amd/lib/libfplhip.so
