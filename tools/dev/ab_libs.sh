#!/bin/bash
# A/B of prebuilt library variants under tools/dev/libs (experiment helper)
mkdir -p gpurun_out/ab
cp flypylib_amd/lib/libfplhip.so /tmp/orig.so
for v in "$@"; do
  cp tools/dev/libs/$v.so flypylib_amd/lib/libfplhip.so
  python tools/bench_configs.py --what unet --unet-size 756 2>gpurun_out/ab/$v.err | head -1 > gpurun_out/ab/$v.json
  python -c "
import json
d=json.loads(open('gpurun_out/ab/$v.json').read()); print('$v', round(d['mvox_s']), d['kernels'])"
done
cp /tmp/orig.so flypylib_amd/lib/libfplhip.so
