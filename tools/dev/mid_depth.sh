#!/bin/bash
# A/B of the mid kernel's weight prefetch depth (experiment)
mkdir -p gpurun_out/wg
for d in 0 2 3 4 6 8; do
  if [ $d = 0 ]; then unset FPL_MID_WGLOBAL; else export FPL_MID_WGLOBAL=$d; fi
  python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('D=$d', d['value'], r['avg_launch_ms'])"
done
