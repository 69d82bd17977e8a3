#!/bin/bash
# rocm-smi power / sclk while a bare MFMA loop runs for ~5 s (random operands)
for shape in 16 32; do
  (cd tools/micro && ./mfma_rate random $shape 250) &
  pid=$!
  sleep 2
  for i in 1 2 3 4; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed "s/=\+//g" | tr '\n' ' '; echo
    sleep 0.5
  done
  wait $pid
done
