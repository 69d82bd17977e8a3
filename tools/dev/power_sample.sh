#!/bin/bash
# sample rocm-smi (power, sclk) while the headline step runs in a loop (run ON the GPU box)
python bench.py --no-cpu-baseline --no-legs --steps 300 --warmup 5 > /tmp/b.json 2>/dev/null &
pid=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo
  sleep 0.7
done
wait $pid
python -c "import json; d=json.load(open('/tmp/b.json')); print(d['ms_per_step'], d['roofline']['kernel_ms_total'])"
