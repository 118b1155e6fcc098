#!/bin/bash
# Development aid: sample the GPU's engine clock (rocm-smi) while a conv layer runs back to back.
# usage: tools/clock_probe.sh <layer> <mode>
python tools/bench_conv.py --layers "$1" --modes "$2" --pro --reps 1500 > /tmp/clock_probe_bench.txt 2>&1 &
BP=$!
sleep 2.0
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2
  rocm-smi --showpower 2>/dev/null | grep -i "power" | head -1
  sleep 0.3
done
wait $BP
grep "D\.\|G\." /tmp/clock_probe_bench.txt
