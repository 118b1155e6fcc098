#!/bin/bash
# Development aid: engine clock and socket power (rocm-smi) while a bf16 layer runs back to back.
# usage: tools/clock_probe_bf16.sh <layer> <mode> [MPGAN_DBG_HB value]
export MPGAN_DBG_HB=${3:-0}
python tools/bench_bf16.py --layers "$1" --modes "$2" --reps 1200 > /tmp/clock_probe_bf16.txt 2>&1 &
BP=$!
sleep 4.0
for i in 1 2 3 4; do
  rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1
  rocm-smi --showpower 2>/dev/null | grep -i "power" | head -1
  sleep 0.25
done
wait $BP
grep "D\." /tmp/clock_probe_bf16.txt
