#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3q
mkdir -p $O
cd $R
for d in 0 20 16 6; do
  echo "== MPGAN_DBG_HB=$d" | tee -a $O/clock.txt
  timeout -k 10 120 bash tools/clock_probe_bf16.sh D.conv3 fwd $d 2>&1 | tee -a $O/clock.txt
done
