#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3n
mkdir -p $O
cd $R
for d in 0 32 16 20 6 4; do
  echo "MPGAN_DBG_HB=$d" | tee -a $O/whatif2.txt
  MPGAN_DBG_HB=$d timeout -k 10 200 python tools/bench_bf16.py --layers D.conv2 --modes fwd,dgrad,wgrad --reps 5 2>&1 | grep -v amdgpu.ids | tee -a $O/whatif2.txt || exit 1
done
