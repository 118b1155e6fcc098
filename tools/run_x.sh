#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -q --tb=short 2>&1 | tail -4 || exit 1
for d in 0 6 32; do
  echo "MPGAN_DBG_HB=$d"
  MPGAN_DBG_HB=$d timeout -k 10 200 python tools/bench_bf16.py --layers D.conv2 --modes fwd,dgrad --reps 5 2>&1 | grep -v amdgpu.ids
done
