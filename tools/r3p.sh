#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3p
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -q --tb=short > $O/tests.log 2>&1; rc=$?
tail -12 $O/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/bench_bf16.py --reps 8 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt || exit 1
