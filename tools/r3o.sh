#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3o
mkdir -p $O
cd $R
echo "[1] bf16 tests"
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -q --tb=short -k "wide or conv_forward or conv_backward" > $O/tests.log 2>&1; rc=$?
tail -8 $O/tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
echo "[2] layer bench"
timeout -k 10 300 python tools/bench_bf16.py --layers D.conv3,D.conv4 --modes fwd,dgrad --reps 5 2>&1 | grep -v amdgpu.ids | tee $O/lb_wide.txt || exit 1
bash tools/r3n.sh
