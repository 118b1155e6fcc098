#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3m
mkdir -p $O
cd $R
echo "[1] bf16 tests"
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -q --tb=short -k "wide or conv_forward or conv_backward" > $O/tests.log 2>&1; rc=$?
tail -15 $O/tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
echo "[2] layer bench, wide on / off"
timeout -k 10 300 python tools/bench_bf16.py --layers D.conv3,D.conv4 --modes fwd,dgrad --reps 5 > $O/lb_wide.txt 2>&1 || exit 1
cat $O/lb_wide.txt
MPGAN_DBG_HB_WIDE=0 timeout -k 10 300 python tools/bench_bf16.py --layers D.conv3,D.conv4 --modes fwd,dgrad --reps 5 > $O/lb_old.txt 2>&1 || exit 1
cat $O/lb_old.txt
