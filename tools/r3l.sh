#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3l
mkdir -p $O
cd $R
echo "[1] full gpu tests"
timeout -k 10 1000 python -m pytest tests -m gpu -q --tb=short > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
echo "[2] profiles part a"
timeout -k 10 700 bash tools/collect_profiles.sh a > $O/collect_a.log 2>&1; rc=$?
tail -3 $O/collect_a.log; echo "collect rc=$rc"
exit $rc
