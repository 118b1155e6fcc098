#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3r
mkdir -p $O
cd $R
echo "[1] bf16 + c5 tests"
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py tests/test_c5_step_gpu.py tests/test_fullsize_gpu.py -m gpu -q --tb=short > $O/tests.log 2>&1; rc=$?
tail -6 $O/tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
echo "[2] C5 bench"
timeout -k 10 400 python bench.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; rc=$?
cut -c1-400 $O/bench_c5.json; echo "bench rc=$rc"
MPGAN_DBG_HB_WIDE=0 timeout -k 10 400 python bench.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-phases --no-gfwd > $O/bench_c5_old.json 2> $O/bench_c5_old.err
cut -c1-300 $O/bench_c5_old.json
