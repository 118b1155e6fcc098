#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3t
mkdir -p $O
cd $R
echo "[1] gpu tests (conv, networks, fullsize)"
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_networks_gpu.py tests/test_fullsize_gpu.py tests/test_norm_loss_gpu.py -m gpu -q --tb=short > $O/tests.log 2>&1; rc=$?
tail -6 $O/tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
echo "[2] C3 bench, dma"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gfwd --no-phases 2> $O/b1.err | cut -c1-260 | tee -a $O/ab.txt
echo "[2] C3 bench, pipe"
MPGAN_DBG_NO_DMA=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gfwd --no-phases 2> $O/b2.err | cut -c1-260 | tee -a $O/ab.txt
done
