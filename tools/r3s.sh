#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3s
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -k "dma" > $O/tests.log 2>&1; rc=$?
tail -15 $O/tests.log; [ $rc -ne 0 ] && exit $rc
for i in 1 2; do
echo "dma"; timeout -k 10 200 python tools/bench_conv.py --layers D.conv2,D.conv3,D.conv4 --modes dgrad --reps 10 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt || exit 1
echo "pipe"; MPGAN_DBG_NO_DMA=1 timeout -k 10 200 python tools/bench_conv.py --layers D.conv2,D.conv3,D.conv4 --modes dgrad --reps 10 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt || exit 1
done
