set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
echo "[1] tests"; timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py tests/test_c5_step_gpu.py tests/test_conv_gpu.py -m gpu -q --tb=short > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -5 $O/tests.log
echo "[2] layer bench bf16"; timeout -k 10 300 python tools/bench_bf16.py --reps 5 > $O/layer_bench_bf16.txt 2>&1; cat $O/layer_bench_bf16.txt
echo "[3] C5 bench"; timeout -k 10 400 python bench.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "rc=$?"; tail -3 $O/bench_c5.err
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_sq_bf16 -- python3 $R/tools/bench_bf16.py --reps 2 > $O/pmc_sq_bf16.log 2>&1; echo rc=$?
echo done
