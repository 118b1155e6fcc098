set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
echo "[1] tests"; timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_norm_loss_gpu.py "tests/test_fullsize_gpu.py::test_full_gd_step_at_c3_matches_oracle" "tests/test_networks_gpu.py" tests/test_c5_step_gpu.py -m gpu -q --tb=short > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -8 $O/tests.log
echo "[2] phases"; timeout -k 10 200 python tools/phase_times.py --steps 5 --probe gbwd > $O/ph_gbwd.txt 2>&1; echo rc=$?
echo "[3] bench"; timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
echo "[4] single-stream kernel stats"
MPGAN_SINGLE_STREAM=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- python3 $R/bench.py --no-cpu-baseline --no-gfwd --no-phases --steps 5 > $O/stats_single.log 2>&1; echo rc=$?
echo done
