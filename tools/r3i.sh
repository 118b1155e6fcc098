set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3i
mkdir -p $O
cd $R
echo "[1] tests"; timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_networks_gpu.py tests/test_variant_b_gpu.py "tests/test_fullsize_gpu.py::test_full_gd_step_at_c3_matches_oracle" -m gpu -q --tb=short > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -5 $O/tests.log
timeout -k 10 200 python tools/kernel_phases.py --what fwd > $O/kp_fwd.txt 2>&1; echo rc=$?
timeout -k 10 200 python tools/kernel_phases.py --what bwd > $O/kp_bwd.txt 2>&1; echo rc=$?
echo "[3] bench"; timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
echo done
