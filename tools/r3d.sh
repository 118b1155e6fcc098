set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
echo "[1] tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q --tb=short > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -8 $O/tests.log
echo "[2] phases"; timeout -k 10 200 python tools/phase_times.py --steps 5 --probe gbwd > $O/ph_gbwd.txt 2>&1; echo rc=$?
echo "[2b] stamps"; timeout -k 10 200 python tools/kernel_phases.py --what fwd > $O/kp_fwd.txt 2>&1; echo rc=$?
timeout -k 10 200 python tools/kernel_phases.py --what bwd > $O/kp_bwd.txt 2>&1; echo rc=$?
echo "[3] bench"; timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
echo "[4] single-stream kernel stats"
MPGAN_SINGLE_STREAM=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- python3 $R/bench.py --no-cpu-baseline --no-gfwd --no-phases --steps 5 > $O/stats_single.log 2>&1; echo rc=$?
echo done
