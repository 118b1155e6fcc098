set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
echo "[1] tests"; timeout -k 10 1000 python -X faulthandler -m pytest tests -m gpu -q --tb=short > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -6 $O/tests.log
echo "[3] bench"; timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
echo "[2] phases (single stream, no probe)"; timeout -k 10 200 python tools/phase_times.py --steps 5 > $O/phase_times.txt 2>&1; echo rc=$?
echo done
