set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
echo "[1] tests"; timeout -k 10 1000 python -m pytest tests -m gpu -q --tb=short > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -6 $O/tests.log
echo "[2] phases (single stream, no probe)"; timeout -k 10 200 python tools/phase_times.py --steps 5 > $O/phase_times.txt 2>&1; echo rc=$?
echo "[3] C5 bench"; timeout -k 10 400 python bench.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "rc=$?"
echo "[4] C5 G probes"; timeout -k 10 300 python tools/phase_times.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 3 --probe gfwd > $O/c5_gfwd.txt 2>&1; echo rc=$?
timeout -k 10 300 python tools/phase_times.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 3 --probe gbwd > $O/c5_gbwd.txt 2>&1; echo rc=$?
echo done
