set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
echo "[1] tests"; timeout -k 10 900 python -m pytest tests -m gpu -q --tb=short -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -5 $O/tests.log
echo "[2] smoke"; timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
echo "[3] bench"; timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
echo "[4] phases"; timeout -k 10 200 python tools/phase_times.py --steps 5 --probe gbwd > $O/ph_gbwd.txt 2>&1; echo rc=$?
timeout -k 10 200 python tools/phase_times.py --steps 5 --probe gfwd > $O/ph_gfwd.txt 2>&1; echo rc=$?
echo "[5] variant B stats"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_vb -- python3 $R/tools/bench_variant_b.py --batch 7 --steps 3 > $O/stats_vb.log 2>&1; echo rc=$?
echo done
