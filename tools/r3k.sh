set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-gfwd --steps 20 > $O/bench_two_$i.json 2> $O/bench_two_$i.err; echo "rc=$?"
MPGAN_D_WGRAD_MAIN=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-gfwd --steps 20 > $O/bench_main_$i.json 2> $O/bench_main_$i.err; echo "rc=$?"
done
echo done
