#!/bin/bash
# Round-4 evidence for profiles/ -- run on the GPU box as:  gpurun -- 'bash tools/collect_profiles.sh [part]'
# (parts: b = counters FIRST -- tools/publish_profiles.sh turns them into profiles/traffic*.json, which the bench lines of
#  parts a and c then carry --, a = bench + kernel stats, c = C5 + variant B + tables, d = inference; one gpurun call per part
#  keeps a call inside its time limit).  Every rocprofv3 run has the program directly after `--`; counters are
#  collected in passes of their own (--kernel-trace --pmc only).
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
PART=${1:-bacd}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
if [[ $PART == *a* ]]; then
echo "[a1] default bench"; python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/bench_default.err $O/bench_default.json > $O/bench_default.log
echo "[a2] kernel stats, default bench"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 $R/bench.py --no-cpu-baseline --no-gfwd --no-phases > $O/stats_default.log 2>&1
echo "[a3] kernel stats, single stream"
MPGAN_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- python3 $R/bench.py --no-cpu-baseline --no-gfwd --no-phases > $O/stats_single.log 2>&1
echo "[a4] kernel stats, G forward only / G backward (single stream)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_gfwd -- python3 $R/tools/gfwd_loop.py --reps 20 > $O/stats_gfwd.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_gbwd -- python3 $R/tools/gfwd_loop.py --reps 10 --backward > $O/stats_gbwd.log 2>&1
echo "[a5] phases, in-kernel stamps"
python3 $R/tools/phase_times.py --steps 5 > $O/phase_times.txt 2>&1
python3 $R/tools/phase_times.py --steps 5 --probe gbwd > $O/phase_times_gbwd_calls.txt 2>&1
python3 $R/tools/kernel_phases.py --what fwd > $O/kernel_phases_gfwd.txt 2>&1
python3 $R/tools/kernel_phases.py --what bwd > $O/kernel_phases_gbwd.txt 2>&1
fi
if [[ $PART == *b* ]]; then
echo "[b1] PMC FETCH_SIZE / WRITE_SIZE, bench"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gfwd --no-phases > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gfwd --no-phases > $O/pmc_write.log 2>&1
echo "[b2] PMC FETCH_SIZE / WRITE_SIZE, G forward"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_gfwd_fetch -- python3 $R/tools/gfwd_loop.py --reps 3 > $O/pmc_gfwd_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_gfwd_write -- python3 $R/tools/gfwd_loop.py --reps 3 > $O/pmc_gfwd_write.log 2>&1
echo "[b3] SQ counters: D's dense kernels, the generator's kernels (forward + backward), the bf16 kernels"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_sq -- python3 $R/tools/bench_conv.py --layers D.conv2,D.conv3,D.conv4 --modes fwd,dgrad,wgrad --pro --reps 2 > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_sq_g -- python3 $R/tools/gfwd_loop.py --reps 2 --backward > $O/pmc_sq_g.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_sq_bf16 -- python3 $R/tools/bench_bf16.py --reps 2 > $O/pmc_sq_bf16.log 2>&1
echo "[b4] PMC FETCH_SIZE / WRITE_SIZE, C5 bf16"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_c5_fetch -- python3 $R/bench.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-gfwd --no-phases > $O/pmc_c5_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_c5_write -- python3 $R/bench.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-gfwd --no-phases > $O/pmc_c5_write.log 2>&1
fi
if [[ $PART == *c* ]]; then
echo "[c1] C5: bf16 storage 3-D line + kernel stats"
python3 $R/bench.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5_bf16.json 2> $O/bench_c5_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5_bf16 -- python3 $R/bench.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-gfwd --no-phases > $O/stats_c5_bf16.log 2>&1
python3 $R/tools/phase_times.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 3 --probe gfwd > $O/c5_gfwd_calls.txt 2>&1
python3 $R/tools/phase_times.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 3 --probe gbwd > $O/c5_gbwd_calls.txt 2>&1
echo "[c2] per-layer tables"
python3 $R/tools/bench_conv.py --layers all --pro --reps 10 > $O/layer_bench.txt 2>&1
python3 $R/tools/bench_bf16.py --reps 5 > $O/layer_bench_bf16.txt 2>&1
echo "[c3] variant B at the reference's scale: wall time, kernel stats, per-family table"
python3 $R/tools/bench_variant_b.py --batch 7 --steps 3 > $O/variant_b_bs7.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_vb -- python3 $R/tools/bench_variant_b.py --batch 7 --steps 3 > $O/stats_vb.log 2>&1
python3 $R/tools/bench_variant_b.py --batch 7 --steps 2 --families > $O/variant_b_families.txt 2>&1
fi
if [[ $PART == *d* ]]; then
echo "[d1] N1: eval-mode generator forward (eager, HIP graph, train mode)"
python3 $R/tools/bench_infer.py > $O/infer.txt 2>&1
fi
echo done
