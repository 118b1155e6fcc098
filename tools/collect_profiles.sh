#!/bin/bash
# Round-2 evidence for profiles/ -- run on the GPU box as:  gpurun -- 'bash tools/collect_profiles.sh'
# Every rocprofv3 run has the program directly after `--`; counters are collected in passes of their own.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[1] default bench"; python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/bench_default.err $O/bench_default.json > $O/bench_default.log
echo "[2] kernel stats, default bench"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 $R/bench.py --no-cpu-baseline --no-gfwd > $O/stats_default.log 2>&1
echo "[3] kernel stats, single stream"
MPGAN_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- python3 $R/bench.py --no-cpu-baseline --no-gfwd > $O/stats_single.log 2>&1
echo "[4] kernel stats, G forward only"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_gfwd -- python3 $R/tools/gfwd_loop.py --reps 20 > $O/stats_gfwd.log 2>&1
echo "[5] PMC FETCH_SIZE / WRITE_SIZE, bench"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gfwd > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gfwd > $O/pmc_write.log 2>&1
echo "[6] PMC FETCH_SIZE / WRITE_SIZE, G forward"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_gfwd_fetch -- python3 $R/tools/gfwd_loop.py --reps 3 > $O/pmc_gfwd_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_gfwd_write -- python3 $R/tools/gfwd_loop.py --reps 3 > $O/pmc_gfwd_write.log 2>&1
echo "[7] C5: bf16 storage 3-D line + kernel stats; fp32 3-D line + kernel stats"
python3 $R/bench.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5_bf16.json 2> $O/bench_c5_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5_bf16 -- python3 $R/bench.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-gfwd > $O/stats_c5_bf16.log 2>&1
python3 $R/bench.py --dims 3 --size 128 --batch 4 --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_f32.json 2> $O/bench_c5_f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5_f32 -- python3 $R/bench.py --dims 3 --size 128 --batch 4 --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-gfwd > $O/stats_c5_f32.log 2>&1
echo "[8] SQ counters of D's dense kernels"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 $R/tools/bench_conv.py --layers D.conv2,D.conv3,D.conv4 --modes fwd,dgrad,wgrad --pro --reps 2 > $O/pmc_sq.log 2>&1
echo "[9] per-layer tables"
python3 $R/tools/bench_conv.py --layers all --pro --reps 10 > $O/layer_bench.txt 2>&1
python3 $R/tools/bench_bf16.py --reps 5 > $O/layer_bench_bf16.txt 2>&1
python3 $R/tools/phase_times.py --steps 5 > $O/phase_times.txt 2>&1
python3 $R/tools/bench_variant_b.py --batch 7 --steps 3 > $O/variant_b_bs7.txt 2>&1
echo "[10] PMC FETCH_SIZE / WRITE_SIZE, C5 bf16"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_c5_fetch -- python3 $R/bench.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-gfwd > $O/pmc_c5_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_c5_write -- python3 $R/bench.py --dims 3 --size 128 --batch 4 --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-gfwd > $O/pmc_c5_write.log 2>&1
echo done
