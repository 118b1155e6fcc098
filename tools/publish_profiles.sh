#!/bin/bash
# Copy the summaries of the newest tools/collect_profiles.sh run (gpurun_out/r02/) into profiles/ (tracked).
set -eo pipefail
cd "$(dirname "$0")/.."
P=gpurun_out/r02
newest() { ls -t $1 | head -1; }
cp $P/bench_default.log profiles/r02_bench_default.log
cp "$(newest "$P/stats_default/runc/*_kernel_stats.csv")" profiles/r02_bench_default_kernel_stats.csv
cp "$(newest "$P/stats_single/runc/*_kernel_stats.csv")" profiles/r02_bench_single_stream_kernel_stats.csv
cp "$(newest "$P/stats_gfwd/runc/*_kernel_stats.csv")" profiles/r02_gfwd_kernel_stats.csv
cp "$(newest "$P/stats_c5_bf16/runc/*_kernel_stats.csv")" profiles/r02_c5_bf16_kernel_stats.csv
cp "$(newest "$P/stats_c5_f32/runc/*_kernel_stats.csv")" profiles/r02_c5_f32_kernel_stats.csv
cat $P/bench_c5_bf16.err $P/bench_c5_bf16.json > profiles/r02_bench_c5_bf16.log
cat $P/bench_c5_f32.err $P/bench_c5_f32.json > profiles/r02_bench_c5_f32.log
cp $P/layer_bench.txt profiles/r02_layer_bench.txt
cp $P/layer_bench_bf16.txt profiles/r02_layer_bench_bf16.txt
cp $P/phase_times.txt profiles/r02_phase_times.txt
cp $P/variant_b_bs7.txt profiles/r02_variant_b_bs7.txt
python3 tools/make_traffic.py "$(newest "$P/pmc_fetch/runc/*_counter_collection.csv")" "$(newest "$P/pmc_write/runc/*_counter_collection.csv")" profiles/traffic.json | head -4
python3 tools/make_traffic.py "$(newest "$P/pmc_gfwd_fetch/runc/*_counter_collection.csv")" "$(newest "$P/pmc_gfwd_write/runc/*_counter_collection.csv")" profiles/r02_gfwd_traffic.json | head -3
python3 tools/make_sq_summary.py "$(newest "$P/pmc_sq/runc/*_counter_collection.csv")" profiles/r02_pmc_dconv_sq_counters.csv
if ls $P/pmc_c5_fetch/runc/*_counter_collection.csv > /dev/null 2>&1; then
  python3 tools/make_traffic.py "$(newest "$P/pmc_c5_fetch/runc/*_counter_collection.csv")" "$(newest "$P/pmc_c5_write/runc/*_counter_collection.csv")" profiles/traffic_c5_bf16.json conv_bf16.hip | head -6
fi
