#!/bin/bash
# Copy the summaries of the newest tools/collect_profiles.sh run (gpurun_out/r04/) into profiles/ (tracked).
set -eo pipefail
cd "$(dirname "$0")/.."
P=gpurun_out/r04
T=r04
newest() { ls -t $1 | head -1; }
have() { ls $1 > /dev/null 2>&1; }
if have "$P/bench_default.log"; then cp $P/bench_default.log profiles/${T}_bench_default.log; fi
for tag in default single gfwd gbwd c5_bf16 vb; do
  if have "$P/stats_$tag/runc/*_kernel_stats.csv"; then
    name=$tag
    [[ $tag == default ]] && name=bench_default
    [[ $tag == single ]] && name=bench_single_stream
    [[ $tag == vb ]] && name=variant_b
    cp "$(newest "$P/stats_$tag/runc/*_kernel_stats.csv")" profiles/${T}_${name}_kernel_stats.csv
  fi
done
for f in phase_times phase_times_gbwd_calls kernel_phases_gfwd kernel_phases_gbwd layer_bench layer_bench_bf16 variant_b_bs7 variant_b_families c5_gfwd_calls c5_gbwd_calls infer; do
  if have "$P/$f.txt"; then grep -v "amdgpu.ids\|RuntimeWarning\|mean = lambda" $P/$f.txt > profiles/${T}_$f.txt; fi
done
if have "$P/bench_c5_bf16.json"; then cat $P/bench_c5_bf16.err $P/bench_c5_bf16.json > profiles/${T}_bench_c5_bf16.log; fi
if have "$P/pmc_fetch/runc/*_counter_collection.csv"; then
  python3 tools/make_traffic.py "$(newest "$P/pmc_fetch/runc/*_counter_collection.csv")" "$(newest "$P/pmc_write/runc/*_counter_collection.csv")" profiles/traffic.json | head -4
fi
if have "$P/pmc_gfwd_fetch/runc/*_counter_collection.csv"; then
  python3 tools/make_traffic.py "$(newest "$P/pmc_gfwd_fetch/runc/*_counter_collection.csv")" "$(newest "$P/pmc_gfwd_write/runc/*_counter_collection.csv")" profiles/${T}_gfwd_traffic.json | head -3
fi
for tag in sq sq_g sq_bf16; do
  if have "$P/pmc_$tag/runc/*_counter_collection.csv"; then
    out=pmc_dconv_sq_counters
    [[ $tag == sq_g ]] && out=pmc_generator_sq_counters
    [[ $tag == sq_bf16 ]] && out=pmc_bf16_sq_counters
    python3 tools/make_sq_summary.py "$(newest "$P/pmc_$tag/runc/*_counter_collection.csv")" profiles/${T}_$out.csv
  fi
done
if have "$P/pmc_c5_fetch/runc/*_counter_collection.csv"; then
  python3 tools/make_traffic.py "$(newest "$P/pmc_c5_fetch/runc/*_counter_collection.csv")" "$(newest "$P/pmc_c5_write/runc/*_counter_collection.csv")" profiles/traffic_c5_bf16.json conv_bf16.hip | head -6
fi
ls profiles | grep $T
