#!/bin/bash
# Copies what tools/profile_round.sh left in OUTDIR (merged back under gpurun_out/) into profiles/rNN under the names
# DESIGN.md and bench.py refer to:   tools/collect_profiles.sh gpurun_out/r02prof3 profiles/r02
set -e
SRC=$1; DST=$2
mkdir -p "$DST"
cp "$SRC/stats_f32/run_kernel_stats.csv"        "$DST/final_f32_kernel_stats.csv"
cp "$SRC/stats_bf16/run_kernel_stats.csv"       "$DST/final_bf16_kernel_stats.csv"
cp "$SRC/stats_dp1/run_kernel_stats.csv"        "$DST/final_dp_path_n1_eager_kernel_stats.csv"
cp "$SRC/stats_cfg45_f32/run_kernel_stats.csv"  "$DST/final_configs45_f32_kernel_stats.csv"
cp "$SRC/stats_cfg45_bf16/run_kernel_stats.csv" "$DST/final_configs45_bf16_kernel_stats.csv"
python3 tools/pmc_summary.py "$SRC/pmc_fetch/run_counter_collection.csv" "$SRC/pmc_write/run_counter_collection.csv" > "$DST/pmc_traffic.json"
python3 tools/sq_summary.py "$SRC/pmc_sq/run_counter_collection.csv" > "$DST/sq_counters.json"
python3 tools/mfma_summary.py "$SRC/pmc_mfma/run_counter_collection.csv" "$SRC/pmc_mfma_cfg45/run_counter_collection.csv" > "$DST/mfma_counters.json"
cp "$SRC/bench_default.json"      "$DST/final_bench_f32.json"
cp "$SRC/bench_bf16.json"         "$DST/final_bench_bf16.json"
cp "$SRC/bench_dp1_graph.json"    "$DST/final_bench_dp_path_n1_graph.json"
cp "$SRC/bench_three_launch.json" "$DST/final_bench_three_launch_GNN_MLP_CHAIN0.json"
cp "$SRC/configs_all.jsonl"       "$DST/final_configs_all.jsonl"
cp "$SRC/host_path.txt"           "$DST/final_host_path.txt"
cp "$SRC/trainer.txt"             "$DST/final_trainer.txt"
ls -l "$DST" | head -60
