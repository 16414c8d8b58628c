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
: > "$DST/final_step_trace_configs45.txt"   # (the loop below appends: one section per config and dtype)
for c in 4 5; do for d in f32 bf16; do
  n=$(python3 - "$SRC/trace_cfg${c}_$d/run_kernel_trace.csv" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows][-400:]
# the period of the dispatch sequence = launches per step
for p in range(3, 60):
    if all(names[-1 - i] == names[-1 - i - p] for i in range(2 * p)):
        print(p); break
else:
    print(16)
PY
)
  { echo "== config key $c, $d: the last two steps' dispatches ($n launches per step; under the profiler)"; python3 tools/step_trace.py "$SRC/trace_cfg${c}_$d/run_kernel_trace.csv" $((2 * n)); } >> "$DST/final_step_trace_configs45.txt"
done; done
cp "$SRC/rowblock_probe.log" "$DST/rowblock_probe_final.log"
cp "$SRC/tile_probe.log" "$DST/tile_probe_final.log"
cp "$SRC/config5_hybrid_choices.jsonl" "$DST/config5_hybrid_choices.jsonl"
for r in direct direct_rs; do cp "$SRC/bench_library_8_shared_$r.json" "$DST/bench_library_8_replicas_sharing_one_gpu_$r.json"; cp "$SRC/bench_library_8_shared_${r}_bf16.json" "$DST/bench_library_8_replicas_sharing_one_gpu_${r}_bf16.json"; done
cp "$SRC/bench_default.json"      "$DST/final_bench_f32.json"
cp "$SRC/bench_bf16.json"         "$DST/final_bench_bf16.json"
cp "$SRC/bench_dp1_graph.json"    "$DST/final_bench_dp_path_n1_graph.json"
cp "$SRC/bench_three_launch.json" "$DST/final_bench_three_launch_GNN_MLP_CHAIN0.json"
cp "$SRC/configs_all.jsonl"       "$DST/final_configs_all.jsonl"
cp "$SRC/host_path.txt"           "$DST/final_host_path.txt"
cp "$SRC/trainer.txt"             "$DST/final_trainer.txt"
[ -f "$SRC/inference.jsonl" ] && cp "$SRC/inference.jsonl" "$DST/final_inference.jsonl"
[ -f "$SRC/stats_inference/run_kernel_stats.csv" ] && cp "$SRC/stats_inference/run_kernel_stats.csv" "$DST/final_inference_kernel_stats.csv"
[ -f "$SRC/bench_n2_rehearsal_two_ranks_sharing_one_gpu.json" ] && cp "$SRC/bench_n2_rehearsal_two_ranks_sharing_one_gpu.json" "$DST/"
[ -f "$SRC/bench_default_after_profiling.json" ] && cp "$SRC/bench_default_after_profiling.json" "$DST/final_bench_f32_after_profiler_passes.json"
ls -l "$DST" | head -60
