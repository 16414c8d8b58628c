#!/bin/bash
# Collects the round's profiling evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh OUTDIR
# kernel trace + stats of bench.py (default, bf16, data-parallel path), separate PMC passes (FETCH_SIZE, WRITE_SIZE,
# SQ counters), the other BASELINE configs.  rocprofv3 gets the program itself after `--` (python3 ...).
set -o pipefail
OUT=${1:-gpurun_out/prof_round}
ROOT=$(pwd)
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-other-configs"
run() { # name, rocprof args..., -- command
    local name=$1; shift
    timeout -k 10 300 rocprofv3 "$@" > "$ROOT/$OUT/$name.log" 2>&1 || echo "$name failed" >> "$ROOT/$OUT/failures.txt"
}
# the unprofiled bench lines FIRST, on the box as it comes (after the profiler passes below one box gave 15.3 us/step where the
# same build gives 13.5 before them and on every other box)
( cd "$ROOT" && python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
  python3 bench.py --dtype bf16 --no-cpu-baseline --no-other-configs > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err" )
run stats_f32   --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_f32"   -o run -- $B
run stats_bf16  --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_bf16"  -o run -- $B --dtype bf16
run stats_dp1   --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_dp1"   -o run -- $B --dp-path --dp-mode eager
run pmc_fetch   --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$ROOT/$OUT/pmc_fetch" -o run -- $B
run pmc_write   --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$ROOT/$OUT/pmc_write" -o run -- $B
run pmc_sq      --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d "$ROOT/$OUT/pmc_sq" -o run -- $B
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run pmc_mfma    --kernel-trace --pmc $MF --output-format csv -d "$ROOT/$OUT/pmc_mfma" -o run -- $B
run pmc_mfma_cfg45 --kernel-trace --pmc $MF --output-format csv -d "$ROOT/$OUT/pmc_mfma_cfg45" -o run -- python3 $ROOT/tools/bench_configs.py 4 5 f32 bf16 --steps 24
run stats_cfg45_f32  --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_cfg45_f32"  -o run -- python3 $ROOT/tools/bench_configs.py 4 5 f32 --steps 60
run stats_cfg45_bf16 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_cfg45_bf16" -o run -- python3 $ROOT/tools/bench_configs.py 4 5 bf16 --steps 60
# one config per trace, for the per-step dispatch lists (tools/step_trace.py wants the tail of ONE timed loop)
for c in 4 5; do for d in f32 bf16; do
run trace_cfg${c}_$d --kernel-trace --output-format csv -d "$ROOT/$OUT/trace_cfg${c}_$d" -o run -- python3 $ROOT/tools/bench_configs.py $c $d --steps 40
done; done
cd "$ROOT"
python3 bench.py > "$OUT/bench_default_after_profiling.json" 2> "$OUT/bench_default_after_profiling.err"
python3 bench.py --dp-path --no-cpu-baseline --no-other-configs --steps 2048 --warmup 256 > "$OUT/bench_dp1_graph.json" 2> "$OUT/bench_dp1_graph.err"
GNN_MLP_CHAIN=0 python3 bench.py --no-cpu-baseline --no-other-configs > "$OUT/bench_three_launch.json" 2> "$OUT/bench_three_launch.err"
python3 tools/bench_configs.py 1 2 4 5 f32 bf16 --graph > "$OUT/configs_all.jsonl" 2> "$OUT/configs_all.err"
python3 tools/bench_host_path.py > "$OUT/host_path.txt" 2>&1
# the in-library data-parallel handle, replicas sharing this one GPU (a rehearsal of the control flow, not a scaling number)
for r in direct direct_rs; do
python3 bench.py --gpus 8 --dp-impl library --dp-reducer $r --share-gpu --no-cpu-baseline --steps 512 --warmup 128 > "$OUT/bench_library_8_shared_$r.json" 2> "$OUT/bench_library_8_shared_$r.err"
python3 bench.py --gpus 8 --dp-impl library --dp-reducer $r --share-gpu --no-cpu-baseline --steps 512 --warmup 128 --dtype bf16 > "$OUT/bench_library_8_shared_${r}_bf16.json" 2>> "$OUT/bench_library_8_shared_$r.err"
done
tools/rowblock_probe > "$OUT/rowblock_probe.log" 2>&1
tools/tile_probe > "$OUT/tile_probe.log" 2>&1
for m in 0 1 2 3; do echo "GNN_MLP_HYBRID=$m"; GNN_MLP_HYBRID=$m python3 tools/bench_configs.py 5 f32 --steps 200 2>/dev/null; done > "$OUT/config5_hybrid_choices.jsonl"
python3 tools/bench_trainer.py > "$OUT/trainer.txt" 2>&1
# round 4: evaluation as one call (MT:181-197), the N > 1 record rehearsed with two ranks on this one GPU, the exchange primitive
python3 tools/bench_configs.py inference 128 1024 4096 16384 > "$OUT/inference.jsonl" 2>&1
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_inference" -o run -- python3 $ROOT/tools/bench_configs.py inference 16384 > "$ROOT/$OUT/stats_inference.log" 2>&1 )
python3 bench.py --gpus 2 --backend gloo --share-gpu --steps 256 --warmup 64 --no-cpu-baseline > "$OUT/bench_n2_rehearsal_two_ranks_sharing_one_gpu.json" 2> "$OUT/bench_n2_rehearsal.err"
[ -x tools/_build/exchange_probe ] && tools/_build/exchange_probe 32 > "$OUT/exchange_probe.log" 2>&1
[ -x tools/_build/rowblock_cluster_probe ] && tools/_build/rowblock_cluster_probe 128 > "$OUT/rowblock_cluster_probe.log" 2>&1
ls -R "$OUT" | head -80
