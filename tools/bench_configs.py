#!/usr/bin/env python3
"""Bench over the OTHER BASELINE.json configs (bench.py is the driver's contract and measures configs[1]):
one JSON line per (config, dtype) with the whole-step rate and a `roofline` object in bench.py's format.

  python tools/bench_configs.py [1] [2] [4] [5] [f32] [bf16] [--steps N]

A step is one gradientStep on synthetic batches resident in HBM (gnn_mlp_train_range).  Algorithmic
work per step is SURVEY 8d's accounting: FLOP = (6P - 2 d0 d1) B; bytes = weights (P read forward +
(P - d0 d1) read backward + P gradient + 5P update) x 4 B (f32 masters; bf16 operand reads count 2 B)
+ activations B (d0 + 4 sum_{l>=1} d_l) x element size.  `roofline` is quoted on the WHOLE STEP for
these configs (a step is a chain of 3(L-1) GEMMs that all run at the same roof): bound = MFMA when the
step's arithmetic intensity is above the dtype's ridge, else HBM.  Per-kernel times of the first forward
GEMM and the first-layer gradient GEMM (the two largest) come from the dispatches' own timestamps.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_amd  # noqa: E402

CONFIGS = {
    "1": ([784, 100, 50, 10], 32, "configs[0] shape: 784-100-50-10, batch 32"),
    "2": ([784, 300, 100, 10], 128, "configs[1]/[2] shape: 784-300-100-10, batch 128"),
    "4": ([4096, 2048, 2048, 1024], 512, "configs[3]: 4096-2048-2048-1024, batch 512 (MFMA-bound GEMM roofline run)"),
    "5": ([784, 1024, 1024, 1024, 10], 256, "configs[4]: 784-1024-1024-1024-10, batch 256"),
}
PEAK_TF = {"f32": 157.3, "bf16": 2500.0}   # MI355X_MICROARCH.md dense MFMA peaks
HBM_GBS = 8000.0


def run(key, dtype, steps, graph=False, n_batches=8, timed_kernels=True, general=None):
    """graph: the hipGraph-captured training step of gnn_mlp_train_range (GNN_MLP_GRAPH=1, read at create): one pass over
    the resident batches captured once, replayed; `steps` should then be a multiple of 2 * n_batches.
    general: (inner_act, last_act) -> the same shape as a GeneralNeuralNet (element-wise output + half-squared loss,
    GNN:215-218, 236-239, 267-271) instead of SoftmaxCrossEntropyNeuralNet."""
    import torch
    dims, B, label = CONFIGS[key]
    rng = np.random.default_rng(0)
    nb = n_batches
    X = rng.random((B * nb, dims[0])) * (rng.random((B * nb, dims[0])) < 0.19)
    Y = np.eye(dims[-1])[rng.integers(0, dims[-1], B * nb)]
    old = os.environ.get("GNN_MLP_GRAPH")
    os.environ["GNN_MLP_GRAPH"] = "1" if graph else "0"   # (read once, at create)
    try:
        dt_enum = gnn_amd.DTYPE_BF16 if dtype == "bf16" else gnn_amd.DTYPE_F32
        if general is not None:
            net = gnn_amd.GeneralNeuralNet(dims, inner_act=general[0], last_act=general[1], dtype=dt_enum, max_batch=B)
            label = label + ", GeneralNeuralNet %s/%s + half-squared loss" % tuple(general)
        else:
            net = gnn_amd.SoftmaxCrossEntropyNeuralNet(dims, dtype=dt_enum, max_batch=B)
    finally:
        if old is None:
            del os.environ["GNN_MLP_GRAPH"]
        else:
            os.environ["GNN_MLP_GRAPH"] = old
    if dims[0] > 1000 or len(dims) > 4:
        net.set_weights(net.get_weights() * 0.05)   # keep the softmax unsaturated at these widths
    net.upload_dataset(X, Y)
    side = torch.cuda.Stream()
    net.set_stream(side.cuda_stream)
    net.train_range(0, B, max(20, 2 * nb), 0.0125, 0.9)   # (under `graph` this captures the pass and replays it once)
    net.synchronize()
    # wall clock AND two events on the stream the kernels are launched on (a torch stream bound to the handle)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    net.synchronize()
    t0 = time.perf_counter()
    e0.record(side)
    net.train_range(0, B, steps, 0.0125, 0.9)
    e1.record(side)
    net.synchronize()
    dt = (time.perf_counter() - t0) / steps
    dt_ev = e0.elapsed_time(e1) * 1e-3 / steps
    fwd_us = grad_us = 0.0
    fwd_n = grad_n = 0
    if timed_kernels:
        net.timing_enable(True)
        net.train_range(0, B, min(steps, 100), 0.0125, 0.9)
        net.synchronize()
        fwd_us, fwd_n = net.timing_read(0)
        grad_us, grad_n = net.timing_read(1)
        net.timing_enable(False)
    P = sum(dims[l] * dims[l + 1] for l in range(len(dims) - 1))
    flop = (6 * P - 2 * dims[0] * dims[1]) * B
    eo = 2 if dtype == "bf16" else 4
    nbytes = eo * (2 * P - dims[0] * dims[1]) + 4 * 6 * P + eo * B * (dims[0] + 4 * sum(dims[1:]))
    tf = flop / dt / 1e12
    gbs = nbytes / dt / 1e9
    ridge = PEAK_TF[dtype] * 1e12 / (HBM_GBS * 1e9)
    ai = flop / nbytes
    bound = "mfma" if ai >= ridge else "hbm"
    ach, peak, unit = (tf, PEAK_TF[dtype], "TFLOP/s") if bound == "mfma" else (gbs, HBM_GBS, "GB/s")
    kern = {}
    if fwd_n:
        f = 2.0 * B * dims[0] * dims[1]
        kern["first forward GEMM (%dx%dx%d)" % (B, dims[0], dims[1])] = {
            "avg_us": round(fwd_us, 2), "launches": fwd_n, "tflops": round(f / (fwd_us * 1e-6) / 1e12, 2),
            "mfma_frac": round(f / (fwd_us * 1e-6) / 1e12 / PEAK_TF[dtype], 4)}
    if grad_n:
        f = 2.0 * B * dims[0] * dims[1]
        kern["first-layer gradient GEMM + update (%dx%dx%d)" % (dims[0], dims[1], B)] = {
            "avg_us": round(grad_us, 2), "launches": grad_n, "tflops": round(f / (grad_us * 1e-6) / 1e12, 2),
            "mfma_frac": round(f / (grad_us * 1e-6) / 1e12 / PEAK_TF[dtype], 4)}
    line = {"metric": "training samples/sec, %s" % label, "value": round(B / dt, 1), "unit": "samples/s", "n_gpus": 1,
            "steps": steps, "ms_per_step": round(dt * 1e3, 5), "us_per_step_events": round(dt_ev * 1e6, 2), "dtype": dtype, "data": "synthetic",
            "config": {"workload": label, "dims": dims, "batch": B, "step_launches": net.step_launches, "hipgraph": bool(graph),
                       "batches_resident": nb, "specialization": net.specialization, "rowblock_state": net.rowblock_state},
            "roofline": {"bound": bound, "kernel": "whole gradientStep (forward + backward GEMM chain + update)",
                         "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4), "traffic": None,
                         "flop_per_step": flop, "algorithmic_bytes_per_step": nbytes,
                         "arithmetic_intensity_flop_per_byte": round(ai, 1), "ridge_flop_per_byte": round(ridge, 1),
                         "tflops": round(tf, 2), "gbs": round(gbs, 1),
                         "mfma_frac": round(tf / PEAK_TF[dtype], 4), "hbm_frac": round(gbs / HBM_GBS, 4),
                         "kernels": kern}}
    net.close()
    return line


def run_inference(key="2", rows=60000, max_batch=4096, reps=5, dtype="f32"):
    """testOnTrainingData (MT:181-197) as ONE call: propagate + the `>=` argmax + the comparison with the expected class over
    `rows` resident 784-pixel rows (MNIST's training-set size), gnn_mlp_count_hits_range -- blocks of max_batch rows, no host work
    in between, one count read back.  Algorithmic work per row: 2 P FLOP (the forward products), d_0 x 4 bytes of input;
    arithmetic intensity ~170 FLOP/B, far above the f32 ridge: quoted against the MFMA peak."""
    dims, _, label = CONFIGS[key]
    rng = np.random.default_rng(1)
    pix = rng.integers(0, 256, (rows, dims[0]), dtype=np.uint8)
    pix[rng.random((rows, dims[0])) < 0.8] = 0
    lab = rng.integers(0, dims[-1], rows, dtype=np.uint8)
    net = gnn_amd.SoftmaxCrossEntropyNeuralNet(dims, dtype=gnn_amd.DTYPE_BF16 if dtype == "bf16" else gnn_amd.DTYPE_F32, max_batch=max_batch)
    net.upload_dataset_u8(pix, lab)
    hits = net.count_hits_range(0, rows)     # warm (and the answer: ~10 % of a random net's labels are right)
    t0 = time.perf_counter()
    for _ in range(reps):
        assert net.count_hits_range(0, rows) == hits
    dt = (time.perf_counter() - t0) / reps
    # the same loop one ABI call per block (argmax_range + a comparison on the host): the only form until round 4
    t1 = time.perf_counter()
    per_block = sum(int((net.argmax_range(f, min(max_batch, rows - f)) == lab[f:f + max_batch]).sum()) for f in range(0, rows, max_batch))
    dt_blocks = time.perf_counter() - t1
    assert abs(per_block - hits) <= 10   # (the one-call form walks larger blocks through other kernels: only near-ties may move)
    P = sum(dims[l] * dims[l + 1] for l in range(len(dims) - 1))
    flop = 2.0 * P * rows
    nbytes = 4.0 * rows * (dims[0] + 1) + 4.0 * P
    tf = flop / dt / 1e12
    out = {"metric": "inference samples/sec (propagate + argmax + hit count, MT:181-197), %s" % label.split(":")[0], "value": round(rows / dt, 1),
           "unit": "samples/s", "rows": rows, "max_batch": max_batch, "ms_per_pass": round(dt * 1e3, 4), "dtype": dtype, "hits": int(hits),
           "per_block_calls_samples_per_s": round(rows / dt_blocks, 1),
           "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_TF[dtype], "unit": "TFLOP/s", "frac": round(tf / PEAK_TF[dtype], 4),
                        "flop_per_pass": flop, "algorithmic_bytes_per_pass": nbytes, "hbm_frac": round(nbytes / dt / 1e9 / HBM_GBS, 4)}}
    net.close()
    return out


def main():
    args = sys.argv[1:]
    if args and args[0] == "inference":
        for mb in [int(a) for a in args[1:]] or [128, 1024, 4096, 16384]:
            print(json.dumps(run_inference(max_batch=mb)), flush=True)
        return
    steps = None
    if "--steps" in args:
        i = args.index("--steps")
        steps = int(args[i + 1])
        del args[i:i + 2]
    graph = "--graph" in args
    dtypes = [a for a in args if a in ("f32", "bf16")] or ["f32"]
    which = [a for a in args if a in CONFIGS] or ["1", "2", "5", "4"]
    for key in which:
        for dtype in dtypes:
            n = steps or (200 if key in ("4", "5") else 2000)
            print(json.dumps(run(key, dtype, n, n_batches=16 if key == "5" else 8)), flush=True)
            if graph and key == "5":   # configs[4] names the hipGraph-captured step: both forms, side by side
                print(json.dumps(run(key, dtype, n - n % 32, graph=True, n_batches=16)), flush=True)


if __name__ == "__main__":
    main()
