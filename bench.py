#!/usr/bin/env python3
"""bench.py -- training samples/sec of the MLP mini-batch SGD path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one gradientStep (SCE:297-346: forward + backward + [all-reduce] + momentum update)
on one batch of synthetic 784-dim inputs already resident in HBM.  The workload at every N is
BASELINE.json configs[1] per GPU: 784-300-100-10, fp32, batch 128 per GPU (weak scaling: the
global batch is 128*N, sharded by rows, ONE all-reduce(SUM) of the flat weight gradient per step
over RCCL, identical update on every rank).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DIMS = [784, 300, 100, 10]
BATCH = 128
STEP, MOMENTUM = 0.0125, 0.9  # MT:227-229
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def pmc_traffic(kernel_substring):
    """L2<->fabric bytes per launch of a kernel from the committed rocprofv3 PMC passes
    (profiles/r01/pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs of this very command,
    FETCH_SIZE doubled as the gfx950 guide prescribes).  bench.py cannot collect PMCs itself."""
    path = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
    try:
        with open(path) as f:
            for name, d in json.load(f)["kernels"].items():
                if kernel_substring in name:
                    return round(d["traffic_bytes_per_launch"])
    except Exception:
        pass
    return None


def synthetic(n, seed):
    """X ~ U[0,1) 784-dim, uniform one-hot labels; generated here, never shipped."""
    rng = np.random.default_rng(seed)
    X = rng.random((n, DIMS[0]))
    Y = np.eye(DIMS[-1])[rng.integers(0, DIMS[-1], n)]
    return X, Y


def cpu_baseline(seconds=12.0):
    """The oracle (serial fp64 C restatement of the Java loop, NOT a JVM run) timed on one host
    core on a bounded sample of the same workload.  Reported baseline only."""
    from oracle import oracle
    oracle.build()
    net = oracle.OracleNet(DIMS)
    X, Y = synthetic(BATCH * 4, 1234)
    net.gradient_step(X[:BATCH], Y[:BATCH], STEP, MOMENTUM)  # warm
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds:
        r = (steps % 4) * BATCH
        net.gradient_step(X[r:r + BATCH], Y[r:r + BATCH], STEP, MOMENTUM)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": round(steps * BATCH / dt, 1), "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": "%d gradientSteps of batch %d on 784-300-100-10 (fp64, per-sample loop order of "
                      "SCE:297-346; C restatement, not a JVM run)" % (steps, BATCH)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="data-parallel path: eager steps, no hipGraph replay")
    ap.add_argument("--dp-path", action="store_true",
                    help="run the data-parallel code path (compute -> all_reduce -> update) even at N=1")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend; 'gloo' + --share-gpu rehearses the N>1 control flow on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    args = ap.parse_args()

    # stdout carries ONE JSON line: libraries that chat on fd 1 (RCCL prints a version banner there,
    # gloo its connection report) are routed to stderr until the line is printed
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import gnn_amd

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.dp_path:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    K, W = args.steps, args.warmup
    n_batches = 64
    X, Y = synthetic(BATCH * n_batches, 1000 + rank)  # each rank owns its row shard of the global batch
    net = gnn_amd.SoftmaxCrossEntropyNeuralNet(DIMS, device=local_rank, max_batch=BATCH)
    net.upload_dataset(X, Y)

    def barrier():
        if dist is not None:
            dist.barrier()
        net.synchronize()
        torch.cuda.synchronize()

    if dist is None:
        def run(first_batch, n):
            net.train_range((first_batch % n_batches) * BATCH, BATCH, n, STEP, MOMENTUM)
    else:
        # data parallel (graph-neural-net_amd/data_parallel.py): kernels on torch's current stream,
        # gradient buffer owned by torch so that RCCL reduces it in place, one all-reduce per step
        from gnn_amd import data_parallel as dp
        side = torch.cuda.Stream()
        stepper = dp.DataParallelStep(dp.HipEngine(net, torch, stream=side), dist, always_reduce=args.dp_path)
        graphed = None
        if not args.no_graph and args.backend == "nccl":  # (a gloo collective cannot be captured)
            try:  # one pass over the 64 resident batches as ONE graph launch
                graphed = dp.GraphedSteps(stepper, torch, side, [b * BATCH for b in range(n_batches)],
                                          BATCH, STEP, MOMENTUM)
            except Exception as e:  # capture of the collective not available: eager steps
                if rank == 0:
                    print("graph capture failed (%s: %s); running eager" % (type(e).__name__, str(e).splitlines()[0]), file=sys.stderr)
                graphed = None
                side = torch.cuda.Stream()       # the stream of a failed capture may stay unusable
                stepper.engine.rebind_stream(side)

        def run(first_batch, n, eager=False):
            s = 0
            with torch.cuda.stream(side):
                while s < n:
                    b = (first_batch + s) % n_batches
                    if graphed is not None and not eager and b == 0 and n - s >= n_batches:
                        graphed.replay()
                        s += n_batches
                    else:
                        stepper.step(b * BATCH, BATCH, STEP, MOMENTUM)
                        s += 1

    run(0, W)
    barrier()
    t0 = time.perf_counter()
    run(0 if dist is not None else W, K)  # data parallel: start on a graph boundary (64 resident batches)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- dominant-kernel roofline: HIP events on the kernel's own stream, over the same step loop
    roofline = None
    cpu = None
    # The kernel-timing pass steps the net again. On the data-parallel path a step contains a
    # collective, so EVERY rank takes part (eagerly: kernels inside a replayed graph are not
    # timed); only rank 0 records and reports.
    nt = min(K, 1000) if dist is None else min(K, 256)
    if rank == 0:
        net.timing_enable(True)
    if dist is None:
        run(W + K, nt)
    else:
        run(K, nt, eager=True)
    barrier()
    if rank == 0:
        fwd_us, fwd_n = net.timing_read(0)
        grad_us, grad_n = net.timing_read(1)
        mid_us, mid_n = net.timing_read(3)
        net.timing_enable(False)
        P_all = sum(DIMS[l] * DIMS[l + 1] for l in range(len(DIMS) - 1))
        P_mid = P_all - DIMS[0] * DIMS[1]
        e4 = 4  # bytes per element (f32)
        # algorithmic work per launch (SURVEY 8d accounting):
        #   fwd_first : 2*B*d0*d1 FLOP; reads A_0 (B*d0) + W_0 (d0*d1), writes A_1 (B*d1)
        #   middle    : 2*B*2*(P - d0 d1) FLOP; reads W_1.. once per use (fwd + bwd), A_1, Y; writes A_2.., delta_1..
        #   grad      : 2*B*P FLOP; reads A_l, delta_{l+1} for every layer and W, V (2P); writes W, V (2P)
        #   (data-parallel path: the same kernel stores G instead -- P written, W and V untouched;
        #    the update is sgd_momentum_kernel after the all-reduce)
        dp = dist is not None
        grad_name = "grad_update(all layers, 784x300xB + ..., stores G; update after the all-reduce)" if dp \
            else "grad_update(all layers, 784x300xB + ...)"
        kernels = {
            "fwd_first(128x784x300)": (fwd_us, 2.0 * BATCH * DIMS[0] * DIMS[1],
                                       e4 * (BATCH * DIMS[0] + DIMS[0] * DIMS[1] + BATCH * DIMS[1])),
            "middle(fwd L2.. + softmax + bwd-data)": (mid_us, 2.0 * BATCH * 2 * P_mid,
                                                      e4 * (2 * P_mid + BATCH * (DIMS[1] + 2 * DIMS[-1] + 2 * sum(DIMS[1:])))),
            grad_name: (grad_us, 2.0 * BATCH * P_all,
                        e4 * ((1 if dp else 4) * P_all + BATCH * (sum(DIMS[:-1]) + sum(DIMS[1:])))),
        }
        step_us = dt / K * 1e6
        # roofline kernel: the one that moves the most bytes and FLOPs -- every layer's G = A^T.delta
        # with the momentum update fused.  Its arithmetic intensity (13 FLOP/B) is below the f32
        # ridge (157.3 TFLOP/s / 8 TB/s = 19.7 FLOP/B), so the bound that applies is HBM.
        name = grad_name
        us, flop, nbytes = kernels[name]
        ach = nbytes / (us * 1e-6) / 1e9 if us > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": None if dp else pmc_traffic("grad_update_kernel<true"),
                    "avg_launch_us": round(us, 3), "launches": grad_n,
                    "algorithmic_bytes_per_launch": nbytes, "flop_per_launch": flop,
                    "arithmetic_intensity_flop_per_byte": round(flop / nbytes, 2),
                    "mfma_frac": round(flop / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4) if us > 0 else None,
                    # SURVEY 8d, whole step: 6P - 2 d0 d1 FLOP per sample; bytes = weights (P + (P - d0 d1) + P + 5P)
                    # + activations B (d0 + 4 sum_{l>=1} d_l), 4 B each
                    "whole_step": {"flop": (6 * P_all - 2 * DIMS[0] * DIMS[1]) * BATCH,
                                   "algorithmic_bytes": e4 * (8 * P_all - DIMS[0] * DIMS[1] + BATCH * (DIMS[0] + 4 * sum(DIMS[1:]))),
                                   "us": round(step_us, 3),
                                   "mfma_frac": round((6 * P_all - 2 * DIMS[0] * DIMS[1]) * BATCH / (step_us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                                   "hbm_frac": round(e4 * (8 * P_all - DIMS[0] * DIMS[1] + BATCH * (DIMS[0] + 4 * sum(DIMS[1:]))) / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                    "other": {k: {"avg_us": round(v[0], 3), "flop": v[1], "algorithmic_bytes": v[2],
                                  "tflops": round(v[1] / (v[0] * 1e-6) / 1e12, 3) if v[0] > 0 else None,
                                  "gbs": round(v[2] / (v[0] * 1e-6) / 1e9, 1) if v[0] > 0 else None,
                                  "share_of_step": round(v[0] / step_us, 3)} for k, v in kernels.items()}}
        if dist is None and not args.no_cpu_baseline:
            cpu = cpu_baseline()

    lockstep = None
    if dist is not None:
        # every rank applied the same all-reduced gradient: the replicas must hold identical weights
        lockstep = stepper.replicas_in_lockstep(torch, device="cuda" if args.backend == "nccl" else "cpu")
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        total = K * BATCH * world
        line = {
            "metric": "training samples/sec, 784-300-100-10 MLP batch 128",
            "value": round(total / dt, 1), "unit": "samples/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "784-300-100-10 SoftmaxCrossEntropyNeuralNet gradientStep, fp32, "
                                   "batch 128 per GPU (BASELINE configs[1])",
                       "global_batch": BATCH * world, "parallelism": "dp%d" % world,
                       "dp_mode": (None if dist is None else ("hipGraph replay of 64 steps" if graphed is not None else "eager")),
                       "dp_replicas_identical": lockstep,
                       "step": STEP, "momentum": MOMENTUM, "inner_activation": "leaky_relu"},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
