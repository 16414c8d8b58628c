#!/usr/bin/env python3
"""Development bench: generic (sizes from kernel arguments) vs run-time instantiated (hiprtc) vs
prebuilt kernels of the fused path, per shape."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_amd

for dims, B in (([784, 256, 64, 10], 128), ([784, 300, 100, 10], 128), ([100, 64, 48, 32, 10], 64)):
    rng = np.random.default_rng(0)
    X = rng.random((B * 8, dims[0])); Y = np.eye(dims[-1])[rng.integers(0, dims[-1], B * 8)]
    for mode in ("runtime", "jit", "prebuilt-or-auto"):
        os.environ["GNN_MLP_JIT"] = "0" if mode == "runtime" else "1"
        os.environ["GNN_MLP_STATIC"] = "1" if mode == "prebuilt-or-auto" else "0"
        net = gnn_amd.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
        net.upload_dataset(X, Y)
        t0 = time.perf_counter()
        if mode == "jit":
            net.specialize()
        tj = time.perf_counter() - t0
        net.train_range(0, B, 200, 0.0125, 0.9); net.synchronize()
        t0 = time.perf_counter(); net.train_range(0, B, 2000, 0.0125, 0.9); net.synchronize(); dt = time.perf_counter() - t0
        print("%s B=%d %-16s specialization=%d (%.2f s): %.2f us/step" % (
            "-".join(map(str, dims)), B, mode, net.specialization, tj, dt / 2000 * 1e6), flush=True)
