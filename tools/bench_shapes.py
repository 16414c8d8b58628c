#!/usr/bin/env python3
"""Development bench: step time of assorted net shapes (which path each takes)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_amd
shapes = [([784, 512, 256, 10], 128), ([784, 1024, 10], 128), ([3072, 512, 128, 10], 128), ([784, 300, 100, 10], 512),
          ([784, 300, 100, 10], 32), ([784, 64, 10], 128), ([256, 256, 256, 256, 10], 128)]
for dims, B in shapes:
    rng = np.random.default_rng(0)
    X = rng.random((B * 4, dims[0])) * (rng.random((B * 4, dims[0])) < 0.3); Y = np.eye(dims[-1])[rng.integers(0, dims[-1], B * 4)]
    net = gnn_amd.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    net.set_weights(net.get_weights() * 0.1)
    net.upload_dataset(X, Y)
    net.train_range(0, B, 100, 0.0125, 0.9); net.synchronize()
    t0 = time.perf_counter(); net.train_range(0, B, 1000, 0.0125, 0.9); net.synchronize(); dt = time.perf_counter() - t0
    P = sum(dims[l] * dims[l + 1] for l in range(len(dims) - 1))
    print("%-24s B=%-4d specialization=%d launches=%d  %.2f us/step  %.3g samples/s  %.2f TFLOP/s" % (
        "-".join(map(str, dims)), B, net.specialization, net.step_launches, dt / 1000 * 1e6, 1000 * B / dt,
        (6 * P - 2 * dims[0] * dims[1]) * B / (dt / 1000) / 1e12), flush=True)
