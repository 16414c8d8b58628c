#!/usr/bin/env python3
"""NeuralNetTrainer.train (exact epoch sampler + gnn_mlp_train_sampled) against gnn_mlp_train_range on the SAME
resident data set: what the reference's sampling loop (NNT:82-85) costs over stepping through contiguous batches."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_amd  # noqa: E402
from gnn_amd import trainer as tr  # noqa: E402

dims, n, B, steps = [784, 300, 100, 10], 6000, 128, 3000
rng = np.random.default_rng(0)
X = rng.random((n, 784)) * (rng.random((n, 784)) < 0.19)
Y = np.eye(10)[rng.integers(0, 10, n)]
for dtype, name in ((gnn_amd.DTYPE_F32, "f32"), (gnn_amd.DTYPE_BF16, "bf16")):
    net = gnn_amd.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B, dtype=dtype)
    t = tr.NeuralNetTrainer(X, Y, net)
    t.train(200, 0.0125, B, 0.9)
    net.synchronize()
    t0 = time.perf_counter()
    t.train(steps, 0.0125, B, 0.9)
    net.synchronize()
    dt = time.perf_counter() - t0
    net.train_range(0, B, 200, 0.0125, 0.9)
    net.synchronize()
    t1 = time.perf_counter()
    net.train_range(0, B, steps, 0.0125, 0.9)
    net.synchronize()
    dr = time.perf_counter() - t1
    print("%s train_sampled: %.2f us/step (%.3g samples/s); train_range on the same data: %.2f us/step"
          % (name, dt / steps * 1e6, steps * B / dt, dr / steps * 1e6))
    net.close()
