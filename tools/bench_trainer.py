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

# The loop MNISTTrainer actually runs (MT:150 passes a ProgressBar AND an observer, NNT:68-72): gradientStep + validate(1 % of the
# data) per iteration, at MNIST's size (60 000 rows: 601 validation rows), on a handle sized for training batches of 128.
import io
n2 = 60000
X2 = rng.random((n2, 784), dtype=np.float32) * (rng.random((n2, 784), dtype=np.float32) < 0.19)
Y2 = np.eye(10, dtype=np.float32)[rng.integers(0, 10, n2)]
net = gnn_amd.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
t = tr.NeuralNetTrainer(X2, Y2, net)
obs = io.StringIO()
t.train(200, 0.0125, B, 0.9, observer=obs)
net.synchronize()
it = 2000
t0 = time.perf_counter()
t.train(it, 0.0125, B, 0.9, observer=obs)
net.synchronize()
dt = time.perf_counter() - t0
print("observed loop (step + validate(601 rows) per iteration, 60 000 rows resident): %.2f us/iteration (%.3g training samples/s)" % (dt / it * 1e6, it * B / dt))
t0 = time.perf_counter()
hits = net.count_hits_range(0, n2)
dt = time.perf_counter() - t0
print("testOnTrainingData over the 60 000 rows on the same handle (max_batch 128): %.2f ms, %.3g rows/s" % (dt * 1e3, n2 / dt))
net.close()
