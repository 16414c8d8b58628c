#!/usr/bin/env python3
"""f32 leg of tools/bench_trainer.py alone (for a kernel trace): NeuralNetTrainer.train on sampled batches."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_amd  # noqa: E402
from gnn_amd import trainer as tr  # noqa: E402

dims, n, B, steps = [784, 300, 100, 10], 6000, 128, 2000
rng = np.random.default_rng(0)
X = rng.random((n, 784)) * (rng.random((n, 784)) < 0.19)
Y = np.eye(10)[rng.integers(0, 10, n)]
net = gnn_amd.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
t = tr.NeuralNetTrainer(X, Y, net)
t.train(200, 0.0125, B, 0.9)
net.synchronize()
t0 = time.perf_counter()
t.train(steps, 0.0125, B, 0.9)
net.synchronize()
dt = time.perf_counter() - t0
print("f32 train_sampled: %.2f us/step (rowblock_state %d)" % (dt / steps * 1e6, net.rowblock_state))
