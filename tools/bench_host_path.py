#!/usr/bin/env python3
"""PCIe-inclusive rate: gnn_mlp_gradient_step with HOST fp64 batches (the literal NeuralNet.gradientStep
call shape), i.e. 128 x (784 + 10) x 8 B crossing PCIe per step plus the fp64->f32 conversion."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_amd
dims, B = [784, 300, 100, 10], 128
rng = np.random.default_rng(0)
X = rng.random((B * 16, 784)); Y = np.eye(10)[rng.integers(0, 10, B * 16)]
net = gnn_amd.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
for s in range(50):
    net.gradientStep(X[(s % 16) * B:(s % 16 + 1) * B], 0.0125, 0.9, False, expected=Y[(s % 16) * B:(s % 16 + 1) * B])
net.synchronize()
n = 1000
t0 = time.perf_counter()
for s in range(n):
    r = (s % 16) * B
    net.gradientStep(X[r:r + B], 0.0125, 0.9, False, expected=Y[r:r + B])
net.synchronize()
dt = time.perf_counter() - t0
print("host-batch gradientStep: %.1f us/step, %.3g samples/s (PCIe + conversion inclusive)" % (dt / n * 1e6, n * B / dt))
