#!/usr/bin/env python3
"""Development bench over the other BASELINE.json configs (not the driver's contract, which is
bench.py): steps/s, samples/s and FLOP rate of one gradientStep on synthetic data in HBM."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_amd

CONFIGS = {
    "1": ([784, 100, 50, 10], 32),
    "2": ([784, 300, 100, 10], 128),
    "4": ([4096, 2048, 2048, 1024], 512),
    "5": ([784, 1024, 1024, 1024, 10], 256),
}

def main():
    args = sys.argv[1:]
    dtype = gnn_amd.DTYPE_BF16 if "bf16" in args else gnn_amd.DTYPE_F32
    which = [a for a in args if a != "bf16"] or ["1", "2", "5", "4"]
    for key in which:
        dims, B = CONFIGS[key]
        rng = np.random.default_rng(0)
        nb = 8
        X = rng.random((B * nb, dims[0])) * (rng.random((B * nb, dims[0])) < 0.19)
        Y = np.eye(dims[-1])[rng.integers(0, dims[-1], B * nb)]
        net = gnn_amd.SoftmaxCrossEntropyNeuralNet(dims, dtype=dtype, max_batch=B)
        if dims[0] > 1000:
            net.set_weights(net.get_weights() * 0.05)
        net.upload_dataset(X, Y)
        steps = 200 if key in ("4", "5") else 2000
        net.train_range(0, B, 20, 0.0125, 0.9); net.synchronize()
        t0 = time.perf_counter()
        net.train_range(0, B, steps, 0.0125, 0.9); net.synchronize()
        dt = time.perf_counter() - t0
        P = sum(dims[l] * dims[l + 1] for l in range(len(dims) - 1))
        flop = (6 * P - 2 * dims[0] * dims[1]) * B
        print(("bf16 " if dtype else "f32  ") + "config %s %s B=%d: %.2f us/step, %.3g samples/s, %.2f TFLOP/s (%.1f%% of 157.3 fp32 MFMA)" % (
            key, "-".join(map(str, dims)), B, dt / steps * 1e6, steps * B / dt, flop / (dt / steps) / 1e12,
            100 * flop / (dt / steps) / 157.3e12), flush=True)
        net.close()

if __name__ == "__main__":
    main()
