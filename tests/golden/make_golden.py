"""Writes tests/golden/golden.json from the CPU oracle (oracle/mlp_oracle.c).

The reference (Java) holds no fixtures for this path and cannot run here (no JDK), so these
vectors are the ORACLE's outputs, not the reference's: they pin the oracle against regressions
(tests/test_oracle.py) and give the GPU tests fixed targets (tests/test_golden_gpu.py).
Inputs are not stored: they are regenerated from java.util.Random seeds
(oracle.synthetic_batch), weights from Random(1) as in SCE:111,149.

    python -m tests.golden.make_golden        # rewrites golden.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LEAKY, SIGMOID, TANH, RELU, IDENT = range(5)

CASES = [
    # name, dims, out_kind, inner, last, B, seed, keep, steps, step, momentum
    ("sce_leaky_784-100-50-10_b32", [784, 100, 50, 10], 0, LEAKY, IDENT, 32, 101, 0.19, 10, 0.01, 0.9),
    ("sce_leaky_784-300-100-10_b128", [784, 300, 100, 10], 0, LEAKY, IDENT, 128, 202, 1.0, 3, 0.0125, 0.9),
    ("sce_leaky_784-300-100-10_b16_mnistlike", [784, 300, 100, 10], 0, LEAKY, IDENT, 16, 303, 0.19, 10, 0.0125, 0.9),
    ("sce_sigmoid_784-300-100-10_b16", [784, 300, 100, 10], 0, SIGMOID, IDENT, 16, 404, 1.0, 5, 0.0125, 0.9),
    ("gnn_sigmoid_30-21-18-5_b13", [30, 21, 18, 5], 1, SIGMOID, SIGMOID, 13, 505, 1.0, 10, 0.1, 0.9),
    ("sce_tanh_20-17-33-7_b19", [20, 17, 33, 7], 0, TANH, IDENT, 19, 606, 1.0, 10, 0.05, 0.5),
]


def layer_slices(dims):
    off = 0
    for l in range(len(dims) - 1):
        n = dims[l] * dims[l + 1]
        yield l, slice(off, off + n)
        off += n


def summarize(flat, dims):
    out = {}
    for l, sl in layer_slices(dims):
        v = flat[sl]
        out["L%d" % l] = {"head": [float(x) for x in v[:8]], "tail": [float(x) for x in v[-8:]],
                          "sum": float(v.sum()), "abs_sum": float(np.abs(v).sum()),
                          "abs_max": float(np.abs(v).max())}
    return out


def make_case(spec):
    from oracle import oracle
    name, dims, out_kind, inner, last, B, seed, keep, steps, step, mom = spec
    X, Y = oracle.synthetic_batch(dims, B, seed, keep)
    if out_kind == 1:
        Y = np.empty((B, dims[-1]))
        oracle.lib().oracle_fill_uniform(seed + 2, Y.size, 1.0, oracle._dp(Y))
    net = oracle.OracleNet(dims, out_kind=out_kind, inner_act=inner, last_act=last)
    net.set_alloc_per_sample(0)
    case = {"name": name, "dims": dims, "out_kind": out_kind, "inner_act": inner, "last_act": last,
            "B": B, "seed": seed, "keep": keep, "steps": steps, "step": step, "momentum": mom}
    out = net.propagate(X)
    logits = net.logits(X)
    srt = np.sort(logits if out_kind == 0 else out, axis=1)
    case["output_first4"] = [float(v) for v in out[:4].ravel()]
    case["loss"] = [float(v) for v in net.calculate_loss(X, Y)]
    case["labels"] = [int(v) for v in net.argmax(X)]
    case["min_top2_margin"] = float((srt[:, -1] - srt[:, -2]).min())
    g = sum(net.calculate_weight_gradient(X[b], Y[b]) for b in range(B))
    case["gradient"] = summarize(g, dims)
    for s in range(steps):
        Xs, Ys = oracle.synthetic_batch(dims, B, seed + 10 * (s + 1), keep)
        if out_kind == 1:
            oracle.lib().oracle_fill_uniform(seed + 10 * (s + 1) + 2, Ys.size, 1.0, oracle._dp(Ys))
        net.gradient_step(Xs, Ys, step, mom)
    case["weights_after"] = summarize(net.get_weights(), dims)
    case["momentum_after"] = summarize(net.get_momentum(), dims)
    return case


def case_inputs(case):
    """(X, Y) of the propagate/loss/gradient part and a generator of the per-step batches."""
    from oracle import oracle
    dims, B, seed, keep = case["dims"], case["B"], case["seed"], case["keep"]
    X, Y = oracle.synthetic_batch(dims, B, seed, keep)
    if case["out_kind"] == 1:
        oracle.lib().oracle_fill_uniform(seed + 2, Y.size, 1.0, oracle._dp(Y))

    def batches():
        for s in range(case["steps"]):
            Xs, Ys = oracle.synthetic_batch(dims, B, seed + 10 * (s + 1), keep)
            if case["out_kind"] == 1:
                oracle.lib().oracle_fill_uniform(seed + 10 * (s + 1) + 2, Ys.size, 1.0, oracle._dp(Ys))
            yield Xs, Ys
    return X, Y, batches


def generate():
    return {"format": 1,
            "note": "oracle outputs (fp64 C restatement of the Java loops); parity unpinned by the reference",
            "cases": [make_case(c) for c in CASES]}


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json")
    with open(path, "w") as f:
        json.dump(generate(), f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")
