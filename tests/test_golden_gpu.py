"""HIP path vs the committed golden fixtures (tests/golden/golden.json: oracle outputs, inputs
regenerated from java.util.Random seeds -- the generator is oracle code, used here only to
rebuild the INPUTS; the expected values come from the JSON file).

Tolerances as in tests/test_parity_gpu.py; checksums (sum over a layer) use the layer's
abs_sum * 2e-6 as the scale."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

with open(os.path.join(HERE, "golden", "golden.json")) as _f:
    GOLD = json.load(_f)


def check_summary(flat, dims, gold, rel, atol, what):
    off = 0
    for l in range(len(dims) - 1):
        n = dims[l] * dims[l + 1]
        v = flat[off:off + n]
        off += n
        g = gold["L%d" % l]
        tol = rel * g["abs_max"] + atol
        assert np.abs(v[:8] - np.array(g["head"])).max() <= tol, (what, l, "head")
        assert np.abs(v[-8:] - np.array(g["tail"])).max() <= tol, (what, l, "tail")
        assert abs(v.sum() - g["sum"]) <= rel * g["abs_sum"] + atol * n, (what, l, "sum")
        assert abs(np.abs(v).sum() - g["abs_sum"]) <= rel * g["abs_sum"] + atol * n, (what, l, "abs_sum")


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_golden_case(gnn, case):
    from tests.golden import make_golden
    dims, B = case["dims"], case["B"]
    X, Y, batches = make_golden.case_inputs(case)
    if case["out_kind"] == 0:
        net = gnn.SoftmaxCrossEntropyNeuralNet(dims, inner_act=case["inner_act"], max_batch=B)
    else:
        net = gnn.GeneralNeuralNet(dims, inner_act=case["inner_act"], last_act=case["last_act"], max_batch=B)
    out = net.propagate(X)
    assert np.abs(out[:4].ravel() - np.array(case["output_first4"])).max() <= 2e-4
    loss = net.calculateLoss(X, Y)
    gl = np.array(case["loss"])
    assert np.all(np.abs(loss - gl) <= 2e-4 * np.abs(gl) + 2e-4)
    assert case["min_top2_margin"] > 1e-3
    assert list(net.argmax(X)) == case["labels"]          # bit-exact class labels
    g = net.calculateWeightGradient(X, Y)
    gflat = np.concatenate([g[l].ravel() for l in sorted(g)])
    check_summary(gflat, dims, case["gradient"], 3e-5, 1e-9, "gradient")
    for Xs, Ys in batches():
        net.gradientStep(Xs, case["step"], case["momentum"], False, expected=Ys)
    assert net.time == case["steps"]
    check_summary(net.get_weights(), dims, case["weights_after"], 0.0, 2e-6 * case["steps"], "weights")
    check_summary(net.get_momentum(), dims, case["momentum_after"], 0.0, 2e-6 * case["steps"], "momentum")
