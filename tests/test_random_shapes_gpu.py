"""Seeded sweep over net shapes and batch sizes: the kernels' shape-dependent logic (LDS plan, slab
staging with partly idle threads, K splits, column groups, ragged row blocks, the hybrid choices)
against the fp64 oracle, and the run-time instantiated kernel against the runtime-shape one
(bitwise).  PARITY UNPINNED BY THE REFERENCE (see test_parity_gpu.py); tolerances as there."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W_ATOL = 2e-6
P_ATOL = 2e-4


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    L = int(rng.integers(3, 7))
    dims = [int(rng.integers(1, 200))]
    dims += [int(rng.integers(1, 400 if i == 0 else 130)) for i in range(L - 2)]
    dims += [int(rng.integers(2, 40))]
    B = int(rng.integers(1, 70))
    inner = int(rng.integers(0, 4))
    return dims, B, inner


@pytest.mark.parametrize("seed", range(20))
def test_random_shape_against_oracle(gnn, oracle_mod, seed):
    dims, B, inner = random_case(seed)
    rng = np.random.default_rng(seed)
    X = rng.random((B, dims[0])) * (rng.random((B, dims[0])) < 0.5)
    Y = np.eye(dims[-1])[rng.integers(0, dims[-1], B)]
    net = gnn.SoftmaxCrossEntropyNeuralNet(dims, inner_act=inner, max_batch=B)
    ref = oracle_mod.OracleNet(dims, inner_act=inner)
    ref.set_alloc_per_sample(0)
    # small nets at the Random(1) init: keep logits moderate so that f32 softmax error stays in tolerance
    w = net.get_weights() * 0.5
    net.set_weights(w); ref.set_weights(w)
    assert np.abs(net.propagate(X) - ref.propagate(X)).max() <= P_ATOL, (dims, B, inner)
    g = net.calculateWeightGradient(X, Y)
    gr = sum(ref.calculate_weight_gradient(X[b], Y[b]) for b in range(B))
    off = 0
    for l in range(len(dims) - 1):
        n = dims[l] * dims[l + 1]
        grl = gr[off:off + n].reshape(dims[l], dims[l + 1]); off += n
        assert np.abs(g[l] - grl).max() <= 3e-5 * np.abs(grl).max() + 1e-9, (dims, B, inner, l)
    for s in range(2):
        net.gradientStep(X, 0.0125, 0.9, False, expected=Y)
        ref.gradient_step(X, Y, 0.0125, 0.9)
    assert np.abs(net.get_weights() - ref.get_weights()).max() <= 3 * W_ATOL, (dims, B, inner)
    assert np.abs(net.get_momentum() - ref.get_momentum()).max() <= 3 * W_ATOL, (dims, B, inner)
    lab = net.argmax(X)
    z = ref.propagate(X)
    srt = np.sort(z, axis=1)
    sure = (srt[:, -1] - srt[:, -2]) > 1e-3     # softmax is monotone: same order as the logits
    assert np.array_equal(lab[sure], ref.argmax(X)[sure]), (dims, B, inner)


@pytest.mark.parametrize("seed", range(0, 20, 3))
def test_random_shape_specialised_is_bitwise(gnn, seed):
    """The hiprtc instantiation of the row-block kernel for the drawn shape (compile-time plan and
    slab staging) equals the runtime-shape kernel bit for bit."""
    if os.environ.get("GNN_MLP_PATH") or os.environ.get("GNN_MLP_JIT") == "0":
        pytest.skip("path forced by the environment")
    dims, B, inner = random_case(seed)
    rng = np.random.default_rng(seed)
    X = rng.random((B, dims[0]))
    Y = np.eye(dims[-1])[rng.integers(0, dims[-1], B)]
    a = gnn.SoftmaxCrossEntropyNeuralNet(dims, inner_act=inner, max_batch=B)
    b = gnn.SoftmaxCrossEntropyNeuralNet(dims, inner_act=inner, max_batch=B)
    if a.specialization != 0:
        pytest.skip("prebuilt shape")
    if b.specialize() != 2:
        pytest.skip("this shape does not take the row-block path")
    a.upload_dataset(X, Y); b.upload_dataset(X, Y)
    for s in range(3):
        a.gradient_step_range(0, B, 0.0125, 0.9)
        b.gradient_step_range(0, B, 0.0125, 0.9)
    assert a.specialization == 0 and b.specialization == 2
    assert np.array_equal(a.get_weights(), b.get_weights()), (dims, B, inner)
    assert np.array_equal(a.propagate(X), b.propagate(X)), (dims, B, inner)
