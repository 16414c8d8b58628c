"""GeneralNeuralNet (GNN:183-221 forward, GNN:230-242 loss, GNN:251-307 backward with loss'(f_last(z))*f_last'(z),
GNN:317-366 step) at the shapes BASELINE.json names -- 784-300-100-10 at batch 128 and 784-100-50-10 at batch 32 --
through the C ABI against the fp64 oracle built with out_kind = OUT_ACT_LOSS: every entry point of the NeuralNet
interface, both step forms (host batch; device-resident ranges = the two-launch path with the PREBUILT static row-block
instance, csrc/launch_small_gnn.hip), f32 and bf16, the element-wise output branch of the large-output GEMM path
(>= 1024 outputs), and the in-library data-parallel form.  PARITY UNPINNED BY THE REFERENCE (tests/test_oracle.py pins
the oracle itself).

The two nets: sigmoid / sigmoid / half-squared loss is the net of the reference's doc/backprop.pdf; leaky-ReLU inner
(MT:234-235) with a sigmoid output is the shipped inner pair under GeneralNeuralNet.  With leaky ReLU the Random(1) logits
reach +-100 and sigmoid(z) is exactly 1.0f for z > 17 while fp64 still separates such outputs: labels are compared where
the fp64 top-2 OUTPUT margin exceeds 1e-4, and the leaky nets run on weights scaled by 0.1 so that most rows
have such a margin (asserted).

Tolerances (f32 MFMA vs fp64 serial loops; as tests/test_parity_gpu.py): outputs 2e-5 abs (element-wise, in [0, 1]);
loss 1e-4 rel + 1e-5 abs; gradients 3e-5 of the layer's max |G|; weights / momentum 2e-6 per step."""
import os

import numpy as np
import pytest

from tests import np_oracle

pytestmark = pytest.mark.gpu

LEAKY, SIGMOID, TANH, RELU, IDENT = range(5)
W_ATOL = 2e-6

SHAPES = [([784, 300, 100, 10], 128), ([784, 100, 50, 10], 32)]
PAIRS = [(SIGMOID, SIGMOID), (LEAKY, SIGMOID)]


def mnist_like(dims, n, seed):
    rng = np.random.default_rng(seed)
    X = rng.random((n, dims[0])) * (rng.random((n, dims[0])) < 0.19)
    Y = np.eye(dims[-1])[rng.integers(0, dims[-1], n)]
    return X, Y


def start_weights(net, inner):
    """Random(1) weights as created (sigmoid inner: logits within +-3.4); scaled by 0.1 for leaky ReLU (see the header)."""
    w = net.get_weights()
    if inner == LEAKY:
        net.set_weights(w * 0.1)
        w = net.get_weights()
    return w


def per_layer(flat, dims):
    return np_oracle.split(np.asarray(flat), dims)


def forced_path():
    return bool(os.environ.get("GNN_MLP_PATH") or os.environ.get("GNN_MLP_CHAIN") == "0" or os.environ.get("GNN_MLP_ROWBLOCK") == "0"
                or os.environ.get("GNN_MLP_STATIC") == "0")


@pytest.mark.parametrize("inner,last", PAIRS, ids=["sigmoid-sigmoid", "leaky-sigmoid"])
@pytest.mark.parametrize("dims,B", SHAPES, ids=["784-300-100-10", "784-100-50-10"])
def test_general_net_interface_at_baseline_shapes(gnn, oracle_mod, dims, B, inner, last):
    X, Y = mnist_like(dims, B, seed=11)
    net = gnn.GeneralNeuralNet(dims, inner_act=inner, last_act=last, max_batch=B)
    ref = oracle_mod.OracleNet(dims, out_kind=oracle_mod.OUT_ACT_LOSS, inner_act=inner, last_act=last)
    ref.set_alloc_per_sample(0)
    # A2: the same Random(1) draws as the softmax class (GNN:158-175)
    assert np.array_equal(net.get_weights(), ref.get_weights().astype(np.float32).astype(np.float64))
    w = start_weights(net, inner)
    ref.set_weights(w)
    if not forced_path():
        assert net.specialization == 1, "GeneralNeuralNet of a BASELINE shape must take the prebuilt static kernels"
        assert net.step_launches == 2 and net.rowblock_state == 2
    # propagate (GNN:183-221): element-wise last activation, one output row per input row
    out, outr = net.propagate(X), ref.propagate(X)
    assert out.shape == outr.shape == (B, dims[-1])
    assert np.abs(out - outr).max() <= 2e-5
    o1 = net.propagate(X[0])                                   # the reference's own call shape (NN:16)
    assert o1.shape == (dims[-1],) and np.abs(o1 - outr[0]).max() <= 2e-5
    # calculateLoss (GNN:230-242)
    l, lr = net.calculateLoss(X, Y), ref.calculate_loss(X, Y)
    assert np.all(np.abs(l - lr) <= 1e-4 * np.abs(lr) + 1e-5)
    assert abs(net.calculateLoss(X[0], Y[0]) - lr[0]) <= 1e-4 * abs(lr[0]) + 1e-5
    # argmax (MT:166-168 on GeneralNeuralNet outputs), bit-exact where the fp64 top-2 OUTPUT margin exceeds 1e-4
    s = np.sort(outr, axis=1)
    safe = (s[:, -1] - s[:, -2]) > 1e-4
    assert safe.mean() >= 0.9, "fixture: too many near-ties (%d of %d rows safe)" % (safe.sum(), B)
    assert np.array_equal(net.argmax(X)[safe], ref.argmax(X)[safe])
    # calculateWeightGradient (GNN:251-307): one sample (the reference's call) and the batch sum, every element
    g1 = net.calculateWeightGradient(X[1], Y[1])
    g1r = per_layer(ref.calculate_weight_gradient(X[1], Y[1]), dims)
    gb = net.calculateWeightGradient(X, Y)
    gbr = per_layer(sum(ref.calculate_weight_gradient(X[b], Y[b]) for b in range(B)), dims)
    for l_ in range(len(dims) - 1):
        assert np.abs(g1[l_] - g1r[l_]).max() <= 3e-5 * np.abs(g1r[l_]).max() + 1e-10, "layer %d (one sample)" % l_
        assert np.abs(gb[l_] - gbr[l_]).max() <= 3e-5 * np.abs(gbr[l_]).max() + 1e-10, "layer %d (batch)" % l_
    # gradientStep (GNN:317-366), five host batches: weights AND momentum
    steps = 5
    for k in range(steps):
        Xs, Ys = mnist_like(dims, B, seed=200 + k)
        net.gradientStep(Xs, 0.1, 0.9, False, expected=Ys)
        ref.gradient_step(Xs, Ys, 0.1, 0.9)
    assert net.time == steps == ref.time
    assert np.abs(net.get_weights() - ref.get_weights()).max() <= W_ATOL * steps
    assert np.abs(net.get_momentum() - ref.get_momentum()).max() <= W_ATOL * steps


@pytest.mark.parametrize("inner,last", PAIRS, ids=["sigmoid-sigmoid", "leaky-sigmoid"])
@pytest.mark.parametrize("dims,B", SHAPES, ids=["784-300-100-10", "784-100-50-10"])
def test_general_net_two_launch_training_loop(gnn, oracle_mod, dims, B, inner, last):
    """The loop NNT:82-85 on device-resident rows (gnn_mlp_train_range): the two-launch step with the static GeneralNeuralNet
    row-block instance, against the oracle stepping on the same batches, and bitwise against the same steps taken one call
    at a time; then loss / argmax of resident rows."""
    nb, steps = 3, 7
    X, Y = mnist_like(dims, B * nb, seed=31)
    a = gnn.GeneralNeuralNet(dims, inner_act=inner, last_act=last, max_batch=B)
    b = gnn.GeneralNeuralNet(dims, inner_act=inner, last_act=last, max_batch=B)
    ref = oracle_mod.OracleNet(dims, out_kind=oracle_mod.OUT_ACT_LOSS, inner_act=inner, last_act=last)
    ref.set_alloc_per_sample(0)
    w = start_weights(a, inner)
    b.set_weights(w); ref.set_weights(w)
    a.upload_dataset(X, Y); b.upload_dataset(X, Y)
    a.train_range(0, B, steps, 0.1, 0.9)
    for k in range(steps):
        sl = slice((k % nb) * B, (k % nb + 1) * B)
        b.gradient_step_range((k % nb) * B, B, 0.1, 0.9)
        ref.gradient_step(X[sl], Y[sl], 0.1, 0.9)
    assert a.time == steps == b.time
    assert np.array_equal(a.get_weights(), b.get_weights())           # however the chain is cut: the same bits
    assert np.array_equal(a.get_momentum(), b.get_momentum())
    assert np.abs(a.get_weights() - ref.get_weights()).max() <= W_ATOL * steps
    assert np.abs(a.get_momentum() - ref.get_momentum()).max() <= W_ATOL * steps
    lr = ref.calculate_loss(X[:B], Y[:B])
    assert np.all(np.abs(a.loss_range(0, B) - lr) <= 1e-4 * np.abs(lr) + 1e-5)
    outr = ref.propagate(X[:B])
    s = np.sort(outr, axis=1)
    safe = (s[:, -1] - s[:, -2]) > 1e-4
    assert np.array_equal(a.argmax_range(0, B)[safe], ref.argmax(X[:B])[safe])


@pytest.mark.parametrize("poison", ["last_column", "first_column"])
def test_general_net_argmax_nan_rules_at_baseline_shape(gnn, oracle_mod, poison):
    """MT:166-168 on element-wise outputs at 784-300-100-10 (the static instances' output rule): a NaN at an index > 0 is
    never selected (`NaN >= x` is false), a NaN at index 0 is sticky (`x >= NaN` is false for every later x)."""
    dims, B = [784, 300, 100, 10], 16
    net = gnn.GeneralNeuralNet(dims, inner_act=SIGMOID, last_act=IDENT, max_batch=B)
    ref = oracle_mod.OracleNet(dims, out_kind=oracle_mod.OUT_ACT_LOSS, inner_act=SIGMOID, last_act=IDENT)
    w = net.get_weights()
    w[-1 if poison == "last_column" else -dims[-1]] = np.nan
    net.set_weights(w); ref.set_weights(w)
    X, Y = mnist_like(dims, B, seed=5)
    want = ref.argmax(X)
    assert np.all(want == 0) if poison == "first_column" else np.all(want != dims[-1] - 1)
    assert np.array_equal(net.argmax(X), want)
    net.upload_dataset(X, Y)
    assert np.array_equal(net.argmax_range(0, B), want)


@pytest.mark.parametrize("inner,last", PAIRS, ids=["sigmoid-sigmoid", "leaky-sigmoid"])
@pytest.mark.parametrize("dims,B", SHAPES, ids=["784-300-100-10", "784-100-50-10"])
def test_general_net_bf16_against_bf16_oracle(gnn, dims, B, inner, last):
    """GNN_DTYPE_BF16 on GeneralNeuralNet: every GEMM operand rounded to bf16, f32 accumulate / masters / momentum, against
    tests/np_oracle.py's *_bf16(out_kind = 1) -- the same roundings in fp64 (tolerances of tests/test_bf16_gpu.py)."""
    X, Y = mnist_like(dims, B, seed=41)
    net = gnn.GeneralNeuralNet(dims, inner_act=inner, last_act=last, dtype=gnn.DTYPE_BF16, max_batch=B)
    w0 = start_weights(net, inner)
    Ws = np_oracle.split(w0, dims)
    X32 = X.astype(np.float32).astype(np.float64)
    Z, A, out = np_oracle.forward_bf16(Ws, X32, inner, 1, last)
    assert np.abs(net.propagate(X) - out).max() <= 5e-3
    lr = (0.5 * (out - Y) ** 2).sum(axis=1)
    assert np.all(np.abs(net.calculateLoss(X, Y) - lr) <= 1e-2 * np.abs(lr) + 1e-3)
    g = net.calculateWeightGradient(X, Y)
    gq = per_layer(np_oracle.gradient_bf16(Ws, X32, Y, inner, 1, last), dims)
    for l_ in range(len(dims) - 1):
        assert np.abs(g[l_] - gq[l_]).max() <= 4e-3 * np.abs(gq[l_]).max() + 1e-7, "layer %d" % l_
    # three steps, host batches and then the two-launch loop on resident rows
    w, v = w0.copy(), np.zeros_like(w0)
    for k in range(3):
        Xs, Ys = mnist_like(dims, B, seed=50 + k)
        net.gradientStep(Xs, 0.1, 0.9, False, expected=Ys)
        w, v = np_oracle.gradient_step_bf16(w, v, dims, Xs.astype(np.float32).astype(np.float64), Ys, 0.1, 0.9, inner, 1, last)
    assert np.abs(net.get_weights() - w).max() <= 2e-4
    Xd, Yd = mnist_like(dims, 2 * B, seed=60)
    net.upload_dataset(Xd, Yd)
    net.train_range(0, B, 2, 0.1, 0.9)
    Xd32 = Xd.astype(np.float32).astype(np.float64)
    for k in range(2):
        w, v = np_oracle.gradient_step_bf16(w, v, dims, Xd32[k * B:(k + 1) * B], Yd[k * B:(k + 1) * B], 0.1, 0.9, inner, 1, last)
    assert net.time == 5
    assert np.abs(net.get_weights() - w).max() <= 4e-4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_general_net_elementwise_output_of_1024_columns(gnn, dtype):
    """The element-wise output rule at GEMM size: 4096-2048-2048-1024 (BASELINE configs[3]'s net) as a GeneralNeuralNet at
    batch 64 -- output_layer_kernel's last_act / loss' branch (csrc/kernels.h) over 1024 columns after the per-layer MFMA GEMMs --
    every output, loss, and every element of the whole batch's gradient against the fp64 matrix form (GNN:215-218, 236-239,
    267-271, 279-300), then one update step (GNN:344-366)."""
    dims, B = [4096, 2048, 2048, 1024], 64
    bf = dtype == "bf16"
    rng = np.random.default_rng(9)
    X = rng.random((B, dims[0])) * (rng.random((B, dims[0])) < 0.19)
    Y = rng.random((B, dims[-1]))                      # a general target (GeneralNeuralNet is not tied to one-hot rows)
    net = gnn.GeneralNeuralNet(dims, inner_act=SIGMOID, last_act=SIGMOID, dtype=gnn.DTYPE_BF16 if bf else gnn.DTYPE_F32, max_batch=B)
    net.set_weights(net.get_weights() * 0.05)          # sum of 2048 sigmoids x U[-.5,.5): keeps the outputs off saturation
    w = net.get_weights()
    Ws = np_oracle.split(w, dims)
    X32 = X.astype(np.float32).astype(np.float64)
    if bf:
        _, _, outr = np_oracle.forward_bf16(Ws, X32, SIGMOID, 1, SIGMOID)
        gr = np_oracle.gradient_bf16(Ws, X32, Y, SIGMOID, 1, SIGMOID)
    else:
        _, outr = np_oracle.forward(Ws, X32, SIGMOID, 1, SIGMOID)
        gr = np_oracle.gradient(Ws, X32, Y, SIGMOID, 1, SIGMOID)
    out = net.propagate(X)
    assert np.abs(out - outr).max() <= (5e-3 if bf else 2e-5)
    lr = (0.5 * (outr - Y) ** 2).sum(axis=1)
    assert np.all(np.abs(net.calculateLoss(X, Y) - lr) <= (1e-2 if bf else 1e-4) * np.abs(lr) + 1e-5)
    s = np.sort(outr, axis=1)
    safe = (s[:, -1] - s[:, -2]) > (2e-2 if bf else 1e-4)
    assert np.array_equal(net.argmax(X)[safe], outr.argmax(axis=1)[safe])
    g = net.calculateWeightGradient(X, Y)
    grl = per_layer(gr, dims)
    for l_ in range(len(dims) - 1):
        assert np.abs(g[l_] - grl[l_]).max() <= (4e-3 if bf else 3e-5) * np.abs(grl[l_]).max() + 1e-9, "layer %d" % l_
    net.gradientStep(X, 0.1, 0.9, False, expected=Y)
    step = np_oracle.gradient_step_bf16 if bf else np_oracle.gradient_step
    w_ref, v_ref = step(w, np.zeros_like(w), dims, X32, Y, 0.1, 0.9, SIGMOID, 1, SIGMOID)
    gmax = max(np.abs(x).max() for x in grl)
    tol = 0.1 / B * (4e-3 if bf else 3e-5) * gmax + 2e-7 * np.abs(w).max() + 1e-9
    assert np.abs(net.get_momentum() - v_ref).max() <= tol
    assert np.abs(net.get_weights() - w_ref).max() <= tol


@pytest.mark.parametrize("reducer", ["direct", "direct_rs"])
def test_general_net_two_replica_data_parallel(gnn, oracle_mod, reducer):
    """gnn_mlp_dp_* on a GeneralNeuralNet: two replicas (sharing the one device there is), the batch's rows dealt in two
    blocks, the partial gradients summed in rank order, the identical update with batchSize = B (GNN:327-353) -- replicas
    bitwise identical, equal to the oracle stepping on the whole batch."""
    dims, B, nb, steps = [784, 300, 100, 10], 128, 2, 4
    X, Y = mnist_like(dims, B * nb, seed=71)
    net = gnn.DataParallelNeuralNet(dims, devices=[0, 0], out_kind=gnn.OUT_ACT_LOSS, inner_act=SIGMOID, last_act=SIGMOID,
                                    max_batch=B, reducer=gnn.REDUCE_DIRECT_RS if reducer == "direct_rs" else gnn.REDUCE_DIRECT)
    ref = oracle_mod.OracleNet(dims, out_kind=oracle_mod.OUT_ACT_LOSS, inner_act=SIGMOID, last_act=SIGMOID)
    ref.set_alloc_per_sample(0)
    net.upload_dataset(X, Y)
    net.train_range(0, B, steps, 0.1, 0.9)
    net.synchronize()
    for k in range(steps):
        sl = slice((k % nb) * B, (k % nb + 1) * B)
        ref.gradient_step(X[sl], Y[sl], 0.1, 0.9)
    assert net.time == steps and net.replicas_identical()
    assert np.abs(net.get_weights() - ref.get_weights()).max() <= W_ATOL * steps
    # a host batch through the same handle (NN:51), ragged split 65 / 64
    Xh, Yh = mnist_like(dims, 129 - 1, seed=72)
    net.gradientStep(Xh[:127], 0.1, 0.9, False, expected=Yh[:127])
    ref.gradient_step(Xh[:127], Yh[:127], 0.1, 0.9)
    assert net.replicas_identical()
    assert np.abs(net.get_weights() - ref.get_weights()).max() <= W_ATOL * (steps + 1)
    assert np.abs(net.propagate(X[:B]) - ref.propagate(X[:B])).max() <= 2e-5
