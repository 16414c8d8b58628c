"""GPU half of tests/test_reference_labels.py: the labels the reference holds (fixture tests/golden/reference_labels.json, the
first 2 048 of data/train-labels-idx1-ubyte) through the product's encoding and through a long training trajectory.

The IMAGE files are absent from the reference, so the pixels are synthetic: one sparse prototype per class plus noise, as
bytes -- every row carries the label the reference's file gives it.  What is pinned by reference-held data is therefore the
label side (format, order, encoding, which rows the sampler visits and what their targets are); the arithmetic of the
trajectory is checked against the ORACLE (parity unpinned by the reference, tests/test_oracle.py)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def labels():
    with open(os.path.join(HERE, "golden", "reference_labels.json")) as f:
        return np.array(json.load(f)["train"]["first_labels"], dtype=np.uint8)


def prototype_images(labels, seed=3, keep=0.19, noise=40):
    """One sparse byte pattern per class (19 % of the 784 pixels lit, MNIST's density) + per-row noise on the lit pixels and a
    few stray ones: separable, not trivially so."""
    rng = np.random.default_rng(seed)
    proto = rng.integers(96, 256, (10, 784)) * (rng.random((10, 784)) < keep)
    pix = proto[labels].astype(np.int64)
    lit = pix > 0
    pix = pix + lit * rng.integers(-noise, noise + 1, pix.shape)
    pix = pix + (~lit & (rng.random(pix.shape) < 0.02)) * rng.integers(0, 128, pix.shape)
    return np.clip(pix, 0, 255).astype(np.uint8)


def test_one_hot_upload_of_the_reference_labels(gnn, labels):
    """gnn_mlp_upload_dataset_u8: label byte -> one-hot 1.0 row on the GPU (MT:112-118), pixel -> (byte & 0xff) / 255.0 (MT:98),
    against the same rows encoded on the host in fp64 and uploaded as doubles: the same bits in HBM, hence the same loss of
    every row and the same weights after steps."""
    dims, B, N = [784, 100, 50, 10], 32, 512
    pix = prototype_images(labels[:N])
    a = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    b = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    a.upload_dataset_u8(pix, labels[:N])
    Y = np.zeros((N, 10)); Y[np.arange(N), labels[:N]] = 1.0           # MT:113-117
    b.upload_dataset(pix.astype(np.float64) / 255.0, Y)
    for first in range(0, N, B):
        assert np.array_equal(a.loss_range(first, B), b.loss_range(first, B))
    a.train_range(0, B, N // B, 0.01, 0.9); b.train_range(0, B, N // B, 0.01, 0.9)
    assert np.array_equal(a.get_weights(), b.get_weights())
    # and the label really is the target: moving one label moves that row's loss only
    c = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    lab2 = labels[:N].copy(); lab2[5] = (lab2[5] + 1) % 10
    c.upload_dataset_u8(pix, lab2)
    b2 = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    b2.upload_dataset_u8(pix, labels[:N])
    la, lc = b2.loss_range(0, B), c.loss_range(0, B)
    assert la[5] != lc[5] and np.array_equal(np.delete(la, 5), np.delete(lc, 5))


def test_long_trajectory_on_reference_labels(gnn, oracle_mod, labels):
    """BASELINE configs[0] (784-100-50-10, batch 32, leaky ReLU, step 0.01, momentum 0.9 -- the reference's own run,
    logs/trainLog.csv:3) for 300 iterations of NeuralNetTrainer.train (NNT:82-85) over 2 048 rows labelled by the reference's
    file: the HIP path (gnn_mlp_train_sampled: device-resident rows, the exact epoch sampler, the two-launch step) against
    the oracle stepping on the batches the ORACLE's sampler draws -- 4.7 epochs; 2 048 = 64 x 32, so the list refills between
    batches only and every batch has 32 distinct rows (the mid-batch refill is tests/test_trainer_gpu.py's).  Weights within 2e-6 per step; the 1 % validation loss (NNT:102-113, 21 rows) every 50
    iterations within 1e-3 relative; accuracy (MT:159-175) over all rows at the end: labels equal wherever the oracle's top-2
    logit margin exceeds 1e-3, and the accuracies equal when no row is below it."""
    dims, B, N, iters, every = [784, 100, 50, 10], 32, 2048, 300, 50
    lab = labels[:N]
    pix = prototype_images(lab)
    X = pix.astype(np.float64) / 255.0
    Y = np.eye(10)[lab]
    net = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    trainer = gnn.NeuralNetTrainer(pix, lab, net, raw_u8=True)
    ref = oracle_mod.OracleNet(dims)
    ref.set_alloc_per_sample(0)
    smp = oracle_mod.Sampler(N)
    vs = N // 100 + 1                                                  # NNT:65
    for done in range(every, iters + 1, every):
        trainer.train(every, 0.01, B, 0.9, False)
        for _ in range(every):
            idx = smp.sample(B)
            assert idx.size == B
            ref.gradient_step(X[idx], Y[idx], 0.01, 0.9)
        assert net.time == done == ref.time
        dw = np.abs(net.get_weights() - ref.get_weights()).max()
        dv = np.abs(net.get_momentum() - ref.get_momentum()).max()
        assert dw <= 2e-6 * done and dv <= 2e-6 * done, (done, dw, dv)
        v, vr = trainer.validate(vs), float(ref.calculate_loss(X[:vs], Y[:vs]).mean())
        assert abs(v - vr) <= 1e-3 * abs(vr) + 1e-6, (done, v, vr)
    # the run has learnt something (prototype classes are separable) and both sides agree on how much
    logits = ref.logits(X)
    s = np.sort(logits, axis=1)
    safe = (s[:, -1] - s[:, -2]) > 1e-3
    want = ref.argmax(X)
    got = np.concatenate([net.argmax_range(f, 256) for f in range(0, N, 256)]) if net.max_batch >= 256 else \
        np.concatenate([net.argmax_range(f, B) for f in range(0, N, B)])
    assert np.array_equal(got[safe], want[safe])
    acc, acc_ref = gnn.accuracy(net, lab), float((want == lab).mean())
    assert acc_ref > 0.9
    assert abs(acc - acc_ref) <= (~safe).sum() / N
    if safe.all():
        assert acc == acc_ref
