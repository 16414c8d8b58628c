"""NeuralNetTrainer mirror and MNIST data handling (N2) on the GPU."""
import io
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sampler_matches_oracle_restatement(gnn, oracle_mod):
    """product sampler (Fenwick tree) == oracle sampler (ArrayList-style memmove), incl. the
    mid-batch refill and the duplicate collapse (NNT:143-168)."""
    for master, batch, draws in ((60000, 128, 40), (10, 4, 12), (37, 16, 20), (1000, 999, 5)):
        a, b = gnn.Sampler(master), oracle_mod.Sampler(master)
        for _ in range(draws):
            x, y = a.sample(batch), b.sample(batch)
            assert np.array_equal(x, y)
    assert gnn.Sampler(60000).sample(5)[0] == 8985      # Random(1).nextInt(60000)


def test_train_equals_stepwise_and_oracle(gnn, oracle_mod):
    dims, N, B, iters = [784, 100, 50, 10], 300, 32, 25      # crosses an epoch boundary (300 / 32)
    rng = np.random.default_rng(2)
    pix = rng.integers(0, 256, (N, 784), dtype=np.uint8)
    pix[rng.random((N, 784)) < 0.8] = 0
    lab = rng.integers(0, 10, N, dtype=np.uint8)
    X, Y = pix / 255.0, np.eye(10)[lab]
    a = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    b = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    ta = gnn.NeuralNetTrainer(pix, lab, a, raw_u8=True)
    tb = gnn.NeuralNetTrainer(X, Y, b)
    ta.train(iters, 0.01, B, 0.9, False)                      # gnn_mlp_train_sampled
    obs = io.StringIO()
    tb.train(iters, 0.01, B, 0.9, False, observer=obs)        # the observed loop: steps + validation passes on the device
    assert a.time == iters == b.time
    assert np.array_equal(a.get_weights(), b.get_weights())
    lines = obs.getvalue().strip().split("\n")
    assert len(lines) == iters and lines[0].startswith("0,") and lines[-1].startswith("%d," % (iters - 1))
    # the same draws through the oracle
    ref = oracle_mod.OracleNet(dims)
    ref.set_alloc_per_sample(0)
    smp = oracle_mod.Sampler(N)
    for _ in range(iters):
        idx = smp.sample(B)
        ref.gradient_step(X[idx], Y[idx], 0.01, 0.9)
    assert np.abs(a.get_weights() - ref.get_weights()).max() <= 2e-6 * iters
    vs = N // 100 + 1
    v_ref = ref.calculate_loss(X[:vs], Y[:vs]).mean()
    assert abs(tb.validate(vs) - v_ref) <= 2e-4 * abs(v_ref) + 2e-4
    assert float(lines[-1].split(",")[1]) == pytest.approx(round(v_ref, 2), abs=0.011)
    acc = gnn.accuracy(a, lab)
    acc_ref = float((ref.argmax(X) == lab).mean())
    assert abs(acc - acc_ref) <= 2.0 / N


def test_idx_reader(gnn, tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (7, 28, 28), dtype=np.uint8)
    lab = rng.integers(0, 10, 7, dtype=np.uint8)
    (tmp_path / "img").write_bytes(struct.pack(">iiii", 2051, 7, 28, 28) + img.tobytes())
    (tmp_path / "lab").write_bytes(struct.pack(">ii", 2049, 7) + lab.tobytes())
    assert np.array_equal(gnn.read_idx_images(tmp_path / "img"), img.reshape(7, 784))
    assert np.array_equal(gnn.read_idx_labels(tmp_path / "lab"), lab)
    (tmp_path / "bad").write_bytes(struct.pack(">ii", 2051, 7) + lab.tobytes())
    with pytest.raises(ValueError):
        gnn.read_idx_labels(tmp_path / "bad")


def test_checkpoint_round_trip(gnn, tmp_path):
    dims, B = [784, 100, 50, 10], 32
    rng = np.random.default_rng(4)
    X = rng.random((B * 4, 784)); Y = np.eye(10)[rng.integers(0, 10, B * 4)]
    a = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    a.upload_dataset(X, Y)
    a.train_range(0, B, 3, 0.0125, 0.9)
    a.save_checkpoint(tmp_path / "ck.bin")
    a.train_range(0, B, 2, 0.0125, 0.9)
    b = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    b.load_checkpoint(tmp_path / "ck.bin")
    assert b.time == 3
    b.upload_dataset(X, Y)
    b.train_range(0, B, 2, 0.0125, 0.9)        # resume: momentum restored too
    assert np.array_equal(a.get_weights(), b.get_weights()) and a.time == b.time == 5
    c = gnn.SoftmaxCrossEntropyNeuralNet([784, 100, 40, 10], max_batch=B)
    with pytest.raises(gnn.GnnError):
        c.load_checkpoint(tmp_path / "ck.bin")
    # same dims, another net: the header carries output kind, activations, loss and dtype
    for other in (gnn.SoftmaxCrossEntropyNeuralNet(dims, inner_act=gnn.ACT_SIGMOID, max_batch=B),
                  gnn.SoftmaxCrossEntropyNeuralNet(dims, dtype=gnn.DTYPE_BF16, max_batch=B),
                  gnn.GeneralNeuralNet(dims, max_batch=B)):
        with pytest.raises(gnn.GnnError, match="configuration differs"):
            other.load_checkpoint(tmp_path / "ck.bin")
    # truncation, trailing bytes and a flipped payload byte are caught by length + checksum
    raw = (tmp_path / "ck.bin").read_bytes()
    flipped = bytearray(raw); flipped[len(raw) // 2] ^= 1
    for name, data in (("short", raw[:-9]), ("long", raw + b"x"), ("flip", bytes(flipped))):
        (tmp_path / name).write_bytes(data)
        with pytest.raises(gnn.GnnError, match="checksum"):
            gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B).load_checkpoint(tmp_path / name)


def test_train_sampled_ragged_batches_and_chunks(gnn, oracle_mod):
    """gnn_mlp_train_sampled on the fused path reads the sampled rows through the index vector
    inside its kernels (no gather pass) and samples ahead on a worker thread in chunks of 256
    iterations: a batch size that is not a multiple of the 4-row blocks / 16-row padding, an epoch
    that ends mid-batch, and a run longer than one chunk must equal index-by-index stepping
    (bitwise) and the oracle driven by the oracle's own sampler."""
    dims, N, B = [64, 40, 24, 10], 101, 27
    rng = np.random.default_rng(5)
    X = rng.random((N, dims[0])) * (rng.random((N, dims[0])) < 0.5)
    lab = rng.integers(0, 10, N)
    Y = np.eye(10)[lab]
    a = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    b = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    ta = gnn.NeuralNetTrainer(X, Y, a)
    iters = 300                                               # two sampler chunks (256 + 44), 80 epochs
    ta.train(iters, 0.01, B, 0.9, False)
    b.upload_dataset(X, Y)
    smp = gnn.Sampler(N)
    ref = oracle_mod.OracleNet(dims)
    ref.set_alloc_per_sample(0)
    osmp = oracle_mod.Sampler(N)
    for i in range(iters):
        idx = smp.sample(B)
        b.gradient_step_indexed(idx, 0.01, 0.9, False)
        if i < 40:
            oidx = osmp.sample(B)
            assert np.array_equal(np.asarray(idx), np.asarray(oidx))
            ref.gradient_step(X[oidx], Y[oidx], 0.01, 0.9)
            if i == 39:
                assert np.abs(b.get_weights() - ref.get_weights()).max() <= 2e-6 * 40
    assert a.time == iters == b.time
    assert np.array_equal(a.get_weights(), b.get_weights())
    assert np.array_equal(a.get_momentum(), b.get_momentum())


def test_count_hits_range_is_the_reference_test_loop(gnn, oracle_mod):
    """gnn_mlp_count_hits_range = testOnTrainingData / testOnTestData (MT:159-197) on the device: rows walked in blocks of
    max_batch (a ragged last block), `>=` argmax against the expected class (the LAST index holding 1, MT:186-188), one count.
    Against the per-block argmax_range loop (same kernels: equal), and against the oracle's propagate + argmax rule on every
    row whose top-2 logit margin is safe."""
    dims, N, B = [784, 100, 50, 10], 1000, 128          # 7 full blocks + 104 rows
    rng = np.random.default_rng(8)
    lab = rng.integers(0, 10, N)
    proto = rng.random((10, 784)) * (rng.random((10, 784)) < 0.2)
    X = np.clip(proto[lab] + 0.15 * rng.standard_normal((N, 784)) * (proto[lab] > 0), 0, 1)
    Y = np.eye(10)[lab]
    net = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    trainer = gnn.NeuralNetTrainer(X, Y, net)
    trainer.train(60, 0.01, 32, 0.9, False)             # somewhere between chance and perfect
    hits = net.count_hits_range(0, N)
    by_blocks = sum(int((net.argmax_range(f, min(B, N - f)) == lab[f:f + B]).sum()) for f in range(0, N, B))
    assert hits == by_blocks
    assert gnn.accuracy(net) == hits / N == gnn.accuracy(net, lab)
    assert net.count_hits_range(100, 333) == sum(int((net.argmax_range(f, min(B, 433 - f)) == lab[f:min(f + B, 433)]).sum()) for f in range(100, 433, B))
    ref = oracle_mod.OracleNet(dims)
    ref.set_weights(net.get_weights())
    logits = ref.logits(X)
    s = np.sort(logits, axis=1)
    safe = (s[:, -1] - s[:, -2]) > 1e-3
    want = ref.argmax(X)
    assert abs(hits - int((want == lab).sum())) <= int((~safe).sum())
    assert 0.2 < hits / N <= 1.0
    # an expected row with two ones: the LAST one counts (MT:186-188); with none: class 0
    Y2 = Y.copy(); Y2[0] = 0; Y2[0, 2] = 1; Y2[0, 7] = 1; Y2[1] = 0
    net.upload_dataset(X, Y2)
    lab2 = lab.copy(); lab2[0] = 7; lab2[1] = 0
    assert net.count_hits_range(0, 2) == int((net.argmax_range(0, 2) == lab2[:2]).sum())
    with pytest.raises(gnn.GnnError):
        net.count_hits_range(N - 5, 6)


class _Monitor:
    def __init__(self): self.steps, self.finished = 0, 0
    def step(self): self.steps += 1
    def finish(self): self.finished += 1


def test_observed_training_loops_run_on_the_device(gnn, oracle_mod):
    """NNT:68-72 / 75-79: gradientStep + validate(validationSize) per iteration.  gnn_mlp_train_sampled_observed keeps both on the
    device (the validation losses of a call come back in one readback) -- against the same loop made of one ABI call per
    action (host sampler, indexed step, a loss readback per iteration): the SAME weights bit for bit, the same validation
    losses up to the order of an fp64 sum of the same f32 values, the reference's line format; and the oracle's validate()."""
    dims, N, B, iters = [784, 100, 50, 10], 640, 32, 45     # validation size 7; two epochs and a bit
    rng = np.random.default_rng(12)
    pix = rng.integers(0, 256, (N, 784), dtype=np.uint8)
    pix[rng.random((N, 784)) < 0.8] = 0
    lab = rng.integers(0, 10, N, dtype=np.uint8)
    X, Y = pix / 255.0, np.eye(10)[lab]
    a = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    b = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    c = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=4)   # validation size 7 > max_batch: two blocks per validation pass
    ta, tb, tc = gnn.NeuralNetTrainer(pix, lab, a, raw_u8=True), gnn.NeuralNetTrainer(pix, lab, b, raw_u8=True), gnn.NeuralNetTrainer(pix, lab, c, raw_u8=True)
    oa, ob, oc = io.StringIO(), io.StringIO(), io.StringIO()
    mon = _Monitor()
    ta.OBSERVER_BURST = 16                                   # three device loops: 16 + 16 + 13 iterations
    ta.train(iters, 0.01, B, 0.9, False, monitor=mon, observer=oa)
    tb.train_stepwise(iters, 0.01, B, 0.9, False, observer=ob)
    assert mon.steps == iters and mon.finished == 1
    assert a.time == iters == b.time
    assert np.array_equal(a.get_weights(), b.get_weights()) and np.array_equal(a.get_momentum(), b.get_momentum())
    la, lb = oa.getvalue().strip().split("\n"), ob.getvalue().strip().split("\n")
    assert len(la) == iters == len(lb)
    for i, (x, y) in enumerate(zip(la, lb)):
        assert x.split(",")[0] == str(i) == y.split(",")[0]
        assert len(x.split(",")[1].split(".")[1]) == 2                          # "%d,%.2f" (NNT:71)
        assert abs(float(x.split(",")[1]) - float(y.split(",")[1])) <= 0.0100001   # (a sum that rounds the other way at a .005)
    # the unrounded values, straight from the ABI, against the oracle's validate() on the oracle's trajectory
    import ctypes as C
    vs = N // 100 + 1
    val = np.empty(10)
    d = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    td = gnn.NeuralNetTrainer(pix, lab, d, raw_u8=True)
    gnn.load_library().gnn_mlp_train_sampled_observed(d._h, td.sampler._h, 10, B, 0.01, 0.9, 0, vs, val.ctypes.data_as(C.POINTER(C.c_double)))
    ref = oracle_mod.OracleNet(dims)
    ref.set_alloc_per_sample(0)
    smp = oracle_mod.Sampler(N)
    for i in range(10):
        idx = smp.sample(B)
        ref.gradient_step(X[idx], Y[idx], 0.01, 0.9)
        vr = float(ref.calculate_loss(X[:vs], Y[:vs]).mean())                    # NNT:102-113
        assert abs(val[i] - vr) <= 2e-4 * abs(vr) + 2e-4, (i, val[i], vr)
    # observer only (NNT:75-79), validation rows in two blocks: the same steps again
    tc.train(iters, 0.01, 4, 0.9, False, observer=oc)
    assert c.time == iters and len(oc.getvalue().strip().split("\n")) == iters
    # monitor only (NNT:82-86): no validation, the device loop in bursts
    e = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    te = gnn.NeuralNetTrainer(pix, lab, e, raw_u8=True)
    te.OBSERVER_BURST = 20
    m2 = _Monitor()
    te.train(iters, 0.01, B, 0.9, False, monitor=m2)
    assert m2.steps == iters and m2.finished == 1 and np.array_equal(e.get_weights(), a.get_weights())
    with pytest.raises(gnn.GnnError):
        gnn._capi.check(gnn.load_library().gnn_mlp_train_sampled_observed(d._h, td.sampler._h, 3, B, 0.01, 0.9, 0, N + 1, val.ctypes.data_as(C.POINTER(C.c_double))))


@pytest.mark.parametrize("kind", ["softmax_f32", "softmax_bf16", "general_f32"])
def test_count_hits_large_blocks_and_both_classes(gnn, oracle_mod, kind):
    """Evaluation over a data set in LARGE blocks takes the per-layer GEMM chain (from 2 048 rows per block on, plan.hip: do_forward)
    instead of the small-net kernels: the same count as with 128-row blocks up to rows whose top-2 margin is within rounding, for both
    NeuralNet classes and in bf16; against the oracle's rule on every safe row (f32)."""
    dims, N = [784, 300, 100, 10], 5000
    rng = np.random.default_rng(21)
    lab = rng.integers(0, 10, N)
    proto = rng.random((10, 784)) * (rng.random((10, 784)) < 0.2)
    X = np.clip(proto[lab] + 0.2 * rng.standard_normal((N, 784)) * (proto[lab] > 0), 0, 1)
    Y = np.eye(10)[lab]
    bf16 = kind.endswith("bf16")
    make = (lambda mb: gnn.GeneralNeuralNet(dims, inner_act="sigmoid", last_act="sigmoid", max_batch=mb)) if kind.startswith("general") else \
           (lambda mb: gnn.SoftmaxCrossEntropyNeuralNet(dims, dtype=gnn.DTYPE_BF16 if bf16 else gnn.DTYPE_F32, max_batch=mb))
    small, big = make(128), make(4096)
    tr = gnn.NeuralNetTrainer(X, Y, small)
    tr.train(80, 0.1 if kind.startswith("general") else 0.01, 64, 0.9, False)
    w = small.get_weights()
    big.set_weights(w)
    big.upload_dataset(X, Y)
    hs, hb = small.count_hits_range(0, N), big.count_hits_range(0, N)     # 40 blocks of 128 (8 left over) / 4096 + 904 rows
    assert 0.15 * N < hs <= N
    if not bf16:
        ref = (oracle_mod.OracleNet(dims, out_kind=oracle_mod.OUT_ACT_LOSS, inner_act=1, last_act=1) if kind.startswith("general")
               else oracle_mod.OracleNet(dims))
        ref.set_weights(w)
        out = ref.propagate(X[:1500]) if kind.startswith("general") else ref.logits(X[:1500])
        s = np.sort(out, axis=1)
        unsafe = int(((s[:, -1] - s[:, -2]) <= (1e-4 if kind.startswith("general") else 1e-3)).sum())
        want = int((ref.argmax(X[:1500]) == lab[:1500]).sum())
        assert abs(small.count_hits_range(0, 1500) - want) <= unsafe
        assert abs(big.count_hits_range(0, 1500) - want) <= unsafe
        assert abs(hs - hb) <= 3          # (another summation order in the large blocks: only near-ties may move)
    else:
        assert abs(hs - hb) <= 0.01 * N   # (bf16: operands rounded the same way, accumulation order differs)
    assert big.count_hits_range(4096, 904) + big.count_hits_range(0, 4096) == hb


def test_evaluation_workspace_blocks_above_max_batch(gnn, monkeypatch):
    """A handle sized for training batches (max_batch 32) evaluates and validates in blocks of up to 16 384 rows through its
    evaluation workspace (csrc/plan.hip: EvalScope): against GNN_MLP_EVAL_ROWS=0 (blocks of max_batch, the form until round 4).
    Below 2 048 rows per block the kernels are the same per row (exact equality); above, the GEMM chain (near-ties may move)."""
    dims, N, B = [784, 100, 50, 10], 3000, 32
    rng = np.random.default_rng(33)
    lab = rng.integers(0, 10, N)
    proto = rng.random((10, 784)) * (rng.random((10, 784)) < 0.2)
    X = np.clip(proto[lab] + 0.25 * rng.standard_normal((N, 784)) * (proto[lab] > 0), 0, 1)
    Y = np.eye(10)[lab]
    ws = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    monkeypatch.setenv("GNN_MLP_EVAL_ROWS", "0")
    plain = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B)
    monkeypatch.delenv("GNN_MLP_EVAL_ROWS")
    ta, tb = gnn.NeuralNetTrainer(X, Y, ws), gnn.NeuralNetTrainer(X, Y, plain)
    import io
    oa, ob = io.StringIO(), io.StringIO()
    ta.train(40, 0.01, B, 0.9, False, observer=oa)       # validation: 31 rows = one block of 31 in the plain form too
    tb.train(40, 0.01, B, 0.9, False, observer=ob)
    assert np.array_equal(ws.get_weights(), plain.get_weights())
    assert oa.getvalue() == ob.getvalue()
    assert ws.count_hits_range(0, 1900) == plain.count_hits_range(0, 1900)         # one block of 1 900 / 60 blocks of 32: the same kernels per row
    assert abs(ws.count_hits_range(0, N) - plain.count_hits_range(0, N)) <= 3     # one block of 3 000: the GEMM chain
    assert ws.count_hits_range(100, 7) == plain.count_hits_range(100, 7)
    # steps after an evaluation: the workspace was swapped out again
    ta.train(5, 0.01, B, 0.9, False); tb.train(5, 0.01, B, 0.9, False)
    assert np.array_equal(ws.get_weights(), plain.get_weights())
