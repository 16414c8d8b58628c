"""CPU tests that pin the oracle (test infrastructure) -- no GPU needed.

PARITY UNPINNED BY THE REFERENCE: it holds no tests, golden vectors or fixtures for this path
 and cannot be run here (Java, no JDK).  So the oracle is pinned by:
  1. java.util.Random known answers derived from the documented LCG,
  2. an independent numpy matrix-form oracle (tests/np_oracle.py) that must agree to 1e-12,
  3. a finite-difference check of calculateWeightGradient against calculateLoss,
  4. hand-worked 2-2-2 vectors,
  5. the committed golden fixtures (tests/golden/golden.json, made by tests/golden/make_golden.py).
"""
import json
import math
import os

import numpy as np
import pytest

from tests import np_oracle

LEAKY, SIGMOID, TANH, RELU, IDENT = range(5)
HERE = os.path.dirname(os.path.abspath(__file__))


# ---- 1. java.util.Random ---------------------------------------------------------------------
def test_java_random_known_answers(oracle_mod):
    R = oracle_mod.JavaRandom
    r = R(1)
    assert [r.next_double() for _ in range(4)] == [0.7308781907032909, 0.41008081149220166,
                                                   0.20771484130971707, 0.3327170559595112]
    assert R(1).next_int() == -1155869325
    r = R(1)
    assert [r.next_int(60000) for _ in range(5)] == [8985, 4588, 21847, 50313, 24254]
    r = R(1)
    assert [r.next_int(10) for _ in range(5)] == [5, 8, 7, 3, 4]
    assert R(0).next_int() == -1155484576                 # widely published JDK values
    assert R(42).next_int() == -1170105035
    assert R(42).next_gaussian() == pytest.approx(1.1419053154730547, abs=1e-15)
    r = R(7)
    draws = [r.next_int(16) for _ in range(200)]          # power-of-two branch
    assert min(draws) >= 0 and max(draws) <= 15 and len(set(draws)) == 16


def test_init_spot_values(oracle_mod):
    """appendLayer (SCE:139-156): Random(1), layer by layer, row-major, nextDouble()-0.5."""
    w = oracle_mod.OracleNet([784, 300, 100, 10]).get_weights()
    assert w.size == 266200
    assert list(w[:3]) == [0.23087819070329085, -0.08991918850779834, -0.29228515869028293]
    assert w[300] == -0.46020562391264386                  # W_0[1][0]
    assert w[784 * 300] == -0.10748465918004868            # W_1[0][0]
    assert w[-1] == -0.3508213257407772                    # W_2[99][9]
    w = oracle_mod.OracleNet([784, 100, 50, 10]).get_weights()
    assert w.size == 83900
    assert w[100] == 0.17993542019102504
    assert w[78400] == 0.3467531210407554
    assert w[-1] == 0.48691319924705945
    assert w.min() >= -0.5 and w.max() < 0.5


# ---- 2. C oracle vs numpy matrix-form oracle ---------------------------------------------------
CASES = [
    ([784, 100, 50, 10], 8, LEAKY, 0, IDENT),
    ([20, 17, 33, 7], 19, TANH, 0, IDENT),
    ([12, 9, 4], 5, SIGMOID, 0, IDENT),
    ([64, 10], 7, RELU, 0, IDENT),
    ([30, 21, 18, 5], 13, SIGMOID, 1, SIGMOID),
    ([30, 21, 18, 5], 13, LEAKY, 1, TANH),
    ([9, 8, 7, 6, 5], 3, TANH, 1, IDENT),
]


@pytest.mark.parametrize("dims,B,inner,out_kind,last", CASES)
def test_c_oracle_matches_numpy_oracle(oracle_mod, dims, B, inner, out_kind, last):
    rng = np.random.default_rng(11)
    X = rng.random((B, dims[0])) - (0.3 if inner != LEAKY else 0.0)
    Y = np.eye(dims[-1])[rng.integers(0, dims[-1], B)] if out_kind == 0 else rng.random((B, dims[-1]))
    net = oracle_mod.OracleNet(dims, out_kind=out_kind, inner_act=inner, last_act=last)
    w0 = net.get_weights() * (0.3 if dims[0] > 100 else 1.0)
    net.set_weights(w0)
    Ws = np_oracle.split(w0, dims)
    _, out = np_oracle.forward(Ws, X, inner, out_kind, last)
    assert np.allclose(net.propagate(X), out, rtol=1e-12, atol=1e-15)
    assert np.allclose(net.calculate_loss(X, Y), np_oracle.loss(Ws, X, Y, inner, out_kind, last),
                       rtol=1e-12, atol=1e-14)
    g = sum(net.calculate_weight_gradient(X[b], Y[b]) for b in range(B))
    gn = np_oracle.gradient(Ws, X, Y, inner, out_kind, last)
    assert np.allclose(g, gn, rtol=1e-11, atol=1e-13 * np.abs(gn).max())
    w, v = w0.copy(), np.zeros_like(w0)
    for s in range(3):
        net.gradient_step(X, Y, 0.05, 0.9)
        w, v = np_oracle.gradient_step(w, v, dims, X, Y, 0.05, 0.9, inner, out_kind, last)
    assert net.time == 3
    assert np.allclose(net.get_weights(), w, rtol=1e-11, atol=1e-13)
    assert np.allclose(net.get_momentum(), v, rtol=1e-10, atol=1e-14)


def test_alloc_mode_does_not_change_results(oracle_mod):
    dims = [12, 9, 4]
    X, Y = oracle_mod.synthetic_batch(dims, 6, 5)
    a, b = oracle_mod.OracleNet(dims), oracle_mod.OracleNet(dims)
    b.set_alloc_per_sample(0)
    for _ in range(3):
        a.gradient_step(X, Y, 0.1, 0.9)
        b.gradient_step(X, Y, 0.1, 0.9)
    assert np.array_equal(a.get_weights(), b.get_weights())


# ---- 3. finite differences ----------------------------------------------------------------------
@pytest.mark.parametrize("out_kind,inner,last", [(0, SIGMOID, IDENT), (0, TANH, IDENT), (1, SIGMOID, SIGMOID),
                                                 (1, TANH, IDENT)])
def test_gradient_finite_difference(oracle_mod, out_kind, inner, last):
    dims = [5, 4, 3, 3]
    rng = np.random.default_rng(3)
    x = rng.random(5)
    y = np.eye(3)[1] if out_kind == 0 else rng.random(3)
    net = oracle_mod.OracleNet(dims, out_kind=out_kind, inner_act=inner, last_act=last)
    w = net.get_weights()
    g = net.calculate_weight_gradient(x, y)
    eps = 1e-6
    num = np.empty_like(w)
    for i in range(w.size):
        wp, wm = w.copy(), w.copy()
        wp[i] += eps
        wm[i] -= eps
        net.set_weights(wp)
        lp = net.calculate_loss(x, y)
        net.set_weights(wm)
        lm = net.calculate_loss(x, y)
        num[i] = (lp - lm) / (2 * eps)
    assert np.abs(num - g).max() < 1e-8


# ---- 4. hand-worked 2-2-2 vectors ----------------------------------------------------------------
def test_hand_worked_2_2_2_softmax(oracle_mod):
    """2-2-2 SCE net with identity-like arithmetic written out by hand (leaky ReLU on positive
    values is the identity).  W0 = [[0.1, 0.2], [0.3, 0.4]], W1 = [[0.5, -0.5], [0.25, 0.75]],
    x = (1, 2), y = (0, 1)."""
    net = oracle_mod.OracleNet([2, 2, 2])
    net.set_weights(np.array([0.1, 0.2, 0.3, 0.4, 0.5, -0.5, 0.25, 0.75]))
    x, y = np.array([1.0, 2.0]), np.array([0.0, 1.0])
    z1 = [1 * 0.1 + 2 * 0.3, 1 * 0.2 + 2 * 0.4]            # 0.7, 1.0
    z2 = [z1[0] * 0.5 + z1[1] * 0.25, z1[0] * -0.5 + z1[1] * 0.75]   # 0.6, 0.4
    e = [math.exp(z2[0]), math.exp(z2[1])]
    p = [e[0] / (e[0] + e[1]), e[1] / (e[0] + e[1])]
    assert np.allclose(net.propagate(x), p, rtol=1e-15)
    assert net.calculate_loss(x, y) == pytest.approx(-math.log(p[1]), rel=1e-14)
    d2 = [p[0] - 0.0, p[1] - 1.0]                            # SCE:250
    g1 = [[d2[0] * z1[0], d2[1] * z1[0]], [d2[0] * z1[1], d2[1] * z1[1]]]
    d1 = [(0.5 * d2[0] + -0.5 * d2[1]) * 1.0, (0.25 * d2[0] + 0.75 * d2[1]) * 1.0]   # z1 > 0 -> f' = 1
    g0 = [[d1[0] * 1.0, d1[1] * 1.0], [d1[0] * 2.0, d1[1] * 2.0]]
    g = net.calculate_weight_gradient(x, y)
    assert np.allclose(g, np.array(g0 + g1).ravel(), rtol=1e-14)
    net.gradient_step(x[None], y[None], 0.5, 0.9)
    w = net.get_weights()
    assert w[0] == pytest.approx(0.1 - 0.5 * g0[0][0], rel=1e-14)
    assert w[7] == pytest.approx(0.75 - 0.5 * g1[1][1], rel=1e-14)
    v = net.get_momentum()
    assert v[7] == pytest.approx(0.5 * g1[1][1], rel=1e-14)
    # second step on the same sample: momentum term 0.9*prev enters (SCE:333)
    g_again = net.calculate_weight_gradient(x, y)
    net.gradient_step(x[None], y[None], 0.5, 0.9)
    assert net.get_momentum()[7] == pytest.approx(0.5 * g_again[7] + 0.9 * v[7], rel=1e-14)


def test_hand_worked_2_2_2_sigmoid_squared(oracle_mod):
    """GeneralNeuralNet, sigmoid everywhere, 0.5*(a-y)^2 -- the network of doc/backprop.pdf.
    Note the quirk SCE:183-186 / GNN:202-205: the activation is applied to the RAW INPUT too."""
    sig = lambda t: 1.0 / (1.0 + math.exp(-t))
    net = oracle_mod.OracleNet([2, 2, 2], out_kind=1, inner_act=SIGMOID, last_act=SIGMOID)
    W0 = [[0.15, 0.25], [0.20, 0.30]]
    W1 = [[0.40, 0.50], [0.45, 0.55]]
    net.set_weights(np.array(W0 + W1).ravel())
    x, y = [0.05, 0.10], [0.01, 0.99]
    a0 = [sig(x[0]), sig(x[1])]                              # quirk: f(input)
    z1 = [a0[0] * W0[0][0] + a0[1] * W0[1][0], a0[0] * W0[0][1] + a0[1] * W0[1][1]]
    a1 = [sig(z1[0]), sig(z1[1])]
    z2 = [a1[0] * W1[0][0] + a1[1] * W1[1][0], a1[0] * W1[0][1] + a1[1] * W1[1][1]]
    a2 = [sig(z2[0]), sig(z2[1])]
    assert np.allclose(net.propagate(np.array(x)), a2, rtol=1e-15)
    loss = 0.5 * (a2[0] - y[0]) ** 2 + 0.5 * (a2[1] - y[1]) ** 2
    assert net.calculate_loss(np.array(x), np.array(y)) == pytest.approx(loss, rel=1e-14)
    d2 = [(a2[j] - y[j]) * a2[j] * (1 - a2[j]) for j in range(2)]          # GNN:267-271
    d1 = [(W1[j][0] * d2[0] + W1[j][1] * d2[1]) * a1[j] * (1 - a1[j]) for j in range(2)]
    g = net.calculate_weight_gradient(np.array(x), np.array(y))
    expect = [a0[0] * d1[0], a0[0] * d1[1], a0[1] * d1[0], a0[1] * d1[1],
              a1[0] * d2[0], a1[0] * d2[1], a1[1] * d2[0], a1[1] * d2[1]]
    assert np.allclose(g, expect, rtol=1e-13)


# ---- argmax rule, encoding, sampler ----------------------------------------------------------------
def test_argmax_rule(oracle_mod):
    """MT:166-168: `>=` -> ties resolve to the highest index; NaN at 0 is sticky."""
    am = oracle_mod.argmax_rule
    assert am([0.1, 0.7, 0.2]) == 1
    assert am([0.5, 0.5, 0.0]) == 1
    assert am([0.2, 0.2, 0.2, 0.2]) == 3
    assert am([0.1, float("nan"), 0.05]) == 0
    assert am([float("nan"), 0.9, 0.1]) == 0
    assert am([1.0]) == 0


def test_encoding(oracle_mod):
    import ctypes as C
    raw = np.array([0, 1, 127, 128, 255], dtype=np.uint8)
    out = np.empty(5)
    oracle_mod.lib().oracle_encode_image(raw.ctypes.data_as(C.POINTER(C.c_uint8)), 5,
                                         out.ctypes.data_as(C.POINTER(C.c_double)))
    assert list(out) == [0.0, 1 / 255.0, 127 / 255.0, 128 / 255.0, 1.0]     # MT:98
    lab = np.empty(10)
    oracle_mod.lib().oracle_encode_label(3, 10, lab.ctypes.data_as(C.POINTER(C.c_double)))
    assert lab.sum() == 1.0 and lab[3] == 1.0                                # MT:112-118


def test_sampler_without_replacement(oracle_mod):
    """NNT:143-168: nextInt(size) draws with removal; refill mid-batch; duplicates collapse."""
    s = oracle_mod.Sampler(60000)
    first = s.sample(128)
    assert first[0] == 8985 and first.size == 128          # Random(1).nextInt(60000) = 8985
    s = oracle_mod.Sampler(10)
    a = s.sample(4)
    b = s.sample(4)
    assert len(set(a) | set(b)) == 8                       # no replacement inside an epoch
    c = s.sample(4)                                        # 2 left, refill after 2 draws (NNT:149-151)
    assert set(c[:2]) == set(range(10)) - set(a) - set(b)
    assert 2 <= c.size <= 4                                # a post-refill duplicate collapses (H11)
    r = oracle_mod.JavaRandom(1)
    lst = list(range(10))
    exp = [lst.pop(r.next_int(len(lst))) for _ in range(4)]
    assert list(a) == exp


# ---- 5. golden fixtures ---------------------------------------------------------------------------
def test_golden_fixtures_match_oracle(oracle_mod):
    """Regression pin: tests/golden/golden.json was written by tests/golden/make_golden.py from
    this oracle; inputs are regenerated from java.util.Random seeds, so the file is small."""
    from tests.golden import make_golden
    with open(os.path.join(HERE, "golden", "golden.json")) as f:
        gold = json.load(f)
    fresh = make_golden.generate()
    assert gold["format"] == fresh["format"]
    assert len(gold["cases"]) == len(fresh["cases"])
    for a, b in zip(gold["cases"], fresh["cases"]):
        assert a["name"] == b["name"]
        for k in a:
            if isinstance(a[k], list) and a[k] and isinstance(a[k][0], float):
                assert np.allclose(a[k], b[k], rtol=1e-12, atol=1e-300), (a["name"], k)
            else:
                assert a[k] == b[k], (a["name"], k)


def test_train_log_row_reproduces_the_references_own_log(tmp_path):
    """tests/golden/reference_train_log_rows.json holds the five rows of the reference's logs/trainLog.csv (DATA written
    by MNISTTrainer.logTest, MT:211-219): the row writer must reproduce each row from its parsed fields."""
    import gnn_amd
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_train_log_rows.json")))["rows"]
    assert len(rows) == 5
    out = tmp_path / "logs" / "trainLog.csv"
    for r in rows:
        dims, it, step, batch, mom, noise, tr, te = r.split(",")
        args = ([int(d) for d in dims.split("-")], int(it), float(step), int(batch), float(mom), noise == "true", float(tr), float(te))
        assert gnn_amd.train_log_row(*args) == r + "\n"
        gnn_amd.log_test(out, *args)
    assert out.read_text().splitlines() == rows
