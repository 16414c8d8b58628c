"""Second, independent CPU oracle (test infrastructure): numpy fp64 MATRIX form of the
reference's math.  It shares no code with oracle/mlp_oracle.c (which follows the Java loop
nests sample by sample); the two must agree to ~1e-12 relative (tests/test_oracle.py).

Citations: SCE = /root/reference/src/SoftmaxCrossEntropyNeuralNet.java, GNN = GeneralNeuralNet.java.
"""
import numpy as np


def act(kind, z):
    if kind == 0:
        return np.where(z > 0, z, 0.01 * z)          # MT:234
    if kind == 1:
        return 1.0 / (1.0 + np.exp(-z))
    if kind == 2:
        return np.tanh(z)
    if kind == 3:
        return np.where(z > 0, z, 0.0)
    return z


def act_prime(kind, z):
    if kind == 0:
        return np.where(z <= 0, 0.01, 1.0)            # MT:235
    if kind == 1:
        s = 1.0 / (1.0 + np.exp(-z))
        return s * (1 - s)
    if kind == 2:
        return 1 - np.tanh(z) ** 2
    if kind == 3:
        return np.where(z <= 0, 0.0, 1.0)
    return np.ones_like(z)


def split(flat, dims):
    out, off = [], 0
    for l in range(len(dims) - 1):
        n = dims[l] * dims[l + 1]
        out.append(flat[off:off + n].reshape(dims[l], dims[l + 1]))
        off += n
    return out


def forward(Ws, X, inner, out_kind=0, last=1):
    """Returns pre-activations Z[0..L-1] (Z[0] = X) and the output (SCE:164-198 / GNN:183-221)."""
    Z = [np.asarray(X, dtype=np.float64)]
    for W in Ws:
        Z.append(act(inner, Z[-1]) @ W)               # f applied to the raw input too (SCE:183-186)
    if out_kind == 0:
        e = np.exp(Z[-1])                             # un-normalised, as SCE:368
        out = e / e.sum(axis=1, keepdims=True)
    else:
        out = act(last, Z[-1])
    return Z, out


def loss(Ws, X, Y, inner, out_kind=0, last=1):
    Z, out = forward(Ws, X, inner, out_kind, last)
    if out_kind == 0:
        return -(Y * np.log(out)).sum(axis=1)         # SCE:213-217
    return (0.5 * (out - Y) ** 2).sum(axis=1)        # GNN:236-239


def gradient(Ws, X, Y, inner, out_kind=0, last=1):
    """Sum over the batch of the per-sample weight gradients, flat (SCE:229-287 + SCE:305-322)."""
    Z, out = forward(Ws, X, inner, out_kind, last)
    L = len(Ws) + 1
    if out_kind == 0:
        D = out - Y                                   # SCE:249-251
    else:
        D = (out - Y) * act_prime(last, Z[-1])        # GNN:267-271
    G = [None] * (L - 1)
    for l in range(L - 2, -1, -1):
        G[l] = act(inner, Z[l]).T @ D                 # SCE:253-258, 279-283
        if l >= 1:
            D = (D @ Ws[l].T) * act_prime(inner, Z[l])  # SCE:272-278
    return np.concatenate([g.ravel() for g in G])


def gradient_step(w, v, dims, X, Y, step, momentum, inner, out_kind=0, last=1):
    """One gradientStep (SCE:297-346) on flat weights w / momentum v; returns new (w, v)."""
    g = gradient(split(w, dims), X, Y, inner, out_kind, last)
    adj = step * g / X.shape[0] + momentum * v        # SCE:333
    return w - adj, adj


# ---- bf16-operand variant (GNN_DTYPE_BF16): GEMM operands rounded to bf16, everything else fp64 ----
def bf16_round(x):
    """fp64/fp32 -> fp32 -> bf16 (round to nearest even, what v_cvt_pk_bf16_f32 does) -> fp64."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64).reshape(np.shape(x))


def _jittered(a, jitter):
    """`jitter` = (relative size, numpy Generator) or None: the operand as ANOTHER summation order of the f32 accumulation that
    produced it would have left it -- perturbed by ~1e-6 relative BEFORE it is rounded to bf16.  A value that sits within that
    distance of a bf16 rounding boundary then lands on the neighbouring bf16 value (0.4 % away): with bf16 operands, runs that
    differ only in summation order differ by such flips, and this is how the tests measure what ONE contract's results scatter by."""
    if jitter is None:
        return a
    rel, rng = jitter
    return a * (1.0 + rel * rng.standard_normal(np.shape(a)))


def _forward_bf16(Ws, X, inner, out_kind, last, jitter):
    q = bf16_round
    A = [act(inner, np.asarray(X, dtype=np.float64))]
    Aq = []                                            # the bf16 operand each activation matrix becomes, ONCE (forward and gradient use the same)
    Z = [np.asarray(X, dtype=np.float64)]
    for l, W in enumerate(Ws):
        Aq.append(q(_jittered(A[-1], jitter if l > 0 else None)))   # (the inputs are data, not sums: never jittered)
        z = Aq[-1] @ q(W)
        Z.append(z)
        if l < len(Ws) - 1:
            A.append(act(inner, z))
    if out_kind == 0:
        zz = Z[-1] - Z[-1].max(axis=1, keepdims=True)
        e = np.exp(zz)
        out = e / e.sum(axis=1, keepdims=True)
    else:
        out = act(last, Z[-1])
    return Z, A, Aq, out


def forward_bf16(Ws, X, inner, out_kind=0, last=1, jitter=None):
    """Activations / pre-activations as the bf16 GPU path forms them: every matrix product takes
    bf16-rounded operands and accumulates exactly; activations are kept unrounded between layers
    (they are fp32 in HBM and rounded when a tile is staged)."""
    Z, A, _, out = _forward_bf16(Ws, X, inner, out_kind, last, jitter)
    return Z, A, out


def gradient_bf16(Ws, X, Y, inner, out_kind=0, last=1, jitter=None):
    q = bf16_round
    Z, A, Aq, out = _forward_bf16(Ws, X, inner, out_kind, last, jitter)
    L = len(Ws) + 1
    D = out - Y if out_kind == 0 else (out - Y) * act_prime(last, Z[-1])
    G = [None] * (L - 1)
    for l in range(L - 2, -1, -1):
        Dq = q(_jittered(D, jitter))                   # (one rounding per delta matrix, used by both products, as on the GPU)
        G[l] = Aq[l].T @ Dq
        if l >= 1:
            D = (Dq @ q(Ws[l]).T) * act_prime(inner, Z[l])
    return np.concatenate([g.ravel() for g in G])


def gradient_step_bf16(w, v, dims, X, Y, step, momentum, inner, out_kind=0, last=1, jitter=None):
    g = gradient_bf16(split(w, dims), X, Y, inner, out_kind, last, jitter)
    adj = step * g / X.shape[0] + momentum * v
    return w - adj, adj
