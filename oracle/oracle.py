"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product path (graph-neural-net_amd/) never does.

PARITY UNPINNED BY THE REFERENCE: see the header of mlp_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

ACT_LEAKY_RELU, ACT_SIGMOID, ACT_TANH, ACT_RELU, ACT_IDENTITY = range(5)
OUT_SOFTMAX_CE, OUT_ACT_LOSS = 0, 1
LOSS_HALF_SQUARED = 0


def build(force=False):
    """Compile liboracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "mlp_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    L.oracle_create.restype = C.c_void_p
    L.oracle_create.argtypes = [ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64]
    L.oracle_destroy.argtypes = [C.c_void_p]
    L.oracle_propagate.argtypes = [C.c_void_p, dp, dp]
    L.oracle_last_logits.argtypes = [C.c_void_p, dp]
    L.oracle_loss.restype = C.c_double
    L.oracle_loss.argtypes = [C.c_void_p, dp, dp]
    L.oracle_weight_gradient.argtypes = [C.c_void_p, dp, dp, dp]
    L.oracle_gradient_step.restype = C.c_int
    L.oracle_gradient_step.argtypes = [C.c_void_p, dp, dp, C.c_int, C.c_double, C.c_double, C.c_int]
    L.oracle_argmax.restype = C.c_int
    L.oracle_argmax.argtypes = [dp, C.c_int]
    for name in ("get_weights", "set_weights", "get_momentum", "set_momentum"):
        getattr(L, "oracle_" + name).argtypes = [C.c_void_p, dp]
    L.oracle_num_params.restype = C.c_long
    L.oracle_num_params.argtypes = [C.c_void_p]
    L.oracle_time.restype = C.c_int
    L.oracle_time.argtypes = [C.c_void_p]
    L.oracle_set_alloc_per_sample.argtypes = [C.c_void_p, C.c_int]
    L.oracle_encode_image.argtypes = [C.POINTER(C.c_uint8), C.c_int, dp]
    L.oracle_encode_label.argtypes = [C.c_int, C.c_int, dp]
    L.oracle_sampler_create.restype = C.c_void_p
    L.oracle_sampler_create.argtypes = [C.c_int, C.c_int64]
    L.oracle_sampler_destroy.argtypes = [C.c_void_p]
    L.oracle_sampler_sample.restype = C.c_int
    L.oracle_sampler_sample.argtypes = [C.c_void_p, C.c_int, ip]
    L.oracle_fill_uniform.argtypes = [C.c_int64, C.c_long, C.c_double, dp]
    L.oracle_fill_onehot.argtypes = [C.c_int64, C.c_long, C.c_int, dp]
    # java.util.Random
    L.jrandom_seed.argtypes = [C.c_void_p, C.c_int64]
    L.jrandom_next_int.restype = C.c_int32
    L.jrandom_next_int.argtypes = [C.c_void_p]
    L.jrandom_next_int_bound.restype = C.c_int32
    L.jrandom_next_int_bound.argtypes = [C.c_void_p, C.c_int32]
    L.jrandom_next_double.restype = C.c_double
    L.jrandom_next_double.argtypes = [C.c_void_p]
    L.jrandom_next_gaussian.restype = C.c_double
    L.jrandom_next_gaussian.argtypes = [C.c_void_p]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class JavaRandom:
    """java.util.Random (mlp_oracle.c jrandom_*)."""

    def __init__(self, seed=1):
        self._buf = C.create_string_buffer(32)
        lib().jrandom_seed(self._buf, seed)

    def next_int(self, bound=None):
        if bound is None:
            return lib().jrandom_next_int(self._buf)
        return lib().jrandom_next_int_bound(self._buf, bound)

    def next_double(self):
        return lib().jrandom_next_double(self._buf)

    def next_gaussian(self):
        return lib().jrandom_next_gaussian(self._buf)


class OracleNet:
    """fp64 serial restatement of SoftmaxCrossEntropyNeuralNet / GeneralNeuralNet."""

    def __init__(self, dims, out_kind=OUT_SOFTMAX_CE, inner_act=ACT_LEAKY_RELU,
                 last_act=ACT_SIGMOID, loss=LOSS_HALF_SQUARED, seed=1):
        self.dims = [int(d) for d in dims]
        arr = (C.c_int32 * len(self.dims))(*self.dims)
        self._h = lib().oracle_create(arr, len(self.dims), out_kind, inner_act, last_act, loss, seed)
        if not self._h:
            raise ValueError("oracle_create failed for dims %r" % (dims,))
        self.n_params = lib().oracle_num_params(self._h)

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def time(self):
        return lib().oracle_time(self._h)

    def set_alloc_per_sample(self, flag):
        lib().oracle_set_alloc_per_sample(self._h, int(flag))

    def propagate(self, x):
        x = _f64(x)
        single = x.ndim == 1
        X = x.reshape(-1, self.dims[0])
        out = np.empty((X.shape[0], self.dims[-1]))
        for b in range(X.shape[0]):
            lib().oracle_propagate(self._h, _dp(X[b]), _dp(out[b]))
        return out[0] if single else out

    def logits(self, x):
        X = _f64(x).reshape(-1, self.dims[0])
        out = np.empty((X.shape[0], self.dims[-1]))
        tmp = np.empty(self.dims[-1])
        for b in range(X.shape[0]):
            lib().oracle_propagate(self._h, _dp(X[b]), _dp(tmp))
            lib().oracle_last_logits(self._h, _dp(out[b]))
        return out

    def calculate_loss(self, x, y):
        X = _f64(x).reshape(-1, self.dims[0])
        Y = _f64(y).reshape(-1, self.dims[-1])
        out = np.array([lib().oracle_loss(self._h, _dp(X[b]), _dp(Y[b])) for b in range(X.shape[0])])
        return out[0] if np.ndim(x) == 1 else out

    def calculate_weight_gradient(self, x, y):
        x, y = _f64(x), _f64(y)
        g = np.empty(self.n_params)
        lib().oracle_weight_gradient(self._h, _dp(x), _dp(y), _dp(g))
        return g

    def gradient_step(self, X, Y, step, momentum, noise=False):
        X = _f64(X).reshape(-1, self.dims[0])
        Y = _f64(Y).reshape(-1, self.dims[-1])
        rc = lib().oracle_gradient_step(self._h, _dp(X), _dp(Y), X.shape[0], step, momentum, int(noise))
        if rc:
            raise ValueError("oracle_gradient_step: empty batch")

    def argmax(self, x):
        out = np.atleast_2d(self.propagate(x))
        return np.array([lib().oracle_argmax(_dp(_f64(r)), out.shape[1]) for r in out], dtype=np.int32)

    def get_weights(self):
        w = np.empty(self.n_params)
        lib().oracle_get_weights(self._h, _dp(w))
        return w

    def set_weights(self, w):
        w = _f64(w)
        assert w.size == self.n_params
        lib().oracle_set_weights(self._h, _dp(w))

    def get_momentum(self):
        w = np.empty(self.n_params)
        lib().oracle_get_momentum(self._h, _dp(w))
        return w

    def set_momentum(self, w):
        w = _f64(w)
        assert w.size == self.n_params
        lib().oracle_set_momentum(self._h, _dp(w))


def argmax_rule(row):
    row = _f64(row)
    return lib().oracle_argmax(_dp(row), row.size)


class Sampler:
    """NeuralNetTrainer.sample (NNT:143-168) over indices 0..master-1."""

    def __init__(self, master_size, seed=1):
        self._h = lib().oracle_sampler_create(master_size, seed)

    def sample(self, batch):
        out = np.empty(batch, dtype=np.int32)
        n = lib().oracle_sampler_sample(self._h, batch, out.ctypes.data_as(C.POINTER(C.c_int32)))
        return out[:n].copy()

    def __del__(self):
        try:
            lib().oracle_sampler_destroy(self._h)
        except Exception:
            pass


def synthetic_batch(dims, B, seed, keep=1.0):
    """Deterministic (X, Y): X ~ U[0,1) from java.util.Random(seed) (thinned to density `keep`),
    Y one-hot with label nextInt(d_out) of Random(seed + 1)."""
    X = np.empty((B, dims[0]))
    Y = np.empty((B, dims[-1]))
    lib().oracle_fill_uniform(seed, X.size, keep, _dp(X))
    lib().oracle_fill_onehot(seed + 1, B, dims[-1], _dp(Y))
    return X, Y
