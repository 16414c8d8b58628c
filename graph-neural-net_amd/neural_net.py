"""Host-side mirror of the reference's `NeuralNet` operator interface (NeuralNet.java:7-67) and
its two implementations, over the C ABI of include/gnn_mlp.h.

The reference is Java and the image has no JDK, so the literal Java classes with `native`
methods live only as source in INTEGRATION.md / java/; this module is the same interface (same
method names, argument meaning and error behaviour) in Python for the parity tests and the
bench.  Method names follow the Java ones; snake_case aliases are provided.

Every method calls the HIP library; nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from . import _capi

ACT_LEAKY_RELU, ACT_SIGMOID, ACT_TANH, ACT_RELU, ACT_IDENTITY = range(5)
OUT_SOFTMAX_CE, OUT_ACT_LOSS = 0, 1
LOSS_HALF_SQUARED = 0
DTYPE_F32, DTYPE_BF16 = 0, 1

_ACT_NAMES = {"leaky_relu": ACT_LEAKY_RELU, "sigmoid": ACT_SIGMOID, "tanh": ACT_TANH,
              "relu": ACT_RELU, "identity": ACT_IDENTITY}


def _act(a):
    if isinstance(a, str):
        return _ACT_NAMES[a]
    return int(a)


def _f64(a, cols):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if a.ndim != 2 or a.shape[1] != cols:
        # the reference asserts input.length == layerDims[0] (SCE:166)
        raise ValueError("expected rows of length %d, got shape %r" % (cols, a.shape))
    return a


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class NeuralNet:
    """NeuralNet.java:7-67 -- propagate / calculateLoss / calculateWeightGradient /
    gradientStep / getInputDim / getOutputDim, batched: a 2-D array is a batch of rows."""

    def __init__(self, layer_dims, out_kind, inner_act, last_act, loss, seed=1, dtype=DTYPE_F32,
                 device=0, max_batch=1024):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        dims = [int(d) for d in layer_dims]
        arr = (C.c_int32 * len(dims))(*dims)
        _capi.check(self._lib.gnn_mlp_create(arr, len(dims), out_kind, _act(inner_act), _act(last_act),
                                             loss, seed, dtype, device, max_batch, C.byref(self._h)))
        self.layer_dims = dims
        self.max_batch = max_batch
        self.n_params = self._lib.gnn_mlp_num_params(self._h)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.gnn_mlp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- NeuralNet.java ---------------------------------------------------------------------
    def getInputDim(self):
        return self._lib.gnn_mlp_input_dim(self._h)

    def getOutputDim(self):
        return self._lib.gnn_mlp_output_dim(self._h)

    def propagate(self, input):
        """NN:16.  1-D input -> 1-D output; 2-D -> one output row per input row."""
        single = np.ndim(input) == 1
        X = _f64(input, self.layer_dims[0])
        out = np.empty((X.shape[0], self.layer_dims[-1]))
        _capi.check(self._lib.gnn_mlp_propagate(self._h, _dp(X), X.shape[0], _dp(out)))
        return out[0] if single else out

    def calculateLoss(self, input, expected):
        """NN:27."""
        single = np.ndim(input) == 1
        X = _f64(input, self.layer_dims[0])
        Y = _f64(expected, self.layer_dims[-1])
        if X.shape[0] != Y.shape[0]:
            raise ValueError("input / expected row counts differ")
        out = np.empty(X.shape[0])
        _capi.check(self._lib.gnn_mlp_loss(self._h, _dp(X), _dp(Y), X.shape[0], _dp(out)))
        return float(out[0]) if single else out

    def calculateWeightGradient(self, input, expected):
        """NN:39.  Returns {layer index: ndarray[d_l, d_{l+1}]} like the reference's
        Map<Integer,double[][]>; for a batch the per-sample gradients are summed."""
        X = _f64(input, self.layer_dims[0])
        Y = _f64(expected, self.layer_dims[-1])
        flat = np.empty(self.n_params)
        _capi.check(self._lib.gnn_mlp_weight_gradient(self._h, _dp(X), _dp(Y), X.shape[0], _dp(flat)))
        return self._split(flat)

    def gradientStep(self, batch, step, momentum, noise=False, expected=None):
        """NN:51.  `batch` is either a mapping/sequence of (input, expected) pairs like the
        reference's Map<double[],double[]> (iteration order = row order), or an input matrix
        with `expected` given separately."""
        if expected is None:
            items = list(batch.items()) if hasattr(batch, "items") else list(batch)
            if not items:
                raise ValueError("batch must be non-empty (SCE:300)")
            X = np.stack([np.asarray(k, dtype=np.float64) for k, _ in items])
            Y = np.stack([np.asarray(v, dtype=np.float64) for _, v in items])
        else:
            X, Y = batch, expected
        X = _f64(X, self.layer_dims[0])
        Y = _f64(Y, self.layer_dims[-1])
        if X.shape[0] != Y.shape[0]:
            raise ValueError("input / expected row counts differ")
        _capi.check(self._lib.gnn_mlp_gradient_step(self._h, _dp(X), _dp(Y), X.shape[0], float(step),
                                                    float(momentum), int(bool(noise))))

    # snake_case aliases
    get_input_dim = getInputDim
    get_output_dim = getOutputDim
    calculate_loss = calculateLoss
    calculate_weight_gradient = calculateWeightGradient
    gradient_step = gradientStep

    # -- extensions needed for parity tests (weights are private in the reference, SCE:15) ----
    def argmax(self, input):
        X = _f64(input, self.layer_dims[0])
        out = np.empty(X.shape[0], dtype=np.int32)
        _capi.check(self._lib.gnn_mlp_argmax(self._h, _dp(X), X.shape[0], out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def _split(self, flat):
        out, off = {}, 0
        for l in range(len(self.layer_dims) - 1):
            n = self.layer_dims[l] * self.layer_dims[l + 1]
            out[l] = flat[off:off + n].reshape(self.layer_dims[l], self.layer_dims[l + 1])
            off += n
        return out

    def get_weights(self):
        flat = np.empty(self.n_params)
        _capi.check(self._lib.gnn_mlp_get_weights(self._h, _dp(flat)))
        return flat

    def set_weights(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float64).ravel()
        if flat.size != self.n_params:
            raise ValueError("expected %d weights" % self.n_params)
        _capi.check(self._lib.gnn_mlp_set_weights(self._h, _dp(flat)))

    def get_momentum(self):
        flat = np.empty(self.n_params)
        _capi.check(self._lib.gnn_mlp_get_momentum(self._h, _dp(flat)))
        return flat

    def set_momentum(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float64).ravel()
        if flat.size != self.n_params:
            raise ValueError("expected %d values" % self.n_params)
        _capi.check(self._lib.gnn_mlp_set_momentum(self._h, _dp(flat)))

    @property
    def time(self):
        return self._lib.gnn_mlp_time(self._h)

    def save_checkpoint(self, path):
        _capi.check(self._lib.gnn_mlp_save_checkpoint(self._h, str(path).encode()))

    def load_checkpoint(self, path):
        _capi.check(self._lib.gnn_mlp_load_checkpoint(self._h, str(path).encode()))

    # -- device-resident data (the trainer's side of the boundary, NNT:28-43,143-168) ---------
    def upload_dataset(self, X, Y):
        X = _f64(X, self.layer_dims[0])
        Y = _f64(Y, self.layer_dims[-1])
        if X.shape[0] != Y.shape[0]:
            raise ValueError("input / expected row counts differ")
        _capi.check(self._lib.gnn_mlp_upload_dataset(self._h, _dp(X), _dp(Y), X.shape[0]))

    def upload_dataset_u8(self, pixels, labels):
        pixels = np.ascontiguousarray(pixels, dtype=np.uint8).reshape(-1, self.layer_dims[0])
        labels = np.ascontiguousarray(labels, dtype=np.uint8).ravel()
        if pixels.shape[0] != labels.size:
            raise ValueError("pixel / label row counts differ")
        u8 = C.POINTER(C.c_uint8)
        _capi.check(self._lib.gnn_mlp_upload_dataset_u8(self._h, pixels.ctypes.data_as(u8),
                                                        labels.ctypes.data_as(u8), labels.size))

    @property
    def dataset_size(self):
        return self._lib.gnn_mlp_dataset_size(self._h)

    def gradient_step_indexed(self, idx, step, momentum, noise=False):
        idx = np.ascontiguousarray(idx, dtype=np.int32).ravel()
        _capi.check(self._lib.gnn_mlp_gradient_step_indexed(
            self._h, idx.ctypes.data_as(C.POINTER(C.c_int32)), idx.size, float(step), float(momentum),
            int(bool(noise))))

    def gradient_step_range(self, first, B, step, momentum, noise=False):
        _capi.check(self._lib.gnn_mlp_gradient_step_range(self._h, int(first), int(B), float(step),
                                                          float(momentum), int(bool(noise))))

    def train_range(self, first, B, n_steps, step, momentum):
        _capi.check(self._lib.gnn_mlp_train_range(self._h, int(first), int(B), int(n_steps), float(step),
                                                  float(momentum)))

    def loss_range(self, first, B):
        out = np.empty(B)
        _capi.check(self._lib.gnn_mlp_loss_range(self._h, int(first), int(B), _dp(out)))
        return out

    def argmax_range(self, first, B):
        out = np.empty(B, dtype=np.int32)
        _capi.check(self._lib.gnn_mlp_argmax_range(self._h, int(first), int(B),
                                                   out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def count_hits_range(self, first=0, n=None):
        """testOnTrainingData / testOnTestData (MT:159-197) over dataset rows [first, first + n): the number of rows whose `>=`
        argmax equals the expected class (the last index whose expected value is 1); one readback."""
        n = self.dataset_size - int(first) if n is None else int(n)
        out = C.c_int64()
        _capi.check(self._lib.gnn_mlp_count_hits_range(self._h, int(first), n, C.byref(out)))
        return out.value

    # -- data-parallel hooks ------------------------------------------------------------------
    @property
    def grad_elems(self):
        return self._lib.gnn_mlp_grad_elems(self._h)

    def grad_device_ptr(self):
        p = C.c_void_p()
        _capi.check(self._lib.gnn_mlp_grad_device_ptr(self._h, C.byref(p)))
        return p.value

    def bind_grad_buffer(self, dev_ptr, n_elems):
        _capi.check(self._lib.gnn_mlp_bind_grad_buffer(self._h, C.c_void_p(dev_ptr), int(n_elems)))

    def set_stream(self, hip_stream):
        _capi.check(self._lib.gnn_mlp_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def compute_gradient_range(self, first, B_local):
        _capi.check(self._lib.gnn_mlp_compute_gradient_range(self._h, int(first), int(B_local)))

    def compute_gradient(self, X, Y):
        X = _f64(X, self.layer_dims[0])
        Y = _f64(Y, self.layer_dims[-1])
        _capi.check(self._lib.gnn_mlp_compute_gradient(self._h, _dp(X), _dp(Y), X.shape[0]))

    def apply_update(self, B_global, step, momentum):
        _capi.check(self._lib.gnn_mlp_apply_update(self._h, int(B_global), float(step), float(momentum)))

    # -- one process per GPU, the exchange inside the library's step loop (gnn_mlp_rccl_*) --------
    @staticmethod
    def rccl_unique_id():
        """128 bytes from ncclGetUniqueId: rank 0 makes them, the caller hands them to every rank."""
        buf = C.create_string_buffer(128)
        _capi.check(_capi.load().gnn_mlp_rccl_unique_id(buf))
        return buf.raw

    def rccl_attach(self, unique_id, n_ranks, rank):
        if len(unique_id) != 128:
            raise ValueError("an RCCL unique id is 128 bytes")
        _capi.check(self._lib.gnn_mlp_rccl_attach(self._h, C.c_char_p(bytes(unique_id)), int(n_ranks), int(rank)))

    def rccl_detach(self):
        _capi.check(self._lib.gnn_mlp_rccl_detach(self._h))

    def rccl_train_range(self, first, B_local, n_steps, step, momentum):
        """n_steps global gradientSteps: this rank's rows [first + s B_local, ...) per step, one ncclAllReduce of the flat gradient
        per step inside the call, the identical update with batchSize = B_local * n_ranks."""
        _capi.check(self._lib.gnn_mlp_rccl_train_range(self._h, int(first), int(B_local), int(n_steps), float(step), float(momentum)))

    def hint_next_range(self, first, B):
        """The next gradient computation will run on dataset rows [first, first+B) (speed only)."""
        _capi.check(self._lib.gnn_mlp_hint_next_range(self._h, int(first), int(B)))

    def synchronize(self):
        _capi.check(self._lib.gnn_mlp_synchronize(self._h))

    def forget_lookahead(self):
        """Drop the first-layer sums made ahead for the next batch and any pending hint (results unchanged)."""
        _capi.check(self._lib.gnn_mlp_forget_lookahead(self._h))

    def advance_time(self, steps):
        _capi.check(self._lib.gnn_mlp_advance_time(self._h, int(steps)))

    def recover_stream(self):
        """After a failed stream capture: leave capture mode, drop the sticky HIP error."""
        _capi.check(self._lib.gnn_mlp_recover_stream(self._h))

    # -- shape specialisation -----------------------------------------------------------------
    def specialize(self):
        """Instantiate the fused path's kernels for this net's layer sizes (hiprtc)."""
        _capi.check(self._lib.gnn_mlp_specialize(self._h))
        return self.specialization

    @property
    def specialization(self):
        """0 generic kernels, 1 prebuilt instantiation, 2 instantiated at run time."""
        return self._lib.gnn_mlp_specialization(self._h)

    @property
    def step_launches(self):
        """2 two-launch path, 3 fused three-launch path, 0 per-layer GEMMs."""
        return self._lib.gnn_mlp_step_launches(self._h)

    @property
    def rowblock_state(self):
        """The two-launch step's training kernel: 0 middle4_kernel, 1 rowblock runtime shape, 2 prebuilt, 3 instantiated at run time."""
        return self._lib.gnn_mlp_rowblock_state(self._h)

    @property
    def plan_note(self):
        """Why the net is not on the two-launch path ('' when it is)."""
        s = self._lib.gnn_mlp_plan_note(self._h)
        return s.decode() if s else ""

    # -- measurement --------------------------------------------------------------------------
    def timing_enable(self, on=True):
        _capi.check(self._lib.gnn_mlp_timing_enable(self._h, int(bool(on))))

    def timing_read(self, which):
        mean = C.c_double()
        cnt = C.c_int64()
        _capi.check(self._lib.gnn_mlp_timing_read(self._h, int(which), C.byref(mean), C.byref(cnt)))
        return mean.value, cnt.value


class SoftmaxCrossEntropyNeuralNet(NeuralNet):
    """SoftmaxCrossEntropyNeuralNet.java: configurable inner activation, softmax output,
    cross-entropy loss.  ctor SCE:103 takes (layerDims, innerActivationFunc,
    innerActivationPrime); the two lambdas become one `inner_act` enum value (the shipped pair is
    leaky ReLU, MT:234-235)."""

    def __init__(self, layer_dims, inner_act=ACT_LEAKY_RELU, seed=1, dtype=DTYPE_F32, device=0,
                 max_batch=1024):
        super().__init__(layer_dims, OUT_SOFTMAX_CE, inner_act, ACT_IDENTITY, LOSS_HALF_SQUARED,
                         seed=seed, dtype=dtype, device=device, max_batch=max_batch)


class GeneralNeuralNet(NeuralNet):
    """GeneralNeuralNet.java: ctor GNN:112-115 takes (layerDims, inner f, inner f', last f,
    last f', loss, loss'); each (f, f') pair becomes one enum value, (loss, loss') one."""

    def __init__(self, layer_dims, inner_act=ACT_SIGMOID, last_act=ACT_SIGMOID, loss=LOSS_HALF_SQUARED,
                 seed=1, dtype=DTYPE_F32, device=0, max_batch=1024):
        super().__init__(layer_dims, OUT_ACT_LOSS, inner_act, last_act, loss, seed=seed, dtype=dtype,
                         device=device, max_batch=max_batch)


REDUCE_RCCL, REDUCE_DIRECT, REDUCE_DIRECT_RS = 0, 1, 2


class _ReplicaView(NeuralNet):
    """A replica of a DataParallelNeuralNet seen through the single-net interface (borrowed handle:
    closing the view does not destroy it)."""

    def __init__(self, lib, handle, layer_dims, max_batch):
        self._lib, self._h = lib, handle
        self.layer_dims = list(layer_dims)
        self.max_batch = max_batch
        self.n_params = lib.gnn_mlp_num_params(handle)

    def close(self):
        self._h = C.c_void_p()


class DataParallelNeuralNet:
    """The NeuralNet interface over gnn_mlp_dp_* (include/gnn_mlp.h): ONE object, N device replicas,
    gradientStep sharded by rows inside the library (what a single-threaded JVM caller uses; the
    one-process-per-GPU form is data_parallel.py).  propagate / calculateLoss / argmax run on replica 0
    (all replicas are bitwise identical)."""

    def __init__(self, layer_dims, devices=(0,), out_kind=OUT_SOFTMAX_CE, inner_act=ACT_LEAKY_RELU,
                 last_act=ACT_IDENTITY, loss=LOSS_HALF_SQUARED, seed=1, dtype=DTYPE_F32, max_batch=1024,
                 reducer=REDUCE_RCCL):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        dims = [int(d) for d in layer_dims]
        devs = [int(d) for d in devices]
        _capi.check(self._lib.gnn_mlp_dp_create((C.c_int32 * len(dims))(*dims), len(dims), out_kind, _act(inner_act),
                                                _act(last_act), loss, seed, dtype, (C.c_int32 * len(devs))(*devs),
                                                len(devs), max_batch, reducer, C.byref(self._h)))
        self.layer_dims, self.devices, self.max_batch = dims, devs, max_batch
        self.replicas = []
        for r in range(len(devs)):
            h = C.c_void_p()
            _capi.check(self._lib.gnn_mlp_dp_replica(self._h, r, C.byref(h)))
            self.replicas.append(_ReplicaView(self._lib, h, dims, max_batch))
        self.n_params = self.replicas[0].n_params

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            for v in self.replicas:
                v.close()
            self._lib.gnn_mlp_dp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # NeuralNet.java
    def getInputDim(self):
        return self.layer_dims[0]

    def getOutputDim(self):
        return self.layer_dims[-1]

    def propagate(self, input):
        return self.replicas[0].propagate(input)

    def calculateLoss(self, input, expected):
        return self.replicas[0].calculateLoss(input, expected)

    def calculateWeightGradient(self, input, expected):
        return self.replicas[0].calculateWeightGradient(input, expected)

    def argmax(self, input):
        return self.replicas[0].argmax(input)

    def gradientStep(self, batch, step, momentum, noise=False, expected=None):
        """NN:51, the rows dealt to the replicas in contiguous blocks."""
        if expected is None:
            items = list(batch.items()) if hasattr(batch, "items") else list(batch)
            if not items:
                raise ValueError("batch must be non-empty (SCE:300)")
            X = np.stack([np.asarray(k, dtype=np.float64) for k, _ in items])
            Y = np.stack([np.asarray(v, dtype=np.float64) for _, v in items])
        else:
            X, Y = batch, expected
        X = _f64(X, self.layer_dims[0])
        Y = _f64(Y, self.layer_dims[-1])
        if X.shape[0] != Y.shape[0]:
            raise ValueError("input / expected row counts differ")
        _capi.check(self._lib.gnn_mlp_dp_gradient_step(self._h, _dp(X), _dp(Y), X.shape[0], float(step),
                                                       float(momentum), int(bool(noise))))

    gradient_step = gradientStep

    def upload_dataset(self, X, Y):
        X = _f64(X, self.layer_dims[0])
        Y = _f64(Y, self.layer_dims[-1])
        _capi.check(self._lib.gnn_mlp_dp_upload_dataset(self._h, _dp(X), _dp(Y), X.shape[0]))

    def gradient_step_range(self, first, B, step, momentum, noise=False):
        _capi.check(self._lib.gnn_mlp_dp_gradient_step_range(self._h, int(first), int(B), float(step), float(momentum),
                                                             int(bool(noise))))

    def train_range(self, first, B, n_steps, step, momentum):
        _capi.check(self._lib.gnn_mlp_dp_train_range(self._h, int(first), int(B), int(n_steps), float(step), float(momentum)))

    def set_weights(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float64).ravel()
        _capi.check(self._lib.gnn_mlp_dp_set_weights(self._h, _dp(flat)))

    def get_weights(self):
        return self.replicas[0].get_weights()

    def get_momentum(self):
        return self.replicas[0].get_momentum()

    @property
    def time(self):
        return self.replicas[0].time

    def synchronize(self):
        _capi.check(self._lib.gnn_mlp_dp_synchronize(self._h))

    def replicas_identical(self):
        out = C.c_int()
        _capi.check(self._lib.gnn_mlp_dp_replicas_identical(self._h, C.byref(out)))
        return bool(out.value)
