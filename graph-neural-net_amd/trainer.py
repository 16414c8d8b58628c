"""Host-side mirror of NeuralNetTrainer.java and of MNISTTrainer's data handling, over the C ABI.

NeuralNetTrainer (NNT:11-170): epoch sampler without replacement, `train` loop, 1 % validation
loss with the observer line format "%d,%.2f".  The training set lives in HBM (uploaded once by
the constructor); a sample() draw is a list of row indices, gathered on the GPU.

MNIST side (MNISTTrainer.java): IDX parsing (MT:26-66, 76-118) -- raw bytes go to the GPU, which
applies the reference's encoding (pixel/255.0, one-hot) -- and the accuracy loops with the `>=`
argmax (MT:159-197).
"""
import ctypes as C
import struct

import numpy as np

from . import _capi


class Sampler:
    """NNT:143-168 over dataset row indices (gnn_sampler_* in include/gnn_mlp.h)."""

    def __init__(self, master_size, seed=1):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        _capi.check(self._lib.gnn_sampler_create(int(master_size), int(seed), C.byref(self._h)))
        self.master_size = int(master_size)

    def sample(self, batch_size):
        out = np.empty(int(batch_size), dtype=np.int32)
        n = C.c_int()
        _capi.check(self._lib.gnn_sampler_sample(self._h, int(batch_size), out.ctypes.data_as(C.POINTER(C.c_int32)),
                                                 C.byref(n)))
        return out[:n.value].copy()

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.gnn_sampler_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NeuralNetTrainer:
    """NeuralNetTrainer(Map<double[],double[]> data, NeuralNet net) (NNT:28-43): `data` is given as
    two row-aligned matrices (or raw uint8 pixels + labels); master order = row order."""

    def __init__(self, data_x, data_y, net, raw_u8=False, seed=1):
        self.net = net
        if raw_u8:
            net.upload_dataset_u8(data_x, data_y)
        else:
            net.upload_dataset(data_x, data_y)
        self.size = net.dataset_size
        self.sampler = Sampler(self.size, seed)       # random = new Random(1) (NNT:42)

    def train(self, iterations, stepSize, batchSize, momentum, noise=False, monitor=None, observer=None):
        """NNT:60-92.  observer: a text stream receiving "%d,%.2f\\n" % (i, validation loss) per
        iteration (NNT:71); monitor: an object with step() / finish() (ProgressBar)."""
        if not (iterations > 0 and stepSize > 0):
            raise ValueError("iterations and stepSize must be positive (NNT:62)")
        if not (0 < batchSize < self.size):
            raise ValueError("batchSize must be positive and below the data size (NNT:63)")
        validation_size = self.size // 100 + 1                       # NNT:65
        if observer is None and monitor is None:                     # fastest variant, NNT:88-90
            _capi.check(self.net._lib.gnn_mlp_train_sampled(self.net._h, self.sampler._h, int(iterations),
                                                            int(batchSize), float(stepSize), float(momentum),
                                                            int(bool(noise))))
            return
        if observer is None:                                         # only progress monitored, NNT:75-79 has no validation
            done = 0
            while done < iterations:                                 # (the monitor steps in bursts: a device loop per burst)
                n = min(self.OBSERVER_BURST, iterations - done)
                _capi.check(self.net._lib.gnn_mlp_train_sampled(self.net._h, self.sampler._h, n, int(batchSize), float(stepSize),
                                                                float(momentum), int(bool(noise))))
                for _ in range(n):
                    monitor.step()
                done += n
            monitor.finish()
            return
        # observed (NNT:68-72 with a monitor, NNT:75-79 without): gradientStep + validate(validationSize) per iteration, both on
        # the device; the validation losses of a burst come back in ONE readback and are printed in the reference's format
        done = 0
        while done < iterations:
            n = min(self.OBSERVER_BURST, iterations - done) if monitor is not None else iterations - done
            val = np.empty(n)
            _capi.check(self.net._lib.gnn_mlp_train_sampled_observed(
                self.net._h, self.sampler._h, n, int(batchSize), float(stepSize), float(momentum), int(bool(noise)),
                int(validation_size), val.ctypes.data_as(C.POINTER(C.c_double))))
            for k in range(n):
                observer.write("%d,%.2f\n" % (done + k, val[k]))      # NNT:71
                if monitor is not None:
                    monitor.step()
            done += n
        if monitor is not None:
            monitor.finish()

    OBSERVER_BURST = 256   # iterations per device loop when a progress monitor wants to be stepped

    def train_stepwise(self, iterations, stepSize, batchSize, momentum, noise=False, observer=None):
        """The observed loop NNT:75-79 one ABI call per action (sample on the host, an indexed step, a validation readback per
        iteration) -- the form `train` had until round 4; kept for the tests that compare the device loop against it."""
        validation_size = self.size // 100 + 1
        for i in range(iterations):
            self.net.gradient_step_indexed(self.sampler.sample(batchSize), stepSize, momentum, noise)
            if observer is not None:
                observer.write("%d,%.2f\n" % (i, self.validate(validation_size)))

    def validate(self, batchSize):
        """NNT:102-113: mean loss over the first batchSize samples in master order."""
        total, first = 0.0, 0
        while first < batchSize:
            n = min(self.net.max_batch, batchSize - first)
            total += float(self.net.loss_range(first, n).sum())
            first += n
        return total / batchSize


# ---- MNIST (MNISTTrainer.java) ---------------------------------------------------------------
def read_idx_images(path):
    """MT:37-47: magic 2051, count, rows, cols (big-endian int32, MT:76-80), then raw bytes."""
    with open(path, "rb") as f:
        magic, n, rows, cols = struct.unpack(">iiii", f.read(16))
        if magic != 2051:
            raise ValueError("%s: bad IDX image magic %d (MT:39 asserts 2051)" % (path, magic))
        data = np.frombuffer(f.read(n * rows * cols), dtype=np.uint8)
    if data.size != n * rows * cols:
        raise ValueError("%s: truncated" % path)
    return data.reshape(n, rows * cols)


def read_idx_labels(path):
    """MT:38, 40: magic 2049, count, then one byte per label (MT:112-118)."""
    with open(path, "rb") as f:
        magic, n = struct.unpack(">ii", f.read(8))
        if magic != 2049:
            raise ValueError("%s: bad IDX label magic %d (MT:38 asserts 2049)" % (path, magic))
        data = np.frombuffer(f.read(n), dtype=np.uint8)
    if data.size != n:
        raise ValueError("%s: truncated" % path)
    return data


def accuracy(net, labels=None, first=0, n=None):
    """testOnTrainingData / testOnTestData (MT:159-197) over dataset rows [first, first+len):
    hit when the `>=` argmax of propagate() equals the label.  With labels = None the expected classes are the dataset's own
    expected rows (MT:186-188) and the whole loop runs on the device (gnn_mlp_count_hits_range: one readback)."""
    if labels is None:
        n = net.dataset_size - first if n is None else n
        return net.count_hits_range(first, n) / n
    labels = np.asarray(labels)
    hits, off = 0, 0
    while off < labels.size:
        n = min(net.max_batch, labels.size - off)
        hits += int((net.argmax_range(first + off, n) == labels[off:off + n]).sum())
        off += n
    return hits / labels.size


def train_log_row(net_dim, iterations, step_size, batch_size, momentum, noise, training_acc, test_acc):
    """The row MNISTTrainer.logTest appends to logs/trainLog.csv (MT:211-219):
    `dims,iterations,step,batch,momentum,noise,trainAcc,testAcc` in the reference's number formats
    (`%d`, `%.5f`, `%d`, `%.3f`, Java's boolean text, `%.5f,%.5f`)."""
    dims = "-".join(str(int(d)) for d in net_dim)
    return "%s,%d,%.5f,%d,%.3f,%s,%.5f,%.5f\n" % (dims, iterations, step_size, batch_size, momentum,
                                                 "true" if noise else "false", training_acc, test_acc)


def log_test(path, net_dim, iterations, step_size, batch_size, momentum, noise, training_acc, test_acc):
    """Appends one train_log_row to `path` (the reference opens logs/trainLog.csv in append mode, MT:61)."""
    import os
    d = os.path.dirname(str(path))
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "a") as f:
        f.write(train_log_row(net_dim, iterations, step_size, batch_size, momentum, noise, training_acc, test_acc))
