"""MI355X-native (gfx950) MLP mini-batch SGD path behind the reference's NeuralNet interface.

The directory name carries the reference's name and is not a Python identifier; import it as

    import gnn_amd                      # repo-root shim (gnn_amd.py)

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI), the ctypes binding and
the host-side mirror of the NeuralNet / NeuralNetTrainer interface.
"""
from . import build as build_lib  # noqa: F401
from ._capi import GnnError, load as load_library, lib_path  # noqa: F401
from .neural_net import (  # noqa: F401
    ACT_IDENTITY, ACT_LEAKY_RELU, ACT_RELU, ACT_SIGMOID, ACT_TANH, DTYPE_BF16, DTYPE_F32,
    LOSS_HALF_SQUARED, OUT_ACT_LOSS, OUT_SOFTMAX_CE, REDUCE_DIRECT, REDUCE_DIRECT_RS, REDUCE_RCCL, DataParallelNeuralNet, GeneralNeuralNet,
    NeuralNet, SoftmaxCrossEntropyNeuralNet)
from .trainer import (  # noqa: F401
    NeuralNetTrainer, Sampler, accuracy, log_test, read_idx_images, read_idx_labels, train_log_row)
