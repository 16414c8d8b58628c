"""The C-ABI library loads on a CPU-only box and exports every symbol include/gnn_mlp.h
declares (no compute calls here: without a GPU they fail loudly, which is also checked)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gnn_mlp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gnn_(?:mlp|sampler)_\w+)\s*\(", text)))


def test_header_declares_the_interface():
    names = declared_symbols()
    for required in ("gnn_mlp_create", "gnn_mlp_destroy", "gnn_mlp_propagate", "gnn_mlp_loss",
                     "gnn_mlp_weight_gradient", "gnn_mlp_gradient_step", "gnn_mlp_input_dim",
                     "gnn_mlp_output_dim", "gnn_mlp_argmax", "gnn_mlp_get_weights", "gnn_mlp_set_weights",
                     "gnn_mlp_get_momentum", "gnn_mlp_last_error"):
        assert required in names


def test_library_exports_every_declared_symbol(gnn):
    lib = ctypes.CDLL(gnn.lib_path())
    for name in declared_symbols():
        assert hasattr(lib, name), "libgnn_mlp_hip.so does not export %s" % name


def test_binding_covers_every_declared_symbol(gnn):
    import gnn_amd._capi as capi
    bound = sorted(n for n, _, _ in capi.SYMBOLS)
    assert bound == declared_symbols()


def test_no_silent_cpu_fallback(gnn):
    """Without a GPU the product path must fail loudly (GNN_ERR_NO_DEVICE), never compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(gnn.GnnError) as e:
        gnn.SoftmaxCrossEntropyNeuralNet([4, 3, 2])
    assert e.value.code == 4


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "graph-neural-net_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.lower() or fn == "__none__", "%s mentions the oracle" % fn
