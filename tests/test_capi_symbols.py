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


def test_java_binding_sources_are_complete_and_consistent():
    """java/ cannot be compiled here (no JDK), but it can be COMPLETE: every `native` method of HipNeuralNet.java
    has its Java_HipNeuralNet_* function in gnn_mlp_jni.c and vice versa, every gnn_mlp_* the shim calls is
    declared in include/gnn_mlp.h, and no JNI critical section wraps the (blocking) library calls."""
    import re
    java = open(os.path.join(ROOT, "java", "HipNeuralNet.java")).read()
    shim = open(os.path.join(ROOT, "java", "gnn_mlp_jni.c")).read()
    header = open(os.path.join(ROOT, "include", "gnn_mlp.h")).read()
    natives = set(re.findall(r"private static native \w+ (native\w+)\(", java))
    impl = set(re.findall(r"Java_HipNeuralNet_(native\w+)\(", shim))
    assert natives and natives == impl, (natives ^ impl)
    for fn in set(re.findall(r"\b(gnn_mlp_\w+)\(", shim)):
        assert re.search(r"\b%s\(" % fn, header), fn
    code = re.sub(r"/\*.*?\*/", "", shim, flags=re.S)      # comments may name what is NOT used
    assert "GetPrimitiveArrayCritical" not in code
    for cls in ("HipSoftmaxCrossEntropyNeuralNet", "HipGeneralNeuralNet", "TrainLog"):
        assert os.path.exists(os.path.join(ROOT, "java", cls + ".java")), cls
    # the row format of MNISTTrainer.logTest (MT:211-219) as the reference's own log shows it (logs/trainLog.csv:4)
    tl = open(os.path.join(ROOT, "java", "TrainLog.java")).read()
    assert '",%d,%.5f,%d,%.3f,"' in tl and '"%.5f,%.5f\\n"' in tl and '"%d,%.2f\\n"' in tl
