import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure); builds liboracle.so on first use."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gnn():
    """The product package.  GPU tests fail (not skip) if the HIP library is missing."""
    import gnn_amd
    gnn_amd.load_library()
    return gnn_amd


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
