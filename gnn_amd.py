"""Import shim: `import gnn_amd` loads the package in ./graph-neural-net_amd/ (a directory name
that is not a Python identifier) under the module name `gnn_amd`."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph-neural-net_amd")
_spec = importlib.util.spec_from_file_location(
    "gnn_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gnn_amd"] = _mod
_spec.loader.exec_module(_mod)
