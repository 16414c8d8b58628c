"""ctypes binding of include/gnn_mlp.h (libgnn_mlp_hip.so).  No fallback: if the library is
missing or no GPU is visible every call raises."""
import ctypes as C
import os

from . import build as _build

_lib = None


class GnnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gnn_mlp status %d: %s" % (code, msg))
        self.code = code


OK, ERR_BAD_ARG, ERR_HIP, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_STATE = range(6)

# every symbol include/gnn_mlp.h declares: (name, restype, argtypes)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_u8 = C.POINTER(C.c_uint8)
_H = C.c_void_p
SYMBOLS = [
    ("gnn_mlp_create", C.c_int, [_ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(_H)]),
    ("gnn_mlp_destroy", C.c_int, [_H]),
    ("gnn_mlp_input_dim", C.c_int, [_H]),
    ("gnn_mlp_output_dim", C.c_int, [_H]),
    ("gnn_mlp_num_params", C.c_int64, [_H]),
    ("gnn_mlp_time", C.c_int, [_H]),
    ("gnn_mlp_last_error", C.c_char_p, []),
    ("gnn_mlp_propagate", C.c_int, [_H, _dp, C.c_int, _dp]),
    ("gnn_mlp_loss", C.c_int, [_H, _dp, _dp, C.c_int, _dp]),
    ("gnn_mlp_weight_gradient", C.c_int, [_H, _dp, _dp, C.c_int, _dp]),
    ("gnn_mlp_gradient_step", C.c_int, [_H, _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_int]),
    ("gnn_mlp_argmax", C.c_int, [_H, _dp, C.c_int, _ip]),
    ("gnn_mlp_get_weights", C.c_int, [_H, _dp]),
    ("gnn_mlp_set_weights", C.c_int, [_H, _dp]),
    ("gnn_mlp_get_momentum", C.c_int, [_H, _dp]),
    ("gnn_mlp_set_momentum", C.c_int, [_H, _dp]),
    ("gnn_mlp_save_checkpoint", C.c_int, [_H, C.c_char_p]),
    ("gnn_mlp_load_checkpoint", C.c_int, [_H, C.c_char_p]),
    ("gnn_mlp_upload_dataset", C.c_int, [_H, _dp, _dp, C.c_int64]),
    ("gnn_mlp_upload_dataset_u8", C.c_int, [_H, _u8, _u8, C.c_int64]),
    ("gnn_mlp_dataset_size", C.c_int64, [_H]),
    ("gnn_mlp_gradient_step_indexed", C.c_int, [_H, _ip, C.c_int, C.c_double, C.c_double, C.c_int]),
    ("gnn_mlp_gradient_step_range", C.c_int, [_H, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_int]),
    ("gnn_mlp_train_range", C.c_int, [_H, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double]),
    ("gnn_mlp_loss_range", C.c_int, [_H, C.c_int64, C.c_int, _dp]),
    ("gnn_mlp_argmax_range", C.c_int, [_H, C.c_int64, C.c_int, _ip]),
    ("gnn_mlp_count_hits_range", C.c_int, [_H, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    ("gnn_sampler_create", C.c_int, [C.c_int32, C.c_int64, C.POINTER(_H)]),
    ("gnn_sampler_destroy", C.c_int, [_H]),
    ("gnn_sampler_sample", C.c_int, [_H, C.c_int, _ip, C.POINTER(C.c_int)]),
    ("gnn_mlp_train_sampled", C.c_int, [_H, _H, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]),
    ("gnn_mlp_train_sampled_observed", C.c_int, [_H, _H, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, _dp]),
    ("gnn_mlp_grad_elems", C.c_int64, [_H]),
    ("gnn_mlp_grad_device_ptr", C.c_int, [_H, C.POINTER(C.c_void_p)]),
    ("gnn_mlp_bind_grad_buffer", C.c_int, [_H, C.c_void_p, C.c_int64]),
    ("gnn_mlp_set_stream", C.c_int, [_H, C.c_void_p]),
    ("gnn_mlp_compute_gradient_range", C.c_int, [_H, C.c_int64, C.c_int]),
    ("gnn_mlp_compute_gradient", C.c_int, [_H, _dp, _dp, C.c_int]),
    ("gnn_mlp_apply_update", C.c_int, [_H, C.c_int, C.c_double, C.c_double]),
    ("gnn_mlp_hint_next_range", C.c_int, [_H, C.c_int64, C.c_int]),
    ("gnn_mlp_synchronize", C.c_int, [_H]),
    ("gnn_mlp_rccl_unique_id", C.c_int, [C.c_void_p]),
    ("gnn_mlp_rccl_attach", C.c_int, [_H, C.c_void_p, C.c_int, C.c_int]),
    ("gnn_mlp_rccl_detach", C.c_int, [_H]),
    ("gnn_mlp_rccl_train_range", C.c_int, [_H, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double]),
    ("gnn_mlp_dp_create", C.c_int, [_ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, _ip, C.c_int, C.c_int, C.c_int, C.POINTER(_H)]),
    ("gnn_mlp_dp_destroy", C.c_int, [_H]),
    ("gnn_mlp_dp_num_replicas", C.c_int, [_H]),
    ("gnn_mlp_dp_replica", C.c_int, [_H, C.c_int, C.POINTER(_H)]),
    ("gnn_mlp_dp_gradient_step", C.c_int, [_H, _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_int]),
    ("gnn_mlp_dp_upload_dataset", C.c_int, [_H, _dp, _dp, C.c_int64]),
    ("gnn_mlp_dp_gradient_step_range", C.c_int, [_H, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_int]),
    ("gnn_mlp_dp_train_range", C.c_int, [_H, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double]),
    ("gnn_mlp_dp_set_weights", C.c_int, [_H, _dp]),
    ("gnn_mlp_dp_synchronize", C.c_int, [_H]),
    ("gnn_mlp_dp_replicas_identical", C.c_int, [_H, C.POINTER(C.c_int)]),
    ("gnn_mlp_forget_lookahead", C.c_int, [_H]),
    ("gnn_mlp_advance_time", C.c_int, [_H, C.c_int]),
    ("gnn_mlp_recover_stream", C.c_int, [_H]),
    ("gnn_mlp_specialize", C.c_int, [_H]),
    ("gnn_mlp_specialization", C.c_int, [_H]),
    ("gnn_mlp_step_launches", C.c_int, [_H]),
    ("gnn_mlp_plan_note", C.c_char_p, [_H]),
    ("gnn_mlp_rowblock_state", C.c_int, [_H]),
    ("gnn_mlp_timing_enable", C.c_int, [_H, C.c_int]),
    ("gnn_mlp_timing_read", C.c_int, [_H, C.c_int, _dp, C.POINTER(C.c_int64)]),
]


def lib_path():
    return _build.LIB_PATH


def load():
    """Loads libgnn_mlp_hip.so and binds every symbol; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950).  There is no CPU fallback for this path." % path)
    L = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(code):
    if code != OK:
        msg = load().gnn_mlp_last_error()
        raise GnnError(code, msg.decode() if msg else "")
