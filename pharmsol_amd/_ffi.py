"""ctypes binding of ``libpmx_hip.so`` (the C ABI of ``include/pmx.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make``.  If it is
missing this module raises: there is no Python or CPU fallback for the compute path.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# PMX_LIB: developer override for same-device A/B runs of two builds (tools/ab_build.sh)
LIB_PATH = os.environ.get("PMX_LIB") or os.path.join(_HERE, "lib", "libpmx_hip.so")
_lib = None

# every symbol include/pmx.h declares: (name, restype, argtypes)
_PD = C.POINTER(_abi.pmx_population_desc)
_MD = C.POINTER(_abi.pmx_model_desc)
SYMBOLS = [
    ("pmx_abi_version", C.c_int32, []),
    ("pmx_sizeof_model_desc", C.c_int64, []),
    ("pmx_sizeof_population_desc", C.c_int64, []),
    ("pmx_sizeof_struct", C.c_int64, [C.c_char_p]),
    ("pmx_device_count", C.c_int32, []),
    ("pmx_population_create", C.c_int32, [_PD, C.c_int32, C.POINTER(C.c_void_p)]),
    ("pmx_population_create_shard", C.c_int32, [_PD, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]),
    ("pmx_shard_bounds", C.c_int32, [_PD, C.c_int32, C.c_void_p]),
    ("pmx_shard_rows", C.c_int32, [_PD, C.c_int32, C.c_void_p, C.c_void_p]),
    ("pmx_comm_unique_id", C.c_int32, [C.c_void_p]),
    ("pmx_comm_create", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    ("pmx_comm_destroy", None, [C.c_void_p]),
    ("pmx_comm_size", C.c_int32, [C.c_void_p]),
    ("pmx_comm_rank", C.c_int32, [C.c_void_p]),
    ("pmx_allgather_predictions", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ("pmx_population_destroy", None, [C.c_void_p]),
    ("pmx_population_n_subjects", C.c_int64, [C.c_void_p]),
    ("pmx_population_n_observations", C.c_int64, [C.c_void_p]),
    ("pmx_population_n_events", C.c_int64, [C.c_void_p]),
    ("pmx_population_device", C.c_int32, [C.c_void_p]),
    ("pmx_population_observation_offsets", C.c_int32, [C.c_void_p, C.c_void_p]),
    ("pmx_population_observation_info", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pmx_model_create", C.c_int32, [_MD, C.POINTER(C.c_void_p)]),
    ("pmx_model_create_custom", C.c_int32, [_MD, C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]),
    ("pmx_debug_jit_source", C.c_int32, [_MD, C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]),
    ("pmx_model_create_user", C.c_int32, [_MD, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("pmx_debug_jit_source_user", C.c_int32, [_MD, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("pmx_free_text", None, [C.c_void_p]),
    ("pmx_model_destroy", None, [C.c_void_p]),
    ("pmx_predict", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    ("pmx_predict_device", C.c_int32,
     [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("pmx_prediction_buffer_create", C.c_int32,
     [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_double)]),
    ("pmx_prediction_buffer_create_pitched", C.c_int32,
     [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_double)]),
    ("pmx_prediction_buffer_destroy", None, [C.c_void_p]),
    ("pmx_time_predict_device", C.c_int32,
     [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.POINTER(C.c_double)]),
    ("pmx_predict_state_device", C.c_int32,
     [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("pmx_predict_batch", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pmx_predict_batch_device", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pmx_loglik", C.c_int32,
     [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    ("pmx_loglik_device", C.c_int32,
     [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("pmx_loglik_batch", C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pmx_loglik_batch_device", C.c_int32,
     [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pmx_recommended_ld", C.c_int64, [C.c_int64]),
    ("pmx_measure_write_ceiling", C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.POINTER(C.c_double)]),
    ("pmx_host_alloc", C.c_int32, [C.c_int64, C.POINTER(C.c_void_p)]),
    ("pmx_host_free", None, [C.c_void_p]),
    ("pmx_last_kernel_name", C.c_char_p, []),
    ("pmx_last_error", C.c_char_p, []),
    ("pmx_debug_compile", C.c_int32, [_PD, _MD, C.POINTER(_abi.pmx_op_stream_view)]),
    ("pmx_debug_free", None, [C.POINTER(_abi.pmx_op_stream_view)]),
    ("pmx_debug_class_plan", C.c_int32, [_PD, _MD, C.POINTER(C.c_int64)]),
    ("pmx_debug_reload_env", None, []),
]


def lib():
    """Load libpmx_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make`).  pharmsol_amd has no CPU fallback.")
        # PyTorch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  If libpmx_hip.so were loaded
        # first, the process would end up with TWO HIP runtimes and torch would see no GPU; loading torch
        # first makes the dynamic loader hand torch's runtime to this library too, so device pointers and
        # streams are shared.  (Pure C/C++ clients link /opt/rocm's runtime and never meet torch.)
        try:
            import torch  # noqa: F401
        except ImportError:  # the C ABI itself does not need torch
            pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.pmx_abi_version() != _abi.PMX_ABI_VERSION:
            raise ImportError("libpmx_hip.so ABI version mismatch")
        if L.pmx_sizeof_model_desc() != C.sizeof(_abi.pmx_model_desc):
            raise ImportError("pmx_model_desc layout drift between include/pmx.h and pharmsol_amd/_abi.py")
        if L.pmx_sizeof_population_desc() != C.sizeof(_abi.pmx_population_desc):
            raise ImportError("pmx_population_desc layout drift")
        _lib = L
    return _lib


def check(rc: int, allow_pair_failures: bool = False) -> int:
    if rc == _abi.PMX_OK or (allow_pair_failures and rc == _abi.PMX_ERR_PAIR_FAILED):
        return rc
    raise _abi.PmxError(rc, lib().pmx_last_error().decode())
