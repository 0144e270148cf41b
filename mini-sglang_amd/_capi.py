"""ctypes binding of libmsgl_hip.so (the C ABI declared in include/msgl_hip.h).

There is no fallback: if the shared library is missing or a symbol cannot be resolved the
import of this module raises, and every call that returns a non-zero status raises
``RuntimeError(msgl_last_error())``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSGL_HIP_LIB", os.path.join(_HERE, "libmsgl_hip.so"))

ABI_VERSION = 17

# name -> (restype, argtypes); mirrors include/msgl_hip.h and include/msgl_comm.h one to one
SIGNATURES = {
    "msgl_abi_version": (c_int, []),
    "msgl_last_error": (c_char_p, []),
    "msgl_set_variant": (c_int, [c_int, c_int]),
    "msgl_get_variant": (c_int, [c_int]),
    "msgl_store_kv": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p],
    ),
    "msgl_paged_decode_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "msgl_paged_decode": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
         c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
         c_void_p, c_void_p, c_void_p],
    ),
    "msgl_paged_decode_static": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
         c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    ),
    "msgl_decode_plan": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "msgl_shared_prefix_decode": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p,
         c_void_p, c_int, c_void_p, c_void_p],
    ),
    "msgl_paged_extend": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
         c_int64, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_size_t, c_int64, c_void_p],
    ),
    "msgl_paged_extend_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "msgl_paged_extend_key_splits": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "msgl_paged_extend_workspace_bytes_for": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "msgl_extend_plan_items": (c_int64, [c_int, c_int]),
    "msgl_extend_plan": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "msgl_rmsnorm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "msgl_fused_add_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "msgl_silu_and_mul": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p]),
    "msgl_rope_neox": (
        c_int,
        [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    ),
    "msgl_qknorm_rope_store": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
         c_int, c_int, c_int, c_int, c_int, c_void_p],
    ),
    "msgl_argmax": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_void_p]),
    "msgl_argmax_val": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "msgl_lm_head_argmax_workspace_bytes": (c_size_t, [c_int, c_int]),
    "msgl_lm_head_argmax": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "msgl_prepare_decode": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, c_int, c_int, c_void_p],
    ),
    "msgl_prepare_decode_paged": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p,
         c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    ),
    "msgl_stream_read": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "msgl_stream_read_cfg": (c_int, [c_void_p, c_size_t, c_void_p, c_int, c_int, c_void_p]),
    "msgl_stream_read_span": (c_int, [c_void_p, c_size_t, c_void_p, c_int, c_int, c_int, c_void_p]),
    # include/msgl_comm.h
    "msgl_ar_create": (c_int, [c_int, c_int, c_size_t, ctypes.POINTER(c_void_p)]),
    "msgl_ar_export": (c_int, [c_void_p, c_void_p]),
    "msgl_ar_open": (c_int, [c_void_p, c_void_p]),
    "msgl_ar_allreduce_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_size_t, c_void_p]),
    "msgl_ar_num_steps": (c_int, [c_void_p, c_int64, c_size_t]),
    "msgl_ar_allreduce_bf16_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_size_t, c_int, c_void_p]),
    "msgl_ar_set_timeout_ms": (c_int, [c_void_p, c_int]),
    "msgl_ar_check": (c_int, [c_void_p]),
    "msgl_ar_destroy": (c_int, [c_void_p]),
}


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or __graft_entry__.build()); there is no non-HIP fallback"
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.msgl_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {got}, binding expects {ABI_VERSION}")
    return lib


lib = _load()


def last_error() -> str:
    return (lib.msgl_last_error() or b"").decode("utf-8", "replace")


def check(status: int) -> None:
    if status != 0:
        raise RuntimeError(f"msgl status {status}: {last_error()}")
