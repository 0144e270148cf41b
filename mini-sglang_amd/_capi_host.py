"""ctypes binding of libmsgl_host.so (the host-side C ABI declared in include/msgl_host.h).  No fallback: a
missing library or symbol raises at import."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSGL_HOST_LIB", os.path.join(_HERE, "libmsgl_host.so"))

_I32P, _I64P = POINTER(c_int32), POINTER(c_int64)

# name -> (restype, argtypes); mirrors include/msgl_host.h one to one
SIGNATURES = {
    "msgl_radix_create": (c_void_p, [c_int]),
    "msgl_radix_destroy": (None, [c_void_p]),
    "msgl_radix_reset": (None, [c_void_p]),
    "msgl_radix_match_prefix": (c_int64, [c_void_p, _I32P, c_int64, _I32P, c_int64, _I64P]),
    "msgl_radix_insert": (c_int64, [c_void_p, _I32P, _I32P, c_int64]),
    "msgl_radix_inc_lock_ref": (c_int, [c_void_p, c_int64]),
    "msgl_radix_dec_lock_ref": (c_int, [c_void_p, c_int64]),
    "msgl_radix_evict": (c_int64, [c_void_p, c_int64, _I32P, c_int64, _I64P, c_int64, _I64P]),
    "msgl_radix_evictable_size": (c_int64, [c_void_p]),
    "msgl_radix_protected_size": (c_int64, [c_void_p]),
    "msgl_radix_node_count": (c_int64, [c_void_p]),
    "msgl_pages_create": (c_void_p, [c_int64, c_int]),
    "msgl_pages_destroy": (None, [c_void_p]),
    "msgl_pages_clear": (None, [c_void_p]),
    "msgl_pages_free_count": (c_int64, [c_void_p]),
    "msgl_pages_alloc": (c_int64, [c_void_p, c_int64, _I32P]),
    "msgl_pages_free": (c_int64, [c_void_p, _I32P, c_int64]),
    "msgl_pages_peek": (c_int64, [c_void_p, _I32P, c_int64]),
    "msgl_pages_assign": (c_int64, [c_void_p, _I32P, c_int64]),
}


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}`; there is no pure-Python fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = _load()
