"""ctypes binding of liblnx_hip.so (the C ABI declared in include/lnx.h).

The library is the product: there is no Python/CPU fallback.  If it is missing or does not
load, importing anything that needs a kernel raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblnx_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU, ACT_GELU_BWD, ACT_RELU_BWD = 0, 1, 2, 3, 4
ADDR_PLAIN, ADDR_PATCH2 = 0, 1


class LnxError(RuntimeError):
    pass


class RowMap(C.Structure):
    _fields_ = [("group", C.c_int), ("pad", C.c_int), ("off", C.c_int)]


class GemmArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("W", C.c_void_p), ("ldw", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("out_f32", C.c_int),
        ("a_mode", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Cin", C.c_int),
        ("c_mode", C.c_int), ("c_map", RowMap),
        ("bias", C.c_void_p),
        ("c2", C.c_void_p), ("ldc2", C.c_int64),
        ("act", C.c_int),
        ("aux", C.c_void_p), ("ldaux", C.c_int64),
        ("gamma", C.c_void_p), ("rowscale", C.c_void_p), ("rows_per_sample", C.c_int),
        ("res", C.c_void_p), ("ldres", C.c_int64),
    ]


class WgradArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("dY", C.c_void_p), ("lddy", C.c_int64),
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("a_mode", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Cin", C.c_int),
        ("dW", C.c_void_p), ("lddw", C.c_int64),
        ("k_perm_c", C.c_int),
        ("db", C.c_void_p),
        ("splits", C.c_int),
    ]


_lib = None


def lib() -> C.CDLL:
    """Load the HIP library or fail loudly (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LnxError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C linnaeus_amd/csrc`). linnaeus_amd has no CPU fallback."
            )
        _lib = C.CDLL(LIB_PATH)
        _lib.lnx_last_error.restype = C.c_char_p
        for name in EXPORTS:
            if not hasattr(_lib, name):
                raise LnxError(f"{LIB_PATH} does not export {name}; rebuild it")
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise LnxError(f"{what} failed (rc={rc}): {lib().lnx_last_error().decode()}")


# every symbol include/lnx.h declares (kept in sync by tests/test_abi.py)
EXPORTS = [
    "lnx_last_error", "lnx_version", "lnx_device_cus",
    "lnx_gemm_nt", "lnx_gemm_tn",
]
