"""ctypes binding of libdfusion_hip.so (the C ABI declared in include/dfusion.h).

This is the only way the Python host layer reaches the GPU for the hot path.  There is no CPU
fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdfusion_hip.so")
_lib = None

_vp, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

# symbol -> (restype, argtypes); must list every function include/dfusion.h declares
SIGNATURES = {
    "df_last_error": (ctypes.c_char_p, []),
    "df_version": (_i, []),
    "df_knn_device": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "df_knn": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` at the repo root (needs hipcc). "
                "densefusion_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().df_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libdfusion_hip {what} failed ({status}): {msg}")


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def dptr(t) -> int:
    """Device pointer of a contiguous CUDA(HIP) tensor."""
    if not t.is_cuda:
        raise RuntimeError("densefusion_amd ops need device tensors (no CPU path); got a CPU tensor")
    if not t.is_contiguous():
        raise RuntimeError("densefusion_amd ops need contiguous tensors")
    return t.data_ptr()
