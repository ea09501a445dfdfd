"""Oracle: loader for the plain-C k-NN restatement (oracle/knn_ref.c).  TEST INFRASTRUCTURE."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_knn.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "knn_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_knn.restype = ctypes.c_int
        _lib.oracle_knn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6
    return _lib


def knn_ref(ref: np.ndarray, query: np.ndarray, k: int = 1, threads: int = 0) -> np.ndarray:
    """KNearestNeighbor(k)(ref[B,D,R], query[B,D,Q]) -> int64 [B,k,Q], 1-based
    (lib/knn/__init__.py:15-23 call convention)."""
    ref = np.ascontiguousarray(ref, dtype=np.float32)
    query = np.ascontiguousarray(query, dtype=np.float32)
    assert ref.ndim == 3 and query.ndim == 3 and ref.shape[:2] == query.shape[:2]
    B, D, R = ref.shape
    Q = query.shape[2]
    idx = np.empty((B, k, Q), dtype=np.int64)
    rc = _load().oracle_knn(ref.ctypes.data, query.ctypes.data, idx.ctypes.data, B, D, R, Q, k, threads)
    if rc != 0:
        raise RuntimeError("oracle_knn: bad arguments")
    return idx
