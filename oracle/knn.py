"""Oracle wrapper for oracle/knn_ref.c (ctypes).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libknn_ref.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.knn_ref.restype = ctypes.c_int
        _lib.knn_ref.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 7
    return _lib


def knn_graph(x, y=None, k=9, dilation=1, relative_pos=None, normalize=True, return_dist=False):
    """x (B,C,N[,1]) float32 array/tensor -> edge_index int64 (2,B,N,k) [, dist (B,N,M)]."""
    xa = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(x.shape[0], x.shape[1], -1))
    B, C, N = xa.shape
    ya = None
    M = N
    if y is not None:
        ya = np.ascontiguousarray(np.asarray(y, dtype=np.float32).reshape(y.shape[0], y.shape[1], -1))
        M = ya.shape[2]
    rp = None
    if relative_pos is not None:
        rp = np.ascontiguousarray(np.asarray(relative_pos, dtype=np.float32).reshape(N, M))
    K = k * dilation
    edge = np.zeros((2, B, N, (K + dilation - 1) // dilation), dtype=np.int64)
    dist = np.zeros((B, N, M), dtype=np.float32) if return_dist else None
    ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    rc = _load().knn_ref(ptr(xa), ptr(ya), ptr(rp), ptr(edge), ptr(dist), B, C, N, M, K, dilation, int(normalize))
    if rc != 0:
        raise RuntimeError("knn_ref failed")
    return (edge, dist) if return_dist else edge
