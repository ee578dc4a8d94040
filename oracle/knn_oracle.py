"""ORACLE (test infrastructure): ctypes binding of oracle/knn_oracle.c plus a NumPy float64
cross-check.  See knn_oracle.c for the reference citations and the fixed summation/tie order."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libknn_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        _lib.knn_oracle_scores.argtypes = [fp, ctypes.c_int64, ctypes.c_int, fp, ctypes.c_int, fp]
        _lib.knn_oracle_topk_ip.argtypes = [fp, ctypes.c_int64, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, ip, fp]
        _lib.knn_oracle_select.argtypes = [fp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ip, fp]
        _lib.knn_oracle_threads.restype = ctypes.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def threads():
    return lib().knn_oracle_threads()


def scores(X, Q):
    X = np.ascontiguousarray(X, np.float32)
    Q = np.ascontiguousarray(Q, np.float32)
    out = np.empty((Q.shape[0], X.shape[0]), np.float32)
    lib().knn_oracle_scores(_fp(X), X.shape[0], X.shape[1], _fp(Q), Q.shape[0], _fp(out))
    return out


def topk_ip(X, Q, k):
    X = np.ascontiguousarray(X, np.float32)
    Q = np.ascontiguousarray(Q, np.float32)
    idx = np.empty((Q.shape[0], k), np.int64)
    val = np.empty((Q.shape[0], k), np.float32)
    lib().knn_oracle_topk_ip(_fp(X), X.shape[0], X.shape[1], _fp(Q), Q.shape[0], k,
                             idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _fp(val))
    return idx, val


def select(sc, k):
    sc = np.ascontiguousarray(sc, np.float32)
    idx = np.empty((sc.shape[0], k), np.int64)
    val = np.empty((sc.shape[0], k), np.float32)
    lib().knn_oracle_select(_fp(sc), sc.shape[1], sc.shape[0], k, idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _fp(val))
    return idx, val


def topk_numpy_f64(X, Q, k):
    """Independent cross-check: float64 scores, stable (score desc, idx asc) order."""
    s = np.asarray(Q, np.float64) @ np.asarray(X, np.float64).T
    order = np.lexsort((np.broadcast_to(np.arange(s.shape[1]), s.shape), -s), axis=1)[:, :k]
    return order, np.take_along_axis(s, order, 1)
