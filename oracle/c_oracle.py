"""ctypes binding of oracle/oracle_c.c (TEST INFRASTRUCTURE ONLY; see that file's header)."""
import ctypes
import os

import numpy as np

_here = os.path.dirname(os.path.abspath(__file__))
_path = os.path.join(_here, "_build", "liboracle_c.so")


def load():
    if not os.path.exists(_path):
        raise ImportError(f"{_path} missing: run `make -C oracle` (or __graft_entry__.build())")
    lib = ctypes.CDLL(_path)
    lib.orc_num_threads.restype = ctypes.c_int
    lib.orc_gather_concat_f32.restype = ctypes.c_int64
    return lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _table_ptrs(tables):
    arr = (ctypes.POINTER(ctypes.c_float) * len(tables))()
    for i, t in enumerate(tables):
        assert t.dtype == np.float32 and t.flags.c_contiguous
        arr[i] = _fp(t)
    return arr


def num_threads():
    return load().orc_num_threads()


def gather_concat(tables, ids):
    lib = load()
    ids = np.ascontiguousarray(ids, np.int32)
    B, F = ids.shape
    vocab = np.array([t.shape[0] for t in tables], np.int64)
    dim = np.array([t.shape[1] for t in tables], np.int32)
    out = np.empty((B, int(dim.sum())), np.float32)
    lib.orc_gather_concat_f32(_table_ptrs(tables), vocab.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                              dim.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.c_int32(F),
                              ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.c_int64(B), _fp(out))
    return out


def pairwise_dot(x):
    lib = load()
    x = np.ascontiguousarray(x, np.float32)
    B, n, D = x.shape
    P = n * (n - 1) // 2
    out = np.empty((B, P), np.float32)
    lib.orc_pairwise_dot_f32(_fp(x), ctypes.c_int64(B), ctypes.c_int32(n), ctypes.c_int32(D), _fp(out),
                             ctypes.c_int64(P))
    return out


def dlrm_gather_dot(tables, ids, dense, scratch=None, out=None):
    lib = load()
    ids = np.ascontiguousarray(ids, np.int32)
    dense = np.ascontiguousarray(dense, np.float32)
    B, F = ids.shape
    D = tables[0].shape[1]
    n = F + 1
    P = n * (n - 1) // 2
    vocab = np.array([t.shape[0] for t in tables], np.int64)
    if scratch is None:
        scratch = np.empty((B, n * D), np.float32)
    if out is None:
        out = np.empty((B, P + D), np.float32)
    lib.orc_dlrm_gather_dot_f32(_table_ptrs(tables), vocab.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                ctypes.c_int32(F), ctypes.c_int32(D),
                                ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _fp(dense),
                                ctypes.c_int64(B), _fp(scratch), _fp(out))
    return out


def fm_layer(first, w, second):
    lib = load()
    first = np.ascontiguousarray(first, np.float32)
    second = np.ascontiguousarray(second, np.float32)
    w = np.ascontiguousarray(w, np.float32).reshape(-1)
    B, L1 = first.shape
    out = np.empty((B, 1), np.float32)
    lib.orc_fm_layer_f32(_fp(first), ctypes.c_int32(L1), _fp(w), _fp(second), ctypes.c_int32(second.shape[1]),
                         ctypes.c_int64(B), _fp(out))
    return out


def cross(x, W, Bv):
    lib = load()
    x = np.ascontiguousarray(x, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    Bv = np.ascontiguousarray(Bv, np.float32)
    B, dim = x.shape
    out = np.empty((B, dim), np.float32)
    lib.orc_cross_f32(_fp(x), ctypes.c_int32(dim), _fp(W), _fp(Bv), ctypes.c_int32(W.shape[0]), ctypes.c_int64(B),
                      _fp(out))
    return out
