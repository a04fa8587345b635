"""ctypes access to oracle/c/libcsoracle.so (test infrastructure only)."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c", "libcsoracle.so")


class cs_hist(C.Structure):
    _fields_ = [("T", C.c_int64), ("B", C.c_int64), ("K", C.c_int64), ("tail", C.c_int64), ("full", C.c_int32)]


def load():
    if not os.path.exists(_PATH):
        raise RuntimeError(f"{_PATH} missing: run `make -C oracle/c` (or __graft_entry__.build())")
    L = C.CDLL(_PATH)
    i64, u32, u64, i32 = C.c_int64, C.c_uint32, C.c_uint64, C.c_int32
    P = C.POINTER
    L.history_init.argtypes = [P(cs_hist), i64, i64, i64, C.c_void_p]
    L.history_len.argtypes = [P(cs_hist)]; L.history_len.restype = i64
    L.history_update_available.argtypes = [P(cs_hist), C.c_void_p, C.c_void_p]
    L.history_enqueue.argtypes = [P(cs_hist), C.c_void_p]; L.history_enqueue.restype = i64
    L.history_slot.argtypes = [P(cs_hist), i64]; L.history_slot.restype = i64
    L.multinomial_left.argtypes = [C.c_void_p, i64, C.c_void_p, i64, C.c_void_p, C.c_void_p]
    L.systematic_index.argtypes = [C.c_void_p, i64, C.c_float, i64, C.c_void_p]
    L.cs_uniform_rows.argtypes = [u64, u32, u32, i64, i64, i32, C.c_void_p]
    L.cs_uniform64_rows.argtypes = [u64, u32, u32, i64, i64, i32, C.c_void_p]
    L.cs_hist_draw.argtypes = [u64, u32, u32, i64, i64]; L.cs_hist_draw.restype = i64
    return L


def multinomial_left(L, probs, u):
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    u = np.ascontiguousarray(u, dtype=np.float64)
    idx = np.empty(u.shape, dtype=np.int64)
    cum = np.empty(probs.shape[-1], dtype=np.float32)
    for r in range(probs.shape[0]):
        L.multinomial_left(probs[r].ctypes.data, probs.shape[1], u[r].ctypes.data, u.shape[1], idx[r].ctypes.data,
                           cum.ctypes.data)
    return idx


def uniform64_rows(L, seed, stream, it, row0, rows, cols):
    out = np.empty((rows, cols), dtype=np.float64)
    L.cs_uniform64_rows(seed, stream, it, row0, rows, cols, out.ctypes.data)
    return out


def uniform_rows(L, seed, stream, it, row0, rows, cols):
    out = np.empty((rows, cols), dtype=np.float32)
    L.cs_uniform_rows(seed, stream, it, row0, rows, cols, out.ctypes.data)
    return out
