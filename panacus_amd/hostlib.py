"""ctypes binding of libpanacus_host.so -- the C++ host layer above the device ABI
(closed-form growth of src/graph_broker/hist.rs:51-187, GFA front end, table writers)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .thresholds import Threshold

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpanacus_host.so")
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m panacus_amd._build`")
    L = C.CDLL(LIB_PATH)
    u64p, f64p = C.POINTER(C.c_uint64), C.POINTER(C.c_double)
    L.pnh_calc_growth.restype = C.c_int64
    L.pnh_calc_growth.argtypes = [u64p, C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_double, C.c_uint, f64p]
    L.pnh_calc_growth_branch.restype = C.c_int64
    L.pnh_calc_growth_branch.argtypes = [C.c_int, u64p, C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_double,
                                         C.c_uint, f64p]
    L.pnh_choose_log2.restype = C.c_double
    L.pnh_choose_log2.argtypes = [C.c_uint64, C.c_uint64]
    _lib = L
    return L


def calc_growth(hist, coverage: Threshold, quorum: Threshold, n_threads: int = 0) -> np.ndarray:
    """Hist::calc_growth (hist.rs:51-66): n = len(hist)-1 f64 values (no leading NaN)."""
    h = np.ascontiguousarray(hist, dtype=np.uint64)
    out = np.zeros(max(len(h) - 1, 1), dtype=np.float64)
    n = load().pnh_calc_growth(h.ctypes.data_as(C.POINTER(C.c_uint64)), len(h), coverage.kind, float(coverage.value),
                               quorum.kind, float(quorum.value), n_threads, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:n]


def calc_growth_branch(branch: str, hist, coverage: Threshold, quorum: Threshold, n_threads: int = 0) -> np.ndarray:
    h = np.ascontiguousarray(hist, dtype=np.uint64)
    out = np.zeros(max(len(h) - 1, 1), dtype=np.float64)
    b = {"union": 0, "core": 1, "quorum": 2}[branch]
    n = load().pnh_calc_growth_branch(b, h.ctypes.data_as(C.POINTER(C.c_uint64)), len(h), coverage.kind,
                                      float(coverage.value), quorum.kind, float(quorum.value), n_threads,
                                      out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:n]


def calc_all_growths(hist, thresholds, n_threads: int = 0):
    """Hist::calc_all_growths (hist.rs:68-87): one curve per (coverage, quorum) pair, NaN row 0."""
    out = []
    for c, q in zip(thresholds.coverage, thresholds.quorum):
        g = calc_growth(hist, c, q, n_threads)
        out.append(np.concatenate([[np.nan], g]))
    return out
