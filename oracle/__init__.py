"""CPU ORACLE -- test infrastructure, NOT product code.

ctypes front end of ``liboracle.so`` (oracle/panacus_oracle.c), the plain-C restatement of
the reference's coverage-histogram / growth path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker / the timed CPU baseline.  ``panacus_amd`` never does.

Parity status: pinned against the reference's own known-answer vectors
(tests/test_oracle_golden.py); see oracle/panacus_oracle.h for the list.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

NODE, BP, EDGE = 0, 1, 2
GROUP_PATHID, GROUP_SAMPLE, GROUP_HAPLOTYPE, GROUP_FILE = 0, 1, 2, 3
ABSOLUTE, RELATIVE = 0, 1


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "panacus_oracle.c")
    hdr = os.path.join(_HERE, "panacus_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u64p = C.POINTER(C.c_uint64)
        u32p = C.POINTER(C.c_uint32)
        u8p = C.POINTER(C.c_uint8)
        f64p = C.POINTER(C.c_double)
        L.orc_graph_from_gfa.restype = C.c_void_p
        L.orc_graph_from_gfa.argtypes = [C.c_char_p, C.c_int]
        L.orc_graph_free.argtypes = [C.c_void_p]
        for name in ("orc_graph_n_nodes", "orc_graph_n_edges", "orc_graph_n_paths"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_graph_node_lens.restype = u32p
        L.orc_graph_node_lens.argtypes = [C.c_void_p]
        L.orc_graph_path_display.restype = C.c_char_p
        L.orc_graph_path_display.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_graph_last_error.restype = C.c_char_p
        L.orc_graph_path_order.restype = C.c_int64
        L.orc_graph_path_order.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, u64p, u64p, u64p]
        L.orc_graph_path_order_masked.restype = C.c_int64
        L.orc_graph_path_order_masked.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                                  u64p, u64p, u64p]
        L.orc_graph_exclude_flags.restype = C.c_int
        L.orc_graph_exclude_flags.argtypes = [C.c_void_p, C.c_int, C.c_char_p, u8p]
        L.orc_graph_masked_table.restype = C.c_int64
        L.orc_graph_masked_table.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.POINTER(C.c_uint64)),
                                             C.POINTER(C.c_uint64), u8p, C.POINTER(C.POINTER(C.c_uint64)),
                                             C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
        L.orc_hist_apply_uncovered.restype = None
        L.orc_hist_apply_uncovered.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                               C.c_uint64, C.POINTER(C.c_uint64)]
        L.orc_graph_group_name.restype = C.c_char_p
        L.orc_graph_group_name.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_graph_item_table.restype = C.c_int64
        L.orc_graph_item_table.argtypes = [C.c_void_p, C.c_int, C.POINTER(u64p), u64p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_coverage.argtypes = [u64p, u64p, u64p, u64p, C.c_uint64, C.c_uint64, u8p, u32p]
        L.orc_hist.argtypes = [u32p, C.c_uint64, C.c_uint64, u32p, u64p]
        L.orc_choose.restype = C.c_double
        L.orc_choose.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_growth.restype = C.c_int64
        L.orc_growth.argtypes = [u64p, C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_double, f64p]
        L.orc_by_group.restype = C.c_int64
        L.orc_by_group.argtypes = [u64p, u64p, u64p, u64p, C.c_uint64, C.c_uint64, u8p, u64p, C.POINTER(u64p)]
        L.orc_ordered_growth.argtypes = [u64p, u64p, C.c_uint64, C.c_uint64, C.c_int, C.c_double,
                                         C.c_int, C.c_double, u32p, f64p]
        L.orc_exp2.argtypes = [f64p, f64p, C.c_uint64]
        L.orc_log2.argtypes = [f64p, f64p, C.c_uint64]
        L.orc_similarity.restype = C.c_int
        L.orc_similarity.argtypes = [u64p, u64p, C.c_uint64, C.c_uint64, u32p, u64p, u64p,
                                     C.POINTER(C.c_float)]
        L.orc_table_row.argtypes = [u64p, u64p, C.c_uint64, C.c_uint64, C.c_uint64, u64p]
        u32p_ = C.POINTER(C.c_uint32)
        L.orc_by_group_values.restype = C.c_int64
        L.orc_by_group_values.argtypes = [u64p, u64p, u64p, u64p, C.c_uint64, C.c_uint64, u8p, u64p, C.POINTER(u64p),
                                          C.POINTER(u32p_)]
        L.orc_table_row_values.restype = C.c_int
        L.orc_table_row_values.argtypes = [u64p, u64p, u32p_, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, u64p]
        for name in ("orc_growth_union", "orc_growth_core"):
            getattr(L, name).argtypes = [u64p, C.c_uint64, C.c_int, C.c_double, f64p]
        L.orc_growth_quorum.argtypes = [u64p, C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_double, f64p]
        L.pansyn_node_lens.argtypes = [C.c_uint64, C.c_uint64, u32p]
        L.pansyn_splitmix64.restype = C.c_uint64
        L.pansyn_splitmix64.argtypes = [C.c_uint64]
        L.pansyn_node_len.restype = C.c_uint32
        L.pansyn_node_len.argtypes = [C.c_uint64, C.c_uint64]
        L.pansyn_node_thr.restype = C.c_uint64
        L.pansyn_node_thr.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.pansyn_generate.restype = C.c_int64
        L.pansyn_generate.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(u64p), u64p]
        L.pansyn_rearrange.restype = None
        L.pansyn_rearrange.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, u64p, u64p]
        _lib = L
    return _lib


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class Graph:
    """GraphStorage + GraphMask of the reference, restated (graph.rs:195-375, abacus.rs:242-347)."""

    def __init__(self, gfa_file: str, index_edges: bool = False):
        self._h = lib().orc_graph_from_gfa(os.fsencode(gfa_file), int(index_edges))
        if not self._h:
            raise ValueError(lib().orc_graph_last_error().decode())
        self.gfa_file = gfa_file
        self.has_edges = index_edges

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_graph_free(self._h)
            self._h = None

    @property
    def n_nodes(self):
        return int(lib().orc_graph_n_nodes(self._h))

    @property
    def n_edges(self):
        return int(lib().orc_graph_n_edges(self._h))

    @property
    def n_paths(self):
        return int(lib().orc_graph_n_paths(self._h))

    @property
    def node_lens(self) -> np.ndarray:
        p = lib().orc_graph_node_lens(self._h)
        return np.ctypeslib.as_array(p, shape=(self.n_nodes + 1,)).copy()

    def path_names(self):
        return [lib().orc_graph_path_display(self._h, i).decode() for i in range(self.n_paths)]

    def n_items(self, count_type: int) -> int:
        return self.n_edges if count_type == EDGE else self.n_nodes

    def path_order(self, group_mode=GROUP_PATHID, group_file=None, order_file=None, subset_file=None,
                   exclude_file=None):
        """-> (path_idx[u64], group_id[u64], group_names)"""
        P = self.n_paths
        pi = np.zeros(max(P, 1), dtype=np.uint64)
        gi = np.zeros(max(P, 1), dtype=np.uint64)
        n_out = C.c_uint64(0)
        ng = lib().orc_graph_path_order_masked(
            self._h, group_mode,
            os.fsencode(group_file) if group_file else None,
            os.fsencode(order_file) if order_file else None,
            os.fsencode(subset_file) if subset_file else None,
            os.fsencode(exclude_file) if exclude_file else None,
            _p(pi, C.c_uint64), _p(gi, C.c_uint64), C.byref(n_out))
        if ng < 0:
            raise ValueError(lib().orc_graph_last_error().decode())
        names = [lib().orc_graph_group_name(self._h, g).decode() for g in range(ng)]
        return pi[: n_out.value].copy(), gi[: n_out.value].copy(), names

    def exclude_flags(self, count_type: int, exclude_file: str) -> np.ndarray:
        """ActiveTable.items of a whole-path exclude list (call path_order first)."""
        flags = np.zeros(self.n_items(count_type) + 1, dtype=np.uint8)
        if lib().orc_graph_exclude_flags(self._h, count_type, os.fsencode(exclude_file), _p(flags, C.c_uint8)) != 0:
            raise ValueError(lib().orc_graph_last_error().decode())
        return flags

    def masked_table(self, count_type: int, subset_file=None, exclude_file=None):
        """ItemTable, exclude flags and uncovered bps under -s / -e BED lists (call path_order first).
        -> (items[u64], prefsum[u64, P+1], exclude[u8, n_items+1], uncov_ids[u64], uncov_bps[u64])"""
        pre = np.zeros(self.n_paths + 1, dtype=np.uint64)
        flags = np.zeros(self.n_items(count_type) + 1, dtype=np.uint8)
        ptr, uid, ubp = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        nu = C.c_uint64(0)
        n = lib().orc_graph_masked_table(self._h, count_type,
                                         os.fsencode(subset_file) if subset_file else None,
                                         os.fsencode(exclude_file) if exclude_file else None,
                                         C.byref(ptr), _p(pre, C.c_uint64), _p(flags, C.c_uint8),
                                         C.byref(uid), C.byref(ubp), C.byref(nu))
        if n < 0:
            raise ValueError(lib().orc_graph_last_error().decode())
        items = np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].copy()
        lib().orc_free(ptr)
        k = nu.value
        ids = np.ctypeslib.as_array(uid, shape=(max(k, 1),))[:k].copy() if uid else np.zeros(0, np.uint64)
        bps = np.ctypeslib.as_array(ubp, shape=(max(k, 1),))[:k].copy() if ubp else np.zeros(0, np.uint64)
        if uid:
            lib().orc_free(uid)
        if ubp:
            lib().orc_free(ubp)
        return items, pre, flags, ids, bps

    def item_table(self, count_type: int):
        """-> (items[u64], prefsum[u64, P+1])  (the reference's ItemTable, util.rs:81-93)"""
        pre = np.zeros(self.n_paths + 1, dtype=np.uint64)
        ptr = C.POINTER(C.c_uint64)()
        n = lib().orc_graph_item_table(self._h, count_type, C.byref(ptr), _p(pre, C.c_uint64))
        if n < 0:
            raise ValueError(lib().orc_graph_last_error().decode())
        items = np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].copy()
        lib().orc_free(ptr)
        return items, pre


def coverage(items, prefsum, path_idx, group_id, n_items, exclude=None) -> np.ndarray:
    items, prefsum, path_idx, group_id = map(_u64, (items, prefsum, path_idx, group_id))
    out = np.zeros(n_items + 1, dtype=np.uint32)
    ex = None if exclude is None else np.ascontiguousarray(exclude, dtype=np.uint8)
    lib().orc_coverage(_p(items, C.c_uint64), _p(prefsum, C.c_uint64), _p(path_idx, C.c_uint64),
                       _p(group_id, C.c_uint64), len(path_idx), n_items, _p(ex, C.c_uint8),
                       _p(out, C.c_uint32))
    return out


def hist_apply_uncovered(countable, uncov_ids, uncov_bps, hist_in) -> np.ndarray:
    """construct_hist_bps' fix-up for partially covered nodes (abacus.rs:779-785)."""
    countable = np.ascontiguousarray(countable, dtype=np.uint32)
    ids, bps = _u64(uncov_ids), _u64(uncov_bps)
    out = _u64(hist_in).copy()
    lib().orc_hist_apply_uncovered(_p(countable, C.c_uint32), _p(ids, C.c_uint64), _p(bps, C.c_uint64), len(ids),
                                   _p(out, C.c_uint64))
    return out


def hist(countable, n_groups, weights=None) -> np.ndarray:
    countable = np.ascontiguousarray(countable, dtype=np.uint32)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.uint32)
    out = np.zeros(n_groups + 1, dtype=np.uint64)
    lib().orc_hist(_p(countable, C.c_uint32), len(countable) - 1, n_groups, _p(w, C.c_uint32),
                   _p(out, C.c_uint64))
    return out


def choose(n, k) -> float:
    return float(lib().orc_choose(n, k))


def growth(hist_arr, coverage_thr=(ABSOLUTE, 1), quorum_thr=(RELATIVE, 0.0)) -> np.ndarray:
    """Hist::calc_growth (hist.rs:51-66); returns n values (no leading NaN)."""
    h = _u64(hist_arr)
    out = np.zeros(max(len(h) - 1, 1), dtype=np.float64)
    n = lib().orc_growth(_p(h, C.c_uint64), len(h), coverage_thr[0], float(coverage_thr[1]),
                         quorum_thr[0], float(quorum_thr[1]), _p(out, C.c_double))
    return out[:n]


def growth_branch(branch: str, hist_arr, coverage_thr=(ABSOLUTE, 1), quorum_thr=(RELATIVE, 0.0)) -> np.ndarray:
    """calc_growth_union/_core/_quorum called directly (hist.rs:89-187), bypassing the dispatch."""
    h = _u64(hist_arr)
    n = len(h) - 1
    out = np.zeros(max(n, 1), dtype=np.float64)
    if branch == "union":
        lib().orc_growth_union(_p(h, C.c_uint64), n, coverage_thr[0], float(coverage_thr[1]), _p(out, C.c_double))
    elif branch == "core":
        lib().orc_growth_core(_p(h, C.c_uint64), n, coverage_thr[0], float(coverage_thr[1]), _p(out, C.c_double))
    elif branch == "quorum":
        lib().orc_growth_quorum(_p(h, C.c_uint64), n, coverage_thr[0], float(coverage_thr[1]),
                                quorum_thr[0], float(quorum_thr[1]), _p(out, C.c_double))
    else:
        raise ValueError(branch)
    return out[:n]


def by_group(items, prefsum, path_idx, group_id, n_items, exclude=None):
    """AbacusByGroup r, c (abacus.rs:859-986)."""
    items, prefsum, path_idx, group_id = map(_u64, (items, prefsum, path_idx, group_id))
    r = np.zeros(n_items + 2, dtype=np.uint64)
    ex = None if exclude is None else np.ascontiguousarray(exclude, dtype=np.uint8)
    ptr = C.POINTER(C.c_uint64)()
    nnz = lib().orc_by_group(_p(items, C.c_uint64), _p(prefsum, C.c_uint64), _p(path_idx, C.c_uint64),
                             _p(group_id, C.c_uint64), len(path_idx), n_items, _p(ex, C.c_uint8),
                             _p(r, C.c_uint64), C.byref(ptr))
    c = np.ctypeslib.as_array(ptr, shape=(max(nnz, 1),))[:nnz].copy()
    lib().orc_free(ptr)
    return r, c


def by_group_values(items, prefsum, path_idx, group_id, n_items, exclude=None):
    """AbacusByGroup r, c, v with report_values (abacus.rs:859-986)."""
    items, prefsum, path_idx, group_id = map(_u64, (items, prefsum, path_idx, group_id))
    r = np.zeros(n_items + 2, dtype=np.uint64)
    ex = None if exclude is None else np.ascontiguousarray(exclude, dtype=np.uint8)
    ptr, vptr = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)()
    nnz = lib().orc_by_group_values(_p(items, C.c_uint64), _p(prefsum, C.c_uint64), _p(path_idx, C.c_uint64),
                                    _p(group_id, C.c_uint64), len(path_idx), n_items, _p(ex, C.c_uint8),
                                    _p(r, C.c_uint64), C.byref(ptr), C.byref(vptr))
    c = np.ctypeslib.as_array(ptr, shape=(max(nnz, 1),))[:nnz].copy()
    v = np.ctypeslib.as_array(vptr, shape=(max(nnz, 1),))[:nnz].copy()
    lib().orc_free(ptr)
    lib().orc_free(vptr)
    return r, c, v


def ordered_growth(r, c, n_groups, coverage_thr=(ABSOLUTE, 1), quorum_thr=(RELATIVE, 0.0),
                   weights=None) -> np.ndarray:
    """AbacusByGroup::calc_growth (abacus.rs:989-1032)."""
    r, c = _u64(r), _u64(c)
    if len(c) == 0:
        c = np.zeros(1, dtype=np.uint64)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.uint32)
    out = np.zeros(max(n_groups, 1), dtype=np.float64)
    lib().orc_ordered_growth(_p(r, C.c_uint64), _p(c, C.c_uint64), len(r) - 2, n_groups,
                             coverage_thr[0], float(coverage_thr[1]), quorum_thr[0],
                             float(quorum_thr[1]), _p(w, C.c_uint32), _p(out, C.c_double))
    return out[:n_groups]


def exp2(x) -> np.ndarray:
    """libm exp2 (numpy's own exp2 is a different implementation and differs in the last bit)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros_like(x)
    lib().orc_exp2(_p(x, C.c_double), _p(y, C.c_double), x.size)
    return y


def log2(x) -> np.ndarray:
    """libm log2 (numpy's own log2 is a different implementation)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros_like(x)
    lib().orc_log2(_p(x, C.c_double), _p(y, C.c_double), x.size)
    return y


def similarity(r, c, n_groups, node_lens=None):
    """Similarity::set_table before clustering (similarity.rs:119-165):
    (inter[G, G] u64, path_lens[G] u64, jaccard[G, G] f32); raises where the reference panics."""
    r, c = _u64(r), _u64(c)
    if len(c) == 0:
        c = np.zeros(1, dtype=np.uint64)
    w = None if node_lens is None else np.ascontiguousarray(node_lens, dtype=np.uint32)
    G = int(n_groups)
    inter = np.zeros((G, G), dtype=np.uint64)
    lens = np.zeros(G, dtype=np.uint64)
    table = np.zeros((G, G), dtype=np.float32)
    rc = lib().orc_similarity(_p(r, C.c_uint64), _p(c, C.c_uint64), len(r) - 2, G, _p(w, C.c_uint32),
                              _p(inter, C.c_uint64), _p(lens, C.c_uint64), _p(table, C.c_float))
    if rc:
        raise KeyError("a group holds no item: the reference panics on path_lens[&i]")
    return inter, lens, table


CLUSTER_METHODS = ["single", "complete", "average", "weighted", "ward", "centroid", "median"]


def similarity_order(table, method="centroid"):
    """Similarity::set_table after the Jaccard table (similarity.rs:166-182) -> (reordered table,
    perm) with perm[k] = input index of the group in row / column k."""
    t = np.ascontiguousarray(table, dtype=np.float32).copy()
    n = t.shape[0]
    perm = np.zeros(max(n, 1), dtype=np.uint64)
    L = lib()
    L.orc_similarity_order.restype = C.c_int
    L.orc_similarity_order.argtypes = [C.POINTER(C.c_float), C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    if L.orc_similarity_order(_p(t, C.c_float), n, CLUSTER_METHODS.index(method), _p(perm, C.c_uint64)):
        raise IndexError("no groups: calculate_distances underflows in the reference (similarity.rs:248)")
    return t, perm[:n]


def table_rows(r, c, n_groups, node_lens=None) -> np.ndarray:
    """Body of AbacusByGroup::to_tsv without `total` (abacus.rs:1093-1112): [n_items, G] u64,
    row i-1 = item i."""
    r, c = _u64(r), _u64(c)
    if len(c) == 0:
        c = np.zeros(1, dtype=np.uint64)
    n_items = len(r) - 2
    out = np.zeros((n_items, int(n_groups)), dtype=np.uint64)
    for i in range(1, n_items + 1):
        bp = 1 if node_lens is None else int(node_lens[i])
        lib().orc_table_row(_p(r, C.c_uint64), _p(c, C.c_uint64), i, int(n_groups), bp,
                            out[i - 1].ctypes.data_as(C.POINTER(C.c_uint64)))
    return out


def table_rows_values(r, c, v, n_groups, bps=None, is_edge=False) -> np.ndarray:
    """Body of AbacusByGroup::to_tsv without `total` with multiplicities (abacus.rs:1098-1108, 1158-1166):
    [n_items, G] u64; bps[i] = the bp factor of item i (node_len - uncovered), None = 1.
    Raises where the reference's edge branch indexes past the end of v."""
    r, c = _u64(r), _u64(c)
    v = np.ascontiguousarray(v, dtype=np.uint32)
    nnz = len(c)
    if nnz == 0:
        c, v = np.zeros(1, dtype=np.uint64), np.zeros(1, dtype=np.uint32)
    n_items = len(r) - 2
    out = np.zeros((n_items, int(n_groups)), dtype=np.uint64)
    for i in range(1, n_items + 1):
        bp = 1 if bps is None else int(bps[i])
        rc = lib().orc_table_row_values(_p(r, C.c_uint64), _p(c, C.c_uint64), _p(v, C.c_uint32), nnz, i, int(n_groups), bp,
                                        int(is_edge), out[i - 1].ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc != 0:
            raise IndexError("v[j] out of range (the reference panics)")
    return out


def pansyn(seed: int, n_nodes: int, n_paths: int):
    """pansyn-v1 CSR: (items[u64], prefsum[u64, P+1], node_lens[u32, N+1])."""
    pre = np.zeros(n_paths + 1, dtype=np.uint64)
    ptr = C.POINTER(C.c_uint64)()
    n = lib().pansyn_generate(seed, n_nodes, n_paths, C.byref(ptr), _p(pre, C.c_uint64))
    items = np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].copy()
    lib().orc_free(ptr)
    lens = np.zeros(n_nodes + 1, dtype=np.uint32)
    lib().pansyn_node_lens(seed, n_nodes, _p(lens, C.c_uint32))
    return items, pre, lens


def pansyn_rearranged(seed: int, n_nodes: int, n_paths: int):
    """pansyn-v1r: pansyn-v1 with 1 % of the 64-step blocks of every path reversed, 0.1 % replaced by a copy of an earlier
    block and 0.05 % moved elsewhere in the id space (paths that are not sorted by id)."""
    items, pre, lens = pansyn(seed, n_nodes, n_paths)
    items = np.ascontiguousarray(items)
    lib().pansyn_rearrange(seed, n_nodes, n_paths, _p(items, C.c_uint64), _p(pre, C.c_uint64))
    return items, pre, lens
