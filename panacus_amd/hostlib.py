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
    if os.environ.get("PANACUS_AMD_CRASH_LOG"):  # the C stack (and the tail of a captured stderr) of a process that dies in native code goes to this file
        L.pnh_install_crash_handler.argtypes = [C.c_char_p]
        L.pnh_install_crash_handler(os.environ["PANACUS_AMD_CRASH_LOG"].encode())
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


def calc_growths(hist, pairs, n_threads: int = 0):
    """All (coverage, quorum) pairs of one histogram in one parallel region.
    pairs: list of (Threshold, Threshold). -> list of n-value curves (no NaN row)."""
    L = load()
    L.pnh_calc_all_growths.restype = C.c_int64
    L.pnh_calc_all_growths.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                       C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_uint32, C.c_uint, C.POINTER(C.c_double)]
    h = np.ascontiguousarray(hist, dtype=np.uint64)
    T = len(pairs)
    n = max(len(h) - 1, 0)
    ck = (C.c_int * T)(*[c.kind for c, _ in pairs])
    cv = (C.c_double * T)(*[float(c.value) for c, _ in pairs])
    qk = (C.c_int * T)(*[q.kind for _, q in pairs])
    qv = (C.c_double * T)(*[float(q.value) for _, q in pairs])
    out = np.zeros((T, max(n, 1)), dtype=np.float64)
    L.pnh_calc_all_growths(h.ctypes.data_as(C.POINTER(C.c_uint64)), len(h), ck, cv, qk, qv, T, n_threads,
                           out.ctypes.data_as(C.POINTER(C.c_double)))
    return [out[t, :n].copy() for t in range(T)]


def calc_growths_begin(hist, pairs, n_threads: int = 0):
    """First half of calc_growths: sets the jobs up and enqueues the device part of the quorum
    closed form (set_quorum_offload) if there is one.  Pass the result to calc_growths_end."""
    L = load()
    L.pnh_calc_all_growths_begin.restype = C.c_void_p
    L.pnh_calc_all_growths_begin.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                             C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_uint32, C.c_uint]
    h = np.ascontiguousarray(hist, dtype=np.uint64)
    T = len(pairs)
    ck = (C.c_int * T)(*[c.kind for c, _ in pairs])
    cv = (C.c_double * T)(*[float(c.value) for c, _ in pairs])
    qk = (C.c_int * T)(*[q.kind for _, q in pairs])
    qv = (C.c_double * T)(*[float(q.value) for _, q in pairs])
    handle = L.pnh_calc_all_growths_begin(h.ctypes.data_as(C.POINTER(C.c_uint64)), len(h), ck, cv, qk, qv, T, n_threads)
    return (handle, T, max(len(h) - 1, 0))


_PAIR_ARRAYS = {}  # threshold pairs as the four C arrays of pnh_calc_all_growths_begin (a pipelined loop asks with the same pairs every step)


def _pair_arrays(pairs):
    key = tuple((c.kind, float(c.value), q.kind, float(q.value)) for c, q in pairs)
    a = _PAIR_ARRAYS.get(key)
    if a is None:
        T = len(pairs)
        a = ((C.c_int * T)(*[k[0] for k in key]), (C.c_double * T)(*[k[1] for k in key]),
             (C.c_int * T)(*[k[2] for k in key]), (C.c_double * T)(*[k[3] for k in key]))
        if len(_PAIR_ARRAYS) > 64:
            _PAIR_ARRAYS.clear()
        _PAIR_ARRAYS[key] = a
    return a


def growth_tables_begin(n_groups: int, pairs) -> bool:
    """Before the coverage pass is enqueued: the first part of the device tables of (n_groups, pairs) -- two small kernels that must
    not run beside the pass -- is started now (pnx_growth_tables_begin).  False: the device path does not take these arguments."""
    L = load()
    if not getattr(L, "_tables_begin_bound", False):
        L.pnh_growth_tables_begin.restype = C.c_int
        L.pnh_growth_tables_begin.argtypes = [C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_uint32]
        L._tables_begin_bound = True
    ck, cv, qk, qv = _pair_arrays(pairs)
    return bool(L.pnh_growth_tables_begin(n_groups, ck, cv, qk, qv, len(pairs)))


def calc_growths_begin_on_device(n_groups: int, pairs):
    """The curves of the histogram of the coverage pass enqueued LAST on the offload context (set_quorum_offload),
    computed on the device without the histogram visiting the host.  None when the device path cannot take it (no
    context, libm restatements not confirmed, n out of range): fetch the histogram and use calc_growths_begin."""
    L = load()
    if not getattr(L, "_begin_bound", False):
        L.pnh_calc_all_growths_begin.restype = C.c_void_p
        L.pnh_calc_all_growths_begin.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                                 C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_uint32, C.c_uint]
        L._begin_bound = True
    ck, cv, qk, qv = _pair_arrays(pairs)
    handle = L.pnh_calc_all_growths_begin(None, n_groups + 1, ck, cv, qk, qv, len(pairs), 0)
    return (handle, len(pairs), n_groups) if handle else None


def histgrowth_resident(ctx, n_groups: int, pairs, *, drop_derived=False, drop_tables=False, hist_out=None, growth_out=None):
    """One complete histgrowth from the resident steps of `ctx` (a capi.Context whose order is set) in ONE native call
    (pnh_histgrowth_resident): pass, histogram, the curves of every pair -- from the device when ctx is the offload context
    (set_quorum_offload).  -> (hist u64[n_groups + 1], [curve f64[n_groups] per pair]); the arrays are hist_out / growth_out if given."""
    L = load()
    if not getattr(L, "_hg_bound", False):
        L.pnh_histgrowth_resident.restype = C.c_int
        L.pnh_histgrowth_resident.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                              C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L._hg_bound = True
    ck, cv, qk, qv = _pair_arrays(pairs)
    T = len(pairs)
    h = hist_out if hist_out is not None else np.zeros(n_groups + 1, dtype=np.uint64)
    g = growth_out if growth_out is not None else np.zeros((T, n_groups), dtype=np.float64)
    rc = L.pnh_histgrowth_resident(ctx._h, n_groups, ck, cv, qk, qv, T, (1 if drop_derived else 0) | (2 if drop_tables else 0),
                                   h.ctypes.data, g.ctypes.data)
    if rc:
        raise RuntimeError(f"pnh_histgrowth_resident failed ({rc}): {ctx._L.pnx_last_error(ctx._h).decode()} {L.pnh_last_error().decode()}")
    return h, [g[t] for t in range(T)]


def calc_growths_end(pending):
    """Second half: waits for the device part, finishes on the host threads. -> list of curves"""
    handle, T, n = pending
    L = load()
    L.pnh_calc_all_growths_end.restype = C.c_int64
    L.pnh_calc_all_growths_end.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
    out = np.zeros((T, max(n, 1)), dtype=np.float64)
    L.pnh_calc_all_growths_end(handle, n, T, out.ctypes.data_as(C.POINTER(C.c_double)))
    return [out[t, :n].copy() for t in range(T)]


def calc_all_growths(hist, thresholds, n_threads: int = 0):
    """Hist::calc_all_growths (hist.rs:68-87): one curve per (coverage, quorum) pair, NaN row 0."""
    curves = calc_growths(hist, list(zip(thresholds.coverage, thresholds.quorum)), n_threads)
    return [np.concatenate([[np.nan], g]) for g in curves]


# ---------------------------------------------------------------------------------------------
# GFA front end (panacus_amd/host/gfa_graph.cpp)
# ---------------------------------------------------------------------------------------------
NODE, BP, EDGE = 0, 1, 2
GROUP_PATHID, GROUP_SAMPLE, GROUP_HAPLOTYPE, GROUP_FILE = 0, 1, 2, 3


def _bind_graph(L):
    if getattr(L, "_graph_bound", False):
        return
    u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.pnh_last_error.restype = C.c_char_p
    L.pnh_graph_load.restype = C.c_void_p
    L.pnh_graph_load.argtypes = [C.c_char_p, C.c_int]
    L.pnh_graph_free.argtypes = [C.c_void_p]
    L.pnh_graph_save_cache.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.pnh_graph_from_cache.restype = C.c_void_p
    L.pnh_graph_from_cache.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    for n in ("pnh_graph_n_nodes", "pnh_graph_n_edges", "pnh_graph_n_paths"):
        getattr(L, n).restype = C.c_uint64
        getattr(L, n).argtypes = [C.c_void_p]
    L.pnh_graph_name_kind.argtypes = [C.c_void_p, C.c_char_p]
    L.pnh_graph_node_lens.restype = u32p
    L.pnh_graph_node_lens.argtypes = [C.c_void_p]
    L.pnh_graph_path_name.restype = C.c_uint64
    L.pnh_graph_path_name.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64]
    L.pnh_graph_item_table.restype = C.c_int64
    L.pnh_graph_item_table.argtypes = [C.c_void_p, C.c_int, u32p, u64p]
    L.pnh_graph_path_order.restype = C.c_int64
    L.pnh_graph_path_order.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, u32p, u32p,
                                       u64p, C.c_char_p, C.c_uint64]
    L.pnh_graph_exclude_flags.restype = C.c_int
    L.pnh_graph_exclude_flags.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint8)]
    L.pnh_graph_cut_upload.restype = C.c_int
    L.pnh_graph_cut_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int,
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.pnh_graph_masked_table.restype = C.c_int
    L.pnh_graph_masked_table.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, u64p, u64p, u32p,
                                         u64p, C.POINTER(C.c_uint8), u32p, u64p]
    L.pnh_graph_edge_relabel.restype = C.c_int
    L.pnh_graph_edge_relabel.argtypes = [C.c_void_p, u32p]
    L._graph_bound = True


class GfaGraph:
    """GraphStorage + GraphMask of the reference (graph.rs:163-375, abacus.rs:46-347), host side."""

    def __init__(self, gfa_file: str, index_edges: bool = False):
        self._L = load()
        _bind_graph(self._L)
        self._h = self._L.pnh_graph_load(os.fsencode(gfa_file), int(index_edges))
        if not self._h:
            raise ValueError(self._L.pnh_last_error().decode())

    @classmethod
    def from_cache(cls, cache_file: str, gfa_file: str, need_edges: bool = False):
        """the graph from a .pcsr cache, or None when the cache is missing / stale / without edges"""
        L = load()
        _bind_graph(L)
        h = L.pnh_graph_from_cache(os.fsencode(cache_file), os.fsencode(gfa_file), int(need_edges))
        if not h:
            return None
        g = cls.__new__(cls)
        g._L, g._h = L, h
        return g

    def save_cache(self, cache_file: str, gfa_file: str):
        if self._L.pnh_graph_save_cache(self._h, os.fsencode(cache_file), os.fsencode(gfa_file)):
            raise OSError(self._L.pnh_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.pnh_graph_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_nodes(self):
        return int(self._L.pnh_graph_n_nodes(self._h))

    def name_kind(self):
        """-> (kind, prefix): 0 names the device does not take, 1 number = rank, 2 number through a table, 3 hashed bytes"""
        buf = C.create_string_buffer(9)
        k = int(self._L.pnh_graph_name_kind(self._h, buf))
        return k, buf.value.decode()

    @property
    def n_edges(self):
        return int(self._L.pnh_graph_n_edges(self._h))

    @property
    def n_paths(self):
        return int(self._L.pnh_graph_n_paths(self._h))

    def n_items(self, count_type):
        return self.n_edges if count_type == EDGE else self.n_nodes

    @property
    def node_lens(self):
        p = self._L.pnh_graph_node_lens(self._h)
        return np.ctypeslib.as_array(p, shape=(self.n_nodes + 1,)).copy()

    def path_names(self):
        out = []
        buf = C.create_string_buffer(4096)
        for i in range(self.n_paths):
            self._L.pnh_graph_path_name(self._h, i, buf, 4096)
            out.append(buf.value.decode())
        return out

    def item_table(self, count_type):
        pre = np.zeros(self.n_paths + 1, dtype=np.uint64)
        n = self._L.pnh_graph_item_table(self._h, count_type, None, pre.ctypes.data_as(C.POINTER(C.c_uint64)))
        if n < 0:
            raise ValueError(self._L.pnh_last_error().decode())
        items = np.zeros(max(n, 1), dtype=np.uint32)
        n = self._L.pnh_graph_item_table(self._h, count_type, items.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         pre.ctypes.data_as(C.POINTER(C.c_uint64)))
        if n < 0:
            raise ValueError(self._L.pnh_last_error().decode())
        return items[:n], pre

    def exclude_flags(self, count_type, exclude_file, group_mode=GROUP_PATHID, group_file=None):
        flags = np.zeros(self.n_items(count_type) + 1, dtype=np.uint8)
        rc = self._L.pnh_graph_exclude_flags(self._h, count_type, group_mode,
                                             os.fsencode(group_file) if group_file else None,
                                             os.fsencode(exclude_file), flags.ctypes.data_as(C.POINTER(C.c_uint8)))
        if rc != 0:
            raise ValueError(self._L.pnh_last_error().decode())
        return flags

    def masked_table(self, count_type, subset_file=None, exclude_file=None, group_mode=GROUP_PATHID, group_file=None):
        """ItemTable, exclude flags and uncovered bps under BED -s / -e lists (coordinates allowed).
        -> (items[u32], prefsum[u64], exclude[u8, n_items+1], uncov_ids[u32], uncov_bps[u64])"""
        enc = lambda f: os.fsencode(f) if f else None  # noqa: E731
        n_steps, n_unc = C.c_uint64(0), C.c_uint64(0)
        args = (self._h, count_type, group_mode, enc(group_file), enc(subset_file), enc(exclude_file))
        if self._L.pnh_graph_masked_table(*args, C.byref(n_steps), C.byref(n_unc), None, None, None, None, None) != 0:
            raise ValueError(self._L.pnh_last_error().decode())
        items = np.zeros(max(n_steps.value, 1), dtype=np.uint32)
        pre = np.zeros(self.n_paths + 1, dtype=np.uint64)
        flags = np.zeros(self.n_items(count_type) + 1, dtype=np.uint8)
        ids = np.zeros(max(n_unc.value, 1), dtype=np.uint32)
        bps = np.zeros(max(n_unc.value, 1), dtype=np.uint64)
        rc = self._L.pnh_graph_masked_table(*args, C.byref(n_steps), C.byref(n_unc),
                                            items.ctypes.data_as(C.POINTER(C.c_uint32)),
                                            pre.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            flags.ctypes.data_as(C.POINTER(C.c_uint8)),
                                            ids.ctypes.data_as(C.POINTER(C.c_uint32)),
                                            bps.ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc != 0:
            raise ValueError(self._L.pnh_last_error().decode())
        return items[: n_steps.value], pre, flags, ids[: n_unc.value], bps[: n_unc.value]

    def cut_upload(self, ctx, count_type, subset_file=None, exclude_file=None, group_mode=GROUP_PATHID, group_file=None,
                   growth_weights=False):
        """The walks cut by the -s / -e lists ON THE DEVICE into the capi.Context `ctx` (pnx_set_csr_cut + the host's
        replay of the partial pieces) -> (uncov_ids[u32], uncov_bps[u64]); the cut table is ctx's resident graph."""
        enc = lambda f: os.fsencode(f) if f else None  # noqa: E731
        cap = self.n_items(count_type) + 1
        n_unc = C.c_uint64(cap)
        ids = np.zeros(cap, dtype=np.uint32)
        bps = np.zeros(cap, dtype=np.uint64)
        rc = self._L.pnh_graph_cut_upload(self._h, ctx._h, count_type, group_mode, enc(group_file), enc(subset_file),
                                          enc(exclude_file), int(growth_weights), C.byref(n_unc),
                                          ids.ctypes.data_as(C.POINTER(C.c_uint32)), bps.ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc != 0:
            raise ValueError(self._L.pnh_last_error().decode())
        ctx.n_items = self.n_items(count_type)
        return ids[: n_unc.value], bps[: n_unc.value]

    def edge_keys(self) -> np.ndarray:
        """key[edge id] = (smaller node id << 32) | larger node id of the canonical edge; [0] = 0: the
        `item_key` of pnx_set_csr_keyed (capi.Context.set_csr(..., item_key=...))"""
        out = np.zeros(self.n_edges + 1, dtype=np.uint64)
        self._L.pnh_graph_edge_keys.restype = C.c_int
        self._L.pnh_graph_edge_keys.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        if self._L.pnh_graph_edge_keys(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))) != 0:
            raise ValueError(self._L.pnh_last_error().decode())
        return out

    def edge_relabel(self) -> np.ndarray:
        """new_id[old edge id] = rank by canonical (smaller node, larger node, orientations); [0] = 0.
        What the CLI renumbers edge steps by before the upload (hist / growth do not depend on ids)."""
        out = np.zeros(self.n_edges + 1, dtype=np.uint32)
        if self._L.pnh_graph_edge_relabel(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32))) != 0:
            raise ValueError(self._L.pnh_last_error().decode())
        return out

    def path_order(self, group_mode=GROUP_PATHID, group_file=None, order_file=None, subset_file=None,
                   exclude_file=None):
        P = max(self.n_paths, 1)
        pi = np.zeros(P, dtype=np.uint32)
        gi = np.zeros(P, dtype=np.uint32)
        n_out = C.c_uint64(0)
        cap = 1 << 20
        while True:
            buf = C.create_string_buffer(cap)
            ng = self._L.pnh_graph_path_order(self._h, group_mode, os.fsencode(group_file) if group_file else None,
                                              os.fsencode(order_file) if order_file else None,
                                              os.fsencode(subset_file) if subset_file else None,
                                              os.fsencode(exclude_file) if exclude_file else None,
                                              pi.ctypes.data_as(C.POINTER(C.c_uint32)),
                                              gi.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n_out), buf, cap)
            if ng < 0 and b"too small" in self._L.pnh_last_error():
                cap *= 4
                continue
            break
        if ng < 0:
            raise ValueError(self._L.pnh_last_error().decode())
        names = buf.value.decode().split("\n") if ng else []
        return pi[: n_out.value].copy(), gi[: n_out.value].copy(), names


# ---------------------------------------------------------------------------------------------
# CLI in-process (panacus_amd/host/commands.cpp) and table helpers
# ---------------------------------------------------------------------------------------------
def run_cli(args):
    """Runs `panacus-amd <args...>` inside this process. -> (exit_code, stdout_text, stderr_text)"""
    L = load()
    L.pnh_run_cli.restype = C.c_int
    L.pnh_run_cli.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.c_uint64,
                              C.POINTER(C.c_uint64)]
    joined = "\n".join(["panacus-amd"] + [str(a) for a in args]).encode()
    cap = 1 << 22
    while True:
        out, err = C.create_string_buffer(cap), C.create_string_buffer(1 << 16)
        ol, el = C.c_uint64(0), C.c_uint64(0)
        rc = L.pnh_run_cli(joined, out, cap, C.byref(ol), err, 1 << 16, C.byref(el))
        if ol.value + 1 > cap:
            cap = ol.value + 16
            continue
        return rc, out.value.decode(), err.value.decode()


run_cli_inprocess = run_cli  # (tests/conftest.py may point run_cli at the panacus-amd binary; this name always runs in this process)


def format_f64(x: float) -> str:
    L = load()
    L.pnh_format_f64.restype = C.c_uint64
    L.pnh_format_f64.argtypes = [C.c_double, C.c_char_p, C.c_uint64]
    buf = C.create_string_buffer(512)
    L.pnh_format_f64(float(x), buf, 512)
    return buf.value.decode()


def format_f32(x) -> str:
    """Rust `{}` of an f32 (the cells of the similarity table)."""
    L = load()
    L.pnh_format_f32.restype = C.c_uint64
    L.pnh_format_f32.argtypes = [C.c_float, C.c_char_p, C.c_uint64]
    buf = C.create_string_buffer(512)
    L.pnh_format_f32(float(x), buf, 512)
    return buf.value.decode()


def json_f64(x: float) -> str:
    """an f64 as the report JSON prints it (serde_json / ryu)"""
    L = load()
    L.pnh_json_f64.restype = C.c_uint64
    L.pnh_json_f64.argtypes = [C.c_double, C.c_char_p, C.c_uint64]
    buf = C.create_string_buffer(64)
    L.pnh_json_f64(float(x), buf, 64)
    return buf.value.decode()


def json_f32(x) -> str:
    L = load()
    L.pnh_json_f32.restype = C.c_uint64
    L.pnh_json_f32.argtypes = [C.c_float, C.c_char_p, C.c_uint64]
    buf = C.create_string_buffer(64)
    L.pnh_json_f32(float(x), buf, 64)
    return buf.value.decode()


CLUSTER_METHODS = ["single", "complete", "average", "weighted", "ward", "centroid", "median"]


def similarity_order(table, method="centroid") -> np.ndarray:
    """perm[k] = input index of the group printed in row / column k of `similarity`
    (similarity.rs:166-182: Euclidean row distances, kodama linkage, dendrogram order)."""
    L = load()
    t = np.ascontiguousarray(table, dtype=np.float32)
    n = t.shape[0]
    perm = np.zeros(n, dtype=np.uint64)
    L.pnh_similarity_order.argtypes = [C.POINTER(C.c_float), C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    L.pnh_last_error.restype = C.c_char_p
    if L.pnh_similarity_order(t.ctypes.data_as(C.POINTER(C.c_float)), n, CLUSTER_METHODS.index(method),
                              perm.ctypes.data_as(C.POINTER(C.c_uint64))):
        raise RuntimeError(L.pnh_last_error().decode())
    return perm


def linkage(condensed, n, method="centroid"):
    """kodama::linkage restated on the host: (c1[n-1], c2[n-1], dissimilarity[n-1]) with SciPy labels"""
    L = load()
    d = np.ascontiguousarray(condensed, dtype=np.float32)
    c1 = np.zeros(max(n - 1, 1), dtype=np.uint64)
    c2 = np.zeros(max(n - 1, 1), dtype=np.uint64)
    ds = np.zeros(max(n - 1, 1), dtype=np.float32)
    u64p = C.POINTER(C.c_uint64)
    L.pnh_linkage.argtypes = [C.POINTER(C.c_float), C.c_uint64, C.c_int, u64p, u64p, C.POINTER(C.c_float)]
    L.pnh_last_error.restype = C.c_char_p
    if L.pnh_linkage(d.ctypes.data_as(C.POINTER(C.c_float)), n, CLUSTER_METHODS.index(method), c1.ctypes.data_as(u64p),
                     c2.ctypes.data_as(u64p), ds.ctypes.data_as(C.POINTER(C.c_float))):
        raise RuntimeError(L.pnh_last_error().decode())
    k = max(n - 1, 0)
    return c1[:k], c2[:k], ds[:k]


def pool_threads() -> int:
    """threads of the host worker pool (hardware threads cut down to the cgroup CPU quota, max 64)"""
    L = load()
    L.pnh_pool_threads.restype = C.c_uint32
    return int(L.pnh_pool_threads())


def usable_cpus() -> int:
    """hardware threads of this process cut down to its cgroup CPU bandwidth quota"""
    L = load()
    L.pnh_usable_cpus.restype = C.c_uint32
    return int(L.pnh_usable_cpus())


def set_quorum_offload(ctx=None, min_n: int = 256):
    """Quorum closed form with n >= min_n: the O(n^3) inner sums run on the GPU of `ctx`
    (a capi.Context), bit-identical to the host path; None switches it off."""
    L = load()
    L.pnh_set_quorum_offload.argtypes = [C.c_void_p, C.c_uint64]
    L.pnh_set_quorum_offload(None if ctx is None else ctx._h, int(min_n))


def device_growth_usable() -> bool:
    """True iff the restated log2 and exp2 both reproduce this platform's libm bit for bit (whole closed forms may then
    run on the device)"""
    L = load()
    return bool(L.pnh_device_growth_usable())


def log2_restated(x) -> np.ndarray:
    """log2 by the restatement of libm's algorithm (csrc/log2_exact.hpp) on the host"""
    L = load()
    L.pnh_log2_restated.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_uint64]
    L.pnh_log2_restated.restype = None
    a = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros_like(a)
    L.pnh_log2_restated(a.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double)), len(a))
    return y


def quorum_offload_usable() -> bool:
    """True iff the restated exp2 reproduces this platform's libm bit for bit"""
    L = load()
    return bool(L.pnh_quorum_offload_usable())
