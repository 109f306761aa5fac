"""ctypes binding of libpanacus_hip.so -- the C ABI declared in include/panacus_amd.h.

This is the same seam a Rust host would bind with ``extern "C"``.  There
is no CPU fallback: loading fails loudly if the HIP library has not been built, and every
call fails with PNX_ENODEV when no GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpanacus_hip.so")

PNX_OK, PNX_EINVAL, PNX_ENODEV, PNX_EHIP, PNX_ENOMEM, PNX_ELIMIT = 0, -1, -2, -3, -4, -5
K_INDEX, K_SCATTER, K_COVER, K_HIST, K_MASK, K_GROWTH, K_PAIRS, K_COUNT = range(8)
KERNEL_SLOT_NAMES = ["index", "scatter", "cover", "hist", "mask", "growth", "pairs"]
CFG_CACHE_INDEX, CFG_TILE_BLOCKS, CFG_KEEP_PRESENCE, CFG_COVER_VARIANT, CFG_INDEX_COARSE, CFG_COVER_WAVES, CFG_USE_WEIGHTS, CFG_BLOCKING_SYNC, CFG_COVER_SPLIT, CFG_INDEX_BY_ENTRY, CFG_COVER_SKIP, CFG_INDEX_PROBE, CFG_COMM_REDUCE_HIST, CFG_OVERLAP_PHASES, CFG_SORT_SHUFFLED, CFG_PAIRS_VARIANT, CFG_ROWS_LAYOUT, CFG_DROP_DERIVED, CFG_MAX_IN_FLIGHT, CFG_DROP_GROWTH_TABLES, CFG_HIST_IN_COVER, CFG_COVER_ROUTE, CFG_ROWS_KERNEL = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23

# every symbol include/panacus_amd.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "pnx_init", "pnx_free", "pnx_last_error", "pnx_version", "pnx_abi_version", "pnx_set_csr_gfa_sized", "pnx_gfa_walks_sized", "pnx_growth_tables_begin", "pnx_set_csr", "pnx_set_csr_keyed", "pnx_set_csr_pansyn", "pnx_set_csr_pansyn_shard", "pnx_set_csr_pansyn_rearranged", "pnx_set_exclude",
    "pnx_get_csr", "pnx_set_order", "pnx_hist", "pnx_hist_async", "pnx_hist_device", "pnx_hist_fetch", "pnx_hist_enqueued", "pnx_hist_enqueued_on",
    "pnx_sync", "pnx_stream", "pnx_ordered_growth", "pnx_ordered_growth_async",
    "pnx_ordered_growth_device", "pnx_ordered_growth_fetch", "pnx_ordered_growth_enqueued", "pnx_profile_enable", "pnx_profile_read",
    "pnx_profile_reset", "pnx_profile_select", "pnx_config", "pnx_info", "pnx_info_sized", "pnx_group_intersections",
    "pnx_group_intersections_device", "pnx_presence_row_words", "pnx_presence", "pnx_quorum_sums",
    "pnx_quorum_sums_async", "pnx_quorum_sums_fetch", "pnx_exp2_exact", "pnx_group_visit_counts", "pnx_share_csr",
    "pnx_comm_unique_id", "pnx_comm_init", "pnx_comm_allreduce_u64", "pnx_comm_free", "pnx_comm_barrier",
    "pnx_set_csr_cut", "pnx_set_weights", "pnx_exclude_items", "pnx_get_exclude", "pnx_preload", "pnx_init_flags", "pnx_prepare",
    "pnx_log2_exact", "pnx_growth_closed_form_async", "pnx_growth_closed_form_fetch", "pnx_gfa_text_upload", "pnx_set_csr_gfa",
    "pnx_profile_sample", "pnx_gfa_walks", "pnx_set_csr_walks",
]


class PnxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"panacus_amd error {code}: {msg}")
        self.code = code


class PnxInfo(C.Structure):
    _fields_ = [("n_steps", C.c_uint64), ("n_items", C.c_uint32), ("n_paths", C.c_uint32),
                ("n_ordered", C.c_uint32), ("n_groups", C.c_uint32), ("n_tiles", C.c_uint32),
                ("tile_items", C.c_uint32), ("n_general_paths", C.c_uint32), ("weighted", C.c_uint32),
                ("n_run_paths", C.c_uint32), ("n_scatter_paths", C.c_uint32), ("n_runs", C.c_uint64),
                ("n_reruns", C.c_uint64), ("n_sorted_paths", C.c_uint32), ("rows_tile_major", C.c_uint32),
                ("n_rows", C.c_uint64), ("n_rows_in_order", C.c_uint64), ("n_growth_table_builds", C.c_uint64),
                ("n_band_passes", C.c_uint32), ("band_route_failed", C.c_uint32), ("n_rows_q_passes", C.c_uint64),
                ("n_spilled_last", C.c_uint32), ("band_splits", C.c_uint32), ("n_spilled_total", C.c_uint64),
                ("n_spill_bursts_last", C.c_uint32), ("n_loose_groups_last", C.c_uint32), ("n_path_cuts", C.c_uint32),
                ("n_band_entries", C.c_uint32), ("n_sorted_copies", C.c_uint32), ("reserved1", C.c_uint32)]


class PnxWalks(C.Structure):  # pnx_walks (include/panacus_amd.h)
    _fields_ = [("walk_node", C.POINTER(C.c_uint32)), ("walk_backward", C.POINTER(C.c_uint8)),
                ("walk_off", C.POINTER(C.c_uint64)), ("path_start", C.POINTER(C.c_uint64)),
                ("path_mode", C.POINTER(C.c_uint8)), ("n_paths", C.c_uint32), ("n_nodes", C.c_uint32),
                ("node_len", C.POINTER(C.c_uint32)), ("edge_item", C.POINTER(C.c_uint32)),
                ("edge_off", C.POINTER(C.c_uint64)), ("edge_uv", C.POINTER(C.c_uint64)), ("edge_oo", C.POINTER(C.c_uint8)),
                ("n_items", C.c_uint32), ("count_type", C.c_int),
                ("track_covered", C.c_int), ("inc_off", C.POINTER(C.c_uint64)), ("inc_iv", C.POINTER(C.c_uint64)),
                ("exc_off", C.POINTER(C.c_uint64)), ("exc_iv", C.POINTER(C.c_uint64))]


class PnxGfaSteps(C.Structure):  # pnx_gfa_steps
    _fields_ = [("text", C.c_char_p), ("text_bytes", C.c_uint64), ("n_paths", C.c_uint32), ("n_nodes", C.c_uint32),
                ("col_begin", C.POINTER(C.c_uint64)), ("col_end", C.POINTER(C.c_uint64)), ("is_walk", C.POINTER(C.c_uint8)),
                ("id_of_name", C.POINTER(C.c_uint32)), ("n_names", C.c_uint64),
                ("edge_uv", C.POINTER(C.c_uint64)), ("edge_oo", C.POINTER(C.c_uint8)), ("n_edges", C.c_uint32),
                ("name_off", C.POINTER(C.c_uint64)), ("name_len", C.POINTER(C.c_uint8)),
                ("link_off", C.POINTER(C.c_uint64)), ("n_links", C.c_uint64), ("link_lo", C.c_uint64), ("link_hi", C.c_uint64),
                ("name_lo", C.c_uint64), ("name_hi", C.c_uint64), ("name_prefix", C.c_char * 8), ("name_prefix_len", C.c_uint32)]


LINKS_FIND = 0xFFFFFFFFFFFFFFFF  # pnx_gfa_steps.n_links: the library finds the L lines itself


def _find_links(g, find_links, find_names=None, name_prefix=None):
    if name_prefix:
        g.name_prefix = name_prefix[:8].ljust(8, b"\0") if len(name_prefix) < 8 else name_prefix[:8]
        g.name_prefix_len = len(name_prefix)
    if find_names is not None and find_names is not False:  # PNX_NAMES_FIND: True, or the byte range of the S lines
        g.n_names = LINKS_FIND
        if find_names is not True:
            g.name_lo, g.name_hi = int(find_names[0]), int(find_names[1])
    if find_links is None or find_links is False:
        return
    g.n_links = LINKS_FIND
    if find_links is not True:
        g.link_lo, g.link_hi = int(find_links[0]), int(find_links[1])


class PnxPieceEvent(C.Structure):  # pnx_piece_event
    _fields_ = [("step", C.c_uint64), ("last_full", C.c_uint64), ("path", C.c_uint32), ("item", C.c_uint32),
                ("a", C.c_uint32), ("b", C.c_uint32), ("piece", C.c_uint32), ("kind", C.c_uint8),
                ("flagged", C.c_uint8), ("pad", C.c_uint8 * 2)]


WALK_SKIP, WALK_WHOLE, WALK_CUT = 0, 1, 2

_lib = None


def load() -> C.CDLL:
    """Load libpanacus_hip.so (built in-tree by panacus_amd._build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m panacus_amd._build` "
            "(hipcc, gfx950). panacus_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u32p, u64p, u8p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
    L.pnx_init.argtypes = [C.POINTER(vp), C.c_int]
    L.pnx_init_flags.argtypes = [C.POINTER(vp), C.c_int, C.c_uint32]
    L.pnx_free.argtypes = [vp]
    L.pnx_free.restype = None
    L.pnx_last_error.argtypes = [vp]
    L.pnx_last_error.restype = C.c_char_p
    L.pnx_version.restype = C.c_char_p
    L.pnx_set_csr.argtypes = [vp, u32p, u64p, C.c_uint32, C.c_uint32, u32p, u8p]
    L.pnx_set_csr_keyed.argtypes = [vp, u32p, u64p, C.c_uint32, C.c_uint32, u32p, u8p, u64p]
    L.pnx_set_csr_pansyn.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
    L.pnx_set_csr_pansyn_shard.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
    L.pnx_set_csr_pansyn_rearranged.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
    L.pnx_set_exclude.argtypes = [vp, u8p]
    L.pnx_prepare.argtypes = [vp]
    L.pnx_gfa_text_upload.argtypes = [vp, C.c_char_p, C.c_uint64]
    L.pnx_set_csr_gfa.argtypes = [vp, C.POINTER(PnxGfaSteps), u32p, u8p]
    L.pnx_gfa_walks.argtypes = [vp, C.POINTER(PnxGfaSteps), C.POINTER(C.c_uint64)]
    L.pnx_set_csr_walks.argtypes = [vp, C.c_uint32, u32p, u8p, u64p, u8p, C.c_uint32]
    L.pnx_get_exclude.argtypes = [vp, u8p]
    L.pnx_set_weights.argtypes = [vp, u32p]
    L.pnx_exclude_items.argtypes = [vp, u32p, C.c_uint32]
    L.pnx_set_csr_cut.argtypes = [vp, C.POINTER(PnxWalks), u32p, u64p, C.POINTER(PnxPieceEvent), C.c_uint64, u64p]
    L.pnx_get_csr.argtypes = [vp, u64p, u32p, u64p, u32p]
    L.pnx_set_order.argtypes = [vp, u32p, u32p, C.c_uint32, C.c_uint32]
    L.pnx_hist.argtypes = [vp, u32p, u64p]
    L.pnx_hist_async.argtypes = [vp]
    L.pnx_hist_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.pnx_hist_fetch.argtypes = [vp, u32p, u64p]
    L.pnx_hist_enqueued.argtypes = [vp, C.POINTER(vp)]
    L.pnx_hist_enqueued_on.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.pnx_sync.argtypes = [vp]
    L.pnx_stream.argtypes = [vp]
    L.pnx_stream.restype = vp
    L.pnx_ordered_growth.argtypes = [vp, u32p, C.c_uint32, u32p, u32p, C.c_uint32, u64p]
    L.pnx_ordered_growth_async.argtypes = [vp, u32p, C.c_uint32, u32p, u32p, C.c_uint32]
    L.pnx_ordered_growth_device.argtypes = [vp, C.POINTER(vp)]
    L.pnx_ordered_growth_fetch.argtypes = [vp, u64p]
    L.pnx_ordered_growth_enqueued.argtypes = [vp, C.POINTER(vp)]
    L.pnx_group_intersections.argtypes = [vp, u64p]
    L.pnx_group_intersections_device.argtypes = [vp, C.POINTER(vp)]
    L.pnx_presence_row_words.argtypes = [vp]
    L.pnx_presence_row_words.restype = C.c_uint64
    L.pnx_presence.argtypes = [vp, u64p]
    L.pnx_share_csr.argtypes = [vp, vp]
    L.pnx_comm_unique_id.argtypes = [u8p]
    L.pnx_comm_init.argtypes = [vp, u8p, C.c_int, C.c_int]
    L.pnx_comm_allreduce_u64.argtypes = [vp, vp, C.c_size_t]
    L.pnx_comm_free.argtypes = [vp]
    L.pnx_comm_barrier.argtypes = [vp]
    L.pnx_group_visit_counts.argtypes = [vp, C.c_uint32, C.c_uint32, u32p]
    f64p = C.POINTER(C.c_double)
    L.pnx_quorum_sums.argtypes = [vp, C.c_uint32, C.c_uint32, u32p, f64p, f64p, f64p, C.POINTER(f64p)]
    L.pnx_exp2_exact.argtypes = [vp, f64p, f64p, C.c_uint64]
    L.pnx_log2_exact.argtypes = [vp, f64p, f64p, C.c_uint64]
    L.pnx_growth_closed_form_async.argtypes = [vp, u64p, C.c_uint32, C.c_uint32, u32p, u32p, f64p]
    L.pnx_growth_closed_form_fetch.argtypes = [vp, f64p]
    L.pnx_profile_enable.argtypes = [vp, C.c_int]
    L.pnx_profile_select.argtypes = [vp, C.c_uint32]
    L.pnx_profile_sample.argtypes = [vp, C.c_uint32]
    L.pnx_profile_read.argtypes = [vp, C.POINTER(C.c_double), u64p]
    L.pnx_profile_reset.argtypes = [vp]
    L.pnx_config.argtypes = [vp, C.c_int, C.c_int64]
    L.pnx_info.argtypes = [vp, C.POINTER(PnxInfo)]
    L.pnx_info_sized.argtypes = [vp, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    _lib = L
    return L


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


class Context:
    """One engine context = one GPU (one process per GPU)."""

    def __init__(self, device: int = 0, one_shot: bool = False):
        """one_shot: pnx_init_flags(PNX_INIT_ONE_SHOT) -- the two extra pass streams are made when passes first overlap"""
        self._L = load()
        h = C.c_void_p()
        rc = self._L.pnx_init_flags(C.byref(h), device, 1 if one_shot else 0)
        if rc != PNX_OK:
            raise PnxError(rc, self._L.pnx_last_error(None).decode())
        self._h = h
        self.n_groups = 0
        self.n_items = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.pnx_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc):
        if rc != PNX_OK:
            raise PnxError(rc, self._L.pnx_last_error(self._h).decode())

    # ---- graph ----
    def set_csr(self, items, path_off, n_items, weights=None, exclude=None, item_key=None):
        """item_key (n_items+1 u64, optional): sort keys the paths follow (edges: canonical ends); the
        library then renumbers the items internally, results stay in the caller's ids"""
        items = np.ascontiguousarray(items, dtype=np.uint32)
        path_off = np.ascontiguousarray(path_off, dtype=np.uint64)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.uint32)
        ex = None if exclude is None else np.ascontiguousarray(exclude, dtype=np.uint8)
        # the ABI takes no length for `items`: the library copies path_off[n_paths] ids from it
        if len(path_off) < 1:
            raise ValueError("path_off must hold n_paths+1 offsets (at least one)")
        if int(path_off[-1]) != len(items):
            raise ValueError(f"path_off[-1] = {int(path_off[-1])} but items has {len(items)} entries")
        if w is not None and len(w) != n_items + 1:
            raise ValueError("weights must have n_items+1 entries")
        if ex is not None and len(ex) != n_items + 1:
            raise ValueError("exclude must have n_items+1 entries")
        if item_key is not None:
            key = np.ascontiguousarray(item_key, dtype=np.uint64)
            if len(key) != n_items + 1:
                raise ValueError("item_key must have n_items+1 entries")
            self._ck(self._L.pnx_set_csr_keyed(self._h, _ptr(items, C.c_uint32), _ptr(path_off, C.c_uint64), len(path_off) - 1,
                                               n_items, _ptr(w, C.c_uint32), _ptr(ex, C.c_uint8), _ptr(key, C.c_uint64)))
        else:
            self._ck(self._L.pnx_set_csr(self._h, _ptr(items, C.c_uint32), _ptr(path_off, C.c_uint64),
                                         len(path_off) - 1, n_items, _ptr(w, C.c_uint32), _ptr(ex, C.c_uint8)))
        self.n_items = n_items

    def set_exclude(self, exclude=None):
        """replace the exclusion flags (n_items+1 u8, None = no exclusion) of the resident graph"""
        ex = None if exclude is None else np.ascontiguousarray(exclude, dtype=np.uint8)
        if ex is not None and len(ex) != self.n_items + 1:
            raise ValueError("exclude must have n_items+1 entries")
        self._ck(self._L.pnx_set_exclude(self._h, _ptr(ex, C.c_uint8)))

    def get_exclude(self) -> np.ndarray:
        ex = np.zeros(self.info().n_items + 1, dtype=np.uint8)
        self._ck(self._L.pnx_get_exclude(self._h, _ptr(ex, C.c_uint8)))
        return ex

    def set_weights(self, weights):
        w = np.ascontiguousarray(weights, dtype=np.uint32)
        if len(w) != self.n_items + 1:
            raise ValueError("weights must have n_items+1 entries")
        self._ck(self._L.pnx_set_weights(self._h, _ptr(w, C.c_uint32)))

    def exclude_items(self, ids):
        v = np.ascontiguousarray(ids, dtype=np.uint32)
        self._ck(self._L.pnx_exclude_items(self._h, _ptr(v, C.c_uint32), len(v)))

    def set_csr_cut(self, walk_node, walk_off, node_len, path_mode, inc, exc=None, path_start=None, walk_backward=None,
                    count_type=0, edge_item=None, edge_off=None, n_items=None, weights=None, item_key=None,
                    track_covered=False, max_events=None, edge_uv=None, edge_oo=None):
        """pnx_set_csr_cut: inc / exc = per path a list of (start, end) pairs (sorted, disjoint, not touching;
        exc None = no exclude list).  -> list of event dicts (bp counts)"""
        keep = []

        def arr(x, dt):
            a = np.ascontiguousarray(x, dtype=dt)
            if len(a) == 0:
                a = np.zeros(1, dtype=dt)
            keep.append(a)
            return a

        P = len(walk_off) - 1

        def lists(ls):
            off = np.zeros(P + 1, dtype=np.uint64)
            flat = []
            for k in range(P):
                for s, e in ls[k]:
                    flat += [int(s), int(e)]
                off[k + 1] = len(flat) // 2
            return arr(off, np.uint64), arr(np.array(flat, dtype=np.uint64), np.uint64)

        n_nodes = len(node_len) - 1
        w = PnxWalks()
        # walk_node None: the walks a preceding gfa_walks() left on the device (walk_off = the offsets it returned)
        w.walk_node = None if walk_node is None else _ptr(arr(walk_node, np.uint32), C.c_uint32)
        w.walk_backward = None if walk_backward is None else _ptr(arr(walk_backward, np.uint8), C.c_uint8)
        w.walk_off = _ptr(arr(walk_off, np.uint64), C.c_uint64)
        w.path_start = _ptr(arr(np.zeros(P, dtype=np.uint64) if path_start is None else path_start, np.uint64), C.c_uint64)
        w.path_mode = _ptr(arr(path_mode, np.uint8), C.c_uint8)
        w.n_paths, w.n_nodes = P, n_nodes
        w.node_len = _ptr(arr(node_len, np.uint32), C.c_uint32)
        if edge_off is not None:
            w.edge_item = _ptr(arr(edge_item, np.uint32), C.c_uint32)
            w.edge_off = _ptr(arr(edge_off, np.uint64), C.c_uint64)
        if edge_uv is not None:
            w.edge_uv = _ptr(arr(edge_uv, np.uint64), C.c_uint64)
            w.edge_oo = _ptr(arr(edge_oo, np.uint8), C.c_uint8)
        w.n_items = n_nodes if n_items is None else n_items
        w.count_type = count_type
        w.track_covered = int(track_covered)
        io, ii = lists(inc)
        w.inc_off, w.inc_iv = _ptr(io, C.c_uint64), _ptr(ii, C.c_uint64)
        n_iv = int(io[-1])
        if exc is not None:
            eo, ei = lists(exc)
            w.exc_off, w.exc_iv = _ptr(eo, C.c_uint64), _ptr(ei, C.c_uint64)
            n_iv += int(eo[-1])
        cap = 2 * n_iv + 16 if max_events is None else max_events
        ev = (PnxPieceEvent * max(cap, 1))()
        n_ev = C.c_uint64(0)
        wt = None if weights is None else arr(weights, np.uint32)
        key = None if item_key is None else arr(item_key, np.uint64)
        self._ck(self._L.pnx_set_csr_cut(self._h, C.byref(w), _ptr(wt, C.c_uint32), _ptr(key, C.c_uint64), ev, cap, C.byref(n_ev)))
        self.n_items = int(w.n_items)
        return [dict(step=e.step, last_full=e.last_full, path=e.path, item=e.item, a=e.a, b=e.b, piece=e.piece,
                     kind=e.kind, flagged=e.flagged) for e in ev[: n_ev.value]]

    def set_csr_pansyn(self, seed, n_nodes, n_paths, with_weights=False):
        self._ck(self._L.pnx_set_csr_pansyn(self._h, seed, n_nodes, n_paths, int(with_weights)))
        self.n_items = n_nodes

    def set_csr_pansyn_rearranged(self, seed, n_nodes, n_paths, with_weights=False):
        """pnx_set_csr_pansyn_rearranged: pansyn-v1r, paths that are not sorted by id (inversions, jumps back, translocations)"""
        self._ck(self._L.pnx_set_csr_pansyn_rearranged(self._h, seed, n_nodes, n_paths, int(with_weights)))
        self.n_items = n_nodes

    def set_csr_pansyn_shard(self, seed, node_lo, n_nodes, n_paths, with_weights=False):
        """pnx_set_csr_pansyn_shard: the nodes node_lo + 1 .. node_lo + n_nodes of the pansyn graph as items 1 .. n_nodes"""
        self._ck(self._L.pnx_set_csr_pansyn_shard(self._h, seed, node_lo, n_nodes, n_paths, int(with_weights)))
        self.n_items = n_nodes

    @staticmethod
    def preload(device: int = 0, what: int = 7):
        """pnx_preload: load the device code of the named routes (1 GFA text, 2 edges, 4 pass) before their first use"""
        rc = load().pnx_preload(C.c_int(device), C.c_uint32(what))
        if rc != PNX_OK:
            raise PnxError(rc, "pnx_preload failed")

    def set_csr_gfa(self, text: bytes, col_begin, col_end, is_walk, n_nodes, id_of_name=None, weights=None, exclude=None, upload_first=False,
                    edge_uv=None, edge_oo=None, name_off=None, name_len=None, link_off=None, find_links=None, find_names=None, name_prefix=None):
        """pnx_set_csr_gfa: the node ItemTable from the step columns of GFA text, tokenised on the device; with edge_uv / edge_oo
        (n_edges + 1 entries, [0] unused) the EDGE ItemTable of the same walks; name_off / name_len: segment names that are not
        numbers, looked up in a hash table on the device; link_off: the L lines parsed on the device (edge counts); find_links:
        True or a byte range (lo, hi) -- the library finds the L lines itself (PNX_LINKS_FIND)"""
        cb = np.ascontiguousarray(col_begin, dtype=np.uint64)
        ce = np.ascontiguousarray(col_end, dtype=np.uint64)
        wk = np.ascontiguousarray(is_walk, dtype=np.uint8)
        names = None if id_of_name is None else np.ascontiguousarray(id_of_name, dtype=np.uint32)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.uint32)
        ex = None if exclude is None else np.ascontiguousarray(exclude, dtype=np.uint8)
        g = PnxGfaSteps()
        if upload_first:
            self._ck(self._L.pnx_gfa_text_upload(self._h, text, len(text)))
            g.text, g.text_bytes = None, 0
        else:
            g.text, g.text_bytes = text, len(text)
        g.n_paths, g.n_nodes = len(cb), n_nodes
        g.col_begin, g.col_end, g.is_walk = _ptr(cb, C.c_uint64), _ptr(ce, C.c_uint64), _ptr(wk, C.c_uint8)
        g.id_of_name, g.n_names = _ptr(names, C.c_uint32), (0 if names is None else len(names))
        uv = None if edge_uv is None else np.ascontiguousarray(edge_uv, dtype=np.uint64)
        oo = None if edge_oo is None else np.ascontiguousarray(edge_oo, dtype=np.uint8)
        g.edge_uv, g.edge_oo, g.n_edges = _ptr(uv, C.c_uint64), _ptr(oo, C.c_uint8), (0 if uv is None else len(uv) - 1)
        no = None if name_off is None else np.ascontiguousarray(name_off, dtype=np.uint64)
        nl = None if name_len is None else np.ascontiguousarray(name_len, dtype=np.uint8)
        lo = None if link_off is None else np.ascontiguousarray(link_off, dtype=np.uint64)
        g.name_off, g.name_len = _ptr(no, C.c_uint64), _ptr(nl, C.c_uint8)
        g.link_off, g.n_links = _ptr(lo, C.c_uint64), (0 if lo is None else len(lo))
        _find_links(g, find_links, find_names, name_prefix)
        self._ck(self._L.pnx_set_csr_gfa(self._h, C.byref(g), _ptr(w, C.c_uint32), _ptr(ex, C.c_uint8)))
        self.n_items = int(self.info().n_items)

    def gfa_walks(self, text: bytes, col_begin, col_end, is_walk, n_nodes, id_of_name=None, name_off=None, name_len=None, link_off=None, find_links=None, find_names=None, name_prefix=None) -> np.ndarray:
        """pnx_gfa_walks: the walks of GFA text tokenised on the device and kept there for set_csr_cut(walk_node=None, ...);
        -> their n_paths + 1 offsets"""
        cb = np.ascontiguousarray(col_begin, dtype=np.uint64)
        ce = np.ascontiguousarray(col_end, dtype=np.uint64)
        wk = np.ascontiguousarray(is_walk, dtype=np.uint8)
        names = None if id_of_name is None else np.ascontiguousarray(id_of_name, dtype=np.uint32)
        g = PnxGfaSteps()
        g.text, g.text_bytes = text, len(text)
        g.n_paths, g.n_nodes = len(cb), n_nodes
        g.col_begin, g.col_end, g.is_walk = _ptr(cb, C.c_uint64), _ptr(ce, C.c_uint64), _ptr(wk, C.c_uint8)
        g.id_of_name, g.n_names = _ptr(names, C.c_uint32), (0 if names is None else len(names))
        no = None if name_off is None else np.ascontiguousarray(name_off, dtype=np.uint64)
        nl = None if name_len is None else np.ascontiguousarray(name_len, dtype=np.uint8)
        lo = None if link_off is None else np.ascontiguousarray(link_off, dtype=np.uint64)
        g.name_off, g.name_len = _ptr(no, C.c_uint64), _ptr(nl, C.c_uint8)
        g.link_off, g.n_links = _ptr(lo, C.c_uint64), (0 if lo is None else len(lo))
        _find_links(g, find_links, find_names, name_prefix)
        off = np.zeros(len(cb) + 1, dtype=np.uint64)
        self._ck(self._L.pnx_gfa_walks(self._h, C.byref(g), _ptr(off, C.c_uint64)))
        return off

    def set_csr_walks(self, n_nodes, weights=None, exclude=None, edge_uv=None, edge_oo=None, edges_from_links=False):
        """pnx_set_csr_walks: the walks gfa_walks left on the device become the resident graph (node table, or the edge table of
        the same paths) without the text being tokenised again"""
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.uint32)
        x = None if exclude is None else np.ascontiguousarray(exclude, dtype=np.uint8)
        uv = None if edge_uv is None else np.ascontiguousarray(edge_uv, dtype=np.uint64)
        oo = None if edge_oo is None else np.ascontiguousarray(edge_oo, dtype=np.uint8)
        self._ck(self._L.pnx_set_csr_walks(self._h, n_nodes, _ptr(w, C.c_uint32), _ptr(x, C.c_uint8), _ptr(uv, C.c_uint64), _ptr(oo, C.c_uint8),
                                           0xFFFFFFFF if edges_from_links else (0 if uv is None else len(uv) - 1)))
        self.n_items = int(self.info().n_items)

    def prepare(self):
        self._ck(self._L.pnx_prepare(self._h))

    def get_csr(self, want_weights=False):
        n = C.c_uint64(0)
        self._ck(self._L.pnx_get_csr(self._h, C.byref(n), None, None, None))
        info = self.info()
        items = np.zeros(max(n.value, 1), dtype=np.uint32)
        off = np.zeros(info.n_paths + 1, dtype=np.uint64)
        w = np.zeros(info.n_items + 1, dtype=np.uint32) if want_weights else None
        self._ck(self._L.pnx_get_csr(self._h, C.byref(n), _ptr(items, C.c_uint32), _ptr(off, C.c_uint64),
                                     _ptr(w, C.c_uint32)))
        return items[: n.value], off, w

    def get_weights(self) -> np.ndarray:
        """the resident weights (n_items+1 u32, caller ids) without the steps"""
        n = C.c_uint64(0)
        w = np.zeros(self.info().n_items + 1, dtype=np.uint32)
        self._ck(self._L.pnx_get_csr(self._h, C.byref(n), None, None, _ptr(w, C.c_uint32)))
        return w

    def set_order(self, path_idx, group_id, n_groups=None):
        pi = np.ascontiguousarray(path_idx, dtype=np.uint32)
        gi = np.ascontiguousarray(group_id, dtype=np.uint32)
        if n_groups is None:
            n_groups = int(gi[-1]) + 1 if len(gi) else 0
        self._ck(self._L.pnx_set_order(self._h, _ptr(pi, C.c_uint32), _ptr(gi, C.c_uint32), len(pi), n_groups))
        self.n_groups = n_groups

    # ---- hist ----
    def hist(self, want_countable=True):
        cnt = np.zeros(self.n_items + 1, dtype=np.uint32) if want_countable else None
        h = np.zeros(self.n_groups + 1, dtype=np.uint64)
        self._ck(self._L.pnx_hist(self._h, _ptr(cnt, C.c_uint32), _ptr(h, C.c_uint64)))
        return cnt, h

    def hist_async(self):
        self._ck(self._L.pnx_hist_async(self._h))

    def hist_fetch(self, want_countable=False):
        cnt = np.zeros(self.n_items + 1, dtype=np.uint32) if want_countable else None
        h = np.zeros(self.n_groups + 1, dtype=np.uint64)
        self._ck(self._L.pnx_hist_fetch(self._h, _ptr(cnt, C.c_uint32), _ptr(h, C.c_uint64)))
        return cnt, h

    def hist_device(self):
        dh, dc = C.c_void_p(), C.c_void_p()
        self._ck(self._L.pnx_hist_device(self._h, C.byref(dh), C.byref(dc)))
        return dh.value, dc.value

    def hist_enqueued(self) -> int:
        """device pointer of the (G+1) u64 counters of the pass enqueued last (no wait)"""
        d = C.c_void_p()
        self._ck(self._L.pnx_hist_enqueued(self._h, C.byref(d)))
        return d.value

    def hist_enqueued_on(self):
        """(device pointer of the counters of the pass enqueued last, the stream they are produced on)"""
        d, st = C.c_void_p(), C.c_void_p()
        self._ck(self._L.pnx_hist_enqueued_on(self._h, C.byref(d), C.byref(st)))
        return d.value, st.value

    def sync(self):
        self._ck(self._L.pnx_sync(self._h))

    def stream(self) -> int:
        return self._L.pnx_stream(self._h)

    # ---- growth ----
    def ordered_growth(self, cov_thr, quorum_tab, perms=None):
        """cov_thr: T u32; quorum_tab: T x G u32; perms: R x G u32 or None. -> (R, T, G) u64"""
        G = self.n_groups
        ct = np.ascontiguousarray(cov_thr, dtype=np.uint32)
        qt = np.ascontiguousarray(quorum_tab, dtype=np.uint32).reshape(len(ct), G)
        if perms is None:
            R, pp = 1, None
        else:
            pp = np.ascontiguousarray(perms, dtype=np.uint32).reshape(-1, G)
            R = pp.shape[0]
        out = np.zeros((R, len(ct), G), dtype=np.uint64)
        self._ck(self._L.pnx_ordered_growth(self._h, _ptr(pp, C.c_uint32), R, _ptr(ct, C.c_uint32),
                                            _ptr(qt, C.c_uint32), len(ct), _ptr(out, C.c_uint64)))
        return out

    def ordered_growth_async(self, cov_thr, quorum_tab, perms=None):
        G = self.n_groups
        ct = np.ascontiguousarray(cov_thr, dtype=np.uint32)
        qt = np.ascontiguousarray(quorum_tab, dtype=np.uint32).reshape(len(ct), G)
        pp = None if perms is None else np.ascontiguousarray(perms, dtype=np.uint32).reshape(-1, G)
        R = 1 if pp is None else pp.shape[0]
        self._ck(self._L.pnx_ordered_growth_async(self._h, _ptr(pp, C.c_uint32), R, _ptr(ct, C.c_uint32),
                                                  _ptr(qt, C.c_uint32), len(ct)))
        return (R, len(ct), G)

    def ordered_growth_fetch(self, shape):
        out = np.zeros(shape, dtype=np.uint64)
        self._ck(self._L.pnx_ordered_growth_fetch(self._h, _ptr(out, C.c_uint64)))
        return out

    def ordered_growth_enqueued(self) -> int:
        """device pointer of the R*T*G u64 result of the growth call enqueued last (no wait)"""
        d = C.c_void_p()
        self._ck(self._L.pnx_ordered_growth_enqueued(self._h, C.byref(d)))
        return d.value

    def ordered_growth_device(self) -> int:
        d = C.c_void_p()
        self._ck(self._L.pnx_ordered_growth_device(self._h, C.byref(d)))
        return d.value

    # ---- "next" rows: similarity / table ----
    def group_intersections(self) -> np.ndarray:
        """inter[a, b] = items (or bp, when weighted) shared by groups a and b; diagonal = path_lens
        (Similarity::set_table, similarity.rs:119-150)."""
        G = int(self.info().n_groups)
        out = np.zeros((G, G), dtype=np.uint64)
        self._ck(self._L.pnx_group_intersections(self._h, _ptr(out, C.c_uint64)))
        return out

    def group_intersections_device(self) -> int:
        d = C.c_void_p()
        self._ck(self._L.pnx_group_intersections_device(self._h, C.byref(d)))
        return d.value

    def presence(self) -> np.ndarray:
        """[G, row_words] u64 plain bit rows: bit i % 64 of word i // 64 = item i in the group."""
        G = int(self.info().n_groups)
        rw = int(self._L.pnx_presence_row_words(self._h))
        out = np.zeros((G, rw), dtype=np.uint64)
        if G and rw:
            self._ck(self._L.pnx_presence(self._h, _ptr(out, C.c_uint64)))
        return out

    # ---- multi-GPU (RCCL communicator owned by the context) ----
    @staticmethod
    def comm_unique_id() -> bytes:
        """128 bytes that rank 0 hands to the other processes (any transport) for comm_init"""
        L = load()
        buf = (C.c_uint8 * 128)()
        rc = L.pnx_comm_unique_id(buf)
        if rc != PNX_OK:
            raise PnxError(rc, "pnx_comm_unique_id failed (librccl.so not loadable?)")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._ck(self._L.pnx_comm_init(self._h, buf, rank, world))

    def comm_allreduce_u64(self, d_ptr: int, n: int):
        """in-place sum over the ranks of n u64 at device pointer d_ptr, enqueued on the context's stream"""
        self._ck(self._L.pnx_comm_allreduce_u64(self._h, C.c_void_p(d_ptr), n))

    def comm_barrier(self):
        self._ck(self._L.pnx_comm_barrier(self._h))

    def comm_free(self):
        self._ck(self._L.pnx_comm_free(self._h))

    def share_csr(self, src: "Context"):
        """read the graph resident in `src` (same device) without a copy; keep `src` open meanwhile"""
        self._ck(self._L.pnx_share_csr(self._h, src._h))
        self._csr_owner = src  # keeps the owner alive as long as this context
        self.n_items = src.n_items
        self.n_groups = 0

    def group_visit_counts(self, item_lo: int, item_hi: int) -> np.ndarray:
        """[G, item_hi - item_lo] u32: steps of each group's paths on the items lo..hi-1 (AbacusByGroup.v, dense)."""
        G = int(self.info().n_groups)
        out = np.zeros((G, max(item_hi - item_lo, 0)), dtype=np.uint32)
        buf = out if out.size else np.zeros(1, dtype=np.uint32)
        self._ck(self._L.pnx_group_visit_counts(self._h, item_lo, item_hi, _ptr(buf, C.c_uint32)))
        return out

    def exp2_exact(self, x) -> np.ndarray:
        """device restatement of the platform libm's exp2 (bit-exact by construction; test hook)"""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        self._ck(self._L.pnx_exp2_exact(self._h, _ptr(x, C.c_double), _ptr(y, C.c_double), x.size))
        return y

    # ---- measurement / tunables ----
    def log2_exact(self, x) -> np.ndarray:
        a = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(a)
        self._ck(self._L.pnx_log2_exact(self._h, _ptr(a, C.c_double), _ptr(y, C.c_double), len(a)))
        return y

    def growth_closed_form_async(self, hist, n, branch, cov_abs, quorum_rel):
        """pnx_growth_closed_form_async; hist None = the counters of the pass enqueued last"""
        h = None if hist is None else np.ascontiguousarray(hist, dtype=np.uint64)
        br = np.ascontiguousarray(branch, dtype=np.uint32)
        cv = np.ascontiguousarray(cov_abs, dtype=np.uint32)
        q = np.ascontiguousarray(quorum_rel, dtype=np.float64)
        self._ck(self._L.pnx_growth_closed_form_async(self._h, _ptr(h, C.c_uint64), n, len(br), _ptr(br, C.c_uint32), _ptr(cv, C.c_uint32),
                                                      _ptr(q, C.c_double)))
        return (len(br), n)

    def growth_closed_form_fetch(self, shape) -> np.ndarray:
        out = np.zeros(shape, dtype=np.float64)
        self._ck(self._L.pnx_growth_closed_form_fetch(self._h, _ptr(out, C.c_double)))
        return out

    def profile_enable(self, on=True):
        self._ck(self._L.pnx_profile_enable(self._h, int(on)))

    def profile_select(self, slots=None):
        """time only the given slot indices (None = all)"""
        mask = 0xFFFFFFFF if slots is None else sum(1 << int(k) for k in slots)
        self._ck(self._L.pnx_profile_select(self._h, mask))

    def profile_sample(self, every: int):
        """time only every `every`-th launch of the selected slots"""
        self._ck(self._L.pnx_profile_sample(self._h, int(every)))

    def profile_read(self):
        ms = (C.c_double * K_COUNT)()
        n = (C.c_uint64 * K_COUNT)()
        self._ck(self._L.pnx_profile_read(self._h, ms, n))
        return {KERNEL_SLOT_NAMES[i]: (float(ms[i]), int(n[i])) for i in range(K_COUNT)}

    def profile_reset(self):
        self._ck(self._L.pnx_profile_reset(self._h))

    def config(self, key, value):
        self._ck(self._L.pnx_config(self._h, key, int(value)))

    def info(self) -> PnxInfo:
        out = PnxInfo()
        lib_bytes = C.c_size_t(0)  # (the sized entry: this binding stays valid when the library's struct grows)
        self._ck(self._L.pnx_info_sized(self._h, C.byref(out), C.sizeof(out), C.byref(lib_bytes)))
        return out
