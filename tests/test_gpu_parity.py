"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the reference's
golden vectors.  Integer work -> bit-exact equality everywhere."""
import math
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from panacus_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _load_gfa(ctx, gfa, count, group_mode=orc.GROUP_PATHID, exclude=None):
    g = orc.Graph(gfa, index_edges=(count == orc.EDGE))
    pi, gi, names = g.path_order(group_mode)
    items, pre = g.item_table(count)
    n = g.n_items(count)
    w = g.node_lens if count == orc.BP else None
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=w, exclude=exclude)
    ctx.set_order(pi.astype(np.uint32), gi.astype(np.uint32), len(names))
    return g, pi, gi, names, items, pre, n, w


# ---------------------------------------------------------------------------------------------
# reference golden vectors through the HIP path
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("count,key", [(orc.NODE, "chrM_sample_node"), (orc.EDGE, "chrM_sample_edge"),
                                       (orc.BP, "chrM_sample_bp")])
def test_chrM_golden(ctx, golden, golden_dir, count, key):
    # src/graph_broker/abacus.rs:1487-1630
    _load_gfa(ctx, os.path.join(golden_dir, "chrM_test.gfa"), count, orc.GROUP_SAMPLE)
    cnt, h = ctx.hist()
    assert cnt.tolist() == golden[key]["countable"]
    assert h.tolist() == golden[key]["hist"]


def test_cdbg_golden(ctx, golden, golden_dir):
    # src/graph_broker/abacus.rs:1424-1435
    _load_gfa(ctx, os.path.join(golden_dir, "cdbg.gfa"), orc.NODE)
    cnt, h = ctx.hist()
    assert cnt.tolist() == golden["cdbg_node"]["countable"]


def test_t_groups_golden(ctx, golden, golden_dir):
    # tests/test_files/t_groups.hist.tsv
    _load_gfa(ctx, os.path.join(golden_dir, "t_groups.gfa"), orc.NODE)
    cnt, h = ctx.hist()
    assert h.tolist() == golden["t_groups_node_hist"]["hist"]


def test_ordered_growth_fixture_values(ctx, golden_dir):
    """Same expectations as tests/test_oracle_golden.py::test_ordered_growth_survey_values."""
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    g, pi, gi, names, items, pre, n, w = _load_gfa(ctx, os.path.join(golden_dir, "t_groups.gfa"), orc.NODE)
    G = len(names)
    cov = [coverage_abs(Threshold(ABSOLUTE, 1), G)] * 2
    qt = np.stack([quorum_table(Threshold(RELATIVE, 0.0), G), quorum_table(Threshold(RELATIVE, 0.5), G)])
    out = ctx.ordered_growth(cov, qt)
    assert out[0, 0].tolist() == [2, 5, 8, 9, 10, 10]
    assert out[0, 1].tolist() == [2, 5, 5, 5, 5, 0]
    _load_gfa(ctx, os.path.join(golden_dir, "t_groups.gfa"), orc.BP)
    out = ctx.ordered_growth(cov[:1], qt[:1])
    assert out[0, 0].tolist() == [9, 14, 38, 39, 50, 50]

    _load_gfa(ctx, os.path.join(golden_dir, "chrM_test.gfa"), orc.NODE, orc.GROUP_SAMPLE)
    cov = [1, 2, 1]
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), 4) for q in (0.0, 0.0, 0.5)])
    out = ctx.ordered_growth(cov, qt)
    assert out[0, 0].tolist() == [89, 106, 140, 154]
    assert out[0, 1].tolist() == [87, 101, 115, 115]
    assert out[0, 2].tolist() == [89, 106, 106, 120]
    _load_gfa(ctx, os.path.join(golden_dir, "chrM_test.gfa"), orc.BP, orc.GROUP_SAMPLE)
    out = ctx.ordered_growth([1], qt[:1])
    assert out[0, 0].tolist() == [16569, 17147, 17183, 17197]


# ---------------------------------------------------------------------------------------------
# synthetic generator: device CSR == CPU CSR
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,p", [(1, 1), (777, 5), (2048, 16), (2049, 17), (50_000, 33)])
def test_pansyn_device_equals_cpu(ctx, n, p):
    items, pre, lens = orc.pansyn(42, n, p)
    ctx.set_csr_pansyn(42, n, p, with_weights=True)
    d_items, d_off, d_w = ctx.get_csr(want_weights=True)
    assert d_off.tolist() == pre.tolist()
    assert np.array_equal(d_items, items.astype(np.uint32))
    assert np.array_equal(d_w[1:], lens[1:])


# ---------------------------------------------------------------------------------------------
# hist parity on synthetic graphs (monotone fast path), several shapes / groupings
# ---------------------------------------------------------------------------------------------
def _oracle_hist(items, pre, pi, gi, n, G, w=None, exclude=None):
    cov = orc.coverage(items, pre, pi, gi, n, exclude)
    return cov, orc.hist(cov, G, w)


@pytest.mark.parametrize("n,p,tile_blocks", [(5_000, 7, 1), (70_000, 32, 1), (70_000, 32, 2), (300_000, 64, 1)])
def test_hist_pansyn_vs_oracle(ctx, n, p, tile_blocks):
    from panacus_amd import capi
    items, pre, lens = orc.pansyn(42, n, p)
    ctx.config(capi.CFG_TILE_BLOCKS, tile_blocks)
    try:
        ctx.set_csr_pansyn(42, n, p, with_weights=False)
        # group = path
        pi = np.arange(p, dtype=np.uint64)
        ctx.set_order(pi, pi, p)
        cnt, h = ctx.hist()
        ocov, oh = _oracle_hist(items, pre, pi, pi, n, p)
        assert np.array_equal(cnt, ocov)
        assert np.array_equal(h, oh)
        assert ctx.info().n_general_paths == 0  # all paths took the tile route
        # group = sample (pairs of paths), reversed visiting order of the groups
        G = (p + 1) // 2
        order = np.arange(p, dtype=np.uint64)[::-1].copy()
        gid = np.repeat(np.arange(G, dtype=np.uint64), 2)[:p] if p % 2 == 0 else None
        if gid is not None:
            ctx.set_order(order, gid, G)
            cnt, h = ctx.hist()
            ocov, oh = _oracle_hist(items, pre, order, gid, n, G)
            assert np.array_equal(cnt, ocov)
            assert np.array_equal(h, oh)
    finally:
        ctx.config(capi.CFG_TILE_BLOCKS, 1)


def test_hist_bp_and_exclude(ctx):
    n, p = 40_000, 12
    items, pre, lens = orc.pansyn(7, n, p)
    rng = np.random.default_rng(1)
    excl = (rng.random(n + 1) < 0.1).astype(np.uint8)
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=lens, exclude=excl)
    pi = np.arange(p, dtype=np.uint64)
    ctx.set_order(pi, pi, p)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, pi, pi, n, p, lens, excl)
    assert np.array_equal(cnt, ocov)
    assert np.array_equal(h, oh)
    assert int(h.sum()) == int(lens[1:].sum())


def test_hist_subset_order_and_empty(ctx):
    n, p = 10_000, 9
    items, pre, lens = orc.pansyn(3, n, p)
    ctx.set_csr(items.astype(np.uint32), pre, n)
    # only some paths are visited (excluded / subset paths are simply absent from the order)
    pi = np.array([4, 0, 7], dtype=np.uint64)
    gi = np.array([0, 1, 2], dtype=np.uint64)
    ctx.set_order(pi, gi, 3)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, pi, gi, n, 3)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    # empty order: every item has coverage 0
    ctx.set_order(np.zeros(0, np.uint32), np.zeros(0, np.uint32), 0)
    cnt, h = ctx.hist()
    assert cnt[0] == 0xFFFFFFFF and not cnt[1:].any()
    assert h.tolist() == [n]


def test_hist_ragged_and_empty_paths(ctx):
    # empty paths, a single-step path, repeated steps, ids at tile borders
    n = 5000
    paths = [[], [1], [2047, 2048, 2049, 2048], [n, n, n], list(range(1, n + 1)), [], [4096, 1, 4095]]
    items = np.array([x for p in paths for x in p], dtype=np.uint64)
    pre = np.cumsum([0] + [len(p) for p in paths]).astype(np.uint64)
    ctx.set_csr(items.astype(np.uint32), pre, n)
    pi = np.arange(len(paths), dtype=np.uint64)
    ctx.set_order(pi, pi, len(paths))
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, pi, pi, n, len(paths))
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)


@pytest.fixture
def route(request, ctx):
    """PNX_CFG_COVER_VARIANT for one test: 3 = over path rows, the coverage kernel adds the histogram itself (the default), 30 =
    over path rows with the separate histogram kernel (PNX_CFG_HIST_IN_COVER 0), 2 = over the packed steps with the boundary
    index, the run index and the sort of shuffled paths (round 2's default, kept as a cross-check)"""
    from panacus_amd import capi
    ctx.config(capi.CFG_COVER_VARIANT, 3 if request.param == 30 else request.param)
    ctx.config(capi.CFG_HIST_IN_COVER, 0 if request.param == 30 else 1)
    yield request.param
    ctx.config(capi.CFG_COVER_VARIANT, 3)
    ctx.config(capi.CFG_HIST_IN_COVER, 1)


@pytest.fixture
def sort_shuffled(request, ctx):
    """PNX_CFG_SORT_SHUFFLED for one test (1 = default: shuffled paths are sorted by id when the graph is
    prepared; 0: they take the atomic scatter route), restored afterwards on the shared context"""
    from panacus_amd import capi
    ctx.config(capi.CFG_SORT_SHUFFLED, request.param)
    yield request.param
    ctx.config(capi.CFG_SORT_SHUFFLED, 1)


@pytest.mark.parametrize("route,sort_shuffled", [(3, 1), (30, 1), (2, 1), (2, 0)], indirect=True)
def test_hist_unsorted_paths_sorted_or_on_the_scatter_route(ctx, route, sort_shuffled):
    n, p = 30_000, 10
    items, pre, lens = orc.pansyn(11, n, p)
    rng = np.random.default_rng(5)
    items = items.copy()
    for k in (1, 4, 8):  # shuffle three paths completely
        seg = items[pre[k]:pre[k + 1]]
        rng.shuffle(seg)
    # one path that looks monotone at its ends and on most boundaries but has a late outlier
    seg = items[pre[2]:pre[2 + 1]]
    seg[len(seg) // 2] = seg[3]
    ctx.set_csr(items.astype(np.uint32), pre, n)
    pi = np.arange(p, dtype=np.uint64)
    gi = (pi // 2).astype(np.uint64)
    ctx.set_order(pi, gi, 5)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, pi, gi, n, 5)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    info = ctx.info()
    if route == 2:
        # the three shuffled paths: sorted once (tile route) or left to the atomic route; the one with a single outlier
        # is cut into runs either way
        assert (info.n_sorted_paths, info.n_scatter_paths) == ((3, 0) if sort_shuffled else (0, 3)) and info.n_run_paths == 1
        assert ctx.info().n_general_paths == (1 if sort_shuffled else 4)
    else:  # path rows: no routes, nothing sorted, nothing run again
        assert (info.n_sorted_paths, info.n_scatter_paths, info.n_run_paths) == (0, 0, 0) and info.n_rows > 0
    got, off, _ = ctx.get_csr()  # the caller's order, whatever the library did with its copy
    assert np.array_equal(got, items.astype(np.uint32)) and np.array_equal(off, pre)
    # second call (classification cached) gives the same answer
    cnt2, h2 = ctx.hist()
    assert np.array_equal(cnt2, ocov) and np.array_equal(h2, oh)


def test_hist_many_groups_uses_wide_counters(ctx):
    # > 255 groups -> 12-plane counters; items present in every group reach count G
    n, p = 3000, 300
    items, pre, lens = orc.pansyn(5, n, p)
    ctx.set_csr(items.astype(np.uint32), pre, n)
    pi = np.arange(p, dtype=np.uint64)
    ctx.set_order(pi, pi, p)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, pi, pi, n, p)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    assert cnt[1:].max() == p


def test_rejects_bad_input(ctx):
    from panacus_amd.capi import PnxError
    with pytest.raises(PnxError):
        ctx.set_csr(np.array([1, 2, 9], np.uint32), np.array([0, 3], np.uint64), 5)  # id 9 > n_items
    with pytest.raises(PnxError):
        ctx.set_csr(np.array([0], np.uint32), np.array([0, 1], np.uint64), 5)  # id 0 is reserved
    ctx.set_csr(np.array([1, 2], np.uint32), np.array([0, 2], np.uint64), 5)
    with pytest.raises(PnxError):
        ctx.set_order(np.array([0, 0], np.uint32), np.array([0, 2], np.uint32), 3)  # gap in group ids
    with pytest.raises(PnxError):
        ctx.set_order(np.array([3], np.uint32), np.array([0], np.uint32), 1)  # path index out of range


def test_rejects_bad_calls_of_the_later_entry_points():
    """every entry point fails with a message instead of crashing: wrong order of calls, null
    pointers, out-of-range sizes, unknown tunables"""
    import ctypes as C
    from panacus_amd import capi
    from panacus_amd.capi import PnxError
    with capi.Context(0) as c:
        L, h = c._L, c._h
        with pytest.raises(PnxError):   # nothing uploaded yet
            c.group_intersections()
        with pytest.raises(PnxError):
            c.hist_enqueued()
        with pytest.raises(PnxError):
            c.config(capi.CFG_COVER_SPLIT, 3)
        with pytest.raises(PnxError):
            c.config(capi.CFG_COVER_SKIP, 7)
        with pytest.raises(PnxError):
            c.config(capi.CFG_INDEX_BY_ENTRY, -1)
        with pytest.raises(PnxError):
            c.config(999, 1)
        assert L.pnx_presence(h, None) != 0
        assert L.pnx_quorum_sums(h, 0, 1, None, None, None, None, None) != 0
        out = C.POINTER(C.c_double)()
        assert L.pnx_quorum_sums_fetch(h, C.byref(out)) != 0      # nothing was enqueued
        z = np.zeros(4, dtype=np.float64)
        u = np.zeros(2, dtype=np.uint32)
        f64p, u32p = C.POINTER(C.c_double), C.POINTER(C.c_uint32)
        assert L.pnx_quorum_sums_async(h, 100000, 1, u.ctypes.data_as(u32p), z.ctypes.data_as(f64p),
                                       z.ctypes.data_as(f64p), z.ctypes.data_as(f64p)) != 0  # n too large
        assert b"pnx_quorum_sums" in L.pnx_last_error(h)
        # a context that failed a call is still usable
        c.set_csr(np.array([1, 2, 2, 3], np.uint32), np.array([0, 2, 4], np.uint64), 3)
        c.set_order(np.array([0, 1], np.uint32), np.array([0, 1], np.uint32), 2)
        assert c.group_intersections().tolist() == [[2, 1], [1, 2]]
        assert c.hist()[1].tolist() == [0, 2, 1]
    assert capi.load().pnx_hist(None, None, None) != 0  # NULL context


# ---------------------------------------------------------------------------------------------
# ordered / permuted growth parity
# ---------------------------------------------------------------------------------------------
def _oracle_growth(items, pre, n, G, path_groups, perm, cov_thr, q, w=None):
    """ordered growth for the group order `perm` (rank -> group id): relabel and rerun the oracle"""
    rank_of = np.empty(G, dtype=np.uint64)
    rank_of[perm] = np.arange(G, dtype=np.uint64)
    # visiting order: groups by rank, paths of a group contiguous
    pi, gi = [], []
    for rank, g in enumerate(perm):
        for p in np.nonzero(path_groups == g)[0]:
            pi.append(p)
            gi.append(rank)
    r, c = orc.by_group(items, pre, np.array(pi, np.uint64), np.array(gi, np.uint64), n)
    return orc.ordered_growth(r, c, G, (orc.ABSOLUTE, cov_thr), (orc.RELATIVE, q), w)


@pytest.mark.parametrize("weighted", [False, True])
def test_ordered_growth_vs_oracle(ctx, weighted):
    from panacus_amd.pansyn import random_orders
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    n, p = 20_000, 24
    items, pre, lens = orc.pansyn(9, n, p)
    w = lens if weighted else None
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=w)
    path_groups = (np.arange(p) // 2).astype(np.uint64)  # 12 groups of 2 paths
    G = 12
    pi = np.arange(p, dtype=np.uint64)
    ctx.set_order(pi, path_groups, G)
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5), (3, 0.3), (1, 1.0), (0, 0.1)]
    cov = [coverage_abs(Threshold(ABSOLUTE, c), G) for c, _ in pairs]
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), G) for _, q in pairs])
    perms = random_orders(42, 5, G)
    out = ctx.ordered_growth(cov, qt, perms)
    assert out.shape == (5, len(pairs), G)
    for r in range(5):
        for t, (c, q) in enumerate(pairs):
            exp = _oracle_growth(items, pre, n, G, path_groups, perms[r], c, q, w)
            assert out[r, t].tolist() == [int(x) for x in exp], (r, c, q)
    # identity order == reference ordered-histgrowth
    out1 = ctx.ordered_growth(cov, qt)
    for t, (c, q) in enumerate(pairs):
        exp = _oracle_growth(items, pre, n, G, path_groups, np.arange(G), c, q, w)
        assert out1[0, t].tolist() == [int(x) for x in exp]


def test_ordered_growth_wide_weights(ctx):
    """weights >= 2^16 take the u32 staging of the growth kernel; sums exceed 2^32"""
    from panacus_amd.pansyn import random_orders
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    n, p = 12_000, 20
    items, pre, lens = orc.pansyn(17, n, p)
    w = (lens.astype(np.uint64) * 60_001 + 70_000).clip(0, 0xFFFFFFFF).astype(np.uint32)
    w[0] = 0
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=w)
    pg = np.arange(p, dtype=np.uint64)
    ctx.set_order(pg, pg, p)
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5), (1, 0.25)]
    cov = [coverage_abs(Threshold(ABSOLUTE, c), p) for c, _ in pairs]
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), p) for _, q in pairs])
    perms = random_orders(3, 3, p)
    out = ctx.ordered_growth(cov, qt, perms)
    assert int(out.max()) > (1 << 32)
    for r in range(3):
        for t, (c, q) in enumerate(pairs):
            exp = _oracle_growth(items, pre, n, p, pg, perms[r], c, q, w)
            assert out[r, t].tolist() == [int(x) for x in exp], (r, c, q)


def test_ordered_growth_many_groups(ctx):
    from panacus_amd.pansyn import random_orders
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    n, p = 6000, 300  # > 255 groups: 12-plane quorum counters
    items, pre, lens = orc.pansyn(13, n, p)
    ctx.set_csr(items.astype(np.uint32), pre, n)
    pg = np.arange(p, dtype=np.uint64)
    ctx.set_order(pg, pg, p)
    pairs = [(1, 0.0), (1, 0.5)]
    cov = [coverage_abs(Threshold(ABSOLUTE, c), p) for c, _ in pairs]
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), p) for _, q in pairs])
    perms = random_orders(1, 2, p)
    out = ctx.ordered_growth(cov, qt, perms)
    for r in range(2):
        for t, (c, q) in enumerate(pairs):
            exp = _oracle_growth(items, pre, n, p, pg, perms[r], c, q)
            assert out[r, t].tolist() == [int(x) for x in exp]


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("p,groups_of", [(40, 1), (130, 1), (264, 1), (96, 2), (530, 1)])
def test_ordered_growth_two_ranks_per_step(ctx, p, groups_of, weighted, monkeypatch):
    """k_growth_fused<..., ALT>: tables whose every pair of ranks is d = (1, 0) (q = 0.5) take two ranks per ripple over the slack's
    planes, any other table one; full batches of 16 ranks + tail ranks, one and two quorum pairs per launch, quorum pairs with a
    coverage mask (folded into the sign masks of the slack), q = 0 pairs beside them; the same call with the short step switched
    off (PNX_GROWTH_STEP=1) gives the same numbers.  Rule: abacus.rs:1001-1010."""
    from panacus_amd.pansyn import random_orders
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    n = 9000
    items, pre, lens = orc.pansyn(100 + p, n, p)
    w = lens if weighted else None
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=w)
    G = p // groups_of
    pg = (np.arange(p) // groups_of).astype(np.uint64)
    ctx.set_order(np.arange(p, dtype=np.uint64), pg, G)
    perms = random_orders(5, 3, G)
    for pairs in ([(1, 0.0), (2, 0.0), (1, 0.5)],            # cfg4's: one table that alternates
                  [(2, 0.5), (1, 0.5), (1, 0.0)],            # two of them, one under a coverage mask
                  [(1, 0.5), (3, 0.3)],                      # one that alternates beside one that does not: single steps
                  [(2, 0.25), (1, 0.9), (1, 1.0), (3, 0.0)],  # none alternates; three quorum pairs = two launches
                  [(1, 0.5)],
                  # 2, 4, 5, 6 accumulators in one launch: 4, 8, 10, 12 registers per row in the reduce-scatter (pairs and lone ones)
                  [(1, 0.0), (2, 0.0)],
                  [(1, 0.0), (2, 0.0), (3, 0.0), (1, 0.5)],
                  [(1, 0.0), (2, 0.0), (3, 0.0), (1, 0.5), (2, 0.5)],
                  [(1, 0.0), (2, 0.0), (3, 0.0), (4, 0.0), (1, 0.5), (2, 0.5)],
                  [(1, 0.0), (2, 0.0), (3, 0.0), (5, 0.0), (1, 0.3), (2, 0.7)]):
        cov = [coverage_abs(Threshold(ABSOLUTE, c), G) for c, _ in pairs]
        qt = np.stack([quorum_table(Threshold(RELATIVE, q), G) for _, q in pairs])
        out = ctx.ordered_growth(cov, qt, perms)
        for r in range(len(perms)):
            for t, (c, q) in enumerate(pairs):
                exp = _oracle_growth(items, pre, n, G, pg, perms[r], c, q, w)
                assert out[r, t].tolist() == [int(x) for x in exp], (pairs, r, c, q)
        monkeypatch.setenv("PNX_GROWTH_STEP", "1")
        out1 = ctx.ordered_growth(cov, qt, perms)
        monkeypatch.delenv("PNX_GROWTH_STEP")
        assert np.array_equal(out, out1), pairs


@pytest.mark.parametrize("route,sort_shuffled", [(3, 1), (30, 1), (2, 1), (2, 0)], indirect=True)
def test_growth_after_scatter_route(ctx, route, sort_shuffled):
    """edge-like (unsorted) paths: sorted at preparation, or the presence matrix comes from the scatter route"""
    from panacus_amd.thresholds import RELATIVE, Threshold, quorum_table
    n, p = 9000, 8
    items, pre, lens = orc.pansyn(21, n, p)
    rng = np.random.default_rng(2)
    items = items.copy()
    for k in range(p):
        rng.shuffle(items[pre[k]:pre[k + 1]])
    ctx.set_csr(items.astype(np.uint32), pre, n)
    pg = np.arange(p, dtype=np.uint64)
    ctx.set_order(pg, pg, p)
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), p) for q in (0.0, 0.6)])
    out = ctx.ordered_growth([1, 2], qt)
    for t, (c, q) in enumerate([(1, 0.0), (2, 0.6)]):
        exp = _oracle_growth(items, pre, n, p, pg, np.arange(p), c, q)
        assert out[0, t].tolist() == [int(x) for x in exp]


# ---------------------------------------------------------------------------------------------
# size-independent properties at a larger size (no oracle run)
# ---------------------------------------------------------------------------------------------
def test_properties_large(ctx):
    from panacus_amd.thresholds import RELATIVE, Threshold, quorum_table
    n, p = 2_000_000, 64
    ctx.set_csr_pansyn(42, n, p, with_weights=False)
    pi = np.arange(p, dtype=np.uint32)
    ctx.set_order(pi, pi, p)
    cnt, h = ctx.hist()
    assert int(h.sum()) == n                                  # every item lands in exactly one bin
    assert np.array_equal(np.bincount(cnt[1:], minlength=p + 1).astype(np.uint64), h)
    info = ctx.info()
    # sum of coverages == number of distinct (item, path) incidences <= S
    assert int(cnt[1:].astype(np.uint64).sum()) <= info.n_steps
    cnt2, h2 = ctx.hist()                                     # idempotent
    assert np.array_equal(h, h2) and np.array_equal(cnt, cnt2)
    # order invariance of the histogram
    ctx.set_order(pi[::-1].copy(), pi, p)
    _, h3 = ctx.hist(want_countable=False)
    assert np.array_equal(h, h3)
    # growth at q=0, c=1: last value == #items covered at least once; curve is monotone
    ctx.set_order(pi, pi, p)
    out = ctx.ordered_growth([1], quorum_table(Threshold(RELATIVE, 0.0), p)[None, :])
    curve = out[0, 0]
    assert int(curve[-1]) == n - int(h[0])
    assert np.all(np.diff(curve.astype(np.int64)) >= 0)


# ---------------------------------------------------------------------------------------------
# every kernel variant / index setting gives the same exact answer
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant,coarse", [(v, c) for v in (0, 1, 2) for c in (1, 3, 8, 64)] + [(3, 8)])
def test_variants_agree_with_oracle(ctx, variant, coarse):
    from panacus_amd import capi
    n, p = 150_000, 20
    items, pre, lens = orc.pansyn(17, n, p)
    rng = np.random.default_rng(3)
    items = items.copy()
    rng.shuffle(items[pre[6]:pre[7]])           # one general path
    items[pre[9] + 1000] = items[pre[9] + 5]     # one late outlier in an otherwise sorted path
    ctx.config(capi.CFG_COVER_VARIANT, variant)
    ctx.config(capi.CFG_INDEX_COARSE, coarse)
    try:
        ctx.set_csr(items.astype(np.uint32), pre, n, weights=lens)
        pi = np.arange(p, dtype=np.uint64)
        gi = (pi // 2).astype(np.uint64)
        ctx.set_order(pi, gi, 10)
        cnt, h = ctx.hist()
        ocov, oh = _oracle_hist(items, pre, pi, gi, n, 10, lens)
        assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
        # long segments (> 1024 steps per tile and path): dense paths with every step doubled
        dense = np.repeat(np.arange(1, 6001, dtype=np.uint64), 2)
        items2 = np.concatenate([dense, dense[::-1]])
        pre2 = np.array([0, len(dense), 2 * len(dense)], dtype=np.uint64)
        ctx.set_csr(items2.astype(np.uint32), pre2, 6000)
        ctx.set_order(np.array([0, 1], np.uint32), np.array([0, 1], np.uint32), 2)
        cnt, h = ctx.hist()
        assert h.tolist() == [0, 0, 6000] and cnt[1:].tolist() == [2] * 6000
    finally:
        ctx.config(capi.CFG_COVER_VARIANT, 3)
        ctx.config(capi.CFG_INDEX_COARSE, 8)


def test_growth_arbitrary_quorum_table(ctx):
    """pnx_ordered_growth takes the quorum bound as a table; tables that do not rise by 0/1 per
    rank take the comparison kernel.  Reference rule (abacus.rs:1001-1010) evaluated directly."""
    n, p = 5000, 10
    items, pre, lens = orc.pansyn(31, n, p)
    ctx.set_csr(items.astype(np.uint32), pre, n)
    pg = np.arange(p, dtype=np.uint64)
    ctx.set_order(pg, pg, p)
    tabs = np.array([[0, 2, 2, 5, 1, 1, 7, 3, 3, 9], [1, 1, 1, 1, 1, 1, 1, 1, 1, 1], [0, 1, 1, 2, 2, 3, 3, 4, 4, 5]],
                    dtype=np.uint32)
    cov = [1, 2, 1]
    out = ctx.ordered_growth(cov, tabs)
    r, c = orc.by_group(items, pre, pg, pg, n)
    for t in range(3):
        exp = np.zeros(p, dtype=np.int64)
        for i in range(1, n + 1):
            grp = c[r[i]:r[i + 1]]
            if len(grp) < cov[t]:
                continue
            k = 0
            for j in range(int(grp[0]), p):
                if k < len(grp) - 1 and grp[k + 1] <= j:
                    k += 1
                if k + 1 >= tabs[t, int(grp[k])]:
                    exp[j] += 1
        assert out[0, t].tolist() == exp.tolist(), t


def _prefix_set_growth(X, w, deg_ok, q):
    """AbacusByGroup::calc_growth (abacus.rs:1001-1010) WITHOUT the walk over k: X[rank, item] is the
    presence of the item in the group visited at that rank.  For every rank j: cnt = how many of the
    ranks 0..j hold the item, last = the last of them; the item counts iff it has been seen and
    cnt >= ceil((last + 1) * q).  Vectorised over items with cumulative sums / maxima."""
    G, n = X.shape
    cnt = np.cumsum(X, axis=0, dtype=np.int64)
    ranks = np.arange(G, dtype=np.int64)[:, None]
    last = np.maximum.accumulate(np.where(X > 0, ranks, -1), axis=0)
    need = np.ceil((last + 1.0) * q).astype(np.int64)
    ok = (last >= 0) & (cnt >= need) & deg_ok[None, :]
    return (ok * w[None, :].astype(np.int64)).sum(axis=1)


@pytest.mark.parametrize("weighted", [False, True])
def test_ordered_growth_rule_from_the_presence_matrix(ctx, weighted):
    """The reference holds no number for ordered growth (tests/ordered_histgrowth.rs:14 checks a header),
    so the q > 0 rule -- the bound uses the rank of the LAST containing group, `c[k] + 1` -- is pinned
    by checks that do not restate its loop:
      (ii) for any (c, q) and any order, res[j] equals a brute force over PREFIX SETS taken from the
           device's own presence export (K6): counts and last-seen ranks from cumulative sums;
      (i)  q = 1: an item counts at rank j iff it is in EVERY group visited up to its last sighting, so
           res[G-1] = weight of the items whose groups are exactly the ranks 0..deg-1 -- in particular
           >= hist[G] * w (items of all groups) and, at rank 0, the size of the first group;
      (iii) q = 0: the curve is the running union."""
    from panacus_amd.pansyn import random_orders
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    n, p = 30_000, 28
    items, pre, lens = orc.pansyn(77, n, p)
    w_all = lens if weighted else None
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=w_all)
    path_groups = (np.arange(p) * 3 // 4 // 2).astype(np.uint64)  # uneven groups
    G = int(path_groups.max()) + 1
    ctx.set_order(np.arange(p, dtype=np.uint64), path_groups, G)
    cntv, h = ctx.hist()
    bits = ctx.presence()
    X0 = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, 1: n + 1].astype(np.int64)  # [group, item]
    assert np.array_equal(X0.sum(axis=0), cntv[1:].astype(np.int64))
    w = (lens[1:].astype(np.int64) if weighted else np.ones(n, dtype=np.int64))
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5), (3, 0.3), (1, 1.0), (2, 1.0), (1, 0.05), (G, 0.7)]
    cov = [coverage_abs(Threshold(ABSOLUTE, c), G) for c, _ in pairs]
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), G) for _, q in pairs])
    perms = np.concatenate([np.arange(G, dtype=np.uint32)[None, :], random_orders(5, 6, G)])
    out = ctx.ordered_growth(cov, qt, perms)
    deg = X0.sum(axis=0)
    for r in range(len(perms)):
        X = X0[perms[r].astype(np.int64)]          # rows in visiting order
        for t, (c, q) in enumerate(pairs):
            exp = _prefix_set_growth(X, w, deg >= max(1, c), q)
            assert out[r, t].tolist() == exp.tolist(), (r, c, q)
        # (i) q = 1 (pair index 4, c = 1)
        first_absent = np.where((X == 0).any(axis=0), (X == 0).argmax(axis=0), G)
        prefix_items = (deg >= 1) & (first_absent == deg)          # groups are exactly ranks 0..deg-1
        assert int(out[r, 4, -1]) == int(w[prefix_items].sum())
        assert int(out[r, 4, -1]) >= int(w[deg == G].sum())
        assert int(out[r, 4, 0]) == int(w[X[0] > 0].sum())
        # (iii) q = 0: running union
        seen = np.maximum.accumulate(X, axis=0)
        assert out[r, 0].tolist() == (seen * w[None, :]).sum(axis=1).tolist()


@pytest.mark.parametrize("route", [3, 30, 2], indirect=True)
def test_keyed_upload_reports_everything_in_the_callers_ids(ctx, tmp_path, route):
    """pnx_set_csr_keyed: edge steps numbered like the reference does (order of the L lines, here
    shuffled) + one key per edge (its canonical ends).  The library renumbers the edges internally on
    the device, so the paths take the tile route instead of the atomic scatter route -- and the
    coverage vector, the presence rows, the visit counts and the CSR read-back are all in the
    reference's edge ids, equal to the oracle's."""
    from panacus_amd import hostlib as hl
    src = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "40000", "--paths", "10", "--links", "-o", src])
    assert rc == 0, err
    lines = open(src).read().split("\n")
    links = [l for l in lines if l.startswith("L\t")]
    rng = np.random.default_rng(3)
    shuf = str(tmp_path / "shuf.gfa")
    with open(shuf, "w") as f:
        f.write("\n".join([l for l in lines if l and not l.startswith("L\t")] + [links[i] for i in rng.permutation(len(links))]) + "\n")
    g = orc.Graph(shuf, index_edges=True)
    hg = hl.GfaGraph(shuf, index_edges=True)
    pi, gi, names = g.path_order(orc.GROUP_SAMPLE)
    G = len(names)
    items, pre = g.item_table(orc.EDGE)
    hitems, hpre = hg.item_table(hl.EDGE)
    assert np.array_equal(items, hitems.astype(np.uint64))          # same (reference) edge ids on both sides
    n = g.n_edges
    keys = hg.edge_keys()
    excl = (rng.random(n + 1) < 0.03).astype(np.uint8)
    excl[0] = 0
    # without keys: the reference's ids do not follow the paths -> such paths are sorted at preparation
    ctx.set_csr(hitems, hpre, n, exclude=excl)
    ctx.set_order(pi.astype(np.uint32), gi.astype(np.uint32), G)
    cnt_plain, h_plain = ctx.hist()
    assert (ctx.info().n_sorted_paths > 0) == (route == 2) and ctx.info().n_scatter_paths == 0
    back, back_off, _ = ctx.get_csr()
    assert np.array_equal(back, hitems) and np.array_equal(back_off, hpre)
    # with keys: tile route, same answers
    ctx.set_csr(hitems, hpre, n, exclude=excl, item_key=keys)
    ctx.set_order(pi.astype(np.uint32), gi.astype(np.uint32), G)
    cnt, h = ctx.hist()
    info = ctx.info()
    assert info.n_scatter_paths == 0 and info.n_sorted_paths == 0
    ocov = orc.coverage(items, pre, pi, gi, n, excl)
    assert np.array_equal(cnt, ocov) and np.array_equal(cnt, cnt_plain)
    assert np.array_equal(h, orc.hist(ocov, G)) and np.array_equal(h, h_plain)
    # presence rows in the caller's ids
    bits = ctx.presence()
    got = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, : n + 1]
    r, c = orc.by_group(items, pre, pi, gi, n, excl)
    assert np.array_equal(got[:, 1:].T, orc.table_rows(r, c, G))
    # visit counts of a slice of caller ids
    lo, hi = 1000, 9000
    vc = ctx.group_visit_counts(lo, hi)
    path_group = np.full(len(pre) - 1, -1, dtype=np.int64)
    path_group[pi.astype(np.int64)] = gi.astype(np.int64)
    exp = np.zeros((G, hi - lo), dtype=np.int64)
    for p_ in range(len(pre) - 1):
        if path_group[p_] < 0:
            continue
        ids = items[int(pre[p_]):int(pre[p_ + 1])].astype(np.int64)
        ids = ids[(ids >= lo) & (ids < hi) & (excl[ids] == 0)]
        np.add.at(exp[path_group[p_]], ids - lo, 1)
    assert np.array_equal(vc.astype(np.int64), exp)
    # the CSR comes back as it was uploaded; a new exclusion list is taken in caller ids
    back, off, _ = ctx.get_csr()
    assert np.array_equal(back, hitems) and np.array_equal(off, hpre)
    excl2 = (rng.random(n + 1) < 0.2).astype(np.uint8)
    excl2[0] = 0
    ctx.set_exclude(excl2)
    cnt2, h2 = ctx.hist()
    assert np.array_equal(cnt2, orc.coverage(items, pre, pi, gi, n, excl2))
    # ordered growth and the intersections do not depend on numbering
    ctx.set_exclude(None)
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    out = ctx.ordered_growth([coverage_abs(Threshold(ABSOLUTE, 1), G)], quorum_table(Threshold(RELATIVE, 0.4), G)[None, :])
    r0, c0 = orc.by_group(items, pre, pi, gi, n)
    assert out[0, 0].tolist() == [int(x) for x in orc.ordered_growth(r0, c0, G, (orc.ABSOLUTE, 1), (orc.RELATIVE, 0.4))]
    exp_inter, _, _ = orc.similarity(r0, c0, G)
    assert np.array_equal(ctx.group_intersections(), exp_inter)
    # keys that already rise with the ids: nothing is renumbered, nothing changes
    ctx.set_csr(hitems, hpre, n, item_key=np.arange(n + 1, dtype=np.uint64))
    ctx.set_order(pi.astype(np.uint32), gi.astype(np.uint32), G)
    cnt3, _ = ctx.hist()
    assert np.array_equal(cnt3, orc.coverage(items, pre, pi, gi, n))


def test_native_rccl_communicator_single_rank(tmp_path):
    """pnx_comm_*: the library's own RCCL communicator (dlopen of librccl.so).  With one rank the sums
    are identities, but every call takes the real path: id, ncclCommInitRank, the all-reduce of flags +
    histogram behind every coverage pass (also behind the re-run of a pass whose paths are not
    tile-monotone), and pnx_comm_allreduce_u64 on the buffer of a growth call in flight."""
    from panacus_amd import capi
    from panacus_amd.distributed import native_comm_init
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    n, p = 50_000, 12
    items, pre, lens = orc.pansyn(61, n, p)
    jit = items.copy()
    for k in range(p):   # nearly monotone paths: a window reversed every 300 steps
        seg = jit[int(pre[k]):int(pre[k + 1])]
        m = (len(seg) // 300) * 300
        v = seg[:m].reshape(-1, 300)
        v[:, :40] = v[:, :40][:, ::-1].copy()
    pi = np.arange(p, dtype=np.uint64)
    ocov = orc.coverage(jit, pre, pi, pi, n)
    oh = orc.hist(ocov, p)
    with capi.Context(0) as c:
        with pytest.raises(capi.PnxError):
            c.comm_allreduce_u64(0, 4)                       # no communicator yet
        with pytest.raises(capi.PnxError):
            c.comm_init(bytes(128), 3, 2)                    # rank outside the world
        uid = native_comm_init(c, 0, 1, str(tmp_path / "comm.id"))
        assert len(uid) == 128 and not os.path.exists(str(tmp_path / "comm.id"))   # removed once every rank holds the communicator
        with pytest.raises(capi.PnxError):
            c.comm_init(uid, 0, 1)                           # one communicator per context
        c.set_csr(jit.astype(np.uint32), pre, n)
        c.set_order(pi, pi, p)
        cnt, h = c.hist()                                    # over path rows: one pass, one collective
        assert c.info().n_reruns == 0 and np.array_equal(cnt, ocov) and np.array_equal(h, oh)
        c.config(capi.CFG_COVER_VARIANT, 2)
        c.set_csr(jit.astype(np.uint32), pre, n)
        c.set_order(pi, pi, p)
        cnt, h = c.hist()                                    # over the steps: first pass classifies, is re-run: two collectives
        assert c.info().n_reruns >= 1 and c.info().n_run_paths == p
        assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
        c.hist_async()
        c.hist_async()
        assert np.array_equal(c.hist_fetch()[1], oh) and np.array_equal(c.hist_fetch()[1], oh)
        cov = [coverage_abs(Threshold(ABSOLUTE, 1), p)]
        qt = quorum_table(Threshold(RELATIVE, 0.3), p)[None, :]
        shape = c.ordered_growth_async(cov, qt)
        c.comm_allreduce_u64(c.ordered_growth_enqueued(), int(np.prod(shape)))
        out = c.ordered_growth_fetch(shape)
        r, cc = orc.by_group(jit, pre, pi, pi, n)
        assert out[0, 0].tolist() == [int(x) for x in orc.ordered_growth(r, cc, p, (orc.ABSOLUTE, 1), (orc.RELATIVE, 0.3))]
        c.config(capi.CFG_COMM_REDUCE_HIST, 0)
        assert np.array_equal(c.hist()[1], oh)
        c.comm_free()
        assert np.array_equal(c.hist()[1], oh)


def test_two_passes_in_flight(ctx):
    """pnx_hist_async may be called twice before the first result is fetched; results come back
    oldest first and are identical; a third enqueue is refused; a violation found in an
    in-flight pass is repaired for both."""
    from panacus_amd import capi
    n, p = 60_000, 12
    items, pre, lens = orc.pansyn(23, n, p)
    pi = np.arange(p, dtype=np.uint64)
    ocov, oh = _oracle_hist(items, pre, pi, pi, n, p)
    ctx.set_csr(items.astype(np.uint32), pre, n)
    ctx.set_order(pi, pi, p)
    ctx.hist_async()
    ctx.hist_async()
    with pytest.raises(capi.PnxError):
        ctx.hist_async()
    _, h1 = ctx.hist_fetch()
    cnt, h2 = ctx.hist_fetch(want_countable=True)
    assert np.array_equal(h1, oh) and np.array_equal(h2, oh) and np.array_equal(cnt, ocov)
    # unsorted path discovered while two passes are in flight
    items2 = items.copy()
    np.random.default_rng(0).shuffle(items2[pre[3]:pre[4]])
    items2[pre[7] + 500] = items2[pre[7] + 3]
    ocov2, oh2 = _oracle_hist(items2, pre, pi, pi, n, p)
    ctx.set_csr(items2.astype(np.uint32), pre, n)
    ctx.set_order(pi, pi, p)
    ctx.hist_async()
    ctx.hist_async()
    _, h1 = ctx.hist_fetch()
    cnt, h2 = ctx.hist_fetch(want_countable=True)
    assert np.array_equal(h1, oh2) and np.array_equal(h2, oh2) and np.array_equal(cnt, ocov2)
    # blocking call after an async one returns its own (newest) pass
    ctx.hist_async()
    cnt, h = ctx.hist()
    assert np.array_equal(h, oh2) and np.array_equal(cnt, ocov2)
    # PNX_CFG_MAX_IN_FLIGHT: four passes (each with its own coverage vector and counters), oldest first, a fifth refused
    ctx.config(capi.CFG_MAX_IN_FLIGHT, 4)
    try:
        orders = [pi, pi[::-1].copy(), np.roll(pi, 3), pi]
        for k in range(4):
            ctx.hist_async()
        with pytest.raises(capi.PnxError):
            ctx.hist_async()
        with pytest.raises(capi.PnxError):
            ctx.config(capi.CFG_MAX_IN_FLIGHT, 2)   # not while passes are in flight
        for k in range(4):
            cnt, h = ctx.hist_fetch(want_countable=True)
            assert np.array_equal(h, oh2) and np.array_equal(cnt, ocov2), k
        with pytest.raises(capi.PnxError):
            ctx.config(capi.CFG_MAX_IN_FLIGHT, 5)
    finally:
        ctx.sync()
        ctx.config(capi.CFG_MAX_IN_FLIGHT, 2)


def test_one_shot_context_makes_its_pass_streams_when_passes_overlap():
    """pnx_init_flags(PNX_INIT_ONE_SHOT): no phase streams up front; synchronous calls run on one stream, the first pass enqueued
    behind one in flight makes the streams and drains the single-stream pass first -- same numbers every way"""
    from panacus_amd import capi
    n, p = 50_000, 10
    items, pre, lens = orc.pansyn(29, n, p)
    pi = np.arange(p, dtype=np.uint64)
    ocov, oh = _oracle_hist(items, pre, pi, pi, n, p)
    with capi.Context(one_shot=True) as c:
        c.set_csr(items.astype(np.uint32), pre, n)
        c.set_order(pi, pi, p)
        for _ in range(3):                       # the one-shot route, then the rows: one stream each time
            cnt, h = c.hist()
            assert np.array_equal(h, oh) and np.array_equal(cnt, ocov)
        c.config(capi.CFG_MAX_IN_FLIGHT, 4)
        for rounds in range(3):
            for _ in range(4):
                c.hist_async()
            for _ in range(4):
                cnt, h = c.hist_fetch(want_countable=True)
                assert np.array_equal(h, oh) and np.array_equal(cnt, ocov)
        cnt, h = c.hist()
        assert np.array_equal(h, oh) and np.array_equal(cnt, ocov)


# ---------------------------------------------------------------------------------------------
# run route: nearly monotone paths (local back-steps across tile borders) stay off the atomics
# ---------------------------------------------------------------------------------------------
def _jitter(items, pre, paths, rng, width=40, every=300):
    """reverse short windows of a sorted path: ids step back locally, also across tile borders"""
    items = items.copy()
    for k in paths:
        seg = items[pre[k]:pre[k + 1]]
        for s in range(0, len(seg) - width, every):
            seg[s:s + width] = seg[s:s + width][::-1].copy()
    return items


@pytest.mark.parametrize("route,sort_shuffled", [(3, 1), (30, 1), (2, 1), (2, 0)], indirect=True)
def test_run_route_near_monotone_paths(ctx, route, sort_shuffled):
    from panacus_amd.thresholds import RELATIVE, Threshold, quorum_table
    n, p = 120_000, 16
    items, pre, lens = orc.pansyn(41, n, p)
    rng = np.random.default_rng(9)
    items = _jitter(items, pre, [0, 3, 4, 9, 15], rng)
    rng.shuffle(items[pre[6]:pre[7]])  # one path with random ids: scatter route
    reruns = ctx.info().n_reruns
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=lens)
    pi = np.arange(p, dtype=np.uint64)
    gi = (pi // 2).astype(np.uint64)
    ctx.set_order(pi, gi, 8)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, pi, gi, n, 8, lens)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    info = ctx.info()
    if route == 2:
        assert info.n_run_paths == 5 and info.n_runs > 0
        assert (info.n_sorted_paths, info.n_scatter_paths) == ((1, 0) if sort_shuffled else (0, 1))
    else:
        assert (info.n_run_paths, info.n_runs, info.n_sorted_paths, info.n_scatter_paths, info.n_reruns - reruns) == (0, 0, 0, 0, 0)
    # a different visiting order / grouping re-sorts the runs
    order = pi[::-1].copy()
    g2 = np.arange(p, dtype=np.uint64)
    ctx.set_order(order, g2, p)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, order, g2, n, p, lens)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    # only some paths visited (runs of unvisited paths are ignored)
    sub = np.array([3, 9, 1], dtype=np.uint64)
    ctx.set_order(sub, np.arange(3, dtype=np.uint64), 3)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, sub, np.arange(3, dtype=np.uint64), n, 3, lens)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    # growth on top of the run route (presence rows come out of the same bitmaps)
    ctx.set_order(pi, gi, 8)
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), 8) for q in (0.0, 0.5)])
    out = ctx.ordered_growth([1, 2], qt)
    pg = gi
    for t, (c, q) in enumerate([(1, 0.0), (2, 0.5)]):
        exp = _oracle_growth(items, pre, n, 8, pg, np.arange(8), c, q, lens)
        assert out[0, t].tolist() == [int(x) for x in exp]


@pytest.mark.parametrize("route,tile_blocks", [(3, 1), (30, 1), (2, 1), (2, 2)], indirect=["route"])
def test_run_route_all_paths_and_rebuild(ctx, route, tile_blocks):
    from panacus_amd import capi
    n, p = 50_000, 9
    items, pre, lens = orc.pansyn(43, n, p)
    items = _jitter(items, pre, range(p), np.random.default_rng(1), width=25, every=120)
    ctx.config(capi.CFG_TILE_BLOCKS, tile_blocks)
    ctx.config(capi.CFG_CACHE_INDEX, 0)  # index (and run index) rebuilt in every call
    try:
        ctx.set_csr(items.astype(np.uint32), pre, n)
        pi = np.arange(p, dtype=np.uint64)
        ctx.set_order(pi, pi, p)
        ocov, oh = _oracle_hist(items, pre, pi, pi, n, p)
        for _ in range(3):
            cnt, h = ctx.hist()
            assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
        assert ctx.info().n_run_paths == (p if route == 2 else 0) and ctx.info().n_scatter_paths == 0
        ctx.hist_async()
        ctx.hist_async()
        _, h1 = ctx.hist_fetch()
        _, h2 = ctx.hist_fetch()
        assert np.array_equal(h1, oh) and np.array_equal(h2, oh)
    finally:
        ctx.config(capi.CFG_TILE_BLOCKS, 1)
        ctx.config(capi.CFG_CACHE_INDEX, 1)


def test_many_groups_wide_paths(ctx):
    """G = 5000 groups: 16-plane coverage counters, histogram bins beyond the LDS limit (global
    atomics), growth refuses politely when its LDS accumulators cannot hold G."""
    from panacus_amd import capi
    n, p = 1500, 5000
    items, pre, lens = orc.pansyn(77, n, p)
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=lens)
    pi = np.arange(p, dtype=np.uint64)
    ctx.set_order(pi, pi, p)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, pi, pi, n, p, lens)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    assert cnt[1:].max() > 4096
    qt = np.zeros((4, p), dtype=np.uint32)
    with pytest.raises(capi.PnxError):
        ctx.ordered_growth([1, 1, 1, 1], qt)  # 4 x 5000 x 8 B + weight planes > 150 KiB of LDS
    out = ctx.ordered_growth([1], qt[:1])  # one pair fits
    r, c = orc.by_group(items, pre, pi, pi, n)
    exp = orc.ordered_growth(r, c, p, (orc.ABSOLUTE, 1), (orc.RELATIVE, 0.0), lens)
    assert out[0, 0].tolist() == [int(x) for x in exp]


def test_edge_paths_of_sorted_links_take_the_tile_route(ctx, tmp_path):
    """edge ids follow the L-line order; when links are listed by source node (the usual case)
    the edge ids along a sorted path are sorted too, so edge counting stays off the atomics"""
    from panacus_amd import hostlib as hl
    path = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "40000", "--paths", "10", "--links", "-o", path])
    assert rc == 0, err
    g = hl.GfaGraph(path, index_edges=True)
    items, pre = g.item_table(hl.EDGE)
    pi, gi, names = g.path_order()
    ctx.set_csr(items, pre, g.n_edges)
    ctx.set_order(pi, gi, len(names))
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items.astype(np.uint64), pre, pi.astype(np.uint64), gi.astype(np.uint64), g.n_edges, len(names))
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    info = ctx.info()
    assert info.n_scatter_paths == 0


# ---------------------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties (the oracle is too slow / too big here)
# ---------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------
# "next" rows: similarity intersections (K5) and the presence export behind `table` (K6)
# ---------------------------------------------------------------------------------------------
def _oracle_pairs(items, pre, n, path_groups, G, w=None, exclude=None):
    pi = np.arange(len(pre) - 1, dtype=np.uint64)
    r, c = orc.by_group(items, pre, pi, np.asarray(path_groups, dtype=np.uint64), n, exclude)
    return r, c


@pytest.mark.parametrize("count", [orc.NODE, orc.BP, orc.EDGE])
def test_group_intersections_chrM(ctx, golden_dir, count):
    g, pi, gi, names, items, pre, n, w = _load_gfa(ctx, os.path.join(golden_dir, "chrM_test.gfa"), count,
                                                   orc.GROUP_SAMPLE)
    r, c = orc.by_group(items, pre, pi, gi, n)
    exp, lens, _ = orc.similarity(r, c, len(names), node_lens=w)
    got = ctx.group_intersections()
    assert got.tolist() == exp.tolist()
    assert np.diag(got).tolist() == lens.tolist()


@pytest.fixture
def pairs_variant(request, ctx):
    """PNX_CFG_PAIRS_VARIANT for one test: 1 = int8 MFMA [default], 0 = AND + popcount on the vector ALUs"""
    from panacus_amd import capi
    ctx.config(capi.CFG_PAIRS_VARIANT, request.param)
    yield request.param
    ctx.config(capi.CFG_PAIRS_VARIANT, 1)


@pytest.mark.parametrize("pairs_variant", [1, 0], indirect=True)
@pytest.mark.parametrize("n,p,paths_per_group,weighted", [(5_000, 7, 1, False), (70_000, 70, 1, False),
                                                          (70_000, 70, 1, True), (40_000, 130, 1, False),
                                                          (30_000, 48, 2, True), (20_000, 300, 1, True)])
def test_group_intersections_vs_oracle(ctx, n, p, paths_per_group, weighted, pairs_variant):
    items, pre, lens = orc.pansyn(21, n, p)
    w = lens if weighted else None
    excl = np.zeros(n + 1, dtype=np.uint8)
    excl[5::97] = 1
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=w, exclude=excl)
    pg = (np.arange(p) // paths_per_group).astype(np.uint64)
    G = int(pg.max()) + 1
    ctx.set_order(np.arange(p, dtype=np.uint64), pg, G)
    r, c = _oracle_pairs(items, pre, n, pg, G, exclude=excl)
    exp, _, _ = orc.similarity(r, c, G, node_lens=w)
    got = ctx.group_intersections()
    assert (got == exp).all()
    # repeatable, and a new order gives a new matrix
    assert (ctx.group_intersections() == exp).all()
    ctx.set_order(np.arange(p - 1, dtype=np.uint64), pg[: p - 1], int(pg[: p - 1].max()) + 1)
    G2 = int(pg[: p - 1].max()) + 1
    r2, c2 = orc.by_group(items, pre, np.arange(p - 1, dtype=np.uint64), pg[: p - 1], n, excl)
    exp2, _, _ = orc.similarity(r2, c2, G2, node_lens=w)
    assert (ctx.group_intersections() == exp2).all()


@pytest.mark.parametrize("pairs_variant", [1, 0], indirect=True)
def test_group_intersections_wide_weights(ctx, pairs_variant):
    """weights above 2^16 use the high accumulator planes (vector kernel) / a second launch for digits 3, 4 (MFMA)"""
    n, p = 9_000, 9
    items, pre, lens = orc.pansyn(5, n, p)
    w = lens.copy()
    w[1::3] = w[1::3] * 70_001 + 1_000_000  # up to ~3.5e9 < 2^32
    w = np.minimum(w.astype(np.uint64), 0xFFFFFFFF).astype(np.uint32)
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=w)
    pg = np.arange(p, dtype=np.uint64)
    ctx.set_order(pg, pg, p)
    r, c = _oracle_pairs(items, pre, n, pg, p)
    exp, _, _ = orc.similarity(r, c, p, node_lens=w)
    assert (ctx.group_intersections() == exp).all()


@pytest.mark.parametrize("pairs_variant", [1, 0], indirect=True)
def test_set_weights_replaces_everything_derived_from_the_weights(ctx, pairs_variant):
    """pnx_set_weights after the weight planes (ordered growth) and digits (similarity) of the old weights were built:
    both must follow the new weights, also when those need more digits"""
    from panacus_amd.thresholds import RELATIVE, Threshold, quorum_table
    n, p = 12_000, 10
    items, pre, lens = orc.pansyn(31, n, p)
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=lens)
    pg = np.arange(p, dtype=np.uint64)
    ctx.set_order(pg, pg, p)
    qt = np.stack([quorum_table(Threshold(RELATIVE, 0.0), p)])
    r, c = _oracle_pairs(items, pre, n, pg, p)
    for w in (lens, (lens.astype(np.uint64) * 977 + 5).astype(np.uint32), np.minimum(lens, 3).astype(np.uint32)):
        if w is not lens:
            ctx.set_weights(w)
        exp, _, _ = orc.similarity(r, c, p, node_lens=w)
        assert (ctx.group_intersections() == exp).all()
        out = ctx.ordered_growth([1], qt)
        g = orc.ordered_growth(r, c, p, (orc.ABSOLUTE, 1), (orc.RELATIVE, 0.0), w)
        assert out[0, 0].tolist() == [int(x) for x in g]
        _, h = ctx.hist()
        assert int(h[1:].sum()) == int(w[1:][np.bincount(items.astype(np.int64), minlength=n + 1)[1:] > 0].astype(np.uint64).sum())


@pytest.mark.parametrize("pairs_variant", [1, 0], indirect=True)
def test_group_intersections_equal_the_matrix_product_of_the_presence_export(ctx, pairs_variant):
    """inter = B diag(w) B^T computed by numpy from the device's own presence export (K6): a check of K5 that shares
    nothing with the oracle's walk over (r, c) slices (`similarity` has no golden output upstream)"""
    n, p = 25_000, 37
    items, pre, lens = orc.pansyn(13, n, p)
    excl = np.zeros(n + 1, dtype=np.uint8)
    excl[3::41] = 1
    pg = (np.arange(p) // 2).astype(np.uint64)
    G = int(pg.max()) + 1
    for w in (None, lens):
        ctx.set_csr(items.astype(np.uint32), pre, n, weights=w, exclude=excl)
        ctx.set_order(np.arange(p, dtype=np.uint64), pg, G)
        got = ctx.group_intersections()
        B = np.unpackbits(ctx.presence().view(np.uint8), axis=1, bitorder="little")[:, : n + 1].astype(np.uint64)
        ww = np.ones(n + 1, dtype=np.uint64) if w is None else w.astype(np.uint64)
        assert np.array_equal(got, (B * ww) @ B.T)


def test_presence_export_matches_by_group(ctx):
    n, p = 10_000, 19
    items, pre, _ = orc.pansyn(8, n, p)
    ctx.set_csr(items.astype(np.uint32), pre, n)
    pg = (np.arange(p) // 2).astype(np.uint64)
    G = int(pg.max()) + 1
    ctx.set_order(np.arange(p, dtype=np.uint64), pg, G)
    bits = ctx.presence()
    assert bits.shape[0] == G and bits.shape[1] * 64 >= n + 1
    r, c = _oracle_pairs(items, pre, n, pg, G)
    exp = orc.table_rows(r, c, G)  # [n, G] 0/1
    got = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, : n + 1]
    assert got[:, 0].sum() == 0
    assert (got[:, 1:].T == exp).all()
    assert np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, n + 1:].sum() == 0


def test_group_visit_counts_match_by_group_values(ctx):
    """pnx_group_visit_counts == AbacusByGroup.v (the pointer game of abacus.rs:901-986, restated
    literally in the oracle), in item slices, with excluded items, unordered and empty paths."""
    n, p = 6000, 13
    items, pre, _ = orc.pansyn(21, n, p)
    # make path 3 empty and leave path 5 out of the order
    keep = np.ones(len(items), dtype=bool)
    keep[pre[3]:pre[4]] = False
    lens = np.diff(pre).astype(np.int64)
    lens[3] = 0
    items, pre = items[keep], np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    rng = np.random.default_rng(4)
    excl = (rng.random(n + 1) < 0.1).astype(np.uint8)
    ctx.set_csr(items.astype(np.uint32), pre, n, exclude=excl)
    paths = np.array([q for q in range(p) if q != 5], dtype=np.uint64)
    pg = (np.arange(len(paths)) // 3).astype(np.uint64)
    G = int(pg.max()) + 1
    ctx.set_order(paths, pg, G)
    r, c, v = orc.by_group_values(items, pre, paths, pg, n, excl)
    dense = np.zeros((G, n + 1), dtype=np.uint32)
    for i in range(1, n + 1):
        dense[c[r[i]:r[i + 1]].astype(np.int64), i] = v[r[i]:r[i + 1]]
    assert int(v.max()) > 1
    for lo, hi in ((0, n + 1), (1, 2), (17, 4000), (n, n + 1), (5, 5)):
        got = ctx.group_visit_counts(lo, hi)
        assert got.shape == (G, hi - lo) and np.array_equal(got, dense[:, lo:hi]), (lo, hi)
    # brute force: steps per (group, item)
    brute = np.zeros((G, n + 1), dtype=np.uint32)
    for q, g_ in zip(paths, pg):
        np.add.at(brute[int(g_)], items[pre[int(q)]:pre[int(q) + 1]].astype(np.int64), 1)
    brute[:, excl.astype(bool)] = 0
    assert np.array_equal(brute, dense)
    from panacus_amd import capi
    with pytest.raises(capi.PnxError):
        ctx.group_visit_counts(3, n + 3)


def test_two_contexts_share_one_resident_graph():
    """pnx_share_csr: a second context reads the first one's ItemTable in place -- own order, own
    stream, own results; passes of the two are enqueued interleaved"""
    from panacus_amd import capi
    n, p = 40_000, 24
    items, pre, lens = orc.pansyn(19, n, p)
    rng = np.random.default_rng(3)
    excl = (rng.random(n + 1) < 0.05).astype(np.uint8)
    a, b = capi.Context(0), capi.Context(0)
    try:
        with pytest.raises(capi.PnxError):
            b.share_csr(a)                      # nothing resident yet
        a.set_csr(items.astype(np.uint32), pre, n, weights=lens, exclude=excl)
        with pytest.raises(capi.PnxError):
            a.share_csr(a)
        b.share_csr(a)
        with pytest.raises(capi.PnxError):
            capi.Context(0).share_csr(b)        # a borrower cannot lend
        pa = np.arange(p, dtype=np.uint64)
        ga = pa // 2
        pb = pa[::-1].copy()
        gb = np.arange(p, dtype=np.uint64) // 3
        a.set_order(pa, ga, int(ga.max()) + 1)
        b.set_order(pb, gb, int(gb.max()) + 1)
        for c in (a, b, a, b):
            c.hist_async()
        outs = {"a": [], "b": []}
        for name, c in (("a", a), ("b", b), ("a", a), ("b", b)):
            outs[name].append(c.hist_fetch(want_countable=True))
        for name, (pi, gi) in (("a", (pa, ga)), ("b", (pb, gb))):
            G = int(gi.max()) + 1
            cov = orc.coverage(items, pre, pi, gi, n, excl)
            h = orc.hist(cov, G, lens)
            for cnt, hh in outs[name]:
                assert np.array_equal(cnt, cov) and np.array_equal(hh, h), name
        # the borrower can take a graph of its own afterwards; the lender is untouched
        items2, pre2, _ = orc.pansyn(20, 5000, 6)
        b.set_csr(items2.astype(np.uint32), pre2, 5000)
        o = np.arange(6, dtype=np.uint64)
        b.set_order(o, o, 6)
        cnt2, h2 = b.hist()
        assert np.array_equal(cnt2, orc.coverage(items2, pre2, o, o, 5000))
        cnt, hh = a.hist()
        assert np.array_equal(hh, orc.hist(orc.coverage(items, pre, pa, ga, n, excl), int(ga.max()) + 1, lens))
    finally:
        b.close()
        a.close()


def test_bench_contract_line_small(tmp_path):
    """bench.py end to end on small shapes: one JSON line with the contract's fields plus the
    permuted_growth / shape_10Mx1k / cpu_baseline blocks; the one-shot route (forced: the small shape would take the
    rows), the default choice, the forced single-rank RCCL path with either carrier -- all agree on the checks"""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs, pgs = [], []
    small = ["--nodes", "200000", "--paths", "64", "--steps", "12", "--warmup", "2", "--cpu-passes", "1",
             "--pg-nodes", "150000", "--pg-paths", "40", "--pg-orders", "10", "--pg-reps", "2",
             "--k1-nodes", "150000", "--k1-paths", "96", "--k1-steps", "4", "--resident-steps", "8", "--strong-steps", "6", "--strayed-steps", "6"]
    # FORCE_DIST without RANK / WORLD_SIZE in the environment: bench.py launches its rank(s) itself (the --gpus N path)
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for extra, env in ((["--cover-route", "1"], {}), (["--no-cpu-baseline", "--no-shape-1k", "--no-pmc"], {}),
                       (["--no-cpu-baseline", "--no-shape-1k", "--no-pmc", "--cover-route", "1"], {"PANACUS_BENCH_FORCE_DIST": "1"}),
                       (["--no-cpu-baseline", "--no-shape-1k", "--no-pmc", "--collective", "native"], {"PANACUS_BENCH_FORCE_DIST": "1"})):
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        e = dict(clean, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **env)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + small + extra, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=e, timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        lines = [l for l in r.stdout.decode().split("\n") if l.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "permuted_growth"):
            assert k in d, k
        assert d["steps"] == 12 and d["n_gpus"] == 1 and d["roofline"]["launches"] == 4   # every 3rd of 12 launches is timed
        assert d["checks"]["hist_sum"] == d["checks"]["expected_hist_sum"] == 200000
        rf = d["roofline"]
        one_shot = "--cover-route" in extra
        assert d["checks"]["one_shot_route_held"] is one_shot and rf["kernel"] == ("k_band_cover" if one_shot else "k_rows_build + k_rows_cover")
        # the line's own peak cross-check: algorithmic bytes over the step, and over the kernel that reads the steps, stay below the memory's peak
        assert 0 < rf["whole_step"]["frac"] < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
        assert abs(rf["whole_step"]["achieved"] - rf["algorithmic_bytes_per_launch"] / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6
        assert abs(d["value"] - 200000 * 64 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
        br = d["step_breakdown_ms"]
        assert br["band_cover"] > 0 and br["kernels_of_the_pass"] < d["ms_per_step"]
        if "--no-pmc" not in extra:   # the counter passes are driven by the run itself; on a box without rocprofv3 the line says why
            assert (rf["traffic"] is not None and rf["traffic"] > 0 and d["roofline_valu"]["frac"] > 0) or "rocprofv3" in rf["traffic_source"], rf["traffic_source"]
        else:
            assert rf["traffic"] is None and d["roofline_valu"] is None
        if not env:
            ho, rp = d["hist_only"], d["resident_pass"]
            assert ho["ms_per_call"] > 0 and ho["kernels_ms"]["band_cover"] > 0
            assert rp["ms_per_pass"] > 0 and rp["rows"] > 0 and rp["rows_route"]["cold_first_pass_ms"] > rp["rows_route"]["prepare_ms"] > 0
            assert rp["hist_sum"] == 200000
        pg = d["permuted_growth"]
        for k in ("seconds_per_call", "orders_per_s", "speedup_vs_1", "growth_kernel_ms_rank_max", "allreduce_ms",
                  "presence_pack_ms", "scaling", "sharding", "n_gpus"):
            assert k in pg, k
        assert pg["scaling"] == "strong" and pg["orders"] == 10 and pg["checks"]["sharded_equals_single_gpu"]
        assert pg["seconds_per_call"] > 0 and pg["growth_kernel_ms_rank_max"] > 0
        sim = pg["similarity_intersections"]
        assert sim["kernel_ms"] > 0 and sim["checks"]["symmetric"] and sim["checks"]["diagonal_sum"] > 0
        if env:
            assert pg["allreduce_ms"] > 0 and "rccl" in pg["collective_path"]
            # the headline graph split by node range (one rank here: its range is the whole graph), through the same collective
            sb = d["strong_scaling"]
            for k in ("workload", "n_gpus", "scaling", "sharding", "ms_per_step", "ms_per_step_1gpu", "speedup_vs_1", "value", "rank0", "checks"):
                assert k in sb, (k, sb)
            assert sb["scaling"] == "strong" and sb["n_gpus"] == 1 and sb["speedup_vs_1"] == 1.0
            assert sb["checks"]["hist_sum"] == 200000 and sb["checks"]["sharded_equals_single_gpu"] and sb["rank0"]["n_reruns"] == 0
        else:
            # the headline step on paths that are not sorted by id: forced onto the one-shot route it holds (spills, no rerun, no rows)
            sp = d["strayed_paths"]
            assert sp["checks"]["hist_sum"] == 200000 and sp["ms_per_step"] > 0 and sp["n_reruns"] == 0
            if one_shot:
                assert sp["one_shot_route_held"] and sp["spilled_steps_per_pass"] > 0 and sp["n_rows"] == 0
            if "--no-cpu-baseline" not in extra:
                assert sp["checks"]["hist_agrees_with_oracle"] is True
        if "--no-cpu-baseline" not in extra:
            cb, k1 = d["cpu_baseline"], d["shape_10Mx1k"]
            assert cb["agrees_with_gpu"] is True and cb["kind"] == "port" and cb["cores"] == 3 and cb["value"] > 0
            assert k1["checks"]["hist_sum"] == 150000 and k1["step_breakdown_ms"]["band_cover"] > 0 and k1["hist_only"]["ms_per_call"] > 0
            assert 0 < k1["roofline"]["whole_step"]["frac"] < k1["roofline"]["frac"] < 1
        outs.append(d["checks"])
        pgs.append(pg["checks"])
    for o in outs:
        o.pop("one_shot_route_held")
    assert outs[0] == outs[1] == outs[2] == outs[3]
    assert pgs[0] == pgs[1] == pgs[2] == pgs[3]
    # more ranks than devices: a clear failure, not a silent single-GPU run
    import torch
    want = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(want)] + small, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=clean, timeout=300)
    assert r.returncode != 0 and f"--gpus {want}" in r.stderr.decode() and "device(s) visible" in r.stderr.decode()
    assert not [l for l in r.stdout.decode().split("\n") if l.startswith("{")]


def test_full_size_cfg3_properties():
    """configs[2]: 10M nodes x 256 paths.  Every item lands in exactly one bin; the histogram is
    the bincount of the coverage vector; core nodes reach G; visiting order does not matter."""
    from panacus_amd import capi
    n, p = 10_000_000, 256
    with capi.Context(0) as c:
        c.set_csr_pansyn(42, n, p)
        order = np.arange(p, dtype=np.uint32)
        c.set_order(order, order, p)
        cnt, h = c.hist()
        assert c.info().n_rows == 0 and c.info().n_reruns == 0   # the cold call: the one-shot route over the steps, nothing derived
        assert int(h.sum()) == n and cnt[0] == 0xFFFFFFFF
        assert np.array_equal(np.bincount(cnt[1:], minlength=p + 1).astype(np.uint64), h)
        assert int(cnt[1:].max()) == p and c.info().n_general_paths == 0
        assert int(cnt[1:].astype(np.uint64).sum()) <= c.info().n_steps
        # 20 % of the nodes are in every path (pansyn "core"), 45 % in about one path
        assert abs(int(h[p]) / n - 0.20) < 0.01
        # direct parity on a prefix: node i of pansyn(seed, N, P) does not depend on N, so the first
        # million nodes have the coverage the oracle computes on the (seed, 1 M, P) graph
        n_pre = 1_000_000
        items, pre, _ = orc.pansyn(42, n_pre, p)
        pi = np.arange(p, dtype=np.uint64)
        assert np.array_equal(cnt[1:n_pre + 1], orc.coverage(items, pre, pi, pi, n_pre)[1:])
        del items
        # ... and the WHOLE graph: the resident steps read back, widened to the reference's u64 items, through the serial
        # restatement of abacus.rs:719-787 (about a second of CPU time) -- coverage vector and histogram bit for bit
        items32, off, _ = c.get_csr()
        ocov = orc.coverage(items32.astype(np.uint64), off, pi, pi, n)
        del items32
        assert np.array_equal(cnt, ocov) and np.array_equal(h, orc.hist(ocov, p))
        del ocov
        # ... and sampled nodes anywhere, from the generator's definition
        rng = np.random.default_rng(1)
        nodes = np.unique(np.concatenate([rng.integers(1, n + 1, size=1500), [1, n, n - 1, 2048, 2049, 4_999_999]]))
        assert np.array_equal(cnt[nodes].astype(np.int64), _pansyn_coverage_of(42, nodes, n, p))
        rng = np.random.default_rng(0)
        perm = rng.permutation(p).astype(np.uint32)
        c.set_order(perm, order, p)  # same groups visited in another order
        _, h2 = c.hist(want_countable=False)
        assert np.array_equal(h, h2)
        # -S style grouping (pairs of paths): a checksum that ties the two histograms together
        c.set_order(order, (order // 2).astype(np.uint32), p // 2)
        cnt2, h3 = c.hist()
        assert int(h3.sum()) == n
        assert np.all(cnt2[1:] <= cnt[1:]) and np.all(2 * cnt2[1:].astype(np.int64) >= cnt[1:])


def test_full_size_cfg3_with_paths_that_stray_against_the_oracle():
    """configs[2]'s shape on pansyn-v1r -- 10 M nodes x 256 paths whose steps are NOT sorted by id (1 % of the 64-step blocks
    reversed, 0.1 % copies of earlier blocks, 0.05 % moved elsewhere in the id space): the cold call takes the one-shot
    route and holds it (steps out of their band are spilled and added by the pass's tail: no rerun, no rows), and the WHOLE
    coverage vector and the histogram are the ones the serial restatement of abacus.rs:719-787 computes on the same steps
    (read back from HBM); grouped in pairs as well, where a step spilled by one path of a group may be in band on the other."""
    from panacus_amd import capi
    n, p = 10_000_000, 256
    with capi.Context(0) as c:
        c.set_csr_pansyn_rearranged(42, n, p)
        order = np.arange(p, dtype=np.uint32)
        c.set_order(order, order, p)
        cnt, h = c.hist()
        info = c.info()
        assert info.n_rows == 0 and info.n_reruns == 0 and info.band_route_failed == 0
        assert info.n_spilled_last > 1_000_000 and info.n_spill_bursts_last > 10_000       # ~0.17 % of 0.98 G steps
        assert int(h.sum()) == n and cnt[0] == 0xFFFFFFFF
        items32, off, _ = c.get_csr()
        items = items32.astype(np.uint64)
        del items32
        pi = np.arange(p, dtype=np.uint64)
        ocov = orc.coverage(items, off, pi, pi, n)
        assert np.array_equal(cnt, ocov) and np.array_equal(h, orc.hist(ocov, p))
        # the device generator made the oracle's graph: the first million nodes' worth of every path cannot be compared (the
        # rearrangement is by position), so compare the whole of a smaller graph elsewhere (test_gpu_band) and here the step count
        assert len(items) == int(off[-1]) == int(info.n_steps)
        gid = (pi // 2).astype(np.uint64)
        c.config(capi.CFG_DROP_DERIVED, 0)
        c.set_order(order, gid.astype(np.uint32), p // 2)
        cnt2, h2 = c.hist()
        assert c.info().n_rows == 0 and c.info().n_reruns == 0
        ocov2 = orc.coverage(items, off, pi, gid, n)
        assert np.array_equal(cnt2, ocov2) and np.array_equal(h2, orc.hist(ocov2, p // 2))


def test_node_range_shards_of_the_generated_graph_sum_to_the_whole():
    """pnx_set_csr_pansyn_shard: the nodes lo + 1 .. hi of pansyn(seed, N, P) as items 1 .. hi - lo -- the steps of the whole
    graph restricted to the range, re-based (distributed.shard_csr on the CPU generator's graph).  Items are independent, so
    histograms and ordered-growth curves of the shards add up to those of the whole graph: what bench.py's multi-GPU blocks do
    with one shard per rank and an all-reduce, done here with one GPU, one shard after the other."""
    from panacus_amd import capi
    from panacus_amd.distributed import even_node_range, shard_csr
    from panacus_amd.pansyn import random_orders
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    n, p, world = 150_000, 24, 3
    items, pre, lens = orc.pansyn(9, n, p)
    order = np.arange(p, dtype=np.uint32)
    gid = (order // 2).astype(np.uint32)
    G = p // 2
    cov = [coverage_abs(Threshold(ABSOLUTE, c), G) for c in (1, 2, 1)]
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), G) for q in (0.0, 0.0, 0.5)])
    perms = random_orders(5, 6, G)
    with capi.Context(0) as c:
        c.set_csr_pansyn(9, n, p, with_weights=True)
        c.set_order(order, gid, G)
        cnt_all, h_all = c.hist()
        out_all = c.ordered_growth(cov, qt, perms)
        h_sum = np.zeros_like(h_all)
        out_sum = np.zeros_like(out_all)
        edges = []
        for r in range(world):
            lo, hi = even_node_range(n, world, r)
            edges.append((lo, hi))
            c.set_csr_pansyn_shard(9, lo, hi - lo, p, with_weights=True)
            got, off, w = c.get_csr(want_weights=True)
            want_items, want_off, n_r = shard_csr(items, pre, lo + 1, hi + 1)
            assert n_r == hi - lo and np.array_equal(got, want_items) and np.array_equal(off, want_off)
            assert np.array_equal(w[1:], lens[lo + 1:hi + 1])
            c.set_order(order, gid, G)
            cnt, h = c.hist()
            assert np.array_equal(cnt[1:], cnt_all[lo + 1:hi + 1])
            h_sum += h
            out_sum += c.ordered_growth(cov, qt, perms)
        assert edges[0][0] == 0 and edges[-1][1] == n and all(edges[k][1] == edges[k + 1][0] for k in range(world - 1))
        assert np.array_equal(h_sum, h_all) and np.array_equal(out_sum, out_all)


def test_full_size_10Mx1k_against_the_oracle():
    """north_star's shape (10 M nodes x 1024 paths, 3.9 G steps) through the one-shot route, the WHOLE coverage vector and the
    histogram against the serial restatement of abacus.rs:719-787: with every path its own group the coverage of a graph is
    the sum of the coverages of disjoint sets of its paths, so the oracle runs over four quarters of the paths (the u64
    widening of all steps at once would be 31 GB of host memory)."""
    from panacus_amd import capi
    n, p = 10_000_000, 1024
    with capi.Context(0) as c:
        c.set_csr_pansyn(42, n, p)
        order = np.arange(p, dtype=np.uint32)
        c.set_order(order, order, p)
        cnt, h = c.hist()
        info = c.info()
        assert info.n_rows == 0 and info.n_reruns == 0       # the steps were read once, nothing was derived
        assert int(h.sum()) == n and cnt[0] == 0xFFFFFFFF
        items32, off, _ = c.get_csr()
    total = np.zeros(n + 1, dtype=np.uint64)
    q = p // 4
    for k in range(4):
        a, b = int(off[k * q]), int(off[(k + 1) * q])
        part = items32[a:b].astype(np.uint64)
        pre = (off[k * q:(k + 1) * q + 1] - off[k * q]).astype(np.uint64)
        pi = np.arange(q, dtype=np.uint64)
        cov = orc.coverage(part, pre, pi, pi, n)
        total[1:] += cov[1:]
        del part, cov
    del items32
    assert np.array_equal(cnt[1:].astype(np.uint64), total[1:])
    assert np.array_equal(h, np.bincount(total[1:].astype(np.int64), minlength=p + 1).astype(np.uint64))


def _pansyn_coverage_of(seed, nodes, n_nodes, n_paths):
    """coverage (# paths holding the node) straight from the pansyn-v1 definition, vectorised"""
    M = (1 << 64) - 1

    def sm(x):  # splitmix64 on uint64 arrays (wrap-around arithmetic)
        z = (x + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))

    from panacus_amd.pansyn import key
    nodes = np.asarray(nodes, dtype=np.uint64)
    thr = np.array([orc.lib().pansyn_node_thr(seed, int(i), n_paths) for i in nodes], dtype=np.uint64)
    cov = np.zeros(len(nodes), dtype=np.int64)
    with np.errstate(over="ignore"):
        k5 = np.uint64(key(seed, 5))
        for p in range(n_paths):
            kp = sm(np.array([k5 + np.uint64(p)], dtype=np.uint64))[0]
            h = sm(kp + nodes)
            cov += ((h >> np.uint64(11)) < thr)
    return cov


def test_more_than_2_to_32_steps():
    """S > 2^32 path steps (17 GB of CSR): every step index is 64-bit.  The coverage of sampled
    nodes is checked against the generator's definition, the rest through the histogram."""
    from panacus_amd import capi
    n, p = 9_000_000, 1300
    with capi.Context(0) as c:
        c.set_csr_pansyn(7, n, p)
        assert c.info().n_steps > (1 << 32)
        order = np.arange(p, dtype=np.uint32)
        c.set_order(order, order, p)
        cnt, h = c.hist()
        assert int(h.sum()) == n and c.info().n_general_paths == 0
        assert np.array_equal(np.bincount(cnt[1:], minlength=p + 1).astype(np.uint64), h)
        rng = np.random.default_rng(5)
        sample = np.unique(np.concatenate([rng.integers(1, n + 1, size=600), [1, 2047, 2048, 2049, n - 1, n]]))
        exp = _pansyn_coverage_of(7, sample, n, p)
        assert np.array_equal(cnt[sample].astype(np.int64), exp)
        # the last path starts beyond step 2^32: dropping it lowers exactly its nodes by one
        c.set_order(order[:-1], order[:-1], p - 1)
        cnt2, h2 = c.hist()
        d = cnt[1:].astype(np.int64) - cnt2[1:].astype(np.int64)
        assert d.min() >= 0 and d.max() == 1 and int(h2.sum()) == n


def test_full_size_cfg4_all_orders_and_masked_prefix_parity():
    """configs[3] at full size, all 128 orders.  Direct oracle parity at 10 M x 512 would take the
    serial loop hours, but items are independent: with exclude[i] = (i > 200 000) the full-size run
    must equal the oracle on the 200 000-node prefix graph -- which IS pansyn(42, 200 000, 512), the
    generator being counter-based per (path, node) -- bit for bit, for node and bp counts."""
    from panacus_amd import capi
    from panacus_amd.pansyn import random_orders
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    n, p, m, R = 10_000_000, 512, 200_000, 128
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]
    cov = [coverage_abs(Threshold(ABSOLUTE, cc), p) for cc, _ in pairs]
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), p) for _, q in pairs])
    perms = random_orders(42, R, p)
    items, pre, lens = orc.pansyn(42, m, p)
    pg = np.arange(p, dtype=np.uint64)
    check = [0, 1, 2, 17, 63, 64, 100, 127]
    excl = np.zeros(n + 1, dtype=np.uint8)
    excl[m + 1:] = 1
    with capi.Context(0) as c:
        for weighted in (False, True):
            c.set_csr_pansyn(42, n, p, with_weights=weighted)
            order = np.arange(p, dtype=np.uint32)
            c.set_order(order, order, p)
            w = lens if weighted else None
            if not weighted:
                # all 128 orders on the whole graph: the q = 0 curves end at the same totals, whatever the order
                cnt, h = c.hist()
                full = c.ordered_growth(cov, qt, perms)
                assert full.shape == (R, 3, p)
                assert np.all(full[:, 0, -1] == n - int(h[0])) and np.all(full[:, 1, -1] == n - int(h[0]) - int(h[1]))
                assert np.all(np.diff(full[:, :2].astype(np.int64), axis=2) >= 0)
                # mean over the orders of the first step = mean group size
                sizes = np.array([int((cnt[1:] > 0).sum())])  # sanity anchor only
                assert sizes[0] == n - int(h[0])
            c.set_exclude(excl)
            cnt, h = c.hist()
            ocov = orc.coverage(items, pre, pg, pg, m)
            assert np.array_equal(cnt[: m + 1], ocov) and not cnt[m + 1:].any()
            oh = orc.hist(ocov, p, w)
            assert np.array_equal(h[1:], oh[1:])  # bin 0 additionally holds the excluded items
            out = c.ordered_growth(cov, qt, perms)
            for r in check:
                for t, (cv, q) in enumerate(pairs):
                    exp = _oracle_growth(items, pre, m, p, pg, perms[r], cv, q, w)
                    assert out[r, t].tolist() == [int(x) for x in exp], (weighted, r, cv, q)
            if not weighted:
                # EVERY one of the 128 orders against the oracle, on a shorter prefix (12 500 nodes: the serial loop takes
                # milliseconds per order there); the device's work is the same 10 M x 512 matrix whatever the mask
                m2 = 12_500
                items2, pre2, _ = orc.pansyn(42, m2, p)
                excl2 = np.zeros(n + 1, dtype=np.uint8)
                excl2[m2 + 1:] = 1
                c.set_exclude(excl2)
                c.hist()
                out2 = c.ordered_growth(cov, qt, perms)
                pi2 = np.arange(p, dtype=np.uint64)
                for r in range(R):
                    # one path per group: the paths in the order of their group's rank
                    rows, cols = orc.by_group(items2, pre2, perms[r].astype(np.uint64), pi2, m2)
                    for t, (cv, q) in enumerate(pairs):
                        exp = orc.ordered_growth(rows, cols, p, (orc.ABSOLUTE, cv), (orc.RELATIVE, q), None)
                        assert out2[r, t].tolist() == [int(x) for x in exp], (r, cv, q)
            c.set_exclude(None)


def test_full_size_cfg4_properties():
    """configs[3]: 10M nodes x 512 paths, permuted ordered growth.  For q = 0 every order ends at
    the same total, curves are non-decreasing, and the mean first step equals the mean group
    size; for the quorum pair the last value counts the items present in >= half of the groups."""
    from panacus_amd import capi
    from panacus_amd.pansyn import random_orders
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    n, p = 10_000_000, 512
    with capi.Context(0) as c:
        c.set_csr_pansyn(42, n, p)
        order = np.arange(p, dtype=np.uint32)
        c.set_order(order, order, p)
        cnt, h = c.hist()
        pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]
        cov = [coverage_abs(Threshold(ABSOLUTE, cc), p) for cc, _ in pairs]
        qt = np.stack([quorum_table(Threshold(RELATIVE, q), p) for _, q in pairs])
        perms = random_orders(42, 6, p)
        out = c.ordered_growth(cov, qt, perms)
        for r in range(6):
            assert int(out[r, 0, -1]) == n - int(h[0])                    # deg >= 1
            assert int(out[r, 1, -1]) == n - int(h[0]) - int(h[1])        # deg >= 2
            assert np.all(np.diff(out[r, 0].astype(np.int64)) >= 0)
            assert np.all(np.diff(out[r, 1].astype(np.int64)) >= 0)
            # at the last rank the quorum bound is ceil(G * 0.5) for an item whose last group is
            # the last rank; items seen in >= half of all groups are always counted there
            assert int(out[r, 2, -1]) >= int(h[(p + 1) // 2:].sum()) - 1
        # first rank: res[0] = size of the first group of each order
        sizes = np.array([int(out[r, 0, 0]) for r in range(6)])
        assert np.all(sizes > 0.30 * n) and np.all(sizes < 0.45 * n)
        # identity order == reference ordered-histgrowth semantics, q = 0: distinct items so far
        ident = c.ordered_growth(cov[:1], qt[:1])
        assert int(ident[0, 0, -1]) == n - int(h[0])


def test_full_size_cfg4_similarity_against_the_presence_rows():
    """configs[3]'s matrix (10 M items x 512 groups) through K5 (int8 MFMA): every diagonal entry and 96 random pairs
    against popcounts of the device's own presence export (node counts); for bp, sampled pairs against the weighted
    sums of the same rows, with weights wide enough for three 7-bit digits -- and the masked-prefix parity against the
    oracle (items are independent: exclude[i] = (i > 100 000) must give the oracle's matrix of the prefix graph)"""
    from panacus_amd import capi
    n, p, m = 10_000_000, 512, 100_000
    rng = np.random.default_rng(4)
    pairs = [(int(a), int(b)) for a, b in rng.integers(0, p, size=(96, 2))]
    order = np.arange(p, dtype=np.uint32)
    with capi.Context(0) as c:
        c.set_csr_pansyn(42, n, p, with_weights=True)
        c.set_order(order, order, p)
        w = c.get_weights()
        # node counts
        c.config(capi.CFG_USE_WEIGHTS, 0)
        inter = c.group_intersections()
        rows = c.presence()  # [G, words] u64, bit i % 64 of word i / 64 = item i
        assert (inter == inter.T).all()
        assert np.array_equal(np.diag(inter), np.bitwise_count(rows).sum(axis=1, dtype=np.uint64))
        for a, b in pairs:
            assert int(inter[a, b]) == int(np.bitwise_count(rows[a] & rows[b]).sum(dtype=np.uint64)), (a, b)
        # bp, weights scaled into the three-digit range (pansyn lengths < 2^14; x 9 + 3 < 2^17)
        c.config(capi.CFG_USE_WEIGHTS, 1)
        w3 = (w.astype(np.uint64) * 9 + 3).astype(np.uint32)
        w3[0] = 0
        c.set_weights(w3)
        inter = c.group_intersections()
        wq = np.zeros(rows.shape[1] * 64, dtype=np.uint64)
        wq[: n + 1] = w3
        for a, b in pairs[:10]:
            bits = np.unpackbits((rows[a] & rows[b]).view(np.uint8), bitorder="little")
            assert int(inter[a, b]) == int(wq[bits.astype(bool)].sum(dtype=np.uint64)), (a, b)
        # masked prefix against the oracle
        excl = np.zeros(n + 1, dtype=np.uint8)
        excl[m + 1:] = 1
        c.set_exclude(excl)
        inter = c.group_intersections()
    items, pre, lens = orc.pansyn(42, m, p)
    pg = np.arange(p, dtype=np.uint64)
    r, cc = orc.by_group(items, pre, pg, pg, m)
    exp, _, _ = orc.similarity(r, cc, p, node_lens=(lens.astype(np.uint64) * 9 + 3).astype(np.uint32))
    assert np.array_equal(inter, exp)


@pytest.mark.parametrize("variant", [3, 2])
def test_shuffled_paths_at_scale_are_sorted_once_and_read_back_in_order(variant):
    """2 M nodes x 32 paths (25 M steps), every path shuffled: the preparation sorts them (no scatter route, no run index),
    the coverage vector equals the one of the unshuffled graph, a second order gives the same, and pnx_get_csr still
    returns the caller's order"""
    from panacus_amd import capi
    n, p = 2_000_000, 32
    order = np.arange(p, dtype=np.uint32)
    with capi.Context(0) as c:
        c.config(capi.CFG_COVER_VARIANT, variant)
        c.set_csr_pansyn(7, n, p)
        c.set_order(order, order // 2, p // 2)
        cnt0, h0 = c.hist()
        items, off, _ = c.get_csr()
        sh = items.copy()
        rng = np.random.default_rng(1)
        for k in range(p):
            rng.shuffle(sh[int(off[k]):int(off[k + 1])])
        c.set_csr(sh, off, n)
        c.set_order(order, order // 2, p // 2)
        cnt, h = c.hist()
        info = c.info()
        assert info.n_sorted_paths == (p if variant == 2 else 0) and info.n_scatter_paths == 0 and info.n_run_paths == 0
        # (the default route tries the one-shot pass on this shape; its index kernel's probes find the paths shuffled, the coverage
        # kernel does not start, and the pass runs over path rows: one cheap rerun, remembered for the upload)
        assert info.n_reruns == (1 if variant == 3 else 0) and info.band_route_failed == (1 if variant == 3 else 0)
        assert np.array_equal(cnt, cnt0) and np.array_equal(h, h0)
        c.set_order(order[::-1].copy(), (order // 2)[::-1].max() - (order // 2)[::-1], p // 2)
        cnt2, h2 = c.hist()
        assert np.array_equal(cnt2, cnt0) and np.array_equal(h2, h0)
        back, back_off, _ = c.get_csr()
        assert np.array_equal(back, sh) and np.array_equal(back_off, off)


@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_multi_rank_blocks_on_one_device(ranks):
    """`bench.py --gpus N` with N = 2 and 8 ranks on ONE device (PANACUS_BENCH_ONE_DEVICE=1: every rank on GPU 0, counters
    reduced over gloo): the weak-scaling headline, the `strong_scaling` block (one headline graph split into N node ranges)
    and `permuted_growth` (node-range shards, out[R][T][G] summed over the ranks) run their multi-rank code -- the shards'
    sums equal the single-GPU results bit for bit (the blocks check that themselves and fail the run otherwise).  Not a scaling
    measurement (the ranks share one GPU); SURVEY 8e / VERDICT r5 item 6: the multi-rank path exercised where one GPU is all
    there is."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e = dict(clean, PANACUS_BENCH_ONE_DEVICE="1")
    args = ["--gpus", str(ranks), "--nodes", "400000", "--paths", "48", "--steps", "6", "--warmup", "2", "--strong-steps", "6",
            "--pg-nodes", "400000", "--pg-paths", "40", "--pg-orders", "12", "--pg-reps", "2", "--no-pmc", "--no-cpu-baseline",
            "--cover-route", "1", "--ss2-nodes", "120000", "--ss2-samples", "6"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e,
                       timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().split("\n") if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["one_device"] is True and d["scaling"] == "weak"
    assert d["checks"]["hist_sum"] == d["checks"]["expected_hist_sum"] == ranks * 400000
    ss, pg = d["strong_scaling"], d["permuted_growth"]
    assert ss["n_gpus"] == ranks and ss["scaling"] == "strong" and ss["speedup_vs_1"] > 0
    assert pg["n_gpus"] == ranks and pg["checks"]["sharded_equals_single_gpu"] is True and pg["speedup_vs_1"] > 0
    s2 = d["strong_scaling_pggb_shape"]   # a graph whose steps are not spread evenly over the ids, node ranges balanced by step count
    assert "error" not in s2, s2
    assert s2["n_gpus"] == ranks and s2["checks"]["sharded_equals_single_gpu"] is True and len(s2["steps_per_rank"]) == ranks
    assert s2["imbalance_by_steps"] <= s2["imbalance_even_node_ranges"] + 1e-9 and s2["imbalance_by_steps"] < 1.2
