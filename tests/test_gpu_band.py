"""The one-shot route (csrc/kernels_band.hip): steps -> coverage vector + histogram in ONE read of the ItemTable, for the
first sweep of a graph whose paths follow the order of the ids -- against the CPU oracle (abacus.rs:719-787 restated),
through the C ABI.  Integer work: bit-exact equality everywhere.  Steps that are not where the order of the ids says
(inversions, jumps back, duplications, translocations) are spilled and added by the pass's tail; only a graph whose paths
do not follow the ids at all makes the pass void -- the library then runs it again over path rows, and the caller sees
the same numbers."""
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


@pytest.fixture
def band(ctx):
    """every pass over the steps themselves while no rows exist"""
    from panacus_amd import capi
    ctx.config(capi.CFG_COVER_ROUTE, 1)
    yield ctx
    ctx.config(capi.CFG_COVER_ROUTE, 0)


def _oracle_hist(items, pre, pi, gi, n, G, w=None, exclude=None):
    cov = orc.coverage(items, pre, pi, gi, n, exclude)
    return cov, orc.hist(cov, G, w)


def _concat(segs):
    pre = np.zeros(len(segs) + 1, dtype=np.uint64)
    pre[1:] = np.cumsum([len(s) for s in segs])
    items = np.concatenate(segs).astype(np.uint64) if segs else np.zeros(0, dtype=np.uint64)
    return items, pre


def _sorted_graph(n, seed):
    """paths that run through the ids in order, of every kind: ascending, descending, every step doubled, runs of one
    id, empty, one step, strides that leave most item tiles untouched, both ends only"""
    items, pre, lens = orc.pansyn(seed, n, 6)
    segs = [items[int(pre[k]):int(pre[k + 1])].copy() for k in range(6)]
    segs[1] = np.sort(segs[1])[::-1].copy()
    segs[2] = np.sort(segs[2])
    segs[3] = np.repeat(np.sort(segs[3]), 2)
    segs[4] = np.sort(np.concatenate([segs[4], np.full(300, max(1, n // 3), dtype=np.uint64)]))
    segs[0] = np.sort(segs[0])
    segs[5] = np.sort(segs[5])[::-1].copy()
    segs.append(np.zeros(0, dtype=np.uint64))
    segs.append(np.array([n // 2 + 1], dtype=np.uint64))
    segs.append(np.arange(1, n + 1, 5003, dtype=np.uint64))
    segs.append(np.arange(n, 0, -4999, dtype=np.uint64))
    segs.append(np.array([1, 1, n, n], dtype=np.uint64))
    segs.append(np.array([n, 1], dtype=np.uint64))
    items, pre = _concat(segs)
    return items, pre, lens


@pytest.mark.parametrize("n", [1, 2047, 2048, 8191, 8192, 8193, 40_961, 300_000])
def test_band_route_on_sorted_paths_of_every_kind(band, n):
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    ctx = band
    items, pre, lens = _sorted_graph(n, 3 + n % 7)
    P = len(pre) - 1
    rng = np.random.default_rng(n)
    excl = (rng.random(n + 1) < 0.05).astype(np.uint8)
    excl[0] = 0
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=lens, exclude=excl)
    assert ctx.info().n_rows == 0                      # nothing derived at upload
    reruns = ctx.info().n_reruns
    pi = np.arange(P, dtype=np.uint64)
    ctx.set_order(pi, pi, P)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, pi, pi, n, P, lens, excl)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    # groups of three, visited backwards, two paths left out
    order = np.array([q for q in range(P - 1, -1, -1) if q not in (2, 7)], dtype=np.uint64)
    gid = (np.arange(len(order)) // 3).astype(np.uint64)
    G = int(gid.max()) + 1
    ctx.set_order(order, gid, G)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, order, gid, n, G, lens, excl)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    # the presence matrix written by the same kernel, and growth on top of it
    r, c = orc.by_group(items, pre, order, gid, n, excl)
    bits = ctx.presence()
    got_rows = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, 1: n + 1]
    assert (got_rows.T == (orc.table_rows(r, c, G) != 0)).all()
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), G) for q in (0.0, 0.5)])
    out = ctx.ordered_growth([coverage_abs(Threshold(ABSOLUTE, 1), G)] * 2, qt)
    for t, q in enumerate((0.0, 0.5)):
        exp = orc.ordered_growth(r, c, G, (orc.ABSOLUTE, 1), (orc.RELATIVE, q), lens)
        assert out[0, t].tolist() == [int(x) for x in exp]
    info = ctx.info()
    assert info.n_rows == 0 and info.n_reruns == reruns  # every pass held, no rows were ever derived
    assert info.n_band_passes >= 2 and info.band_route_failed == 0
    # pnx_info_sized: a caller with an older, shorter pnx_info_t gets its prefix and nothing behind it
    import ctypes as C
    from panacus_amd import capi
    buf = (C.c_uint8 * 64)(*([0xAB] * 64))
    lib_bytes = C.c_size_t(0)
    assert capi.load().pnx_info_sized(ctx._h, buf, 24, C.byref(lib_bytes)) == 0
    assert lib_bytes.value == C.sizeof(capi.PnxInfo) and bytes(buf[24:]) == b"\xAB" * 40
    assert bytes(buf[:24]) == bytes(info)[:24]


@pytest.mark.parametrize("seed", range(24))
def test_band_route_fuzz(band, seed):
    """random sorted paths (either direction, random duplicates, random lengths incl. 0) on item counts around the tile
    and band sizes, random grouping, random exclusion; node counts and bp"""
    ctx = band
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([5, 63, 64, 2047, 2049, 8191, 8192, 8200, 16383, 16385, 50_000, 131_071]))
    P = int(rng.integers(1, 40))
    segs = []
    for _ in range(P):
        kind = rng.integers(0, 5)
        ln = 0 if kind == 0 else int(rng.integers(1, max(2, min(3 * n, 60_000))))
        s = np.sort(rng.integers(1, n + 1, size=ln).astype(np.uint64))
        if kind == 2:
            s = s[::-1].copy()
        segs.append(s)
    items, pre = _concat(segs)
    lens = rng.integers(0, 70_000, size=n + 1).astype(np.uint32)
    lens[0] = 0
    excl = (rng.random(n + 1) < rng.choice([0.0, 0.1])).astype(np.uint8)
    excl[0] = 0
    weighted = bool(seed % 2)
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=lens if weighted else None, exclude=excl if excl.any() else None)
    reruns = ctx.info().n_reruns
    order = rng.permutation(P)[: int(rng.integers(1, P + 1))].astype(np.uint64)
    gid = np.cumsum(rng.random(len(order)) < 0.6).astype(np.uint64)
    gid -= gid[0]
    G = int(gid.max()) + 1
    ctx.set_order(order, gid, G)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, order, gid, n, G, lens if weighted else None, excl if excl.any() else None)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    assert ctx.info().n_reruns == reruns and (ctx.info().n_rows == 0 or len(items) == 0)   # (no steps at all: nothing to read once)


def _check(ctx, items, pre, n, order, gid, G, lens=None, excl=None, presence=False):
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, order, gid, n, G, lens, excl)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    if presence:
        r, c = orc.by_group(items, pre, order, gid, n, excl)
        bits = ctx.presence()
        got_rows = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, 1: n + 1]
        assert (got_rows.T == (orc.table_rows(r, c, G) != 0)).all()
        cnt2, h2 = ctx.hist()           # the pass that wrote M counted the same
        assert np.array_equal(cnt2, ocov) and np.array_equal(h2, oh)


def test_steps_out_of_place_are_spilled_and_added_by_the_tail(band):
    """two steps swapped across bands; a path jittered locally; a stretch visited a second time far away; a stretch that
    belongs elsewhere -- no pass is run again, no rows are derived, and the numbers are the oracle's (coverage vector,
    histogram of node counts and of bp, presence matrix)"""
    ctx = band
    n = 60_000
    items, pre, lens = orc.pansyn(5, n, 8)
    segs = [np.sort(items[int(pre[k]):int(pre[k + 1])]) for k in range(8)]
    s = segs[5].copy()
    s[len(s) // 2], s[len(s) // 2 + 9000] = s[len(s) // 2 + 9000], s[len(s) // 2]   # two steps swapped across bands
    segs[5] = s
    for a in range(0, len(segs[2]) - 50, 211):                                        # local inversions, some across a band edge
        segs[2][a:a + 37] = segs[2][a:a + 37][::-1].copy()
    s = segs[3]
    segs[3] = np.concatenate([s[:9000], s[200:500], s[9000:], s[100:164]])           # two stretches visited again, out of place
    s = segs[6][::-1].copy()                                                          # a descending path ...
    s[4000:4100] = np.arange(n - 99, n + 1, dtype=np.uint64)                          # ... with a stretch from the other end of the graph
    segs[6] = s
    segs[7] = np.concatenate([segs[7], segs[7][:3000]])                               # the path's start once more at its end
    items, pre = _concat(segs)
    rng = np.random.default_rng(4)
    excl = (rng.random(n + 1) < 0.03).astype(np.uint8)
    excl[0] = 0
    pi = np.arange(8, dtype=np.uint64)
    before = ctx.info().n_reruns
    for weights, exclude in ((None, None), (lens, excl)):
        ctx.set_csr(items.astype(np.uint32), pre, n, weights=weights, exclude=exclude)
        ctx.set_order(pi, pi, 8)
        _check(ctx, items, pre, n, pi, pi, 8, weights, exclude, presence=True)
        info = ctx.info()
        assert info.n_reruns == before and info.n_rows == 0 and info.band_route_failed == 0
        assert info.n_spilled_last > 0 and info.n_spilled_total >= info.n_spilled_last
        # groups of several paths: a step spilled by one path of a group may be in band on another
        order = np.array([6, 2, 3, 5, 0, 7, 1], dtype=np.uint64)
        gid = np.array([0, 0, 0, 1, 1, 2, 2], dtype=np.uint64)
        ctx.set_order(order, gid, 3)
        _check(ctx, items, pre, n, order, gid, 3, weights, exclude, presence=True)
        assert ctx.info().n_reruns == before and ctx.info().n_rows == 0


@pytest.mark.parametrize("seed", range(16))
def test_band_route_fuzz_with_paths_that_stray(band, seed):
    """random paths sorted by id (either direction), then rearranged at random: blocks reversed, blocks copied from
    elsewhere in the path, blocks moved to other ids, single steps replaced; random grouping, exclusion, node counts and bp"""
    ctx = band
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([700, 8191, 8192, 8200, 16385, 50_000, 131_071, 400_000]))
    P = int(rng.integers(2, 24))
    segs = []
    for _ in range(P):
        ln = int(rng.integers(1, max(2, min(2 * n, 80_000))))
        s = np.sort(rng.integers(1, n + 1, size=ln).astype(np.uint64))
        if rng.random() < 0.3:
            s = s[::-1].copy()
        for _ in range(int(rng.integers(0, 12))):
            if ln < 4:
                break
            a = int(rng.integers(0, ln - 1))
            w = int(rng.integers(1, min(ln - a, 300) + 1))
            kind = rng.integers(0, 4)
            if kind == 0:
                s[a:a + w] = s[a:a + w][::-1].copy()
            elif kind == 1:
                src = int(rng.integers(0, ln - w + 1))
                s[a:a + w] = s[src:src + w].copy()
            elif kind == 2:
                s[a:a + w] = (s[a:a + w] - 1 + int(rng.integers(1, n))) % n + 1
            else:
                s[a] = int(rng.integers(1, n + 1))
        segs.append(s)
    items, pre = _concat(segs)
    lens = rng.integers(0, 70_000, size=n + 1).astype(np.uint32)
    lens[0] = 0
    excl = (rng.random(n + 1) < rng.choice([0.0, 0.1])).astype(np.uint8)
    excl[0] = 0
    weighted = bool(seed % 2)
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=lens if weighted else None, exclude=excl if excl.any() else None)
    reruns = ctx.info().n_reruns
    order = rng.permutation(P)[: int(rng.integers(1, P + 1))].astype(np.uint64)
    gid = np.cumsum(rng.random(len(order)) < 0.5).astype(np.uint64)
    gid -= gid[0]
    G = int(gid.max()) + 1
    ctx.set_order(order, gid, G)
    os.environ["PNX_BAND_SPLITS"] = str(1 + seed % 4)     # bands shared by 1..4 workgroups (capped by the number of groups)
    try:
        _check(ctx, items, pre, n, order, gid, G, lens if weighted else None, excl if excl.any() else None, presence=seed % 3 == 0)
    finally:
        del os.environ["PNX_BAND_SPLITS"]
    info = ctx.info()
    # the scans of the tail are bounded by the size of the graph: a tiny graph with much disorder may be run again over rows
    assert info.n_reruns <= reruns + 2
    if info.n_reruns == reruns:
        assert info.n_rows == 0


def _graph_with_shuffled_paths(n, P, which, seed, half=()):
    items, pre, lens = orc.pansyn(5, n, P)
    rng = np.random.default_rng(seed)
    segs = [np.sort(items[int(pre[k]):int(pre[k + 1])]) for k in range(P)]
    for k in which:
        segs[k] = rng.permutation(segs[k])                 # no order at all: every step is out of place
    for k in half:                                         # in order for the first 55 %, shuffled from there on
        m = len(segs[k]) * 55 // 100
        segs[k] = np.concatenate([segs[k][:m], rng.permutation(segs[k][m:])])
    items, pre = _concat(segs)
    return items, pre, lens


@pytest.mark.parametrize("splits", [1, 3])
def test_paths_that_do_not_follow_the_ids_are_sorted_at_upload_or_left_to_bitmaps(band, splits):
    """three shuffled paths, two that are shuffled over their last 45 %, among sorted ones and one with strays.  The shuffled ones
    are stored a second time at upload, sorted (the pass reads the copies like any path; pnx_get_csr returns what was uploaded);
    the partly shuffled ones are recognised by the index kernel: their groups are taken out of the bands, their steps marked in
    per-group bitmaps and folded in by the pass's tail.  No rerun, no rows, the oracle's numbers (coverage vector, histogram of
    node counts and of bp with exclusion, presence matrix)"""
    ctx = band
    n = 300_000
    items, pre, lens = _graph_with_shuffled_paths(n, 10, (1, 4, 6), 2, half=(8, 3))
    a = int(pre[5]) + 1000
    items[a:a + 300] = items[a + 5000:a + 5300].copy()      # (path 5 strays: spilled steps in the same pass)
    rng = np.random.default_rng(9)
    excl = (rng.random(n + 1) < 0.04).astype(np.uint8)
    excl[0] = 0
    pi = np.arange(10, dtype=np.uint64)
    os.environ["PNX_BAND_SPLITS"] = str(splits)
    try:
        for weights, exclude in ((None, None), (lens, excl)):
            ctx.set_csr(items.astype(np.uint32), pre, n, weights=weights, exclude=exclude)
            before = ctx.info().n_reruns
            assert ctx.info().n_sorted_copies == 3
            got, off, _ = ctx.get_csr()
            assert np.array_equal(got, items.astype(np.uint32)) and np.array_equal(off, pre)   # the ItemTable as uploaded
            ctx.set_order(pi, pi, 10)                       # every path its own group
            _check(ctx, items, pre, n, pi, pi, 10, weights, exclude, presence=True)
            info = ctx.info()
            assert info.n_reruns == before and info.n_rows == 0 and info.band_route_failed == 0
            assert info.n_loose_groups_last == 2 and info.n_spilled_last > 0
            gid = (pi // 2).astype(np.uint64)               # groups of two: a loose path takes its partner along
            ctx.set_order(pi, gid, 5)
            _check(ctx, items, pre, n, pi, gid, 5, weights, exclude, presence=True)
            info = ctx.info()
            assert info.n_reruns == before and info.n_rows == 0 and info.n_loose_groups_last == 2
            order = np.array([9, 6, 0, 8, 3], dtype=np.uint64)      # a subset, the two loose paths in one group
            gid = np.array([0, 1, 1, 2, 2], dtype=np.uint64)
            ctx.set_order(order, gid, 3)
            _check(ctx, items, pre, n, order, gid, 3, weights, exclude, presence=True)
            assert ctx.info().n_reruns == before and ctx.info().n_loose_groups_last == 1
            ctx.set_order(np.array([0, 1, 5], dtype=np.uint64), np.array([0, 1, 2], dtype=np.uint64), 3)   # none of the loose ones
            _check(ctx, items, pre, n, np.array([0, 1, 5], dtype=np.uint64), np.array([0, 1, 2], dtype=np.uint64), 3, weights, exclude)
            assert ctx.info().n_loose_groups_last == 0
    finally:
        del os.environ["PNX_BAND_SPLITS"]


def test_more_loose_groups_than_a_pass_takes_in_void_the_pass_and_the_rows_take_over(band):
    ctx = band
    n = 200_000
    P = 40
    items, pre, _ = _graph_with_shuffled_paths(n, P, (), 3, half=tuple(range(0, P, 2)))     # 20 groups with a partly shuffled path: more than 16
    ctx.set_csr(items.astype(np.uint32), pre, n)
    assert ctx.info().n_sorted_copies == 0
    pi = np.arange(P, dtype=np.uint64)
    ctx.set_order(pi, pi, P)
    before = ctx.info().n_reruns
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, pi, pi, n, P)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    info = ctx.info()
    assert info.n_reruns == before + 1 and info.n_rows > 0         # run again, over rows
    assert info.n_band_passes == 1 and info.band_route_failed == 1
    cnt, h = ctx.hist()                                             # and the graph is remembered: no second attempt
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh) and ctx.info().n_reruns == before + 1
    # the next upload starts from a clean slate (the flags of the groups were set back by the void pass's tail)
    items, pre, _ = _graph_with_shuffled_paths(n, 6, (), 4, half=(2,))
    ctx.set_csr(items.astype(np.uint32), pre, n)
    pi = np.arange(6, dtype=np.uint64)
    ctx.set_order(pi, pi, 6)
    _check(ctx, items, pre, n, pi, pi, 6)
    assert ctx.info().n_reruns == before + 1 and ctx.info().n_loose_groups_last == 1 and ctx.info().n_rows == 0


def _paths_with_large_rearrangements(n, P, seed):
    """pansyn paths, sorted; then per path one LARGE rearrangement: an inversion, a tandem duplication, a stretch moved elsewhere,
    two inversions, an inversion inside a descending path -- each tens of thousands of steps long"""
    items, pre, lens = orc.pansyn(seed, n, P)
    rng = np.random.default_rng(seed)
    segs = []
    for k in range(P):
        s = np.sort(items[int(pre[k]):int(pre[k + 1])])
        ln = len(s)
        w = ln // 12
        a = int(rng.integers(ln // 8, ln - 2 * w - ln // 8))
        kind = k % 6
        if kind == 0:
            s[a:a + w] = s[a:a + w][::-1].copy()
        elif kind == 1:
            s = np.concatenate([s[:a + w], s[a:a + w], s[a + w:]])
        elif kind == 2:
            cut = s[a:a + w].copy()
            rest = np.concatenate([s[:a], s[a + w:]])
            b = int(rng.integers(0, len(rest)))
            s = np.concatenate([rest[:b], cut, rest[b:]])
        elif kind == 3:
            s[a:a + w] = s[a:a + w][::-1].copy()
            s[a + w + 5000:a + 2 * w] = s[a + w + 5000:a + 2 * w][::-1].copy()
        elif kind == 4:
            s = s[::-1].copy()
            s[a:a + w] = s[a:a + w][::-1].copy()
        segs.append(s)                                  # kind 5: left sorted
    items, pre = _concat(segs)
    return items, pre, lens


@pytest.mark.parametrize("splits", [1, 2])
def test_paths_with_large_rearrangements_are_cut_into_pieces_that_follow_the_ids(band, splits):
    """inversions, duplications and translocations of a twelfth of a path: the upload's summaries of the chunks find the breaks,
    the one-shot pass takes the pieces as entries of their own under the path's group -- no rerun, no rows, few spilled steps
    (the chunk a break lies in), and the oracle's numbers (coverage vector, node and bp histograms with exclusion, presence)"""
    ctx = band
    n, P = 1_500_000, 12
    items, pre, lens = _paths_with_large_rearrangements(n, P, 21)
    rng = np.random.default_rng(8)
    excl = (rng.random(n + 1) < 0.02).astype(np.uint8)
    excl[0] = 0
    pi = np.arange(P, dtype=np.uint64)
    os.environ["PNX_BAND_SPLITS"] = str(splits)
    try:
        for weights, exclude in ((None, None), (lens, excl)):
            ctx.set_csr(items.astype(np.uint32), pre, n, weights=weights, exclude=exclude)
            before = ctx.info().n_reruns
            assert ctx.info().n_path_cuts >= 10                 # 2 per inversion / duplication, 2-3 per translocation, 4 for two inversions
            ctx.set_order(pi, pi, P)
            _check(ctx, items, pre, n, pi, pi, P, weights, exclude, presence=weights is None)
            info = ctx.info()
            assert info.n_reruns == before and info.n_rows == 0 and info.band_route_failed == 0 and info.n_loose_groups_last == 0
            assert info.n_band_entries == P + info.n_path_cuts
            assert info.n_spilled_last < len(items) // 100      # (uncut, a twelfth of every path would have been spilled: the pass void)
            order = np.array([7, 0, 3, 5, 1, 2, 10], dtype=np.uint64)       # a subset, groups of several paths
            gid = np.array([0, 0, 1, 1, 1, 2, 3], dtype=np.uint64)
            ctx.set_order(order, gid, 4)
            _check(ctx, items, pre, n, order, gid, 4, weights, exclude, presence=True)
            assert ctx.info().n_reruns == before and ctx.info().n_rows == 0
    finally:
        del os.environ["PNX_BAND_SPLITS"]


def test_a_borrowing_context_takes_the_cuts_and_the_sorted_copies_along():
    """pnx_share_csr: the second context reads the first one's steps in place -- with the cuts of its paths and the sorted copy
    of its shuffled path (they lie behind the steps, in the owner's buffer) --, under an order of its own"""
    from panacus_amd import capi
    n, P = 1_500_000, 12
    items, pre, _ = _paths_with_large_rearrangements(n, P, 21)
    rng = np.random.default_rng(2)
    a0, b0 = int(pre[11]), int(pre[12])
    items[a0:b0] = rng.permutation(items[a0:b0])
    a, b = capi.Context(0), capi.Context(0)
    try:
        for c in (a, b):
            c.config(capi.CFG_COVER_ROUTE, 1)
        a.set_csr(items.astype(np.uint32), pre, n)
        assert a.info().n_path_cuts >= 10 and a.info().n_sorted_copies == 1
        b.share_csr(a)
        pa = np.arange(P, dtype=np.uint64)
        pb = pa[::-1].copy()
        gb = (np.arange(P, dtype=np.uint64) // 3)
        a.set_order(pa, pa, P)
        b.set_order(pb, gb, 4)
        for c, (pi, gi, G) in ((a, (pa, pa, P)), (b, (pb, gb, 4)), (a, (pa, pa, P))):
            cnt, h = c.hist()
            ocov, oh = _oracle_hist(items, pre, pi, gi, n, G)
            assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
            assert c.info().n_reruns == 0      # (a shared graph has its path rows made by the owner: both contexts sweep those)
    finally:
        b.close()
        a.close()


def test_small_disorder_makes_no_cuts(band):
    """pansyn-v1r (blocks of 64 steps reversed, copied, moved) and a path with no order at all: nothing to cut at -- the first is
    spilled, the second stored once more, sorted"""
    ctx = band
    n, P = 600_000, 6
    items, pre, _ = orc.pansyn_rearranged(9, n, P)
    rng = np.random.default_rng(1)
    a, b = int(pre[2]), int(pre[3])
    items[a:b] = rng.permutation(items[a:b])
    ctx.set_csr(items.astype(np.uint32), pre, n)
    assert ctx.info().n_path_cuts == 0
    pi = np.arange(P, dtype=np.uint64)
    before = ctx.info().n_reruns
    ctx.set_order(pi, pi, P)
    _check(ctx, items, pre, n, pi, pi, P)
    info = ctx.info()
    assert info.n_reruns == before and info.n_rows == 0 and info.n_loose_groups_last == 0 and info.n_sorted_copies == 1 and info.n_band_entries == P


@pytest.mark.parametrize("n,p", [(70_000, 5), (300_000, 12), (2_000_000, 24)])
def test_rearranged_pansyn_device_generator_and_pass(band, n, p):
    """pansyn-v1r (1 % of the 64-step blocks reversed, 0.1 % copied from earlier in the path, 0.05 % moved elsewhere): the
    device generator makes the oracle's graph, and the one-shot pass over it holds -- spills, no rerun, no rows"""
    ctx = band
    items, pre, lens = orc.pansyn_rearranged(13, n, p)
    ctx.set_csr_pansyn_rearranged(13, n, p, with_weights=True)
    got, off, w = ctx.get_csr(want_weights=True)
    assert np.array_equal(got, items.astype(np.uint32)) and np.array_equal(off, pre) and np.array_equal(w, lens)
    before = ctx.info().n_reruns
    pi = np.arange(p, dtype=np.uint64)
    ctx.set_order(pi, pi, p)
    _check(ctx, items, pre, n, pi, pi, p, lens)
    gid = (pi // 2).astype(np.uint64)
    ctx.set_order(pi, gid, int(gid.max()) + 1)
    _check(ctx, items, pre, n, pi, gid, int(gid.max()) + 1, lens, presence=n <= 300_000)
    info = ctx.info()
    assert info.n_reruns == before and info.n_rows == 0 and info.n_spilled_last > 0


@pytest.mark.parametrize("sparse", ["0", "1"])
@pytest.mark.parametrize("n_paths,splits", [(180, 3), (252, 4), (400, 3), (1000, 7), (129, 2), (700, 1)])
def test_split_boundaries_anywhere_in_a_window_of_the_visiting_order(band, n_paths, splits, sparse):
    """many short paths (contigs), every path its own group or groups of a few: the visiting order is longer than one 64-entry
    window and the splits begin at any entry -- also at the last batch of a window (180 paths / 3 splits: entry 60), where
    the groups of the window being folded must not be those of the window the loads have moved on to.  sparse = "1": the
    workgroups walk per-band lists of the entries that have steps there (k_band_compact; what such a shape takes by itself),
    "0": the visiting order itself"""
    ctx = band
    rng = np.random.default_rng(n_paths)
    n = 300_000
    span = n // 12
    segs = []
    for p in range(n_paths):
        a = int(rng.integers(1, n - span + 2))
        ids = (a + np.flatnonzero(rng.random(span) < 0.3)).astype(np.uint64)
        segs.append(ids[::-1].copy() if p % 7 == 3 else ids)
    items, pre = _concat(segs)
    ctx.set_csr(items.astype(np.uint32), pre, n)
    before = ctx.info().n_reruns
    for group_size in (1, 3):
        order = rng.permutation(n_paths).astype(np.uint64)
        gid = (np.arange(n_paths) // group_size).astype(np.uint64)
        G = int(gid.max()) + 1
        ctx.set_order(order, gid, G)
        os.environ["PNX_BAND_SPLITS"] = str(splits)
        os.environ["PNX_BAND_SPARSE"] = sparse
        try:
            _check(ctx, items, pre, n, order, gid, G, presence=group_size == 3)
            assert ctx.info().band_splits == splits
        finally:
            del os.environ["PNX_BAND_SPLITS"]
            del os.environ["PNX_BAND_SPARSE"]
    assert ctx.info().n_reruns == before and ctx.info().n_rows == 0


@pytest.mark.parametrize("splits", [2, 3, 16])
def test_bands_shared_by_several_workgroups(band, splits):
    """a small graph: the visiting order is cut at group boundaries, the counters of a band's workgroups meet in the
    coverage vector and K2 takes the histogram from it"""
    ctx = band
    n, p = 150_000, 40
    items, pre, lens = orc.pansyn_rearranged(17, n, p)
    ctx.set_csr_pansyn_rearranged(17, n, p, with_weights=True)
    rng = np.random.default_rng(splits)
    order = rng.permutation(p).astype(np.uint64)
    gid = np.cumsum(rng.random(p) < 0.7).astype(np.uint64)
    gid -= gid[0]
    G = int(gid.max()) + 1
    ctx.set_order(order, gid, G)
    before = ctx.info().n_reruns
    os.environ["PNX_BAND_SPLITS"] = str(splits)
    try:
        _check(ctx, items, pre, n, order, gid, G, lens, presence=True)
        assert ctx.info().band_splits == min(splits, G)
    finally:
        del os.environ["PNX_BAND_SPLITS"]
    assert ctx.info().n_reruns == before and ctx.info().n_rows == 0


def test_band_route_rejects_bad_ids_at_upload(band):
    from panacus_amd import capi
    ctx = band
    n = 10_000
    items, pre, _ = orc.pansyn(9, n, 4)
    for bad_pos, bad_val in ((0, 0), (len(items) // 2, n + 1), (len(items) - 1, 0xFFFFFFFE)):
        bad = items.astype(np.uint32).copy()
        bad[bad_pos] = bad_val
        with pytest.raises(capi.PnxError) as e:
            ctx.set_csr(bad, pre, n)
        assert e.value.code == capi.PNX_EINVAL


def test_first_sweep_takes_the_steps_second_sweep_derives_the_rows():
    """the default policy on a shape the one-shot route fits (enough bands to fill the chip, long segments): nothing is
    derived for the first sweep; a caller that sweeps again gets the rows, and every later pass runs over them"""
    from panacus_amd import capi
    n, p = 5_000_000, 12
    items, pre, _ = orc.pansyn(11, n, p)
    pi = np.arange(p, dtype=np.uint64)
    with capi.Context(0) as c:
        c.set_csr_pansyn(11, n, p)
        c.set_order(pi, pi, p)
        cnt, h = c.hist()
        ocov, oh = _oracle_hist(items, pre, pi, pi, n, p)
        assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
        assert c.info().n_rows == 0 and c.info().n_reruns == 0
        gid = (pi // 3).astype(np.uint64)
        c.set_order(pi[::-1].copy(), gid, 4)
        cnt2, h2 = c.hist()
        ocov2, oh2 = _oracle_hist(items, pre, pi[::-1].copy(), gid, n, 4)
        assert np.array_equal(cnt2, ocov2) and np.array_equal(h2, oh2)
        assert c.info().n_rows == p * ((n + 1 + 2047) // 2048)
        # dropped derived data: the next sweep is a first sweep again
        c.config(capi.CFG_DROP_DERIVED, 0)
        c.set_order(pi, pi, p)
        cnt3, h3 = c.hist()
        assert np.array_equal(cnt3, ocov) and np.array_equal(h3, oh) and c.info().n_rows == 0
        # the same through an uploaded ItemTable (ids validated at upload, no rows derived there)
        c.set_csr(items.astype(np.uint32), pre, n)
        assert c.info().n_rows == 0
        c.set_order(pi, pi, p)
        cnt4, h4 = c.hist()
        assert np.array_equal(cnt4, ocov) and np.array_equal(h4, oh) and c.info().n_rows == 0
        # path rows only
        c.config(capi.CFG_COVER_ROUTE, 2)
        c.config(capi.CFG_DROP_DERIVED, 0)
        c.set_order(pi, pi, p)
        cnt5, h5 = c.hist()
        assert np.array_equal(cnt5, ocov) and np.array_equal(h5, oh) and c.info().n_rows > 0


def test_band_passes_in_flight(band):
    from panacus_amd import capi
    ctx = band
    n, p = 200_000, 10
    items, pre, _ = orc.pansyn(21, n, p)
    ctx.set_csr_pansyn(21, n, p)
    pi = np.arange(p, dtype=np.uint64)
    ctx.set_order(pi, pi, p)
    ocov, oh = _oracle_hist(items, pre, pi, pi, n, p)
    ctx.hist_async()
    ctx.hist_async()
    for _ in range(2):
        cnt, h = ctx.hist_fetch(want_countable=True)
        assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    assert ctx.info().n_rows == 0
