"""Path rows (csrc/kernels_rows.hip) -- the path x item presence table that one read of the steps derives per
upload, and the coverage pass over it -- against the CPU oracle (abacus.rs:719-787 restated), through the C ABI.
Integer work: bit-exact equality everywhere."""
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
def layout(request, ctx):
    from panacus_amd import capi
    ctx.config(capi.CFG_ROWS_LAYOUT, request.param)
    yield request.param
    ctx.config(capi.CFG_ROWS_LAYOUT, 0)


def _oracle_hist(items, pre, pi, gi, n, G, w=None, exclude=None):
    cov = orc.coverage(items, pre, pi, gi, n, exclude)
    return cov, orc.hist(cov, G, w)


def _awkward_graph(n, seed):
    """paths of every kind the reference accepts: sorted, descending, locally jittered, shuffled, every step doubled,
    empty, one step, a stride that leaves most item tiles untouched, a path with both the first and the last item"""
    rng = np.random.default_rng(seed)
    items, pre, lens = orc.pansyn(seed, n, 6)
    segs = [items[int(pre[k]):int(pre[k + 1])].copy() for k in range(6)]
    segs[1] = segs[1][::-1].copy()
    for s in range(0, len(segs[2]) - 50, 211):
        segs[2][s:s + 37] = segs[2][s:s + 37][::-1].copy()
    rng.shuffle(segs[3])
    segs[4] = np.repeat(segs[4], 2)
    segs.append(np.zeros(0, dtype=np.uint64))                                   # empty
    segs.append(np.array([n // 2 + 1], dtype=np.uint64))                         # one step
    segs.append(np.arange(1, n + 1, 5003, dtype=np.uint64))                      # sparse ascending
    segs.append(np.arange(n, 0, -4999, dtype=np.uint64))                         # sparse descending
    segs.append(np.array([1, n, 1, n], dtype=np.uint64))                         # both ends, repeated
    segs.append(rng.integers(1, n + 1, size=9000).astype(np.uint64))             # random ids with repeats
    pre2 = np.zeros(len(segs) + 1, dtype=np.uint64)
    pre2[1:] = np.cumsum([len(s) for s in segs])
    return np.concatenate(segs).astype(np.uint64), pre2, lens


@pytest.mark.parametrize("layout", [1, 2, 0], indirect=True)
@pytest.mark.parametrize("n", [1, 2047, 2048, 40_961, 300_000])
def test_rows_every_kind_of_path(ctx, layout, n):
    from panacus_amd import capi
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    items, pre, lens = _awkward_graph(n, 3 + n % 7)
    P = len(pre) - 1
    rng = np.random.default_rng(n)
    excl = (rng.random(n + 1) < 0.05).astype(np.uint8)
    excl[0] = 0
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=lens, exclude=excl)
    info = ctx.info()
    assert info.n_rows > 0 and (layout == 0 or info.rows_tile_major == (1 if layout == 1 else 0))
    for split in (0, 1, 2, 4, 8):
        ctx.config(capi.CFG_COVER_SPLIT, split)
        # every path its own group, file order
        pi = np.arange(P, dtype=np.uint64)
        ctx.set_order(pi, pi, P)
        cnt, h = ctx.hist()
        ocov, oh = _oracle_hist(items, pre, pi, pi, n, P, lens, excl)
        assert np.array_equal(cnt, ocov) and np.array_equal(h, oh), ("paths", split)
        # groups of three, visited backwards, two paths left out
        order = np.array([q for q in range(P - 1, -1, -1) if q not in (2, 7)], dtype=np.uint64)
        gid = (np.arange(len(order)) // 3).astype(np.uint64)
        G = int(gid.max()) + 1
        ctx.set_order(order, gid, G)
        cnt, h = ctx.hist()
        ocov, oh = _oracle_hist(items, pre, order, gid, n, G, lens, excl)
        assert np.array_equal(cnt, ocov) and np.array_equal(h, oh), ("groups", split)
    ctx.config(capi.CFG_COVER_SPLIT, 0)
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
    got, off, _ = ctx.get_csr()   # the steps themselves are untouched
    assert np.array_equal(got, items.astype(np.uint32)) and np.array_equal(off, pre)


def test_rows_many_short_paths_take_the_span_layout_and_skip_windows(ctx):
    """thousands of contig-like paths, each on one or two item tiles: the rows are laid out path-major over the tiles a
    path really spans, and a coverage wave skips the 64-entry windows of the order that do not reach its tile"""
    n, P = 400_000, 6000
    rng = np.random.default_rng(77)
    segs = []
    for k in range(P):
        a = int(rng.integers(1, n - 3000))
        ln = int(rng.integers(1, 2500))
        s = np.arange(a, a + ln, dtype=np.uint64)
        if k % 3 == 1:
            s = s[::-1].copy()
        if k % 5 == 2:
            s = s[rng.random(ln) < 0.4]
        segs.append(s)
    pre = np.zeros(P + 1, dtype=np.uint64)
    pre[1:] = np.cumsum([len(s) for s in segs])
    items = np.concatenate(segs)
    ctx.set_csr(items.astype(np.uint32), pre, n)
    info = ctx.info()
    assert info.rows_tile_major == 0 and 0 < info.n_rows <= 3 * P
    gid = (np.arange(P) // 40).astype(np.uint64)       # 150 groups of 40 paths
    pi = np.arange(P, dtype=np.uint64)
    G = int(gid.max()) + 1
    ctx.set_order(pi, gid, G)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, pi, gid, n, G)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    # the same paths, the groups dealt out differently (the order within a group is the library's to choose)
    perm = rng.permutation(P).astype(np.uint64)
    ctx.set_order(perm, gid, G)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, perm, gid, n, G)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)


@pytest.mark.parametrize("layout", [1, 2], indirect=True)
def test_rows_reject_bad_ids_and_stay_usable(ctx, layout):
    from panacus_amd import capi
    n = 10_000
    items, pre, _ = orc.pansyn(9, n, 4)
    for bad_pos, bad_val in ((0, 0), (len(items) // 2, n + 1), (len(items) - 1, 0xFFFFFFFE), (5, n + 5000)):
        bad = items.astype(np.uint32).copy()
        bad[bad_pos] = bad_val
        with pytest.raises(capi.PnxError) as e:
            ctx.set_csr(bad, pre, n)
        assert e.value.code == capi.PNX_EINVAL
        with pytest.raises(capi.PnxError):
            ctx.hist()                      # nothing is resident after a rejected upload
    ctx.set_csr(items.astype(np.uint32), pre, n)
    pi = np.arange(4, dtype=np.uint64)
    ctx.set_order(pi, pi, 4)
    cnt, h = ctx.hist()
    ocov, oh = _oracle_hist(items, pre, pi, pi, n, 4)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)


def test_rows_are_derived_once_and_again_on_request(ctx):
    from panacus_amd import capi
    n, p = 120_000, 12
    items, pre, _ = orc.pansyn(4, n, p)
    ctx.set_csr_pansyn(4, n, p)
    assert ctx.info().n_rows == 0             # a generated graph leaves the rows to the first pass (or pnx_prepare)
    ctx.prepare()
    rows = ctx.info().n_rows
    assert rows == p * ((n + 1 + 2047) // 2048) and ctx.info().rows_tile_major == 1
    pi = np.arange(p, dtype=np.uint64)
    ctx.set_order(pi, pi, p)
    ocov, oh = _oracle_hist(items, pre, pi, pi, n, p)
    cnt, h = ctx.hist()
    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    ctx.config(capi.CFG_DROP_DERIVED, 0)
    assert ctx.info().n_rows == 0
    ctx.set_order(pi, pi, p)
    cnt, h = ctx.hist()                       # derived again by the pass itself
    assert ctx.info().n_rows == rows and np.array_equal(cnt, ocov) and np.array_equal(h, oh)
    # the step routes and the rows give the same answer on the same resident graph
    ctx.config(capi.CFG_COVER_VARIANT, 2)
    try:
        ctx.set_order(pi, pi, p)
        cnt2, h2 = ctx.hist()
    finally:
        ctx.config(capi.CFG_COVER_VARIANT, 3)
    assert np.array_equal(cnt2, ocov) and np.array_equal(h2, oh)


def test_rows_are_shared_with_a_borrowing_context():
    from panacus_amd import capi
    n, p = 90_000, 10
    items, pre, _ = orc.pansyn(8, n, p)
    pi = np.arange(p, dtype=np.uint64)
    gid = (pi // 2).astype(np.uint64)
    with capi.Context(0) as a, capi.Context(0) as b:
        a.set_csr(items.astype(np.uint32), pre, n)
        b.share_csr(a)
        assert b.info().n_rows == a.info().n_rows > 0
        a.set_order(pi, pi, p)
        b.set_order(pi[::-1].copy(), gid, p // 2)
        a.hist_async()
        b.hist_async()
        ca, ha = a.hist_fetch(want_countable=True)
        cb, hb = b.hist_fetch(want_countable=True)
        oa = _oracle_hist(items, pre, pi, pi, n, p)
        ob = _oracle_hist(items, pre, pi[::-1].copy(), gid, n, p // 2)
        assert np.array_equal(ca, oa[0]) and np.array_equal(ha, oa[1])
        assert np.array_equal(cb, ob[0]) and np.array_equal(hb, ob[1])


def test_rows_long_path_many_chunks_and_wide_group_counts(ctx):
    """one path of > 16 chunks next to 299 short ones: a build wave walks several chunks, also across path ends;
    300 groups need the 12-plane counters, whose low planes come out of the carry-save tree"""
    n = 100_000
    rng = np.random.default_rng(12)
    segs = [np.arange(1, n + 1, dtype=np.uint64)]
    for k in range(299):
        a = int(rng.integers(1, n - 600))
        segs.append(np.arange(a, a + int(rng.integers(1, 600)), dtype=np.uint64))
    segs.insert(100, np.arange(n, 0, -1, dtype=np.uint64))
    P = len(segs)
    pre = np.zeros(P + 1, dtype=np.uint64)
    pre[1:] = np.cumsum([len(s) for s in segs])
    items = np.concatenate(segs)
    ctx.set_csr(items.astype(np.uint32), pre, n)
    for G, gid in ((P, np.arange(P)), (7, np.arange(P) * 7 // P), (1, np.zeros(P, dtype=np.int64)), (8, np.arange(P) * 8 // P)):
        gid = gid.astype(np.uint64)
        pi = np.arange(P, dtype=np.uint64)
        ctx.set_order(pi, gid, G)
        cnt, h = ctx.hist()
        ocov, oh = _oracle_hist(items, pre, pi, gid, n, G)
        assert np.array_equal(cnt, ocov) and np.array_equal(h, oh), G


@pytest.mark.parametrize("split", [0, 1, 2, 4, 8])
def test_histogram_added_by_the_coverage_kernel(ctx, split):
    """PNX_CFG_HIST_IN_COVER: the coverage kernel over rows adds the histogram itself (construct_hist / construct_hist_bps,
    abacus.rs:746-787) -- node counts from bit masks over the counter planes, bp through per-lane sums -- against the
    oracle and against the separate histogram kernel; every width of the split kernel, groups of 1, 2, 3, 64 and one per
    path, bins that are the hot ones (0, 1, G) or not, excluded nodes, a subset of the paths, a last tile that is not full"""
    from panacus_amd import capi
    n, p = 70_001, 48
    items, pre, lens = orc.pansyn(23, n, p)
    rng = np.random.default_rng(split)
    excl = (rng.random(n + 1) < 0.02).astype(np.uint8)
    try:
        ctx.config(capi.CFG_COVER_SPLIT, split)
        for w, ex in ((None, None), (lens, None), (lens, excl), (None, excl)):
            ctx.set_csr(items.astype(np.uint32), pre, n, weights=w, exclude=ex)
            for G, pi, gid in ((p, np.arange(p), np.arange(p)), (1, np.arange(p), np.zeros(p)), (2, np.arange(p), np.arange(p) % 2),
                               (3, np.arange(p), np.arange(p) * 3 // p), (16, np.arange(4, p - 3), (np.arange(4, p - 3) - 4) * 16 // (p - 7))):
                pi, gid = pi.astype(np.uint64), np.sort(gid.astype(np.uint64))
                ctx.set_order(pi, gid, G)
                got = {}
                for fused in (1, 0):
                    ctx.config(capi.CFG_HIST_IN_COVER, fused)
                    got[fused] = ctx.hist()
                ocov = orc.coverage(items, pre, pi, gid, n, ex)
                oh = orc.hist(ocov, G, w)
                for fused in (1, 0):
                    assert np.array_equal(got[fused][0], ocov) and np.array_equal(got[fused][1], oh), (G, fused, w is not None, ex is not None)
    finally:
        ctx.config(capi.CFG_COVER_SPLIT, 0)
        ctx.config(capi.CFG_HIST_IN_COVER, 1)


def test_histogram_with_more_bins_than_lds_holds(ctx):
    """more than 4095 groups: the bins do not fit a workgroup's LDS, the separate histogram kernel takes over (24-plane counters)"""
    n, p = 30_000, 5000
    rng = np.random.default_rng(3)
    lens_p = rng.integers(1, 40, size=p)
    pre = np.zeros(p + 1, dtype=np.uint64)
    pre[1:] = np.cumsum(lens_p)
    items = np.concatenate([np.sort(rng.integers(1, n + 1, size=k)) for k in lens_p]).astype(np.uint64)
    ctx.set_csr(items.astype(np.uint32), pre, n)
    pi = np.arange(p, dtype=np.uint64)
    for G, gid in ((p, pi), (4095, pi * 4095 // p), (4096, pi * 4096 // p)):
        ctx.set_order(pi, gid.astype(np.uint64), G)
        cnt, h = ctx.hist()
        ocov = orc.coverage(items, pre, pi, gid.astype(np.uint64), n)
        assert np.array_equal(cnt, ocov) and np.array_equal(h, orc.hist(ocov, G)), G


def test_profile_sampling_times_every_nth_launch(ctx):
    """pnx_profile_sample: with `every` = 3, seven passes leave three timed launches of the selected kernel (the 1st, 4th and
    7th); the default times every launch; 0 is refused"""
    from panacus_amd import capi
    n, p = 30_000, 12
    ctx.set_csr_pansyn(3, n, p)
    pi = np.arange(p, dtype=np.uint64)
    ctx.set_order(pi, pi, p)
    ctx.hist()
    try:
        ctx.profile_enable(True)
        ctx.profile_select([capi.K_COVER])
        ctx.profile_sample(3)
        ctx.profile_reset()
        for _ in range(7):
            ctx.hist()
        ms, launches = ctx.profile_read()["cover"]
        assert launches == 3 and ms > 0
        ctx.profile_sample(1)
        ctx.profile_reset()
        for _ in range(4):
            ctx.hist()
        assert ctx.profile_read()["cover"][1] == 4
        with pytest.raises(capi.PnxError):
            ctx.profile_sample(0)
    finally:
        ctx.profile_sample(1)
        ctx.profile_select(None)
        ctx.profile_reset()
        ctx.profile_enable(False)


@pytest.mark.parametrize("n", [2047, 2048, 70_001, 300_000])
def test_four_rows_per_load(ctx, n):
    """PNX_CFG_ROWS_KERNEL 2: k_rows_cover_q -- the four quarters of a wave walk the four parts of the visiting order, one 16-byte
    load per lane fetches four rows, the counters of the quarters are added up at the end -- against the oracle and against
    k_rows_cover (1): groups of every size (so that the quarters' groups end at different slots), parts of unequal length, a
    subset of the paths, paths that miss tiles (rows that are not there), excluded items, bp weights, a last tile that is
    not full"""
    from panacus_amd import capi
    p = 53
    items, pre, lens = orc.pansyn(31, n, p)
    rng = np.random.default_rng(n)
    # some paths only visit a range of the ids: tiles without a row of theirs
    segs = []
    for k in range(p):
        s = items[int(pre[k]):int(pre[k + 1])]
        if k % 5 == 1:
            s = s[(s > n // 3) & (s < 2 * n // 3)]
        elif k % 7 == 2:
            s = s[:0]
        segs.append(np.sort(s))
    items = np.concatenate(segs).astype(np.uint64)
    pre = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.uint64)
    excl = (rng.random(n + 1) < 0.03).astype(np.uint8)
    excl[0] = 0
    try:
        ctx.config(capi.CFG_COVER_ROUTE, 2)   # path rows
        ctx.config(capi.CFG_ROWS_LAYOUT, 1)   # ... tile-major: what the kernel reads
        for w, ex in ((None, None), (lens, excl)):
            ctx.set_csr(items.astype(np.uint32), pre, n, weights=w, exclude=ex)
            orders = [(np.arange(p), np.arange(p)),
                      (np.arange(p), np.sort(rng.integers(0, 9, size=p))),            # groups of random sizes
                      (np.arange(p), np.repeat(np.arange(4), [1, 2, 10, 40])),          # four groups, very unequal
                      (np.arange(3, p - 2), np.arange(p - 5) // 3),
                      (rng.permutation(p)[:30], np.sort(rng.integers(0, 12, size=30)))]
            for pi, gid in orders:
                pi = pi.astype(np.uint64)
                gid = gid.astype(np.uint64)
                gid = np.unique(gid, return_inverse=True)[1].astype(np.uint64)           # 0 .. G - 1 without gaps
                G = int(gid.max()) + 1
                ctx.set_order(pi, gid, G)
                ocov = orc.coverage(items, pre, pi, gid, n, ex)
                oh = orc.hist(ocov, G, w)
                for kern in (2, 1):
                    ctx.config(capi.CFG_ROWS_KERNEL, kern)
                    before = ctx.info().n_rows_q_passes
                    cnt, h = ctx.hist()
                    assert ctx.info().n_rows_q_passes == before + (1 if kern == 2 and G >= 4 else 0), (kern, G)
                    assert np.array_equal(cnt, ocov) and np.array_equal(h, oh), (kern, G, w is not None)
    finally:
        ctx.config(capi.CFG_ROWS_KERNEL, 0)
        ctx.config(capi.CFG_ROWS_LAYOUT, 0)
        ctx.config(capi.CFG_COVER_ROUTE, 0)
