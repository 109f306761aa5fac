"""Randomised differential test of the whole device path against the oracle: random graphs with
sorted, reversed, locally shuffled and fully shuffled paths, duplicate steps, empty paths, random
grouping, partial visiting orders, exclusions and weights -- histogram, coverage vector, presence
matrix, ordered growth and pair intersections must match bit for bit."""
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


def _random_graph(rng, n, p):
    paths = []
    for k in range(p):
        kind = rng.integers(0, 8)
        if kind == 0:
            ids = np.zeros(0, dtype=np.int64)  # empty path
        else:
            # a contiguous stretch of the id space, or all of it
            if rng.random() < 0.5:
                a = int(rng.integers(1, n + 1))
                b = int(rng.integers(a, min(n, a + max(1, n // 3)) + 1))
            else:
                a, b = 1, n
            dens = rng.choice([0.02, 0.3, 0.9])
            ids = a + np.flatnonzero(rng.random(b - a + 1) < dens)
            if kind == 1:
                ids = ids[::-1]                                  # descending
            elif kind == 2 and len(ids) > 8:                     # local disorder (bubbles): near-monotone
                for _ in range(max(1, len(ids) // 200)):
                    i = int(rng.integers(0, len(ids) - 4))
                    j = min(len(ids), i + int(rng.integers(2, 40)))
                    ids[i:j] = ids[i:j][::-1]
            elif kind == 3:
                ids = rng.permutation(ids)                       # edge-like: random order
            elif kind == 4 and len(ids):
                ids = np.repeat(ids, rng.integers(1, 3, size=len(ids)))  # duplicates in place
            elif kind == 5 and len(ids) > 2:
                ids = np.concatenate([ids, ids[: len(ids) // 3]])  # returns to the start: not monotone
        paths.append(ids.astype(np.uint64))
    pre = np.zeros(p + 1, dtype=np.uint64)
    pre[1:] = np.cumsum([len(x) for x in paths])
    items = np.concatenate(paths) if pre[-1] else np.zeros(0, dtype=np.uint64)
    return items, pre


import os

N_SEEDS = int(os.environ.get("PANACUS_FUZZ_SEEDS", "64"))  # more seeds for a soak run


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_graph_matches_oracle(ctx, seed):
    from panacus_amd import capi
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 37, 2047, 2048, 2049, 5000, 23_000, 70_000]))
    p = int(rng.choice([1, 2, 5, 17, 64, 200, 300]))
    items, pre = _random_graph(rng, n, p)
    weighted = bool(rng.integers(0, 2))
    w = None
    if weighted:
        w = rng.integers(1, 200_000 if seed % 3 == 0 else 3000, size=n + 1).astype(np.uint32)
        w[0] = 0
    excl = None
    if rng.random() < 0.4:
        excl = (rng.random(n + 1) < 0.05).astype(np.uint8)
    # random grouping of the paths, then the reference's order: groups contiguous at first visit
    n_g0 = int(rng.integers(1, p + 1))
    grp_of_path = rng.integers(0, n_g0, size=p)
    visit = rng.permutation(p)
    if rng.random() < 0.3:
        visit = visit[: max(1, p // 2)]  # subset of the paths
    seen, order_paths, order_groups = {}, [], []
    for q in visit:
        g = int(grp_of_path[q])
        if g in seen:
            continue
        seen[g] = len(seen)
        for q2 in visit:
            if int(grp_of_path[q2]) == g:
                order_paths.append(int(q2))
                order_groups.append(seen[g])
    pi = np.array(order_paths, dtype=np.uint64)
    gi = np.array(order_groups, dtype=np.uint64)
    G = len(seen)
    ctx.config(capi.CFG_TILE_BLOCKS, 1 + seed % 2 if seed % 5 == 0 else 1)
    ctx.config(capi.CFG_INDEX_BY_ENTRY, seed % 3)  # automatic / per entry / path-major index kernels
    ctx.config(capi.CFG_COVER_SKIP, 1 if seed % 2 else 0)  # window skipping in the plain hist pass
    ctx.config(capi.CFG_COVER_VARIANT, 2 if seed % 4 == 3 else 3)  # over the steps (round 2's routes) / over path rows
    ctx.config(capi.CFG_ROWS_LAYOUT, seed % 3)             # automatic / tile-major / path-major rows
    ctx.config(capi.CFG_HIST_IN_COVER, 0 if seed % 7 == 2 else 1)  # histogram by its own kernel / by the coverage kernel over rows
    ctx.config(capi.CFG_COVER_SPLIT, (0, 1, 2, 4, 8)[seed % 5])
    ctx.set_csr(items.astype(np.uint32), pre, n, weights=w, exclude=excl)
    ctx.set_order(pi, gi, G)
    cnt, h = ctx.hist()
    cov = orc.coverage(items, pre, pi, gi, n, excl)
    assert np.array_equal(np.asarray(cov, dtype=np.uint32), cnt), "coverage vector"
    assert h.tolist() == orc.hist(cov, G, w).tolist(), "histogram"
    # presence matrix + intersections
    r, c = orc.by_group(items, pre, pi, gi, n, excl)
    exp_inter = None
    try:
        exp_inter, _, _ = orc.similarity(r, c, G, node_lens=w)
    except KeyError:
        pass  # a group without items: the reference panics, the device still returns the sums
    got_inter = ctx.group_intersections()
    if exp_inter is not None:
        assert (got_inter == exp_inter).all(), "pair intersections"
    bits = ctx.presence()
    got_rows = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, 1: n + 1]
    assert (got_rows.T == (orc.table_rows(r, c, G) != 0)).all(), "presence matrix"
    # ordered growth for the identity order and one random order
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5), (1, float(rng.random()))]
    cv = [coverage_abs(Threshold(ABSOLUTE, a), G) for a, _ in pairs]
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), G) for _, q in pairs])
    perm = rng.permutation(G).astype(np.uint32)
    out = ctx.ordered_growth(cv, qt, np.stack([np.arange(G, dtype=np.uint32), perm]))
    for ri, pm in enumerate((np.arange(G), perm)):
        # relabel the groups by their rank in the order and ask the oracle for the identity order
        # the oracle wants the groups visited in rank order (ids ascending along the order)
        pi2, gi2 = [], []
        for rank, g in enumerate(pm):
            sel = pi[gi == g]
            pi2.extend(sel.tolist())
            gi2.extend([rank] * len(sel))
        rr, cc = orc.by_group(items, pre, np.array(pi2, np.uint64), np.array(gi2, np.uint64), n, excl)
        for t, (a, q) in enumerate(pairs):
            exp = orc.ordered_growth(rr, cc, G, (orc.ABSOLUTE, a), (orc.RELATIVE, q), w)
            assert out[ri, t].tolist() == [int(x) for x in exp], ("growth", ri, a, q)
    ctx.config(capi.CFG_TILE_BLOCKS, 1)
    ctx.config(capi.CFG_INDEX_BY_ENTRY, 0)
    ctx.config(capi.CFG_COVER_SKIP, 0)
    ctx.config(capi.CFG_COVER_VARIANT, 3)
    ctx.config(capi.CFG_ROWS_LAYOUT, 0)
    ctx.config(capi.CFG_HIST_IN_COVER, 1)
    ctx.config(capi.CFG_COVER_SPLIT, 0)
