"""Subset / exclude lists with BED coordinates in the oracle and the host front end.

PARITY UNPINNED: the reference ships the BED inputs (test/bed_chrM, copied as data into
tests/golden/bed_chrM) but no expected output for them.  What pins the restatement here:
  * a small graph whose expected tables are derived BY HAND from the reference's code
    (graph_broker/util.rs:569-795, src/util.rs:118-310, abacus.rs:779-785, 1187-1229);
  * structural properties on the chrM fixtures (whole-path lists == the unmasked tables, bp
    and node totals are conserved, coordinates that cover a path == the path);
  * the host C++ implementation, written separately, agreeing with the oracle (also fuzzed in
    test_host_gfa_fuzz.py).
"""
import os

import numpy as np
import pytest

import oracle as orc
from panacus_amd import hostlib as hl

TINY = "\n".join([
    "H\tVN:Z:1.0",
    "S\tA\t" + "A" * 10, "S\tB\t" + "C" * 5, "S\tC\t" + "G" * 8,   # node length = sequence length
    "L\tA\t+\tB\t+\t0M", "L\tB\t+\tC\t+\t0M", "L\tA\t+\tC\t-\t0M",
    "P\tp1\tA+,B+,C+\t*",     # A = [0,10), B = [10,15), C = [15,23)
    "P\tp2\tA+,C-\t*",        # A = [0,10), C = [10,18) walked backwards
]) + "\n"


def _hists(g, ct, sf, ef, mode=orc.GROUP_PATHID):
    pi, gi, names = g.path_order(mode, None, None, sf, ef)
    items, pre, fl, ids, bps = g.masked_table(ct, sf, ef)
    n = g.n_items(ct)
    cov = orc.coverage(items, pre, pi, gi, n, fl if ef else None)
    h = orc.hist(cov, len(names), g.node_lens if ct == orc.BP else None)
    return items, pre, fl, ids, bps, cov, orc.hist_apply_uncovered(cov, ids, bps, h), names


@pytest.fixture()
def tiny(tmp_path):
    p = tmp_path / "tiny.gfa"
    p.write_text(TINY)
    return str(p), tmp_path


def test_tiny_subset_interval_by_hand(tiny):
    gfa, d = tiny
    g = orc.Graph(gfa, index_edges=True)
    (d / "s.bed").write_text("p1\t3\t12\n")
    sf = str(d / "s.bed")
    # p1 3..12 meets A in [3,10) and B in [0,2): both pushed, both only partly covered; p2 is not
    # in the subset and gets an empty entry; the subset list is the order source: one group
    items, pre, fl, ids, bps, cov, h, names = _hists(g, orc.NODE, sf, None)
    assert names == ["p1"] and items.tolist() == [1, 2] and pre.tolist() == [0, 2, 2]
    assert cov.tolist()[1:] == [1, 1, 0] and h.tolist() == [1, 2]
    items, pre, fl, ids, bps, cov, h, names = _hists(g, orc.BP, sf, None)
    assert ids.tolist() == [1, 2] and bps.tolist() == [3, 3]       # 10 - 7, 5 - 2
    assert h.tolist() == [8 + 6, 15 - 6]
    # edges: A>B sits at B = [10,15): meets [3,12); B>C sits at C = [15,23): does not
    items, pre, fl, ids, bps, cov, h, names = _hists(g, orc.EDGE, sf, None)
    assert items.tolist() == [1] and h.tolist() == [2, 1]


def test_tiny_exclude_interval_by_hand(tiny):
    gfa, d = tiny
    g = orc.Graph(gfa, index_edges=True)
    (d / "e.bed").write_text("p2\t5\t12\n")
    ef = str(d / "e.bed")
    # the interval touches A ([5,10) of it) and C ([0,2) along the walk = [6,8) of the node):
    # node counts drop both nodes; bp counts keep them (neither is excluded completely)
    items, pre, fl, ids, bps, cov, h, names = _hists(g, orc.NODE, None, ef)
    assert names == ["p1", "p2"] and fl.tolist() == [0, 1, 0, 1]
    assert cov.tolist()[1:] == [0, 1, 0] and h.tolist() == [2, 1, 0]
    items, pre, fl, ids, bps, cov, h, names = _hists(g, orc.BP, None, ef)
    assert fl.sum() == 0 and len(ids) == 0 and h.tolist() == [0, 5, 18]
    # two intervals that together cover C completely exclude it for bp too
    (d / "e2.bed").write_text("p2\t10\t14\np2\t14\t18\n")
    items, pre, fl, ids, bps, cov, h, names = _hists(g, orc.BP, None, str(d / "e2.bed"))
    assert fl.tolist() == [0, 0, 0, 1] and h.tolist() == [8, 5, 10]
    # edge A>C- of p2 sits at C = [10,18): excluded; p1's edges are untouched
    items, pre, fl, ids, bps, cov, h, names = _hists(g, orc.EDGE, None, ef)
    assert fl.tolist() == [0, 0, 0, 1] and h.tolist() == [1, 2, 0]


def test_tiny_whole_path_in_exclude_interval(tiny):
    gfa, d = tiny
    g = orc.Graph(gfa, index_edges=True)
    # an exclude entry without coordinates removes the path from the order and flags all its nodes
    (d / "e.bed").write_text("p2\n")
    items, pre, fl, ids, bps, cov, h, names = _hists(g, orc.BP, None, str(d / "e.bed"))
    assert names == ["p1"] and fl.tolist() == [0, 1, 0, 1] and h.tolist() == [18, 5]


@pytest.mark.parametrize("ct", [orc.NODE, orc.BP, orc.EDGE])
def test_chrM_bed_fixtures_properties(golden_dir, ct):
    gfa = os.path.join(golden_dir, "chrM_test.gfa")
    bed = os.path.join(golden_dir, "bed_chrM")
    g = orc.Graph(gfa, index_edges=True)
    base = _hists(g, ct, None, None)
    total = int(base[6].sum())
    # a list of all paths without coordinates is the unmasked graph, in list order
    full = _hists(g, ct, os.path.join(bed, "inclusion.bed1"), None)
    assert np.array_equal(full[0], base[0]) and np.array_equal(full[1], base[1])
    assert sorted(full[7]) == sorted(base[7]) and full[6].tolist() == base[6].tolist()
    # intervals: every item lands in exactly one bin, so node / bp / edge totals are conserved
    for sf, ef in (("inclusion.bed3", None), (None, "exclusion.bed3"), ("inclusion.bed3", "exclusion.bed3"),
                   ("inclusion_sub.bed1", None), ("inclusion_chm13.bed1", "exclusion.bed3")):
        r = _hists(g, ct, sf and os.path.join(bed, sf), ef and os.path.join(bed, ef))
        assert int(r[6].sum()) == total, (sf, ef)
        assert len(r[0]) <= len(base[0])
    # session-derived values for the record (oracle restatement = definition, not a golden vector)
    r = _hists(g, ct, os.path.join(bed, "inclusion.bed3"), None)
    want = {orc.NODE: [3, 51, 54, 46, 0], orc.BP: [3, 629, 544, 16021, 0], orc.EDGE: [7, 104, 93, 1, 0]}[ct]
    assert r[6].tolist() == want


@pytest.mark.parametrize("ct", [hl.NODE, hl.BP, hl.EDGE])
def test_host_matches_oracle_on_bed_fixtures(golden_dir, tiny, ct):
    bed = os.path.join(golden_dir, "bed_chrM")
    cases = [(os.path.join(golden_dir, "chrM_test.gfa"), sf and os.path.join(bed, sf), ef and os.path.join(bed, ef))
             for sf, ef in (("inclusion.bed3", None), (None, "exclusion.bed3"), ("inclusion.bed3", "exclusion.bed3"),
                            ("inclusion_sub.bed1", "exclusion.bed3"))]
    gfa, d = tiny
    (d / "s.bed").write_text("p1\t3\t12\np2\t0\t4\n")
    (d / "e.bed").write_text("p2\t5\t12\n")
    cases += [(gfa, str(d / "s.bed"), None), (gfa, None, str(d / "e.bed")), (gfa, str(d / "s.bed"), str(d / "e.bed"))]
    for path, sf, ef in cases:
        a, b = hl.GfaGraph(path, index_edges=True), orc.Graph(path, index_edges=True)
        po, ph = b.path_order(orc.GROUP_PATHID, None, None, sf, ef), a.path_order(hl.GROUP_PATHID, None, None, sf, ef)
        assert po[2] == ph[2] and np.array_equal(po[0], ph[0])
        x, y = b.masked_table(ct, sf, ef), a.masked_table(ct, sf, ef)
        for u, v in zip(x, y):
            assert np.array_equal(np.asarray(u, dtype=np.uint64), np.asarray(v, dtype=np.uint64)), (path, sf, ef)


def test_rows_with_start_beyond_end(golden_dir):
    """BED rows with start > end inside one node (found by a soak run, tests/golden/bed_inverted/README.md): the exclude
    side drops such a piece (activate_n_annotate, src/util.rs:163-170), the include side hands it to
    IntervalContainer::add as it is -- host walk == oracle, and the uncovered bp are the ones the soak run disagreed on"""
    import numpy as np
    from panacus_amd import hostlib as hl
    base = os.path.join(golden_dir, "bed_inverted")
    expect = {"6cd09061": (orc.GROUP_PATHID, [10], [7]), "8438eb63": (orc.GROUP_SAMPLE, [5, 53], [12, 6])}
    for d, (mode, ids, bps) in expect.items():
        gfa, sf, ef = (os.path.join(base, d, f) for f in ("r.gfa", "subset.bed", "exclude.bed"))
        g = orc.Graph(gfa, index_edges=True)
        hg = hl.GfaGraph(gfa, index_edges=True)
        g.path_order(mode, None, None, sf, ef)
        for ct in (orc.NODE, orc.BP, orc.EDGE):
            a, b = g.masked_table(ct, sf, ef), hg.masked_table(ct, sf, ef, mode)
            for x, y in zip(a, b):
                assert np.array_equal(np.asarray(x).astype(np.uint64), np.asarray(y).astype(np.uint64)), (d, ct)
            if ct == orc.BP:
                assert a[3].tolist() == ids and a[4].tolist() == bps


def test_walked_path_drops_leading_zero_length_nodes(tmp_path):
    """update_tables keeps a node only if an include interval starts BEFORE its end (`include_coords[i].0 < p + l`,
    graph_broker/util.rs:625): with an exclude list that does not name a path, the path is walked against the whole-path
    interval (0, usize::MAX) and a zero-length node at coordinate 0 is dropped -- unlike a path taken whole (no lists)"""
    import numpy as np
    from panacus_amd import hostlib as hl
    gfa = tmp_path / "z.gfa"
    gfa.write_text("H\tVN:Z:1.0\nS\ta\t\nS\tb\tACGT\nS\tc\t\nP\tp1\ta+,b+,c+\t*\nP\tp2\tb+\t*\n")
    ex = tmp_path / "e.bed"
    ex.write_text("p2\n")
    g = orc.Graph(str(gfa))
    hg = hl.GfaGraph(str(gfa))
    assert g.node_lens.tolist() == [0, 0, 4, 0] and np.array_equal(hg.node_lens, g.node_lens)
    assert g.item_table(orc.NODE)[0].tolist() == [1, 2, 3, 2]       # no lists: every step
    g.path_order(orc.GROUP_PATHID, None, None, None, str(ex))
    for ct in (orc.NODE, orc.BP):
        o, h = g.masked_table(ct, None, str(ex)), hg.masked_table(ct, None, str(ex))
        assert o[0].tolist() == [2, 3, 2] and o[1].tolist() == [0, 2, 3]  # a (length 0 at coordinate 0) is gone, c stays
        assert np.array_equal(h[0].astype(np.uint64), o[0].astype(np.uint64)) and np.array_equal(h[1], o[1])
