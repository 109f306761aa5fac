"""Pins the CPU oracle (oracle/) against the reference's own known-answer vectors
(tests/golden/, extracted by tests/golden/make_golden.py)."""
import itertools
import math
import os

import numpy as np
import pytest

import oracle as orc


def _abacus(gfa, count, group_mode=orc.GROUP_PATHID, group_file=None, order_file=None):
    g = orc.Graph(gfa, index_edges=(count == orc.EDGE))
    pi, gi, names = g.path_order(group_mode, group_file, order_file)
    items, pre = g.item_table(count)
    n = g.n_items(count)
    cov = orc.coverage(items, pre, pi, gi, n)
    return g, pi, gi, names, items, pre, cov


def test_cdbg_node_countable(golden, golden_dir):
    # src/graph_broker/abacus.rs:1424-1435
    g, pi, gi, names, items, pre, cov = _abacus(os.path.join(golden_dir, "cdbg.gfa"), orc.NODE)
    assert cov.tolist() == golden["cdbg_node"]["countable"]
    assert names == golden["cdbg_node"]["groups"]


@pytest.mark.parametrize("count,key", [(orc.NODE, "chrM_sample_node"), (orc.EDGE, "chrM_sample_edge"),
                                       (orc.BP, "chrM_sample_bp")])
def test_chrM_by_sample(golden, golden_dir, count, key):
    # src/graph_broker/abacus.rs:1487-1630 (countables, groups, hists)
    g, pi, gi, names, items, pre, cov = _abacus(os.path.join(golden_dir, "chrM_test.gfa"), count,
                                                orc.GROUP_SAMPLE)
    assert names == golden[key]["groups"]
    assert cov.tolist() == golden[key]["countable"]
    w = g.node_lens if count == orc.BP else None
    assert orc.hist(cov, len(names), w).tolist() == golden[key]["hist"]


def test_chrM_shape(golden_dir):
    g = orc.Graph(os.path.join(golden_dir, "chrM_test.gfa"), index_edges=True)
    assert (g.n_nodes, g.n_edges, g.n_paths) == (154, 205, 4)
    items, pre = g.item_table(orc.NODE)
    assert len(items) == 400
    eitems, epre = g.item_table(orc.EDGE)
    assert len(eitems) == 396


def test_t_groups_hist_and_info(golden, golden_dir):
    # tests/test_files/t_groups.hist.tsv; tests/info.rs:45-48; util.rs:1263-1274
    gfa = os.path.join(golden_dir, "t_groups.gfa")
    g, pi, gi, names, items, pre, cov = _abacus(gfa, orc.NODE)
    assert len(names) == 6
    assert orc.hist(cov, 6).tolist() == golden["t_groups_node_hist"]["hist"]
    # path x = ids 1,3,5,6,8,9,11,12,14,15
    assert items[pre[5]:pre[6]].tolist() == [1, 3, 5, 6, 8, 9, 11, 12, 14, 15]
    # -S: groups y and x each hold 10 nodes / 50 bp
    g2, pi, gi, names, items, pre, cov = _abacus(gfa, orc.BP, orc.GROUP_SAMPLE)
    assert sorted(names) == ["x", "y"]
    lens = g2.node_lens
    for grp in range(2):
        sel = np.zeros(g2.n_nodes + 1, bool)
        for k in np.nonzero(gi == grp)[0]:
            sel[items[pre[pi[k]]:pre[pi[k] + 1]]] = True
        assert sel.sum() == 10 and lens[sel].sum() == 50


def test_choose():
    # src/graph_broker/hist.rs:342-348
    assert abs(orc.choose(5, 0)) < 1e-10 and abs(orc.choose(5, 5)) < 1e-10
    assert abs(orc.choose(5, 1) - math.log2(5)) < 1e-10
    assert abs(orc.choose(5, 4) - math.log2(5)) < 1e-10
    assert abs(orc.choose(5, 2) - math.log2(10)) < 1e-10
    assert orc.choose(5, 6) == 0.0


def test_growth_known_answers_bit_exact(golden):
    # src/graph_broker/hist.rs:352-398 -- the reference uses assert_eq! on f64: exact equality
    ka = golden["growth_known_answers"]
    cov0 = (orc.ABSOLUTE, 0)
    assert orc.growth_branch("union", ka["union"]["hist"], cov0).tolist() == ka["union"]["expected"]
    assert orc.growth_branch("core", ka["core"]["hist"], cov0).tolist() == ka["core"]["expected"]
    assert orc.growth_branch("quorum", ka["quorum"]["hist"], cov0,
                             (orc.RELATIVE, ka["quorum"]["quorum"])).tolist() == ka["quorum"]["expected"]
    # dispatch (hist.rs:51-66): q=0 -> union, q=1 -> core
    assert orc.growth(ka["union"]["hist"], cov0, (orc.RELATIVE, 0.0)).tolist() == ka["union"]["expected"]
    assert orc.growth(ka["core"]["hist"], cov0, (orc.RELATIVE, 1.0)).tolist() == ka["core"]["expected"]


def test_chr22_report_660_values(golden):
    # docs/chr22.hprc-v1.0-pggb.histgrowth.html:266-276: hist -> floored growth, n = 44
    rep = golden["chr22_report"]
    total = 0
    for count in ("bp", "node", "edge"):
        h = rep["hists"][count]
        gr = rep["growths"][count]
        for c, q, curve in zip(gr["coverage"], gr["quorum"], gr["curves"]):
            got = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
            assert [int(math.floor(x)) for x in got] == curve, (count, c, q)
            total += len(curve)
    assert total == 660


def test_by_group_is_sorted_distinct(golden_dir):
    g, pi, gi, names, items, pre, cov = _abacus(os.path.join(golden_dir, "chrM_test.gfa"), orc.NODE,
                                                orc.GROUP_SAMPLE)
    r, c = orc.by_group(items, pre, pi, gi, g.n_nodes)
    assert r[1] == 0 and r[-1] == len(c)
    for i in range(1, g.n_nodes + 1):
        sl = c[r[i]:r[i + 1]]
        assert len(sl) == cov[i]
        assert np.all(np.diff(sl.astype(np.int64)) > 0)


def test_ordered_growth_survey_values(golden_dir):
    """Session-derived expectations recorded in SURVEY.md 8c (literal restatement of
    abacus.rs:989-1032; no reference-produced golden exists for ordered growth)."""
    gfa = os.path.join(golden_dir, "t_groups.gfa")
    g, pi, gi, names, items, pre, cov = _abacus(gfa, orc.NODE)
    r, c = orc.by_group(items, pre, pi, gi, g.n_nodes)
    G = len(names)
    assert orc.ordered_growth(r, c, G).tolist() == [2, 5, 8, 9, 10, 10]
    assert orc.ordered_growth(r, c, G, weights=g.node_lens).tolist() == [9, 14, 38, 39, 50, 50]
    assert orc.ordered_growth(r, c, G, quorum_thr=(orc.RELATIVE, 0.5)).tolist() == [2, 5, 5, 5, 5, 0]

    gfa = os.path.join(golden_dir, "chrM_test.gfa")
    g, pi, gi, names, items, pre, cov = _abacus(gfa, orc.NODE, orc.GROUP_SAMPLE)
    r, c = orc.by_group(items, pre, pi, gi, g.n_nodes)
    assert orc.ordered_growth(r, c, 4).tolist() == [89, 106, 140, 154]
    assert orc.ordered_growth(r, c, 4, coverage_thr=(orc.ABSOLUTE, 2)).tolist() == [87, 101, 115, 115]
    assert orc.ordered_growth(r, c, 4, quorum_thr=(orc.RELATIVE, 0.5)).tolist() == [89, 106, 106, 120]
    assert orc.ordered_growth(r, c, 4, weights=g.node_lens).tolist() == [16569, 17147, 17183, 17197]


def test_histgrowth_chrM_cfg1(golden_dir):
    # BASELINE cfg1: histgrowth chrM -c node -l 1 (group = path id): floor = [100,129,144,154]
    g, pi, gi, names, items, pre, cov = _abacus(os.path.join(golden_dir, "chrM_test.gfa"), orc.NODE)
    h = orc.hist(cov, len(names))
    gr = orc.growth(h, (orc.ABSOLUTE, 1), (orc.RELATIVE, 0.0))
    assert [int(math.floor(x)) for x in gr] == [100, 129, 144, 154]


def test_mean_over_all_orders_equals_union():
    """Cross-check: mean over all G! orders of ordered growth at q=0 equals the
    closed-form union growth for the same coverage threshold."""
    items, pre, lens = orc.pansyn(7, 60, 5)
    G = 5
    n = 60
    base_pi = np.arange(G, dtype=np.uint64)
    cov = orc.coverage(items, pre, base_pi, base_pi, n)
    for cthr in (1, 2):
        h = orc.hist(cov, G)
        union = orc.growth(h, (orc.ABSOLUTE, cthr), (orc.RELATIVE, 0.0))
        acc = np.zeros(G)
        cnt = 0
        for perm in itertools.permutations(range(G)):
            pi = np.asarray(perm, dtype=np.uint64)
            r, c = orc.by_group(items, pre, pi, np.arange(G, dtype=np.uint64), n)
            acc += orc.ordered_growth(r, c, G, coverage_thr=(orc.ABSOLUTE, cthr))
            cnt += 1
        np.testing.assert_allclose(acc / cnt, union, rtol=1e-9, atol=1e-9)


def test_order_file_and_group_file(golden_dir, tmp_path):
    gfa = os.path.join(golden_dir, "cdbg.gfa")
    g = orc.Graph(gfa)
    order = tmp_path / "order.txt"
    order.write_text("d#1#h1\nc#2#h1\na#1#h1\nb#1#h1\nc#1#h1\nc#1#h2\n")
    pi, gi, names = g.path_order(orc.GROUP_PATHID, None, str(order))
    assert names == ["d#1#h1", "c#2#h1", "a#1#h1", "b#1#h1", "c#1#h1", "c#1#h2"]
    assert pi.tolist() == [5, 4, 0, 1, 2, 3]
    # grouping by haplotype: a#1, b#1, c#1 (2 paths, contiguous), c#2, d#1
    pi, gi, names = g.path_order(orc.GROUP_HAPLOTYPE)
    assert names == ["a#1", "b#1", "c#1", "c#2", "d#1"]
    assert gi.tolist() == [0, 1, 2, 2, 3, 4]
