"""CLI surface (panacus_amd/host/commands.cpp, tables.cpp): TSV shapes of SURVEY.md Appendix B,
threshold parsing, growth from a hist TSV (host only), and the GPU commands (marked gpu)."""
import math
import os

import numpy as np
import pytest

import oracle as orc
from panacus_amd import hostlib as hl
from panacus_amd import thresholds as th


def _body(text):
    """table without the '#' comment lines"""
    return "\n".join(l for l in text.split("\n") if not l.startswith("#"))


def test_format_f64_like_rust():
    for x, s in [(0.0, "0"), (1.0, "1"), (0.5, "0.5"), (0.1, "0.1"), (1e-5, "0.00001"), (123456789.0, "123456789"),
                 (1e21, "1000000000000000000000"), (2.5e-7, "0.00000025"), (float("nan"), "NaN"), (-3.0, "-3"),
                 (0.30000000000000004, "0.30000000000000004"), (16569.0, "16569")]:
        assert hl.format_f64(x) == s
        assert th.format_f64(x) == s


def test_format_f32_like_rust():
    # Rust prints the shortest decimal that round-trips as f32
    for x, s in [(1.0, "1"), (0.5, "0.5"), (np.float32(1) / np.float32(3), "0.33333334"), (np.float32(0.1), "0.1"),
                 (np.float32(2) / np.float32(3), "0.6666667"), (np.float32(1e-7), "0.0000001"),
                 (np.float32(16777216.0), "16777216"), (float("nan"), "NaN"), (0.0, "0")]:
        assert hl.format_f32(x) == s
    rng = np.random.default_rng(3)
    for v in rng.random(200).astype(np.float32):
        t = hl.format_f32(v)
        assert np.float32(t) == v and "e" not in t
        # no shorter decimal round-trips
        assert np.float32(t[:-1]) != v or t[-2] == "."


def test_threshold_container():
    tc = th.ThresholdContainer.parse_params("0,0.5,1.0", "1")
    assert [t.value for t in tc.coverage] == [1, 1, 1]
    assert [t.get_string() for t in tc.quorum] == ["0", "0.5", "1"]
    with pytest.raises(ValueError):
        th.ThresholdContainer.parse_params("0,0.5", "1,2,3")
    with pytest.raises(ValueError):
        th.ThresholdContainer.parse_params("1.5", "1")
    with pytest.raises(ValueError):
        th.ThresholdContainer.parse_params("0", "0.5")


def test_growth_from_hist_tsv(golden_dir):
    # src/lib.rs:160-190: `growth hist.tsv` never touches the graph (or the GPU)
    rc, out, err = hl.run_cli(["growth", os.path.join(golden_dir, "t_groups.hist.tsv"), "-l", "1,2", "-q", "0,0.5", "-a"])
    assert rc == 0, err
    lines = out.split("\n")
    assert lines[0].startswith("# target/debug/panacus hist")  # comment of the input is kept
    assert lines[1].startswith("# panacus-amd growth")
    assert lines[2:6] == ["panacus\thist\tgrowth\tgrowth", "count\tnode\tnode\tnode", "coverage\t\t1\t2", "quorum\t\t0\t0.5"]
    hist = [5, 0, 10, 0, 0, 0, 0]
    exp = [orc.growth(hist, (orc.ABSOLUTE, c), (orc.RELATIVE, q)) for c, q in ((1, 0.0), (2, 0.5))]
    assert lines[6] == "0\t5\tNaN\tNaN"
    for i in range(1, 7):
        cells = lines[6 + i].split("\t")
        assert cells == [str(i), str(hist[i])] + [hl.format_f64(math.floor(e[i - 1])) for e in exp]
    assert out.endswith("\n\n")  # writeln!(out, "{table}")


def test_growth_from_hist_errors(tmp_path, golden_dir):
    rc, out, err = hl.run_cli(["growth", os.path.join(golden_dir, "t_groups.hist.tsv"), "-S"])
    assert rc == 1 and "graph mode" in err
    bad = tmp_path / "x.tsv"
    bad.write_text("foo\tbar\n1\t2\n")
    rc, out, err = hl.run_cli(["growth", str(bad)])
    assert rc == 1
    rc, out, err = hl.run_cli(["nosuch", "x"])
    assert rc == 1 and "unknown subcommand" in err
    rc, out, err = hl.run_cli(["hist", "-s", "x", "y.gfa"])
    assert rc == 1 and "cannot open" in err


def test_gpu_commands_fail_loudly_without_gpu(golden_dir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rc, out, err = hl.run_cli(["hist", os.path.join(golden_dir, "chrM_test.gfa")])
    assert rc == 1 and "no CPU fallback" in err


# --------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_cli_hist_chrM(golden_dir):
    # SURVEY.md Appendix B
    rc, out, err = hl.run_cli(["hist", "-S", os.path.join(golden_dir, "chrM_test.gfa")])
    assert rc == 0, err
    assert out.startswith("# panacus-amd hist -S ")
    assert _body(out) == "panacus\thist\ncount\tnode\n\t\n\t\n0\t0\n1\t39\n2\t29\n3\t41\n4\t45\n\n"
    rc, out, err = hl.run_cli(["hist", "-S", "-c", "all", os.path.join(golden_dir, "chrM_test.gfa")])
    assert rc == 0, err
    rows = _body(out).split("\n")
    assert rows[0] == "panacus\thist\thist\thist" and rows[1] == "count\tnode\tbp\tedge"
    assert rows[4:9] == ["0\t0\t0\t0", "1\t39\t616\t80", "2\t29\t31\t59", "3\t41\t601\t66", "4\t45\t15949\t0"]


@pytest.mark.gpu
def test_cli_histgrowth_chrM(golden_dir):
    gfa = os.path.join(golden_dir, "chrM_test.gfa")
    rc, out, err = hl.run_cli(["histgrowth", "-a", "-S", "-l", "1", "-q", "0", gfa])
    assert rc == 0, err
    assert _body(out) == ("panacus\thist\tgrowth\ncount\tnode\tnode\ncoverage\t\t1\nquorum\t\t0\n"
                          "0\t0\tNaN\n1\t39\t100\n2\t29\t129\n3\t41\t144\n4\t45\t154\n\n")
    # BASELINE cfg1: histgrowth chrM -c node -l 1 (group = path id)
    rc, out, err = hl.run_cli(["histgrowth", "-c", "node", "-l", "1", gfa])
    assert rc == 0, err
    assert [r.split("\t")[1] for r in _body(out).split("\n")[5:9]] == ["100", "129", "144", "154"]
    # integrated_test.R grid: -S|-H x node|edge x -q 0,0.5,1.0 -l 0,1,2 against the oracle
    for grp, gm in (("-S", orc.GROUP_SAMPLE), ("-H", orc.GROUP_HAPLOTYPE)):
        for cname, ct in (("node", orc.NODE), ("edge", orc.EDGE), ("bp", orc.BP)):
            rc, out, err = hl.run_cli(["histgrowth", grp, "-c", cname, "-q", "0,0.5,1.0", "-l", "0,1,2", "-a", gfa])
            assert rc == 0, err
            g = orc.Graph(gfa, index_edges=True)
            pi, gi, names = g.path_order(gm)
            items, pre = g.item_table(ct)
            cov = orc.coverage(items, pre, pi, gi, g.n_items(ct))
            h = orc.hist(cov, len(names), g.node_lens if ct == orc.BP else None)
            rows = [r.split("\t") for r in _body(out).split("\n")[4:] if r]
            assert [int(r[1]) for r in rows] == h.tolist()
            for k, (c, q) in enumerate(((0, 0.0), (1, 0.5), (2, 1.0))):
                exp = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
                assert [r[2 + k] for r in rows[1:]] == [hl.format_f64(math.floor(x)) for x in exp]


@pytest.mark.gpu
def test_cli_ordered_histgrowth_chrM(golden_dir, tmp_path):
    gfa = os.path.join(golden_dir, "chrM_test.gfa")
    rc, out, err = hl.run_cli(["ordered-histgrowth", "-S", gfa])
    assert rc == 0, err
    assert _body(out) == ("panacus\tordered-growth\ncount\tnode\ncoverage\t1\nquorum\t0\n"
                          "chm13\t89\ngrch38\t106\nHG00438\t140\nHG00621\t154\n\n")
    order = tmp_path / "order.txt"
    order.write_text("HG00621\nchm13\nHG00438\ngrch38\n")
    rc, out, err = hl.run_cli(["ordered-histgrowth", "-S", "-c", "bp", "-l", "1,2", "-q", "0,0.5", "-O", str(order), gfa])
    assert rc == 0, err
    g = orc.Graph(gfa)
    pi, gi, names = g.path_order(orc.GROUP_SAMPLE, None, str(order))
    items, pre = g.item_table(orc.BP)
    r, c = orc.by_group(items, pre, pi, gi, g.n_nodes)
    rows = [x.split("\t") for x in _body(out).split("\n")[4:] if x]
    assert [x[0] for x in rows] == names == ["HG00621", "chm13", "HG00438", "grch38"]
    for k, (cv, q) in enumerate(((1, 0.0), (2, 0.5))):
        exp = orc.ordered_growth(r, c, 4, (orc.ABSOLUTE, cv), (orc.RELATIVE, q), g.node_lens)
        assert [x[1 + k] for x in rows] == [str(int(v)) for v in exp]


@pytest.mark.gpu
def test_cli_cfg2_shape_hist_bp_on_synthetic_gfa(tmp_path):
    """BASELINE configs[1] at test size: `hist -c bp` on a synthetic GFA file, bit-exact vs the
    oracle run on the same file; plus node/edge/all and -S grouping."""
    path = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "30000", "--paths", "12", "--links", "-o", path])
    assert rc == 0, err
    g = orc.Graph(path, index_edges=True)
    for grp_flag, gm in ((None, orc.GROUP_PATHID), ("-S", orc.GROUP_SAMPLE)):
        pi, gi, names = g.path_order(gm)
        exp = {}
        for cname, ct in (("node", orc.NODE), ("bp", orc.BP), ("edge", orc.EDGE)):
            items, pre = g.item_table(ct)
            cov = orc.coverage(items, pre, pi, gi, g.n_items(ct))
            exp[cname] = orc.hist(cov, len(names), g.node_lens if ct == orc.BP else None)
        args = ["hist", "-c", "all"] + ([grp_flag] if grp_flag else []) + [path]
        rc, out, err = hl.run_cli(args)
        assert rc == 0, err
        rows = [r.split("\t") for r in _body(out).split("\n")[4:] if r]
        assert [int(r[1]) for r in rows] == exp["node"].tolist()
        assert [int(r[2]) for r in rows] == exp["bp"].tolist()
        assert [int(r[3]) for r in rows] == exp["edge"].tolist()
    rc, out, err = hl.run_cli(["hist", "-c", "bp", path])
    assert rc == 0 and _body(out).split("\n")[1] == "count\tbp"


@pytest.mark.gpu
def test_cli_subset_exclude(tmp_path):
    """`-s` / `-e` whole-path lists through the CLI vs the oracle; subset == same graph without
    the other paths; excluded items never count."""
    path = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "20000", "--paths", "10", "--links", "-o", path])
    assert rc == 0, err
    g = orc.Graph(path, index_edges=True)
    names = g.path_names()
    sub = tmp_path / "sub.txt"
    sub.write_text("\n".join(n.split(":")[0] for n in (names[7], names[0], names[3], names[4])) + "\n")
    exc = tmp_path / "exc.txt"
    exc.write_text(names[5].split(":")[0] + "\n")
    for extra, sf, ef in ((["-s", str(sub)], str(sub), None), (["-e", str(exc)], None, str(exc)),
                          (["-S", "-s", str(sub), "-e", str(exc)], str(sub), str(exc))):
        gm = orc.GROUP_SAMPLE if "-S" in extra else orc.GROUP_PATHID
        pi, gi, gnames = g.path_order(gm, None, None, sf, ef)
        for cname, ct in (("node", orc.NODE), ("bp", orc.BP), ("edge", orc.EDGE)):
            items, pre = g.item_table(ct)
            excl = g.exclude_flags(ct, ef) if ef else None
            cov = orc.coverage(items, pre, pi, gi, g.n_items(ct), excl)
            h = orc.hist(cov, len(gnames), g.node_lens if ct == orc.BP else None)
            rc, out, err = hl.run_cli(["histgrowth", "-a", "-c", cname, "-l", "1,2", "-q", "0,0.5"] + extra + [path])
            assert rc == 0, err
            rows = [r.split("\t") for r in _body(out).split("\n")[4:] if r]
            assert [int(r[1]) for r in rows] == h.tolist(), (extra, cname)
            for k, (c, q) in enumerate(((1, 0.0), (2, 0.5))):
                exp = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
                assert [r[2 + k] for r in rows[1:]] == [hl.format_f64(math.floor(x)) for x in exp]
        # ordered growth with the same masks
        rc, out, err = hl.run_cli(["ordered-histgrowth", "-c", "bp", "-l", "1", "-q", "0.3"] + extra + [path])
        assert rc == 0, err
        items, pre = g.item_table(orc.BP)
        excl = g.exclude_flags(orc.BP, ef) if ef else None
        r_, c_ = orc.by_group(items, pre, pi, gi, g.n_nodes, excl)
        exp = orc.ordered_growth(r_, c_, len(gnames), (orc.ABSOLUTE, 1), (orc.RELATIVE, 0.3), g.node_lens)
        rows = [x.split("\t") for x in _body(out).split("\n")[4:] if x]
        assert [x[0] for x in rows] == gnames and [x[1] for x in rows] == [str(int(v)) for v in exp]


def _oracle_masked(g, ct, gm, sf, ef):
    """hist (with the uncovered-bp fix-up) and ordered growth inputs of the oracle under -s / -e lists"""
    pi, gi, gnames = g.path_order(gm, None, None, sf, ef)
    items, pre, fl, ids, bps = g.masked_table(ct, sf, ef)
    excl = fl if ef else None
    cov = orc.coverage(items, pre, pi, gi, g.n_items(ct), excl)
    h = orc.hist(cov, len(gnames), g.node_lens if ct == orc.BP else None)
    h = orc.hist_apply_uncovered(cov, ids, bps, h)
    w = None
    if ct == orc.BP:  # AbacusByGroup::calc_growth adds node_len - uncovered (abacus.rs:1013-1023)
        w = g.node_lens.copy()
        for i, u in zip(ids, bps):
            w[i] = 0 if u > w[i] else w[i] - u
    r_, c_ = orc.by_group(items, pre, pi, gi, g.n_items(ct), excl)
    return gnames, h, r_, c_, w


def _check_cli_masked(g, path, extra, gm, sf, ef):
    for cname, ct in (("node", orc.NODE), ("bp", orc.BP), ("edge", orc.EDGE)):
        gnames, h, r_, c_, w = _oracle_masked(g, ct, gm, sf, ef)
        rc, out, err = hl.run_cli(["histgrowth", "-a", "-c", cname, "-l", "1,2", "-q", "0,0.5"] + extra + [path])
        assert rc == 0, err
        rows = [r.split("\t") for r in _body(out).split("\n")[4:] if r]
        assert [int(r[1]) for r in rows] == h.tolist(), (extra, cname)
        if len(gnames) > 0:
            exp = orc.growth(h, (orc.ABSOLUTE, 2), (orc.RELATIVE, 0.5))
            assert [r[3] for r in rows[1:]] == [hl.format_f64(math.floor(x)) for x in exp]
            rc, out, err = hl.run_cli(["ordered-histgrowth", "-c", cname, "-l", "1,2", "-q", "0.3,0"] + extra + [path])
            assert rc == 0, err
            rows = [x.split("\t") for x in _body(out).split("\n")[4:] if x]
            assert [x[0] for x in rows] == gnames
            for k, (c, q) in enumerate(((1, 0.3), (2, 0.0))):
                exp = orc.ordered_growth(r_, c_, len(gnames), (orc.ABSOLUTE, c), (orc.RELATIVE, q), w)
                assert [x[1 + k] for x in rows] == [str(int(v)) for v in exp], (extra, cname, k)
    # all count types in one call take the same per-type masks
    rc, out, err = hl.run_cli(["hist", "-c", "all"] + extra + [path])
    assert rc == 0, err
    rows = [r.split("\t") for r in _body(out).split("\n")[4:] if r]
    head = _body(out).split("\n")[1].split("\t")[1:]
    for col, cname in enumerate(head):
        ct = {"node": orc.NODE, "bp": orc.BP, "edge": orc.EDGE}[cname]
        assert [int(r[1 + col]) for r in rows] == _oracle_masked(g, ct, gm, sf, ef)[1].tolist(), (extra, cname)


@pytest.mark.gpu
def test_cli_bed_intervals_chrM(golden_dir):
    """-s / -e BED lists with coordinates through the CLI on the GPU against the oracle
    pipeline, on the reference's own BED inputs (test/bed_chrM; no expected outputs exist upstream)."""
    gfa = os.path.join(golden_dir, "chrM_test.gfa")
    bed = os.path.join(golden_dir, "bed_chrM")
    g = orc.Graph(gfa, index_edges=True)
    for sf, ef in (("inclusion.bed3", None), (None, "exclusion.bed3"), ("inclusion.bed3", "exclusion.bed3"),
                   ("inclusion_sub.bed1", "exclusion.bed3"), ("inclusion.bed1", None)):
        sf, ef = sf and os.path.join(bed, sf), ef and os.path.join(bed, ef)
        extra = (["-s", sf] if sf else []) + (["-e", ef] if ef else [])
        _check_cli_masked(g, gfa, extra, orc.GROUP_PATHID, sf, ef)
        _check_cli_masked(g, gfa, extra + ["-S"], orc.GROUP_SAMPLE, sf, ef)


@pytest.mark.gpu
def test_cli_bed_intervals_synthetic(tmp_path):
    """random intervals over a generated graph: partially covered and partially excluded nodes"""
    path = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "3000", "--paths", "8", "--links", "-o", path])
    assert rc == 0, err
    g = orc.Graph(path, index_edges=True)
    names = [n.split(":")[0] for n in g.path_names()]
    items, pre = g.item_table(orc.NODE)
    bp = [int(g.node_lens[items[pre[k]:pre[k + 1]]].sum()) for k in range(g.n_paths)]
    rng = np.random.default_rng(11)
    n_unc = 0
    for rep in range(4):
        def rows(n_rows):
            out_rows = []
            for _ in range(n_rows):
                k = int(rng.integers(0, len(names)))
                lo = int(rng.integers(0, bp[k]))
                out_rows.append(f"{names[k]}\t{lo}\t{lo + int(rng.integers(1, bp[k] // 3 + 2))}")
            return "\n".join(out_rows) + "\n"
        sub, exc = tmp_path / f"s{rep}.bed", tmp_path / f"e{rep}.bed"
        sub.write_text(rows(6))
        exc.write_text(rows(3))
        for sf, ef in ((str(sub), None), (None, str(exc)), (str(sub), str(exc))):
            extra = (["-s", sf] if sf else []) + (["-e", ef] if ef else [])
            _check_cli_masked(g, path, extra, orc.GROUP_PATHID, sf, ef)   # walks tokenised on the device, cut where they are
            if rep == 0:   # ... and the host's step parser in front of the same cut
                os.environ["PANACUS_AMD_HOST_PARSE"] = "1"
                try:
                    _check_cli_masked(g, path, extra, orc.GROUP_PATHID, sf, ef)
                finally:
                    del os.environ["PANACUS_AMD_HOST_PARSE"]
            if sf:
                g.path_order(orc.GROUP_PATHID, None, None, sf, ef)
                n_unc += len(g.masked_table(orc.BP, sf, ef)[3])
    assert n_unc > 0  # the uncovered-bp fix-up was exercised


@pytest.mark.gpu
def test_cli_cfg2_full_size_hist_bp(tmp_path):
    """BASELINE configs[1] at its stated size: `hist -c bp` on a synthetic 1 M-node / 64-path GFA FILE
    (0.3 GB of text with sequences), the whole CLI path (parse -> ItemTable -> GPU -> TSV), bit-exact
    against the oracle on the same file; the .pcsr cache gives the same table."""
    path = str(tmp_path / "cfg2.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "1000000", "--paths", "64", "--sequences", "-o", path])
    assert rc == 0, err
    g = orc.Graph(path, index_edges=False)
    assert g.n_nodes == 1_000_000
    pi, gi, names = g.path_order(orc.GROUP_PATHID)
    assert len(names) == 64
    items, pre = g.item_table(orc.BP)
    cov = orc.coverage(items, pre, pi, gi, g.n_nodes)
    exp_bp = orc.hist(cov, 64, g.node_lens)
    exp_node = orc.hist(cov, 64)
    rc, out, err = hl.run_cli(["hist", "-c", "bp", path])
    assert rc == 0, err
    lines = _body(out).split("\n")
    assert lines[0] == "panacus\thist" and lines[1] == "count\tbp"
    rows = [r.split("\t") for r in lines[4:] if r]
    assert [int(r[0]) for r in rows] == list(range(65))
    assert [int(r[1]) for r in rows] == exp_bp.tolist()
    assert sum(int(r[1]) for r in rows) == int(g.node_lens.sum())
    rc, out2, err = hl.run_cli(["hist", "-c", "bp", "--cache", path])
    assert rc == 0, err
    rc, out3, err = hl.run_cli(["hist", "-c", "node", "--cache", path])
    assert rc == 0, err
    assert _body(out2) == _body(out)
    assert [int(r.split("\t")[1]) for r in _body(out3).split("\n")[4:] if r] == exp_node.tolist()


@pytest.mark.gpu
def test_cli_hist_on_a_gfa_whose_paths_have_large_rearrangements(tmp_path):
    """a GFA file whose P lines hold an inversion, a tandem duplication, a translocation, two inversions, an inversion inside a
    descending path (each a twelfth of the path) and one shuffled path: `hist -c node|bp` and `histgrowth` through the whole
    CLI path (the step columns tokenised on the device, the upload's chunk summaries cutting the paths, the one-shot pass over the
    pieces, the shuffled path's group left to a bitmap) against the oracle on the same file"""
    from test_gpu_band import _paths_with_large_rearrangements
    n, P = 300_000, 34          # (a small graph takes the one-shot route from 32 entries of the visiting order on)
    items, pre, _ = _paths_with_large_rearrangements(n, P - 1, 33)
    rng = np.random.default_rng(3)
    segs = [items[int(pre[k]):int(pre[k + 1])] for k in range(P - 1)]
    segs.append(rng.permutation(segs[2]))                   # one path with no order at all
    lens = rng.integers(1, 9, size=n)
    path = str(tmp_path / "sv.gfa")
    with open(path, "w") as f:
        f.write("H\tVN:Z:1.1\n")
        f.write("".join(f"S\t{i + 1}\t{'A' * int(lens[i])}\n" for i in range(n)))
        for k, sgm in enumerate(segs):
            f.write(f"P\ts{k}#1#c{k}\t" + ",".join(f"{int(x)}+" for x in sgm) + "\t*\n")
    g = orc.Graph(path, index_edges=False)
    pi, gi, names = g.path_order(orc.GROUP_PATHID)
    it, off = g.item_table(orc.NODE)
    cov = orc.coverage(it, off, pi, gi, g.n_nodes)
    for cname, w in (("node", None), ("bp", g.node_lens)):
        exp = orc.hist(cov, P, w)
        rc, out, err = hl.run_cli(["hist", "-c", cname, path])
        assert rc == 0, err
        rows = [r.split("\t") for r in _body(out).split("\n")[4:] if r]
        assert [int(r[1]) for r in rows] == exp.tolist()
    h = orc.hist(cov, P)
    rc, out, err = hl.run_cli(["histgrowth", "-a", "-c", "node", "-l", "1,2", "-q", "0,0.5", path])
    assert rc == 0, err
    rows = [r.split("\t") for r in _body(out).split("\n")[4:] if r]
    assert [int(r[1]) for r in rows] == h.tolist()
    for k, (c, q) in enumerate(((1, 0.0), (2, 0.5))):
        exp = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
        assert [r[2 + k] for r in rows[1:]] == [hl.format_f64(math.floor(x)) for x in exp]


@pytest.mark.gpu
def test_cli_similarity_chrM(golden_dir):
    """`similarity` = the Jaccard table of the groups, rows and columns in the dendrogram order of
    -m/--method (default centroid) like Similarity::set_table (similarity.rs:119-190): every method,
    every count type, against the oracle's table and permutation"""
    gfa = os.path.join(golden_dir, "chrM_test.gfa")
    g = orc.Graph(gfa, index_edges=True)
    seen_orders = set()
    for cname, ct in (("node", orc.NODE), ("bp", orc.BP), ("edge", orc.EDGE)):
        for grp, gmode in (("-S", orc.GROUP_SAMPLE), ("-H", orc.GROUP_HAPLOTYPE)):
            pi, gi, names = g.path_order(gmode)
            items, pre = g.item_table(ct)
            r, c = orc.by_group(items, pre, pi, gi, g.n_items(ct))
            _, _, tab = orc.similarity(r, c, len(names), g.node_lens if ct == orc.BP else None)
            for method in [None] + orc.CLUSTER_METHODS:
                rc, out, err = hl.run_cli(["similarity", grp, "-c", cname] + (["-m", method.upper()] if method else []) + [gfa])
                assert rc == 0, err
                tab2, perm = orc.similarity_order(tab, method or "centroid")
                labels = [names[int(k)] for k in perm]
                rows = [x.split("\t") for x in _body(out).split("\n") if x]
                assert rows[0] == ["group"] + labels
                for i, name in enumerate(labels):
                    assert rows[1 + i] == [name] + [hl.format_f32(v) for v in tab2[i]]
                assert out.endswith("\n\n")
                seen_orders.add(tuple(labels))
    assert len(seen_orders) > 1   # the clustering really reorders something on this graph
    rc, out, err = hl.run_cli(["similarity", "-S", "-m", "upgma", gfa])
    assert rc != 0 and "--method" in err


@pytest.mark.gpu
def test_cli_similarity_synthetic_groups(tmp_path):
    """24 groups of a synthetic pangenome: the CLI's order equals the oracle's for every method"""
    path = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "6000", "--paths", "48", "-o", path])
    assert rc == 0, err
    g = orc.Graph(path, index_edges=False)
    pi, gi, names = g.path_order(orc.GROUP_SAMPLE)
    for cname, ct in (("node", orc.NODE), ("bp", orc.BP)):
        items, pre = g.item_table(ct)
        r, c = orc.by_group(items, pre, pi, gi, g.n_items(ct))
        _, _, tab = orc.similarity(r, c, len(names), g.node_lens if ct == orc.BP else None)
        for method in orc.CLUSTER_METHODS:
            rc, out, err = hl.run_cli(["similarity", "-S", "-c", cname, "--method", method, path])
            assert rc == 0, err
            tab2, perm = orc.similarity_order(tab, method)
            rows = [x.split("\t") for x in _body(out).split("\n") if x]
            assert rows[0] == ["group"] + [names[int(k)] for k in perm]
            assert [row[1:] for row in rows[1:]] == [[hl.format_f32(v) for v in tab2[i]] for i in range(len(names))]


@pytest.mark.gpu
def test_cli_table_total_chrM(golden, golden_dir):
    gfa = os.path.join(golden_dir, "chrM_test.gfa")
    rc, out, err = hl.run_cli(["table", "--total", "-S", gfa])
    assert rc == 0, err
    rows = [x.split("\t") for x in _body(out).split("\n") if x]
    assert rows[0] == ["node", "total"]
    # abacus.rs:1487-1496: coverage of node id i; segment names of chrM_test.gfa in S-line order
    names = [l.split("\t")[1] for l in open(gfa) if l.startswith("S\t")]
    assert [r[0] for r in rows[1:]] == names
    assert [int(r[1]) for r in rows[1:]] == golden["chrM_sample_node"]["countable"][1:]
    rc, out, err = hl.run_cli(["table", "--total", "-S", "-c", "edge", gfa])
    assert rc == 0, err
    rows = [x.split("\t") for x in _body(out).split("\n") if x]
    assert rows[0] == ["edge", "total"]
    assert [int(r[1]) for r in rows[1:]] == golden["chrM_sample_edge"]["countable"][1:]
    assert all(r[0][0] in "<>" for r in rows[1:])


@pytest.mark.gpu
def test_cli_table_per_group(golden_dir, tmp_path):
    """`table` without --total (abacus.rs:1093-1112, 1150-1168): per group the number of steps on the
    item (AbacusByGroup.v) times its bp; the edge branch prints v[group id] (restated as it is)."""
    cases = [(os.path.join(golden_dir, "chrM_test.gfa"), ["-S"], orc.GROUP_SAMPLE, None, None),
             (os.path.join(golden_dir, "t_groups.gfa"), [], orc.GROUP_PATHID, None, None),
             (os.path.join(golden_dir, "chrM_test.gfa"), ["-s", os.path.join(golden_dir, "bed_chrM", "inclusion.bed3"),
                                                          "-e", os.path.join(golden_dir, "bed_chrM", "exclusion.bed3")],
              orc.GROUP_PATHID, os.path.join(golden_dir, "bed_chrM", "inclusion.bed3"),
              os.path.join(golden_dir, "bed_chrM", "exclusion.bed3"))]
    syn = str(tmp_path / "syn.gfa")  # duplicated steps inside paths: multiplicities > 1
    rc, out, err = hl.run_cli(["synth", "--nodes", "5000", "--paths", "9", "--links", "-o", syn])
    assert rc == 0, err
    cases.append((syn, ["-S"], orc.GROUP_SAMPLE, None, None))
    saw_multi = False
    for gfa, extra, gm, sf, ef in cases:
        g = orc.Graph(gfa, index_edges=True)
        for cname, ct in (("node", orc.NODE), ("bp", orc.BP), ("edge", orc.EDGE)):
            pi, gi, names = g.path_order(gm, None, None, sf, ef)
            items, pre, fl, ids, ubp = g.masked_table(ct, sf, ef)
            r, c, v = orc.by_group_values(items, pre, pi, gi, g.n_items(ct), fl if ef else None)
            saw_multi = saw_multi or (len(v) and int(v.max()) > 1)
            bps = None
            if ct == orc.BP:
                bps = g.node_lens.astype(np.uint64)
                bps[ids] -= ubp
            rc, out, err = hl.run_cli(["table", "-c", cname] + extra + [gfa])
            try:
                exp = orc.table_rows_values(r, c, v, len(names), bps, ct == orc.EDGE)
            except IndexError:
                assert rc == 1 and "panic" in err
                continue
            assert rc == 0, err
            rows = [x.split("\t") for x in _body(out).split("\n") if x]
            assert rows[0] == ["edge" if ct == orc.EDGE else "node"] + names
            assert len(rows) - 1 == g.n_items(ct)
            got = np.array([[int(x) for x in row[1:]] for row in rows[1:]], dtype=np.uint64).reshape(len(rows) - 1, len(names))
            assert np.array_equal(got, exp), (gfa, cname)
    assert saw_multi
    # the item range is walked in slices: a tiny slice gives the same table
    full = hl.run_cli(["table", "-c", "bp", "-S", syn])
    os.environ["PANACUS_AMD_TABLE_SLICE"] = "37"
    try:
        sliced = hl.run_cli(["table", "-c", "bp", "-S", syn])
        sliced_e = hl.run_cli(["table", "-c", "edge", "-S", syn])
    finally:
        del os.environ["PANACUS_AMD_TABLE_SLICE"]
    assert full[0] == 0 and _body(full[1]) == _body(sliced[1])
    assert _body(sliced_e[1]) == _body(hl.run_cli(["table", "-c", "edge", "-S", syn])[1])


@pytest.mark.gpu
def test_cli_empty_selection(golden_dir):
    """a subset that names nothing: no groups, every item uncovered (one histogram row), and the
    growth commands print just their headers"""
    gfa = os.path.join(golden_dir, "chrM_test.gfa")
    rc, out, err = hl.run_cli(["hist", "-c", "all", "-s", "matches_no_path_at_all", gfa])
    assert rc == 0, err
    rows = [r.split("\t") for r in _body(out).split("\n") if r]
    assert len(rows) == 5 and rows[4][0] == "0"
    g = orc.Graph(gfa, index_edges=True)
    col = {name: j for j, name in enumerate(rows[1])}
    assert int(rows[4][col["node"]]) == g.n_nodes and int(rows[4][col["edge"]]) == g.n_edges
    assert int(rows[4][col["bp"]]) == int(g.node_lens.sum())
    rc, out, err = hl.run_cli(["ordered-histgrowth", "-s", "matches_no_path_at_all", gfa])
    assert rc == 0, err
    assert len([r for r in _body(out).split("\n") if r]) == 4


@pytest.mark.gpu
def test_cli_cache_gives_identical_tables(golden_dir, tmp_path):
    import shutil
    gfa = str(tmp_path / "chrM_test.gfa")
    shutil.copy(os.path.join(golden_dir, "chrM_test.gfa"), gfa)
    for args in (["hist", "-S", "-c", "all"], ["histgrowth", "-H", "-c", "edge", "-l", "1,2", "-q", "0,0.5"],
                 ["ordered-histgrowth", "-S", "-c", "bp"], ["table", "--total", "-S", "-c", "edge"],
                 ["similarity", "-S", "-c", "bp"]):
        rc, plain, err = hl.run_cli(args + [gfa])
        assert rc == 0, err
        if os.path.exists(gfa + ".pcsr"):
            os.remove(gfa + ".pcsr")
        rc, first, err = hl.run_cli(args + ["--cache", gfa])   # parses and writes the cache
        assert rc == 0, err
        assert os.path.exists(gfa + ".pcsr")
        rc, second, err = hl.run_cli(args + ["--cache", gfa])  # served from the cache
        assert rc == 0, err
        assert _body(plain) == _body(first) == _body(second)


@pytest.mark.gpu
def test_multi_gpu_tool_matches_cli(golden_dir, tmp_path):
    """tools/histgrowth_multi_gpu.py with one process == `panacus-amd histgrowth -a` (the N > 1 path adds
    node-range shards and one all-reduce, covered by tests/test_distributed_gloo.py)"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("hmg", os.path.join(root, "tools", "histgrowth_multi_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gfa = os.path.join(golden_dir, "chrM_test.gfa")
    for cname in ("node", "bp", "edge"):
        out_file = str(tmp_path / f"t_{cname}.tsv")
        text = mod.main(["-c", cname, "-l", "1,2", "-q", "0,0.5", "-S", "-o", out_file, gfa])
        rc, out, err = hl.run_cli(["histgrowth", "-a", "-c", cname, "-l", "1,2", "-q", "0,0.5", "-S", gfa])
        assert rc == 0, err
        assert text == _body(out).rstrip("\n") + "\n"
    # the same through the library's own RCCL communicator (one rank; no torch involved)
    os.environ.update(PANACUS_DIST_BACKEND="native", PANACUS_NATIVE_SINGLE="1", PANACUS_COMM_ID_FILE=str(tmp_path / "c.id"))
    try:
        text = mod.main(["-c", "bp", "-l", "1,2", "-q", "0,0.5", "-S", "-o", str(tmp_path / "n.tsv"), gfa])
    finally:
        for k in ("PANACUS_DIST_BACKEND", "PANACUS_NATIVE_SINGLE", "PANACUS_COMM_ID_FILE"):
            del os.environ[k]
    rc, out, err = hl.run_cli(["histgrowth", "-a", "-c", "bp", "-l", "1,2", "-q", "0,0.5", "-S", gfa])
    assert text == _body(out).rstrip("\n") + "\n"


@pytest.mark.gpu
def test_multi_gpu_tool_two_ranks_non_monotone_paths(tmp_path):
    """Two ranks (node-range shards, both on GPU 0, counters reduced over gloo) on a GFA whose paths are
    NOT tile-monotone: the first coverage pass of every rank only classifies the paths and is re-run by
    the library, so a host that reduced the counters of the first attempt would print a wrong table
    (round-1 advisor finding).  The tool must settle the pass first: its table equals the single-GPU CLI."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "30000", "--paths", "10", "-o", src])
    assert rc == 0, err
    gfa = str(tmp_path / "bumpy.gfa")
    with open(src) as f, open(gfa, "w") as g:
        for line in f:
            if line.startswith("P\t"):
                cols = line.rstrip("\n").split("\t")
                steps = cols[2].split(",")
                for a in range(100, len(steps) - 60, 300):   # a 40-step window reversed every 300 steps
                    steps[a:a + 40] = steps[a:a + 40][::-1]
                cols[2] = ",".join(steps)
                line = "\t".join(cols) + "\n"
            g.write(line)
    for cname in ("node", "bp"):
        rc, ref, err = hl.run_cli(["histgrowth", "-a", "-c", cname, "-l", "1,2", "-q", "0,0.5", gfa])
        assert rc == 0, err
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        out_file = str(tmp_path / f"two_{cname}.tsv")
        procs = []
        for r in range(2):
            e = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                     MASTER_PORT=str(port), PANACUS_DIST_BACKEND="gloo", PANACUS_TOOL_REPORT_RERUNS="1",
                     PNX_COVER_VARIANT="2")   # over the steps: the route on which a first pass is run again
            procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tools", "histgrowth_multi_gpu.py"), "-c", cname,
                                           "-l", "1,2", "-q", "0,0.5", "-o", out_file, gfa], env=e,
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE))
        errs = []
        for pr in procs:
            o, e2 = pr.communicate(timeout=600)
            assert pr.returncode == 0, e2.decode()[-2000:]
            errs.append(e2.decode())
        assert "reruns=0" not in errs[0] and "reruns=" in errs[0]   # the scenario really happened on rank 0
        assert open(out_file).read() == _body(ref).rstrip("\n") + "\n"


@pytest.mark.gpu
def test_cli_edge_counts_do_not_depend_on_link_order(tmp_path):
    """edge ids follow the L lines (graph.rs:282-295); the CLI renumbers them for the device when
    they do not follow the paths.  Sorted and shuffled link sections give the oracle's numbers."""
    path = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "20000", "--paths", "12", "--links", "-o", path])
    assert rc == 0, err
    lines = open(path).read().split("\n")
    links = [l for l in lines if l.startswith("L\t")]
    rng = np.random.default_rng(7)
    shuf = str(tmp_path / "shuf.gfa")
    with open(shuf, "w") as f:
        f.write("\n".join([l for l in lines if l and not l.startswith("L\t")] + [links[i] for i in rng.permutation(len(links))]) + "\n")
    outs = []
    for gfa in (path, shuf):
        g = orc.Graph(gfa, index_edges=True)
        pi, gi, names = g.path_order(orc.GROUP_SAMPLE)
        items, pre = g.item_table(orc.EDGE)
        cov = orc.coverage(items, pre, pi, gi, g.n_edges)
        h = orc.hist(cov, len(names))
        rc, out, err = hl.run_cli(["histgrowth", "-a", "-c", "edge", "-S", "-l", "1,2", "-q", "0,0.5", gfa])
        assert rc == 0, err
        rows = [r.split("\t") for r in _body(out).split("\n")[4:] if r]
        assert [int(r[1]) for r in rows] == h.tolist()
        rc, out2, err = hl.run_cli(["ordered-histgrowth", "-c", "edge", "-S", "-l", "1", "-q", "0.3", gfa])
        assert rc == 0, err
        r_, c_ = orc.by_group(items, pre, pi, gi, g.n_edges)
        exp = orc.ordered_growth(r_, c_, len(names), (orc.ABSOLUTE, 1), (orc.RELATIVE, 0.3))
        orows = [x.split("\t") for x in _body(out2).split("\n")[4:] if x]
        assert [x[1] for x in orows] == [str(int(v)) for v in exp]
        outs.append((_body(out), _body(out2)))
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_cli_pggb_shaped_graph_example_commands(tmp_path):
    """BASELINE configs[4] (HPRC chr22 pggb) is a 402 MB download that is not in the container; this is its
    structural stand-in (`synth --shape pggb`: contig paths per haplotype, 2-part reference names, inversions with
    '-' steps, tandem duplications -> paths on the tile, run and sorted routes at once) through the commands of
    examples/pangenome_growth_pggb.md:13-22 and test/integrated_test.R:95-137, against the oracle pipeline."""
    gfa = str(tmp_path / "pggb.gfa")
    rc, out, err = hl.run_cli(["synth", "--shape", "pggb", "--nodes", "150000", "--samples", "10", "-o", gfa])
    assert rc == 0, err
    g = orc.Graph(gfa, index_edges=True)
    names = g.path_names()
    assert names[0] == "chm13#chr22" and names[1] == "grch38#chr22" and len(names) > 100
    items, pre = g.item_table(orc.NODE)
    assert any(np.any(np.diff(items[pre[k]:pre[k + 1]].astype(np.int64)) < 0) for k in range(2, len(names)))  # back-steps exist
    haps = tmp_path / "haplotypes.txt"
    haps.write_text("".join(n + "\n" for n in names if not n.startswith(("grch38", "chm13"))))
    body = lambda text: [l.split("\t") for l in text.split("\n") if l and not l.startswith("#")]  # noqa: E731
    # 1. the example: histgrowth -l 1,2,1,1,1 -q 0,0,1,0.5,0.1 -S -a -s haplotypes (node), and the same for bp / edge
    pairs = [(1, 0.0, "0"), (2, 0.0, "0"), (1, 1.0, "1"), (1, 0.5, "0.5"), (1, 0.1, "0.1")]
    pi, gi, gnames = g.path_order(orc.GROUP_SAMPLE, None, None, str(haps), None)
    G = len(gnames)
    assert G == 10
    for cname, ct in (("node", orc.NODE), ("bp", orc.BP), ("edge", orc.EDGE)):
        rc, out, err = hl.run_cli(["histgrowth", "-c", cname, "-l", "1,2,1,1,1", "-q", "0,0,1,0.5,0.1", "-S", "-a", "-s", str(haps), gfa])
        assert rc == 0, err
        rows = body(out)
        m_items, m_pre, fl, ids, ubp = g.masked_table(ct, str(haps), None)
        cov = orc.coverage(m_items, m_pre, pi, gi, g.n_items(ct), None)
        h = orc.hist_apply_uncovered(cov, ids, ubp, orc.hist(cov, G, g.node_lens if ct == orc.BP else None))
        assert [int(r[1]) for r in rows[4:]] == h.tolist(), cname
        for k, (c, q, qs) in enumerate(pairs):
            exp = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
            assert (rows[2][2 + k], rows[3][2 + k]) == (str(c), qs)
            assert [r[2 + k] for r in rows[5:]] == [hl.format_f64(math.floor(x)) for x in exp], (cname, c, q)
    # 2. integrated_test.R's grid on the whole graph: -S / -H, node / edge, -q 0,0.5,1.0 -l 0,1,2
    for flag, mode in (("-S", orc.GROUP_SAMPLE), ("-H", orc.GROUP_HAPLOTYPE)):
        pi, gi, gnames = g.path_order(mode)
        G = len(gnames)
        for cname, ct in (("node", orc.NODE), ("edge", orc.EDGE)):
            rc, out, err = hl.run_cli(["histgrowth", "-c", cname, flag, "-q", "0,0.5,1.0", "-l", "0,1,2", "-a", gfa])
            assert rc == 0, err
            rows = body(out)
            t_items, t_pre = g.item_table(ct)
            cov = orc.coverage(t_items, t_pre, pi, gi, g.n_items(ct))
            h = orc.hist(cov, G)
            assert [int(r[1]) for r in rows[4:]] == h.tolist(), (flag, cname)
            for k, (c, q) in enumerate(((0, 0.0), (1, 0.5), (2, 1.0))):
                exp = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
                assert [r[2 + k] for r in rows[5:]] == [hl.format_f64(math.floor(x)) for x in exp], (flag, cname, c, q)
    # 3. ordered growth in file order of the samples + the similarity table, bp
    pi, gi, gnames = g.path_order(orc.GROUP_SAMPLE)
    G = len(gnames)
    rc, out, err = hl.run_cli(["ordered-histgrowth", "-c", "bp", "-S", "-l", "1,2", "-q", "0,0.5", gfa])
    assert rc == 0, err
    rows = body(out)[4:]
    assert [r[0] for r in rows] == gnames
    t_items, t_pre = g.item_table(orc.BP)
    r_, c_ = orc.by_group(t_items, t_pre, pi, gi, g.n_nodes)
    for k, (c, q) in enumerate(((1, 0.0), (2, 0.5))):
        exp = orc.ordered_growth(r_, c_, G, (orc.ABSOLUTE, c), (orc.RELATIVE, q), g.node_lens)
        assert [x[1 + k] for x in rows] == [hl.format_f64(float(v)) for v in exp], (c, q)


@pytest.mark.gpu
def test_cli_commands_on_several_threads_of_one_process(tmp_path):
    """A host that binds the library in-process may run commands on several threads at once (the reference calls the replaced
    functions from rayon workers, src/analyses/ordered_histgrowth.rs:174-188): every command has a GPU context of its own, and
    the closed forms of >= 256 groups are evaluated on THAT context (bound per thread, round 6) -- not on whichever context
    another thread's command registered last.  Four threads, each its own graph and commands, tables equal to the ones the
    same commands print one after the other."""
    import threading
    graphs = []
    for k, (n, p) in enumerate(((30000, 300), (20000, 270), (25000, 40), (12000, 330))):
        path = str(tmp_path / f"g{k}.gfa")
        rc, out, err = hl.run_cli_inprocess(["synth", "--nodes", str(n), "--paths", str(p), "--seed", str(5 + k), "--links", "-o", path])
        assert rc == 0, err
        graphs.append(path)
    cmds = [["histgrowth", "-a", "-c", "node", "-l", "1,2,1", "-q", "0,0,0.5"], ["histgrowth", "-c", "bp", "-l", "1,1", "-q", "0.3,0.9"],
            ["ordered-histgrowth", "-c", "node", "-l", "1,2", "-q", "0,0.5"], ["hist", "-c", "all"]]
    serial = {}
    for g in graphs:
        for c in cmds:
            rc, out, err = hl.run_cli_inprocess(c + [g])
            assert rc == 0, err
            serial[(g, tuple(c))] = _body(out)
    errors = []

    def worker(g):
        try:
            for rep in range(3):
                for c in cmds:
                    rc, out, err = hl.run_cli_inprocess(c + [g])
                    if rc != 0:
                        errors.append((g, c, err))
                    elif _body(out) != serial[(g, tuple(c))]:
                        errors.append((g, c, "table differs from the command run alone"))
        except Exception as e:   # noqa: BLE001
            errors.append((g, None, repr(e)))

    ts = [threading.Thread(target=worker, args=(g,)) for g in graphs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]
