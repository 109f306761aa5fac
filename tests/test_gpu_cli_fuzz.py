"""End-to-end CLI against the oracle pipeline on random GFA files (tests/test_host_gfa_fuzz.py's
generator): hist for every count type and grouping, histgrowth, ordered-histgrowth."""
import math
import os

import numpy as np
import pytest

import oracle as orc
from panacus_amd import hostlib as hl
from test_host_gfa_fuzz import _random_gfa

pytestmark = pytest.mark.gpu
N_SEEDS = int(os.environ.get("PANACUS_FUZZ_SEEDS", "16"))


def _body(text):
    return [l.split("\t") for l in text.split("\n") if l and not l.startswith("#")]


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_cli_on_random_gfa(tmp_path, seed):
    rng = np.random.default_rng(9000 + seed)
    gfa = str(tmp_path / "r.gfa")
    _random_gfa(rng, gfa, crlf=seed % 4 == 3)   # (every fourth file with \r\n line ends: the device reads S / L / P / W lines of those too)
    try:
        g = orc.Graph(gfa, index_edges=True)
    except Exception:
        pytest.skip("generator produced a graph the reference rejects")
    for flag, mode in (("", orc.GROUP_PATHID), ("-S", orc.GROUP_SAMPLE), ("-H", orc.GROUP_HAPLOTYPE)):
        pi, gi, names = g.path_order(mode)
        G = len(names)
        args = ["histgrowth", "-c", "all", "-a", "-l", "1,2", "-q", "0,0.5"] + ([flag] if flag else []) + [gfa]
        rc, out, err = hl.run_cli(args)
        assert rc == 0, err
        rows = _body(out)
        # header rows: panacus / count / coverage / quorum (Appendix B of SURVEY.md); columns are
        # looked up by their header, whatever order the writer uses
        colmap = {(rows[0][j], rows[1][j], rows[2][j], rows[3][j]): j for j in range(1, len(rows[0]))}
        for cname, ct in (("node", orc.NODE), ("bp", orc.BP), ("edge", orc.EDGE)):
            items, pre = g.item_table(ct)
            cov = orc.coverage(items, pre, pi, gi, g.n_items(ct))
            h = orc.hist(cov, G, g.node_lens if ct == orc.BP else None)
            j = colmap[("hist", cname, "", "")]
            assert [int(r[j]) for r in rows[4:]] == h.tolist(), (flag, cname)
            for c, q, qs in ((1, 0.0, "0"), (2, 0.5, "0.5")):
                exp = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
                j = colmap[("growth", cname, str(c), qs)]
                assert rows[4][j] == "NaN"
                got = [r[j] for r in rows[5:]]
                assert got == [hl.format_f64(math.floor(x)) for x in exp], (flag, cname, c, q)
        # ordered growth in file order of the groups
        rc, out, err = hl.run_cli(["ordered-histgrowth", "-c", "bp", "-l", "1,2", "-q", "0,0.3"] + ([flag] if flag else []) + [gfa])
        assert rc == 0, err
        rows = _body(out)[4:]
        assert [r[0] for r in rows] == names
        items, pre = g.item_table(orc.BP)
        r_, c_ = orc.by_group(items, pre, pi, gi, g.n_nodes)
        for k, (c, q) in enumerate(((1, 0.0), (2, 0.3))):
            exp = orc.ordered_growth(r_, c_, G, (orc.ABSOLUTE, c), (orc.RELATIVE, q), g.node_lens)
            assert [x[1 + k] for x in rows] == [hl.format_f64(float(v)) for v in exp], (flag, c, q)


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_cli_bed_lists_and_table_on_random_gfa(tmp_path, seed):
    """-s / -e BED lists (paths, groups, intervals) through hist / ordered-histgrowth / table"""
    from test_host_gfa_fuzz import _random_bed
    rng = np.random.default_rng(13000 + seed)
    gfa = str(tmp_path / "r.gfa")
    _random_gfa(rng, gfa, crlf=False)
    try:
        g = orc.Graph(gfa, index_edges=True)
    except Exception:
        pytest.skip("generator produced a graph the reference rejects")
    items0, pre0 = g.item_table(orc.NODE)
    path_bp = [int(g.node_lens[items0[pre0[k]:pre0[k + 1]]].sum()) for k in range(g.n_paths)]
    flag, mode = [("", orc.GROUP_PATHID), ("-S", orc.GROUP_SAMPLE), ("-H", orc.GROUP_HAPLOTYPE)][seed % 3]
    _, _, gnames = g.path_order(mode)
    sf = ef = None
    if rng.random() < 0.8:
        sf = str(tmp_path / "s.bed")
        _random_bed(rng, sf, g.path_names(), path_bp, gnames if mode != orc.GROUP_PATHID else [])
    if rng.random() < 0.8 or sf is None:
        ef = str(tmp_path / "e.bed")
        _random_bed(rng, ef, g.path_names(), path_bp, gnames if mode != orc.GROUP_PATHID else [])
    extra = ([flag] if flag else []) + (["-s", sf] if sf else []) + (["-e", ef] if ef else [])
    try:
        pi, gi, names = g.path_order(mode, None, None, sf, ef)
        g.masked_table(orc.NODE, sf, ef)
    except ValueError:  # a row the reference panics on (two columns, a coordinate that is no number): the CLI refuses it too
        rc, out, err = hl.run_cli(["hist", "-c", "all"] + extra + [gfa])
        assert rc != 0 and err
        return
    G = len(names)
    rc, out, err = hl.run_cli(["hist", "-c", "all"] + extra + [gfa])
    assert rc == 0, err
    rows = _body(out)
    colmap = {rows[1][j]: j for j in range(1, len(rows[0]))}
    for cname, ct in (("node", orc.NODE), ("bp", orc.BP), ("edge", orc.EDGE)):
        items, pre, fl, ids, ubp = g.masked_table(ct, sf, ef)
        excl = fl if ef else None
        cov = orc.coverage(items, pre, pi, gi, g.n_items(ct), excl)
        h = orc.hist_apply_uncovered(cov, ids, ubp, orc.hist(cov, G, g.node_lens if ct == orc.BP else None))
        assert [int(r[colmap[cname]]) for r in rows[4:]] == h.tolist(), (extra, cname)
        r_, c_, v_ = orc.by_group_values(items, pre, pi, gi, g.n_items(ct), excl)
        bps = None
        if ct == orc.BP:
            bps = g.node_lens.astype(np.uint64)
            bps[ids] -= ubp
        if G:
            w = None if bps is None else bps.astype(np.uint32)
            rc, out2, err = hl.run_cli(["ordered-histgrowth", "-c", cname, "-l", "1,2", "-q", "0,0.4"] + extra + [gfa])
            assert rc == 0, err
            orows = _body(out2)[4:]
            assert [r[0] for r in orows] == names
            for k, (c, q) in enumerate(((1, 0.0), (2, 0.4))):
                exp = orc.ordered_growth(r_, c_, G, (orc.ABSOLUTE, c), (orc.RELATIVE, q), w)
                assert [x[1 + k] for x in orows] == [hl.format_f64(float(x)) for x in exp], (extra, cname, k)
        rc, out3, err = hl.run_cli(["table", "-c", cname] + extra + [gfa])
        try:
            exp = orc.table_rows_values(r_, c_, v_, G, bps, ct == orc.EDGE)
        except IndexError:
            assert rc == 1 and "panic" in err
            continue
        assert rc == 0, err
        trows = _body(out3)
        assert trows[0][1:] == names and len(trows) - 1 == g.n_items(ct)
        got = np.array([[int(x) for x in row[1:]] for row in trows[1:]], dtype=np.uint64).reshape(len(trows) - 1, G)
        assert np.array_equal(got, exp), (extra, cname)
