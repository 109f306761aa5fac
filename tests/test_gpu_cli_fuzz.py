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
    _random_gfa(rng, gfa, crlf=False)
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
