"""The YAML `report` runner and the JSON report surface (src/commands/report.rs,
src/analysis_parameter.rs:83-258, src/html_report.rs:56-66,396-457, analyses/*::generate_report_section).
CPU part: the YAML subset, the ordering of runs and analyses, float formatting as serde_json prints it.
GPU part (marked): the sections of chrM against the tables of the classic subcommands."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest

from panacus_amd import hostlib as hl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHRM = os.path.join(ROOT, "tests", "golden", "chrM_test.gfa")


def _write(tmp_path, text, name="r.yaml"):
    p = str(tmp_path / name)
    with open(p, "w") as f:
        f.write(text)
    return p


def test_dry_run_sorts_runs_and_analyses(tmp_path):
    """AnalysisRun::convert_to_tasks sorts the runs, to_tasks sorts the analyses of a run (derive(Ord):
    variant order Hist < Growth < Table < ... < OrderedGrowth < ... < Similarity), and an OrderedGrowth
    is preceded by its OrderChange task (analysis_parameter.rs:117-151, 239-244)"""
    y = _write(tmp_path, """
# a comment
- graph: z.gfa        # trailing comment
  grouping: Haplotype
  analyses:
    - !Similarity
      count_type: Bp
      cluster_method: Ward
    - !Growth
      coverage: 1,1,2
      quorum: 0,0.9,0
    - !Hist
      count_type: Bp
    - !OrderedGrowth
      coverage: "1"
      order: 'order.txt'
      count_type: Bp
- graph: a.gfa
  name: first run
  subset: sub.bed
  grouping: !Custom groups.tsv
  analyses:
  - !Table
    total: true
  - !Hist
""")
    rc, out, err = hl.run_cli(["report", "--dry-run", y])
    assert rc == 0, err
    lines = [l for l in out.split("\n") if l]
    assert lines == [
        'GraphStateChange { graph: "a.gfa", name: Some("first run"), subset: "sub.bed", exclude: "", grouping: Custom("groups.tsv"), nice: false }',
        "  Analysis Hist { count_type: Node }",
        "  Analysis Table { count_type: Node }",
        'GraphStateChange { graph: "z.gfa", name: None, subset: "", exclude: "", grouping: Haplotype, nice: false }',
        "  Analysis Hist { count_type: Bp }",
        "  Analysis Growth { coverage: 1,1,2, quorum: 0,0.9,0 }",
        '  OrderChange(Some("order.txt"))',
        "  Analysis OrderedGrowth { count_type: Bp }",
        "  Analysis Similarity { count_type: Bp }",
    ]


@pytest.mark.parametrize("text,msg", [
    ("- graph: a.gfa\n  analyses:\n    - !Histo\n", "unknown analysis"),
    ("- graph: a.gfa\n  analyses:\n    - !Hist\n      count: Bp\n", "unknown field `count`"),
    ("- graph: a.gfa\n  analyses:\n    - !Hist\n      count_type: bp\n", "unknown variant `bp`"),
    ("- graf: a.gfa\n  analyses: []\n", "flow collections"),
    ("- graf: a.gfa\n  analyses:\n", "unknown field `graf`"),
    ("- analyses:\n    - !Hist\n", "missing field `graph`"),
    ("- graph: a.gfa\n", "missing field `analyses`"),
    ("- graph: a.gfa\n  analyses:\n    - !Info\n", "outside the hist / growth hot path"),
    ("- graph: a.gfa\n  analyses:\n    - !Table\n      count_type: Edge\n", "missing field `total`"),
    ("- graph: a.gfa\n  grouping: Sampel\n  analyses:\n    - !Hist\n", "unknown variant `Sampel`"),
    ("- graph: a.gfa\n  graph: b.gfa\n  analyses:\n    - !Hist\n", "duplicate key"),
    ("- graph: a.gfa\n\tanalyses:\n", "tabs"),
])
def test_config_errors_name_the_problem(tmp_path, text, msg):
    rc, out, err = hl.run_cli(["report", "--dry-run", _write(tmp_path, text)])
    assert rc != 0 and msg in err, err


def test_report_without_a_file_prints_the_example():
    rc, out, err = hl.run_cli(["report"])
    assert rc == 0 and "# Missing YAML file!" in out and "- !Growth" in out


def test_html_is_out_of_scope(tmp_path):
    rc, out, err = hl.run_cli(["report", _write(tmp_path, "- graph: a.gfa\n  analyses:\n    - !Hist\n")])
    assert rc != 0 and "--json" in err


def test_floats_as_serde_json_prints_them():
    """ryu: shortest round-trip digits, `.0` on integers, decimal notation while the point stays within
    16 (f32: 13) digits / 5 (f32: 6) leading zeros, scientific beyond, no `+` and no padding in exponents"""
    f64 = {0.0: "0.0", 1.0: "1.0", 616.0: "616.0", 5.666666666666667: "5.666666666666667", 1e15: "1000000000000000.0",
           1e16: "1e16", 123456789012345680.0: "1.2345678901234568e17", 1e-5: "0.00001", 1.5e-6: "1.5e-6",
           0.001234: "0.001234", 9007199254740992.0: "9007199254740992.0", -2.5: "-2.5", 8336454.25: "8336454.25",
           1e21: "1e21", 1.7976931348623157e308: "1.7976931348623157e308", 5e-324: "5e-324"}
    for x, s in f64.items():
        assert hl.json_f64(x) == s, x
    assert hl.json_f64(float("nan")) == "null" and hl.json_f64(float("inf")) == "null"
    f32 = {1.0: "1.0", 0.25: "0.25", float(np.float32(1) / np.float32(3)): "0.33333334", 1e-6: "0.000001", 1e-7: "1e-7",
           12345678.0: "12345678.0", 1e13: "1e13", 0.1: "0.1"}
    for x, s in f32.items():
        assert hl.json_f32(np.float32(x)) == s, x
    rng = np.random.default_rng(0)
    for x in np.concatenate([rng.random(200) * 10.0 ** rng.integers(-4, 15, 200), rng.integers(0, 1 << 50, 100).astype(np.float64)]):
        s = hl.json_f64(float(x))
        assert float(s) == float(x) and (("e" in s) or ("." in s))
        if 1e-4 <= x < 1e16:
            assert s == repr(float(x))          # the range where Python's repr is the same notation


def _body_lines(tsv):
    return [l for l in tsv.split("\n") if l and not l.startswith("#")]


@pytest.mark.gpu
def test_report_json_chrM(tmp_path):
    """One run with Hist(bp) + Growth + OrderedGrowth(bp) + Similarity(bp): two of node/bp/edge are never
    asked for together, so the run builds bp only; every section carries the table of the classic
    subcommand and the same numbers as items; the text is what serde_json::to_string_pretty writes."""
    y = _write(tmp_path, f"""
- graph: {CHRM}
  name: chrM demo
  grouping: Sample
  analyses:
    - !Hist
      count_type: Bp
    - !Growth
      coverage: 1,2
      quorum: 0,0.5
    - !OrderedGrowth
      coverage: 1,1
      quorum: 0,0.5
      count_type: Bp
    - !Similarity
      count_type: Bp
""")
    rc, out, err = hl.run_cli(["report", "--json", y])
    assert rc == 0, err
    secs = json.loads(out, object_pairs_hook=OrderedDict)
    assert [s["analysis"] for s in secs] == ["Coverage Histogram", "Pangenome Growth", "Ordered Growth", "Similarity Heatmap"]
    for s in secs:
        assert list(s.keys()) == ["analysis", "run_name", "run_id", "countable", "items", "id", "table", "plot_downloads"]
        assert s["run_name"] == "chrM demo" and s["countable"] == "bp" and s["table"].startswith("`") and s["table"].endswith("`")
        assert s["plot_downloads"] == [["png", "Download as png"], ["svg", "Download as svg"], ["vega-editor", "Open in vega editor"]]
        assert len(s["items"]) == 1 and len(s["items"][0]) == 1
    hist, growth, ordered, sim = secs
    assert hist["run_id"] == "chrm-demo-hist" and hist["id"] == "cov-hist-chrm-demo-hist-bp"
    bar = hist["items"][0]["Bar"]
    assert list(bar.keys()) == ["id", "name", "x_label", "y_label", "labels", "values", "log_toggle"]
    assert bar["values"] == [0.0, 616.0, 31.0, 601.0, 15949.0]      # abacus.rs:1630, the reference's own golden
    assert bar["labels"] == ["0", "1", "2", "3", "4"] and bar["y_label"] == "#bps" and bar["x_label"] == "taxa"
    assert bar["name"] == CHRM and bar["log_toggle"] is True and bar["id"] == hist["id"]
    rc, t_hist, err = hl.run_cli(["hist", "-S", "-c", "bp", CHRM])
    assert _body_lines(hist["table"].strip("`")) == _body_lines(t_hist)
    # growth: MultiBar, NaN row 0 -> 0.0, labels 1..G, quorum label in per cent
    mb = growth["items"][0]["MultiBar"]
    assert list(mb.keys()) == ["id", "names", "x_label", "y_label", "labels", "values", "log_toggle"]
    assert mb["names"] == ["coverage ≥ 1, quorum ≥ 0%", "coverage ≥ 2, quorum ≥ 50%"]
    assert mb["labels"] == ["1", "2", "3", "4"] and mb["log_toggle"] is False and growth["id"] == "pan-growth-chrm-demo-growth-bp"
    rc, t_hg, err = hl.run_cli(["histgrowth", "-S", "-c", "bp", "-l", "1,2", "-q", "0,0.5", CHRM])
    rows = [r.split("\t") for r in _body_lines(t_hg)[4:]]
    for k in range(2):
        assert mb["values"][k][0] == 0.0
        assert [int(np.floor(v)) for v in mb["values"][k][1:]] == [int(r[1 + k]) for r in rows[1:]]
    assert _body_lines(growth["table"].strip("`")) == _body_lines(t_hg)
    # ordered growth: labels are the groups, quorum label without the per cent conversion, y label without '#'
    ob = ordered["items"][0]["MultiBar"]
    assert ordered["id"] == "pan-ordered-growth-chrm-demo-orderedgrowth" and ob["y_label"] == "bps"
    assert ob["names"] == ["coverage ≥ 1, quorum ≥ 0%", "coverage ≥ 1, quorum ≥ 0.5%"]
    rc, t_og, err = hl.run_cli(["ordered-histgrowth", "-S", "-c", "bp", "-l", "1,1", "-q", "0,0.5", CHRM])
    orows = [r.split("\t") for r in _body_lines(t_og)[4:]]
    assert ob["labels"] == [r[0] for r in orows] == ["chm13", "grch38", "HG00438", "HG00621"]
    assert ob["values"][0] == [16569.0, 17147.0, 17183.0, 17197.0]   # session-derived expectation
    for k in range(2):
        assert ob["values"][k] == [float(r[1 + k]) for r in orows]
    assert _body_lines(ordered["table"].strip("`")) == _body_lines(t_og)
    # similarity: heatmap in dendrogram order, labels on both axes
    hm = sim["items"][0]["Heatmap"]
    assert list(hm.keys()) == ["id", "name", "x_labels", "y_labels", "values"]
    rc, t_sim, err = hl.run_cli(["similarity", "-S", "-c", "bp", CHRM])
    srows = [r.split("\t") for r in _body_lines(t_sim)]
    assert hm["x_labels"] == hm["y_labels"] == srows[0][1:]
    assert [[hl.format_f32(np.float32(v)) for v in row] for row in hm["values"]] == [r[1:] for r in srows[1:]]
    # the text itself: two-space pretty printing, one element per line, non-ASCII kept, floats with '.0'
    assert out == json.dumps(secs, indent=2, ensure_ascii=False) + "\n"   # writeln!(out, "{json_text}")
    assert '"coverage ≥ 1, quorum ≥ 0%"' in out


@pytest.mark.gpu
def test_report_requirement_union_builds_all_count_types(tmp_path):
    """Hist(bp) + OrderedGrowth(node) ask for two count types -> GraphBroker builds node, bp AND edge
    (graph_broker.rs:149-160), the Hist analysis reports all three, Growth too; default run name and id"""
    y = _write(tmp_path, f"""
- graph: {CHRM}
  grouping: Haplotype
  analyses:
    - !OrderedGrowth
      count_type: Node
    - !Growth
    - !Hist
      count_type: Bp
""")
    rc, out, err = hl.run_cli(["report", "-j", y])
    assert rc == 0, err
    secs = json.loads(out)
    assert [(s["analysis"], s["countable"]) for s in secs] == [
        ("Coverage Histogram", "node"), ("Coverage Histogram", "bp"), ("Coverage Histogram", "edge"),
        ("Pangenome Growth", "node"), ("Pangenome Growth", "bp"), ("Pangenome Growth", "edge"), ("Ordered Growth", "node")]
    name = f"{CHRM}--Group By Haplotype"
    rid = name.lower().replace(" ", "-").replace("_", "-").replace("#", "-").replace("/", "-").replace('"', "-")
    assert all(s["run_name"] == name for s in secs)
    assert secs[0]["run_id"] == rid + "-hist" and secs[0]["id"] == "cov-hist-" + rid + "-hist-node"
    rc, t_all, err = hl.run_cli(["hist", "-H", "-c", "all", CHRM])
    cols = list(zip(*[r.split("\t")[1:] for r in _body_lines(t_all)[4:]]))
    for k in range(3):
        assert secs[k]["items"][0]["Bar"]["values"] == [float(v) for v in cols[k]]
    # two different by-group count types in one run: the reference refuses
    y2 = _write(tmp_path, f"- graph: {CHRM}\n  analyses:\n    - !OrderedGrowth\n      count_type: Node\n    - !Similarity\n      count_type: Bp\n", "r2.yaml")
    rc, out, err = hl.run_cli(["report", "-j", y2])
    assert rc != 0 and "multiple Abaci By Group" in err


@pytest.mark.gpu
def test_json_flag_of_the_classic_subcommands():
    rc, out, err = hl.run_cli(["histgrowth", "-S", "-c", "node", "-l", "1", "-q", "0", "--json", CHRM])
    assert rc == 0, err
    secs = json.loads(out)
    assert [s["analysis"] for s in secs] == ["Coverage Histogram", "Pangenome Growth"]
    assert secs[0]["items"][0]["Bar"]["values"] == [0.0, 39.0, 29.0, 41.0, 45.0]            # abacus.rs:1525
    assert [int(np.floor(v)) for v in secs[1]["items"][0]["MultiBar"]["values"][0]] == [0, 100, 129, 144, 154]
    rc, out, err = hl.run_cli(["ordered-histgrowth", "-S", "-j", CHRM])
    assert rc == 0, err
    assert json.loads(out)[0]["items"][0]["MultiBar"]["values"] == [[89.0, 106.0, 140.0, 154.0]]
    rc, out, err = hl.run_cli(["table", "--json", CHRM])
    assert rc != 0 and "--json is available" in err


@pytest.mark.gpu
def test_report_nice_true_needs_rank_names(tmp_path):
    """`nice: true` (graph.rs:224-229: a segment's name, parsed as an integer, is its id): accepted where the names are
    the ranks 1..N of the S lines -- the same graph, the same sections as without the flag -- and refused elsewhere,
    where the reference would index node lengths with ids that are not ranks"""
    import json
    ok = tmp_path / "ranks.gfa"
    ok.write_text("H\tVN:Z:1.1\nS\t1\tACGT\nS\t2\tA\nS\t3\tGG\nL\t1\t+\t2\t+\t0M\nL\t2\t+\t3\t+\t0M\n"
                  "P\ta#1#c\t1+,2+,3+\t*\nP\tb#1#c\t1+,3+\t*\n")
    other = tmp_path / "other.gfa"
    other.write_text(ok.read_text().replace("S\t3\t", "S\t7\t").replace("3+", "7+").replace("\t3\t", "\t7\t"))
    outs = []
    for nice in ("true", "false"):
        cfg = tmp_path / f"r_{nice}.yaml"
        cfg.write_text(f"- graph: {ok}\n  nice: {nice}\n  analyses:\n    - !Hist\n      count_type: Bp\n")
        rc, out, err = hl.run_cli(["report", "--json", str(cfg)])
        assert rc == 0, err
        outs.append(json.loads(out))

    def numbers(x):  # every list of numbers in the sections, in order: the plotted data
        if isinstance(x, list) and x and all(isinstance(v, (int, float)) for v in x):
            return [x]
        if isinstance(x, list):
            return [y for v in x for y in numbers(v)]
        if isinstance(x, dict):
            return [y for v in x.values() for y in numbers(v)]
        return []
    assert numbers(outs[0]) == numbers(outs[1]) and numbers(outs[0])
    cfg = tmp_path / "bad.yaml"
    cfg.write_text(f"- graph: {other}\n  nice: true\n  analyses:\n    - !Hist\n      count_type: Bp\n")
    rc, out, err = hl.run_cli(["report", "--json", str(cfg)])
    assert rc != 0 and "nice: true needs segment names" in err
