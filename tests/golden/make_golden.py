#!/usr/bin/env python3
"""Extracts the reference's own known-answer vectors for the hist/growth path into
tests/golden/ (data only -- numbers and test input files, never reference source text).

Run once in the build container (needs /root/reference); the outputs are committed so the
tests never touch /root/reference at run time.

Sources (paths relative to the reference root):
  test/chrM_test.gfa, test/cdbg.gfa, test/test_groups.txt     -- test input graphs
  tests/test_files/t_groups.gfa, tests/test_files/t_groups.hist.tsv
  src/graph_broker/abacus.rs:1424-1435,1487-1630   -- expected countables / hists / groups
  src/graph_broker/hist.rs:342-398                 -- closed-form growth f64 vectors
  docs/chr22.hprc-v1.0-pggb.histgrowth.html:266-276 -- hist -> growth arrays (panacus 0.2.2)
"""
import json
import os
import re
import shutil

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def ints(s):
    return [int(x) for x in re.findall(r"\d+", s)]


def main():
    for src in ("test/chrM_test.gfa", "test/cdbg.gfa", "test/test_groups.txt",
                "tests/test_files/t_groups.gfa", "tests/test_files/t_groups.hist.tsv"):
        shutil.copyfile(os.path.join(REF, src), os.path.join(OUT, os.path.basename(src)))

    golden = {}
    ab = open(os.path.join(REF, "src/graph_broker/abacus.rs")).read().split("\n")

    def block(lo, hi):
        # 1-based inclusive line range, comment markers stripped
        return "\n".join(l.lstrip().lstrip("/") for l in ab[lo - 1:hi])

    def countable(lo, hi):
        txt = block(lo, hi)
        m = re.search(r"countable:\s*vec!\[(.*?)\]", txt, re.S)
        body = m.group(1).replace("CountSize::MAX", "4294967295")
        return ints(body)

    def groups(lo, hi):
        txt = block(lo, hi)
        m = re.search(r"groups:\s*vec!\[(.*?)\]", txt, re.S)
        return re.findall(r'"([^"]+)"', m.group(1))

    def hist(lo, hi):
        m = re.search(r"test_hist = vec!\[(.*?)\]", block(lo, hi))
        return ints(m.group(1))

    golden["cdbg_node"] = {"countable": countable(1420, 1436), "groups": groups(1420, 1436),
                           "source": "src/graph_broker/abacus.rs:1424-1435"}
    golden["chrM_sample_node"] = {"countable": countable(1480, 1527), "groups": groups(1480, 1527),
                                  "hist": hist(1480, 1527),
                                  "source": "src/graph_broker/abacus.rs:1487-1525"}
    golden["chrM_sample_edge"] = {"countable": countable(1529, 1581), "groups": groups(1529, 1581),
                                  "hist": hist(1529, 1581),
                                  "source": "src/graph_broker/abacus.rs:1537-1579"}
    golden["chrM_sample_bp"] = {"countable": countable(1583, 1632), "groups": groups(1583, 1632),
                                "hist": hist(1583, 1632),
                                "source": "src/graph_broker/abacus.rs:1591-1630"}

    # t_groups node hist (hist of t_groups.gfa without grouping)
    rows = [l.split("\t") for l in open(os.path.join(REF, "tests/test_files/t_groups.hist.tsv"))
            if l[0].isdigit()]
    golden["t_groups_node_hist"] = {"hist": [int(r[1]) for r in rows],
                                    "source": "tests/test_files/t_groups.hist.tsv"}

    # closed-form growth f64 known answers
    hs = open(os.path.join(REF, "src/graph_broker/hist.rs")).read()

    def f64vec(name):
        m = re.search(name + r": Vec<f64> = vec!\[(.*?)\];", hs, re.S)
        return [float(x) for x in re.findall(r"[0-9.]+", m.group(1))]

    golden["growth_known_answers"] = {
        "union": {"hist": [0, 5, 3, 2], "coverage": 0, "expected": f64vec("test_growth")},
        "core": {"hist": [0, 5, 3, 2], "coverage": 0, "expected": f64vec("test_core")},
        "quorum": {"hist": [0, 5, 3, 2, 3, 5, 0, 4, 2, 1], "coverage": 0, "quorum": 0.9,
                   "expected": [float(x) for x in re.findall(
                       r"[0-9.]+", re.search(r"fn test_hist_calc_growth_quorum.*?test_growth: Vec<f64> = vec!\[(.*?)\];",
                                             hs, re.S).group(1))]},
        "source": "src/graph_broker/hist.rs:352-398",
    }

    # chr22 report: hist arrays and floored growth curves
    html = open(os.path.join(REF, "docs/chr22.hprc-v1.0-pggb.histgrowth.html")).read()
    chr22 = {"source": "docs/chr22.hprc-v1.0-pggb.histgrowth.html:266-276", "hists": {}, "growths": {}}
    for m in re.finditer(r"new Hist\('(\w+)', \[(.*?)\], \[(.*?)\]\)", html):
        chr22["hists"][m.group(1)] = ints(m.group(3))
    for m in re.finditer(r"new Growth\('(\w+)', \[(.*?)\], \[(.*?)\], \[(.*?)\], \[\[(.*?)\]\]\)", html):
        cov = ints(m.group(3))
        quo = [float(x) for x in re.findall(r"[0-9.]+", m.group(4))]
        curves = [ints(c) for c in m.group(5).split("], [")]
        chr22["growths"][m.group(1)] = {"coverage": cov, "quorum": quo, "curves": curves}
    golden["chr22_report"] = chr22

    with open(os.path.join(OUT, "golden.json"), "w") as f:
        json.dump(golden, f, indent=1)
    n = sum(len(c) for g in chr22["growths"].values() for c in g["curves"])
    print("chr22 growth values:", n)


if __name__ == "__main__":
    main()
