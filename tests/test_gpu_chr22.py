"""BASELINE.json configs[4]: the documentation example on the HPRC v1.0 pggb chr22 graph
(/root/reference/examples/pangenome_growth_pggb.md:13-22) against the numbers the reference publishes for it
(docs/chr22.hprc-v1.0-pggb.histgrowth.html:266-276: three 45-bin histograms and 3 x 5 growth curves of 44 values,
kept in tests/golden/golden.json by make_golden.py).

The graph is a 402 MB download that is not in the build image (no network), so the test SKIPS unless the file is found
-- under PANACUS_CHR22_GFA, or dropped in at the fixed place tests/golden/chr22.hprc-v1.0-pggb.gfa[.gz] (git-ignored):

    wget -P tests/golden https://s3-us-west-2.amazonaws.com/human-pangenomics/pangenomes/freeze/freeze1/pggb/chroms/chr22.hprc-v1.0-pggb.gfa.gz
    python -m pytest tests/test_gpu_chr22.py -m gpu

With the file present it pins `hist` itself (coverage on the device, all three count types, -S grouping, the subset
list of the example) on real data, not only the hist -> growth half that the 660 values pin without the graph."""
import gzip
import math
import os
import time

import pytest

from panacus_amd import hostlib as hl

pytestmark = pytest.mark.gpu

_HERE = os.path.dirname(os.path.abspath(__file__))
GFA = next((f for f in (os.environ.get("PANACUS_CHR22_GFA", ""), os.path.join(_HERE, "golden", "chr22.hprc-v1.0-pggb.gfa"),
                        os.path.join(_HERE, "golden", "chr22.hprc-v1.0-pggb.gfa.gz")) if f and os.path.exists(f)), "")


def _body_rows(text):
    return [l.split("\t") for l in text.split("\n") if l and not l.startswith("#")]


@pytest.mark.skipif(not (GFA and os.path.exists(GFA)), reason="the chr22 pggb graph is neither under PANACUS_CHR22_GFA nor in tests/golden/ (see the module docstring)")
@pytest.mark.parametrize("cname", ["node", "bp", "edge"])
def test_chr22_example_reproduces_the_published_report(golden, tmp_path, cname):
    # step 2 of the example: every path that is not a reference
    sub = str(tmp_path / "haplotypes.txt")
    opener = gzip.open if GFA.endswith(".gz") else open
    with opener(GFA, "rt") as f, open(sub, "w") as out:
        for line in f:
            if line.startswith("P\t"):
                name = line.split("\t", 2)[1]
                if "grch38" not in name and "chm13" not in name:
                    out.write(name + "\n")
    # step 3, per count type
    t0 = time.perf_counter()
    rc, text, err = hl.run_cli(["histgrowth", "-c", cname, "-l", "1,2,1,1,1", "-q", "0,0,1,0.5,0.1", "-S", "-a", "-s", sub, GFA])
    dt = time.perf_counter() - t0
    assert rc == 0, err
    rows = _body_rows(text)
    assert rows[0][:2] == ["panacus", "hist"] and rows[1][1] == cname
    data = rows[4:]
    rep = golden["chr22_report"]
    assert [int(r[1]) for r in data] == rep["hists"][cname], "coverage histogram"
    curves = rep["growths"][cname]["curves"]
    assert rep["growths"][cname]["coverage"] == [1, 2, 1, 1, 1] and rep["growths"][cname]["quorum"] == [0.0, 0.0, 1.0, 0.5, 0.1]
    for k in range(5):
        got = [r[2 + k] for r in data[1:]]
        assert got == [hl.format_f64(float(math.floor(x))) for x in curves[k]], ("growth", k)
    print(f"chr22 histgrowth -c {cname}: {dt:.2f} s whole call")
