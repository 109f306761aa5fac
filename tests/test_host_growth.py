"""Host closed-form growth (panacus_amd/host/growth_closed_form.cpp) against the reference's
known answers and, bit for bit, against the oracle restatement."""
import math

import numpy as np
import pytest

import oracle as orc
from panacus_amd import hostlib
from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold


def test_known_answers_bit_exact(golden):
    ka = golden["growth_known_answers"]
    c0 = Threshold(ABSOLUTE, 0)
    assert hostlib.calc_growth_branch("union", ka["union"]["hist"], c0, Threshold(RELATIVE, 0)).tolist() == ka["union"]["expected"]
    assert hostlib.calc_growth_branch("core", ka["core"]["hist"], c0, Threshold(RELATIVE, 1)).tolist() == ka["core"]["expected"]
    assert hostlib.calc_growth_branch("quorum", ka["quorum"]["hist"], c0, Threshold(RELATIVE, 0.9)).tolist() == ka["quorum"]["expected"]


def test_chr22_report(golden):
    rep = golden["chr22_report"]
    for count in ("bp", "node", "edge"):
        h = rep["hists"][count]
        gr = rep["growths"][count]
        for c, q, curve in zip(gr["coverage"], gr["quorum"], gr["curves"]):
            got = hostlib.calc_growth(h, Threshold(ABSOLUTE, c), Threshold(RELATIVE, q))
            assert [int(math.floor(x)) for x in got] == curve


@pytest.mark.parametrize("n", [5, 44, 130, 257])
@pytest.mark.parametrize("threads", [1, 4])
def test_matches_oracle_bitwise(n, threads):
    rng = np.random.default_rng(n)
    h = rng.integers(0, 10**7, size=n + 1).astype(np.uint64)
    h[rng.integers(0, n + 1, size=3)] = 0  # log2(0) = -inf terms
    for c in (0, 1, 2, n // 3):
        for q in (0.0, 0.1, 0.5, 0.9, 1.0, 1.0 / n):
            a = hostlib.calc_growth(h, Threshold(ABSOLUTE, c), Threshold(RELATIVE, q), threads)
            b = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
            assert a.tobytes() == b.tobytes(), (n, c, q)


def test_all_pairs_in_one_region_bitwise():
    rng = np.random.default_rng(7)
    for n in (3, 60, 256):
        h = rng.integers(0, 10**6, size=n + 1).astype(np.uint64)
        pairs = [(0, 0.0), (1, 0.0), (2, 0.0), (1, 0.5), (1, 1.0), (3, 0.25), (1, 0.1)]
        got = hostlib.calc_growths(h, [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs])
        for (c, q), g in zip(pairs, got):
            assert g.tobytes() == orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q)).tobytes(), (n, c, q)


def test_quorum_large_n_bitwise():
    """n where the tails of the quorum sums are far below half an ulp of the running sum (the
    host skips those libm calls): still bit-identical with the serial restatement"""
    rng = np.random.default_rng(11)
    n = 420
    h = rng.integers(1, 10**6, size=n + 1).astype(np.uint64)
    h[17] = 0
    for c, q in ((1, 0.5), (1, 0.05), (3, 0.95), (40, 0.3)):
        a = hostlib.calc_growth(h, Threshold(ABSOLUTE, c), Threshold(RELATIVE, q))
        b = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
        assert a.tobytes() == b.tobytes(), (c, q)


def test_exp2_restatement_matches_this_libm():
    """the device path of the quorum closed form rests on this; on a libm with another exp2 the
    library detects it and stays on the host"""
    assert hostlib.quorum_offload_usable()


def test_log2_restatement_matches_libm_bitwise():
    """csrc/log2_exact.hpp (the restated glibc log2, table from tools/gen_log2_table.py) against the platform libm: what
    lets whole closed forms run on the device (hostlib.device_growth_usable is the run-time form of this test)"""
    import oracle as orc
    from panacus_amd import hostlib
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.integers(0, 2**63, size=400_000, dtype=np.uint64).view(np.float64),
                        rng.integers(0, 2**30, size=200_000).astype(np.float64),
                        1.0 + (rng.random(200_000) - 0.5) * 2.0 ** -4,
                        np.exp2(2000.0 * rng.random(200_000) - 1000.0),
                        np.array([0.0, 1.0, np.inf, 5e-324, 2.2250738585072014e-308, np.nextafter(1.0, 0), np.nextafter(1.0, 2)])])
    got, exp = hostlib.log2_restated(x), orc.log2(x)
    both_nan = np.isnan(got) & np.isnan(exp)
    assert not ((got.view(np.uint64) != exp.view(np.uint64)) & ~both_nan).any()
    assert hostlib.device_growth_usable()
