"""Quorum closed form with its O(n^3) inner sums on the GPU (pnx_quorum_sums): bit-identical with
the serial restatement of hist.rs:138-187, because the device evaluates the platform libm's exp2
algorithm operation by operation (csrc/exp2_exact.hpp)."""
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


def test_device_exp2_equals_libm_bitwise(ctx):
    rng = np.random.default_rng(0)
    x = np.concatenate([
        -1100.0 * rng.random(1_500_000), -60.0 * rng.random(1_500_000), -1080.0 + 10.0 * rng.random(500_000),
        40.0 * rng.random(500_000) - 20.0, 2000.0 * rng.random(100_000) - 1000.0,
        np.array([0.0, -0.0, 1.0, -1.0, 1023.0, 1023.9999, 1024.0, -1022.0, -1022.5, -1074.0, -1074.9999, -1075.0, -1076.0,
                  1e-300, -1e-300, 2.0 ** -54, 2.0 ** -55, -np.inf, np.inf, 928.0, -928.0, 928.0000001, -928.0000001])])
    got = ctx.exp2_exact(x)
    exp = orc.exp2(x)  # the platform libm (numpy's exp2 is its own implementation)
    bad = np.flatnonzero(got.view(np.uint64) != exp.view(np.uint64))
    assert bad.size == 0, (x[bad[:5]], got[bad[:5]], exp[bad[:5]])


@pytest.mark.parametrize("n", [5, 44, 130, 301])
def test_quorum_growth_offload_bitwise(ctx, n):
    from panacus_amd import hostlib
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold
    assert hostlib.quorum_offload_usable()
    rng = np.random.default_rng(n)
    h = rng.integers(0, 10**7, size=n + 1).astype(np.uint64)
    h[rng.integers(0, n + 1, size=2)] = 0
    pairs = [(1, 0.5), (0, 0.1), (2, 0.9), (max(1, n // 3), 0.3), (1, 2.0 / n), (1, 0.0), (1, 1.0)]
    thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs]
    hostlib.set_quorum_offload(ctx, min_n=1)
    try:
        got = hostlib.calc_growths(h, thr)
    finally:
        hostlib.set_quorum_offload(None)
    host_only = hostlib.calc_growths(h, thr)
    for (c, q), a, b in zip(pairs, got, host_only):
        exp = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
        assert a.tobytes() == exp.tobytes(), (n, c, q)
        assert b.tobytes() == exp.tobytes(), (n, c, q)


def test_quorum_growth_offload_large_n(ctx):
    """n = 1024 (the north_star shape): offload == host path bit for bit"""
    from panacus_amd import hostlib
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold
    rng = np.random.default_rng(5)
    n = 1024
    h = rng.integers(1, 10**6, size=n + 1).astype(np.uint64)
    thr = [(Threshold(ABSOLUTE, 1), Threshold(RELATIVE, 0.5)), (Threshold(ABSOLUTE, 3), Threshold(RELATIVE, 0.2))]
    host_only = hostlib.calc_growths(h, thr)
    hostlib.set_quorum_offload(ctx, min_n=512)
    try:
        got = hostlib.calc_growths(h, thr)
    finally:
        hostlib.set_quorum_offload(None)
    for a, b in zip(got, host_only):
        assert a.tobytes() == b.tobytes()


def test_cli_histgrowth_many_groups_uses_the_offload(tmp_path):
    """520 paths: the CLI's quorum closed form takes the GPU path; output == oracle, and == the
    host-only run (PANACUS_AMD_HOST_GROWTH=1)"""
    import math
    import os
    from panacus_amd import hostlib as hl
    gfa = str(tmp_path / "many.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "3000", "--paths", "520", "--seed", "3", "-o", gfa])
    assert rc == 0, err
    args = ["histgrowth", "-c", "node", "-l", "1,2", "-q", "0.5,0.1", "-a", gfa]
    rc, out, err = hl.run_cli(args)
    assert rc == 0, err
    os.environ["PANACUS_AMD_HOST_GROWTH"] = "1"
    try:
        rc, out_host, err = hl.run_cli(args)
    finally:
        del os.environ["PANACUS_AMD_HOST_GROWTH"]
    assert rc == 0, err
    body = lambda t: [l.split("\t") for l in t.split("\n") if l and not l.startswith("#")]
    rows, rows_host = body(out), body(out_host)
    assert rows == rows_host
    g = orc.Graph(gfa)
    pi, gi, names = g.path_order(orc.GROUP_PATHID)
    assert len(names) == 520
    items, pre = g.item_table(orc.NODE)
    h = orc.hist(orc.coverage(items, pre, pi, gi, g.n_nodes), 520)
    assert [int(r[1]) for r in rows[4:]] == h.tolist()
    for k, (c, q) in enumerate(((1, 0.5), (2, 0.1))):
        exp = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
        assert [r[2 + k] for r in rows[5:]] == [hl.format_f64(math.floor(x)) for x in exp], (c, q)
