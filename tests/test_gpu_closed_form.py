"""Quorum closed form with its O(n^3) inner sums on the GPU (pnx_quorum_sums): bit-identical with
the serial restatement of hist.rs:138-187, because the device evaluates the platform libm's exp2
algorithm operation by operation (csrc/exp2_exact.hpp)."""
import numpy as np
import pytest

import oracle as orc
from panacus_amd import capi

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
        ctx.sync()
        ctx.config(capi.CFG_MAX_IN_FLIGHT, 2)
    host_only = hostlib.calc_growths(h, thr)
    for (c, q), a, b in zip(pairs, got, host_only):
        exp = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
        assert a.tobytes() == exp.tobytes(), (n, c, q)
        assert b.tobytes() == exp.tobytes(), (n, c, q)


@pytest.mark.parametrize("route", ["0", "1"])
@pytest.mark.parametrize("n", [3, 130, 301, 385, 700])
def test_quorum_inner_sums_both_routes_bitwise(ctx, monkeypatch, n, route):
    """the inner sums of the quorum pair through HBM (K7a + K7b, PNX_QUORUM_ROUTE=0) and kept in LDS (k_quorum_fused, =1; both of
    its shapes: <= 384 groups and above) == the host path == the oracle, bit for bit"""
    from panacus_amd import hostlib
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold
    monkeypatch.setenv("PNX_QUORUM_ROUTE", route)
    rng = np.random.default_rng(7 * n)
    h = rng.integers(0, 10**7, size=n + 1).astype(np.uint64)
    h[rng.integers(0, n + 1, size=2)] = 0
    pairs = [(1, 0.5), (0, 0.1), (2, 0.9), (max(1, n // 3), 0.3), (1, 2.0 / n), (1, 0.999), (n, 0.5)]
    thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs]
    host_only = hostlib.calc_growths(h, thr)
    hostlib.set_quorum_offload(ctx, min_n=1)
    try:
        ctx.config(capi.CFG_DROP_GROWTH_TABLES, 0)
        got = hostlib.calc_growths(h, thr)
    finally:
        hostlib.set_quorum_offload(None)
        ctx.sync()
        ctx.config(capi.CFG_DROP_GROWTH_TABLES, 0)
        ctx.config(capi.CFG_MAX_IN_FLIGHT, 2)
    for (c, q), a, b in zip(pairs, got, host_only):
        assert a.tobytes() == b.tobytes(), (n, c, q, route)
        if n <= 301:
            assert a.tobytes() == orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q)).tobytes(), (n, c, q, route)


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
        ctx.sync()
        ctx.config(capi.CFG_MAX_IN_FLIGHT, 2)
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


def _log2_arguments():
    rng = np.random.default_rng(1)
    bits = rng.integers(0, 2**63, size=1_000_000, dtype=np.uint64)
    near1 = 1.0 + (rng.random(500_000) - 0.5) * 2.0 ** -4
    return np.concatenate([
        bits.view(np.float64),                                            # any non-negative bit pattern: subnormals, inf, NaN
        rng.integers(0, 2**24, size=500_000).astype(np.float64),          # histogram bins, small integers
        rng.integers(0, 2**53, size=500_000).astype(np.float64),
        near1, np.exp2(2000.0 * rng.random(500_000) - 1000.0),            # the sums of the quorum branch
        np.arange(0, 5000, dtype=np.float64),
        np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
                  1.0 - float.fromhex('0x1.5b51p-5'), 1.0 + float.fromhex('0x1.6ab2p-5'), np.nextafter(1.0, 0), np.nextafter(1.0, 2)])])


def test_device_log2_equals_libm_bitwise(ctx):
    from panacus_amd import hostlib
    assert hostlib.device_growth_usable()
    x = _log2_arguments()
    exp = orc.log2(x)
    for got in (ctx.log2_exact(x), hostlib.log2_restated(x)):
        nan_both = np.isnan(got) & np.isnan(exp)
        bad = np.flatnonzero((got.view(np.uint64) != exp.view(np.uint64)) & ~nan_both)
        assert bad.size == 0, (x[bad[:5]], got[bad[:5]], exp[bad[:5]])


@pytest.mark.parametrize("n", [2, 5, 44, 130, 256, 257, 301, 700, 1024, 1025])
def test_whole_closed_forms_on_the_device_bitwise(ctx, n):
    """pnx_growth_closed_form: union, core and quorum curves from the histogram to the value on the device == the serial
    restatement of hist.rs:89-187 == the host path, bit for bit; also straight from the counters of a coverage pass"""
    from panacus_amd import hostlib
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold
    rng = np.random.default_rng(100 + n)
    h = rng.integers(0, 10**7, size=n + 1).astype(np.uint64)
    h[rng.integers(0, n + 1, size=2)] = 0
    pairs = [(1, 0.5), (0, 0.1), (2, 0.9), (max(1, n // 3), 0.3), (1, 2.0 / n), (1, 0.0), (2, 0.0), (1, 1.0), (3, 1.0), (n + 5, 0.0),
             (n, 0.5), (1, 0.999)]
    thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs]
    host_only = hostlib.calc_growths(h, thr)
    hostlib.set_quorum_offload(ctx, min_n=1)
    try:
        pending = hostlib.calc_growths_begin(h, thr)
        pending2 = hostlib.calc_growths_begin(h[::-1].copy(), thr[:3])   # two calls in flight
        got = hostlib.calc_growths_end(pending)
        got2 = hostlib.calc_growths_end(pending2)
    finally:
        hostlib.set_quorum_offload(None)
        ctx.sync()
        ctx.config(capi.CFG_MAX_IN_FLIGHT, 2)
    for (c, q), a, b in zip(pairs, got, host_only):
        exp = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
        assert a.tobytes() == exp.tobytes(), (n, c, q)
        assert b.tobytes() == exp.tobytes(), (n, c, q)
    for (c, q), a in zip(pairs[:3], got2):
        assert a.tobytes() == orc.growth(h[::-1].copy(), (orc.ABSOLUTE, c), (orc.RELATIVE, q)).tobytes()


def test_closed_forms_follow_the_pass_on_the_device(ctx):
    """hist == NULL: the curves are computed from the device counters of the pass enqueued last, two passes in flight"""
    from panacus_amd import capi, hostlib
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold
    n, p = 200_000, 96
    items, pre, _ = orc.pansyn(13, n, p)
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5), (1, 1.0)]
    thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs]
    ctx.set_csr_pansyn(13, n, p)
    hostlib.set_quorum_offload(ctx, min_n=1)
    try:
        # four passes and their closed forms in flight
        ctx.config(capi.CFG_MAX_IN_FLIGHT, 4)
        pi = np.arange(p, dtype=np.uint32)
        ctx.set_order(pi, pi, p)
        pend = []
        for _ in range(4):
            ctx.hist_async()
            pend.append(hostlib.calc_growths_begin_on_device(p, thr))
        builds = int(ctx.info().n_growth_table_builds)
        extra = hostlib.calc_growths_begin_on_device(p, thr)   # two calls more than passes may be in flight (the last pass again)
        extra2 = hostlib.calc_growths_begin_on_device(p, thr)
        assert all(x is not None for x in pend) and extra is not None and extra2 is not None
        assert hostlib.calc_growths_begin_on_device(p, thr) is None   # a seventh: refused
        oh = orc.hist(orc.coverage(items, pre, pi.astype(np.uint64), pi.astype(np.uint64), n), p)
        for x in pend:
            assert np.array_equal(ctx.hist_fetch()[1], oh)
            for (c, q), a in zip(pairs, hostlib.calc_growths_end(x)):
                assert a.tobytes() == orc.growth(oh, (orc.ABSOLUTE, c), (orc.RELATIVE, q)).tobytes()
        for x in (extra, extra2):
            for (c, q), a in zip(pairs, hostlib.calc_growths_end(x)):
                assert a.tobytes() == orc.growth(oh, (orc.ABSOLUTE, c), (orc.RELATIVE, q)).tobytes()
        # the (n, thresholds) tables were derived once and read by the calls that followed; other thresholds derive them again,
        # with calls in flight too, and dropping them changes nothing but the time
        assert int(ctx.info().n_growth_table_builds) == builds
        ctx.hist_async()
        a1 = hostlib.calc_growths_begin_on_device(p, thr)
        a2 = hostlib.calc_growths_begin_on_device(p, thr[1:])
        ctx.config(capi.CFG_DROP_GROWTH_TABLES, 0)
        a3 = hostlib.calc_growths_begin_on_device(p, thr[1:])
        assert int(ctx.info().n_growth_table_builds) == builds + 2
        ctx.hist_fetch()
        for pr, x in ((pairs, a1), (pairs[1:], a2), (pairs[1:], a3)):
            for (c, q), a in zip(pr, hostlib.calc_growths_end(x)):
                assert a.tobytes() == orc.growth(oh, (orc.ABSOLUTE, c), (orc.RELATIVE, q)).tobytes()
        ctx.config(capi.CFG_MAX_IN_FLIGHT, 2)
        orders = [(np.arange(p, dtype=np.uint32), np.arange(p, dtype=np.uint32), p),
                  (np.arange(p, dtype=np.uint32), (np.arange(p) // 3).astype(np.uint32), p // 3)]
        for pi, gi, G in orders:
            ctx.set_order(pi, gi, G)
            ctx.hist_async()
            g1 = hostlib.calc_growths_begin_on_device(G, thr)
            ctx.hist_async()
            g2 = hostlib.calc_growths_begin_on_device(G, thr)
            assert g1 is not None and g2 is not None
            _, h1 = ctx.hist_fetch()
            _, h2 = ctx.hist_fetch()
            c1, c2 = hostlib.calc_growths_end(g1), hostlib.calc_growths_end(g2)
            oh = orc.hist(orc.coverage(items, pre, pi.astype(np.uint64), gi.astype(np.uint64), n), G)
            assert np.array_equal(h1, oh) and np.array_equal(h2, oh)
            for (c, q), a, b in zip(pairs, c1, c2):
                exp = orc.growth(oh, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
                assert a.tobytes() == exp.tobytes() and b.tobytes() == exp.tobytes(), (G, c, q)
    finally:
        hostlib.set_quorum_offload(None)
        ctx.sync()
        ctx.config(capi.CFG_MAX_IN_FLIGHT, 2)


def test_reference_vectors_through_the_device_closed_forms(ctx, golden):
    """the reference's own numbers for hist -> growth -- the three exact-equality f64 vectors of hist.rs:352-398 and the 660
    floored values of docs/chr22.hprc-v1.0-pggb.histgrowth.html:266-276 -- with every log2 / exp2 / addition taken on the GPU"""
    import math
    from panacus_amd import hostlib
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold
    hostlib.set_quorum_offload(ctx, min_n=1)
    try:
        ka = golden["growth_known_answers"]
        # (the reference's tests call the three branch functions directly; so does this, through the ABI: branch 0 / 1 / 2,
        # coverage max(1, 0) = 1)
        for branch, kind in enumerate(("union", "core", "quorum")):
            k = ka[kind]
            n = len(k["hist"]) - 1
            shape = ctx.growth_closed_form_async(k["hist"], n, [branch], [1], [k.get("quorum", 0.0)])
            assert ctx.growth_closed_form_fetch(shape)[0].tolist() == k["expected"], kind
        rep = golden["chr22_report"]
        n_vals = 0
        for count in ("bp", "node", "edge"):
            h = np.array(rep["hists"][count], dtype=np.uint64)
            gr = rep["growths"][count]
            thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in zip(gr["coverage"], gr["quorum"])]
            for got, curve in zip(hostlib.calc_growths(h, thr), gr["curves"]):
                assert [int(math.floor(x)) for x in got] == curve
                n_vals += len(curve)
        assert n_vals == 660
    finally:
        hostlib.set_quorum_offload(None)


def test_histgrowth_in_one_native_call_equals_the_calls_one_by_one_and_the_oracle():
    """pnh_histgrowth_resident: pass + histogram + the curves of every pair from the resident steps in one call of the host library
    -- the same numbers as pnx_hist followed by the closed forms, and as the oracle's (hist.rs:68-187 restated), cold and warm,
    with the curves from the device (offload context) and from the host threads"""
    import oracle as orc
    from panacus_amd import capi, hostlib
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold
    n, p = 400_000, 300
    pairs = [(Threshold(ABSOLUTE, 1), Threshold(RELATIVE, 0.0)), (Threshold(ABSOLUTE, 2), Threshold(RELATIVE, 0.0)),
             (Threshold(ABSOLUTE, 1), Threshold(RELATIVE, 0.5))]
    items, pre, _ = orc.pansyn(9, n, p)
    pi = np.arange(p, dtype=np.uint64)
    oh = orc.hist(orc.coverage(items, pre, pi, pi, n), p)
    want = [orc.growth(oh, (orc.ABSOLUTE, int(c.value)), (orc.RELATIVE, q.value)) for c, q in pairs]
    with capi.Context(0) as ctx:
        ctx.set_csr_pansyn(9, n, p, with_weights=False)
        order = np.arange(p, dtype=np.uint32)
        ctx.set_order(order, order, p)
        for offload in (True, False):
            hostlib.set_quorum_offload(ctx if offload else None, 256)
            try:
                for cold in (True, False, True):
                    h, g = hostlib.histgrowth_resident(ctx, p, pairs, drop_derived=cold, drop_tables=cold)
                    assert np.array_equal(h, oh)
                    for a, b in zip(g, want):
                        assert a.tobytes() == np.asarray(b, dtype=np.float64).tobytes()
            finally:
                hostlib.set_quorum_offload(None)
        _, h2 = ctx.hist(want_countable=False)
        assert np.array_equal(h2, oh)


def test_histgrowth_in_one_native_call_on_a_context_that_is_not_the_offload_context():
    """pnh_histgrowth_resident(ctx) while ANOTHER context is the offload context (round-5 advisor): "the counters of the last pass"
    are those of the offload context -- with a different graph and a different number of groups there -- so the call must take
    its curves from ITS OWN histogram (host threads, or the device fed with that histogram), never from the other context's
    pass.  Both contexts' results equal the oracle's."""
    import oracle as orc
    from panacus_amd import capi, hostlib
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold
    pairs = [(Threshold(ABSOLUTE, 1), Threshold(RELATIVE, 0.0)), (Threshold(ABSOLUTE, 1), Threshold(RELATIVE, 0.5))]

    def want(seed, n, p):
        items, pre, _ = orc.pansyn(seed, n, p)
        pi = np.arange(p, dtype=np.uint64)
        oh = orc.hist(orc.coverage(items, pre, pi, pi, n), p)
        return oh, [orc.growth(oh, (orc.ABSOLUTE, int(c.value)), (orc.RELATIVE, q.value)) for c, q in pairs]

    (na, pa), (nb, pb) = (150_000, 320), (90_000, 260)
    oh_a, g_a = want(3, na, pa)
    oh_b, g_b = want(4, nb, pb)
    with capi.Context(0) as a, capi.Context(0) as b:
        for c, seed, n, p in ((a, 3, na, pa), (b, 4, nb, pb)):
            c.set_csr_pansyn(seed, n, p, with_weights=False)
            o = np.arange(p, dtype=np.uint32)
            c.set_order(o, o, p)
        hostlib.set_quorum_offload(b, 256)
        try:
            b.hist(want_countable=False)   # b's last pass: 260 groups
            for _ in range(2):
                h, g = hostlib.histgrowth_resident(a, pa, pairs, drop_derived=True, drop_tables=True)   # a is NOT the offload context
                assert np.array_equal(h, oh_a)
                for x, y in zip(g, g_a):
                    assert x.tobytes() == np.asarray(y, dtype=np.float64).tobytes()
                h, g = hostlib.histgrowth_resident(b, pb, pairs, drop_derived=True, drop_tables=True)   # b is
                assert np.array_equal(h, oh_b)
                for x, y in zip(g, g_b):
                    assert x.tobytes() == np.asarray(y, dtype=np.float64).tobytes()
        finally:
            hostlib.set_quorum_offload(None)
