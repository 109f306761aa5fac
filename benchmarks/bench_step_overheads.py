"""Where a pipelined histgrowth step spends its time on the HOST side: the same loop with and without the closed forms,
with 1..4 passes in flight (10 M x 256, resident)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panacus_amd import capi, hostlib  # noqa: E402
from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold  # noqa: E402


def main():
    n, p, steps = 10_000_000, 256, 300
    thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in ((1, 0.0), (2, 0.0), (1, 0.5))]
    out = {}
    with capi.Context(0) as ctx:
        ctx.set_csr_pansyn(42, n, p)
        order = np.arange(p, dtype=np.uint32)
        ctx.set_order(order, order, p)
        ctx.hist()
        hostlib.set_quorum_offload(ctx, 256)
        for depth in (1, 2, 4):
            ctx.config(capi.CFG_MAX_IN_FLIGHT, depth)
            for mode in ("pass_only", "pass+growth"):
                def enqueue():
                    ctx.hist_async()
                    return hostlib.calc_growths_begin_on_device(p, thr) if mode == "pass+growth" else None
                q = [enqueue() for _ in range(depth)]
                t0 = time.perf_counter()
                for k in range(steps):
                    ctx.hist_fetch(want_countable=False)
                    g = q.pop(0)
                    if k + depth < steps + depth:
                        q.append(enqueue())
                    if g is not None:
                        hostlib.calc_growths_end(g)
                for g in q:
                    ctx.hist_fetch(want_countable=False)
                    if g is not None:
                        hostlib.calc_growths_end(g)
                dt = (time.perf_counter() - t0) / (steps + depth) * 1e3
                ctx.sync()
                out[f"depth{depth}_{mode}_ms"] = round(dt, 4)
        # where the host's time goes at depth 4: the asynchronous calls (pure host work) and the two waits
        ctx.config(capi.CFG_MAX_IN_FLIGHT, 4)
        none = 0
        acc = {"hist_async": 0.0, "growth_begin": 0.0, "hist_fetch(wait)": 0.0, "growth_end(wait)": 0.0}
        pc = time.perf_counter

        def enq():
            a = pc(); ctx.hist_async(); b = pc()
            g = hostlib.calc_growths_begin_on_device(p, thr); c = pc()
            acc["hist_async"] += b - a; acc["growth_begin"] += c - b
            assert g is not None
            return g
        q = [enq() for _ in range(4)]
        for v in acc: acc[v] = 0.0
        for k in range(steps):
            a = pc(); ctx.hist_fetch(want_countable=False); b = pc()
            acc["hist_fetch(wait)"] += b - a
            g = q.pop(0)
            q.append(enq())
            a = pc(); hostlib.calc_growths_end(g); b = pc()
            acc["growth_end(wait)"] += b - a
        for g in q:
            ctx.hist_fetch(want_countable=False); hostlib.calc_growths_end(g)
        ctx.sync()
        out["depth4_host_us_per_step"] = {k: round(v / steps * 1e6, 1) for k, v in acc.items()}
        # pure host cost of the calls (nothing to wait for: tiny graph)
        hostlib.set_quorum_offload(None)
    with capi.Context(0) as c2:
        c2.set_csr_pansyn(1, 3000, 8)
        o = np.arange(8, dtype=np.uint32)
        c2.set_order(o, o, 8)
        c2.hist()
        t0 = time.perf_counter()
        for _ in range(2000):
            c2.hist_async()
            c2.hist_fetch(want_countable=False)
        out["tiny_graph_async_fetch_ms"] = round((time.perf_counter() - t0) / 2000 * 1e3, 4)
    # the same pipelined loop on a graph whose kernels take no time: what the host and the launch path alone allow
    with capi.Context(0) as c3:
        c3.set_csr_pansyn(1, 3000, p)
        c3.set_order(order, order, p)
        c3.hist()
        hostlib.set_quorum_offload(c3, 256)
        for depth in (1, 4):
            c3.config(capi.CFG_MAX_IN_FLIGHT, depth)
            for mode in ("pass_only", "pass+growth"):
                def enq3():
                    c3.hist_async()
                    return hostlib.calc_growths_begin_on_device(p, thr) if mode == "pass+growth" else None
                q = [enq3() for _ in range(depth)]
                t0 = time.perf_counter()
                for k in range(1000):
                    c3.hist_fetch(want_countable=False)
                    g = q.pop(0)
                    q.append(enq3())
                    if g is not None:
                        hostlib.calc_growths_end(g)
                dt = (time.perf_counter() - t0) / 1000 * 1e3
                for g in q:
                    c3.hist_fetch(want_countable=False)
                    if g is not None:
                        hostlib.calc_growths_end(g)
                c3.sync()
                out[f"tiny_depth{depth}_{mode}_ms"] = round(dt, 4)
        hostlib.set_quorum_offload(None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
