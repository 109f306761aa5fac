"""What a graph with ONE path that is not sorted costs on its first sweep, beside the same graph with every path sorted and
through the path rows: a path with no order at all (shuffled) is stored a second time at upload, sorted, and read like any other
(upload_scan.hip); one that is shuffled over its last 45 % is recognised by the index kernel, its group left to a bitmap that
the pass's tail folds in (kernels_band.hip: BandLoose); two steps swapped across bands are spilled and added by the tail.
None runs a pass again (reruns_total stays 0).

  python benchmarks/bench_band_fallback.py [--nodes 6000000] [--paths 16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as orc  # noqa: E402  (the CPU generator only: the graph has to be edited on the host)
from panacus_amd import capi  # noqa: E402


def first_hist_ms(ctx, items32, pre, n, order, route, reps=3):
    ts = []
    for _ in range(reps):
        ctx.config(capi.CFG_COVER_ROUTE, route)
        ctx.set_csr(items32, pre, n)
        ctx.set_order(order, order, len(order))
        ctx.config(capi.CFG_DROP_DERIVED, 0)   # (the rows route derives its rows at upload: the call is timed from the steps, as the one-shot route's)
        t0 = time.perf_counter()
        cnt, h = ctx.hist(want_countable=False)
        ts.append((time.perf_counter() - t0) * 1e3)
    # the kernels of one more such call, timed with HIP events (the events hold the chain up: not part of the wall clock above)
    ctx.config(capi.CFG_COVER_ROUTE, route)
    ctx.set_csr(items32, pre, n)
    ctx.set_order(order, order, len(order))
    ctx.profile_enable(True)
    ctx.profile_reset()
    ctx.hist(want_countable=False)
    pr = ctx.profile_read()
    ctx.profile_enable(False)
    kern = {k: round(v[0] / v[1], 4) for k, v in pr.items() if v[1]}
    return sorted(ts)[len(ts) // 2], h, int(ctx.info().n_reruns), kern


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=6_000_000)
    ap.add_argument("--paths", type=int, default=16)
    args = ap.parse_args()
    n, p = args.nodes, args.paths
    items, pre, _ = orc.pansyn(3, n, p)
    order = np.arange(p, dtype=np.uint32)
    out = {"nodes": n, "paths": p, "steps": int(len(items))}
    rng = np.random.default_rng(1)
    variants = {"sorted": items.astype(np.uint32)}
    a, b = int(pre[3]), int(pre[4])
    sh = items.astype(np.uint32)
    sh[a:b] = rng.permutation(sh[a:b])
    variants["one_path_shuffled"] = sh
    ps = items.astype(np.uint32)
    m = a + (b - a) * 55 // 100
    ps[m:b] = rng.permutation(ps[m:b])
    variants["one_path_partly_shuffled"] = ps
    sw = items.astype(np.uint32)
    m = (a + b) // 2
    sw[m], sw[m + 40_000] = sw[m + 40_000], sw[m]
    variants["two_steps_swapped_across_bands"] = sw
    with capi.Context(0) as ctx:
        first_hist_ms(ctx, variants["one_path_partly_shuffled"], pre, n, order, 0)  # (untimed: every kernel of every route loaded once)
        first_hist_ms(ctx, variants["sorted"], pre, n, order, 2)
        for name, it in variants.items():
            auto_ms, h_auto, reruns, kern = first_hist_ms(ctx, it, pre, n, order, 0)
            rows_ms, h_rows, _, _ = first_hist_ms(ctx, it, pre, n, order, 2)
            out[name] = {"first_hist_ms_auto_route": round(auto_ms, 3), "first_hist_ms_rows_route": round(rows_ms, 3), "kernels_ms_auto_route": kern,
                         "same_hist": bool(np.array_equal(h_auto, h_rows)), "reruns_total": reruns}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
