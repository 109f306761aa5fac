#!/usr/bin/env python3
"""What the first sweep costs on paths with LARGE rearrangements -- the structural variants of real pangenomes, not the
64-step jitter of pansyn-v1r: per path one inversion / tandem duplication / translocation of a given share of its length.
The same graph sorted, and through the path rows, beside it.  The steps are edited on the host (the CPU generator), so the
graph is moderate: 4 M nodes x 32 paths by default.

  python benchmarks/bench_structural_variants.py [--nodes 4000000] [--paths 32]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as orc  # noqa: E402  (the CPU generator, and the checker of the histograms)
from panacus_amd import capi  # noqa: E402


def first_hist(ctx, items32, pre, n, order, route, reps=3):
    ts = []
    for _ in range(reps):
        ctx.config(capi.CFG_COVER_ROUTE, route)
        ctx.set_csr(items32, pre, n)
        ctx.set_order(order, order, len(order))
        ctx.config(capi.CFG_DROP_DERIVED, 0)   # (the rows route derives its rows at upload: the call is timed from the steps, as the one-shot route's)
        r0 = int(ctx.info().n_reruns)
        t0 = time.perf_counter()
        _, h = ctx.hist(want_countable=False)
        ts.append((time.perf_counter() - t0) * 1e3)
    info = ctx.info()
    return sorted(ts)[len(ts) // 2], h, int(info.n_reruns) - r0, int(info.n_spilled_last), int(info.n_loose_groups_last), int(info.n_rows), int(info.n_path_cuts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=4_000_000)
    ap.add_argument("--paths", type=int, default=32)
    ap.add_argument("--check", action="store_true", help="compare every histogram with the oracle's (slow)")
    args = ap.parse_args()
    n, p = args.nodes, args.paths
    items, pre, _ = orc.pansyn(11, n, p)
    order = np.arange(p, dtype=np.uint32)
    base = items.astype(np.uint32)
    rng = np.random.default_rng(5)

    def edit(kind, share, every=1):
        out = [None] * p
        for k in range(p):
            s = base[int(pre[k]):int(pre[k + 1])].copy()
            ln = len(s)
            w = int(ln * share)
            if k % every == 0 and w > 1:
                a = int(rng.integers(ln // 10, ln - w - ln // 10))
                if kind == "inversion":
                    s[a:a + w] = s[a:a + w][::-1].copy()
                elif kind == "duplication":        # the stretch once more right behind itself (the path grows)
                    s = np.concatenate([s[:a + w], s[a:a + w], s[a + w:]])
                elif kind == "translocation":      # the stretch cut out and put back elsewhere
                    cut = s[a:a + w].copy()
                    rest = np.concatenate([s[:a], s[a + w:]])
                    b = int(rng.integers(0, len(rest)))
                    s = np.concatenate([rest[:b], cut, rest[b:]])
            out[k] = s
        off = np.zeros(p + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(s) for s in out])
        return np.concatenate(out), off

    shapes = {"sorted": (base, pre)}
    for kind in ("inversion", "duplication", "translocation"):
        for share in (0.002, 0.01, 0.05):
            shapes[f"{kind}_{share:g}_every_path"] = edit(kind, share)
    shapes["inversion_0.05_every_4th_path"] = edit("inversion", 0.05, 4)
    res = {"nodes": n, "paths": p, "steps": int(len(base))}
    with capi.Context(0) as ctx:
        for name, (it, off) in shapes.items():
            ms, h, reruns, spilled, loose, rows, cuts = first_hist(ctx, it, off, n, order, 0)
            ms_rows, h_rows, _, _, _, _, _ = first_hist(ctx, it, off, n, order, 2)
            r = {"first_hist_ms": round(ms, 3), "rows_route_ms": round(ms_rows, 3), "reruns": reruns, "spilled": spilled, "loose_groups": loose, "path_cuts": cuts,
                 "one_shot_held": reruns == 0 and rows == 0, "same_hist_as_rows_route": bool(np.array_equal(h, h_rows))}
            if args.check:
                pi = order.astype(np.uint64)
                r["hist_equals_oracle"] = bool(np.array_equal(h, orc.hist(orc.coverage(it.astype(np.uint64), off, pi, pi, n), p)))
            res[name] = r
    print(json.dumps(res))


if __name__ == "__main__":
    main()
