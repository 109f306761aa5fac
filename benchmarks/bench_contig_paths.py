#!/usr/bin/env python3
"""hist on a "contig-shaped" graph: many paths, each covering one contiguous stretch of the id
space (like the haplotype contigs of an assembly-based pangenome), grouped into few samples.
Most (path, tile) pairs are empty, which stresses the per-entry overhead of the coverage kernel
rather than its streaming rate.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=4_000_000)
    ap.add_argument("--paths", type=int, default=4000)
    ap.add_argument("--groups", type=int, default=80)
    ap.add_argument("--span", type=float, default=0.05, help="fraction of the id space one path covers")
    ap.add_argument("--density", type=float, default=0.38)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--check", action="store_true", help="compare with the oracle (slow)")
    args = ap.parse_args()
    from panacus_amd import capi
    rng = np.random.default_rng(3)
    N, P, G = args.nodes, args.paths, args.groups
    span = max(1, int(N * args.span))
    chunks, off = [], [0]
    for p in range(P):
        a = int(rng.integers(1, N - span + 2))
        ids = a + np.flatnonzero(rng.random(span) < args.density).astype(np.uint32)
        if p % 7 == 3:
            ids = ids[::-1]
        chunks.append(ids.astype(np.uint32))
        off.append(off[-1] + len(ids))
    items = np.concatenate(chunks)
    pre = np.array(off, dtype=np.uint64)
    order = np.argsort(rng.integers(0, G, size=P), kind="stable").astype(np.uint32)  # paths grouped into samples
    grp = np.sort(rng.integers(0, G, size=P)).astype(np.uint32)
    grp = np.unique(grp, return_inverse=True)[1].astype(np.uint32)
    n_groups = int(grp.max()) + 1
    ctx = capi.Context(0)
    ctx.config(capi.CFG_CACHE_INDEX, 0)  # a command line run builds the index once per pass: time it
    ctx.set_csr(items, pre, N)
    ctx.set_order(order, grp, n_groups)
    cnt, h = ctx.hist()
    if args.check:
        import oracle as orc
        cov = orc.coverage(items.astype(np.uint64), pre, order.astype(np.uint64), grp.astype(np.uint64), N)
        assert np.array_equal(np.asarray(cov, dtype=np.uint32)[1:], cnt[1:]), "differs from the oracle"
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        ctx.hist(want_countable=False)
    wall = (time.perf_counter() - t0) / args.reps
    prof = ctx.profile_read()
    info = ctx.info()
    S = int(info.n_steps)
    # the COLD call: everything derived dropped before it (what a command line run pays once per count type)
    cold = {}
    for route, name in ((0, "default_route"), (2, "rows_route"), (1, "one_shot_route")):
        ctx.config(capi.CFG_COVER_ROUTE, route)
        ts = []
        for k in range(5):
            ctx.config(capi.CFG_DROP_DERIVED, 0)
            ctx.set_order(order, grp, n_groups)
            ctx.sync()
            t1 = time.perf_counter()
            _, hc = ctx.hist(want_countable=False)
            if k >= 2:
                ts.append((time.perf_counter() - t1) * 1e3)
            assert np.array_equal(hc, h), (name, k, int(ctx.info().n_rows), int(ctx.info().n_reruns), int((hc != h).sum()), hc[:6].tolist(), h[:6].tolist())
        ctx.profile_reset()
        ctx.config(capi.CFG_DROP_DERIVED, 0)
        ctx.set_order(order, grp, n_groups)
        ctx.hist(want_countable=False)
        ci = ctx.info()
        cold[name] = {"ms_per_call": sorted(ts)[len(ts) // 2], "kernels_ms": {k: v[0] / max(v[1], 1) for k, v in ctx.profile_read().items() if v[1]},
                      "route": "one-shot" if int(ci.n_rows) == 0 else "rows", "n_reruns": int(ci.n_reruns), "band_splits": int(ci.band_splits),
                      "frac_of_hbm_peak_on_4S_bytes": 4.0 * S / (sorted(ts)[len(ts) // 2] * 1e-3) / 8e12}
    ctx.config(capi.CFG_COVER_ROUTE, 0)
    out = {"benchmark": "contig_paths", "nodes": N, "paths": P, "groups": n_groups, "steps": S,
           "tiles": int(info.n_tiles), "general_paths": int(info.n_general_paths),
           "nonempty_path_tile_fraction": float(args.span + 1.0 / max(int(info.n_tiles), 1)),
           "ms_per_hist": wall * 1e3,
           "kernels_ms": {k: v[0] / max(v[1], 1) for k, v in prof.items() if v[1]},
           "ideal_stream_ms_at_5.6TBs": 4.0 * S / 5.6e12 * 1e3, "hist_sum_ok": int(h.sum()) == N, "cold": cold}
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
