"""Path rows against the step routes on one GPU: same coverage vector and histogram, time per pass,
time to derive the rows (the cold path of an upload).

  python benchmarks/bench_rows.py [--nodes 10000000] [--paths 256] [--steps 50]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panacus_amd import capi  # noqa: E402


def timed_passes(ctx, steps):
    ctx.hist(want_countable=False)
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.hist_async()
        ctx.hist_fetch()
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    per = {k: round(ms / n, 4) for k, (ms, n) in prof.items() if n}
    return dt * 1e3, per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--paths", type=int, default=256)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--splits", default="0,1,2,4,8")
    ap.add_argument("--layouts", default="1,2")
    args = ap.parse_args()
    n, p = args.nodes, args.paths
    out = {"nodes": n, "paths": p}
    with capi.Context(0) as ctx:
        order = np.arange(p, dtype=np.uint32)
        # reference: round 2's route over the packed steps
        ctx.config(capi.CFG_COVER_VARIANT, 2)
        ctx.set_csr_pansyn(42, n, p, with_weights=False)
        ctx.set_order(order, order, p)
        cnt2, h2 = ctx.hist()
        out["steps"] = int(ctx.info().n_steps)
        ms, per = timed_passes(ctx, args.steps)
        out["variant2"] = {"ms_per_pass": round(ms, 4), "kernels_ms": per}
        ctx.config(capi.CFG_COVER_VARIANT, 3)
        for layout in [int(x) for x in args.layouts.split(",")]:
            ctx.config(capi.CFG_ROWS_LAYOUT, layout)
            res = {}
            prep = []
            for _ in range(4):
                ctx.config(capi.CFG_DROP_DERIVED, 0)
                t0 = time.perf_counter()
                ctx.prepare()
                prep.append((time.perf_counter() - t0) * 1e3)
            res["prepare_ms"] = [round(x, 3) for x in prep]
            info = ctx.info()
            res["n_rows"] = int(info.n_rows)
            res["tile_major"] = int(info.rows_tile_major)
            ctx.set_order(order, order, p)
            cnt3, h3 = ctx.hist()
            res["same_hist"] = bool(np.array_equal(h2, h3))
            res["same_coverage"] = bool(np.array_equal(cnt2, cnt3))
            for split in [int(x) for x in args.splits.split(",")]:
                ctx.config(capi.CFG_COVER_SPLIT, split)
                ms, per = timed_passes(ctx, args.steps)
                _, h = ctx.hist(want_countable=False)
                res[f"split{split}"] = {"ms_per_pass": round(ms, 4), "kernels_ms": per, "same_hist": bool(np.array_equal(h2, h))}
            ctx.config(capi.CFG_COVER_SPLIT, 0)
            # cold: derived data dropped, one hist call
            cold = []
            for _ in range(3):
                ctx.config(capi.CFG_DROP_DERIVED, 0)
                t0 = time.perf_counter()
                ctx.hist(want_countable=False)
                cold.append((time.perf_counter() - t0) * 1e3)
            res["cold_first_pass_ms"] = [round(x, 3) for x in cold]
            out[f"rows_layout{layout}"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
