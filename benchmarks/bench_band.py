"""The one-shot route over the steps (kernels_band.hip) against the path rows on one GPU: same coverage vector and
histogram, time per cold pass.

  python benchmarks/bench_band.py [--nodes 10000000] [--paths 256] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panacus_amd import capi  # noqa: E402


def timed(ctx, steps, drop):
    ctx.hist(want_countable=False)

    def loop():
        ts = []
        for _ in range(steps):
            if drop:
                ctx.config(capi.CFG_DROP_DERIVED, 0)
            t0 = time.perf_counter()
            ctx.hist(want_countable=False)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return ts

    ts = loop()  # wall clock of a call, no HIP events around the kernels
    ctx.profile_reset()
    ctx.profile_enable(True)
    loop()
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    per = {k: round(ms / n, 4) for k, (ms, n) in prof.items() if n}
    return {"ms_median": round(ts[len(ts) // 2], 4), "ms_min": round(ts[0], 4), "kernels_ms": per}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--paths", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--groups", type=int, default=0, help="0: group = path; else paths per group")
    args = ap.parse_args()
    n, p = args.nodes, args.paths
    out = {"nodes": n, "paths": p}
    with capi.Context(0) as ctx:
        order = np.arange(p, dtype=np.uint32)
        grp = order // args.groups if args.groups else order
        G = int(grp.max()) + 1
        ctx.set_csr_pansyn(42, n, p, with_weights=False)
        ctx.set_order(order, grp.astype(np.uint32), G)
        ctx.config(capi.CFG_COVER_ROUTE, 2)
        cnt_r, h_r = ctx.hist()
        S = int(ctx.info().n_steps)
        out["steps"] = S
        out["rows_cold"] = timed(ctx, args.steps, True)
        out["rows_warm"] = timed(ctx, args.steps, False)
        ctx.config(capi.CFG_DROP_DERIVED, 0)
        ctx.config(capi.CFG_COVER_ROUTE, 1)
        cnt_b, h_b = ctx.hist()
        out["same_hist"] = bool(np.array_equal(h_r, h_b))
        out["same_coverage"] = bool(np.array_equal(cnt_r, cnt_b))
        out["reruns"] = int(ctx.info().n_reruns)
        out["band"] = timed(ctx, args.steps, True)
        B = 4 * S + 8 * (p + 1) + 4 * n + 8 * (G + 1)
        out["algorithmic_bytes"] = B
        out["band_frac_of_8TBs_call"] = round(B / (out["band"]["ms_median"] * 1e-3) / 8e12, 4)
        k = out["band"]["kernels_ms"]
        if "cover" in k:
            out["band_frac_of_8TBs_kernels"] = round(B / ((k.get("cover", 0) + k.get("index", 0) + k.get("hist", 0)) * 1e-3) / 8e12, 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
