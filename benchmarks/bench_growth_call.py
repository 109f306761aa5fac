#!/usr/bin/env python3
"""Where the time of one ordered-growth call goes beyond its kernels (cfg4 shape by default): the enqueue (host tables, uploads,
mask kernel, launches), the wait, the copy of the result -- with the library's kernel timing off."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--paths", type=int, default=512)
    ap.add_argument("--orders", type=int, default=128)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch
    torch.cuda.set_device(0)
    from panacus_amd import capi
    from panacus_amd.pansyn import random_orders
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table
    N, P, R = args.nodes, args.paths, args.orders
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]
    ctx = capi.Context(0)
    ctx.set_csr_pansyn(42, N, P)
    order = np.arange(P, dtype=np.uint32)
    ctx.set_order(order, order, P)
    cov = [coverage_abs(Threshold(ABSOLUTE, c), P) for c, _ in pairs]
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), P) for _, q in pairs])
    perms = random_orders(42, R, P)
    ctx.config(capi.CFG_KEEP_PRESENCE, 1)
    ctx.hist(want_countable=False)
    ctx.ordered_growth(cov, qt, perms)
    res = {}
    for prof in (False, True):
        ctx.profile_enable(prof)
        ctx.profile_reset()
        t_enq = t_wait = t_fetch = t_whole = 0.0
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            shape = ctx.ordered_growth_async(cov, qt, perms)
            t1 = time.perf_counter()
            ctx.ordered_growth_device()
            t2 = time.perf_counter()
            out = ctx.ordered_growth_fetch(shape)
            t3 = time.perf_counter()
            t_enq += t1 - t0
            t_wait += t2 - t1
            t_fetch += t3 - t2
        for _ in range(args.reps):
            t0 = time.perf_counter()
            out = ctx.ordered_growth(cov, qt, perms)
            t_whole += time.perf_counter() - t0
        k = 1e3 / args.reps
        res["kernel_timing_on" if prof else "kernel_timing_off"] = {
            "enqueue_ms": t_enq * k, "wait_ms": t_wait * k, "fetch_ms": t_fetch * k, "one_call_ms": t_whole * k,
            "growth_kernels_ms": (ctx.profile_read()["growth"][0] / (2 * args.reps)) if prof else None}
    print(json.dumps({"shape": [N, P, R], **res, "check_last": [int(out[0, t, -1]) for t in range(3)]}))


if __name__ == "__main__":
    main()
