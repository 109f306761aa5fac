#!/usr/bin/env python3
"""What ONE of eight node-range shards costs on one MI355X -- every part of the multi-GPU estimate measured, none assumed
(node-range shards, one all-reduce of small counters).

For shard r of D = 8 of BASELINE.json configs[2] (histgrowth, 10 M x 256) and configs[3] (ordered-histgrowth over 128 orders,
10 M x 512), and for the whole graph on the same GPU in the same run:

  cold_hist     pnx_hist from the resident steps, everything derived dropped before the call (median of the calls after the
                first two): wall clock, kernel times (HIP events, separate calls), the route it took
  cold_pack     the same call with the presence matrix written (what a growth call needs first)
  growth        pnx_ordered_growth over all 128 orders on the shard's matrix: wall clock and kernel time
  allreduce     RCCL all-reduce (sum) of the 128 x 3 x 512 u64 result in place on the library's buffer and stream, through the
                library's own communicator -- ONE rank: the launch + kernel floor of the collective, not its wire time
                (1.5 MB over xGMI at ~150 GB/s per link adds ~10-20 us per ring step)

and the estimate that follows from them: (pack + growth of the whole graph) / (pack + growth of the shard + all-reduce).
Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cold_calls(ctx, capi, order, P, calls, want_m):
    ctx.config(capi.CFG_KEEP_PRESENCE, 1 if want_m else 0)
    ts = []
    for k in range(calls + 2):
        ctx.config(capi.CFG_DROP_DERIVED, 0)
        ctx.set_order(order, order, P)
        ctx.sync()
        t0 = time.perf_counter()
        ctx.hist(want_countable=False)
        if k >= 2:
            ts.append((time.perf_counter() - t0) * 1e3)
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(4):
        ctx.config(capi.CFG_DROP_DERIVED, 0)
        ctx.set_order(order, order, P)
        ctx.hist(want_countable=False)
    pr = ctx.profile_read()
    ctx.profile_enable(False)
    info = ctx.info()
    kern = {k: v[0] / v[1] for k, v in pr.items() if v[1]}
    route = ("one-shot over the steps (k_band_cover%s, %d workgroup(s) per band)" % (", WRITE_M" if want_m else "", int(info.band_splits))
             if int(info.n_rows) == 0 else "path rows (k_rows_build + k_rows_cover)")
    return {"ms_per_call": median(ts), "ms_min": min(ts), "kernels_ms": kern, "kernels_sum_ms": sum(kern.values()),
            "fixed_ms": median(ts) - sum(kern.values()), "route": route, "n_reruns": int(info.n_reruns)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--shard", type=int, default=3)
    ap.add_argument("--orders", type=int, default=128)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--calls", type=int, default=9)
    ap.add_argument("--no-whole", action="store_true", help="skip the whole-graph legs (the shard's numbers only)")
    ap.add_argument("--all-shards", action="store_true",
                    help="measure EVERY shard (a step of the job is the MAX over the ranks) and price the estimate on the slowest one")
    ap.add_argument("--allreduce-us", type=float, default=50.0,
                    help="what the estimate charges for the all-reduce of the result among the D ranks (microseconds; one rank's launch "
                         "floor is measured, the wire of D ranks is not: a stated latency, not a measurement)")
    args = ap.parse_args()
    from panacus_amd import capi
    from panacus_amd.distributed import even_node_range
    from panacus_amd.pansyn import random_orders
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table

    N, D, R = args.nodes, args.shards, args.orders
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]
    lo, hi = even_node_range(N, D, args.shard)
    out = {"what": f"shard {args.shard} of {D} (nodes {lo + 1} .. {hi}) of pansyn-v1 seed {args.seed}, {N} nodes, on one GPU; and the whole graph",
           "hbm_peak_GBps": 8000.0}

    def legs(P, lo_, hi_, growth):
        ctx = capi.Context(0)
        ctx.set_csr_pansyn_shard(args.seed, lo_, hi_ - lo_, P)
        order = np.arange(P, dtype=np.uint32)
        ctx.set_order(order, order, P)
        S = int(ctx.info().n_steps)
        res = {"nodes": hi_ - lo_, "paths": P, "steps": S}
        res["cold_hist"] = cold_calls(ctx, capi, order, P, args.calls, False)
        b_hist = 4 * S + 8 * (P + 1) + 4 * (hi_ - lo_) + 8 * (P + 1)
        res["cold_hist"]["frac_of_hbm_peak_on_algorithmic_bytes"] = b_hist / (res["cold_hist"]["ms_per_call"] * 1e-3) / 1e9 / 8000.0
        if growth:
            res["cold_pack"] = cold_calls(ctx, capi, order, P, args.calls, True)
            b_pack = 4 * S + 8 * P * ((hi_ - lo_ + 1 + 63) // 64)
            res["cold_pack"]["algorithmic_bytes"] = b_pack
            res["cold_pack"]["frac_of_hbm_peak_on_algorithmic_bytes"] = b_pack / (res["cold_pack"]["ms_per_call"] * 1e-3) / 1e9 / 8000.0
            cov = [coverage_abs(Threshold(ABSOLUTE, c), P) for c, _ in pairs]
            qt = np.stack([quorum_table(Threshold(RELATIVE, q), P) for _, q in pairs])
            perms = random_orders(args.seed, R, P)
            ctx.config(capi.CFG_KEEP_PRESENCE, 1)
            ctx.hist(want_countable=False)
            ctx.ordered_growth(cov, qt, perms[:1])
            ts = []
            ctx.profile_enable(True)
            ctx.profile_reset()
            for _ in range(args.calls):
                t0 = time.perf_counter()
                g = ctx.ordered_growth(cov, qt, perms)
                ts.append((time.perf_counter() - t0) * 1e3)
            gk = ctx.profile_read()["growth"]
            ctx.profile_enable(False)
            res["growth"] = {"orders": R, "ms_per_call": median(ts), "ms_min": min(ts), "kernel_ms": gk[0] / max(gk[1], 1),
                             "fixed_ms": median(ts) - gk[0] / max(gk[1], 1), "check_last": [int(g[0, t, -1]) for t in range(len(pairs))]}
            # the collective's floor: one rank, the library's communicator, in place on the growth result
            try:
                ctx.config(capi.CFG_COMM_REDUCE_HIST, 0)
                ctx.comm_init(capi.Context.comm_unique_id(), 0, 1)
                ctx.ordered_growth_async(cov, qt, perms)
                d_out = ctx.ordered_growth_enqueued()
                n_words = R * len(pairs) * P
                ctx.comm_allreduce_u64(d_out, n_words)
                ctx.sync()
                ar = []
                for _ in range(args.calls):
                    t0 = time.perf_counter()
                    ctx.comm_allreduce_u64(d_out, n_words)
                    ctx.sync()
                    ar.append((time.perf_counter() - t0) * 1e3)
                res["allreduce"] = {"bytes": 8 * n_words, "ms_per_call_one_rank": median(ar), "ms_min": min(ar)}
                ctx.comm_free()
            except capi.PnxError as e:
                res["allreduce"] = {"error": str(e)}
        ctx.close()
        return res

    out["cfg3_shard"] = legs(256, lo, hi, False)
    out["cfg4_shard"] = legs(512, lo, hi, True)
    if args.all_shards:
        # every shard, one after the other on this GPU: what the slowest rank of D would take
        per = []
        for r in range(D):
            lo_r, hi_r = even_node_range(N, D, r)
            if r == args.shard:
                s3, s4 = out["cfg3_shard"], out["cfg4_shard"]
            else:
                s3, s4 = legs(256, lo_r, hi_r, False), legs(512, lo_r, hi_r, True)
            per.append({"shard": r, "nodes": hi_r - lo_r, "cfg3_steps": s3["steps"], "cfg3_cold_hist_ms": s3["cold_hist"]["ms_per_call"],
                        "cfg3_route": s3["cold_hist"]["route"], "cfg4_steps": s4["steps"], "cfg4_cold_pack_ms": s4["cold_pack"]["ms_per_call"],
                        "cfg4_pack_frac_of_hbm_peak": s4["cold_pack"]["frac_of_hbm_peak_on_algorithmic_bytes"],
                        "cfg4_growth_ms": s4["growth"]["ms_per_call"], "cfg4_growth_kernel_ms": s4["growth"]["kernel_ms"],
                        "allreduce_one_rank_ms": s4.get("allreduce", {}).get("ms_per_call_one_rank")})
        out["all_shards"] = per
    if not args.no_whole:
        out["cfg3_whole"] = legs(256, 0, N, False)
        out["cfg4_whole"] = legs(512, 0, N, True)
        s, w = out["cfg4_shard"], out["cfg4_whole"]
        ar = s.get("allreduce", {}).get("ms_per_call_one_rank", 0.0) or 0.0
        one = w["cold_pack"]["ms_per_call"] + w["growth"]["ms_per_call"]
        eight = s["cold_pack"]["ms_per_call"] + s["growth"]["ms_per_call"] + ar
        if args.all_shards:
            # the job's call is the slowest rank's (pack + growth) + the collective among D ranks at a STATED latency
            worst = max(x["cfg4_cold_pack_ms"] + x["cfg4_growth_ms"] for x in out["all_shards"])
            w3 = out["cfg3_whole"]
            worst3 = max(x["cfg3_cold_hist_ms"] for x in out["all_shards"])
            out["estimate_all_shards"] = {
                "permuted_growth_one_gpu_ms": one, "slowest_shard_ms": worst, "allreduce_charged_ms": args.allreduce_us * 1e-3,
                "permuted_growth_speedup": one / (worst + args.allreduce_us * 1e-3),
                "hist_one_gpu_ms": w3["cold_hist"]["ms_per_call"], "hist_slowest_shard_ms": worst3,
                "hist_speedup": w3["cold_hist"]["ms_per_call"] / (worst3 + args.allreduce_us * 1e-3),
                "from": f"every one of the {D} shards measured in this run, one after the other on ONE GPU; the job's time is the SLOWEST shard's "
                        f"(cold pack + growth call) + {args.allreduce_us:.0f} us charged for the all-reduce among {D} ranks (a stated latency: RCCL "
                        f"has not run on two devices here); speedup = the whole graph on one GPU / that",
            }
        out["estimate_8_gpus_permuted_growth"] = {
            "one_gpu_ms": one, "eight_gpus_ms": eight, "speedup": one / eight,
            "from": "cold pack + growth call of the whole graph on one GPU / (cold pack + growth call of ONE shard of 8 + the all-reduce's "
                    "one-rank floor); every term measured in this run, wall clock of the calls (launches and the host's share included)",
            "growth_only_speedup": w["growth"]["ms_per_call"] / (s["growth"]["ms_per_call"] + ar),
        }
        s3, w3 = out["cfg3_shard"], out["cfg3_whole"]
        out["estimate_8_gpus_hist"] = {"one_gpu_ms": w3["cold_hist"]["ms_per_call"], "shard_ms": s3["cold_hist"]["ms_per_call"],
                                       "speedup_before_allreduce": w3["cold_hist"]["ms_per_call"] / s3["cold_hist"]["ms_per_call"]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
