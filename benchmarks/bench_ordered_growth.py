#!/usr/bin/env python3
"""BASELINE.json configs[3]: ordered-histgrowth over R random group orders (permuted growth)
on a 10M-node / 512-path pansyn-v1 graph, thresholds -l 1,2,1 -q 0,0,0.5.

One process per GPU; with WORLD_SIZE > 1 the R orders are sharded over the ranks
(permutation sharding, replicated presence matrix) and the per-rank results are summed with an
RCCL all-reduce -- strong scaling (total work fixed).  Prints one JSON line on rank 0.
"""
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
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--pairs", default="1:0,2:0,1:0.5", help="coverage:quorum pairs")
    ap.add_argument("--bp", action="store_true", help="count bp (node lengths as weights) instead of nodes")
    ap.add_argument("--warm-full", action="store_true", help="warm up with all orders (profiling: every growth launch is then the same)")
    args = ap.parse_args()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from panacus_amd import capi
    from panacus_amd.distributed import split_orders
    from panacus_amd.pansyn import random_orders
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table

    N, P, R = args.nodes, args.paths, args.orders
    pairs = [(int(a.split(':')[0]), float(a.split(':')[1])) for a in args.pairs.split(',')]
    ctx = capi.Context(local_rank)
    ctx.set_csr_pansyn(args.seed, N, P, with_weights=args.bp)
    order = np.arange(P, dtype=np.uint32)
    ctx.set_order(order, order, P)
    cov = [coverage_abs(Threshold(ABSOLUTE, c), P) for c, _ in pairs]
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), P) for _, q in pairs])
    perms = random_orders(args.seed, R, P)
    mine = list(split_orders(R, world, rank))
    my_perms = perms[mine]
    ctx.config(capi.CFG_KEEP_PRESENCE, 1)
    t0 = time.perf_counter()
    ctx.hist(want_countable=False)           # builds the presence matrix (K0, K1 with WRITE_M, K2)
    t_pack = time.perf_counter() - t0
    ctx.ordered_growth(cov, qt, my_perms if args.warm_full else my_perms[:1])  # warm-up (masks, first launch)
    ctx.profile_enable(True)
    ctx.profile_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        out = ctx.ordered_growth(cov, qt, my_perms)
        if world > 1:
            full = torch.zeros((R, len(pairs), P), dtype=torch.int64, device=f"cuda:{local_rank}")
            full[mine] = torch.from_numpy(out.view(np.int64)).to(full.device)
            dist.all_reduce(full)
            torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / args.reps
    prof = ctx.profile_read()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        info = ctx.info()
        n_words = (N + 1 + 63) // 64
        b_growth = R * (8 * P * n_words + 8 * len(pairs) * P)
        gms, gn = prof["growth"]
        print(json.dumps({
            "metric": "ordered_growth_permuted", "count": "bp" if args.bp else "node",
            "workload": f"{R} random orders x {len(pairs)} threshold pairs, "
            f"{N} nodes x {P} groups (BASELINE.json configs[3])", "n_gpus": world, "scaling": "strong",
            "seconds_per_call": dt, "orders_per_s": R / dt, "M_node_group_orders_per_s": N * P * R / dt / 1e6,
            "presence_pack_s": t_pack, "growth_kernels_ms_per_call_rank0": gms / max(args.reps, 1),
            "algorithmic_bytes": b_growth, "algorithmic_GBps": b_growth / dt / 1e9,
            "steps_in_csr": int(info.n_steps),
            "check_last": [int(out[0, t, -1]) for t in range(len(pairs))],
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
