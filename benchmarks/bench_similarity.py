#!/usr/bin/env python3
"""Group x group intersections (the `similarity` accumulation, similarity.rs:119-150) on the
presence matrix of a pansyn-v1 graph: K5 timed with HIP events on the context's stream.

Work model (DESIGN_DEADENDS.md section 4, K5): the kernel is VALU-bound -- one AND + one popcount-accumulate per
(pair, 32-item word), pairs counted on and above the diagonal at tile granularity.
Prints one JSON line.
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
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--bp", action="store_true", help="weighted by node length (16 weight planes)")
    ap.add_argument("--check-nodes", type=int, default=200_000, help="oracle check on a small graph first (0 = skip)")
    ap.add_argument("--variant", type=int, default=1, help="PNX_CFG_PAIRS_VARIANT: 1 = int8 MFMA [default], 0 = AND + popcount")
    args = ap.parse_args()
    from panacus_amd import capi

    ctx = capi.Context(0)
    ctx.config(capi.CFG_PAIRS_VARIANT, args.variant)
    if args.check_nodes:
        import oracle as orc
        n, p = args.check_nodes, min(args.paths, 96)
        items, pre, lens = orc.pansyn(args.seed, n, p)
        ctx.set_csr(items.astype(np.uint32), pre, n, weights=lens if args.bp else None)
        pg = np.arange(p, dtype=np.uint64)
        ctx.set_order(pg, pg, p)
        r, c = orc.by_group(items, pre, pg, pg, n)
        exp, _, _ = orc.similarity(r, c, p, node_lens=lens if args.bp else None)
        assert (ctx.group_intersections() == exp).all(), "K5 differs from the oracle"

    N, P = args.nodes, args.paths
    ctx.set_csr_pansyn(args.seed, N, P, with_weights=args.bp)
    order = np.arange(P, dtype=np.uint32)
    ctx.set_order(order, order, P)
    ctx.config(capi.CFG_KEEP_PRESENCE, 1)
    ctx.hist(want_countable=False)
    inter = ctx.group_intersections()  # warm-up
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        inter = ctx.group_intersections()
    wall = (time.perf_counter() - t0) / args.reps
    ms, launches = ctx.profile_read()["pairs"]
    k_ms = ms / max(launches, 1)
    info = ctx.info()
    row_words = ((N + 1 + 2047) // 2048) * 64
    side = (P + 63) // 64
    tile_pairs = side * (side + 1) // 2
    pair_words = tile_pairs * 4096 * row_words
    planes = 16 if args.bp else 0
    valu_ops = pair_words * (1 + 3 * planes if args.bp else 2)
    # 256 CUs x 4 SIMD x 16 lanes at 2.4 GHz
    peak_ops = 256 * 4 * 16 * 2.4e9
    # matrix cores: 128 x 128 tiles on and above the diagonal, one int8 MAC per (pair, item, 7-bit digit of the weight)
    side128 = (P + 127) // 128
    digits = 1
    if args.bp:  # 7-bit digits of the widest weight (pansyn node lengths stay below 2^14: two digits)
        wmax = int(ctx.get_weights().max())
        digits = max(1, (wmax.bit_length() + 6) // 7)
    mfma_ops = 2 * (side128 * (side128 + 1) // 2) * 128 * 128 * row_words * 32 * digits
    peak_i8 = 5.0e15  # dense int8 = 2 x the 2.5 PFLOP/s bf16 peak (the guide's 32x32x32 i8 micro-benchmark floor: 4.4e15)
    out = {
        "benchmark": "group_intersections", "nodes": N, "groups": P, "weighted": bool(args.bp),
        "variant": "int8 MFMA" if args.variant == 1 else "AND + popcount", "weight_digits": digits,
        "mfma_ops": mfma_ops if args.variant == 1 else None,
        "mfma_frac_of_peak": (mfma_ops / (k_ms * 1e-3) / peak_i8) if args.variant == 1 else None,
        "steps_in_csr": int(info.n_steps), "kernel_ms": k_ms, "wall_ms_per_call": wall * 1e3,
        "pair_words_per_s": pair_words / (k_ms * 1e-3),
        "item_pairs_per_s": pair_words * 32 / (k_ms * 1e-3),
        # the AND + popcount cost model prices the vector-ALU variant only; the matrix-core variant does none of those operations
        # (a fraction above 1 -- r04 printed 1.26 and 18.1 -- is that model applied to the wrong kernel, not evidence)
        "valu_lane_ops": valu_ops if args.variant == 0 else None,
        "valu_frac_of_peak": (valu_ops / (k_ms * 1e-3) / peak_ops) if args.variant == 0 else None,
        "presence_bytes": P * row_words * 4,
        "presence_read_amplification_by_tiling": 2 * tile_pairs / side,
        "checks": {"diag_sum": int(np.diag(inter).sum()), "symmetric": bool((inter == inter.T).all())},
    }
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
