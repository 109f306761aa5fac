#!/usr/bin/env python3
"""How fast is the slow route?  All paths of a pansyn graph are shuffled (no path is
tile-monotone any more, like edge-id paths), so every step goes through the global-atomicOr
scatter kernel and K1 only merges.  Prints the kernel times."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from panacus_amd import capi
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    ctx = capi.Context(0)
    ctx.set_csr_pansyn(42, n, p)
    items, off, _ = ctx.get_csr()
    rng = np.random.default_rng(0)
    items = items.copy()
    for k in range(p):
        rng.shuffle(items[off[k]:off[k + 1]])
    ctx.set_csr(items, off, n)
    order = np.arange(p, dtype=np.uint32)
    ctx.set_order(order, order, p)
    ctx.hist(want_countable=False)  # first call discovers the general paths (wasted pass + rerun)
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        _, h = ctx.hist(want_countable=False)
    dt = (time.perf_counter() - t0) / reps
    prof = ctx.profile_read()
    S = int(len(items))
    print(json.dumps({"nodes": n, "paths": p, "steps": S, "general_paths": int(ctx.info().n_general_paths),
                      "ms_per_hist": dt * 1e3, "scatter_ms": prof["scatter"][0] / reps, "cover_ms": prof["cover"][0] / reps,
                      "G_steps_per_s": S / dt / 1e9, "hist_sum": int(h.sum())}))


if __name__ == "__main__":
    main()
