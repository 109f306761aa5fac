#!/usr/bin/env python3
"""K0 (tile boundary index) on its own: ms per rebuild for a list of shapes and index settings.
One JSON line per (shape, setting)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="10000000x256,10000000x1024")
    ap.add_argument("--coarse", default="8")
    ap.add_argument("--probe", default="16")
    ap.add_argument("--tile-blocks", default="1")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args()
    from panacus_amd import capi
    for shape in args.shapes.split(","):
        n, p = (int(x) for x in shape.split("x"))
        ctx = capi.Context(0)
        ctx.config(capi.CFG_CACHE_INDEX, 0)
        ctx.set_csr_pansyn(args.seed, n, p, with_weights=False)
        order = np.arange(p, dtype=np.uint32)
        ctx.set_order(order, order, p)
        for tb, coarse, probe in ((int(t), int(c), int(w)) for t in args.tile_blocks.split(",") for w in args.probe.split(",")
                                  for c in args.coarse.split(",")):
            ctx.config(capi.CFG_TILE_BLOCKS, tb)
            ctx.config(capi.CFG_INDEX_COARSE, coarse)
            ctx.config(capi.CFG_INDEX_PROBE, probe)
            _, h0 = ctx.hist(want_countable=False)
            ctx.profile_enable(True)
            ctx.profile_reset()
            for _ in range(args.reps):
                _, h = ctx.hist(want_countable=False)
            prof = ctx.profile_read()
            ctx.profile_enable(False)
            assert int(h.sum()) == n and np.array_equal(h, h0)
            print(json.dumps({"nodes": n, "paths": p, "tile_blocks": tb, "index_coarse": coarse, "probe_ids": probe,
                              "tile_index_ms": prof["index"][0] / max(prof["index"][1], 1),
                              "tile_cover_ms": prof["cover"][0] / max(prof["cover"][1], 1),
                              "hist_sum_ok": True, "hist_head": [int(x) for x in h[:3]]}), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
