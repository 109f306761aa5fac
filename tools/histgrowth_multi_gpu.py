#!/usr/bin/env python3
"""`histgrowth` of a GFA file on all GPUs of a node: one process per GPU, node-range shards,
one RCCL all-reduce of the histogram counters.

PANACUS_DIST_BACKEND selects who carries the collective:
  native  the library's own RCCL communicator (pnx_comm_init; no torch at all): the all-reduce of the
          flags + histogram follows every coverage pass on the library's stream, re-runs stay matched
          across ranks.  The 128-byte id travels through PANACUS_COMM_ID_FILE
          [default: a name private to the launch in $XDG_RUNTIME_DIR or ~/.cache/panacus_amd, removed once every rank has it]
  nccl    [default] torch.distributed over RCCL, on the verified device counters
  gloo    torch.distributed over gloo, on the fetched host counters (tests: two ranks on one GPU)

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        tools/histgrowth_multi_gpu.py -c bp -l 1,2 -q 0,0.5 [-S|-H|-g FILE] graph.gfa

Every rank parses the GFA (the parser runs at GB/s), keeps the steps whose id lies in its range,
and runs the unchanged single-GPU kernels on them; rank 0 evaluates the closed forms and prints
the same table as `panacus-amd histgrowth -a`.  With one process (no launcher) it is the
single-GPU path through the same code.
"""
from __future__ import annotations

import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("gfa")
    ap.add_argument("-c", "--count", default="node", choices=["node", "bp", "edge"])
    ap.add_argument("-l", "--coverage", default="1")
    ap.add_argument("-q", "--quorum", default="0")
    ap.add_argument("-S", "--groupby-sample", action="store_true")
    ap.add_argument("-H", "--groupby-haplotype", action="store_true")
    ap.add_argument("-g", "--groupby", default=None)
    ap.add_argument("-o", "--output", default=None, help="write the table here instead of stdout (rank 0)")
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = torch = None
    # PANACUS_DIST_BACKEND=gloo reduces the fetched host counters over gloo instead (tests: two ranks
    # on one GPU, which RCCL refuses)
    backend = os.environ.get("PANACUS_DIST_BACKEND", "nccl")
    if world > 1 and backend != "native":  # torch is only the carrier of the collective
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    from panacus_amd import capi, hostlib as hl
    from panacus_amd.distributed import plan_node_shards, shard_csr, shard_weights
    from panacus_amd.thresholds import ThresholdContainer, format_f64

    ct = {"node": hl.NODE, "bp": hl.BP, "edge": hl.EDGE}[args.count]
    mode = hl.GROUP_FILE if args.groupby else hl.GROUP_SAMPLE if args.groupby_sample else \
        hl.GROUP_HAPLOTYPE if args.groupby_haplotype else hl.GROUP_PATHID
    g = hl.GfaGraph(args.gfa, index_edges=(ct == hl.EDGE))
    items, pre = g.item_table(ct)
    n_items = g.n_edges if ct == hl.EDGE else g.n_nodes
    pi, gi, names = g.path_order(mode, args.groupby)
    G = len(names)
    weights = g.node_lens if ct == hl.BP else None

    cuts = plan_node_shards(items, n_items, world)
    lo, hi = int(cuts[rank]), int(cuts[rank + 1])
    it_r, off_r, n_r = shard_csr(items, pre, lo, hi)
    ctx = capi.Context(local_rank)
    ctx.set_csr(it_r, off_r, n_r, weights=shard_weights(weights, lo, hi))
    ctx.set_order(pi, gi, G)
    if backend == "native" and (world > 1 or os.environ.get("PANACUS_NATIVE_SINGLE") == "1"):
        from panacus_amd.distributed import native_comm_init
        native_comm_init(ctx, rank, world)  # PANACUS_COMM_ID_FILE, or a name private to this launch (distributed.default_comm_id_file)
    # A one-shot host for ARBITRARY graphs verifies the pass BEFORE it reduces: a real GFA may hold
    # paths that are not tile-monotone; the first pass then only classifies them, the library builds
    # the run index and runs the pass again inside pnx_hist_fetch -- counters reduced from the first
    # attempt would be incomplete (bench.py pipelines instead, on paths known to be monotone, and
    # fails loudly on a re-run).
    ctx.hist_async()
    _, h = ctx.hist_fetch(want_countable=False)   # settled: verified, re-run if it had to be
    if world > 1 and backend == "nccl":
        # RCCL all-reduce of the verified pass's device counters, on the library's stream
        ext = torch.cuda.ExternalStream(ctx.stream(), device=f"cuda:{local_rank}")

        class _Dev:
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}

        d_hist, _ = ctx.hist_device()
        t = torch.as_tensor(_Dev(d_hist, G + 1), device=f"cuda:{local_rank}")
        with torch.cuda.stream(ext):
            tot = t.clone()
            dist.all_reduce(tot)
            host = tot.cpu()
        ext.synchronize()
        h = host.numpy().view(np.uint64).copy()
        del t, tot, ext
    elif world > 1 and backend != "native":
        from panacus_amd.distributed import allreduce_counters
        h = allreduce_counters(h)
    # native: pnx_hist_fetch already returned the global histogram
    # every shard counts its own sentinel-free items; items in no group of any shard are bin 0
    text = None
    if rank == 0:
        tc = ThresholdContainer.parse_params(args.quorum, args.coverage)
        hl.set_quorum_offload(ctx)
        curves = hl.calc_growths(h, list(zip(tc.coverage, tc.quorum)))
        hl.set_quorum_offload(None)
        T = len(curves)
        rows = ["\t".join(["panacus", "hist"] + ["growth"] * T),
                "\t".join(["count"] + [args.count] * (T + 1)),
                "\t".join(["coverage", ""] + [c.get_string() for c in tc.coverage]),
                "\t".join(["quorum", ""] + [q.get_string() for q in tc.quorum])]
        rows.append("\t".join(["0", str(int(h[0]))] + ["NaN"] * T))
        for i in range(1, G + 1):
            rows.append("\t".join([str(i), str(int(h[i]))] + [format_f64(math.floor(c[i - 1])) for c in curves]))
        text = "\n".join(rows) + "\n"
        if args.output:
            with open(args.output, "w") as f:
                f.write(text)
        else:
            sys.stdout.write(text)
    reruns = int(ctx.info().n_reruns)
    if world > 1 and backend != "native":
        torch.cuda.synchronize()
        dist.destroy_process_group()
    ctx.close()
    if os.environ.get("PANACUS_TOOL_REPORT_RERUNS") and rank == 0:
        sys.stderr.write(f"reruns={reruns}\n")
    return text


if __name__ == "__main__":
    main()
