#!/usr/bin/env python3
"""bench.py -- histgrowth throughput of the MI355X hot path (BASELINE.json metric).

One "step" = one full histgrowth pass over one synthetic pangenome resident in HBM:
  tile index (K0) -> tile coverage (K1) -> histogram (K2) on the GPU, the (G+1)-bin
  histogram back to the host, then the exact closed-form growth curves (f64, host threads)
  for the configured (coverage, quorum) pairs -- i.e. `panacus histgrowth -c node -l 1,2,1
  -q 0,0,0.5` after the GFA has been turned into the CSR.  The tile index is rebuilt in every
  step (PNX_CFG_CACHE_INDEX = 0), so nothing is cached across the timed passes.

Workload at N = 1: BASELINE.json configs[2] ("histgrowth ... on 10M-node / 256-path
synthetic"), generator pansyn-v1 seed 42.  With --gpus N each rank owns one node-range
shard of the same shape (weak scaling: the global graph has N x 10M nodes, seeds 42+rank),
the per-rank histograms are summed with an RCCL all-reduce on the device counters, and
rank 0 evaluates the closed forms.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


class _DevArray:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, n, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def algorithmic_bytes_hist(S, P, N, G, weighted=False):
    """B_hist = 4*S + 8*(P+1) + 4*N [+ 4*N if bp] + 8*(G+1)."""
    return 4 * S + 8 * (P + 1) + 4 * N + (4 * N if weighted else 0) + 8 * (G + 1)


def pmc_traffic_from_profiles(nodes, paths):
    """HBM bytes per k_tile_cover launch from the committed rocprofv3 PMC summaries
    (profiles/, separate --pmc FETCH_SIZE / WRITE_SIZE passes on this same workload).
    gfx950 reports half of the bytes of wide streaming reads, hence 2 x FETCH_SIZE."""
    if (nodes, paths) != (10_000_000, 256):
        return None, None
    vals = {}
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(ROOT, "profiles", f"r01_hist_cfg3_pmc_{name}.csv")
        if not os.path.exists(path):
            return None, None
        for line in open(path):
            if "k_tile_cover" in line and f",{name}," in line:
                vals[name] = float(line.rsplit(",", 2)[1])
    if len(vals) != 2:
        return None, None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "profiles/r01_hist_cfg3_pmc_{FETCH,WRITE}_SIZE.csv (2*FETCH_SIZE + WRITE_SIZE, KiB)"


def cpu_baseline(sample_nodes, n_paths, pairs, seed=42, min_seconds=10.0, max_reps=40):
    """The oracle (a plain-C port of the reference's serial loops) on a bounded sample of the
    same workload, timed on this host: coverage + hist + closed-form growth, repeated until
    about `min_seconds` of CPU work has been measured."""
    import oracle as orc
    items, pre, _ = orc.pansyn(seed, sample_nodes, n_paths)
    pi = np.arange(n_paths, dtype=np.uint64)
    reps, total = 0, 0.0
    while reps < max_reps and (total < min_seconds or reps == 0):
        t0 = time.perf_counter()
        cov = orc.coverage(items, pre, pi, pi, sample_nodes)
        h = orc.hist(cov, n_paths)
        for c, q in pairs:
            orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))
        total += time.perf_counter() - t0
        reps += 1
    dt = total / reps
    return {
        "value": sample_nodes * n_paths / dt / 1e6,
        "unit": "M node*paths/s",
        "cores": 1,
        "kind": "port",
        "sample": f"pansyn-v1 seed {seed}, {sample_nodes} nodes x {n_paths} paths "
                  f"({len(items)} steps); serial coverage+hist+closed-form growth, "
                  f"{reps} passes, {dt:.3f} s each ({total:.1f} s of CPU work)",
    }, h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--paths", type=int, default=256)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--tile-blocks", type=int, default=1)
    ap.add_argument("--cover-variant", type=int, default=None)
    ap.add_argument("--index-coarse", type=int, default=None)
    ap.add_argument("--cover-waves", type=int, default=None)
    ap.add_argument("--cover-split", type=int, default=None)
    ap.add_argument("--cpu-sample-nodes", type=int, default=4_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--growth-threads", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=1,
                    help="contexts (streams) over the one resident graph whose passes alternate.  1 [default]: one "
                         "stream, the coverage kernel is timed alone (what `roofline` is defined on); 2: +11 %% passes/s, "
                         "but two coverage kernels then overlap and a launch takes 1.1 ms (DESIGN.md section 5)")
    ap.add_argument("--no-quorum-offload", action="store_true")
    ap.add_argument("--quorum-offload-min-n", type=int, default=512)
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: panacus_amd has no CPU fallback")
    blocking = world > 1 or os.environ.get("PANACUS_BENCH_BLOCKING") == "1"
    if blocking:
        # Several ranks share the host (and possibly one cgroup CPU quota): waits must sleep, not
        # spin, or the waiting ranks eat the CPU time rank 0 needs for the closed forms.  The flag
        # has to be set before the HIP context of the device exists.
        try:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipSetDevice(local_rank)
            hip.hipSetDeviceFlags(0x4)  # hipDeviceScheduleBlockingSync
        except OSError:
            pass
    torch.cuda.set_device(local_rank)
    dist = None
    # PANACUS_BENCH_FORCE_DIST=1 runs the multi-GPU code path (RCCL all-reduce on the device
    # counters) with a single rank, so it can be exercised on a 1-GPU box
    use_dist = world > 1 or os.environ.get("PANACUS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from panacus_amd import capi, hostlib
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold

    N, P = args.nodes, args.paths
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]  # -l 1,2,1 -q 0,0,0.5
    thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs]

    def make_context(owner=None):
        c = capi.Context(local_rank)
        c.config(capi.CFG_TILE_BLOCKS, args.tile_blocks)
        c.config(capi.CFG_CACHE_INDEX, 0)
        if blocking:
            c.config(capi.CFG_BLOCKING_SYNC, 1)
        if args.index_coarse is not None:
            c.config(capi.CFG_INDEX_COARSE, args.index_coarse)
        if args.cover_waves is not None:
            c.config(capi.CFG_COVER_WAVES, args.cover_waves)
        if args.cover_split is not None:
            c.config(capi.CFG_COVER_SPLIT, args.cover_split)
        if args.cover_variant is not None:
            c.config(capi.CFG_COVER_VARIANT, args.cover_variant)
        if owner is None:
            c.set_csr_pansyn(args.seed + rank, N, P, with_weights=False)
        else:
            c.share_csr(owner)  # the same ItemTable in HBM, no copy
        order = np.arange(P, dtype=np.uint32)
        c.set_order(order, order, P)
        return c

    class Lane:
        """One context = one stream with two passes in flight.  Two lanes over the same resident
        graph alternate their passes, so that the short latency-bound kernels of one lane's pass
        (tile index, histogram, the result copy) run beside the coverage kernel of the other's."""

        def __init__(self, ctx, group):
            self.ctx = ctx
            self.group = group
            self.hist_views = {}
            if use_dist:
                # the per-shard counters are summed with an RCCL all-reduce that is ENQUEUED behind the
                # pass on the library's own stream (a collective on another stream would have to wait
                # for free CUs until the next pass's coverage kernel -- one resident wave per tile -- ends)
                self.ext = torch.cuda.ExternalStream(ctx.stream(), device=f"cuda:{local_rank}")
                self.ring = [{"tmp": torch.zeros(P + 1, dtype=torch.int64, device=f"cuda:{local_rank}"),
                              "host": torch.zeros(P + 1, dtype=torch.int64).pin_memory(),
                              "ev": torch.cuda.Event(blocking=blocking), "reruns": 0} for _ in range(2)]
                self.enq = self.fin = 0

        def enqueue(self):
            ctx = self.ctx
            ctx.hist_async()
            if use_dist:
                d_hist = ctx.hist_enqueued()
                t = self.hist_views.get(d_hist)
                if t is None:
                    t = self.hist_views[d_hist] = torch.as_tensor(_DevArray(d_hist, P + 1), device=f"cuda:{local_rank}")
                slot = self.ring[self.enq % 2]
                self.enq += 1
                with torch.cuda.stream(self.ext):
                    slot["tmp"].copy_(t)
                    dist.all_reduce(slot["tmp"], group=self.group)  # RCCL, int64 sum == uint64 sum for counts < 2^63
                    slot["host"].copy_(slot["tmp"], non_blocking=True)
                    slot["ev"].record(self.ext)
                slot["reruns"] = int(ctx.info().n_reruns)

        def settle(self):
            """wait for the OLDEST enqueued pass of this lane; multi-GPU: its all-reduced counters"""
            ctx = self.ctx
            if use_dist:
                slot = self.ring[self.fin % 2]
                self.fin += 1
                slot["ev"].synchronize()
                ctx.hist_fetch(want_countable=False)  # verifies and retires the pass
                if int(ctx.info().n_reruns) != slot["reruns"]:
                    # A pass that fails its verification is run again by the library, and its reduced
                    # counters would be stale.  It cannot happen here (pansyn paths are tile-monotone);
                    # a host for arbitrary graphs settles the first pass before it pipelines.  Failing
                    # is better than an unmatched collective.
                    raise RuntimeError("a coverage pass was re-run inside the pipelined multi-GPU loop")
                return slot["host"].numpy().view(np.uint64).copy()
            _, h = ctx.hist_fetch(want_countable=False)
            return h

        def close(self):
            if use_dist:
                # torch objects that were used on the library's stream (pinned buffers record it when
                # they are freed) must go before the stream does
                self.ring.clear()
                self.hist_views.clear()
                del self.ext

    n_lanes = max(1, args.lanes)
    ctx = make_context()
    all_lanes = lanes = [Lane(ctx, None)]
    for _ in range(1, n_lanes):
        # every lane has its own communicator: its collectives are ordered on its own stream
        lanes.append(Lane(make_context(ctx), dist.new_group() if use_dist else None))
    info = ctx.info()
    S = int(info.n_steps)

    # large group counts: the O(n^3) inner sums of the quorum closed form run on the GPU
    # (bit-identical, see csrc/kernels_closed_form.hip); below 512 groups the host is faster
    if rank == 0 and not args.no_quorum_offload:
        hostlib.set_quorum_offload(ctx, args.quorum_offload_min_n)

    def growth_begin(h):
        """rank 0: set the closed forms up and enqueue their device part (if any) behind the pass
        that is running"""
        return hostlib.calc_growths_begin(h, thr, args.growth_threads) if rank == 0 else None

    def growth_end(pending):
        return hostlib.calc_growths_end(pending) if rank == 0 else None

    def run(n_steps, lanes=None):
        """n_steps complete histgrowth passes, dealt round-robin to the lanes.  Consecutive passes
        are independent; every lane keeps two of its own in flight: while the host evaluates the
        closed forms of pass k, later passes run and the next one of that lane is already enqueued
        behind them.  Every pass is finished inside the call."""
        h = growths = None
        lanes = all_lanes if lanes is None else lanes
        L = len(lanes)
        enqueued = 0
        for _ in range(min(n_steps, 2 * L)):
            lanes[enqueued % L].enqueue()
            enqueued += 1
        for k in range(n_steps):
            h = lanes[k % L].settle()
            pending = growth_begin(h)
            if enqueued < n_steps:
                lanes[enqueued % L].enqueue()
                enqueued += 1
            growths = growth_end(pending)
        return h, growths

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        for ln in lanes:
            ln.ctx.sync()

    if args.warmup:
        run(args.warmup)
    barrier()
    # timed region: HIP events (on the context's stream) around the dominant kernel only; the
    # other kernels are timed in a short untimed tail so that their event records do not sit
    # between the kernels of the measured passes
    for ln in lanes:
        ln.ctx.profile_enable(True)
        ln.ctx.profile_select([capi.K_COVER])
        ln.ctx.profile_reset()
    t0 = time.perf_counter()
    h, growths = run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    prof = {"cover": (0.0, 0)}
    for ln in lanes:  # the coverage kernel of every lane, as it ran beside the other lanes' short kernels
        ms, cnt = ln.ctx.profile_read()["cover"]
        prof["cover"] = (prof["cover"][0] + ms, prof["cover"][1] + cnt)
        ln.ctx.profile_select(None)
        ln.ctx.profile_reset()
        if ln.ctx is not ctx:
            ln.ctx.profile_enable(False)
    run(5, lanes[:1])  # index and histogram kernels on their own (one lane)
    barrier()
    prof_tail = ctx.profile_read()
    ctx.profile_enable(False)
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # latency of one un-pipelined pass (device + fetch + closed form), for the record
    barrier()
    l0 = time.perf_counter()
    for _ in range(3):
        run(1)
    barrier()
    latency_ms = (time.perf_counter() - l0) / 3 * 1e3

    # host-side share of a step (closed-form growth), measured separately on rank 0
    growth_ms = None
    if rank == 0:
        g0 = time.perf_counter()
        for _ in range(3):
            hostlib.calc_growths(h, thr, args.growth_threads)
        growth_ms = (time.perf_counter() - g0) / 3 * 1e3

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * N * P / (dt / args.steps) / 1e6
        cover_ms, cover_n = prof["cover"]
        index_ms, index_n = prof_tail["index"]
        hist_ms, hist_n = prof_tail["hist"]
        cover_avg_ms = cover_ms / max(cover_n, 1)
        B = algorithmic_bytes_hist(S, P, N, P)
        achieved = B / (cover_avg_ms * 1e-3) / 1e9 if cover_avg_ms > 0 else 0.0
        device_ms = cover_avg_ms + index_ms / max(index_n, 1) + hist_ms / max(hist_n, 1)
        traffic, traffic_src = pmc_traffic_from_profiles(N, P)
        out = {
            "metric": "histgrowth_throughput",
            "value": value,
            "unit": "M node*paths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"histgrowth -c node -l 1,2,1 -q 0,0,0.5 on pansyn-v1 synthetic, "
                            f"{N} nodes x {P} paths per GPU (BASELINE.json configs[2])",
                "nodes_per_gpu": N, "paths": P, "groups": P, "steps_in_csr": S, "seed": args.seed,
                "threshold_pairs": pairs, "tile_items": int(info.tile_items),
                "parallelism": "node-range shards, RCCL all-reduce of hist counters" if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_tile_cover",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": B,
                "avg_launch_ms": cover_avg_ms,
                "launches": cover_n,
            },
            "breakdown_ms": {
                "tile_index": index_ms / max(index_n, 1), "tile_cover": cover_avg_ms,
                "hist": hist_ms / max(hist_n, 1), "device_total": device_ms,
                "lanes": len(lanes),
                "host_closed_form_growth": growth_ms, "host_threads": hostlib.pool_threads(), "host_usable_cpus": hostlib.usable_cpus(),
                "quorum_inner_sums_on_gpu": bool(not args.no_quorum_offload and P >= args.quorum_offload_min_n
                                                 and hostlib.quorum_offload_usable()),
                "single_pass_latency": latency_ms,
            },
            "hbm_gbs_whole_device_pass": B / (device_ms * 1e-3) / 1e9 if device_ms > 0 else 0.0,
            "checks": {"hist_sum": int(h.sum()), "expected_hist_sum": world * N,
                       "growth_last_floor": [int(np.floor(g[-1])) for g in growths]},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                cb, _ = cpu_baseline(min(args.cpu_sample_nodes, N), P, pairs, args.seed)
                out["cpu_baseline"] = cb
            except Exception as e:  # the oracle is optional test infrastructure
                out["cpu_baseline"] = {"error": str(e)}
    for ln in lanes:
        ln.close()
    if use_dist:
        torch.cuda.synchronize()
        dist.destroy_process_group()
    hostlib.set_quorum_offload(None)
    for ln in reversed(lanes):  # borrowers of the resident graph before its owner
        ln.ctx.close()
    if rank == 0:
        # RCCL writes a version banner through C stdio; push it out before the one JSON line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
