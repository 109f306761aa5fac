#!/usr/bin/env python3
"""bench.py -- histgrowth throughput of the MI355X hot path (BASELINE.json metric).

One "step" = ONE COMPLETE `panacus histgrowth -c node -l 1,2,1 -q 0,0,0.5` call on a synthetic pangenome whose u32
ItemTable is resident in HBM -- and nothing else: before every step everything that an earlier step derived is dropped
(PNX_CFG_DROP_DERIVED: path rows, boundary index; PNX_CFG_DROP_GROWTH_TABLES: the (n, thresholds) tables of the closed
forms), so a step is what the reference's one sweep per abacus is (graph_broker.rs:389-432, abacus.rs:539-586):

  band edges of the ordered paths (k_band_index) -> the steps, read ONCE, to coverage vector + histogram (k_band_cover,
  csrc/kernels_band.hip) -> the (G+1) counters published to the host (k_hist_publish) -> the exact closed-form growth
  curves (f64, bit-identical to glibc: csrc/kernels_closed_form.hip) from the pass's device counters, their tables derived
  inside the step on a side stream -> curves on the host.

`value` = N x P / ms_per_step; `roofline` = SURVEY 8(d)'s algorithmic bytes / the HIP-event time of the kernel that reads
the steps / 8 TB/s (every 4th launch of the timed region is timed; events around every launch would put gaps between the
kernels); `roofline.traffic` = that kernel's HBM bytes from rocprofv3 counter passes driven by this run.  The pipelined
pass over RESIDENT path rows that earlier rounds reported as the headline is in `resident_pass` (what a caller that sweeps
the same graph again gets from the second sweep on), the derivation of those rows in `rows_route`.

`python bench.py --gpus N` launches its N ranks itself (one process per GPU, rendezvous on 127.0.0.1) unless
RANK / WORLD_SIZE are already in the environment (torch.distributed.run, the driver's way).

Workload at N = 1: BASELINE.json configs[2] ("histgrowth ... on 10M-node / 256-path synthetic"), generator pansyn-v1
seed 42.  With --gpus N the headline `value` is WEAK scaling (the contract's per-GPU work fixed: every rank owns a graph
of the same shape, seeds 42+rank, i.e. one node-range shard of an N x 10M-node graph); the per-rank histograms are summed
with an RCCL all-reduce on the device counters, and rank 0 evaluates the closed forms.

Besides the headline the same JSON line carries:
  * "strong_scaling" (N > 1) -- the SAME 10M-node / 256-path graph split into N node ranges (SURVEY 8e's primary
    partitioning): rank r holds the steps whose id falls into its range, the same one-shot step, the (G+1) counters
    all-reduced; rank 0 then runs alone on the whole graph, so `speedup_vs_1` comes from one run on one box.
  * "permuted_growth" -- BASELINE.json configs[3]: ordered-histgrowth over R = 128 random group orders on a
    10M-node / 512-path graph, STRONG scaling by node ranges as well: rank r holds the nodes of its range -- 1 / N of
    the steps, of the presence matrix and of every order's work --, evaluates ALL R orders on them, and the
    out[R][T][G] counters of the ranks are summed with one RCCL all-reduce in place on the library's device buffer
    and stream.  For N > 1 rank 0 also times the whole graph alone, so that `speedup_vs_1` comes from one run on one box.
  * "strayed_paths" (N = 1 only) -- the headline step on pansyn-v1r: paths that are NOT sorted by id.
  * "shape_10Mx1k" (N = 1 only) -- north_star's 10M-node / 1k-path histgrowth shape, the same step.
  * "cpu_baseline" (N = 1 only) -- the oracle (serial port of the reference's loops; closed forms one thread per
    threshold pair like hist.rs:68-81) on the FULL headline workload, its results compared bit for bit.

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


def moved_bytes_rows_pass(rows_in_order, N, G):
    """What a coverage pass over RESIDENT path rows moves through HBM: the 256-byte rows of the ordered paths on the tiles
    they span, the coverage vector, the counters (the steps themselves are not touched by such a pass)."""
    return 256 * rows_in_order + 4 * (N + 1) + 8 * (G + 1)


ROOFLINE_NOTE = ("achieved = SURVEY 8(d)'s algorithmic bytes of one histgrowth pass (4 B per path step + 4 B per item + the offsets and "
                 "counters) / the HIP-event duration of k_band_cover, the kernel that reads the steps, measured on its stream for every "
                 "4th launch of the timed region; frac = achieved / 8 TB/s.  traffic = HBM bytes of the same kernel from rocprofv3 PMC "
                 "passes driven by this run (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; gfx950 reports half of wide streaming reads).  "
                 "whole_step prices the same bytes on ms_per_step: band-edge index, publish, closed forms (tables derived inside the "
                 "step), launches and the host's share included")


def pmc_leg(argv_child, kernels, counters_sets, timeout=240):
    """HBM / SQ counters of the named kernels, measured by THIS run: one child `bench.py` per counter set under
    `rocprofv3 --kernel-trace --pmc ...` (counter passes only -- no trace domains mixed in), per-launch averages read
    from the rocpd database.  Returns ({kernel: {counter: avg}}, note); ({}, why) when rocprofv3 is unusable."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found"
    out = {}
    tmp_root = tempfile.mkdtemp(prefix="pnx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", PANACUS_BENCH_CHILD="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        for i, cs in enumerate(counters_sets):
            d = os.path.join(tmp_root, f"p{i}")
            cmd = [exe, "--kernel-trace", "--pmc"] + cs + ["-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__)] + argv_child
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return out, f"rocprofv3 pass {cs} failed (rc {r.returncode}): {r.stderr.decode(errors='replace')[-300:]}"
            db = sqlite3.connect(dbs[0])
            agg = {}
            for kname, cname, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
                for want in kernels:
                    if want in kname:
                        a = agg.setdefault((want, cname), [0, 0.0])
                        a[0] += 1
                        a[1] += val
            db.close()
            for (want, cname), (n, tot) in agg.items():
                out.setdefault(want, {})[cname] = tot / n
                out[want]["launches_" + cname] = n
    except Exception as e:  # profiling is evidence, not the measurement
        return out, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp_root, ignore_errors=True)
    return out, "measured by this run: rocprofv3 --kernel-trace --pmc passes over child runs of this script on the same workload"



def launch_ranks(n, force_dist):
    """`bench.py --gpus N` without a launcher: start the N ranks (one process per GPU), hand them RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*, pass rank 0's stdout through.  Fails loudly when fewer than N devices are visible."""
    import ctypes
    import socket
    import subprocess
    have = -1
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        cnt = ctypes.c_int(0)
        if hip.hipGetDeviceCount(ctypes.byref(cnt)) == 0:
            have = cnt.value
    except OSError:
        pass
    if have < 0:
        import torch
        have = torch.cuda.device_count()
    # PANACUS_BENCH_ONE_DEVICE=1 (tests): the N ranks share device 0 and reduce their counters over gloo -- every line of the
    # multi-rank path (node-range shards, the reduced flags, the collectives' call pattern) runs on a box with ONE GPU; the
    # numbers of such a run say nothing about scaling and the line says so ("one_device": true)
    one_device = os.environ.get("PANACUS_BENCH_ONE_DEVICE") == "1"
    if have < n and not (one_device and have >= 1):
        raise SystemExit(f"bench.py --gpus {n}: only {have} HIP device(s) visible to this process -- nothing was measured")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if one_device else str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PANACUS_BENCH_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if force_dist:
            env["PANACUS_BENCH_FORCE_DIST"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    raise SystemExit(rc)



def cpu_baseline(ctx, n_nodes, n_paths, pairs, seed=42, passes=3, sample_nodes=None):
    """The oracle (a plain-C port of the reference's loops, u64 items like the reference) on the
    headline workload, timed on this host: the serial coverage loop (abacus.rs:719-744 is serial in
    the reference too, despite its comment) + construct_hist + the closed-form growth curves, one
    thread per (coverage, quorum) pair as the reference's rayon par_iter does (hist.rs:68-81).
    The input is the graph that is resident on the GPU, read back once (generating 10^9 steps with
    the serial CPU generator would take minutes); `sample_nodes` < n_nodes (hosts short of memory: the u64 steps of the headline graph are
    7.8 GB) times a smaller pansyn graph instead and says so."""
    import threading

    import oracle as orc
    n = n_nodes
    if sample_nodes is not None and sample_nodes < n_nodes:
        n = sample_nodes
        items, pre, _ = orc.pansyn(seed, n, n_paths)  # the CPU generator (bit-identical to the device one)
    else:
        items32, pre, _ = ctx.get_csr()
        items = items32.astype(np.uint64)  # the reference's ItemIdSize
        del items32
    pi = np.arange(n_paths, dtype=np.uint64)
    growths = [None] * len(pairs)

    def one_pair(k):
        c, q = pairs[k]
        growths[k] = orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q))

    total = 0.0
    t_cov = t_growth = 0.0
    for _ in range(passes):
        t0 = time.perf_counter()
        cov = orc.coverage(items, pre, pi, pi, n)
        h = orc.hist(cov, n_paths)
        t1 = time.perf_counter()
        th = [threading.Thread(target=one_pair, args=(k,)) for k in range(len(pairs))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        t2 = time.perf_counter()
        total += t2 - t0
        t_cov += t1 - t0
        t_growth += t2 - t1
    dt = total / passes
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        model = "unknown"
    return {
        "value": n * n_paths / dt / 1e6,
        "unit": "M node*paths/s",
        "cores": len(pairs),
        "kind": "port",
        "sample": f"pansyn-v1 seed {seed}, {n} nodes x {n_paths} paths ({len(items)} steps"
                  f"{'' if n == n_nodes else ', a SMALLER graph than the headline workload'}); {passes} passes, {dt:.3f} s each: "
                  f"serial coverage+hist {t_cov / passes:.3f} s (1 thread, as the reference), closed-form growth "
                  f"{t_growth / passes:.3f} s ({len(pairs)} threads, one per threshold pair as hist.rs:68-81)",
        "host": {"nproc": os.cpu_count(), "cpu_model": model},
    }, h, growths



def permuted_growth_block(args, torch, dist, use_dist, world, rank, local_rank, blocking):
    """BASELINE.json configs[3], strong scaling by NODE-RANGE sharding: rank r holds the nodes of its range --
    1 / N of the steps, of the presence matrix and of every order's work -- evaluates ALL R orders on them, and the
    out[R][T][G] counters of the ranks are summed with one RCCL all-reduce on the device buffer.  Both halves of a call
    scale: the presence pack (one read of the rank's steps) and the growth kernels."""
    from panacus_amd import capi
    from panacus_amd.distributed import even_node_range
    from panacus_amd.pansyn import random_orders
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table

    N, P, R, reps = args.pg_nodes, args.pg_paths, args.pg_orders, max(1, args.pg_reps)
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]
    T = len(pairs)
    dev = f"cuda:{local_rank}"
    order = np.arange(P, dtype=np.uint32)
    cov = [coverage_abs(Threshold(ABSOLUTE, c), P) for c, _ in pairs]
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), P) for _, q in pairs])
    perms = random_orders(args.seed, R, P)

    def make(lo, hi):
        c = capi.Context(local_rank)
        if blocking:
            c.config(capi.CFG_BLOCKING_SYNC, 1)
        c.config(capi.CFG_KEEP_PRESENCE, 1)
        c.set_csr_pansyn_shard(args.seed, lo, hi - lo, P, with_weights=False)
        c.set_order(order, order, P)
        return c

    def sync_all(c):
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        c.sync()

    def pack_time(c, solo=False):
        """the presence matrix of the groups from the resident steps: ONE read of them (the one-shot route writes the matrix
        beside the coverage vector; a shard too small for it derives the path rows first), then resident"""
        c.hist(want_countable=False)   # first call: allocations, code objects
        c.profile_enable(True)
        packs = []
        for _ in range(3):
            c.config(capi.CFG_DROP_DERIVED, 0)
            c.set_order(order, order, P)
            c.profile_reset()
            if solo:  # (rank 0 alone on the whole graph: no barrier)
                torch.cuda.synchronize()
                c.sync()
            else:
                sync_all(c)
            t0 = time.perf_counter()
            c.hist(want_countable=False)
            packs.append(time.perf_counter() - t0)
        pk = c.profile_read()
        info = c.info()
        route = "one-shot over the steps (k_band_cover, WRITE_M)" if int(info.n_rows) == 0 else "path rows (k_rows_build + k_rows_cover)"
        return sorted(packs)[1], pk, info, route

    lo, hi = even_node_range(N, world, rank)
    ctx = make(lo, hi)
    pack_s, pk, info, pack_route = pack_time(ctx)
    steps_mine = int(info.n_steps)

    # ---- the sharded call: all R orders on my nodes -> out[R][T][G] -> RCCL all-reduce on the library's stream and buffer ----
    # warm-up with ALL orders: masks, first launch, and the buffers of a call of this size (warmed with one order the five timed
    # calls shared 8 ms of first-call allocations: 13.5 ms per call against 11.7 in steady state, K4 itself 11.5)
    ctx.ordered_growth(cov, qt, perms)
    ctx.profile_reset()
    ar_ms = 0.0
    if not use_dist:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            full_host = ctx.ordered_growth(cov, qt, perms)
        dt = (time.perf_counter() - t0) / reps
        gk = ctx.profile_read()["growth"]
        gk_ms = gk[0] / max(gk[1], 1)
    else:
        ext = torch.cuda.ExternalStream(ctx.stream(), device=dev)
        host = torch.zeros((R, T, P), dtype=torch.int64).pin_memory()
        ev_a = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev_b = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        done = torch.cuda.Event(blocking=blocking)
        views = {}
        native = args.collective == "native"
        if native:
            ctx.config(capi.CFG_COMM_REDUCE_HIST, 0)  # (the pack's histogram is not asked for here)
            uid = [capi.Context.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(uid[0], rank, world)

        def call(k):
            ctx.ordered_growth_async(cov, qt, perms)
            d_out = ctx.ordered_growth_enqueued()
            t = views.get(d_out)
            if t is None:
                t = views[d_out] = torch.as_tensor(_DevArray(d_out, R * T * P), device=dev).view(R, T, P)
            with torch.cuda.stream(ext):
                ev_a[k].record(ext)
                if native:
                    ctx.comm_allreduce_u64(t.data_ptr(), t.numel())  # the library's communicator, same stream
                else:
                    dist.all_reduce(t)  # RCCL, in place on the library's buffer; int64 sum == u64 sum (counts < 2^63)
                ev_b[k].record(ext)
                if rank == 0:
                    host.copy_(t, non_blocking=True)
                done.record(ext)
            done.synchronize()

        call(reps)  # warm-up (communicator, first launches)
        sync_all(ctx)
        ctx.profile_reset()
        t0 = time.perf_counter()
        for k in range(reps):
            call(k)
        sync_all(ctx)
        dt = (time.perf_counter() - t0) / reps
        gk = ctx.profile_read()["growth"]
        ar_ms = sum(ev_a[k].elapsed_time(ev_b[k]) for k in range(reps)) / reps
        red = torch.tensor([dt, gk[0] / max(gk[1], 1), ar_ms, pack_s], dtype=torch.float64, device=dev)
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        dt, gk_ms, ar_ms, pack_s = (float(x) for x in red.tolist())
        full_host = host.numpy().view(np.uint64).copy() if rank == 0 else None
        if native:
            ctx.comm_free()
        views.clear()
        del ext, host
    # ---- the same resident presence matrix through `similarity`'s group x group intersections (int8 MFMA):
    # rank 0's shard only, a "next"-row figure recorded with the driver's run; checked against the histogram-free identities
    sim = None
    if rank == 0:
        inter = ctx.group_intersections()  # warm-up: partial-sum buffers
        ctx.profile_reset()
        for _ in range(3):
            inter = ctx.group_intersections()
        sim_ms, sim_n = ctx.profile_read()["pairs"]
        sim_ms /= max(sim_n, 1)
        row_words = ((hi - lo + 1 + 2047) // 2048) * 64
        side = (P + 127) // 128
        ops = 2 * (side * (side + 1) // 2) * 128 * 128 * row_words * 32
        sim = {"kernel_ms": sim_ms, "int8_mfma_ops": ops, "nodes": hi - lo,
               "mfma_frac_of_5_POPs_peak": ops / (sim_ms * 1e-3) / 5.0e15 if sim_ms > 0 else None,
               "checks": {"symmetric": bool((inter == inter.T).all()), "diagonal_sum": int(np.diag(inter).sum())}}
    ctx.profile_enable(False)
    if use_dist:
        torch.cuda.synchronize()
    ctx.close()

    # ---- one GPU alone on the WHOLE graph (rank 0, same run, same box): what the sharded call is compared with ----
    out = None
    if rank == 0:
        if world == 1:
            t1, single, pack1_s, k1_ms = dt, full_host, pack_s, gk_ms
        else:
            c1 = make(0, N)
            pack1_s, _, _, _ = pack_time(c1, solo=True)
            c1.ordered_growth(cov, qt, perms[:1])
            c1.profile_enable(True)
            c1.profile_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                single = c1.ordered_growth(cov, qt, perms)
            t1 = (time.perf_counter() - t0) / reps
            k1 = c1.profile_read()["growth"]
            k1_ms = k1[0] / max(k1[1], 1)
            c1.profile_enable(False)
            c1.close()
        if not np.array_equal(full_host, single):
            raise SystemExit("permuted growth: the sum over the node-range shards differs from the single-GPU result")
        n_words = (N + 1 + 63) // 64
        b_growth = R * (8 * P * n_words + 8 * T * P)             # SURVEY 8(d), all R orders, whole graph
        b_pack_mine = 4 * steps_mine + 8 * P * ((hi - lo + 1 + 63) // 64)  # SURVEY 8(d), this rank's shard
        cover_ms = pk["cover"][0] / max(pk["cover"][1], 1)
        out = {
            "workload": f"ordered-histgrowth -c node -l 1,2,1 -q 0,0,0.5 over {R} random group orders (pansyn stream 7, seed "
                        f"{args.seed}), {N} nodes x {P} paths (BASELINE.json configs[3])",
            "n_gpus": world, "scaling": "strong",
            "sharding": "node ranges: rank r holds the nodes of its range (1/N of the steps and of the presence matrix) and evaluates "
                        "ALL orders on them; RCCL all-reduce (sum) of out[R][T][G] in place on the library's device buffer and stream",
            "orders": R, "threshold_pairs": pairs, "nodes_per_rank_max": (N + world - 1) // world, "reps": reps,
            "seconds_per_call": dt, "orders_per_s": R / dt,
            "M_node_group_orders_per_s": N * P * R / dt / 1e6,
            "seconds_per_call_1gpu": t1, "speedup_vs_1": t1 / dt,
            "growth_kernel_ms_rank_max": gk_ms, "growth_kernel_ms_1gpu": k1_ms,
            "allreduce_ms": ar_ms,
            "collective_path": ("none (one rank)" if not use_dist else "rccl through the library's own communicator (pnx_comm_allreduce_u64) on pnx_stream()"
                                if args.collective == "native" else "rccl via torch.distributed (nccl backend) on pnx_stream()"),
            "presence_pack_ms": pack_s * 1e3, "presence_pack_ms_1gpu": pack1_s * 1e3, "presence_pack_route": pack_route,
            "presence_pack_kernels_ms": {k: v[0] / max(v[1], 1) for k, v in pk.items() if v[1]},
            "presence_pack_cover_kernel_ms": cover_ms,
            "presence_pack_algorithmic_bytes_rank0": b_pack_mine,
            "presence_pack_frac_of_hbm_peak": b_pack_mine / pack_s / 1e9 / HBM_PEAK_GBS if pack_s > 0 else None,
            # a COLD call: the presence pack from the resident steps + the growth call; both are sharded
            "seconds_per_call_incl_pack": dt + pack_s,
            "speedup_vs_1_incl_pack": (t1 + pack1_s) / (dt + pack_s),
            "incl_pack_note": "incl_pack = the presence pack from the rank's resident steps (one read) + the growth call; with node-range "
                              "shards neither is replicated",
            "algorithmic_bytes": b_growth, "algorithmic_GBps": b_growth / dt / 1e9,
            "steps_in_csr_rank0": steps_mine,
            "similarity_intersections": sim,
            "checks": {"sharded_equals_single_gpu": True,
                       "growth_last": [int(full_host[0, t, -1]) for t in range(T)]},
        }
    return out


def strong_hist_block(args, torch, dist, use_dist, world, rank, local_rank, blocking, growth_on_device):
    """The headline workload STRONGLY scaled (node-range sharding): ONE pansyn graph of --nodes x --paths, rank r
    holds the nodes of its range -- the steps whose id falls into it, 1 / N of the ItemTable -- and runs the same one-shot step
    on them; the (G+1) counters of the ranks are summed by an RCCL all-reduce in place on the device counters and rank 0
    evaluates the closed forms.  For N > 1 rank 0 afterwards runs the same steps alone on the whole graph, so that
    `speedup_vs_1` comes from one run on one box."""
    from panacus_amd import capi, hostlib
    from panacus_amd.distributed import even_node_range
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold
    N, P, steps = args.nodes, args.paths, max(4, args.strong_steps)
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]
    thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs]
    order = np.arange(P, dtype=np.uint32)

    def run(lo, hi, dist_on, label):
        ctx = capi.Context(local_rank)
        if blocking:
            ctx.config(capi.CFG_BLOCKING_SYNC, 1)
        if args.cover_route is not None:
            ctx.config(capi.CFG_COVER_ROUTE, args.cover_route)
        ctx.set_csr_pansyn_shard(args.seed, lo, hi - lo, P, with_weights=False)
        ctx.set_order(order, order, P)
        if rank == 0:
            hostlib.set_quorum_offload(ctx, args.quorum_offload_min_n)
        stepper = OneShot(ctx, P, thr, rank=rank, world=world if dist_on else 1, use_dist=dist_on, dist=dist, torch=torch,
                          local_rank=local_rank, collective=args.collective, blocking=blocking,
                          growth_on_device=growth_on_device and rank == 0, growth_threads=args.growth_threads)

        def barrier():
            if dist_on:
                dist.barrier()
            torch.cuda.synchronize()
            ctx.sync()

        dt, h, growths, prof = timed_steps(stepper, steps, 10, barrier, max(1, min(4, steps // 4)))
        if dist_on:
            tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        info = ctx.info()
        per = {k: (v[0] / v[1] if v[1] else None) for k, v in prof.items()}
        res = {"label": label, "nodes": hi - lo, "steps_in_csr": int(info.n_steps), "ms_per_step": dt / steps * 1e3,
               "kernels_ms": {"band_index": per.get("index"), "band_cover": per.get("cover"), "tail_or_hist": per.get("hist")},
               "route": ("one-shot over the steps, %d workgroup(s) per band" % int(info.band_splits)) if int(info.n_rows) == 0 else "path rows",
               "n_reruns": int(info.n_reruns), "hist_sum": int(h.sum()),
               "growth_last_floor": [int(np.floor(g[-1])) for g in growths] if growths is not None else None}
        stepper.close()
        if rank == 0:
            hostlib.set_quorum_offload(None)
        if dist_on:
            torch.cuda.synchronize()
        ctx.close()
        return res

    lo, hi = even_node_range(N, world, rank)
    sharded = run(lo, hi, use_dist, f"rank {rank} of {world}")
    out = None
    if rank == 0:
        alone = sharded if world == 1 else run(0, N, False, "rank 0 alone on the whole graph")
        if sharded["hist_sum"] != N or alone["hist_sum"] != N or sharded["growth_last_floor"] != alone["growth_last_floor"]:
            raise SystemExit("strong scaling block: the sharded histogram / curves differ from the single-GPU ones")
        S = alone["steps_in_csr"]
        B = algorithmic_bytes_hist(S, P, N, P)
        out = {
            "workload": f"histgrowth -c node -l 1,2,1 -q 0,0,0.5 on ONE pansyn-v1 graph (seed {args.seed}), {N} nodes x {P} paths, split into "
                        f"{world} node range(s)",
            "n_gpus": world, "scaling": "strong", "steps": steps,
            "sharding": "node ranges (SURVEY 8e): rank r holds the steps whose id falls into its range; RCCL all-reduce (sum) of the "
                        "(G+1) counters in place on the device; closed forms on rank 0",
            "ms_per_step": sharded["ms_per_step"], "ms_per_step_1gpu": alone["ms_per_step"],
            "speedup_vs_1": alone["ms_per_step"] / sharded["ms_per_step"],
            "value": N * P / (sharded["ms_per_step"] * 1e-3) / 1e6, "unit": "M node*paths/s",
            "frac_of_aggregate_hbm_peak_on_algorithmic_bytes": B / (sharded["ms_per_step"] * 1e-3) / 1e9 / (HBM_PEAK_GBS * world),
            "rank0": sharded, "alone": None if world == 1 else alone,
            "checks": {"hist_sum": sharded["hist_sum"], "sharded_equals_single_gpu": True},
        }
    return out


def strong_pggb_block(args, torch, dist, use_dist, world, rank, local_rank, blocking):
    """A second strong-scaling case (N > 1) on a graph whose steps are NOT spread evenly over the ids: a pggb-shaped pangenome
    (`panacus-amd synth --shape pggb`: contig paths per haplotype, inversions, tandem duplications; grouped by sample as
    `histgrowth -S`), parsed from its GFA file, split into N node ranges BALANCED BY STEP COUNT (distributed.plan_node_shards:
    a step costs the rank that owns its id) -- a step of the job is the MAX over the ranks, so ranges of equal node count would
    let the densest range set the pace.  The line reports both plans' imbalance (largest shard's steps / mean) beside the
    measured step; rank 0 afterwards runs alone on the whole graph (`speedup_vs_1` from one run on one box)."""
    import shutil
    import tempfile
    from panacus_amd import capi, hostlib
    from panacus_amd.distributed import even_node_range, plan_node_shards, shard_csr
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]
    thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs]
    steps = max(4, args.strong_steps)
    # (every step that can fail on ONE rank is followed by an agreement of all ranks: a rank that raised alone would leave the
    # others inside the next collective for ever)
    def all_agree(ok, what):
        if use_dist:
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = bool(int(t.item()))
        if not ok:
            raise RuntimeError(f"strong scaling (pggb shape): {what} failed on a rank")

    box = [None, None]
    if rank == 0:
        try:
            box[0] = tempfile.mkdtemp(prefix="pnx_bench_pggb_")
            rc, msg, err = hostlib.run_cli(["synth", "--shape", "pggb", "--nodes", str(args.ss2_nodes), "--samples", str(args.ss2_samples),
                                            "--seed", str(args.seed), "-o", os.path.join(box[0], "pggb.gfa")])
            if rc != 0:
                box[1] = err
        except Exception as e:  # noqa: BLE001
            box[1] = repr(e)
    if use_dist:
        dist.broadcast_object_list(box, src=0)
    tmp = box[0]
    if box[1] is not None or tmp is None:
        if rank == 0 and tmp:
            shutil.rmtree(tmp, ignore_errors=True)
        raise RuntimeError(f"strong scaling (pggb shape): the graph could not be written: {box[1]}")
    gfa = os.path.join(tmp, "pggb.gfa")
    items = pre = pi = gi = names = None
    n = G = 0
    ok = True
    try:
        g = hostlib.GfaGraph(gfa)
        items, pre = g.item_table(hostlib.NODE)
        pi, gi, names = g.path_order(hostlib.GROUP_SAMPLE)
        n, G = g.n_nodes, len(names)
        g.close()
    except Exception:  # noqa: BLE001
        ok = False
    try:
        all_agree(ok, "reading the graph")  # (also: every rank has read the file)
    finally:
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)
    cuts = plan_node_shards(items, n, world)
    per_item = np.bincount(items.astype(np.int64), minlength=n + 1)
    cum = np.concatenate([[0], np.cumsum(per_item)])  # cum[k] = steps with id < k
    by_steps = [int(cum[int(cuts[r + 1])] - cum[int(cuts[r])]) for r in range(world)]
    even = []
    for r in range(world):
        lo, hi = even_node_range(n, world, r)
        even.append(int(cum[hi + 1] - cum[lo + 1]))
    mean = len(items) / world

    def run(lo, hi, dist_on, label):  # ids lo .. hi - 1
        ctx = None
        up_ok = True
        try:
            it, off, n_r = shard_csr(items, pre, lo, hi)
            ctx = capi.Context(local_rank)
            if blocking:
                ctx.config(capi.CFG_BLOCKING_SYNC, 1)
            ctx.set_csr(it, off, n_r)
            ctx.set_order(pi, gi, G)
        except Exception:  # noqa: BLE001
            up_ok = False
        if dist_on:
            all_agree(up_ok, "the upload of a shard")
        elif not up_ok:
            raise RuntimeError("strong scaling (pggb shape): the upload of the whole graph failed")
        stepper = OneShot(ctx, G, thr, rank=rank, world=world if dist_on else 1, use_dist=dist_on, dist=dist, torch=torch,
                          local_rank=local_rank, collective=args.collective, blocking=blocking, growth_on_device=False,
                          growth_threads=args.growth_threads)

        def barrier():
            if dist_on:
                dist.barrier()
            torch.cuda.synchronize()
            ctx.sync()

        dt, h, growths, prof = timed_steps(stepper, steps, 4, barrier, max(1, min(4, steps // 4)))
        if dist_on:
            tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        info = ctx.info()
        res = {"label": label, "nodes": hi - lo, "steps_in_csr": int(info.n_steps), "ms_per_step": dt / steps * 1e3,
               "route": "one-shot over the steps" if int(info.n_rows) == 0 else "path rows", "n_reruns": int(info.n_reruns),
               "hist": [int(x) for x in h], "growth_last_floor": [int(np.floor(gr[-1])) for gr in growths] if growths is not None else None}
        stepper.close()
        if dist_on:
            torch.cuda.synchronize()
        ctx.close()
        return res

    sharded = run(int(cuts[rank]), int(cuts[rank + 1]), use_dist, f"rank {rank} of {world}")
    out = None
    if rank == 0:
        alone = sharded if world == 1 else run(1, n + 1, False, "rank 0 alone on the whole graph")
        if sharded["hist"] != alone["hist"] or sum(alone["hist"]) != n or sharded["growth_last_floor"] != alone["growth_last_floor"]:
            raise SystemExit("strong scaling (pggb shape): the sharded histogram / curves differ from the single-GPU ones")
        out = {
            "workload": f"histgrowth -S -c node -l 1,2,1 -q 0,0,0.5 on a pggb-shaped graph (synth --shape pggb, seed {args.seed}): {n} nodes, "
                        f"{len(pre) - 1} paths in {G} groups, {len(items)} steps, split into {world} node range(s) of about equal STEP count",
            "n_gpus": world, "scaling": "strong", "steps": steps,
            "sharding": "node ranges balanced by step count (distributed.plan_node_shards); all-reduce (sum) of the (G+1) counters; closed forms on rank 0",
            "steps_per_rank": by_steps, "imbalance_by_steps": max(by_steps) / mean,
            "steps_per_rank_if_even_node_ranges": even, "imbalance_even_node_ranges": max(even) / mean,
            "ms_per_step": sharded["ms_per_step"], "ms_per_step_1gpu": alone["ms_per_step"],
            "speedup_vs_1": alone["ms_per_step"] / sharded["ms_per_step"],
            "rank0": {k: v for k, v in sharded.items() if k != "hist"},
            "alone": None if world == 1 else {k: v for k, v in alone.items() if k != "hist"},
            "checks": {"hist_sum": sum(sharded["hist"]), "sharded_equals_single_gpu": True},
        }
    return out


class OneShot:
    """One complete histgrowth call from the resident u32 ItemTable, nothing derived kept from call to call."""

    def __init__(self, ctx, P, thr, *, rank=0, world=1, use_dist=False, dist=None, torch=None, local_rank=0, collective="torch",
                 blocking=False, growth_on_device=False, growth_threads=0):
        from panacus_amd import capi, hostlib
        self.capi, self.hostlib = capi, hostlib
        self.ctx, self.P, self.thr, self.rank = ctx, P, thr, rank
        self.use_dist, self.dist, self.torch = use_dist, dist, torch
        self.growth_on_device, self.growth_threads = growth_on_device, growth_threads
        self.native = use_dist and collective == "native"
        self.dev = f"cuda:{local_rank}"
        self.views, self.ext = {}, {}
        if self.native:
            # the library reduces flags + histogram behind every pass by itself (pnx_comm_init)
            uid = [type(ctx).comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(uid[0], rank, world)
        elif use_dist:
            self.host = torch.zeros(P + 1, dtype=torch.int64).pin_memory()
            self.ev = torch.cuda.Event(blocking=blocking)

    def step(self):
        capi, hostlib, ctx, torch = self.capi, self.hostlib, self.ctx, self.torch
        device_side = self.growth_on_device and self.rank == 0
        if device_side and not self.use_dist and not os.environ.get("PANACUS_BENCH_KEEP_TABLES") and not os.environ.get("PANACUS_BENCH_PYTHON_STEP"):
            # one GPU: the whole call is ONE native call of the host library (pnh_histgrowth_resident: the library calls of the lines
            # below, in their order -- everything derived dropped, the tables' first kernels, the pass, the curves behind it, the
            # fetches -- without Python between them: ~15 us of a 0.74 ms step)
            return hostlib.histgrowth_resident(ctx, self.P, self.thr, drop_derived=True, drop_tables=True)
        ctx.config(capi.CFG_DROP_DERIVED, 0)            # no rows, no index: the pass starts from the steps
        if self.rank == 0 and not os.environ.get("PANACUS_BENCH_KEEP_TABLES"):  # (the variable: an experiment, never a reported number)
            ctx.config(capi.CFG_DROP_GROWTH_TABLES, 0)  # ... and the closed forms from (n, thresholds)
        if device_side:
            # the thresholds are known: the two small table kernels go first, while the pass's kernels are being launched (beside the
            # pass the perc_mult rows -- LDS round trips per lane -- take 0.5 ms instead of 23 us and hold the curves up behind it)
            hostlib.growth_tables_begin(self.P, self.thr)
        ctx.hist_async()
        pend = None
        if device_side and (not self.use_dist or self.native):
            # the curves follow the pass on the device, from its own (all-reduced) counters; their tables are derived on a side
            # stream while the coverage kernel runs
            pend = hostlib.calc_growths_begin_on_device(self.P, self.thr)
        if self.use_dist and not self.native:
            # the collective follows the counters on the stream of the pass, IN PLACE on the pass's device counters (the
            # library's own host copy was written by the kernel that published them, before this point of the stream) ...
            d_hist, st = ctx.hist_enqueued_on()
            ext = self.ext.get(st)
            if ext is None:
                ext = self.ext[st] = torch.cuda.ExternalStream(st, device=self.dev)
            t = self.views.get(d_hist)
            if t is None:
                t = self.views[d_hist] = torch.as_tensor(_DevArray(d_hist, self.P + 1), device=self.dev)
            reruns = int(ctx.info().n_reruns)
            with torch.cuda.stream(ext):
                self.dist.all_reduce(t)  # RCCL, int64 sum == uint64 sum for counts < 2^63
                self.host.copy_(t, non_blocking=True)
            if device_side:
                # ... and the curves follow the reduced counters on the same stream, as with one GPU: no host round trip between
                # the all-reduce and the closed forms, the tables derived beside the pass
                pend = hostlib.calc_growths_begin_on_device(self.P, self.thr)
            with torch.cuda.stream(ext):
                self.ev.record(ext)
            self.ev.synchronize()
            ctx.hist_fetch(want_countable=False)  # verifies and retires the pass
            if int(ctx.info().n_reruns) != reruns:
                # a pass that fails its verification is run again by the library, and its reduced counters would be stale; it
                # cannot happen here (pansyn paths are sorted) -- failing is better than an unmatched collective
                raise RuntimeError("a coverage pass was re-run inside the multi-GPU loop")
            h = self.host.numpy().view(np.uint64).copy()
        else:
            _, h = ctx.hist_fetch(want_countable=False)
        growths = None
        if self.rank == 0:
            if pend is None:
                pend = hostlib.calc_growths_begin(h, self.thr, self.growth_threads)
            growths = hostlib.calc_growths_end(pend)
        return h, growths

    def close(self):
        if self.native:
            self.ctx.comm_free()
        # torch objects that were used on the library's stream must go before the stream does
        self.views.clear()
        self.ext.clear()
        for a in ("host", "ev"):
            if hasattr(self, a):
                delattr(self, a)


def timed_steps(stepper, steps, warmup, barrier, sample_every):
    """W untimed steps, then exactly K steps between barriers; HIP events around every sample_every-th launch of the
    three kernels of a pass.  -> (seconds, last histogram, last curves, {slot: (ms, launches)})"""
    from panacus_amd import capi
    ctx = stepper.ctx
    h = growths = None
    for _ in range(warmup):
        h, growths = stepper.step()
    barrier()
    ctx.profile_enable(True)
    ctx.profile_select([capi.K_INDEX, capi.K_COVER, capi.K_HIST])
    ctx.profile_sample(sample_every)
    ctx.profile_reset()
    barrier()
    each = []
    t0 = time.perf_counter()
    for _ in range(steps):
        h, growths = stepper.step()
        each.append(time.perf_counter())
    barrier()
    dt = time.perf_counter() - t0
    # (the spread of the steps of the timed region, for the record: a step ends when its curves are on the host)
    raw = [b - a for a, b in zip([t0] + each[:-1], each)]
    per = sorted(raw)
    stepper.step_spread_ms = {"min": per[0] * 1e3, "median": per[len(per) // 2] * 1e3, "max": per[-1] * 1e3, "max_at_step": raw.index(per[-1])}
    prof = ctx.profile_read()
    ctx.profile_select(None)
    ctx.profile_sample(1)
    ctx.profile_reset()
    ctx.profile_enable(False)
    return dt, h, growths, prof


def closed_form_costs(ctx, h, thr, growth_threads, reps=3):
    """what the closed forms of one histogram cost by themselves (host histogram in, curves out): with the (n, thresholds)
    tables derived inside the call, and with the tables kept"""
    from panacus_amd import capi, hostlib
    first, later = [], []
    for _ in range(reps):
        ctx.config(capi.CFG_DROP_GROWTH_TABLES, 0)
        for acc in (first, later):
            g0 = time.perf_counter()
            hostlib.calc_growths_end(hostlib.calc_growths_begin(h, thr, growth_threads))
            acc.append((time.perf_counter() - g0) * 1e3)
    return {"tables_and_curves_ms": min(first), "curves_with_kept_tables_ms": min(later)}


def hist_only_block(ctx, N, P, S, steps):
    """the same one-shot pass without the closed forms behind it: `panacus hist` from the resident steps (derived data dropped
    before every call) -- the coverage kernel alone on the chip, no table kernels beside it"""
    from panacus_amd import capi
    ctx.sync()
    ts = []
    for k in range(steps + 2):
        ctx.config(capi.CFG_DROP_DERIVED, 0)
        t0 = time.perf_counter()
        ctx.hist(want_countable=False)
        if k >= 2:
            ts.append((time.perf_counter() - t0) * 1e3)
    ctx.profile_enable(True)
    ctx.profile_select([capi.K_INDEX, capi.K_COVER, capi.K_HIST])
    ctx.profile_reset()
    for _ in range(max(4, steps // 4)):
        ctx.config(capi.CFG_DROP_DERIVED, 0)
        ctx.hist(want_countable=False)
    prof = ctx.profile_read()
    ctx.profile_select(None)
    ctx.profile_reset()
    ctx.profile_enable(False)
    ts.sort()
    B = algorithmic_bytes_hist(S, P, N, P)
    per = {k: (v[0] / v[1] if v[1] else None) for k, v in prof.items()}
    ms = ts[len(ts) // 2]
    return {"what": "pnx_hist on the resident steps, derived data dropped before every call; wall clock of the call (median), kernels "
                    "timed in separate calls with HIP events around each",
            "ms_per_call": ms, "ms_min": ts[0], "calls": len(ts), "value": N * P / (ms * 1e-3) / 1e6,
            "frac_of_hbm_peak_on_algorithmic_bytes": B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "kernels_ms": {"band_index": per.get("index"), "band_cover": per.get("cover"), "hist_publish": per.get("hist")},
            "band_cover_frac_of_hbm_peak": (B / (per["cover"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if per.get("cover") else None}


def resident_pass_block(ctx, P, thr, growth_on_device, growth_threads, steps, depth):
    """What a caller that sweeps the SAME resident graph again gets: the path rows derived once (rows_route.prepare_ms), every
    further pass over the rows, pipelined `depth` deep, closed forms from kept tables.  NOT the headline: no panacus
    command sweeps one ItemTable twice with the same grouping."""
    from panacus_amd import capi, hostlib
    ctx.sync()
    ctx.config(capi.CFG_COVER_ROUTE, 2)
    ctx.config(capi.CFG_MAX_IN_FLIGHT, depth)
    prep, cold = [], []
    for _ in range(3):
        ctx.config(capi.CFG_DROP_DERIVED, 0)
        t0 = time.perf_counter()
        ctx.prepare()
        prep.append((time.perf_counter() - t0) * 1e3)
    for _ in range(3):
        ctx.config(capi.CFG_DROP_DERIVED, 0)
        t0 = time.perf_counter()
        ctx.hist(want_countable=False)
        cold.append((time.perf_counter() - t0) * 1e3)

    def enqueue():
        ctx.hist_async()
        return hostlib.calc_growths_begin_on_device(P, thr) if growth_on_device else None

    def run(n):
        h = growths = None
        queue = [enqueue() for _ in range(min(n, depth))]
        enq = len(queue)
        for _ in range(n):
            _, h = ctx.hist_fetch(want_countable=False)
            pending = queue.pop(0) or hostlib.calc_growths_begin(h, thr, growth_threads)
            if enq < n:
                queue.append(enqueue())
                enq += 1
            growths = hostlib.calc_growths_end(pending)
        return h, growths

    run(4)
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_select([capi.K_COVER])
    ctx.profile_sample(max(1, min(8, steps // 4)))
    ctx.profile_reset()
    t0 = time.perf_counter()
    h, _ = run(steps)
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    ms, cnt = ctx.profile_read()["cover"]
    ctx.profile_select(None)
    ctx.profile_sample(1)
    ctx.profile_reset()
    ctx.profile_enable(False)
    info = ctx.info()
    moved = moved_bytes_rows_pass(int(info.n_rows_in_order), int(info.n_items), P)
    k1 = ms / max(cnt, 1)
    out = {
        "what": "pipelined passes over RESIDENT path rows (derived once per upload), closed forms from kept tables: the second and "
                "later sweeps of one graph -- not a histgrowth call",
        "ms_per_pass": dt * 1e3, "passes": steps, "passes_in_flight": depth,
        "k_rows_cover_avg_launch_ms": k1, "moved_bytes_per_pass": moved,
        "k_rows_cover_frac_of_hbm_peak_on_moved_bytes": moved / (k1 * 1e-3) / 1e9 / HBM_PEAK_GBS if k1 > 0 else None,
        "rows": int(info.n_rows), "rows_bytes": 256 * int(info.n_rows),
        "rows_route": {"prepare_ms": sorted(prep)[1], "cold_first_pass_ms": sorted(cold)[1],
                       "what": "the same cold call through the path rows (PNX_CFG_COVER_ROUTE 2): steps -> rows -> pass; what round 3 ran"},
        "hist_sum": int(h.sum()),
    }
    ctx.config(capi.CFG_COVER_ROUTE, 0)
    ctx.config(capi.CFG_DROP_DERIVED, 0)
    ctx.config(capi.CFG_MAX_IN_FLIGHT, 2)
    return out


def step_report(ctx, N, P, S, dt, steps, prof, world=1):
    """the numbers of a timed one-shot loop: per-step time, kernel split, roofline of the kernel that reads the steps"""
    ms_per_step = dt / steps * 1e3
    B = algorithmic_bytes_hist(S, P, N, P)
    per = {k: (v[0] / v[1] if v[1] else None) for k, v in prof.items()}
    cover_ms, index_ms, hist_ms = per.get("cover"), per.get("index"), per.get("hist")
    kernels = sum(x for x in (cover_ms, index_ms, hist_ms) if x)
    achieved = B / (cover_ms * 1e-3) / 1e9 if cover_ms else 0.0
    info = ctx.info()
    one_shot = int(info.n_rows) == 0 and int(info.n_reruns) == 0
    return ms_per_step, B, cover_ms, {
        "bound": "hbm",
        "kernel": "k_band_cover" if one_shot else "k_rows_build + k_rows_cover",
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": None, "traffic_source": None,
        "algorithmic_bytes_per_launch": B,
        "avg_launch_ms": cover_ms, "launches": prof.get("cover", (0, 0))[1],
        "whole_step": {"ms": ms_per_step, "achieved": B / (ms_per_step * 1e-3) / 1e9, "frac": B / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "note": ROOFLINE_NOTE,
    }, {"band_index": index_ms, "band_cover": cover_ms, "hist_publish": hist_ms, "kernels_of_the_pass": kernels,
        "everything_else": ms_per_step - kernels,
        "everything_else_is": "closed-form evaluation behind the pass (k_cf_eval; the tables are derived beside the pass), kernel launches, "
                              "the wait for the results, Python"}


def shape_1k_block(args, local_rank):
    """north_star's shape: histgrowth on a 10M-node / 1k-path pansyn graph, one GPU, the same one-shot step as the headline
    (the O(n^3) quorum tables of n = 1024 are derived inside every step, beside the pass)."""
    from panacus_amd import capi, hostlib
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold
    N, P, steps = args.k1_nodes, args.k1_paths, max(2, args.k1_steps)
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]
    thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs]
    ctx = capi.Context(local_rank)
    if args.cover_route is not None:
        ctx.config(capi.CFG_COVER_ROUTE, args.cover_route)
    ctx.set_csr_pansyn(args.seed, N, P, with_weights=False)
    order = np.arange(P, dtype=np.uint32)
    ctx.set_order(order, order, P)
    hostlib.set_quorum_offload(ctx, args.quorum_offload_min_n)
    on_device = P >= args.quorum_offload_min_n and P <= 2048 and hostlib.device_growth_usable()
    stepper = OneShot(ctx, P, thr, growth_on_device=on_device, growth_threads=args.growth_threads)

    def barrier():
        ctx.sync()

    dt, h, growths, prof = timed_steps(stepper, steps, 2, barrier, max(1, min(4, steps // 4)))
    S = int(ctx.info().n_steps)
    ms_per_step, B, cover_ms, roofline, breakdown = step_report(ctx, N, P, S, dt, steps, prof)
    cf = closed_form_costs(ctx, h, thr, args.growth_threads) if on_device else None
    hist_only = hist_only_block(ctx, N, P, S, 8)
    resident = resident_pass_block(ctx, P, thr, on_device, args.growth_threads, max(8, steps), 4) if not args.no_resident else None
    if int(h.sum()) != N:
        raise SystemExit(f"shape_10Mx1k: histogram sums to {int(h.sum())}, expected {N}")
    stepper.close()
    hostlib.set_quorum_offload(None)
    ctx.close()
    return {
        "workload": f"histgrowth -c node -l 1,2,1 -q 0,0,0.5 on pansyn-v1 seed {args.seed}, {N} nodes x {P} paths (north_star's shape)",
        "steps": steps, "ms_per_step": ms_per_step, "value": N * P / (ms_per_step * 1e-3) / 1e6, "unit": "M node*paths/s",
        "steps_in_csr": S, "roofline": roofline, "step_breakdown_ms": breakdown,
        "closed_forms": cf, "closed_forms_on_gpu": bool(on_device),
        "hist_only": hist_only, "resident_pass": resident,
        "checks": {"hist_sum": int(h.sum()), "growth_last_floor": [int(np.floor(g[-1])) for g in growths]},
    }


def strayed_block(args, local_rank):
    """The headline step on paths that are NOT sorted by id: pansyn-v1r (pnx_set_csr_pansyn_rearranged -- of the 64-step blocks of
    every path 1 % reversed in place, 0.1 % replaced by a copy of an earlier block, 0.05 % moved elsewhere in the id space),
    same nodes x paths, same thresholds.  The one-shot route has to hold (no rerun, no rows): the steps that are not in the
    band their position says are spilled by k_band_cover and added by k_band_tail.  The histogram is checked against the
    oracle on the same graph (read back from HBM)."""
    from panacus_amd import capi, hostlib
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold
    N, P, steps = args.nodes, args.paths, max(4, args.strayed_steps)
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]
    thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs]
    ctx = capi.Context(local_rank)
    if args.cover_route is not None:
        ctx.config(capi.CFG_COVER_ROUTE, args.cover_route)
    ctx.set_csr_pansyn_rearranged(args.seed, N, P, with_weights=False)
    order = np.arange(P, dtype=np.uint32)
    ctx.set_order(order, order, P)
    hostlib.set_quorum_offload(ctx, args.quorum_offload_min_n)
    on_device = args.quorum_offload_min_n <= P <= 2048 and hostlib.device_growth_usable()
    stepper = OneShot(ctx, P, thr, growth_on_device=on_device, growth_threads=args.growth_threads)

    def barrier():
        ctx.sync()

    # (60 untimed steps first: ~40 ms after the block before this one has freed its 16 GB, one HIP call of this process stalls for
    # ~37 ms -- seen at the 30th step of this loop in every full run, never when the block runs by itself)
    dt, h, growths, prof = timed_steps(stepper, steps, 60, barrier, max(1, min(4, steps // 4)))
    info = ctx.info()
    S = int(info.n_steps)
    ms_per_step, B, cover_ms, roofline, breakdown = step_report(ctx, N, P, S, dt, steps, prof)
    hist_only = hist_only_block(ctx, N, P, S, 12)
    held = int(info.n_rows) == 0 and int(info.n_reruns) == 0
    agrees = None
    if not args.no_cpu_baseline:
        try:
            import oracle as orc
            items32, pre, _ = ctx.get_csr()
            pi = np.arange(P, dtype=np.uint64)
            ocov = orc.coverage(items32.astype(np.uint64), pre, pi, pi, N)
            del items32
            agrees = bool(np.array_equal(orc.hist(ocov, P), h))
            del ocov
        except Exception as e:  # the oracle is optional test infrastructure
            agrees = f"{type(e).__name__}: {e}"
        if agrees is False:
            raise SystemExit("bench: the histogram of the rearranged graph differs from the CPU oracle")
    if int(h.sum()) != N:
        raise SystemExit(f"strayed_paths: histogram sums to {int(h.sum())}, expected {N}")
    stepper.close()
    hostlib.set_quorum_offload(None)
    ctx.close()
    return {
        "workload": f"histgrowth -c node -l 1,2,1 -q 0,0,0.5 on pansyn-v1r seed {args.seed} (paths not sorted by id: 1 % of the 64-step "
                    f"blocks reversed, 0.1 % copied from earlier in the path, 0.05 % moved elsewhere), {N} nodes x {P} paths",
        "steps": steps, "ms_per_step": ms_per_step, "value": N * P / (ms_per_step * 1e-3) / 1e6, "unit": "M node*paths/s",
        "step_spread_ms": getattr(stepper, "step_spread_ms", None),
        "steps_in_csr": S, "spilled_steps_per_pass": int(info.n_spilled_last), "n_reruns": int(info.n_reruns), "n_rows": int(info.n_rows),
        "one_shot_route_held": held, "roofline": roofline, "step_breakdown_ms": breakdown, "hist_only": hist_only,
        "checks": {"hist_sum": int(h.sum()), "hist_agrees_with_oracle": agrees,
                   "growth_last_floor": [int(np.floor(g[-1])) for g in growths]},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--paths", type=int, default=256)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cover-route", type=int, default=None, help="PNX_CFG_COVER_ROUTE: 0 chosen per pass [default], 1 one-shot, 2 path rows")
    ap.add_argument("--cpu-sample-nodes", type=int, default=0,
                    help="0 [default]: the CPU baseline runs on the full headline graph; > 0: on a pansyn graph of that many nodes")
    ap.add_argument("--cpu-passes", type=int, default=3)
    ap.add_argument("--no-permuted-growth", action="store_true")
    ap.add_argument("--permuted-timeout", type=float, default=300.0, help="N > 1: seconds the permuted-growth block may take before the line is printed without it")
    ap.add_argument("--pg-nodes", type=int, default=10_000_000)
    ap.add_argument("--pg-paths", type=int, default=512)
    ap.add_argument("--pg-orders", type=int, default=128)
    ap.add_argument("--pg-reps", type=int, default=5)
    ap.add_argument("--no-shape-1k", action="store_true")
    ap.add_argument("--k1-nodes", type=int, default=10_000_000)
    ap.add_argument("--k1-paths", type=int, default=1024)
    ap.add_argument("--k1-steps", type=int, default=20)
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong_scaling block (one headline graph split by node range)")
    ap.add_argument("--strong-steps", type=int, default=40)
    ap.add_argument("--ss2-nodes", type=int, default=1_000_000, help="N > 1: nodes of the pggb-shaped graph of the second strong-scaling block")
    ap.add_argument("--ss2-samples", type=int, default=44)
    ap.add_argument("--no-strayed", action="store_true", help="skip the strayed_paths block (the headline step on pansyn-v1r: paths not sorted by id)")
    ap.add_argument("--strayed-steps", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-resident", action="store_true", help="skip the resident_pass blocks (pipelined passes over kept path rows)")
    ap.add_argument("--resident-steps", type=int, default=100)
    ap.add_argument("--depth", type=int, default=4, help="passes in flight in the resident_pass block (1..4)")
    ap.add_argument("--growth-threads", type=int, default=0)
    ap.add_argument("--collective", choices=["torch", "native"], default="torch",
                    help="who carries the RCCL all-reduce when there is one: torch.distributed's nccl backend [default] or the "
                         "library's own communicator (pnx_comm_init; the id travels through torch's store)")
    ap.add_argument("--quorum-offload-min-n", type=int, default=256)
    ap.add_argument("--allow-host-closed-forms", action="store_true",
                    help="do not fail when the closed forms of >= 256 groups cannot run on the GPU (a libm whose log2 / exp2 the "
                         "restatements do not reproduce): the host threads then set the pace of a step")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the rocprofv3 counter passes (child runs of this script) that measure roofline.traffic and roofline_valu")
    ap.add_argument("--headline-only", action="store_true", help="the timed steps and nothing else (what the counter passes run)")
    ap.add_argument("--rearranged", action="store_true", help="measurement: the headline loop on pansyn-v1r (paths not sorted by id) instead of "
                                                               "pansyn-v1 -- what the strayed_paths block runs, for profiling it alone; never the reported line")
    args = ap.parse_args()
    if args.headline_only:
        args.no_permuted_growth = args.no_shape_1k = args.no_cpu_baseline = args.no_resident = args.no_pmc = args.no_strayed = args.no_strong = True

    force_dist = os.environ.get("PANACUS_BENCH_FORCE_DIST") == "1"
    if "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and (args.gpus > 1 or force_dist):
        launch_ranks(args.gpus, force_dist)  # does not return

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
    if torch.cuda.is_available() and torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} device(s) visible")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: panacus_amd has no CPU fallback")
    blocking = world > 1 or os.environ.get("PANACUS_BENCH_BLOCKING") == "1"
    if blocking:
        # Several ranks share the host (and possibly one cgroup CPU quota): waits must sleep, not spin, or the waiting ranks
        # eat the CPU time rank 0 needs.  The flag has to be set before the HIP context of the device exists.
        try:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipSetDevice(local_rank)
            hip.hipSetDeviceFlags(0x4)  # hipDeviceScheduleBlockingSync
        except OSError:
            pass
    torch.cuda.set_device(local_rank)
    dist = None
    # PANACUS_BENCH_FORCE_DIST=1 runs the multi-GPU code path (RCCL all-reduce on the device counters) with a single rank, so
    # it can be exercised on a 1-GPU box
    use_dist = world > 1 or force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if os.environ.get("PANACUS_BENCH_ONE_DEVICE") == "1":  # several ranks on one device: RCCL cannot, gloo can (through the host)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from panacus_amd import capi, hostlib
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold

    N, P = args.nodes, args.paths
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]  # -l 1,2,1 -q 0,0,0.5
    thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs]

    ctx = capi.Context(local_rank)
    if blocking:
        ctx.config(capi.CFG_BLOCKING_SYNC, 1)
    if args.cover_route is not None:
        ctx.config(capi.CFG_COVER_ROUTE, args.cover_route)
    if args.rearranged:
        ctx.set_csr_pansyn_rearranged(args.seed + rank, N, P, with_weights=False)
    else:
        ctx.set_csr_pansyn(args.seed + rank, N, P, with_weights=False)
    order = np.arange(P, dtype=np.uint32)
    ctx.set_order(order, order, P)
    S = int(ctx.info().n_steps)
    try:
        n_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
        clock_ghz = torch.cuda.get_device_properties(local_rank).clock_rate / 1e6
        arch = getattr(torch.cuda.get_device_properties(local_rank), "gcnArchName", "")
    except Exception:
        n_cus, clock_ghz, arch = 256, 2.4, ""

    # from 256 groups on the closed forms run on the GPU, from the histogram to the curve (bit-identical, csrc/kernels_closed_form.hip)
    growth_on_device = False
    if rank == 0:
        hostlib.set_quorum_offload(ctx, args.quorum_offload_min_n)
        growth_on_device = args.quorum_offload_min_n <= P <= 2048 and hostlib.device_growth_usable()
        if args.quorum_offload_min_n <= P <= 2048 and not growth_on_device and not args.allow_host_closed_forms:
            raise SystemExit("bench: the device closed forms are not usable on this box (the restated log2 / exp2 do not reproduce its libm "
                             "bit for bit), so every step would wait for the host threads -- refusing to report that as the MI355X number "
                             "(--allow-host-closed-forms to measure it anyway)")

    stepper = OneShot(ctx, P, thr, rank=rank, world=world, use_dist=use_dist, dist=dist, torch=torch, local_rank=local_rank,
                      collective=args.collective, blocking=blocking, growth_on_device=growth_on_device, growth_threads=args.growth_threads)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    sample_every = max(1, min(4, args.steps // 4))
    dt, h, growths, prof = timed_steps(stepper, args.steps, args.warmup, barrier, sample_every)
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    out = None
    if rank == 0:
        ms_per_step, B, cover_ms, roofline, breakdown = step_report(ctx, N, P, S, dt, args.steps, prof, world)
        roofline["launches_timed"] = f"every {sample_every}th of the {args.steps} launches of the timed region"
        value = world * N * P / (ms_per_step * 1e-3) / 1e6
        cf = closed_form_costs(ctx, h, thr, args.growth_threads) if growth_on_device else None
        # counters of the kernel that reads the steps, measured by child runs under rocprofv3 -- AFTER every timed block of this
        # run (run_pmc is called just before the line is printed): the counter passes of the children leave the driver busy for
        # a while after they exit, and a 37 ms stall of one HIP call inside a later timed block was traced to them
        valu_holder = {"valu": None}

        def run_pmc():
            valu = None
            if world == 1 and not args.no_pmc and os.environ.get("PANACUS_BENCH_CHILD") != "1":
                child = ["--gpus", "1", "--steps", "4", "--warmup", "1", "--nodes", str(N), "--paths", str(P), "--seed", str(args.seed),
                         "--headline-only"]
                if args.cover_route is not None:
                    child += ["--cover-route", str(args.cover_route)]
                if args.allow_host_closed_forms:
                    child += ["--allow-host-closed-forms"]
                one_shot = roofline["kernel"] == "k_band_cover"
                names = ["k_band_cover", "k_band_index"] if one_shot else ["k_rows_build<false>", "k_rows_cover"]
                pm, src = pmc_leg(child, names, [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"]])
                roofline["traffic_source"] = src
                kc = pm.get(names[0], {})
                if "FETCH_SIZE" in kc and "WRITE_SIZE" in kc:
                    roofline["traffic"] = (2.0 * kc["FETCH_SIZE"] + kc["WRITE_SIZE"]) * 1024.0
                ki = pm.get(names[1], {})
                if "FETCH_SIZE" in ki and "WRITE_SIZE" in ki:
                    roofline["traffic_" + ("band_index" if one_shot else "rows_cover")] = (2.0 * ki["FETCH_SIZE"] + ki["WRITE_SIZE"]) * 1024.0
                if "SQ_INSTS_VALU" in kc and cover_ms:
                    # a wave64 vector instruction issues over 2 cycles on a SIMD-32 (MI355X_MICROARCH.md): peak = CUs x 4 SIMDs x clock / 2
                    peak_wi = n_cus * 4 * clock_ghz * 1e9 / 2.0
                    wi = kc["SQ_INSTS_VALU"]
                    valu = {"bound": "valu", "kernel": names[0], "wave_instructions_per_launch": wi,
                            "achieved": wi / (cover_ms * 1e-3), "peak": peak_wi, "unit": "wave64 VALU instr/s", "frac": wi / (cover_ms * 1e-3) / peak_wi,
                            "scalar_instructions_per_launch": kc.get("SQ_INSTS_SALU"), "lds_instructions_per_launch": kc.get("SQ_INSTS_LDS"),
                            "lds_bank_conflict_cycles_per_launch": kc.get("SQ_LDS_BANK_CONFLICT"),
                            "vector_instructions_per_step": wi * 64 / S, "compute_units": n_cus, "clock_ghz": clock_ghz}
            out["roofline_valu"] = valu
        out = {
            "metric": "histgrowth_throughput",
            "value": value,
            "unit": "M node*paths/s",
            "n_gpus": world,
            "one_device": os.environ.get("PANACUS_BENCH_ONE_DEVICE") == "1" and world > 1,  # (true: a test of the multi-rank path on ONE GPU over gloo, not a scaling measurement)
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"histgrowth -c node -l 1,2,1 -q 0,0,0.5 on {'pansyn-v1r (NOT the reported workload)' if args.rearranged else 'pansyn-v1'} synthetic, "
                            f"{N} nodes x {P} paths per GPU (BASELINE.json configs[2]); one step = one complete call from the resident u32 "
                            f"ItemTable, every derived table dropped between steps",
                "nodes_per_gpu": N, "paths": P, "groups": P, "steps_in_csr": S, "seed": args.seed,
                "threshold_pairs": pairs,
                "parallelism": "node-range shards, RCCL all-reduce of hist counters" if world > 1 else "single GPU",
                "collective": args.collective if use_dist else None,
            },
            "roofline": roofline,
            "roofline_valu": None,
            "step_breakdown_ms": breakdown,
            "step_spread_ms": getattr(stepper, "step_spread_ms", None),
            "closed_forms": cf,
            "closed_forms_on_gpu": bool(growth_on_device),
            "host": {"threads": hostlib.pool_threads(), "usable_cpus": hostlib.usable_cpus(), "gpu_arch": arch},
            "checks": {"hist_sum": int(h.sum()), "expected_hist_sum": world * N,
                       "growth_last_floor": [int(np.floor(g[-1])) for g in growths],
                       "one_shot_route_held": roofline["kernel"] == "k_band_cover"},
        }
    # a wrong histogram must not produce a valid-looking line: every rank leaves together
    ok = 1 if (rank != 0 or int(h.sum()) == world * N) else 0
    if use_dist:
        okt = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{local_rank}")
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = int(okt.item())
    if not ok:
        raise SystemExit(f"bench: the histogram does not sum to the number of items ({world * N})")
    # peak cross-check of the line itself: the algorithmic bytes of a step over its time cannot beat the memory
    if rank == 0 and world == 1 and out["roofline"]["whole_step"]["frac"] > 1.0:
        raise SystemExit("bench: algorithmic bytes / ms_per_step exceeds the HBM peak -- the timed step cannot be reading the steps")

    stepper.close()
    if rank == 0 and world == 1 and not args.headline_only:
        out["hist_only"] = hist_only_block(ctx, N, P, S, 20)
    # ---- what a second sweep of the same graph costs (NOT the headline) ----
    if rank == 0 and world == 1 and not args.no_resident:
        out["resident_pass"] = resident_pass_block(ctx, P, thr, growth_on_device, args.growth_threads, args.resident_steps,
                                                   max(1, min(4, args.depth)))

    def reference_binary_leg():
        """if a `panacus` >= 0.4 binary happens to be on this box's PATH it is timed too (`-t 0`: all cores) on a GFA of
        the CPU-runnable size and labelled "reference binary"; the image has none (no Rust toolchain, no network), and then the
        line says so.  Never the thing measured, never on the product path."""
        import shutil
        import subprocess
        import tempfile
        exe = shutil.which("panacus")
        if not exe:
            return {"found": False, "note": "no `panacus` binary on PATH: the CPU baseline is the oracle (kind \"port\")"}
        try:
            from panacus_amd import hostlib
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                gfa = os.path.join(td, "syn.gfa")
                n_ref, p_ref = 1_000_000, 64
                rc, _, err = hostlib.run_cli(["synth", "--nodes", str(n_ref), "--paths", str(p_ref), "--seed", str(args.seed), "-o", gfa])
                if rc != 0:
                    return {"found": True, "path": exe, "error": "synth failed: " + err[-200:]}
                t0 = time.perf_counter()
                r = subprocess.run([exe, "hist", "-c", "node", "-t", "0", gfa], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
                dt = time.perf_counter() - t0
                return {"found": True, "path": exe, "kind": "reference binary", "rc": r.returncode, "command": "panacus hist -c node -t 0 <pansyn 1M x 64 GFA>",
                        "seconds": dt, "value": n_ref * p_ref / dt / 1e6, "unit": "M node*paths/s (end to end, GFA parse included)",
                        "cores": os.cpu_count()}
        except Exception as e:
            return {"found": True, "path": exe, "error": f"{type(e).__name__}: {e}"}

    def run_cpu_baseline():
        """the oracle on the same workload (rank 0, one GPU only), AFTER the other timed blocks: its busy threads and the
        16 GB of host arrays it allocates slow the process down for whatever is timed beside them"""
        try:
            cb, h_cpu, g_cpu = cpu_baseline(ctx, N, P, pairs, args.seed, passes=args.cpu_passes,
                                            sample_nodes=args.cpu_sample_nodes or None)
            if not args.cpu_sample_nodes:
                # same workload on both sides: the results must agree bit for bit
                cb["agrees_with_gpu"] = bool(np.array_equal(h_cpu, h) and
                                             all(a.tobytes() == b.tobytes() for a, b in zip(g_cpu, growths)))
                if not cb["agrees_with_gpu"]:
                    raise SystemExit("bench: histogram / growth of the GPU path differ from the CPU oracle")
            cb["reference_binary"] = reference_binary_leg()
            out["cpu_baseline"] = cb
        except SystemExit:
            raise
        except Exception as e:  # the oracle is optional test infrastructure
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}

    def close_ctx():
        hostlib.set_quorum_offload(None)
        ctx.close()

    if use_dist:
        close_ctx()
    else:
        hostlib.set_quorum_offload(None)
    # ---- the headline workload strongly scaled: one graph split by node range (N > 1; every rank takes part) ----
    if use_dist and not args.no_strong:
        try:
            sb = strong_hist_block(args, torch, dist, use_dist, world, rank, local_rank, blocking, growth_on_device)
        except SystemExit:
            raise
        except Exception as e:
            sb = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            out["strong_scaling"] = sb
        try:
            sb2 = strong_pggb_block(args, torch, dist, use_dist, world, rank, local_rank, blocking)
        except SystemExit:
            raise
        except Exception as e:
            sb2 = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            out["strong_scaling_pggb_shape"] = sb2
    # ---- BASELINE.json configs[3]: permuted growth, strong scaling (every rank takes part) ----
    if not args.no_permuted_growth:
        # With N > 1 ranks this block is the one place where the ranks exchange data (RCCL all-reduce of the curves).  The
        # headline above is measured and complete at this point: a failure or a hang in here must not cost the line.  An
        # exception becomes an "error" entry; a watchdog prints the line without the block and ends the rank if the block has
        # not come back after --permuted-timeout seconds (a rank that died leaves the others inside a collective).
        watchdog = None
        if use_dist:
            import threading

            def give_up():
                if rank == 0:
                    out["permuted_growth"] = {"error": f"no result after {args.permuted_timeout} s: the block was abandoned"}
                    sys.stdout.flush()
                    print(json.dumps(out), flush=True)
                os._exit(0)

            watchdog = threading.Timer(args.permuted_timeout, give_up)
            watchdog.daemon = True
            watchdog.start()
        try:
            pg = permuted_growth_block(args, torch, dist, use_dist, world, rank, local_rank, blocking)
        except SystemExit:
            raise
        except Exception as e:
            if not use_dist:
                raise
            pg = {"error": f"{type(e).__name__}: {e}"}
        if watchdog is not None:
            watchdog.cancel()
        if rank == 0:
            out["permuted_growth"] = pg
    # ---- north_star's 10M x 1k shape (one GPU) ----
    if world == 1 and not use_dist and not args.no_shape_1k:
        out["shape_10Mx1k"] = shape_1k_block(args, local_rank)
    # ---- the headline step on paths that are not sorted by id (one GPU) ----
    if world == 1 and not use_dist and not args.no_strayed:
        out["strayed_paths"] = strayed_block(args, local_rank)
    if not use_dist:
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            run_cpu_baseline()
        close_ctx()
    if use_dist:
        torch.cuda.synchronize()
        dist.destroy_process_group()
    if rank == 0 and world == 1:
        run_pmc()
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
