#!/usr/bin/env python3
"""bench.py -- histgrowth throughput of the MI355X hot path (BASELINE.json metric).

One "step" = one full histgrowth pass over one synthetic pangenome resident in HBM:
  rows of the ordered paths laid out in visiting order -> coverage over the path rows (K1) -> histogram (K2)
  on the GPU, the (G+1)-bin histogram back to the host, then the exact closed-form growth curves (f64, host
  threads + K7) for the configured (coverage, quorum) pairs -- i.e. `panacus histgrowth -c node -l 1,2,1
  -q 0,0,0.5` after the GFA has been turned into the CSR.  Resident when the timed region starts: the u32
  ItemTable and the PATH ROWS that one read of it derives per upload; nothing that depends
  on the visiting order or on an earlier pass.  What that derivation costs is part of the same JSON line:
  `cold` = prepare_ms (steps -> rows) and cold_first_pass_ms (resident u32 steps -> first histogram on the host).

`python bench.py --gpus N` launches its N ranks itself (one process per GPU, rendezvous on 127.0.0.1) unless
RANK / WORLD_SIZE are already in the environment (torch.distributed.run, the driver's way).

Workload at N = 1: BASELINE.json configs[2] ("histgrowth ... on 10M-node / 256-path
synthetic"), generator pansyn-v1 seed 42.  With --gpus N each rank owns one node-range
shard of the same shape (weak scaling: the global graph has N x 10M nodes, seeds 42+rank),
the per-rank histograms are summed with an RCCL all-reduce on the device counters, and
rank 0 evaluates the closed forms.

Besides the headline the same JSON line carries:
  * "permuted_growth" -- BASELINE.json configs[3]: ordered-histgrowth over R = 128 random group
    orders on a 10M-node / 512-path graph, STRONG scaling: the R orders are dealt to the ranks
    (permutation sharding, presence matrix replicated), every rank's out[R/N][T][G] is summed
    into the full out[R][T][G] with an RCCL all-reduce enqueued behind the growth kernels on the
    library's own stream, on the device buffer (pnx_ordered_growth_enqueued).  For N > 1 rank 0
    also times all R orders alone, so that `speedup_vs_1` comes from one run on one box.
  * "shape_10Mx1k" (N = 1 only) -- north_star's 10M-node / 1k-path histgrowth shape with its
    kernel breakdown.
  * "cpu_baseline" (N = 1 only) -- the oracle (serial port of the reference's loops; closed forms
    one thread per threshold pair like hist.rs:68-81) on the FULL headline workload.

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


def moved_bytes_hist(rows_in_order, N, G, weighted=False):
    """What a coverage pass over path rows HAS to move through HBM: the 256-byte rows of the ordered paths on the
    tiles they span (read once), the coverage vector (written by K1; K2 reads it back, and the weights if bp), the
    counters.  The steps themselves are not touched by a pass."""
    return 256 * rows_in_order + 4 * (N + 1) + 8 * (G + 1)


ROOFLINE_NOTE = ("frac = achieved / peak on the bytes the timed kernel MOVES (256-byte path rows in, coverage vector out), its "
                 "launch time measured with HIP events in this run; frac_algorithmic prices the same launch on SURVEY 8(d)'s "
                 "algorithmic bytes (4 B per path step) -- it exceeds 1 because a pass over path rows does not read the steps: "
                 "they are read ONCE per upload, by the kernel that derives the rows (see `cold`: that read is priced there, "
                 "on the same algorithmic bytes).  traffic = HBM bytes of the same kernel from rocprofv3 PMC passes driven by "
                 "this run (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; gfx950 reports half of wide streaming reads)")


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
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} HIP device(s) visible to this process -- nothing was measured")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
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
    """BASELINE.json configs[3], strong scaling by permutation sharding (module docstring)."""
    from panacus_amd import capi
    from panacus_amd.distributed import split_orders
    from panacus_amd.pansyn import random_orders
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold, coverage_abs, quorum_table

    N, P, R, reps = args.pg_nodes, args.pg_paths, args.pg_orders, max(1, args.pg_reps)
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]
    T = len(pairs)
    dev = f"cuda:{local_rank}"
    ctx = capi.Context(local_rank)
    if blocking:
        ctx.config(capi.CFG_BLOCKING_SYNC, 1)
    ctx.config(capi.CFG_KEEP_PRESENCE, 1)
    ctx.set_csr_pansyn(args.seed, N, P, with_weights=False)  # the SAME graph on every rank
    order = np.arange(P, dtype=np.uint32)
    ctx.set_order(order, order, P)
    cov = [coverage_abs(Threshold(ABSOLUTE, c), P) for c, _ in pairs]
    qt = np.stack([quorum_table(Threshold(RELATIVE, q), P) for _, q in pairs])
    perms = random_orders(args.seed, R, P)
    mine = list(split_orders(R, world, rank))
    my_perms = perms[mine] if mine else perms[:0]

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    # what every rank has to derive before it can evaluate an order: the path rows (one read of the steps) ...
    ctx.prepare()                    # first call: allocations
    prep = []
    for _ in range(3):
        ctx.config(capi.CFG_DROP_DERIVED, 0)
        sync_all()
        t0 = time.perf_counter()
        ctx.prepare()
        prep.append(time.perf_counter() - t0)
    prepare_s = sorted(prep)[len(prep) // 2]
    ctx.set_order(order, order, P)
    # ... and the presence matrix of the groups (K1 over the rows with the row stores, K2): built once per rank, then resident
    ctx.hist(want_countable=False)   # first call: allocations
    ctx.profile_enable(True)
    ctx.profile_reset()
    sync_all()
    t0 = time.perf_counter()
    ctx.hist(want_countable=False)
    pack_s = time.perf_counter() - t0
    pk = ctx.profile_read()
    info = ctx.info()

    # ---- all R orders on one GPU: the single-GPU time (every rank could; rank 0's is reported) ----
    ctx.ordered_growth(cov, qt, perms[:1])  # masks, first launch
    ctx.profile_reset()
    single = None
    t1 = None
    if rank == 0:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            single = ctx.ordered_growth(cov, qt, perms)
        t1 = (time.perf_counter() - t0) / reps
    k1 = ctx.profile_read()["growth"]
    ctx.profile_reset()

    if not use_dist:
        dt, full_host, ar_ms, gk_ms = t1, single, 0.0, k1[0] / max(k1[1], 1)
    else:
        # ---- the sharded call: my orders -> full[R][T][G] (zeros elsewhere) -> RCCL all-reduce, all on
        # the library's stream and on device buffers; rank 0 then copies the 8*R*T*G bytes to the host
        ext = torch.cuda.ExternalStream(ctx.stream(), device=dev)
        full = torch.zeros((R, T, P), dtype=torch.int64, device=dev)
        host = torch.zeros((R, T, P), dtype=torch.int64).pin_memory()
        ev_a = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev_b = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        done = torch.cuda.Event(blocking=blocking)
        idx = torch.tensor(mine, dtype=torch.int64, device=dev)
        views = {}
        native = args.collective == "native"
        if native:
            ctx.config(capi.CFG_COMM_REDUCE_HIST, 0)  # orders are sharded here, not items: nothing to reduce per pass
            uid = [capi.Context.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(uid[0], rank, world)

        def call(k):
            if mine:
                ctx.ordered_growth_async(cov, qt, my_perms)
                d_out = ctx.ordered_growth_enqueued()
                t = views.get(d_out)
                if t is None:
                    t = views[d_out] = torch.as_tensor(_DevArray(d_out, len(mine) * T * P), device=dev).view(len(mine), T, P)
            with torch.cuda.stream(ext):
                ev_a[k].record(ext)
                full.zero_()
                if mine:
                    full.index_copy_(0, idx, t)
                if native:
                    ctx.comm_allreduce_u64(full.data_ptr(), full.numel())  # the library's communicator, same stream
                else:
                    dist.all_reduce(full)  # RCCL; int64 sum == u64 sum (counts < 2^63)
                ev_b[k].record(ext)
                if rank == 0:
                    host.copy_(full, non_blocking=True)
                done.record(ext)
            done.synchronize()

        call(reps)  # warm-up (communicator, first launches)
        sync_all()
        ctx.profile_reset()
        t0 = time.perf_counter()
        for k in range(reps):
            call(k)
        sync_all()
        dt = (time.perf_counter() - t0) / reps
        gk = ctx.profile_read()["growth"]
        ar_ms = sum(ev_a[k].elapsed_time(ev_b[k]) for k in range(reps)) / reps
        red = torch.tensor([dt, gk[0] / max(gk[1], 1), ar_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        dt, gk_ms, ar_ms = (float(x) for x in red.tolist())
        full_host = host.numpy().view(np.uint64).copy() if rank == 0 else None
        if native:
            ctx.comm_free()
        views.clear()
        del ext, full, host, idx
    # ---- the same resident presence matrix through `similarity`'s group x group intersections (int8 MFMA):
    # rank 0 only, a "next"-row figure recorded with the driver's run; checked against the histogram-free identities
    sim = None
    if rank == 0:
        inter = ctx.group_intersections()  # warm-up: partial-sum buffers
        ctx.profile_reset()
        for _ in range(3):
            inter = ctx.group_intersections()
        sim_ms, sim_n = ctx.profile_read()["pairs"]
        sim_ms /= max(sim_n, 1)
        row_words = ((N + 1 + 2047) // 2048) * 64
        side = (P + 127) // 128
        ops = 2 * (side * (side + 1) // 2) * 128 * 128 * row_words * 32
        sim = {"kernel_ms": sim_ms, "int8_mfma_ops": ops, "mfma_frac_of_5_POPs_peak": ops / (sim_ms * 1e-3) / 5.0e15 if sim_ms > 0 else None,
               "checks": {"symmetric": bool((inter == inter.T).all()), "diagonal_sum": int(np.diag(inter).sum())}}
    ctx.profile_enable(False)

    out = None
    if rank == 0:
        if not np.array_equal(full_host, single):
            raise SystemExit("permuted growth: the sharded result differs from the single-GPU result")
        n_words = (N + 1 + 63) // 64
        b_growth = R * (8 * P * n_words + 8 * T * P)       # SURVEY 8(d), all R orders
        b_pack = 4 * int(info.n_steps) + 8 * P * n_words
        b_pack_layout = 256 * int(info.n_rows_in_order) + 8 * P * n_words + 4 * (N + 1)  # path rows in; presence rows + coverage vector out
        cover_ms = pk["cover"][0] / max(pk["cover"][1], 1)
        out = {
            "workload": f"ordered-histgrowth -c node -l 1,2,1 -q 0,0,0.5 over {R} random group orders (pansyn stream 7, seed "
                        f"{args.seed}), {N} nodes x {P} paths (BASELINE.json configs[3])",
            "n_gpus": world, "scaling": "strong",
            "sharding": "orders: rank r evaluates orders r, r+N, ...; presence matrix replicated on every rank; "
                        "RCCL all-reduce (sum) of out[R][T][G] on the device buffer, enqueued on the library's stream",
            "orders": R, "threshold_pairs": pairs, "orders_per_rank_max": (R + world - 1) // world, "reps": reps,
            "seconds_per_call": dt, "orders_per_s": R / dt,
            "M_node_group_orders_per_s": N * P * R / dt / 1e6,
            "seconds_per_call_1gpu": t1, "speedup_vs_1": t1 / dt,
            "growth_kernel_ms_rank_max": gk_ms, "growth_kernel_ms_1gpu": k1[0] / max(k1[1], 1),
            "allreduce_ms": ar_ms,
            "collective_path": ("none (one rank)" if not use_dist else "rccl through the library's own communicator (pnx_comm_allreduce_u64) on pnx_stream()"
                                if args.collective == "native" else "rccl via torch.distributed (nccl backend) on pnx_stream()"),
            "prepare_ms": prepare_s * 1e3,
            "presence_pack_ms": pack_s * 1e3, "presence_pack_cover_kernel_ms": cover_ms,
            "presence_pack_algorithmic_bytes": b_pack, "presence_pack_moved_bytes": b_pack_layout,
            "presence_pack_cover_kernel_GBps_on_moved_bytes": b_pack_layout / (cover_ms * 1e-3) / 1e9 if cover_ms > 0 else None,
            # a COLD call: every rank derives the rows and packs the presence matrix itself (replicated work: it does not scale)
            "seconds_per_call_incl_pack": dt + pack_s + prepare_s,
            "speedup_vs_1_incl_pack": (t1 + pack_s + prepare_s) / (dt + pack_s + prepare_s),
            "incl_pack_note": "incl_pack = prepare (steps -> path rows) + presence pack + the growth call, all replicated on every rank",
            "algorithmic_bytes": b_growth, "algorithmic_GBps": b_growth / dt / 1e9,
            "steps_in_csr": int(info.n_steps),
            "similarity_intersections": sim,
            "checks": {"sharded_equals_single_gpu": True,
                       "growth_last": [int(full_host[0, t, -1]) for t in range(T)]},
        }
    if use_dist:
        torch.cuda.synchronize()
    ctx.close()
    return out


def cold_numbers(ctx, order, G, reps=3):
    """What an upload costs before its first histogram: prepare_ms = the steps resident in HBM -> path rows (one read of
    the steps, synchronous); cold_first_pass_ms = the same + the first pass + its histogram on the host (pnx_hist on a
    graph whose derived data were dropped).  Medians of `reps` runs on a warm context (buffers exist)."""
    from panacus_amd import capi
    ctx.sync()
    prep, cold = [], []
    for _ in range(reps):
        ctx.config(capi.CFG_DROP_DERIVED, 0)
        t0 = time.perf_counter()
        ctx.prepare()
        prep.append((time.perf_counter() - t0) * 1e3)
    for _ in range(reps):
        ctx.config(capi.CFG_DROP_DERIVED, 0)
        ctx.set_order(order, order, G)
        t0 = time.perf_counter()
        ctx.hist(want_countable=False)
        cold.append((time.perf_counter() - t0) * 1e3)
    info = ctx.info()
    return {"prepare_ms": sorted(prep)[reps // 2], "cold_first_pass_ms": sorted(cold)[reps // 2],
            "rows": int(info.n_rows), "rows_bytes": 256 * int(info.n_rows), "rows_tile_major": bool(info.rows_tile_major)}


def shape_1k_block(args, local_rank):
    """north_star's shape: histgrowth on a 10M-node / 1k-path pansyn graph, one GPU, same step as the
    headline (index rebuilt every pass, closed forms included; the O(n^3) quorum sums of n = 1024 run
    on the GPU, bit-identical), with the kernel breakdown."""
    from panacus_amd import capi, hostlib
    from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold
    N, P, steps = args.k1_nodes, args.k1_paths, max(2, args.k1_steps)
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]
    thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs]
    ctx = capi.Context(local_rank)
    ctx.config(capi.CFG_CACHE_INDEX, 0)
    depth = max(1, min(4, args.depth))
    ctx.config(capi.CFG_MAX_IN_FLIGHT, depth)
    ctx.set_csr_pansyn(args.seed, N, P, with_weights=False)
    order = np.arange(P, dtype=np.uint32)
    ctx.set_order(order, order, P)
    offload = not args.no_quorum_offload and P >= args.quorum_offload_min_n
    if not args.no_quorum_offload:
        hostlib.set_quorum_offload(ctx, args.quorum_offload_min_n)

    on_device = offload and hostlib.device_growth_usable()

    def enqueue():
        """one pass; with the whole closed forms on the device, the curves follow it there (the histogram does not
        visit the host in between)"""
        ctx.hist_async()
        return hostlib.calc_growths_begin_on_device(P, thr) if on_device else None

    def run(n):
        h = growths = None
        queue = []
        enq = 0
        for _ in range(min(n, depth if P <= 511 else 2)):
            queue.append(enqueue())
            enq += 1
        for _ in range(n):
            _, h = ctx.hist_fetch(want_countable=False)
            pending = queue.pop(0) or hostlib.calc_growths_begin(h, thr, args.growth_threads)
            if enq < n:
                queue.append(enqueue())
                enq += 1
            growths = hostlib.calc_growths_end(pending)
        return h, growths

    run(3)
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_select([capi.K_COVER])
    ctx.profile_reset()
    t0 = time.perf_counter()
    h, growths = run(steps)
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    cover = ctx.profile_read()["cover"]
    ctx.profile_select(None)
    ctx.profile_reset()
    ctx.config(capi.CFG_OVERLAP_PHASES, 0)  # every kernel on its own
    run(4)
    ctx.sync()
    tail = ctx.profile_read()
    ctx.profile_enable(False)
    info = ctx.info()
    S = int(info.n_steps)
    B = algorithmic_bytes_hist(S, P, N, P)
    B_layout = moved_bytes_hist(int(info.n_rows_in_order), N, P)
    cover_beside_ms = cover[0] / max(cover[1], 1)
    cover_ms = tail["cover"][0] / max(tail["cover"][1], 1)
    index_ms = tail["scatter"][0] / max(tail["scatter"][1], 1)
    hist_ms = tail["hist"][0] / max(tail["hist"][1], 1)
    cold = cold_numbers(ctx, order, P)
    if int(h.sum()) != N:
        raise SystemExit(f"shape_10Mx1k: histogram sums to {int(h.sum())}, expected {N}")
    hostlib.set_quorum_offload(None)
    ctx.close()
    return {
        "workload": f"histgrowth -c node -l 1,2,1 -q 0,0,0.5 on pansyn-v1 seed {args.seed}, {N} nodes x {P} paths (north_star's shape)",
        "steps": steps, "ms_per_step": dt * 1e3, "value": N * P / dt / 1e6, "unit": "M node*paths/s",
        "steps_in_csr": S, "algorithmic_bytes_per_pass": B, "moved_bytes_per_pass": B_layout,
        "cold": cold,
        "breakdown_ms": {"rows_order": index_ms, "rows_cover": cover_ms, "hist": hist_ms,
                         "device_total": index_ms + cover_ms + hist_ms,
                         "rows_cover_beside_the_other_phases": cover_beside_ms,
                         "quorum_inner_sums_on_gpu": bool(offload and hostlib.quorum_offload_usable()),
                         "closed_forms_on_gpu": bool(on_device)},
        "roofline_frac_rows_cover": B_layout / (cover_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if cover_ms > 0 else None,
        "roofline_frac_whole_step": B_layout / dt / 1e9 / HBM_PEAK_GBS,
        "frac_algorithmic_rows_cover": B / (cover_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if cover_ms > 0 else None,
        "frac_algorithmic_whole_step": B / dt / 1e9 / HBM_PEAK_GBS,
        "frac_algorithmic_cold_first_pass": B / (cold["cold_first_pass_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "checks": {"hist_sum": int(h.sum()), "growth_last_floor": [int(np.floor(g[-1])) for g in growths]},
    }


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
    ap.add_argument("--cpu-sample-nodes", type=int, default=0,
                    help="0 [default]: the CPU baseline runs on the full headline graph; > 0: on a pansyn graph of that many nodes")
    ap.add_argument("--cpu-passes", type=int, default=3)
    ap.add_argument("--no-permuted-growth", action="store_true")
    ap.add_argument("--pg-nodes", type=int, default=10_000_000)
    ap.add_argument("--pg-paths", type=int, default=512)
    ap.add_argument("--pg-orders", type=int, default=128)
    ap.add_argument("--pg-reps", type=int, default=5)
    ap.add_argument("--no-shape-1k", action="store_true")
    ap.add_argument("--k1-nodes", type=int, default=10_000_000)
    ap.add_argument("--k1-paths", type=int, default=1024)
    ap.add_argument("--k1-steps", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--growth-threads", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=1,
                    help="contexts (streams) over the one resident graph whose passes alternate.  1 [default]: one "
                         "stream, the coverage kernel is timed alone (what `roofline` is defined on); 2: +11 %% passes/s, "
                         "but two coverage kernels then overlap and a launch takes 1.1 ms (DESIGN.md section 5)")
    ap.add_argument("--collective", choices=["torch", "native"], default="torch",
                    help="who carries the RCCL all-reduce when there is one: torch.distributed's nccl backend [default] or the "
                         "library's own communicator (pnx_comm_init; the id travels through torch's store)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the three phases of a pass (index | coverage kernel | histogram) on one stream instead of three")
    ap.add_argument("--no-quorum-offload", action="store_true")
    ap.add_argument("--quorum-offload-min-n", type=int, default=256)
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the rocprofv3 counter passes (child runs of this script) that measure roofline.traffic and roofline_valu")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-path block (prepare_ms, cold_first_pass_ms)")
    ap.add_argument("--rows-layout", type=int, default=None)
    ap.add_argument("--depth", type=int, default=4,
                    help="passes a context keeps in flight (PNX_CFG_MAX_IN_FLIGHT, 1..4): the latency of a step -- pass, closed forms on "
                         "their own streams, the host's share -- is several times the duration of a pass")
    args = ap.parse_args()

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
        c.config(capi.CFG_MAX_IN_FLIGHT, max(1, min(4, args.depth)))
        if blocking:
            c.config(capi.CFG_BLOCKING_SYNC, 1)
        if args.no_overlap:
            c.config(capi.CFG_OVERLAP_PHASES, 0)
        if args.index_coarse is not None:
            c.config(capi.CFG_INDEX_COARSE, args.index_coarse)
        if args.cover_waves is not None:
            c.config(capi.CFG_COVER_WAVES, args.cover_waves)
        if args.cover_split is not None:
            c.config(capi.CFG_COVER_SPLIT, args.cover_split)
        if args.cover_variant is not None:
            c.config(capi.CFG_COVER_VARIANT, args.cover_variant)
        if args.rows_layout is not None:
            c.config(capi.CFG_ROWS_LAYOUT, args.rows_layout)
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
            self.pending = []  # closed forms enqueued behind the passes in flight (None: the host starts them when it has the histogram)
            self.native = use_dist and args.collective == "native"
            if self.native:
                # the library reduces flags + histogram behind every pass by itself (pnx_comm_init): the lane
                # is then the plain single-GPU pipeline
                uid = [type(ctx).comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                ctx.comm_init(uid[0], rank, world)
            elif use_dist:
                # the per-shard counters are summed with an RCCL all-reduce that is ENQUEUED behind the
                # pass on the library's own stream (a collective on another stream would have to wait
                # for free CUs until the next pass's coverage kernel -- one resident wave per tile -- ends)
                self.ext = {}  # torch views of the library's streams
                self.ring = [{"tmp": torch.zeros(P + 1, dtype=torch.int64, device=f"cuda:{local_rank}"),
                              "host": torch.zeros(P + 1, dtype=torch.int64).pin_memory(),
                              "ev": torch.cuda.Event(blocking=blocking), "reruns": 0} for _ in range(depth)]
                self.enq = self.fin = 0

        def enqueue(self):
            ctx = self.ctx
            ctx.hist_async()
            if growth_on_device and self.ctx is all_lanes[0].ctx and (not use_dist or self.native):
                # the closed forms follow the pass on the device, from its own (all-reduced) counters
                self.pending.append(hostlib.calc_growths_begin_on_device(P, thr))
            else:
                self.pending.append(None)
            if use_dist and not self.native:
                # the collective follows the counters on the stream of the pass's histogram phase: the coverage
                # kernel of the next pass is not held back
                d_hist, st = ctx.hist_enqueued_on()
                ext = self.ext.get(st)
                if ext is None:
                    ext = self.ext[st] = torch.cuda.ExternalStream(st, device=f"cuda:{local_rank}")
                t = self.hist_views.get(d_hist)
                if t is None:
                    t = self.hist_views[d_hist] = torch.as_tensor(_DevArray(d_hist, P + 1), device=f"cuda:{local_rank}")
                slot = self.ring[self.enq % depth]
                self.enq += 1
                with torch.cuda.stream(ext):
                    slot["tmp"].copy_(t)
                    dist.all_reduce(slot["tmp"], group=self.group)  # RCCL, int64 sum == uint64 sum for counts < 2^63
                    slot["host"].copy_(slot["tmp"], non_blocking=True)
                    slot["ev"].record(ext)
                slot["reruns"] = int(ctx.info().n_reruns)

        def settle(self):
            """wait for the OLDEST enqueued pass of this lane; multi-GPU: its all-reduced counters"""
            ctx = self.ctx
            if use_dist and not self.native:
                slot = self.ring[self.fin % depth]
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
            if self.native:
                self.ctx.comm_free()
            elif use_dist:
                # torch objects that were used on the library's stream (pinned buffers record it when
                # they are freed) must go before the stream does
                self.ring.clear()
                self.hist_views.clear()
                self.ext.clear()

    n_lanes = max(1, args.lanes)
    depth = max(1, min(4, args.depth))
    ctx = make_context()
    growth_on_device = False  # set below, once the offload context is known
    all_lanes = lanes = []
    lanes.append(Lane(ctx, None))
    for _ in range(1, n_lanes):
        # every lane has its own communicator: its collectives are ordered on its own stream
        lanes.append(Lane(make_context(ctx), dist.new_group() if use_dist else None))
    S = int(ctx.info().n_steps)
    try:
        n_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
        clock_ghz = torch.cuda.get_device_properties(local_rank).clock_rate / 1e6
    except Exception:
        n_cus, clock_ghz = 256, 2.4
    # ---- the cold path of an upload, measured before anything else is resident: steps -> path rows, and the first histogram
    cold = None
    if not args.no_cold:
        ctx.hist(want_countable=False)  # allocations, code objects
        cold = cold_numbers(ctx, np.arange(P, dtype=np.uint32), P)
    else:
        ctx.prepare()
    info = ctx.info()
    rows_in_order = int(info.n_rows_in_order)

    # large group counts: the O(n^3) inner sums of the quorum closed form run on the GPU
    # (bit-identical, see csrc/kernels_closed_form.hip); below 512 groups the host is faster
    if rank == 0 and not args.no_quorum_offload:
        hostlib.set_quorum_offload(ctx, args.quorum_offload_min_n)
        growth_on_device = P >= args.quorum_offload_min_n and P <= 2048 and hostlib.device_growth_usable()

    if growth_on_device and cold is not None:
        # first closed-form call of a context: the (n, thresholds) tables are derived (log2 table, running sums, perc_mult,
        # the quorum pair's inner sums), then the evaluation; calls that find the tables are the evaluation alone
        _, h0 = ctx.hist(want_countable=False)
        first, later = [], []
        for _ in range(3):
            ctx.config(capi.CFG_DROP_GROWTH_TABLES, 0)
            for acc in (first, later):
                g0 = time.perf_counter()
                hostlib.calc_growths_end(hostlib.calc_growths_begin(h0, thr, args.growth_threads))
                acc.append((time.perf_counter() - g0) * 1e3)
        cold["growth_tables_ms"] = min(first)
        cold["growth_call_ms"] = min(later)

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
        for _ in range(min(n_steps, depth * L)):
            lanes[enqueued % L].enqueue()
            enqueued += 1
        late = None  # the closed forms of the previous pass: collected one pass late, when they are long done
        for k in range(n_steps):
            h = lanes[k % L].settle()
            on_device = lanes[k % L].pending.pop(0)
            pending = on_device or growth_begin(h)
            if enqueued < n_steps:
                lanes[enqueued % L].enqueue()
                enqueued += 1
            if on_device is None:  # host threads (+ one quorum offload at a time): finished here
                growths = growth_end(pending)
                continue
            if late is not None:
                growths = growth_end(late)
            late = pending
        if late is not None:
            growths = growth_end(late)
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
    # (every 8th launch: two event records around EVERY coverage kernel keep the stream from running the kernels back to back --
    # 0.185 against 0.140 ms per step measured)
    sample_every = max(1, min(8, args.steps // 4))
    for ln in lanes:
        ln.ctx.profile_enable(True)
        ln.ctx.profile_select([capi.K_COVER])
        ln.ctx.profile_sample(sample_every)
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
        ln.ctx.profile_sample(1)
        ln.ctx.profile_reset()
        if ln.ctx is not ctx:
            ln.ctx.profile_enable(False)
    ctx.config(capi.CFG_OVERLAP_PHASES, 0)
    run(5, lanes[:1])  # every kernel on its own: one lane, the three phases of a pass on one stream
    barrier()
    prof_tail = ctx.profile_read()
    ctx.profile_enable(False)
    ctx.config(capi.CFG_OVERLAP_PHASES, 0 if args.no_overlap else 1)
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
        index_ms, index_n = prof_tail["scatter"]   # k_rows_order: the rows of the ordered paths in visiting order
        hist_ms, hist_n = prof_tail["hist"]
        cover_avg_ms = cover_ms / max(cover_n, 1)
        B = algorithmic_bytes_hist(S, P, N, P)
        B_moved = moved_bytes_hist(rows_in_order, N, P)
        achieved_alg = B / (cover_avg_ms * 1e-3) / 1e9 if cover_avg_ms > 0 else 0.0
        achieved = B_moved / (cover_avg_ms * 1e-3) / 1e9 if cover_avg_ms > 0 else 0.0
        cover_alone_ms = prof_tail["cover"][0] / max(prof_tail["cover"][1], 1)
        device_ms = cover_alone_ms + index_ms / max(index_n, 1) + hist_ms / max(hist_n, 1)
        if cold is not None:
            # a pass that kept nothing would read the steps every time; the cheapest full read of the steps measured here is the
            # derivation itself, so: prepare + k * step <= k * prepare  <=>  k >= prepare / (prepare - step)
            cold["value_cold_first_pass"] = world * N * P / (cold["cold_first_pass_ms"] * 1e-3) / 1e6
            cold["frac_algorithmic_cold_first_pass"] = B / (cold["cold_first_pass_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            cold["frac_algorithmic_prepare"] = 4 * S / (cold["prepare_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            cold["passes_to_break_even"] = (int(np.ceil(cold["prepare_ms"] / (cold["prepare_ms"] - ms_per_step)))
                                            if cold["prepare_ms"] > ms_per_step else None)
            cold["note"] = ("prepare_ms: the u32 steps resident in HBM -> path rows (one read of the steps, ids validated on the way), "
                            "synchronous; cold_first_pass_ms: pnx_hist on a graph whose derived data were dropped = prepare + one pass + "
                            "the histogram on the host; frac_algorithmic_*: SURVEY 8(d)'s algorithmic bytes over those times -- the read "
                            "of the steps is priced HERE, not in `roofline`; passes_to_break_even: against re-reading the steps in "
                            "every pass, taking the derivation itself as the cheapest full read of the steps")
        # counters of the dominant kernel (and of the kernel that derives the rows), measured by child runs under rocprofv3
        traffic = traffic_src = None
        valu = None
        cold_traffic = None
        if world == 1 and not args.no_pmc and os.environ.get("PANACUS_BENCH_CHILD") != "1":
            child = ["--gpus", "1", "--steps", "4", "--warmup", "1", "--nodes", str(N), "--paths", str(P), "--seed", str(args.seed),
                     "--no-permuted-growth", "--no-shape-1k", "--no-cpu-baseline", "--no-pmc"]
            for flag, val in (("--cover-split", args.cover_split), ("--rows-layout", args.rows_layout), ("--cover-variant", args.cover_variant)):
                if val is not None:
                    child += [flag, str(val)]
            pm, traffic_src = pmc_leg(child, ["k_rows_cover", "k_rows_build<false>"], [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"]])
            kc = pm.get("k_rows_cover", {})
            if "FETCH_SIZE" in kc and "WRITE_SIZE" in kc:
                traffic = (2.0 * kc["FETCH_SIZE"] + kc["WRITE_SIZE"]) * 1024.0
            if "SQ_INSTS_VALU" in kc:
                # a wave64 vector instruction issues over 2 cycles on a SIMD-32 (MI355X_MICROARCH.md): peak = CUs x 4 SIMDs x clock / 2
                peak_wi = n_cus * 4 * clock_ghz * 1e9 / 2.0
                wi = kc["SQ_INSTS_VALU"]
                valu = {"bound": "valu", "kernel": "k_rows_cover", "wave_instructions_per_launch": wi,
                        "achieved": wi / (cover_avg_ms * 1e-3), "peak": peak_wi, "unit": "wave64 VALU instr/s", "frac": wi / (cover_avg_ms * 1e-3) / peak_wi,
                        "scalar_instructions_per_launch": kc.get("SQ_INSTS_SALU"), "lds_instructions_per_launch": kc.get("SQ_INSTS_LDS"),
                        "per_row": wi * 64 / max(rows_in_order, 1) / 64, "compute_units": n_cus, "clock_ghz": clock_ghz}
            kb = pm.get("k_rows_build<false>", {})
            if "FETCH_SIZE" in kb and "WRITE_SIZE" in kb:
                cold_traffic = (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0
                if cold is not None:
                    cold["build_kernel_traffic_bytes"] = cold_traffic
                    cold["build_kernel_valu_wave_instructions"] = kb.get("SQ_INSTS_VALU")
                    cold["build_kernel_lds_instructions"] = kb.get("SQ_INSTS_LDS")
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
                "collective": args.collective if use_dist else None,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_rows_cover",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "moved_bytes_per_launch": B_moved,
                "avg_launch_ms": cover_avg_ms,
                "launches": cover_n,
                "launches_timed": f"every {sample_every}th of the {args.steps} launches of the timed region",
                "algorithmic_bytes_per_launch": B,
                "achieved_algorithmic": achieved_alg,
                "frac_algorithmic": achieved_alg / HBM_PEAK_GBS,
                "avg_launch_ms_alone": cover_alone_ms,
                "note": ROOFLINE_NOTE + "; avg_launch_ms is measured over the timed steps, where the order layout of the next pass and "
                        "the histogram of the previous one run beside the kernel on their own streams; avg_launch_ms_alone is the same "
                        "kernel with the phases on one stream (5 extra passes)",
            },
            "roofline_valu": valu,
            "cold": cold,
            "breakdown_ms": {
                "rows_order": index_ms / max(index_n, 1), "rows_cover": cover_alone_ms,
                "hist": hist_ms / max(hist_n, 1), "device_total": device_ms,
                "rows_cover_beside_the_other_phases": cover_avg_ms,
                "lanes": len(lanes), "passes_in_flight": depth,
                "host_closed_form_growth": growth_ms, "host_threads": hostlib.pool_threads(), "host_usable_cpus": hostlib.usable_cpus(),
                "quorum_inner_sums_on_gpu": bool(not args.no_quorum_offload and P >= args.quorum_offload_min_n
                                                 and hostlib.quorum_offload_usable()),
                "closed_forms_on_gpu": bool(growth_on_device),
                "single_pass_latency": latency_ms,
            },
            "hbm_gbs_whole_step_moved_bytes": B_moved / (ms_per_step * 1e-3) / 1e9,
            "hbm_gbs_whole_step_algorithmic": B / (ms_per_step * 1e-3) / 1e9,
            "checks": {"hist_sum": int(h.sum()), "expected_hist_sum": world * N,
                       "growth_last_floor": [int(np.floor(g[-1])) for g in growths]},
        }
    # a wrong histogram must not produce a valid-looking line: every rank leaves together
    ok = 1 if (rank != 0 or int(h.sum()) == world * N) else 0
    if use_dist:
        okt = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{local_rank}")
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = int(okt.item())
    if not ok:
        raise SystemExit(f"bench: the histogram does not sum to the number of items ({world * N})")
    def run_cpu_baseline():
        """the oracle on the same workload (rank 0, one GPU only), AFTER the other timed blocks: its three busy threads and
        the 16 GB of host arrays it allocates left the process slower for the closed forms of the 10 M x 1 k block
        (3.5 against 2.6 ms per step when it ran first)"""
        try:
            cb, h_cpu, g_cpu = cpu_baseline(ctx, N, P, pairs, args.seed, passes=args.cpu_passes,
                                            sample_nodes=args.cpu_sample_nodes or None)
            if not args.cpu_sample_nodes:
                # same workload on both sides: the results must agree bit for bit
                cb["agrees_with_gpu"] = bool(np.array_equal(h_cpu, h) and
                                             all(a.tobytes() == b.tobytes() for a, b in zip(g_cpu, growths)))
                if not cb["agrees_with_gpu"]:
                    raise SystemExit("bench: histogram / growth of the GPU path differ from the CPU oracle")
            out["cpu_baseline"] = cb
        except SystemExit:
            raise
        except Exception as e:  # the oracle is optional test infrastructure
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}

    def close_lanes():
        for ln in lanes:
            ln.close()
        hostlib.set_quorum_offload(None)
        for ln in reversed(lanes):  # borrowers of the resident graph before its owner
            ln.ctx.close()

    if use_dist:
        close_lanes()
    else:
        hostlib.set_quorum_offload(None)
    # ---- BASELINE.json configs[3]: permuted growth, strong scaling (every rank takes part) ----
    if not args.no_permuted_growth:
        pg = permuted_growth_block(args, torch, dist, use_dist, world, rank, local_rank, blocking)
        if rank == 0:
            out["permuted_growth"] = pg
    # ---- north_star's 10M x 1k shape (one GPU) ----
    if world == 1 and not args.no_shape_1k:
        out["shape_10Mx1k"] = shape_1k_block(args, local_rank)
    if not use_dist:
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            run_cpu_baseline()
        close_lanes()
    if use_dist:
        torch.cuda.synchronize()
        dist.destroy_process_group()
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
