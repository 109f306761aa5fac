"""More than one GPU: one process per device, node-range shards or order shards, the exchange an RCCL all-reduce of
small u64 counters over xGMI.  These tests need at least TWO visible devices and skip otherwise
(the build box and the 1-GPU test boxes have one); the same host code runs with two ranks over gloo in
tests/test_distributed_gloo.py and tests/test_host_cli.py, and with one rank through RCCL in tests/test_gpu_parity.py."""
import json
import os
import subprocess
import sys

import pytest

from panacus_amd import hostlib as hl

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


needs_two = pytest.mark.skipif(_devices() < 2, reason="needs at least two visible GPUs")


def _body(text):
    return "\n".join(l for l in text.split("\n") if not l.startswith("#"))


def _clean_env(**extra):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(extra)
    return e


def _bumpy_gfa(tmp_path):
    """paths that are NOT tile-monotone (a 40-step window reversed every 300 steps)"""
    src = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "60000", "--paths", "12", "-o", src])
    assert rc == 0, err
    gfa = str(tmp_path / "bumpy.gfa")
    with open(src) as f, open(gfa, "w") as g:
        for line in f:
            if line.startswith("P\t"):
                cols = line.rstrip("\n").split("\t")
                steps = cols[2].split(",")
                for a in range(100, len(steps) - 60, 300):
                    steps[a:a + 40] = steps[a:a + 40][::-1]
                cols[2] = ",".join(steps)
                line = "\t".join(cols) + "\n"
            g.write(line)
    return gfa


@needs_two
@pytest.mark.parametrize("backend,variant", [("native", "3"), ("native", "2"), ("nccl", "3")])
def test_two_gpus_node_range_shards_equal_one_gpu(tmp_path, backend, variant):
    """histgrowth of one GFA on two GPUs (node-range shards; the library's own communicator reduces flags + histogram
    behind every pass -- with PNX_COVER_VARIANT=2 also behind the RE-RUN of a pass whose paths are not tile-monotone,
    which every rank must take together) prints the table of the single-GPU CLI"""
    import socket
    gfa = _bumpy_gfa(tmp_path)
    for cname in ("node", "bp"):
        rc, ref, err = hl.run_cli(["histgrowth", "-a", "-c", cname, "-l", "1,2", "-q", "0,0.5", gfa])
        assert rc == 0, err
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        out_file = str(tmp_path / f"two_{backend}_{variant}_{cname}.tsv")
        procs = []
        for r in range(2):
            e = _clean_env(RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                           PANACUS_DIST_BACKEND=backend, PANACUS_TOOL_REPORT_RERUNS="1", PNX_COVER_VARIANT=variant,
                           PANACUS_COMM_ID_FILE=str(tmp_path / f"comm_{backend}_{variant}_{cname}.id"),
                           HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "histgrowth_multi_gpu.py"), "-c", cname,
                                           "-l", "1,2", "-q", "0,0.5", "-o", out_file, gfa], env=e,
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE))
        errs = []
        for pr in procs:
            o, e2 = pr.communicate(timeout=600)
            assert pr.returncode == 0, e2.decode()[-2000:]
            errs.append(e2.decode())
        if variant == "2":
            assert "reruns=0" not in errs[0] and "reruns=" in errs[0]   # the re-run really happened, on both ranks together
        assert open(out_file).read() == _body(ref).rstrip("\n") + "\n"
        assert not os.path.exists(str(tmp_path / f"comm_{backend}_{variant}_{cname}.id"))   # the id file does not outlive the launch


@needs_two
@pytest.mark.parametrize("collective", ["torch", "native"])
def test_bench_launches_its_ranks_and_shards_the_nodes(collective):
    """`bench.py --gpus 2` starts two ranks by itself; the JSON line says n_gpus = 2, the weak-scaling histogram covers
    both shards, and the permuted growth summed over the node-range shards equals the single-GPU result on the whole graph bit for
    bit (the bench fails otherwise)"""
    small = ["--nodes", "200000", "--paths", "64", "--steps", "12", "--warmup", "2", "--pg-nodes", "150000", "--pg-paths", "40",
             "--pg-orders", "10", "--pg-reps", "2", "--no-pmc", "--strong-steps", "8", "--collective", collective]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + small, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=_clean_env(HSA_ENABLE_IPC_MODE_LEGACY="0"), timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().split("\n") if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["checks"]["hist_sum"] == d["checks"]["expected_hist_sum"] == 400000
    pg = d["permuted_growth"]
    assert pg["n_gpus"] == 2 and pg["scaling"] == "strong" and pg["checks"]["sharded_equals_single_gpu"] and pg["allreduce_ms"] > 0
    assert pg["nodes_per_rank_max"] == 75000 and "rccl" in pg["collective_path"] and pg["speedup_vs_1"] > 0
    # the headline graph itself split into two node ranges: same histogram and curves as rank 0 alone on the whole
    # graph (the bench fails otherwise), both times from this run
    sb = d["strong_scaling"]
    for k in ("workload", "n_gpus", "scaling", "sharding", "ms_per_step", "ms_per_step_1gpu", "speedup_vs_1", "value", "rank0", "alone", "checks"):
        assert k in sb, (k, sb)
    assert sb["n_gpus"] == 2 and sb["scaling"] == "strong" and sb["checks"]["hist_sum"] == 200000 and sb["checks"]["sharded_equals_single_gpu"]
    assert sb["rank0"]["nodes"] < 200000 and sb["alone"]["nodes"] == 200000 and sb["speedup_vs_1"] > 0


@needs_two
def test_two_gpus_unequal_shards_one_with_paths_that_do_not_follow_the_ids():
    """the ranks of a communicator may take different routes: rank 0 takes the one-shot route and its shard's paths are shuffled (its
    pass is void), rank 1's shard is sorted and takes the path rows.  The flags are reduced with the histogram, so BOTH ranks see the void flag and run the
    pass again -- whatever their own route was -- and the collectives stay matched (a hang or a corrupted
    histogram otherwise).  Checked against one context on the whole graph."""
    code = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch, torch.distributed as dist
rank = int(os.environ["RANK"]); torch.cuda.set_device(rank)
dist.init_process_group("gloo")
from panacus_amd import capi
import oracle as orc
n, p = 120_000, 20                                    # (20 shuffled paths: more groups than a pass leaves to bitmaps)
items, pre, _ = orc.pansyn(3, n, p)
half = 61_440
def shard(lo, hi):
    segs = []
    for k in range(p):
        s = items[int(pre[k]):int(pre[k + 1])]
        s = s[(s > lo) & (s <= hi)] - lo
        segs.append(s)
    return segs
segs = shard(0, half) if rank == 0 else shard(half, n)
if rank == 0:
    rng = np.random.default_rng(1)
    segs = [np.concatenate([s[:len(s) * 55 // 100], rng.permutation(s[len(s) * 55 // 100:])]) for s in segs]   # the last 45 % of every path shuffled: 20 loose groups, more than a pass takes in: rank 0's one-shot pass will be void
pre_r = np.zeros(p + 1, dtype=np.uint64); pre_r[1:] = np.cumsum([len(s) for s in segs])
it_r = np.concatenate(segs).astype(np.uint32)
c = capi.Context(rank)
c.config(capi.CFG_COVER_ROUTE, 1 if rank == 0 else 0)   # rank 0: one-shot; rank 1: whatever the shape says (rows)
uid = [capi.Context.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
c.comm_init(uid[0], rank, 2)
c.set_csr(it_r, pre_r, half if rank == 0 else n - half)
o = np.arange(p, dtype=np.uint32)
c.set_order(o, o, p)
_, h = c.hist(want_countable=False)
_, h2 = c.hist(want_countable=False)
pi = np.arange(p, dtype=np.uint64)
want = orc.hist(orc.coverage(items, pre, pi, pi, n), p)
assert np.array_equal(h, want) and np.array_equal(h2, want), (rank, h.tolist(), want.tolist())
c.comm_free(); c.close()
dist.barrier()
print("ok", rank)
""" % ROOT
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              env=_clean_env(RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                                             HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(2)]
    for pr in procs:
        out, err = pr.communicate(timeout=600)
        assert pr.returncode == 0 and b"ok" in out, err.decode()[-3000:]
