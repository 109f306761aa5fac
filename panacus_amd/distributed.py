"""Multi-GPU host logic: one process per GPU, node-range shards, RCCL all-reduce of counters.

Every quantity on the hot path is a sum over items (nodes / edges), so the graph shards by
item-id range with NO data-path collective: each rank runs the unchanged kernels on the CSR
steps whose id falls into its range, and the only exchange is an all-reduce (sum) of the small
integer counter arrays -- hist[G+1] (<= a few KB) and, for permuted growth, out[R][T][G].
Backend "nccl" is RCCL on ROCm (xGMI); the same code runs over "gloo" on CPU for the tests.
"""
from __future__ import annotations

import numpy as np


def plan_node_shards(items: np.ndarray, n_items: int, world: int) -> np.ndarray:
    """Item-id range boundaries, balanced by STEP count (not by item count).

    Returns `cuts` (world+1 ints, cuts[0] = 1, cuts[-1] = n_items+1); rank r owns ids in
    [cuts[r], cuts[r+1]).  A path step with id i costs one unit for the rank that owns i.
    """
    if world < 1:
        raise ValueError("world must be >= 1")
    steps_per_item = np.bincount(np.asarray(items, dtype=np.int64), minlength=n_items + 1)
    cum = np.cumsum(steps_per_item)
    total = int(cum[-1])
    cuts = [1]
    for r in range(1, world):
        target = total * r // world
        c = int(np.searchsorted(cum, target, side="left")) + 1
        cuts.append(min(max(c, cuts[-1]), n_items + 1))
    cuts.append(n_items + 1)
    return np.asarray(cuts, dtype=np.int64)


def shard_csr(items: np.ndarray, path_off: np.ndarray, lo: int, hi: int):
    """The CSR restricted to item ids in [lo, hi), re-based to 1..(hi-lo).

    Path order and path count are preserved (a path without steps in the range becomes empty),
    so the same visiting order (path_idx, group_id) applies to every shard.
    -> (items_r u32, path_off_r u64, n_items_r)
    """
    items = np.asarray(items)
    path_off = np.asarray(path_off, dtype=np.uint64)
    keep = (items >= lo) & (items < hi)
    kept_before = np.concatenate([[0], np.cumsum(keep, dtype=np.uint64)])
    new_off = kept_before[path_off.astype(np.int64)]
    new_items = (items[keep] - (lo - 1)).astype(np.uint32)
    return new_items, new_off.astype(np.uint64), int(hi - lo)


def shard_weights(weights: np.ndarray | None, lo: int, hi: int):
    if weights is None:
        return None
    w = np.zeros(hi - lo + 1, dtype=np.uint32)
    w[1:] = np.asarray(weights)[lo:hi]
    return w


def allreduce_counters(arr: np.ndarray, device=None) -> np.ndarray:
    """Sum a u64 counter array over all ranks (int64 transport: exact for counts < 2^63)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(arr, dtype=np.uint64)
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().view(np.uint64)


def gather_countable(local: np.ndarray, cuts: np.ndarray, rank: int, device=None) -> np.ndarray:
    """Concatenate the per-shard coverage vectors (local[1:] of every rank) into the global
    countable (index 0 = u32::MAX like the reference)."""
    import torch
    import torch.distributed as dist
    n_items = int(cuts[-1]) - 1
    out = np.zeros(n_items + 1, dtype=np.int64)
    out[int(cuts[rank]):int(cuts[rank + 1])] = np.asarray(local[1:], dtype=np.int64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.from_numpy(out)
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out = t.cpu().numpy()
    out = out.astype(np.uint32)
    out[0] = 0xFFFFFFFF
    return out


def even_node_range(n_items: int, world: int, rank: int):
    """Node-range shard of a graph whose steps are spread evenly over the ids (pansyn): rank r owns the nodes
    lo + 1 .. hi, ranges of whole 2048-item tiles.  -> (lo, hi)"""
    tiles = (n_items + 2047) // 2048
    lo = min(n_items, (tiles * rank // world) * 2048)
    hi = min(n_items, (tiles * (rank + 1) // world) * 2048) if rank + 1 < world else n_items
    return lo, hi


def split_orders(n_orders: int, world: int, rank: int) -> range:
    """Permutation sharding for permuted growth: rank r evaluates orders r, r+world, ..."""
    return range(rank, n_orders, world)


def default_comm_id_file() -> str:
    """Where the ranks of ONE launch meet when nobody named a file: a name that is private to the launch -- the launcher's
    pid (every rank is its child) and, under torchrun, its run id -- in the user's runtime directory rather than in /tmp.
    A launcher that can do better hands its children PANACUS_COMM_ID_FILE inside a directory it created (bench.py does)."""
    import os
    explicit = os.environ.get("PANACUS_COMM_ID_FILE")
    if explicit:
        return explicit
    base = os.environ.get("XDG_RUNTIME_DIR") or os.path.join(os.path.expanduser("~"), ".cache", "panacus_amd")
    os.makedirs(base, mode=0o700, exist_ok=True)
    tag = f"{os.getppid()}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}_{os.environ.get('MASTER_PORT', '0')}"
    return os.path.join(base, f"comm_{tag}.id")


def launch_nonce() -> bytes:
    """16 bytes that every rank of ONE launch computes alike and no other launch does: PANACUS_COMM_NONCE if the launcher
    exports one (required for launchers whose ranks do not share a parent process -- srun, mpirun -- together with
    PANACUS_COMM_ID_FILE), else torchrun's run id + rendezvous port, else the START TIME OF THE PARENT process (the launcher
    that forked every rank) + the rendezvous port.  Not this process's own start time: a rank that starts, or is restarted,
    seconds after rank 0 published the id must still recognise it."""
    import hashlib
    import os
    src = os.environ.get("PANACUS_COMM_NONCE")
    if not src:
        run = os.environ.get("TORCHELASTIC_RUN_ID")
        port = os.environ.get("MASTER_PORT", "0")
        src = f"torchrun|{run}|{port}" if run else f"parent|{os.getppid()}|{_process_start_ticks(os.getppid())}|{port}"
    return hashlib.sha256(src.encode()).digest()[:16]


def native_comm_init(ctx, rank: int, world: int, id_file: str | None = None, timeout_s: float = 120.0):
    """Give `ctx` the library's own RCCL communicator (pnx_comm_init): rank 0 asks the library for the
    128-byte id and publishes it through `id_file` -- [launch nonce 16 B | id 128 B], created exclusively under a temporary
    name and renamed, so a reader never sees half of it and a planted file is never followed; the other ranks wait for a
    file that carries THIS launch's nonce (a stale file of an earlier launch does not, whenever it was written).  Once
    every rank holds the communicator -- one all-reduce through it proves that -- rank 0 removes the file.  No torch
    involved: the same steps a Rust host takes."""
    import os
    import time
    if id_file is None:
        id_file = default_comm_id_file()
    nonce = launch_nonce()
    if rank == 0:
        uid = type(ctx).comm_unique_id()
        try:
            os.unlink(id_file)  # a stale file of an earlier launch that died
        except FileNotFoundError:
            pass
        tmp = f"{id_file}.{os.getpid()}.tmp"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(nonce + uid)
        os.replace(tmp, id_file)
    else:
        t0 = time.time()
        uid = b""
        while True:
            try:
                st = os.stat(id_file, follow_symlinks=False)
                if st.st_size == 16 + 128:
                    with open(id_file, "rb") as f:
                        blob = f.read()
                    if len(blob) == 16 + 128 and blob[:16] == nonce:
                        uid = blob[16:]
                        break
            except FileNotFoundError:
                pass
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"no communicator id of this launch at {id_file} after {timeout_s} s (ranks that do not share a "
                                   f"parent process need PANACUS_COMM_ID_FILE and PANACUS_COMM_NONCE from their launcher)")
            time.sleep(0.01)
    ctx.comm_init(uid, rank, world)
    ctx.comm_barrier()  # every rank holds the communicator once this returns
    if rank == 0:
        try:
            os.unlink(id_file)
        except FileNotFoundError:
            pass
    return uid


def _process_start_ticks(pid: int) -> int:
    """start time of a process in clock ticks since boot (0 if unknown): identifies the process together with its pid"""
    try:
        with open(f"/proc/{pid}/stat") as f:
            return int(f.read().rsplit(")", 1)[1].split()[19])
    except Exception:
        return 0
