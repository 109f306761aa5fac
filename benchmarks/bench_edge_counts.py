#!/usr/bin/env python3
"""`hist -c edge` on a synthetic GFA with L lines: which route the edge ItemTable takes through the
coverage kernel (edge ids follow the first-seen order of the L lines, graph.rs:282-295, so the
steps of a path are only as monotone as the link order of the file) and what a pass costs next to
the node pass of the same graph.  L lines sorted by (from, to) as odgi / pggb write them, and the
same links shuffled; "edge" = the reference's ids through plain pnx_set_csr; "edge_keyed" = the same ids
plus one sort key per edge (its canonical ends) through pnx_set_csr_keyed, which renumbers them on the
device -- what the CLI uploads; "edge_renumbered" = ids ranked on the host (GraphStorage::edge_relabel)
before a plain upload, the round-1 route.  Prints one JSON line per case."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    from panacus_amd import capi, hostlib as hl
    exe = os.path.join(os.path.dirname(hl.LIB_PATH), "panacus-amd")
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        gfa = os.path.join(d, "syn.gfa")
        subprocess.check_call([exe, "synth", "--nodes", str(n), "--paths", str(p), "--links", "-o", gfa], stdout=subprocess.DEVNULL)
        shuf = os.path.join(d, "shuf.gfa")
        lines = open(gfa).read().split("\n")
        links = [l for l in lines if l.startswith("L\t")]
        rest = [l for l in lines if l and not l.startswith("L\t")]
        rng = np.random.default_rng(1)
        perm = rng.permutation(len(links))
        with open(shuf, "w") as f:  # S lines, shuffled L lines, then the paths
            f.write("\n".join([l for l in rest if not l.startswith(("P\t", "W\t"))] + [links[i] for i in perm] +
                              [l for l in rest if l.startswith(("P\t", "W\t"))]) + "\n")
        for name, path in (("links sorted by (from, to)", gfa), ("links shuffled", shuf)):
            g = hl.GfaGraph(path, index_edges=True)
            pi, gi, names = g.path_order()
            res = {"workload": f"{n} nodes x {p} paths, {g.n_edges} edges, {name}"}
            with capi.Context(0) as ctx:
                for cname, ct in (("node", hl.NODE), ("edge", hl.EDGE), ("edge_keyed", hl.EDGE), ("edge_renumbered", hl.EDGE)):
                    items, pre = g.item_table(ct)
                    keys = None
                    if cname == "edge_renumbered":
                        t0 = time.perf_counter()
                        new_id = g.edge_relabel()
                        res["renumber_host_ms"] = (time.perf_counter() - t0) * 1e3
                        items = new_id[items]
                    if cname == "edge_keyed":  # what the CLI uploads
                        keys = g.edge_keys()
                        ctx.set_csr(items, pre, g.n_items(ct))      # warm: allocations
                        t0 = time.perf_counter()
                        ctx.set_csr(items, pre, g.n_items(ct))
                        t_plain = time.perf_counter() - t0
                    t0 = time.perf_counter()
                    ctx.set_csr(items, pre, g.n_items(ct), item_key=keys)
                    if cname == "edge_keyed":
                        res["relabel_on_device_ms"] = (time.perf_counter() - t0 - t_plain) * 1e3
                    ctx.set_order(pi, gi, len(names))
                    t0 = time.perf_counter()
                    ctx.hist(want_countable=False)
                    first_ms = (time.perf_counter() - t0) * 1e3
                    t0 = time.perf_counter()
                    for _ in range(5):
                        _, h = ctx.hist(want_countable=False)
                    dt = (time.perf_counter() - t0) / 5
                    info = ctx.info()
                    res[cname] = {"steps": int(len(items)), "hist_ms": dt * 1e3, "first_hist_ms": first_ms, "general_paths": int(info.n_general_paths),
                                  "run_paths": int(info.n_run_paths), "scatter_paths": int(info.n_scatter_paths),
                                  "runs": int(info.n_runs), "hist_sum": int(h.sum())}
            print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
