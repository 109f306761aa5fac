#!/usr/bin/env python3
"""BASELINE.json configs[1] end to end: `hist -c bp` on a synthetic GFA FILE (pansyn-v1, 1M nodes
x 64 paths by default): host parse -> CSR -> H2D -> kernels -> TSV, next to the oracle (serial C
restatement of the reference) doing the same from the same file.  Prints one JSON line."""
import json
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    from panacus_amd import hostlib as hl
    import oracle as orc
    exe = os.path.join(os.path.dirname(hl.LIB_PATH), "panacus-amd")
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        gfa = os.path.join(d, "syn.gfa")
        t0 = time.perf_counter()
        subprocess.check_call([exe, "synth", "--nodes", str(n), "--paths", str(p), "-o", gfa], stdout=subprocess.DEVNULL)
        t_synth = time.perf_counter() - t0
        size = os.path.getsize(gfa)
        # product: in-process phases
        t0 = time.perf_counter()
        g = hl.GfaGraph(gfa)
        t_load = time.perf_counter() - t0
        t0 = time.perf_counter()
        items, pre = g.item_table(hl.BP)
        t_csr = time.perf_counter() - t0
        from panacus_amd import capi
        ctx = capi.Context(0)
        pi, gi, names = g.path_order()
        t0 = time.perf_counter()
        ctx.set_csr(items, pre, g.n_nodes, weights=g.node_lens)
        ctx.set_order(pi, gi, len(names))
        t_h2d = time.perf_counter() - t0
        t0 = time.perf_counter()
        _, h = ctx.hist(want_countable=False)
        t_hist_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        _, h = ctx.hist(want_countable=False)
        t_hist = time.perf_counter() - t0
        # product: whole CLI process (includes process start, GPU context creation)
        t0 = time.perf_counter()
        out = subprocess.run([exe, "hist", "-c", "bp", gfa], stdout=subprocess.PIPE, check=True).stdout.decode()
        t_cli = time.perf_counter() - t0
        cli_hist = [int(r.split("\t")[1]) for r in out.split("\n") if r and r[0].isdigit()]
        # the same with the binary cache: first run writes <gfa>.pcsr, second run reads it
        t0 = time.perf_counter()
        subprocess.run([exe, "hist", "-c", "bp", "--cache", gfa], stdout=subprocess.PIPE, check=True)
        t_cli_cache_write = time.perf_counter() - t0
        t0 = time.perf_counter()
        out2 = subprocess.run([exe, "hist", "-c", "bp", "--cache", gfa], stdout=subprocess.PIPE, check=True).stdout.decode()
        t_cli_cached = time.perf_counter() - t0
        cached_hist = [int(r.split("\t")[1]) for r in out2.split("\n") if r and r[0].isdigit()]
        cache_mb = os.path.getsize(gfa + ".pcsr") / 1e6
        # oracle: serial restatement from the same file
        t0 = time.perf_counter()
        og = orc.Graph(gfa)
        oitems, opre = og.item_table(orc.BP)
        t_oparse = time.perf_counter() - t0
        opi, ogi, onames = og.path_order()
        t0 = time.perf_counter()
        cov = orc.coverage(oitems, opre, opi, ogi, og.n_nodes)
        oh = orc.hist(cov, len(onames), og.node_lens)
        t_ocount = time.perf_counter() - t0
        ok = h.tolist() == oh.tolist() == cli_hist == cached_hist
        print(json.dumps({
            "workload": f"hist -c bp on synthetic GFA, {n} nodes x {p} paths ({size / 1e6:.0f} MB text, {len(items)} steps)",
            "bit_exact_vs_oracle": ok, "synth_s": t_synth,
            "product_s": {"gfa_load_index": t_load, "csr_build": t_csr, "h2d_upload+order": t_h2d,
                          "hist_first_call": t_hist_first, "hist_steady": t_hist, "cli_whole_process": t_cli,
                          "cli_cache_write_run": t_cli_cache_write, "cli_from_cache": t_cli_cached, "cache_MB": cache_mb},
            "oracle_s": {"parse_twice_serial": t_oparse, "coverage+hist_serial": t_ocount},
            "parse_MB_per_s": size / 1e6 / (t_load + t_csr),
        }))


if __name__ == "__main__":
    main()
