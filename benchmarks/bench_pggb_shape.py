"""BASELINE configs[4] stand-in: a pggb-SHAPED graph of chr22's size through the reference's own commands.

The HPRC v1.0 pggb chr22 GFA (3.76 M nodes, 5.22 M edges, 464 Mbp, 90 haplotype walks in contig paths; a 402 MB
download) is not in the container.  `panacus-amd synth --shape pggb` writes a graph of that shape (NOT its data):
integer segment names in pangenome order, contig paths per haplotype, inversions, tandem duplications.  Timed, whole
process, page cache warm:
  a. `histgrowth -S -q 0,0.5,1.0 -l 0,1,2`          (test/integrated_test.R:123: "~17 s" for the reference, node)
  b. the same with `-c edge`                        (test/integrated_test.R:136: "~79 s")
  c. `histgrowth -l 1,2,1,1,1 -q 0,0,1,0.5,0.1 -S -a -s haplotypes.txt`   (examples/pangenome_growth_pggb.md:21)
  d. `ordered-histgrowth -c bp -S`, e. `similarity -c bp -S`, f. `table -c node -S --total`   (SURVEY 8f rows on the same file)
and the oracle (serial CPU restatement) on command a, whose table the CLI's must equal byte for byte.
The reference's timings are developer comments on unstated hardware, on the REAL file: orientation, not a baseline."""
import argparse
import json
import math
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as orc  # noqa: E402
from panacus_amd import hostlib as hl  # noqa: E402

CLI = os.path.join(os.path.dirname(os.path.abspath(hl.__file__)), "panacus-amd")


def run(args):
    # a pause first: the previous CLI process left through _exit and the kernel is still reclaiming its 2.4 GB mapping and its
    # GPU context -- a process started right behind it waits for that (0.45 s instead of 0.21 s for the same command)
    time.sleep(1.0)
    t0 = time.perf_counter()
    r = subprocess.run([CLI] + args, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return dt, r.stdout


def body(text):
    return [l.split("\t") for l in text.split("\n") if l and not l.startswith("#")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=3_760_000)
    ap.add_argument("--samples", type=int, default=44)
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        gfa = os.path.join(tmp, "pggb.gfa")
        t, msg = run(["synth", "--shape", "pggb", "--nodes", str(a.nodes), "--samples", str(a.samples), "-o", gfa])
        out = {"graph": msg.strip(), "gfa_bytes": os.path.getsize(gfa), "generate_s": t}
        run(["hist", "-S", gfa])  # warm: page cache, code objects
        grid = ["-S", "-q", "0,0.5,1.0", "-l", "0,1,2"]
        ts = [run(["histgrowth"] + grid + [gfa]) for _ in range(3)]
        out["a_histgrowth_node_s"] = min(x[0] for x in ts)
        table_a = ts[0][1]
        out["b_histgrowth_edge_s"] = min(run(["histgrowth", "-c", "edge"] + grid + [gfa])[0] for _ in range(2))
        names = subprocess.run(["grep", "^P", gfa], capture_output=True, text=True).stdout
        haps = os.path.join(tmp, "haplotypes.txt")
        with open(haps, "w") as f:
            for line in names.split("\n"):
                nm = line.split("\t")[1] if line else ""
                if nm and not nm.startswith(("grch38", "chm13")):
                    f.write(nm + "\n")
        del names
        out["c_example_subset_s"] = min(run(["histgrowth", "-l", "1,2,1,1,1", "-q", "0,0,1,0.5,0.1", "-S", "-a", "-s", haps, gfa])[0]
                                        for _ in range(2))
        # the "next" rows on the same file: ordered growth in file order of the samples, similarity (bp), table --total
        out["d_ordered_histgrowth_bp_s"] = min(run(["ordered-histgrowth", "-c", "bp", "-S", "-l", "1,2", "-q", "0,0.5", gfa])[0] for _ in range(2))
        t_sim, sim_out = run(["similarity", "-c", "bp", "-S", gfa])
        out["e_similarity_bp_s"] = t_sim
        out["e_similarity_rows"] = len(body(sim_out))
        out["f_table_total_node_s"] = run(["table", "-c", "node", "-S", "--total", gfa])[0]
        out["cached_histgrowth_node_s"] = None
        run(["histgrowth", "--cache"] + grid + [gfa])
        out["cached_histgrowth_node_s"] = min(run(["histgrowth", "--cache"] + grid + [gfa])[0] for _ in range(2))
        if not a.no_oracle:
            t0 = time.perf_counter()
            g = orc.Graph(gfa, index_edges=False)
            t_parse = time.perf_counter() - t0
            t0 = time.perf_counter()
            pi, gi, gnames = g.path_order(orc.GROUP_SAMPLE)
            items, pre = g.item_table(orc.NODE)
            cov = orc.coverage(items, pre, pi, gi, g.n_nodes)
            h = orc.hist(cov, len(gnames))
            curves = [orc.growth(h, (orc.ABSOLUTE, c), (orc.RELATIVE, q)) for c, q in ((0, 0.0), (1, 0.5), (2, 1.0))]
            t_rest = time.perf_counter() - t0
            rows = body(table_a)
            same = all([r[1 + k] for r in rows[5:]] == [hl.format_f64(math.floor(x)) for x in curves[k]] for k in range(3))
            out["oracle_cpu"] = {"parse_s": t_parse, "coverage_hist_growth_s": t_rest, "groups": len(gnames),
                                 "steps": int(pre[-1]), "cli_table_identical": bool(same)}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
