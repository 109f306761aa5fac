"""-s / -e interval lists: the walks cut on the device (pnx_set_csr_cut) against the host walk.

Two measurements on one generated graph (pansyn-shaped GFA, `panacus-amd synth`):
  * whole call from the parsed GFA image: GraphStorage::masked_table (the host walk the CLI used in round 1:
    serial over the paths for bp counts, one thread per path otherwise) vs walk_cut + pnx_set_csr_cut +
    the replay of the partial pieces (what the CLI does now); both include the step parse of the text;
  * the library call alone on arrays already in host memory: pnx_set_csr_cut vs a plain pnx_set_csr of the
    same steps (i.e. what the cut costs on top of an upload).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panacus_amd import capi, hostlib as hl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=2_000_000)
    ap.add_argument("--paths", type=int, default=64)
    ap.add_argument("--intervals", type=int, default=200)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        gfa = os.path.join(tmp, "g.gfa")
        rc, _, err = hl.run_cli(["synth", "--nodes", str(a.nodes), "--paths", str(a.paths), "--links", "-o", gfa])
        assert rc == 0, err
        t0 = time.perf_counter()
        hg = hl.GfaGraph(gfa, index_edges=True)
        t_parse = time.perf_counter() - t0
        names = [n.split(":")[0] for n in hg.path_names()]
        items, pre = hg.item_table(hl.NODE)
        lens = hg.node_lens
        bp = [int(lens[items[int(pre[k]):int(pre[k + 1])]].sum(dtype=np.uint64)) for k in range(len(names))]
        rng = np.random.default_rng(3)

        def bed(fn, rows):
            with open(fn, "w") as f:
                for _ in range(rows):
                    k = int(rng.integers(0, len(names)))
                    lo = int(rng.integers(0, bp[k]))
                    f.write(f"{names[k]}\t{lo}\t{lo + int(rng.integers(1, bp[k] // 4))}\n")
        sf, ef = os.path.join(tmp, "s.bed"), os.path.join(tmp, "e.bed")
        bed(sf, a.intervals)
        bed(ef, a.intervals // 4)
        out = {"nodes": a.nodes, "paths": a.paths, "steps": int(pre[-1]), "gfa_bytes": os.path.getsize(gfa),
               "gfa_parse_s": t_parse, "intervals": [a.intervals, a.intervals // 4]}
        with capi.Context() as ctx:
            hg.cut_upload(ctx, hl.NODE, sf, ef)  # warm: code objects, buffers
            for name, ct in (("node", hl.NODE), ("bp", hl.BP), ("edge", hl.EDGE)):
                t0 = time.perf_counter()
                h_items, h_pre, h_fl, h_ids, h_bps = hg.masked_table(ct, sf, ef)
                t_host = time.perf_counter() - t0
                t0 = time.perf_counter()
                uid, ub = hg.cut_upload(ctx, ct, sf, ef)
                t_dev = time.perf_counter() - t0
                got, off, _ = ctx.get_csr()
                same = bool(np.array_equal(off, h_pre) and np.array_equal(got, h_items) and np.array_equal(uid, h_ids)
                            and np.array_equal(ub, h_bps) and np.array_equal(ctx.get_exclude()[1:], h_fl[1:]))
                # the host walk still has to upload its table
                t0 = time.perf_counter()
                ctx.set_csr(h_items, h_pre, len(h_fl) - 1, weights=lens if ct == hl.BP else None, exclude=h_fl)
                t_up = time.perf_counter() - t0
                out[name] = {"host_walk_s": t_host, "host_walk_upload_s": t_up, "device_cut_whole_call_s": t_dev,
                             "kept_steps": int(h_pre[-1]), "uncovered_nodes": int(len(uid)), "identical": same}
            # the library call alone
            P = len(names)
            mode = np.full(P, capi.WALK_CUT, dtype=np.uint8)
            inc = [[] for _ in range(P)]
            for _ in range(a.intervals):
                k = int(rng.integers(0, P))
                lo = int(rng.integers(0, bp[k]))
                inc[k].append((lo, lo + int(rng.integers(1, bp[k] // 4))))

            def joined(v):
                o = []
                for s, e in sorted(v):
                    if o and o[-1][1] >= s:
                        o[-1][1] = max(o[-1][1], e)
                    else:
                        o.append([s, e])
                return [tuple(x) for x in o]
            inc = [joined(v) for v in inc]
            for k in range(P):
                if not inc[k]:
                    mode[k] = capi.WALK_SKIP
            back = np.zeros(len(items), dtype=np.uint8)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.set_csr_cut(items, pre, lens, mode, inc, None, None, back, count_type=1, weights=lens, track_covered=True)
                ts.append(time.perf_counter() - t0)
            kept = int(ctx.info().n_steps)
            tp = []
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.set_csr(items, pre, len(lens) - 1, weights=lens)
                tp.append(time.perf_counter() - t0)
            out["library_call_alone"] = {"pnx_set_csr_cut_s": min(ts), "plain_pnx_set_csr_s": min(tp),
                                         "steps_in": int(pre[-1]), "steps_kept": kept}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
