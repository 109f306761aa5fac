#!/usr/bin/env python3
"""Timeline of the kernels of the last timed steps of a rocprofv3 --kernel-trace run of bench.py --headline-only
(rocpd SQLite database): start / end of every kernel relative to the step's first kernel, and the gaps between them."""
import sqlite3
import sys


def main(db_path, n_steps=3):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ci = {c: i for i, c in enumerate(cols)}
    name_c = "name" if "name" in ci else [c for c in cols if "name" in c][0]
    start_c = "start" if "start" in ci else [c for c in cols if "start" in c][0]
    end_c = "end" if "end" in ci else [c for c in cols if "end" in c][0]
    extra = [c for c in cols if c in ("queue_id", "stream_id", "queue", "stream")]
    rows = sorted(cur.execute("select * from kernels").fetchall(), key=lambda r: r[ci[start_c]])
    idx = [k for k, r in enumerate(rows) if "k_band_index" in r[ci[name_c]]]
    for s in idx[-int(n_steps) - 1:-1]:
        e = idx[idx.index(s) + 1]
        t0 = rows[s][ci[start_c]]
        print("step")
        for r in rows[s:e]:
            nm = r[ci[name_c]].split("(")[0].replace("void ", "").replace("pnx::", "")[:40]
            print(f"  {nm:40s} start {(r[ci[start_c]] - t0) / 1e3:9.2f} us  end {(r[ci[end_c]] - t0) / 1e3:9.2f} us  dur {(r[ci[end_c]] - r[ci[start_c]]) / 1e3:8.2f}"
                  + "".join(f"  {c}={r[ci[c]]}" for c in extra))
        print(f"  next step's index starts at {(rows[e][ci[start_c]] - t0) / 1e3:9.2f} us")


if __name__ == "__main__":
    main(*sys.argv[1:3])
