#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`)
into the per-kernel summary committed under profiles/ (name, calls, total/avg/min/max ns, %)."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select * from kernels").fetchall()
    ci = {c: i for i, c in enumerate(cols)}
    name_c = "name" if "name" in ci else [c for c in cols if "name" in c][0]
    start_c = "start" if "start" in ci else [c for c in cols if "start" in c][0]
    end_c = "end" if "end" in ci else [c for c in cols if "end" in c][0]
    agg = {}
    for r in rows:
        n = r[ci[name_c]]
        d = r[ci[end_c]] - r[ci[start_c]]
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    lines = ["kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent"]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f'"{n}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{a[2]},{a[3]},{100.0 * a[1] / total:.2f}')
    txt = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
