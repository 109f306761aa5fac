#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd SQLite database."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, counter_name, value, duration from counters_collection").fetchall()
    agg = {}
    for k, c, v, d in rows:
        a = agg.setdefault((k.split("(")[0], c), [0, 0.0, 0])
        a[0] += 1
        a[1] += v
        a[2] += d
    lines = ["kernel,counter,launches,avg_value,avg_duration_ns"]
    for (k, c), a in sorted(agg.items()):
        lines.append(f'"{k}",{c},{a[0]},{a[1] / a[0]:.3f},{a[2] / a[0]:.1f}')
    txt = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
