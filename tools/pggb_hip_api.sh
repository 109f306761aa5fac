#!/bin/bash
# HIP API time of one CLI run on the chr22-shaped graph: where the host side of an upload waits (hipMalloc / hipFree / copies /
# synchronisation).   tools/pggb_hip_api.sh ["-c edge"]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CLI=$ROOT/panacus_amd/panacus-amd
mkdir -p /tmp/pg; G=/tmp/pg/pggb.gfa
[ -f $G ] || $CLI synth --shape pggb --nodes 3760000 --samples 44 -o $G 2>&1 | tail -1
$CLI hist -S $G > /dev/null
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pa && rocprofv3 --hip-runtime-trace --stats -d /tmp/pa -o t -- $CLI histgrowth ${1:-} -S -q 0,0.5,1.0 -l 0,1,2 $G > /dev/null 2>&1
python3 - <<'PY'
import glob, sqlite3
db = sqlite3.connect(glob.glob('/tmp/pa/**/*.db', recursive=True)[0])
cur = db.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "regions" if "regions" in names else None
if not view:
    print(names)
else:
    cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
    agg = {}
    ci = {c: i for i, c in enumerate(cols)}
    for r in cur.execute(f"select * from {view}"):
        a = agg.setdefault(r[ci["name"]], [0, 0])
        a[0] += 1
        a[1] += r[ci["end"]] - r[ci["start"]]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"{n:34s} n={a[0]:5d} total={a[1] / 1e6:9.3f} ms  avg={a[1] / a[0] / 1e3:9.1f} us")
    rows = sorted(cur.execute(f"select * from {view}").fetchall(), key=lambda r: r[ci["start"]])
    t0 = rows[0][ci["start"]]
    print("-- every call of 0.3 ms and more, in order (start ms, duration ms)")
    for r in rows:
        d = r[ci["end"]] - r[ci["start"]]
        if d >= 300000:
            print(f"  {(r[ci['start']] - t0) / 1e6:9.3f}  {d / 1e6:8.3f}  {r[ci['name']]}")
PY
