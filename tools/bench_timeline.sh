#!/bin/bash
# Kernel timeline of the last timed steps of `bench.py --headline-only` (run on the GPU box).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS=${1:-"--steps 8 --warmup 2"}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl && rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $ROOT/bench.py --headline-only $ARGS > /tmp/tl.out 2>&1
tail -1 /tmp/tl.out | cut -c1-200
python3 - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/tl/**/*.db', recursive=True)[0])
rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
band = [i for i, r in enumerate(rows) if 'k_band_index' in r[0]]
i0 = band[-3] if len(band) >= 3 else 0
t0 = rows[i0][1]
for name, s, e, q, st in rows[i0:]:
    print(f"{(s-t0)/1e3:9.1f} us +{(e-s)/1e3:7.1f}  q{q} s{st}  {name.split('(')[0][:60]}")
PY
