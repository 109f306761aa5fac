#!/bin/bash
# Kernel timeline of a pipelined loop: which kernel ran when, on which hardware queue and stream.
#   tools/step_timeline.sh [overheads|bench] [percentile of the long k_rows_cover launches to start the window at] [window in us]
# Run on the GPU box (gpurun); GPU_MAX_HW_QUEUES is passed through.
WHAT=${1:-overheads}; PCT=${2:-0.93}; WIN=${3:-600}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "$WHAT" = bench ]; then CMD="python $ROOT/bench.py --no-permuted-growth --no-cpu-baseline --no-pmc --no-shape-1k --no-cold --depth 4"
else CMD="python $ROOT/benchmarks/bench_step_overheads.py"; fi
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl && rocprofv3 --kernel-trace -d /tmp/tl -o t -- $CMD > /tmp/tl.out 2>&1
tail -1 /tmp/tl.out | cut -c1-300
PCT=$PCT WIN=$WIN python3 - <<'PY'
import sqlite3, glob, os
db = sqlite3.connect(glob.glob('/tmp/tl/**/*.db', recursive=True)[0])
rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
covers = [r for r in rows if 'k_rows_cover' in r[0] and r[2] - r[1] > 50000]
t0 = covers[int(len(covers) * float(os.environ['PCT']))][1]
win = float(os.environ['WIN']) * 1e3
for name, s, e, q, st in rows:
    if t0 <= s < t0 + win:
        print(f"{(s-t0)/1e3:8.1f} us +{(e-s)/1e3:7.1f}  q{q} s{st}  {name.split('(')[0][:48]}")
PY
