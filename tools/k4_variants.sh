#!/bin/bash
# K4 (permuted growth, cfg4: 10 M x 512, 128 orders): node, bp with several event-queue lengths, q = 0.3
out=gpurun_out/k4v; mkdir -p $out
B="python benchmarks/bench_ordered_growth.py --reps 3 --warm-full"
$B > $out/node.json 2>$out/node.err
for q in 128 256 512 1024; do PNX_GROWTH_EVQ=$q $B --bp > $out/bp_evq$q.json 2>$out/bp_evq$q.err; done
$B --pairs 1:0,2:0,1:0.3 > $out/node_q03.json 2>$out/node_q03.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/k4v/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], {k: d[k] for k in d if 'ms' in k or 'seconds' in k or k in ('value',)})
    except Exception as e:
        print(f, 'ERR', e)
PY
