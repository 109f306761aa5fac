for flags in "--no-permuted-growth --no-resident --no-shape-1k" "--no-permuted-growth --no-resident" "--no-permuted-growth --no-shape-1k" "--no-resident --no-shape-1k"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-pmc --no-cpu-baseline $flags 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); sp=d['strayed_paths']
        print('$flags', round(d['ms_per_step'],3), d['step_breakdown_ms']['band_index'], 'strayed', round(sp['ms_per_step'],3), round(sp['step_breakdown_ms']['everything_else'],3), round(sp['step_breakdown_ms']['band_index'],4))
"
done
