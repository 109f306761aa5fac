#!/bin/bash
out=gpurun_out/j2; mkdir -p $out
PNX_TEST_CLI_INPROCESS=1 timeout 900 python -m pytest tests/test_host_cli.py tests/test_gpu_closed_form.py tests/test_gpu_band.py -x -q -m gpu -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$? $(tail -1 $out/tests.log)"
for i in 1 2 3; do for seed in 1 0; do PNX_BAND_SEED=$seed python bench.py --headline-only --no-pmc --steps 200 --warmup 20 > $out/head_${seed}_$i.json 2>/dev/null; python - <<PY
import json
d=json.load(open('$out/head_${seed}_$i.json'))
print('headline seed=$seed', d['ms_per_step'], d['step_breakdown_ms']['band_index'], d['step_breakdown_ms']['band_cover'], d['step_breakdown_ms']['everything_else'])
PY
done; done
for seed in 1 0; do PNX_BAND_SEED=$seed python bench.py --no-pmc --no-cpu-baseline --no-permuted-growth --no-resident --steps 40 --warmup 5 > $out/full_$seed.json 2>/dev/null; python - <<PY
import json
d=json.load(open('$out/full_$seed.json'))
print('full seed=$seed', d['ms_per_step'], {k: (d[k].get('ms_per_step'), d[k].get('step_breakdown_ms', {}).get('band_index')) for k in ('shape_10Mx1k', 'strayed_paths') if k in d})
PY
done
tools/k4_variants.sh
