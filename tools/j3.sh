#!/bin/bash
out=gpurun_out/j3; mkdir -p $out
for seed in 1 0; do PNX_BAND_SEED=$seed PNX_BAND_INDEX_TIMING=1 python bench.py --headline-only --no-pmc --steps 8 --warmup 2 2>&1 >/dev/null | grep k_band_index | tail -4 | sed "s/^/seed=$seed /"; done
PNX_TEST_CLI_INPROCESS=1 timeout 900 python -m pytest tests/test_host_cli.py tests/test_gpu_closed_form.py -x -q -m gpu -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$? $(grep -c PASSED $out/tests.log) $(tail -1 $out/tests.log | cut -c1-100)"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "growth or bp or weight" -p no:cacheprovider > $out/tests2.log 2>&1; echo "growth tests rc=$? $(tail -1 $out/tests2.log | cut -c1-100)"
B="python benchmarks/bench_ordered_growth.py --reps 3 --warm-full"
$B --bp 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bp', d['seconds_per_call'], d['growth_kernels_ms_per_call_rank0'])"
$B --bp --pairs 1:0,2:0,1:0.3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bp q03', d['seconds_per_call'], d['growth_kernels_ms_per_call_rank0'])"
