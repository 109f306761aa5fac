#!/bin/bash
# Round 6's measurement evidence on a GPU box (through gpurun, from the repo root): the driver-contract bench lines, rocprofv3
# kernel-trace summaries and the HBM counter passes (separate runs: --kernel-trace + one counter each), the shard-of-8 budget over
# ALL shards, K4 variants.  Results land in gpurun_out/profiles/ and are copied into profiles/ by hand.
set -u
R=r06
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profiles
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }
timeout 900 python $REPO/bench.py > $OUT/${R}_bench.json 2> $OUT/bench.err
tail -c 300 $OUT/${R}_bench.json
timeout 900 python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${R}_bench_steps20.json 2>/dev/null
rm -rf /tmp/p_stats; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o hist -- \
    python $REPO/bench.py --steps 10 --warmup 2 --headline-only > $OUT/${R}_hist_cfg3_bench_under_rocprof.json 2> /dev/null
python $REPO/profiles/summarize_rocprof.py "$(db /tmp/p_stats)" $OUT/${R}_hist_cfg3_kernel_stats.csv > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_pmc; timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_pmc -o pmc -- \
        python $REPO/bench.py --steps 5 --warmup 1 --headline-only > /dev/null 2>&1
    python $REPO/profiles/summarize_pmc.py "$(db /tmp/p_pmc)" $OUT/${R}_hist_cfg3_pmc_$C.csv > /dev/null
done
rm -rf /tmp/p_1k; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_1k -o k1 -- \
    python $REPO/bench.py --paths 1024 --steps 10 --warmup 2 --headline-only > $OUT/${R}_hist_10Mx1024_bench_under_rocprof.json 2>/dev/null
python $REPO/profiles/summarize_rocprof.py "$(db /tmp/p_1k)" $OUT/${R}_hist_10Mx1024_kernel_stats.csv > /dev/null
timeout 900 python $REPO/benchmarks/bench_ordered_growth.py --reps 5 --warm-full > $OUT/${R}_growth_cfg4_bench.json 2>/dev/null
timeout 900 python $REPO/benchmarks/bench_ordered_growth.py --reps 5 --warm-full --bp > $OUT/${R}_growth_cfg4_bp_bench.json 2>/dev/null
timeout 900 python $REPO/benchmarks/bench_ordered_growth.py --reps 5 --warm-full --pairs 1:0,2:0,1:0.3 > $OUT/${R}_growth_cfg4_q03_bench.json 2>/dev/null
rm -rf /tmp/p_g; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_g -o growth -- \
    python $REPO/benchmarks/bench_ordered_growth.py --reps 2 --warm-full > /dev/null 2>&1
python $REPO/profiles/summarize_rocprof.py "$(db /tmp/p_g)" $OUT/${R}_growth_cfg4_kernel_stats.csv > /dev/null
timeout 1500 python $REPO/benchmarks/bench_shard_of_8.py --all-shards --calls 7 > $OUT/${R}_shard_of_8_budget.json 2> $OUT/shard.err
python - <<PY
import json
d = json.load(open("$OUT/${R}_shard_of_8_budget.json"))
print("shard of 8:", d.get("estimate_all_shards"))
PY
