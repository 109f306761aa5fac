#!/bin/bash
# Collects the round's measurement evidence on a GPU box (run through gpurun from the repo
# root): bench JSON lines, rocprofv3 kernel-trace summaries and the PMC passes (separate runs,
# --kernel-trace + one counter each).  Results land in gpurun_out/profiles/ and are copied into
# profiles/ by hand.
set -u
R=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profiles
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }

timeout 600 python $REPO/bench.py > $OUT/${R}_hist_cfg3_bench.json 2> $OUT/bench.err
tail -c 600 $OUT/${R}_hist_cfg3_bench.json

rm -rf /tmp/p_stats; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o hist -- \
    python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${R}_hist_cfg3_bench_under_rocprof.json 2> /dev/null
python $REPO/profiles/summarize_rocprof.py "$(db /tmp/p_stats)" $OUT/${R}_hist_cfg3_kernel_stats.csv > /dev/null

for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_pmc; timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_pmc -o pmc -- \
        python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    python $REPO/profiles/summarize_pmc.py "$(db /tmp/p_pmc)" $OUT/${R}_hist_cfg3_pmc_$C.csv > /dev/null
done

timeout 600 python $REPO/bench.py --paths 1024 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${R}_hist_10Mx1024_bench.json 2>/dev/null

timeout 900 python $REPO/benchmarks/bench_ordered_growth.py --reps 2 > $OUT/${R}_growth_cfg4_bench.json 2>/dev/null
rm -rf /tmp/p_g; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_g -o growth -- \
    python $REPO/benchmarks/bench_ordered_growth.py --reps 2 > /dev/null 2>&1
python $REPO/profiles/summarize_rocprof.py "$(db /tmp/p_g)" $OUT/${R}_growth_cfg4_kernel_stats.csv > /dev/null

timeout 900 python $REPO/benchmarks/bench_ordered_growth.py --reps 2 --bp > $OUT/${R}_growth_cfg4_bp_bench.json 2>/dev/null
timeout 600 python $REPO/benchmarks/bench_gfa_end_to_end.py 1000000 64 > $OUT/${R}_gfa_end_to_end.jsonl 2>/dev/null
timeout 900 python $REPO/benchmarks/bench_gfa_end_to_end.py 4000000 128 >> $OUT/${R}_gfa_end_to_end.jsonl 2>/dev/null
timeout 600 python $REPO/benchmarks/bench_run_route.py > $OUT/${R}_run_route_bench.json 2>/dev/null
timeout 900 python $REPO/benchmarks/bench_contig_paths.py > $OUT/${R}_contig_paths_bench.json 2>/dev/null
timeout 600 python $REPO/benchmarks/bench_similarity.py > $OUT/${R}_similarity_cfg4_bench.jsonl 2>/dev/null
timeout 600 python $REPO/benchmarks/bench_similarity.py --bp >> $OUT/${R}_similarity_cfg4_bench.jsonl 2>/dev/null
rm -rf /tmp/p_s; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_s -o sim -- \
    python $REPO/benchmarks/bench_similarity.py --check-nodes 0 > /dev/null 2>&1
python $REPO/profiles/summarize_rocprof.py "$(db /tmp/p_s)" $OUT/${R}_similarity_cfg4_kernel_stats.csv > /dev/null
timeout 600 python $REPO/benchmarks/bench_edge_counts.py > $OUT/${R}_edge_counts_bench.jsonl 2>/dev/null
ls -la $OUT
