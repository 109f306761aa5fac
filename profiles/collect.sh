#!/bin/bash
# Collects the round's measurement evidence on a GPU box (run through gpurun from the repo
# root): bench JSON lines, rocprofv3 kernel-trace summaries and the PMC passes (separate runs,
# --kernel-trace + counters of one block each; no trace domains mixed in).  Results land in
# gpurun_out/profiles/ and are copied into profiles/ by hand.
set -u
R=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profiles
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }
HEAD_ONLY="--headline-only"

# 1. the driver-contract line: headline (cfg3) + permuted_growth (cfg4) + shape_10Mx1k + cpu_baseline
timeout 900 python $REPO/bench.py > $OUT/${R}_bench.json 2> $OUT/bench.err
tail -c 400 $OUT/${R}_bench.json
# the driver's own invocation (20 timed steps)
timeout 900 python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${R}_bench_steps20.json 2>/dev/null

# 2. headline kernels under rocprofv3: per-kernel times, then HBM traffic and SQ counters of K1
rm -rf /tmp/p_stats; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o hist -- \
    python $REPO/bench.py --steps 10 --warmup 2 $HEAD_ONLY > $OUT/${R}_hist_cfg3_bench_under_rocprof.json 2> /dev/null
python $REPO/profiles/summarize_rocprof.py "$(db /tmp/p_stats)" $OUT/${R}_hist_cfg3_kernel_stats.csv > /dev/null
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
         "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    rm -rf /tmp/p_pmc; timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_pmc -o pmc -- \
        python $REPO/bench.py --steps 5 --warmup 1 $HEAD_ONLY > /dev/null 2>&1
    N=$(echo $C | cut -d' ' -f1); [ "$N" = "SQ_WAVES" ] && N=SQ_waves; [ "$N" = "SQ_INSTS_SALU" ] && N=SQ_insts
    python $REPO/profiles/summarize_pmc.py "$(db /tmp/p_pmc)" $OUT/${R}_hist_cfg3_pmc_$N.csv > /dev/null
done

# 3. north_star's 10M x 1k shape as its own headline run (kernel stats under rocprofv3)
rm -rf /tmp/p_1k; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_1k -o k1 -- \
    python $REPO/bench.py --paths 1024 --steps 10 --warmup 2 $HEAD_ONLY > $OUT/${R}_hist_10Mx1024_bench_under_rocprof.json 2>/dev/null
python $REPO/profiles/summarize_rocprof.py "$(db /tmp/p_1k)" $OUT/${R}_hist_10Mx1024_kernel_stats.csv > /dev/null

# 3b. the same shape, node-range share of one of 8 ranks in permuted growth (1.25 M nodes x 512 paths x 128 orders)
timeout 900 python $REPO/benchmarks/bench_ordered_growth.py --reps 5 --warm-full --nodes 1250000 > $OUT/${R}_growth_cfg4_node_shard_of_8_bench.json 2>/dev/null

# 4. cfg4: presence pack (one-shot route with row stores) + permuted growth (K4): kernel stats and counters
timeout 900 python $REPO/benchmarks/bench_ordered_growth.py --reps 5 --warm-full > $OUT/${R}_growth_cfg4_bench.json 2>/dev/null
timeout 900 python $REPO/benchmarks/bench_ordered_growth.py --reps 5 --warm-full --orders 16 > $OUT/${R}_growth_cfg4_R16_bench.json 2>/dev/null
timeout 900 python $REPO/benchmarks/bench_ordered_growth.py --reps 5 --warm-full --bp > $OUT/${R}_growth_cfg4_bp_bench.json 2>/dev/null
rm -rf /tmp/p_g; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_g -o growth -- \
    python $REPO/benchmarks/bench_ordered_growth.py --reps 2 --warm-full > /dev/null 2>&1
python $REPO/profiles/summarize_rocprof.py "$(db /tmp/p_g)" $OUT/${R}_growth_cfg4_kernel_stats.csv > /dev/null
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
         "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    rm -rf /tmp/p_pmc; timeout 900 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_pmc -o pmc -- \
        python $REPO/benchmarks/bench_ordered_growth.py --reps 1 --warm-full > /dev/null 2>&1
    N=$(echo $C | cut -d' ' -f1); [ "$N" = "SQ_WAVES" ] && N=SQ_waves; [ "$N" = "SQ_INSTS_SALU" ] && N=SQ_insts
    python $REPO/profiles/summarize_pmc.py "$(db /tmp/p_pmc)" $OUT/${R}_growth_cfg4_pmc_$N.csv > /dev/null
done

# 4a. the same counters for 16 orders (one rank's share at 8 GPUs)
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
         "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    rm -rf /tmp/p_pmc; timeout 900 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_pmc -o pmc -- \
        python $REPO/benchmarks/bench_ordered_growth.py --reps 1 --warm-full --orders 16 > /dev/null 2>&1
    N=$(echo $C | cut -d' ' -f1); [ "$N" = "SQ_WAVES" ] && N=SQ_waves; [ "$N" = "SQ_INSTS_SALU" ] && N=SQ_insts
    python $REPO/profiles/summarize_pmc.py "$(db /tmp/p_pmc)" $OUT/${R}_growth_cfg4_R16_pmc_$N.csv > /dev/null
done

# 4b. K5 on the matrix cores (similarity, bp): kernel stats and the MFMA / VALU counters
rm -rf /tmp/p_s; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_s -o sim -- \
    python $REPO/benchmarks/bench_similarity.py --bp --check-nodes 0 --reps 3 > /dev/null 2>&1
python $REPO/profiles/summarize_rocprof.py "$(db /tmp/p_s)" $OUT/${R}_similarity_cfg4_kernel_stats.csv > /dev/null
for C in "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
    rm -rf /tmp/p_pmc; timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_pmc -o pmc -- \
        python $REPO/benchmarks/bench_similarity.py --bp --check-nodes 0 --reps 1 > /dev/null 2>&1
    python $REPO/profiles/summarize_pmc.py "$(db /tmp/p_pmc)" $OUT/${R}_similarity_cfg4_pmc_SQ_mfma.csv > /dev/null
done

# 5. the side benches
timeout 300 python $REPO/benchmarks/bench_rows.py --steps 30 > $OUT/${R}_rows_cfg3_bench.json 2>/dev/null
timeout 300 python $REPO/benchmarks/bench_rows.py --paths 1024 --steps 10 --splits 0 > $OUT/${R}_rows_10Mx1024_bench.json 2>/dev/null
timeout 300 python $REPO/benchmarks/bench_closed_form.py > $OUT/${R}_closed_form_bench.json 2>/dev/null
timeout 300 python $REPO/benchmarks/bench_band.py --steps 30 > $OUT/${R}_band_cfg3_bench.json 2>/dev/null
timeout 300 python $REPO/benchmarks/bench_band.py --paths 1024 --steps 10 > $OUT/${R}_band_10Mx1024_bench.json 2>/dev/null
timeout 300 python $REPO/benchmarks/bench_band_fallback.py > $OUT/${R}_band_fallback_bench.json 2>/dev/null
timeout 600 python $REPO/benchmarks/bench_structural_variants.py --check > $OUT/${R}_structural_variants_bench.json 2>/dev/null
timeout 600 python $REPO/benchmarks/bench_shard_of_8.py > $OUT/${R}_shard_of_8_budget.json 2>/dev/null
for N in 256 1024; do
    rm -rf /tmp/p_cf; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_cf -o cf -- python $REPO/benchmarks/bench_closed_form.py $N > /dev/null 2>&1
    python $REPO/profiles/summarize_rocprof.py "$(db /tmp/p_cf)" $OUT/${R}_closed_form_n${N}_kernel_stats.csv > /dev/null
done
timeout 600 python $REPO/benchmarks/bench_gfa_end_to_end.py 1000000 64 > $OUT/${R}_gfa_end_to_end.jsonl 2>/dev/null
timeout 900 python $REPO/benchmarks/bench_gfa_end_to_end.py 4000000 128 >> $OUT/${R}_gfa_end_to_end.jsonl 2>/dev/null
timeout 600 python $REPO/benchmarks/bench_run_route.py > $OUT/${R}_run_route_bench.json 2>/dev/null
timeout 900 python $REPO/benchmarks/bench_contig_paths.py > $OUT/${R}_contig_paths_bench.json 2>/dev/null
timeout 600 python $REPO/benchmarks/bench_similarity.py > $OUT/${R}_similarity_cfg4_bench.jsonl 2>/dev/null
timeout 600 python $REPO/benchmarks/bench_similarity.py --bp >> $OUT/${R}_similarity_cfg4_bench.jsonl 2>/dev/null
timeout 600 python $REPO/benchmarks/bench_edge_counts.py > $OUT/${R}_edge_counts_bench.jsonl 2>/dev/null
timeout 600 python $REPO/benchmarks/bench_subset_cut.py > $OUT/${R}_subset_cut_bench.json 2>/dev/null
timeout 900 python $REPO/benchmarks/bench_pggb_shape.py > $OUT/${R}_pggb_shape_bench.json 2>/dev/null
# 5b. the chr22-shaped graph through the CLI: whole-process times with the host's phases, then its kernels under rocprofv3
timeout 600 bash $REPO/tools/pggb_time.sh > $OUT/${R}_pggb_cli_phases.txt 2>&1
rm -rf /tmp/p_cli; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_cli -o cli -- \
    $REPO/panacus_amd/panacus-amd histgrowth -S -q 0,0.5,1.0 -l 0,1,2 /tmp/pg/pggb.gfa > /dev/null 2>&1
python $REPO/profiles/summarize_rocprof.py "$(db /tmp/p_cli)" $OUT/${R}_pggb_cli_kernel_stats.csv > /dev/null
[ -x $REPO/benchmarks/micro/h2d_rate ] || hipcc --offload-arch=gfx950 -O2 -o $REPO/benchmarks/micro/h2d_rate $REPO/benchmarks/micro/h2d_rate.hip -lpthread 2>/dev/null
timeout 120 $REPO/benchmarks/micro/h2d_rate /tmp/pg/pggb.gfa 16 > $OUT/${R}_h2d_rate.json 2>/dev/null
rm -rf /tmp/pg
ls -la $OUT
