#!/bin/bash
# Development loop for the one-shot band route (gpurun from the repo root): the band bench, its rocprofv3 kernel-trace
# summary and the counter passes (separate runs, one counter group each, --kernel-trace only).
set -u
TAG=${1:-dev}
ARGS=${2:-"--steps 20"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/band_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }
timeout 300 python $REPO/benchmarks/bench_band.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json; tail -3 $OUT/bench.err
rm -rf /tmp/p_stats; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o band -- \
    python $REPO/benchmarks/bench_band.py $ARGS > /dev/null 2>&1
python $REPO/profiles/summarize_rocprof.py "$(db /tmp/p_stats)" $OUT/kernel_stats.csv > /dev/null
head -8 $OUT/kernel_stats.csv | cut -c1-200
if [ "${3:-pmc}" = "pmc" ]; then
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
         "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    rm -rf /tmp/p_pmc; timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_pmc -o pmc -- \
        python $REPO/benchmarks/bench_band.py --steps 3 > /dev/null 2>&1
    N=$(echo $C | cut -d' ' -f1); [ "$N" = "SQ_WAVES" ] && N=SQ_waves; [ "$N" = "SQ_INSTS_SALU" ] && N=SQ_insts
    python $REPO/profiles/summarize_pmc.py "$(db /tmp/p_pmc)" $OUT/pmc_$N.csv > /dev/null
    grep -E "k_band|kernel" $OUT/pmc_$N.csv | cut -c1-300
done
fi
