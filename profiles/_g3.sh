set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "(TCP|TCC|TA|TD|SQ|SQC|GRBM)_[A-Za-z0-9_]+" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/r02d/counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/r02d/counters.txt
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  rm -rf /tmp/p_pmc
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_pmc -o pmc -- python $GRAFT_REPO_ROOT/benchmarks/bench_tile_index.py --shapes 10000000x1024 --coarse 8 --probe 16 --reps 3 > /dev/null 2> /tmp/pmc.err
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  python $GRAFT_REPO_ROOT/profiles/summarize_pmc.py "$(find /tmp/p_pmc -name '*.db' | head -1)" $GRAFT_REPO_ROOT/gpurun_out/r02d/k0_pmc_$N.csv > /dev/null 2>> /tmp/pmc.err
  grep -iE "error|invalid|not found|unsupported" /tmp/pmc.err | head -3
  grep -E "tile_index" $GRAFT_REPO_ROOT/gpurun_out/r02d/k0_pmc_$N.csv | sed 's/"void pnx::\|"pnx:://' | cut -c1-120
done
