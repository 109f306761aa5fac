set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02a/pytest.log
( time python bench.py ) > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/p_pmc
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_pmc -o pmc -- python $GRAFT_REPO_ROOT/benchmarks/bench_ordered_growth.py --reps 1 > /dev/null 2> /tmp/pmc.err
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  python $GRAFT_REPO_ROOT/profiles/summarize_pmc.py "$(find /tmp/p_pmc -name '*.db' | head -1)" $GRAFT_REPO_ROOT/gpurun_out/r02a/growth_pmc_$N.csv > /dev/null 2>> /tmp/pmc.err
  tail -3 /tmp/pmc.err >> $GRAFT_REPO_ROOT/gpurun_out/r02a/pmc.err
done
cat $GRAFT_REPO_ROOT/gpurun_out/r02a/pytest.log
tail -c 3000 $GRAFT_REPO_ROOT/gpurun_out/r02a/bench.json
tail -5 $GRAFT_REPO_ROOT/gpurun_out/r02a/bench.err
