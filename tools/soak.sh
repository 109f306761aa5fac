# Soak of the in-process CLI fuzz: 8 pytest processes side by side on one GPU, 8000 random GFAs each (tests/test_gpu_cli_fuzz.py);
# a process that does not end in "N passed" is copied to gpurun_out/.  PANACUS_AMD_CRASH_LOG=file makes libpanacus_host.so log the
# C stack of a fatal signal (hostlib.load).  Run on the GPU box: gpurun -- bash tools/soak.sh
cd ${GRAFT_REPO_ROOT:-.}
unset PYTHONFAULTHANDLER
unset PANACUS_AMD_CRASH_LOG
rm -f gpurun_out/crash.txt gpurun_out/soakp_*.log
free -g | head -2; nproc; cat /sys/fs/cgroup/memory.max 2>/dev/null
for round in 1 2; do
  pids=""
  for i in 1 2 3 4 5 6 7 8; do
    ( PANACUS_FUZZ_SEEDS=4000 timeout 1200 python -m pytest tests/test_gpu_cli_fuzz.py -q -m gpu -x -p no:cacheprovider -p no:faulthandler > /tmp/soakp_${round}_$i.log 2>&1; echo "exit $?" >> /tmp/soakp_${round}_$i.log ) &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done
  for i in 1 2 3 4 5 6 7 8; do
    if grep -q "exit 0" /tmp/soakp_${round}_$i.log; then tail -2 /tmp/soakp_${round}_$i.log | head -1; else cp /tmp/soakp_${round}_$i.log gpurun_out/soakp_${round}_$i.log; echo "round $round proc $i: PROBLEM"; tail -40 /tmp/soakp_${round}_$i.log | cut -c1-220; fi
  done
done
cat gpurun_out/crash.txt 2>/dev/null | head -80
