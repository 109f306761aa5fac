#!/bin/bash
# Round 5's last tree (ab_r05/, built from commit 4be75bb) through whole in-process -m gpu sessions on today's boxes: does the
# intermittent abort of round 5 still show there?  usage: tools/r05_sessions.sh <n>
out=$PWD/gpurun_out/r05tree; mkdir -p $out
cd ab_r05
for i in $(seq 1 ${1:-4}); do
    PNX_TEST_CLI_INPROCESS=1 PANACUS_AMD_CRASH_LOG=$out/s$i.crash.txt timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $out/s$i.log 2>&1
    echo "r05tree s$i rc=$? $(tail -1 $out/s$i.log | cut -c1-120)" | tee -a $out/summary.txt
done
