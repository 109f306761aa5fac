#!/bin/bash
# The whole -m gpu suite in ONE pytest process with the CLI commands in that process (the configuration that aborted whole
# sessions at the end of round 5), every in-process command traced, the crash handler installed (C stack + the tail of the
# captured stderr go to <tag>.crash.txt).  usage: tools/whole_suite_inprocess.sh <tag> [guard] [nocapture]
tag=${1:-a}
out=gpurun_out/whole
mkdir -p $out
export PNX_TEST_CLI_INPROCESS=1 PNX_TRACE_CLI=$out/$tag.cli_trace PANACUS_AMD_CRASH_LOG=$PWD/$out/$tag.crash.txt
[ "$2" = guard ] && export PNX_GUARD_ALLOC=1
S=""; [ "$3" = nocapture ] && S="-s"
rm -f $PNX_TRACE_CLI
timeout 1800 python -m pytest tests/ -x -q -m gpu $S -p no:cacheprovider > $out/$tag.log 2>&1
rc=$?
echo "$tag rc=$rc $(tail -1 $out/$tag.log | cut -c1-160)" | tee -a $out/summary.txt
if [ $rc -ne 0 ]; then
    grep -n "Memory access fault\|Aborted\|core dumped\|terminate\|what():\|Fatal Python\|free():\|malloc\|corrupt" $out/$tag.log | head -5 | tee -a $out/summary.txt
    tail -1 $PNX_TRACE_CLI | tee -a $out/summary.txt
    [ -f $out/$tag.crash.txt ] && head -60 $out/$tag.crash.txt | cut -c1-200 | tee -a $out/summary.txt
fi
