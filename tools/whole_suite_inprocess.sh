#!/bin/bash
# The whole -m gpu suite in ONE pytest process with the CLI commands in that process (the configuration that aborted whole
# sessions at the end of round 5), output not captured, every in-process command traced.  usage: tools/whole_suite_inprocess.sh <tag> [guard]
tag=${1:-a}
out=gpurun_out/whole
mkdir -p $out
export PNX_TEST_CLI_INPROCESS=1 PNX_TRACE_CLI=$out/$tag.cli_trace PANACUS_AMD_CRASH_LOG=$PWD/$out/$tag.crash.txt
[ "$2" = guard ] && export PNX_GUARD_ALLOC=1
rm -f $PNX_TRACE_CLI
timeout 1800 python -m pytest tests/ -x -q -m gpu -s -p no:cacheprovider > $out/$tag.log 2>&1
rc=$?
echo "$tag rc=$rc $(tail -1 $out/$tag.log | cut -c1-160)" | tee -a $out/summary.txt
if [ $rc -ne 0 ]; then
    grep -n "Memory access fault\|Aborted\|core dumped\|terminate\|what():\|Fatal Python\|free():\|malloc\|corrupt" $out/$tag.log | head -5 | tee -a $out/summary.txt
    tail -1 $PNX_TRACE_CLI | tee -a $out/summary.txt
fi
