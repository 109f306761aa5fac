#!/bin/bash
# The soak harnesses on a GPU box: product libraries, then ASan+UBSan, then TSan builds (tools/build_sanitized.sh, built in
# the CPU container; the binaries travel).  Logs under gpurun_out/soak/.  usage: tools/soak_gpu.sh [iters_pnx] [iters_cli]
NP=${1:-1500}; NC=${2:-20}
out=gpurun_out/soak; mkdir -p $out
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux
export LD_LIBRARY_PATH=$RT:$LD_LIBRARY_PATH
W=/tmp/soak_work; rm -rf $W; mkdir -p $W
run() {  # name, timeout, command...
    name=$1; to=$2; shift 2
    timeout $to "$@" > $out/$name.log 2>&1
    echo "$name rc=$? $(grep -c 'WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error' $out/$name.log) sanitizer reports; $(tail -1 $out/$name.log | cut -c1-120)" | tee -a $out/summary.txt
}
run product_pnx_2t 600 build_san/product/soak_pnx $NP 2
run product_pnx_4t_keep3 600 build_san/product/soak_pnx $NP 4 3
run product_cli_1t 900 build_san/product/soak_cli $W/p1 $NC 1 tests/golden
run product_cli_3t 900 build_san/product/soak_cli $W/p3 $NC 3 tests/golden
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1
run asan_pnx_2t 900 build_san/asan/soak_pnx $((NP / 3)) 2
run asan_cli_1t 900 build_san/asan/soak_cli $W/a1 $((NC / 2)) 1 tests/golden
run asan_cli_3t 900 build_san/asan/soak_cli $W/a3 $((NC / 2)) 3 tests/golden
export TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 history_size=4 suppressions=$PWD/tools/tsan.supp"
run tsan_pnx_2t 900 build_san/tsan/soak_pnx $((NP / 5)) 2
run tsan_cli_1t 900 build_san/tsan/soak_cli $W/t1 $((NC / 4)) 1 tests/golden
run tsan_cli_3t 900 build_san/tsan/soak_cli $W/t3 $((NC / 4)) 3 tests/golden
for f in $out/asan_*.log $out/tsan_*.log; do grep -A25 "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error" $f | head -150 > $f.first_reports; done
