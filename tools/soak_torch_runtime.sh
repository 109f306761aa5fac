#!/bin/bash
# The soak harnesses against the HIP / HSA runtime that ships inside the torch wheel (ROCm 7.0.2) instead of /opt/rocm's
# (7.2.0): a pytest process imports torch first, so every library of the product resolves libamdhip64.so.7 and
# libhsa-runtime64.so.1 to torch's copies -- the in-process CLI tests run on THAT runtime, the panacus-amd binary on /opt/rocm's.
out=gpurun_out/soak_torch; mkdir -p $out
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
W=/tmp/soak_work_t; rm -rf $W; mkdir -p $W
export PANACUS_AMD_CRASH_LOG=$PWD/$out/crash.txt
run() { name=$1; to=$2; shift 2; LD_LIBRARY_PATH=$TL:$LD_LIBRARY_PATH timeout $to "$@" > $out/$name.log 2>&1; echo "$name rc=$? $(tail -1 $out/$name.log | cut -c1-120)" | tee -a $out/summary.txt; }
LD_LIBRARY_PATH=$TL ldd build_san/product/soak_cli | grep -i "amdhip\|hsa" | tee $out/which_runtime.txt
run torchrt_cli_1t 1200 build_san/product/soak_cli $W/p1 ${1:-60} 1 tests/golden
run torchrt_pnx_2t 600 build_san/product/soak_pnx ${2:-3000} 2
run torchrt_cli_3t 1200 build_san/product/soak_cli $W/p3 ${1:-60} 3 tests/golden
