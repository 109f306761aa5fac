#!/bin/bash
# Sanitizer builds of the product libraries + the soak harnesses, into build_san/<kind>/ (kind: plain | asan | tsan).
#   asan: host code of BOTH libraries with -fsanitize=address,undefined (device code not instrumented: -fno-gpu-sanitize),
#         clang's shared runtime (LD_PRELOAD it under Python: tools/run_with_asan.sh)
#   tsan: host code of both libraries with -fsanitize=thread; the soak executables carry the runtime
# hipcc (clang) compiles everything, so one runtime serves both libraries.  usage: tools/build_sanitized.sh <kind>
set -e
kind=${1:-asan}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build_san/$kind
mkdir -p $out
case $kind in
    plain) SAN="" ;;
    asan) SAN="-fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -fno-omit-frame-pointer" ;;
    tsan) SAN="-fsanitize=thread -shared-libsan -fno-omit-frame-pointer -Wno-option-ignored" ;;
    *) echo "kind: plain | asan | tsan"; exit 2 ;;
esac
HIPCC=/opt/rocm/bin/hipcc
CXX=/opt/rocm/lib/llvm/bin/clang++
hip_src="pnx_api pnx_comm pass_pipeline upload_scan kernels_hist kernels_rows kernels_band kernels_gfa kernels_relabel kernels_cut kernels_growth kernels_pairs kernels_pairs_mfma kernels_closed_form pansyn"
pids=()
for s in $hip_src kernels_cover kernels_runs; do
    extra=""; [ $s = kernels_closed_form ] && extra="-ffp-contract=off"
    ( $HIPCC --offload-arch=gfx950 -O2 -g1 -std=c++17 -fPIC $SAN $extra -Wno-unused-result -c $root/panacus_amd/csrc/$s.hip -o $out/$s.o ) &
    pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
objs=""; for s in $hip_src; do objs="$objs $out/$s.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $SAN -o $out/libpanacus_hip.so $objs -ldl
$HIPCC --offload-arch=gfx950 -shared -fPIC $SAN -o $out/libpanacus_hip_steps.so $out/kernels_cover.o $out/kernels_runs.o -L$out -lpanacus_hip -Wl,-rpath,'$ORIGIN'
host_src="thread_pool growth_closed_form gfa_graph tables synth_gfa linkage mini_yaml report commands host_api"
srcs=""; for s in $host_src; do srcs="$srcs $root/panacus_amd/host/$s.cpp"; done
$CXX -O1 -g -std=c++17 -fPIC -shared -pthread -ffp-contract=off $SAN -o $out/libpanacus_host.so $srcs -L$out -lpanacus_hip -Wl,-rpath,'$ORIGIN' -lz -lm
$CXX -O1 -g -std=c++17 -pthread $SAN -o $out/soak_cli $root/tools/soak_cli.cpp -L$out -lpanacus_host -lpanacus_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib
$CXX -O1 -g -std=c++17 -pthread $SAN -I$root/include -o $out/soak_pnx $root/tools/soak_pnx.cpp -L$out -lpanacus_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib
ls -la $out/*.so $out/soak_*
