#!/bin/bash
# Per-kernel times of one CLI run on the pggb-shaped graph of chr22's size (run on the GPU box): which kernels a real-file
# `histgrowth` runs on the device, and how long each takes.   tools/pggb_kernels.sh [extra CLI options, e.g. "-c edge"]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CLI=$ROOT/panacus_amd/panacus-amd
mkdir -p /tmp/pg; G=/tmp/pg/pggb.gfa
[ -f $G ] || $CLI synth --shape pggb --nodes 3760000 --samples 44 -o $G 2>&1 | tail -1
$CLI hist -S $G > /dev/null
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pk && rocprofv3 --kernel-trace --stats -d /tmp/pk -o t -- $CLI histgrowth ${1:-} -S -q 0,0.5,1.0 -l 0,1,2 $G > /dev/null 2>&1
OUT=${2:-/tmp/pk/kernel_stats.csv}; case $OUT in /*) ;; *) OUT=$ROOT/$OUT;; esac
python3 $ROOT/profiles/summarize_rocprof.py "$(find /tmp/pk -name "*.db" | head -1)" $OUT | head -14 | cut -c1-150
