#!/bin/bash
# rocprofv3 --pmc passes over the permuted-growth benchmark; prints the growth kernel's averages
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in "$@"; do
    rm -rf /tmp/p_pmc
    timeout 900 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_pmc -o pmc -- \
        python $REPO/benchmarks/bench_ordered_growth.py --reps 1 > /dev/null 2>&1
    python $REPO/profiles/summarize_pmc.py "$(find /tmp/p_pmc -name '*.db' | head -1)" 2>/dev/null | grep -E "k_growth" | sed 's/"void pnx::\|"pnx:://; s/<[^>]*>//' | cut -c1-110
done
