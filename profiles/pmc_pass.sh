#!/bin/bash
# one rocprofv3 --pmc pass per counter group over a short bench run; prints K1's averages
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in "$@"; do
    rm -rf /tmp/p_pmc
    timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_pmc -o pmc -- \
        python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    python $REPO/profiles/summarize_pmc.py "$(find /tmp/p_pmc -name '*.db' | head -1)" 2>/dev/null | grep -E "tile_cover|tile_index_fine|k_hist" | sed 's/"void pnx::\|"pnx:://; s/<[^>]*>//' | cut -c1-110
done
