cd ${GRAFT_REPO_ROOT:-.}
CLI=panacus_amd/panacus-amd
mkdir -p /tmp/pg; G=/tmp/pg/pggb.gfa
[ -f $G ] || $CLI synth --shape pggb --nodes 3760000 --samples 44 -o $G 2>&1 | tail -1
$CLI hist -S $G > /dev/null
for c in edge all; do echo "== -c $c"; sleep 1; PANACUS_AMD_HOST_TIMING=1 $CLI histgrowth -c $c -S -q 0,0.5,1.0 -l 0,1,2 $G 2>&1 >/dev/null | grep "host phase"; done
