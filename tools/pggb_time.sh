# whole-process time of the reference's example commands on a pggb-shaped graph of chr22's size, with the host's phases
set -u
cd ${GRAFT_REPO_ROOT:-.}
CLI=panacus_amd/panacus-amd
mkdir -p /tmp/pg; G=/tmp/pg/pggb.gfa
[ -f $G ] || $CLI synth --shape pggb --nodes 3760000 --samples 44 -o $G 2>&1 | tail -1
ls -la $G
$CLI hist -S $G > /dev/null   # warm
TIMEFORMAT="wall %R s"
# (a pause between the runs: a process that leaves through _exit hands its 2.4 GB mapping and its GPU context to the kernel
# to clean up, and a run started right behind it waits for that -- 0.45 s instead of 0.21 s back to back)
for i in 1 2 3; do sleep 1; time $CLI histgrowth -S -q 0,0.5,1.0 -l 0,1,2 $G > /tmp/pg/a.tsv; done
PANACUS_AMD_HOST_TIMING=1 $CLI histgrowth -S -q 0,0.5,1.0 -l 0,1,2 $G 2>&1 >/dev/null | grep "host phase\|host growth"
echo "--- host parser (PANACUS_AMD_HOST_PARSE=1)"
for i in 1 2; do time PANACUS_AMD_HOST_PARSE=1 $CLI histgrowth -S -q 0,0.5,1.0 -l 0,1,2 $G > /tmp/pg/b.tsv; done
cmp /tmp/pg/a.tsv /tmp/pg/b.tsv && echo "same table"
echo "--- bp, all"
sleep 1
time $CLI histgrowth -c bp -S -q 0,0.5,1.0 -l 0,1,2 $G > /dev/null
sleep 1; time $CLI histgrowth -c all -S -q 0,0.5,1.0 -l 0,1,2 $G > /dev/null
echo "--- edge"
sleep 1
time $CLI histgrowth -c edge -S -q 0,0.5,1.0 -l 0,1,2 $G > /dev/null
