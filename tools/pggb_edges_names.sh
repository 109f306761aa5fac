# verdict r3 #4: `-c edge` against `-c node` on the chr22-shaped graph (the L lines parsed on the device), and the same graph
# with segment names that are not numbers (`s12`: the device hashes the name bytes); whole CLI process, best and median of 5
set -u
cd ${GRAFT_REPO_ROOT:-.}
CLI=panacus_amd/panacus-amd
mkdir -p /tmp/pg
G=/tmp/pg/pggb.gfa; N=/tmp/pg/pggb_named.gfa
[ -f $G ] || $CLI synth --shape pggb --nodes 3760000 --samples 44 -o $G 2>&1 | tail -1
[ -f $N ] || $CLI synth --shape pggb --nodes 3760000 --samples 44 --name-prefix s -o $N 2>&1 | tail -1
ls -la $G $N
ARGS="-S -q 0,0.5,1.0 -l 0,1,2"
run() {  # file, count type -> "best median" of 5 runs (seconds)
    python3 - "$@" <<'PY'
import subprocess, sys, time
f, c = sys.argv[1], sys.argv[2]
ts = []
for i in range(6):
    time.sleep(1.0)
    t0 = time.perf_counter()
    subprocess.run(["panacus_amd/panacus-amd", "histgrowth", "-c", c, "-S", "-q", "0,0.5,1.0", "-l", "0,1,2", f], stdout=subprocess.DEVNULL, check=True)
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[1:])
print(f"  -c {c:5s} best {ts[0]:.3f} s  median {ts[len(ts)//2]:.3f} s")
PY
}
for F in $G $N; do
    echo "== $F"
    for c in node bp edge all; do run $F $c; done
done
echo "== phases, -c edge (numbers / names)"
for F in $G $N; do sleep 1; PANACUS_AMD_HOST_TIMING=1 $CLI histgrowth -c edge $ARGS $F 2>&1 >/dev/null | grep "host phase"; done
echo "== same tables as the host's parser"
for F in $G $N; do
    $CLI histgrowth -c all $ARGS $F | grep -v '^#' > /tmp/pg/dev.tsv
    PANACUS_AMD_HOST_PARSE=1 $CLI histgrowth -c all $ARGS $F | grep -v '^#' > /tmp/pg/host.tsv
    cmp /tmp/pg/dev.tsv /tmp/pg/host.tsv && echo "  $F: identical"
done
cmp <($CLI histgrowth -c all $ARGS $G | grep -v '^#') <($CLI histgrowth -c all $ARGS $N | grep -v '^#') && echo "  numbers vs names: identical"
