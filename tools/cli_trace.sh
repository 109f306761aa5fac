cd ${GRAFT_REPO_ROOT:-.}
mkdir -p /tmp/pg; G=/tmp/pg/pggb.gfa
[ -f $G ] || panacus_amd/panacus-amd synth --shape pggb --nodes 3760000 --samples 44 -o $G 2>&1 | tail -1
panacus_amd/panacus-amd hist -S $G > /dev/null
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ht && rocprofv3 --hip-trace --kernel-trace --stats -d /tmp/ht -o t -- $GRAFT_REPO_ROOT/panacus_amd/panacus-amd histgrowth -S -q 0,0.5,1.0 -l 0,1,2 $G > /dev/null 2>&1
python3 - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/ht/**/*.db', recursive=True)[0])
rows = db.execute("select name, start, end from regions order by start").fetchall()
t0 = rows[0][1]
print("first", len(rows))
for name, s, e in rows:
    if (e - s) > 2e6: print(f"{(s-t0)/1e6:9.2f} ms  +{(e-s)/1e6:8.2f}  {name}")
k = db.execute("select name, start, end from kernels order by start").fetchall()
for name, s, e in k[:40]:
    print(f"{(s-t0)/1e6:9.2f} ms  +{(e-s)/1e6:8.3f}  K {name[:60]}")
PY
