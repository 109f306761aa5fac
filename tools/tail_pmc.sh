cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCC_HIT_sum"; do
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -o x --output-format csv -- python $GRAFT_REPO_ROOT/benchmarks/micro/tailprobe.py > /dev/null 2>&1
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_band_tail" in k or "k_band_cover" in k:
        a = agg[(k[:40], r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, c), (n, t) in sorted(agg.items()): print(k, c, n, t / n)
PY
done
