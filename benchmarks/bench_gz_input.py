"""Loading a compressed GFA: plain gzip (one deflate stream: inflated serially, zlib) against BGZF (bgzip / htslib block
gzip: the blocks are found through their BC extra fields and inflated in parallel over the worker pool) against the
uncompressed file.  Host-only (the GPU box's CPUs); prints one JSON line."""
import gzip
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from panacus_amd import hostlib as hl  # noqa: E402
from test_host_gfa import _write_bgzf  # noqa: E402


def main():
    nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    with tempfile.TemporaryDirectory() as tmp:
        plain = os.path.join(tmp, "g.gfa")
        rc, out, err = hl.run_cli(["synth", "--shape", "pggb", "--nodes", str(nodes), "--samples", "20", "-o", plain])
        assert rc == 0, err
        data = open(plain, "rb").read()
        gz, bg = os.path.join(tmp, "g.gfa.gz"), os.path.join(tmp, "g.bgzf.gfa.gz")
        with gzip.open(gz, "wb", compresslevel=1) as f:
            f.write(data)
        _write_bgzf(data, bg)
        res = {"graph": out.strip(), "bytes": len(data), "gz_bytes": os.path.getsize(gz), "bgzf_bytes": os.path.getsize(bg),
               "cpus": os.cpu_count()}
        for name, f in (("plain", plain), ("gzip", gz), ("bgzf", bg)):
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                g = hl.GfaGraph(f)
                ts.append(time.perf_counter() - t0)
                del g
            res[name + "_load_s"] = min(ts)
        print(json.dumps(res))


if __name__ == "__main__":
    main()
