"""Closed-form growth (hist.rs:89-187) for the bench's three threshold pairs: host threads against everything on the GPU
(pnx_growth_closed_form): a first call, which derives the (n, thresholds) tables, calls that find them, two in flight."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panacus_amd import capi, hostlib  # noqa: E402
from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold  # noqa: E402


def main():
    pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]
    thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs]
    out = {}
    with capi.Context(0) as ctx:
        for n in [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else '64,256,512,1024,2048'.split(','))]:
            rng = np.random.default_rng(n)
            h = rng.integers(1, 10**7, size=n + 1).astype(np.uint64)
            res = {}
            reps = 20 if n <= 512 else 5
            t0 = time.perf_counter()
            for _ in range(max(1, reps // 5)):
                ref = hostlib.calc_growths(h, thr)
            res["host_ms"] = (time.perf_counter() - t0) / max(1, reps // 5) * 1e3
            hostlib.set_quorum_offload(ctx, 1)
            got = hostlib.calc_growths(h, thr)
            res["same"] = all(a.tobytes() == b.tobytes() for a, b in zip(ref, got))
            first = []
            for _ in range(3):
                ctx.config(capi.CFG_DROP_GROWTH_TABLES, 0)
                t0 = time.perf_counter()
                hostlib.calc_growths(h, thr)
                first.append((time.perf_counter() - t0) * 1e3)
            res["device_first_call_ms"] = min(first)   # tables derived: log2 table, running sums, perc_mult, the quorum pair's inner sums
            t0 = time.perf_counter()
            for _ in range(reps):
                hostlib.calc_growths(h, thr)
            res["device_ms"] = (time.perf_counter() - t0) / reps * 1e3
            t0 = time.perf_counter()
            q = [hostlib.calc_growths_begin(h, thr)]
            for _ in range(reps):
                q.append(hostlib.calc_growths_begin(h, thr))
                hostlib.calc_growths_end(q.pop(0))
            hostlib.calc_growths_end(q.pop(0))
            res["device_two_in_flight_ms"] = (time.perf_counter() - t0) / (reps + 1) * 1e3
            hostlib.set_quorum_offload(None)
            out[f"n{n}"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
