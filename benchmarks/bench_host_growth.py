#!/usr/bin/env python3
"""Host closed-form growth (a7) wall time vs thread count, for the record in DESIGN.md."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panacus_amd import hostlib  # noqa: E402
from panacus_amd.thresholds import ABSOLUTE, RELATIVE, Threshold  # noqa: E402

rng = np.random.default_rng(1)
pairs = [(1, 0.0), (2, 0.0), (1, 0.5)]
thr = [(Threshold(ABSOLUTE, c), Threshold(RELATIVE, q)) for c, q in pairs]
for n in [int(a) for a in sys.argv[1:]] or [256, 1024]:
    h = rng.integers(1, 10**5, size=n + 1).astype(np.uint64)
    for t in (1, 8, 32, 64, 128, 256, 0):
        hostlib.calc_growths(h, thr, t)
        reps = 3 if t != 1 else 1
        t0 = time.perf_counter()
        for _ in range(reps):
            hostlib.calc_growths(h, thr, t)
        print(f"n={n} threads={t}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms", flush=True)
