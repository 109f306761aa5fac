"""k_band_tail on the strayed headline shape (pansyn-v1r, 10 M x 256): kernel times of cold hist calls; run under rocprofv3 --pmc
for the counters of the three kernels of the pass."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from panacus_amd import capi
n, p = 10_000_000, int(os.environ.get("PROBE_PATHS", "256"))
c = capi.Context(0)
c.set_csr_pansyn_rearranged(42, n, p)
o = np.arange(p, dtype=np.uint32)
c.set_order(o, o, p)
c.hist(want_countable=False)
for rep in range(2):
    c.profile_enable(True); c.profile_reset()
    for _ in range(5):
        c.config(capi.CFG_DROP_DERIVED, 0)
        c.hist(want_countable=False)
    pr = c.profile_read()
    i = c.info()
    print({k: round(v[0]/max(v[1],1),4) for k, v in pr.items() if v[1]}, i.n_spilled_last, i.n_spill_bursts_last, i.n_reruns, flush=True)
