import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from panacus_amd import capi
c = capi.Context(0)
for n, p, rr in ((2_000_000, 256, 0), (2_000_000, 256, 1), (2_000_000, 1024, 0), (2_000_000, 1024, 1), (10_000_000, 512, 0), (10_000_000, 512, 1), (10_000_000, 1024, 0)):
    (c.set_csr_pansyn_rearranged if rr else c.set_csr_pansyn)(42, n, p)
    o = np.arange(p, dtype=np.uint32)
    c.config(capi.CFG_COVER_ROUTE, 1)
    c.set_order(o, o, p)
    c.hist(want_countable=False)
    c.profile_enable(True); c.profile_reset()
    for _ in range(3):
        c.config(capi.CFG_DROP_DERIVED, 0)
        c.hist(want_countable=False)
    pr = c.profile_read()
    i = c.info()
    print(n, p, rr, i.n_steps, {k: round(v[0]/max(v[1],1),4) for k, v in pr.items() if v[1]}, 'spilled', i.n_spilled_last, i.n_spilled_last / i.n_steps, 'bursts', i.n_spill_bursts_last, 'reruns', i.n_reruns, 'splits', i.band_splits, flush=True)
