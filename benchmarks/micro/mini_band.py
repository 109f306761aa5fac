import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import numpy as np
from panacus_amd import capi
import oracle as orc
print("start", flush=True)
c = capi.Context(0)
c.config(capi.CFG_COVER_ROUTE, 1)
n, p = 60000, 8
items, pre, lens = orc.pansyn(5, n, p)
c.set_csr(items.astype(np.uint32), pre, n)
pi = np.arange(p, dtype=np.uint32)
c.set_order(pi, pi, p)
print("hist...", flush=True)
cnt, h = c.hist()
print("ok", h[:5], c.info().n_reruns, flush=True)
