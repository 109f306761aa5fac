import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from panacus_amd import capi
import oracle as orc
def run(N, P, G, span_f, splits, seed=3, desc=True):
    rng = np.random.default_rng(seed)
    span = max(1, int(N * span_f))
    chunks, off = [], [0]
    for p in range(P):
        a = int(rng.integers(1, N - span + 2))
        ids = a + np.flatnonzero(rng.random(span) < 0.38).astype(np.uint32)
        if desc and p % 7 == 3:
            ids = ids[::-1]
        chunks.append(ids.astype(np.uint32)); off.append(off[-1] + len(ids))
    items = np.concatenate(chunks); pre = np.array(off, dtype=np.uint64)
    order = np.argsort(rng.integers(0, G, size=P), kind="stable").astype(np.uint32)
    grp = np.sort(rng.integers(0, G, size=P)).astype(np.uint32)
    grp = np.unique(grp, return_inverse=True)[1].astype(np.uint32)
    ng = int(grp.max()) + 1
    c = capi.Context(0)
    c.config(capi.CFG_COVER_ROUTE, 1)
    c.set_csr(items, pre, N)
    c.set_order(order, grp, ng)
    if splits: os.environ["PNX_BAND_SPLITS"] = str(splits)
    cnt, h = c.hist()
    os.environ.pop("PNX_BAND_SPLITS", None)
    cov = orc.coverage(items.astype(np.uint64), pre, order.astype(np.uint64), grp.astype(np.uint64), N)
    oh = orc.hist(cov, ng)
    bad = np.flatnonzero(cnt != cov)
    i = c.info()
    print(N, P, ng, span_f, 'splits', i.band_splits, 'spilled', i.n_spilled_last, 'reruns', i.n_reruns, 'rows', i.n_rows, 'bad items', len(bad), bad[:8], cnt[bad[:8]], cov[bad[:8]], 'hist ok', np.array_equal(h, oh), flush=True)
    c.close()
#run(400_000, 400, 20, 0.05, 1)
#run(400_000, 400, 20, 0.05, 3)
#run(400_000, 400, 20, 0.05, 1, desc=False)
#run(400_000, 40, 20, 0.05, 1)
#run(400_000, 400, 400, 0.05, 1)
pass

def debug(N, P, G, span_f, seed=3):
    rng = np.random.default_rng(seed)
    span = max(1, int(N * span_f))
    chunks, off, starts = [], [0], []
    for p in range(P):
        a = int(rng.integers(1, N - span + 2)); starts.append(a)
        ids = a + np.flatnonzero(rng.random(span) < 0.38).astype(np.uint32)
        if p % 7 == 3:
            ids = ids[::-1]
        chunks.append(ids.astype(np.uint32)); off.append(off[-1] + len(ids))
    items = np.concatenate(chunks); pre = np.array(off, dtype=np.uint64)
    order = np.argsort(rng.integers(0, G, size=P), kind="stable").astype(np.uint32)
    grp = np.sort(rng.integers(0, G, size=P)).astype(np.uint32)
    grp = np.unique(grp, return_inverse=True)[1].astype(np.uint32)
    ng = int(grp.max()) + 1
    c = capi.Context(0)
    c.config(capi.CFG_COVER_ROUTE, 1)
    c.set_csr(items, pre, N)
    c.set_order(order, grp, ng)
    for splits in (1, 3):
        os.environ["PNX_BAND_SPLITS"] = str(splits)
        cnt, h = c.hist()
        cov = orc.coverage(items.astype(np.uint64), pre, order.astype(np.uint64), grp.astype(np.uint64), N)
        bad = np.flatnonzero(cnt != cov)
        print('splits', splits, 'bad', len(bad), bad[:3], bad[-3:] if len(bad) else None, flush=True)
    bits = c.presence()
    os.environ.pop("PNX_BAND_SPLITS", None)
    rows = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :N + 1]
    # oracle presence per group
    want = np.zeros((ng, N + 1), dtype=np.uint8)
    for k in range(P):
        p = int(order[k]); want[int(grp[k]), items[int(pre[p]):int(pre[p + 1])]] = 1
    diff = np.argwhere(rows != want)
    print('presence diffs', len(diff))
    if len(diff):
        gs = np.unique(diff[:, 0])
        for g in gs[:5]:
            d = diff[diff[:, 0] == g][:, 1]
            print(' group', g, 'extra/missing items', len(d), d.min(), d.max(), 'got', rows[g, d[0]], 'want', want[g, d[0]])
            ks = np.flatnonzero(grp == g)
            print('   entries of group', ks.min(), ks.max(), 'paths', order[ks][:8], 'starts', [starts[int(order[k])] for k in ks[:8]])
            # which entry (any group) has a path covering d.min()
            who = [k for k in range(P) if starts[int(order[k])] <= d.min() < starts[int(order[k])] + span and (d.min() in set(items[int(pre[int(order[k])]):int(pre[int(order[k])+1])].tolist()[:0]) or True)]
            print('   entries whose path spans it', who[:20], [int(grp[k]) for k in who[:20]])
debug(4_000_000, 4000, 80, 0.05)
