"""Row / column order of the `similarity` table: Similarity::set_table after the Jaccard
table (src/analyses/similarity.rs:166-217) = f32 Euclidean row distances, kodama::linkage, observations
in merge order, sort_by_indices.

kodama (crate 0.3.0) is a third-party dependency that is not in the reference tree, and the reference
holds no similarity output: PARITY UNPINNED.  What pins the restatements instead:
  * SciPy's scipy.cluster.hierarchy.linkage -- an independent implementation of the same published
    algorithms (Muellner's fastcluster family: MST / NN-chain / generic, SciPy cluster labels, smaller
    label first) -- gives the same merge steps on tie-free inputs, for all seven methods;
  * a literal Python transcription of the reference's own glue (get_order_from_dendrogram,
    enumerate/sort_by_key, sort_by_indices) applied to those steps gives the same permutation;
  * the oracle (C, literal) and the host (C++, its own structure: row-minimum cache, vectors) agree
    bit for bit on hundreds of random tables including tied ones.
"""
import numpy as np
import pytest

import oracle as orc
from panacus_amd import hostlib as hl

METHODS = orc.CLUSTER_METHODS


def _condensed(table):
    """calculate_distances + euclidean (similarity.rs:238-254) in f32, sequential sums"""
    t = np.asarray(table, dtype=np.float32)
    n = t.shape[0]
    out = []
    for r in range(n - 1):
        for c in range(r + 1, n):
            s = np.float32(0.0)
            for k in range(n):
                d = np.float32(t[r, k] - t[c, k])
                s = np.float32(s + np.float32(d * d))
            out.append(np.sqrt(s, dtype=np.float32))
    return np.asarray(out, dtype=np.float32)


def _reference_glue(steps_c1, steps_c2, n):
    """similarity.rs:169-179 and :194-217, transcribed literally; returns the list 0..n after
    sort_by_indices = which input group sits in every row"""
    indices = []
    for a, b in zip(steps_c1, steps_c2):       # get_order_from_dendrogram
        if a < n:
            indices.append(int(a))
        if b < n:
            indices.append(int(b))
    order = list(enumerate(indices))
    order.sort(key=lambda el: el[1])           # sort_by_key(|el| el.1) (stable)
    order = [el[0] for el in order]
    lst = list(range(n))
    idx = list(order)
    for i in range(len(idx)):                  # sort_by_indices
        while i != idx[i]:
            new_i = idx[i]
            idx[i], idx[new_i] = idx[new_i], idx[i]
            lst[i], lst[new_i] = lst[new_i], lst[i]
    return lst


def _random_jaccard(rng, n, n_items=400, dup=False):
    """a Jaccard table of n random item sets (what the command feeds the clustering)"""
    sets = rng.random((n, n_items)) < rng.uniform(0.15, 0.85, size=(n, 1))
    sets[:, 0] = True
    if dup and n >= 4:
        sets[n - 1] = sets[0]
        sets[n - 2] = sets[1]
    inter = (sets[:, None, :] & sets[None, :, :]).sum(-1).astype(np.uint64)
    lens = np.diag(inter)
    return (inter.astype(np.float32) / (lens[:, None] + lens[None, :] - inter).astype(np.float32)).astype(np.float32)


@pytest.mark.parametrize("method", METHODS)
def test_host_linkage_equals_scipy(method):
    from scipy.cluster.hierarchy import linkage as sp_linkage
    rng = np.random.default_rng(101 + METHODS.index(method))
    for trial in range(40):
        n = int(rng.integers(2, 28))
        pts = rng.normal(size=(n, 6)) * rng.uniform(0.5, 3.0)
        cond = np.asarray([np.linalg.norm(pts[i] - pts[j]) for i in range(n - 1) for j in range(i + 1, n)], dtype=np.float32)
        z = sp_linkage(cond.astype(np.float64), method=method)
        c1, c2, d = hl.linkage(cond, n, method)
        assert np.array_equal(c1, z[:, 0].astype(np.uint64)) and np.array_equal(c2, z[:, 1].astype(np.uint64)), (method, n, trial)
        assert np.allclose(d, z[:, 2], rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize("method", METHODS)
def test_order_is_the_reference_glue_on_scipy_steps(method):
    from scipy.cluster.hierarchy import linkage as sp_linkage
    rng = np.random.default_rng(7 + METHODS.index(method))
    for trial in range(25):
        n = int(rng.integers(2, 20))
        table = _random_jaccard(rng, n)
        cond = _condensed(table)
        z = sp_linkage(cond.astype(np.float64), method=method)
        want = _reference_glue(z[:, 0].astype(int), z[:, 1].astype(int), n)
        perm = hl.similarity_order(table, method)
        assert perm.tolist() == want, (method, n, trial)
        t2, perm_o = orc.similarity_order(table, method)
        assert perm_o.tolist() == want
        assert np.array_equal(t2, table[np.ix_(want, want)])


def test_host_equals_oracle_on_random_tables_with_ties():
    rng = np.random.default_rng(5)
    for trial in range(150):
        n = int(rng.integers(1, 40))
        table = _random_jaccard(rng, n, n_items=int(rng.integers(8, 300)), dup=bool(trial % 3 == 0))
        for method in METHODS:
            t2, perm_o = orc.similarity_order(table, method)
            perm_h = hl.similarity_order(table, method)
            assert perm_h.tolist() == perm_o.tolist(), (trial, n, method)
            assert sorted(perm_h.tolist()) == list(range(n))
            assert np.array_equal(t2, table[np.ix_(perm_o.astype(int), perm_o.astype(int))])


def test_small_cases():
    one = np.ones((1, 1), dtype=np.float32)
    for method in METHODS:
        assert hl.similarity_order(one, method).tolist() == [0]
        assert orc.similarity_order(one, method)[1].tolist() == [0]
    two = np.array([[1, 0.25], [0.25, 1]], dtype=np.float32)
    for method in METHODS:
        assert hl.similarity_order(two, method).tolist() == [0, 1]
    with pytest.raises(IndexError):
        orc.similarity_order(np.zeros((0, 0), dtype=np.float32))
    with pytest.raises(RuntimeError):
        hl.similarity_order(np.zeros((0, 0), dtype=np.float32))
    # three groups, the outer two identical: they merge first (distance 0), whatever the method
    t = np.array([[1, .2, 1], [.2, 1, .2], [1, .2, 1]], dtype=np.float32)
    for method in METHODS:
        assert hl.similarity_order(t, method).tolist() == _reference_glue([0, 1], [2, 3], 3)
