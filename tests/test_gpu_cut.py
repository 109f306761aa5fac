"""Subset / exclude intervals cut on the device (pnx_set_csr_cut, csrc/kernels_cut.hip).

Three kinds of checks:
  * the cut ItemTable, the exclusion flags and the uncovered bp against the oracle's restatement of the
    reference walk (util.rs:412-795) on the reference's own BED inputs and on random GFA / BED files;
  * the raw ABI against a per-BASE model written here (every base of every step is looked up in the
    interval lists one by one: no positions-by-prefix-sum, no interval searches, no piece arithmetic) --
    a check that does not restate the walk it checks (the reference holds no expected outputs for -s/-e);
  * identities: lists that cover everything change nothing, whole-path lists equal path selection.
"""
import os

import numpy as np
import pytest

import oracle as orc
from panacus_amd import capi, hostlib as hl
from test_host_gfa_fuzz import _random_bed, _random_gfa

pytestmark = pytest.mark.gpu
N_SEEDS = int(os.environ.get("PANACUS_FUZZ_SEEDS", "12"))
MAXU = (1 << 64) - 1


def _keep_failure(gfa, sf, ef, note):
    """PANACUS_KEEP_FAILURES=<dir>: the inputs of a failing random case are copied there (soak runs)"""
    keep = os.environ.get("PANACUS_KEEP_FAILURES")
    if not keep:
        return
    import shutil
    import uuid
    d = os.path.join(keep, uuid.uuid4().hex[:8])
    os.makedirs(d, exist_ok=True)
    for f in (gfa, sf, ef):
        if f:
            shutil.copy(f, d)
    with open(os.path.join(d, "note.txt"), "w") as f:
        f.write(note + "\n")


def _compare_with_oracle(g, hg, gfa, sf, ef, mode=orc.GROUP_PATHID):
    try:
        g.path_order(mode, None, None, sf, ef)
    except ValueError:
        with pytest.raises(ValueError):
            hg.path_order(mode, None, None, sf, ef)
        return
    for ct in (orc.NODE, orc.BP, orc.EDGE):
        try:
            items, pre, fl, ids, ubp = g.masked_table(ct, sf, ef)
        except ValueError:  # a row the reference panics on: the device path must refuse it as well
            with capi.Context() as ctx, pytest.raises(ValueError):
                hg.cut_upload(ctx, ct, sf, ef, mode)
            continue
        with capi.Context() as ctx:
            uid, ub = hg.cut_upload(ctx, ct, sf, ef, mode)
            got_items, got_off, _ = ctx.get_csr()
            ok = (np.array_equal(got_off, pre) and np.array_equal(got_items.astype(np.uint64), items.astype(np.uint64)) and
                  np.array_equal(ctx.get_exclude()[1:], fl[1:]) and
                  np.array_equal(uid.astype(np.uint64), ids.astype(np.uint64)) and np.array_equal(ub, ubp))
            if not ok:
                _keep_failure(gfa, sf, ef, f"ct={ct} mode={mode} sf={sf} ef={ef} oracle_unc={ids.tolist()}/{ubp.tolist()} "
                                           f"device_unc={uid.tolist()}/{ub.tolist()}")
            assert np.array_equal(got_off, pre), (ct, sf, ef)
            assert np.array_equal(got_items.astype(np.uint64), items.astype(np.uint64)), (ct, sf, ef)
            assert np.array_equal(ctx.get_exclude()[1:], fl[1:]), (ct, sf, ef)
            assert np.array_equal(uid.astype(np.uint64), ids.astype(np.uint64)) and np.array_equal(ub, ubp), (ct, sf, ef)


def test_cut_chrM_reference_beds(golden_dir):
    gfa = os.path.join(golden_dir, "chrM_test.gfa")
    bed = os.path.join(golden_dir, "bed_chrM")
    g = orc.Graph(gfa, index_edges=True)
    hg = hl.GfaGraph(gfa, index_edges=True)
    for sf, ef in (("inclusion.bed3", None), (None, "exclusion.bed3"), ("inclusion.bed3", "exclusion.bed3"),
                   ("inclusion_sub.bed1", "exclusion.bed3"), ("inclusion.bed1", None), ("inclusion.bed12", "exclusion.bed12")):
        sf, ef = sf and os.path.join(bed, sf), ef and os.path.join(bed, ef)
        if (sf and not os.path.exists(sf)) or (ef and not os.path.exists(ef)):
            continue
        _compare_with_oracle(g, hg, gfa, sf, ef)
        _compare_with_oracle(g, hg, gfa, sf, ef, orc.GROUP_SAMPLE)


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_cut_random_gfa_against_oracle(tmp_path, seed):
    rng = np.random.default_rng(21000 + seed)
    gfa = str(tmp_path / "r.gfa")
    _random_gfa(rng, gfa, crlf=False)
    try:
        g = orc.Graph(gfa, index_edges=True)
    except Exception:
        pytest.skip("generator produced a graph the reference rejects")
    hg = hl.GfaGraph(gfa, index_edges=True)
    items0, pre0 = g.item_table(orc.NODE)
    path_bp = [int(g.node_lens[items0[pre0[k]:pre0[k + 1]]].sum()) for k in range(g.n_paths)]
    mode = [orc.GROUP_PATHID, orc.GROUP_SAMPLE, orc.GROUP_HAPLOTYPE][seed % 3]
    _, _, gnames = g.path_order(mode)
    for rep in range(3):
        sf = ef = None
        if rng.random() < 0.8:
            sf = str(tmp_path / f"s{rep}.bed")
            _random_bed(rng, sf, g.path_names(), path_bp, gnames if mode != orc.GROUP_PATHID else [])
        if rng.random() < 0.8 or sf is None:
            ef = str(tmp_path / f"e{rep}.bed")
            _random_bed(rng, ef, g.path_names(), path_bp, gnames if mode != orc.GROUP_PATHID else [])
        _compare_with_oracle(g, hg, gfa, sf, ef, mode)


# ---- the per-base model ---------------------------------------------------------------------------

def _joined(ivs):
    out = []
    for s, e in sorted(ivs):
        if out and out[-1][1] >= s:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return [tuple(x) for x in out]


def _per_base_model(walk_node, walk_off, node_len, backward, inc, exc, start, count_bp):
    """-> items per path (a node once per include interval that holds one of its bases), node flags,
    per node: the set of its bases (node coordinates) that include intervals of CUT paths cover, as seen
    after the last full sighting, and the bases exclude intervals cover."""
    P = len(walk_off) - 1
    n = len(node_len) - 1
    out_items = []
    flags = np.zeros(n + 1, dtype=np.uint8)
    cov = {}      # node -> set of bases covered since the last full sighting
    partial_exc = {}
    for k in range(P):
        items = []
        pos = start[k]
        for j in range(walk_off[k], walk_off[k + 1]):
            v, l = int(walk_node[j]), int(node_len[walk_node[j]])
            bases = range(pos, pos + l)
            for (s, e) in inc[k]:
                hit = [b - pos for b in bases if s <= b < e]
                if not hit:
                    continue
                items.append(v)
                if backward[j]:
                    hit = [l - 1 - x for x in hit]
                if count_bp:
                    if len(hit) == l:
                        cov.pop(v, None)
                    else:
                        cov.setdefault(v, set()).update(hit)
            for (s, e) in (exc[k] if exc is not None else []):
                hit = [b - pos for b in bases if s <= b < e]
                if not hit:
                    continue
                if backward[j]:
                    hit = [l - 1 - x for x in hit]
                if not count_bp or len(hit) == l:
                    flags[v] = 1
                else:
                    partial_exc.setdefault(v, set()).update(hit)
            pos += l
        out_items.append(items)
    for v, bs in partial_exc.items():
        if len(bs) == node_len[v]:
            flags[v] = 1
    return out_items, flags, cov, partial_exc


@pytest.mark.parametrize("seed", range(8))
def test_cut_raw_abi_against_per_base_model(seed):
    rng = np.random.default_rng(500 + seed)
    n, P = int(rng.integers(20, 120)), int(rng.integers(2, 9))
    node_len = np.concatenate([[0], rng.integers(1, 9, n)]).astype(np.uint32)
    lens = rng.integers(0, 3 * n, P)
    lens[rng.integers(0, P)] = 5000 if seed % 2 else lens[0]  # some paths span several 2048-step chunks
    walk_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    S = int(walk_off[-1])
    walk_node = rng.integers(1, n + 1, S).astype(np.uint32)
    backward = (rng.random(S) < 0.3).astype(np.uint8)
    start = rng.integers(0, 50, P).astype(np.uint64)
    path_bp = [int(node_len[walk_node[walk_off[k]:walk_off[k + 1]]].sum()) for k in range(P)]

    def lists(p_none):
        out = []
        for k in range(P):
            if rng.random() < p_none or path_bp[k] == 0:
                out.append([])
                continue
            ivs = []
            for _ in range(int(rng.integers(1, 6))):
                s = int(start[k]) + int(rng.integers(0, path_bp[k]))
                ivs.append((s, s + int(rng.integers(1, max(2, path_bp[k] // 3)))))
            # sorted, disjoint and NOT touching, as GraphMask's interval sets are
            out.append(_joined(ivs))
        return out

    for count_type in (0, 1):
        inc, exc = lists(0.2), (lists(0.4) if seed % 3 else None)
        mode = np.array([capi.WALK_CUT if inc[k] or (exc and exc[k]) else capi.WALK_SKIP for k in range(P)], dtype=np.uint8)
        with capi.Context() as ctx:
            ev = ctx.set_csr_cut(walk_node, walk_off, node_len, mode, inc, exc, start, backward, count_type=count_type,
                                 weights=node_len if count_type == 1 else None, track_covered=count_type == 1)
            items, off, _ = ctx.get_csr()
            flags = ctx.get_exclude()
        exp_items, exp_flags, cov, pexc = _per_base_model(walk_node, walk_off, node_len, backward, inc, exc, start, count_type == 1)
        for k in range(P):
            if mode[k] == capi.WALK_SKIP:
                assert off[k] == off[k + 1]
                continue
            assert items[int(off[k]):int(off[k + 1])].tolist() == exp_items[k], (seed, count_type, k)
        # the device flags everything a full piece covers; pieces that only JOIN to a whole node are the host's part
        joined = np.zeros(n + 1, dtype=np.uint8)
        got_exc = {}
        for e in ev:
            if e["kind"] == 2 and not e["flagged"]:
                got_exc.setdefault(e["item"], set()).update(range(e["a"], e["b"]))
        for v, bs in got_exc.items():
            if len(bs) == node_len[v]:
                joined[v] = 1
        assert np.array_equal(np.maximum(flags, joined)[1:], exp_flags[1:]), (seed, count_type)
        if count_type == 1:
            got_cov = {}
            for e in sorted(ev, key=lambda x: (x["step"], x["piece"])):
                if e["kind"] == 0 and e["step"] + 1 > e["last_full"]:
                    got_cov.setdefault(e["item"], set()).update(range(e["a"], e["b"]))
            assert got_cov == cov, (seed, count_type)
        else:
            assert not ev


def test_cut_identities(tmp_path):
    """lists that select everything change nothing; whole-path lists equal path selection by pnx_set_order"""
    path = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "60000", "--paths", "12", "--links", "-o", path])
    assert rc == 0, err
    hg = hl.GfaGraph(path, index_edges=True)
    names = [n.split(":")[0] for n in hg.path_names()]
    everything = tmp_path / "all.bed"
    everything.write_text("".join(f"{nm}\t0\t{1 << 40}\n" for nm in names))
    some = tmp_path / "some.bed"
    some.write_text("".join(f"{nm}\n" for nm in names[::3]))
    for ct in (hl.NODE, hl.BP, hl.EDGE):
        items, pre = hg.item_table(ct)
        with capi.Context() as ctx:
            uid, _ = hg.cut_upload(ctx, ct, str(everything), None)
            got, off, _ = ctx.get_csr()
            assert len(uid) == 0 and np.array_equal(off, pre) and np.array_equal(got, items)
        with capi.Context() as ctx:
            hg.cut_upload(ctx, ct, str(some), None)
            got, off, _ = ctx.get_csr()
            for k in range(len(names)):
                seg = got[int(off[k]):int(off[k + 1])]
                if k % 3 == 0:
                    assert np.array_equal(seg, items[int(pre[k]):int(pre[k + 1])])
                else:
                    assert len(seg) == 0


def test_cut_larger_graph_against_host_walk(tmp_path):
    """300 k nodes x 24 paths (6.9 M steps over ~3400 chunks), 40 intervals per list: device cut == host walk"""
    path = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "300000", "--paths", "24", "--links", "-o", path])
    assert rc == 0, err
    hg = hl.GfaGraph(path, index_edges=True)
    names = [n.split(":")[0] for n in hg.path_names()]
    items, pre = hg.item_table(hl.NODE)
    lens = hg.node_lens
    bp = [int(lens[items[int(pre[k]):int(pre[k + 1])]].sum()) for k in range(len(names))]
    rng = np.random.default_rng(5)

    def bed(fn, rows):
        with open(fn, "w") as f:
            for _ in range(rows):
                k = int(rng.integers(0, len(names)))
                lo = int(rng.integers(0, bp[k]))
                f.write(f"{names[k]}\t{lo}\t{lo + int(rng.integers(1, bp[k] // 4))}\n")
    sf, ef = str(tmp_path / "s.bed"), str(tmp_path / "e.bed")
    bed(sf, 40)
    bed(ef, 40)
    n_unc = 0
    for ct in (hl.NODE, hl.BP, hl.EDGE):
        h_items, h_pre, h_fl, h_ids, h_bps = hg.masked_table(ct, sf, ef)
        with capi.Context() as ctx:
            uid, ub = hg.cut_upload(ctx, ct, sf, ef)
            got, off, _ = ctx.get_csr()
            assert np.array_equal(off, h_pre) and np.array_equal(got, h_items), ct
            assert np.array_equal(ctx.get_exclude()[1:], h_fl[1:]), ct
            assert np.array_equal(uid, h_ids) and np.array_equal(ub, h_bps), ct
            n_unc += len(uid)
    assert n_unc > 0


def test_edge_items_looked_up_on_the_device(tmp_path):
    """edge counts: the node walks go up, the library finds the edge of every step pair in its hash table (pnx_walks.edge_uv /
    edge_oo instead of a ready-made edge ItemTable): same table as the host's lookups, on a graph with inversions ('-'
    steps), duplications and shuffled links; an unknown edge fails the call"""
    path = str(tmp_path / "pggb.gfa")
    rc, out, err = hl.run_cli(["synth", "--shape", "pggb", "--nodes", "60000", "--samples", "5", "-o", path])
    assert rc == 0, err
    lines = open(path).read().split("\n")
    links = [l for l in lines if l.startswith("L")]
    rest = [l for l in lines if l and not l.startswith("L")]
    np.random.default_rng(3).shuffle(links)
    path2 = str(tmp_path / "shuffled.gfa")
    open(path2, "w").write("\n".join(rest + links) + "\n")
    for gfa in (path, path2):
        hg = hl.GfaGraph(gfa, index_edges=True)
        items, pre = hg.item_table(hl.EDGE)
        with capi.Context() as ctx:
            hg.cut_upload(ctx, hl.EDGE)  # no lists: every path taken with the whole-path interval
            got, off, _ = ctx.get_csr()
            assert np.array_equal(off, pre) and np.array_equal(got, items), gfa
    # a walk that uses an edge the graph does not have
    node_len = np.array([0, 3, 4, 5, 6], dtype=np.uint32)
    uv = np.array([0, (1 << 32) | 2, (2 << 32) | 3], dtype=np.uint64)
    oo = np.zeros(3, dtype=np.uint8)
    with capi.Context() as ctx:
        mode = np.array([capi.WALK_CUT], dtype=np.uint8)
        ev = ctx.set_csr_cut([1, 2, 3], [0, 3], node_len, mode, [[(0, MAXU)]], count_type=2, edge_uv=uv, edge_oo=oo, n_items=2)
        got, off, _ = ctx.get_csr()
        assert got.tolist() == [1, 2] and off.tolist() == [0, 2] and not ev
        with pytest.raises(capi.PnxError, match="unknown edge in path 0"):
            ctx.set_csr_cut([1, 2, 4], [0, 3], node_len, mode, [[(0, MAXU)]], count_type=2, edge_uv=uv, edge_oo=oo, n_items=2)


def test_cut_at_cfg3_size():
    """BASELINE configs[2]'s graph (10 M nodes x 256 paths, 0.98 G steps, 478 k chunks) through the device cut: the
    whole-path interval reproduces the table; one interval per path keeps exactly the steps whose node starts before the
    interval's end and ends after its start -- checked for sampled paths against numpy prefix sums; exclusion flags =
    the nodes those steps visit (node counts: any touch)."""
    n, p = 10_000_000, 256
    with capi.Context() as src:
        src.set_csr_pansyn(42, n, p, with_weights=True)
        items, off, lens = src.get_csr(want_weights=True)
    mode = np.full(p, capi.WALK_CUT, dtype=np.uint8)
    with capi.Context() as ctx:
        ev = ctx.set_csr_cut(items, off, lens, mode, [[(0, MAXU)]] * p, count_type=0)
        assert not ev and ctx.info().n_steps == len(items)
        got, goff, _ = ctx.get_csr()
        assert np.array_equal(goff, off) and np.array_equal(got, items)
        del got
        # one interval per path, somewhere in its middle; every 3rd path also has an exclude interval
        rng = np.random.default_rng(17)
        sample = [0, 15, 100, 255]  # 15: a descending path
        inc, exc = [], []
        pos = {}
        for k in range(p):
            seg = items[int(off[k]):int(off[k + 1])]
            if k in sample:
                pos[k] = np.concatenate([[0], np.cumsum(lens[seg].astype(np.uint64))])
                total = int(pos[k][-1])
            else:
                total = int(lens[seg].sum(dtype=np.uint64))
            a = int(rng.integers(0, total // 2))
            b = a + int(rng.integers(1, total // 3))
            inc.append([(a, b)])
            exc.append([(b - 1000, b + 1000)] if k % 3 == 0 else [])
        ctx.set_csr_cut(items, off, lens, mode, inc, exc, count_type=0)
        got, goff, _ = ctx.get_csr()
        flags = ctx.get_exclude()
        for k in sample:
            seg = items[int(off[k]):int(off[k + 1])]
            st, en = pos[k][:-1], pos[k][1:]
            (a, b), = inc[k]
            keep = (st < b) & (en > a)
            assert np.array_equal(got[int(goff[k]):int(goff[k + 1])], seg[keep]), k
            if exc[k]:
                (c, d), = exc[k]
                assert flags[seg[(st < d) & (en > c)]].all(), k
        assert int(goff[-1]) == len(got) and 0 < len(got) < len(items)


def test_cut_rows_with_start_beyond_end(golden_dir):
    """the two soak-run cases of tests/golden/bed_inverted (BED rows with start > end inside one node) through the device"""
    base = os.path.join(golden_dir, "bed_inverted")
    for d, mode in (("6cd09061", orc.GROUP_PATHID), ("8438eb63", orc.GROUP_SAMPLE)):
        gfa, sf, ef = (os.path.join(base, d, f) for f in ("r.gfa", "subset.bed", "exclude.bed"))
        _compare_with_oracle(orc.Graph(gfa, index_edges=True), hl.GfaGraph(gfa, index_edges=True), gfa, sf, ef, mode)
