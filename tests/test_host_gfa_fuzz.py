"""Randomised differential test of the host GFA front end against the oracle's independent
parser: random graphs written as GFA text with P and W lines, mixed orientations, numeric and
non-numeric segment names, PanSN and plain path names, header/comment lines, optional CRLF."""
import os

import numpy as np
import pytest

import oracle as orc
from panacus_amd import hostlib as hl

N_SEEDS = int(os.environ.get("PANACUS_FUZZ_SEEDS", "40"))


def _random_gfa(rng, path, crlf):
    n = int(rng.integers(1, 60))
    nice = rng.random() < 0.3
    if nice:
        names = [str(i + 1) for i in range(n)]
    else:
        alphabet = list("abcXYZ019_.")
        names = []
        while len(names) < n:
            s = "".join(rng.choice(alphabet, size=int(rng.integers(1, 7))))
            if s not in names and s not in ("+", "-"):
                names.append(s)
    eol = "\r\n" if crlf else "\n"
    lines = ["H\tVN:Z:1.0"]
    for i, nm in enumerate(names):
        if rng.random() < 0.5:
            # now and then an EMPTY sequence field: a node of length 0 (graph.rs:343-349 takes the field's length)
            seq = "" if rng.random() < 0.04 else "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 30))))
            lines.append(f"S\t{nm}\t{seq}")
        else:
            lines.append(f"S\t{nm}\t*\tLN:i:{int(rng.integers(1, 500))}")
        if rng.random() < 0.1:
            lines.append("# a comment line")
    links = set()
    for _ in range(int(rng.integers(0, 3 * n))):
        a, b = rng.integers(0, n, size=2)
        o1, o2 = rng.choice(["+", "-"], size=2)
        links.add((names[a], o1, names[b], o2))
    links = sorted(links)
    rng.shuffle(links)
    for a, o1, b, o2 in links:
        lines.append(f"L\t{a}\t{o1}\t{b}\t{o2}\t0M")
    have = list(links)  # iterated below: a list, so that a seed gives the same file in every process (a set of str
    # tuples iterates in hash order, and str hashes are randomised per process)

    def flip(o):
        return "-" if o == "+" else "+"

    n_paths = int(rng.integers(1, 12))
    samples = ["s" + str(i) for i in range(int(rng.integers(1, 5)))]
    used = set()
    for k in range(n_paths):
        # a walk along existing links where possible, so that the edge tables are exercised too
        steps = []
        cur = (names[int(rng.integers(0, n))], str(rng.choice(["+", "-"])))
        for _ in range(int(rng.integers(0, 25))):
            steps.append(cur)
            nxt = [(b, o2) for (a, o1, b, o2) in have if a == cur[0] and o1 == cur[1]]
            nxt += [(a, flip(o1)) for (a, o1, b, o2) in have if b == cur[0] and flip(o2) == cur[1]]
            if not nxt:
                break
            cur = nxt[int(rng.integers(0, len(nxt)))]
        if not steps:
            steps = [cur]
        kind = rng.integers(0, 3)
        sample = samples[int(rng.integers(0, len(samples)))]
        hap, ctg = int(rng.integers(0, 3)), "ctg" + str(int(rng.integers(0, 4)))
        if kind == 0:      # W line
            key = (sample, hap, ctg, k)
            walk = "".join((">" if o == "+" else "<") + nm for nm, o in steps)
            lines.append(f"W\t{sample}\t{hap}\t{ctg}{k}\t0\t{10 * len(steps)}\t{walk}")
        else:
            name = f"{sample}#{hap}#{ctg}{k}" if kind == 1 else f"plain{k}"
            if rng.random() < 0.3 and kind == 1:
                name += f":{int(rng.integers(0, 50))}-{int(rng.integers(50, 500))}"
            lines.append("P\t" + name + "\t" + ",".join(nm + o for nm, o in steps) + "\t*")
        used.add(k)
    order = list(range(len(lines)))
    text = eol.join(lines) + eol
    with open(path, "w", newline="") as f:
        f.write(text)


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_gfa_matches_oracle(tmp_path, seed):
    rng = np.random.default_rng(5000 + seed)
    gfa = str(tmp_path / "r.gfa")
    _random_gfa(rng, gfa, crlf=bool(seed % 4 == 3))
    try:
        b = orc.Graph(gfa, index_edges=True)
    except Exception:
        with pytest.raises(Exception):
            hl.GfaGraph(gfa, index_edges=True)
        return
    a = hl.GfaGraph(gfa, index_edges=True)
    assert (a.n_nodes, a.n_edges, a.n_paths) == (b.n_nodes, b.n_edges, b.n_paths)
    assert np.array_equal(a.node_lens, b.node_lens)
    assert a.path_names() == b.path_names()
    for ct in (hl.NODE, hl.EDGE):
        ia, pa = a.item_table(ct)
        ib, pb = b.item_table(ct)
        assert np.array_equal(pa, pb), ct
        assert np.array_equal(ia.astype(np.uint64), ib), ct
    for mode in (hl.GROUP_PATHID, hl.GROUP_SAMPLE, hl.GROUP_HAPLOTYPE):
        pa, ga, na = a.path_order(mode)
        pb, gb, nb = b.path_order(mode)
        assert na == nb and np.array_equal(pa, pb) and np.array_equal(ga, gb), mode
    # the cache gives the same graph back
    a.save_cache(gfa + ".pcsr", gfa)
    c = hl.GfaGraph.from_cache(gfa + ".pcsr", gfa, True)
    for ct in (hl.NODE, hl.EDGE):
        ia, pa = a.item_table(ct)
        ic, pc = c.item_table(ct)
        assert np.array_equal(ia, ic) and np.array_equal(pa, pc)
    assert c.path_names() == a.path_names()


def _random_bed(rng, path, names, path_bp, groups):
    """a BED list over the graph's paths: 1-, 3- and 12-column rows, groups, unknown names, comments"""
    import re
    ids = [re.sub(r":[0-9]+-[0-9]+$", "", nm) for nm in names]
    rows = []
    for _ in range(int(rng.integers(1, 9))):
        k = int(rng.integers(0, len(ids)))
        span = max(int(path_bp[k]), 1)
        kind = rng.random()
        if kind < 0.2:
            rows.append(ids[k])
        elif kind < 0.3 and groups:
            rows.append(groups[int(rng.integers(0, len(groups)))])
        elif kind < 0.34:
            rows.append("nobody#1#knows")
        elif kind < 0.36:  # odd rows: two columns, a coordinate that is no number, a huge coordinate
            rows.append([f"{ids[k]}\t7", f"{ids[k]}\tx\t9", f"{ids[k]}\t3\t99999999999999"][int(rng.integers(0, 3))])
        elif kind < 0.42:
            rows.append("# a comment" if rng.random() < 0.5 else "track name=x")
        elif kind < 0.5:
            st = int(rng.integers(0, span))
            sizes = [int(rng.integers(1, 40)) for _ in range(int(rng.integers(1, 4)))]
            starts, at = [], 0
            for sz in sizes:
                at += int(rng.integers(0, 30))
                starts.append(at)
                at += sz
            rows.append("\t".join([ids[k], str(st), str(st + at), "n", "0", "+", str(st), str(st + at), "0",
                                   str(len(sizes)), ",".join(map(str, sizes)) + ",", ",".join(map(str, starts))]))
        else:
            lo = int(rng.integers(0, span + 20))
            hi = lo + int(rng.integers(0 if rng.random() < 0.1 else 1, max(2, span // int(rng.integers(1, 6)))))
            if rng.random() < 0.15:
                lo = 0
            if rng.random() < 0.15:
                hi = span + int(rng.integers(0, 3))
            if rng.random() < 0.06 and lo > 2:  # start > end, close together: both ends inside one node now and then
                hi = lo - int(rng.integers(1, min(lo, 12)))
            rows.append(f"{ids[k]}\t{lo}\t{hi}" + ("\textra" if rng.random() < 0.1 else ""))
    with open(path, "w") as f:
        f.write("\n".join(rows) + "\n")


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_bed_lists_match_oracle(tmp_path, seed):
    """-s / -e BED lists with coordinates: ItemTable, exclude flags, uncovered bps and the visiting
    order of the host against the oracle's literal restatement (parity unpinned)."""
    rng = np.random.default_rng(9000 + seed)
    gfa = str(tmp_path / "r.gfa")
    _random_gfa(rng, gfa, crlf=False)
    try:
        b = orc.Graph(gfa, index_edges=True)
    except Exception:
        return
    a = hl.GfaGraph(gfa, index_edges=True)
    items, pre = b.item_table(orc.NODE)
    lens = b.node_lens
    path_bp = [int(lens[items[pre[k]:pre[k + 1]]].sum()) for k in range(b.n_paths)]
    for rep in range(3):
        mode = [hl.GROUP_PATHID, hl.GROUP_SAMPLE, hl.GROUP_HAPLOTYPE][rep]
        _, _, gnames = b.path_order(mode)
        sf = ef = None
        if rng.random() < 0.75:
            sf = str(tmp_path / f"s{rep}.bed")
            _random_bed(rng, sf, b.path_names(), path_bp, gnames if mode != hl.GROUP_PATHID else [])
        if rng.random() < 0.75:
            ef = str(tmp_path / f"e{rep}.bed")
            _random_bed(rng, ef, b.path_names(), path_bp, gnames if mode != hl.GROUP_PATHID else [])
        try:
            po = b.path_order(mode, None, None, sf, ef)
        except ValueError:
            with pytest.raises(ValueError):
                a.path_order(mode, None, None, sf, ef)
            continue
        ph = a.path_order(mode, None, None, sf, ef)
        assert po[2] == ph[2] and np.array_equal(po[0], ph[0]) and np.array_equal(po[1], ph[1])
        for ct in (hl.NODE, hl.BP, hl.EDGE):
            try:
                x = b.masked_table(ct, sf, ef)
            except ValueError:  # a row the reference panics on (two columns, a coordinate that is no number)
                with pytest.raises(ValueError):
                    a.masked_table(ct, sf, ef, mode)
                continue
            y = a.masked_table(ct, sf, ef, mode)
            for k, (u, v) in enumerate(zip(x, y)):
                assert np.array_equal(np.asarray(u, dtype=np.uint64), np.asarray(v, dtype=np.uint64)), (ct, k, sf, ef)
