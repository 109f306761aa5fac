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
            seq = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 30))))
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
    have = {(a, o1, b, o2) for a, o1, b, o2 in links}

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
