"""Host GFA front end (panacus_amd/host/gfa_graph.cpp) against the reference's fixtures and
the oracle's independent parser: ids, lengths, PanSN parsing, groups, order, node/edge CSR."""
import gzip
import os

import numpy as np
import pytest

import oracle as orc
from panacus_amd import hostlib as hl

FIX = ["chrM_test.gfa", "cdbg.gfa", "t_groups.gfa"]


@pytest.mark.parametrize("name", FIX)
def test_graph_matches_oracle(golden_dir, name):
    gfa = os.path.join(golden_dir, name)
    a = hl.GfaGraph(gfa, index_edges=True)
    b = orc.Graph(gfa, index_edges=True)
    assert (a.n_nodes, a.n_edges, a.n_paths) == (b.n_nodes, b.n_edges, b.n_paths)
    assert np.array_equal(a.node_lens, b.node_lens)
    assert a.path_names() == b.path_names()
    for ct in (hl.NODE, hl.EDGE):
        ia, pa = a.item_table(ct)
        ib, pb = b.item_table(ct)
        assert np.array_equal(pa, pb)
        assert np.array_equal(ia.astype(np.uint64), ib)
    for mode in (hl.GROUP_PATHID, hl.GROUP_SAMPLE, hl.GROUP_HAPLOTYPE):
        pa, ga, na = a.path_order(mode)
        pb, gb, nb = b.path_order(mode)
        assert na == nb and np.array_equal(pa, pb) and np.array_equal(ga, gb)


def test_chrM_shape_and_groups(golden, golden_dir):
    g = hl.GfaGraph(os.path.join(golden_dir, "chrM_test.gfa"), index_edges=True)
    assert (g.n_nodes, g.n_edges, g.n_paths) == (154, 205, 4)
    pi, gi, names = g.path_order(hl.GROUP_SAMPLE)
    assert names == golden["chrM_sample_node"]["groups"]


def test_group_and_order_files(golden_dir, tmp_path):
    gfa = os.path.join(golden_dir, "cdbg.gfa")
    a, b = hl.GfaGraph(gfa), orc.Graph(gfa)
    order = tmp_path / "order.txt"
    order.write_text("d#1#h1\nc#2#h1\n# comment\na#1#h1\nb#1#h1\nc#1#h1\nc#1#h2\nnosuch\n")
    groups = tmp_path / "groups.tsv"
    groups.write_text("a#1#h1\tG1\nb#1#h1\tG1\nc#1#h1\tG2\nc#1#h2\tG2\n")
    for gm, gf, of in [(hl.GROUP_PATHID, None, str(order)), (hl.GROUP_FILE, str(groups), None),
                       (hl.GROUP_FILE, str(groups), str(order))]:
        pa, ga, na = a.path_order(gm, gf, of)
        pb, gb, nb = b.path_order(gm, gf, of)
        assert na == nb and np.array_equal(pa, pb) and np.array_equal(ga, gb)
    pa, ga, na = a.path_order(hl.GROUP_FILE, str(groups))
    assert na == ["G1", "G2", "c#2#h1", "d#1#h1"]
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.tsv"
        bad.write_text("a#1#h1\tG1\tx\n")
        a.path_order(hl.GROUP_FILE, str(bad))


def _write_gfa(path, items, pre, lens, names=None, walks=(), gz=False, crlf=False):
    """synthetic GFA text from a CSR (segment names are NOT the plain integers, to exercise hashing)"""
    nl = "\r\n" if crlf else "\n"
    n = len(lens) - 1
    out = ["H\tVN:Z:1.0"]
    for i in range(1, n + 1):
        out.append(f"S\ts{i}\t{'A' * int(lens[i])}")
    P = len(pre) - 1
    links = set()
    for p in range(P):
        ids = items[pre[p]:pre[p + 1]]
        ori = ["+" if (int(x) * 7 + p) % 5 else "-" for x in ids]
        for k in range(len(ids) - 1):
            links.add((int(ids[k]), ori[k], int(ids[k + 1]), ori[k + 1]))
        nm = names[p] if names else f"s{p // 2}#{p % 2}#ctg"
        if p in walks:
            s, h, c = nm.split("#")
            walk = "".join((">" if o == "+" else "<") + f"s{int(x)}" for x, o in zip(ids, ori))
            out.append(f"W\t{s}\t{h}\t{c}\t0\t{len(ids)}\t{walk}")
        else:
            out.append(f"P\t{nm}\t" + ",".join(f"s{int(x)}{o}" for x, o in zip(ids, ori)) + "\t*")
    for (u, o1, v, o2) in sorted(links):
        out.append(f"L\ts{u}\t{o1}\ts{v}\t{o2}\t0M")
    data = (nl.join(out) + nl).encode()
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


@pytest.mark.parametrize("gz,crlf", [(False, False), (True, False), (False, True)])
def test_synthetic_text_roundtrip(tmp_path, gz, crlf):
    n, p = 3000, 9
    items, pre, lens = orc.pansyn(5, n, p)
    lens = np.minimum(lens, 40)  # keep the text small
    path = str(tmp_path / ("g.gfa.gz" if gz else "g.gfa"))
    _write_gfa(path, items, pre, lens, walks={2, 5}, gz=gz, crlf=crlf)
    a = hl.GfaGraph(path, index_edges=True)
    assert a.n_nodes == n and a.n_paths == p
    assert np.array_equal(a.node_lens, lens.astype(np.uint32))
    ia, pa = a.item_table(hl.NODE)
    assert np.array_equal(pa, pre) and np.array_equal(ia.astype(np.uint64), items)
    if not gz:  # the oracle reads plain text only
        b = orc.Graph(path, index_edges=True)
        ea, epa = a.item_table(hl.EDGE)
        eb, epb = b.item_table(orc.EDGE)
        assert a.n_edges == b.n_edges
        assert np.array_equal(epa, epb) and np.array_equal(ea.astype(np.uint64), eb)
        for mode in (hl.GROUP_PATHID, hl.GROUP_SAMPLE, hl.GROUP_HAPLOTYPE):
            assert a.path_order(mode)[2] == b.path_order(mode)[2]


def test_pansn_parsing_matches_oracle(tmp_path):
    names = ["plain", "a#1", "a#1#chr1", "a#1#chr1:10-20", "a#b#", "a##b", "x:5-9", "a#1:3-4", "a#b#c#d",
             "#lead", "a#b#c:1-2:3-4", "s#0#ctg:99999999999999999999-5"]
    path = str(tmp_path / "n.gfa")
    with open(path, "w") as f:
        f.write("S\t1\tACGT\n")
        for nm in names:
            f.write(f"P\t{nm}\t1+\t*\n")
    a, b = hl.GfaGraph(path), orc.Graph(path)
    assert a.path_names() == b.path_names()
    for mode in (hl.GROUP_PATHID, hl.GROUP_SAMPLE, hl.GROUP_HAPLOTYPE):
        assert a.path_order(mode)[2] == b.path_order(mode)[2]


def test_errors(tmp_path):
    p = str(tmp_path / "dup.gfa")
    open(p, "w").write("S\t1\tA\nS\t1\tC\n")
    with pytest.raises(ValueError):
        hl.GfaGraph(p)
    p = str(tmp_path / "unk.gfa")
    open(p, "w").write("S\t1\tA\nP\tx\t1+,2+\t*\n")
    g = hl.GfaGraph(p)
    with pytest.raises(ValueError):
        g.item_table(hl.NODE)
    with pytest.raises(ValueError):
        hl.GfaGraph(str(tmp_path / "missing.gfa"))


def test_synth_subcommand_roundtrip(tmp_path):
    """`panacus-amd synth` writes pansyn-v1 as GFA; parsing it gives back the generator's CSR"""
    path = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "2500", "--paths", "10", "--seed", "7", "--links", "-o", path])
    assert rc == 0, err
    items, pre, lens = orc.pansyn(7, 2500, 10)
    g = hl.GfaGraph(path, index_edges=True)
    ia, pa = g.item_table(hl.NODE)
    assert np.array_equal(ia.astype(np.uint64), items) and np.array_equal(pa, pre)
    assert np.array_equal(g.node_lens, lens)
    pi, gi, names = g.path_order(hl.GROUP_SAMPLE)
    assert names == [f"s{k}" for k in range(5)]
    b = orc.Graph(path, index_edges=True)
    ea, _ = g.item_table(hl.EDGE)
    eb, _ = b.item_table(orc.EDGE)
    assert np.array_equal(ea.astype(np.uint64), eb)


def test_subset_and_exclude_lists_match_oracle(golden_dir, tmp_path):
    """-s / -e with whole-path (or group) lists: visiting order, groups and ActiveTable flags.
    No golden output exists in the reference for -s/-e: the oracle restatement is the definition."""
    path = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "4000", "--paths", "12", "--links", "-o", path])
    assert rc == 0, err
    a, b = hl.GfaGraph(path, index_edges=True), orc.Graph(path, index_edges=True)
    names = a.path_names()
    sub = tmp_path / "sub.txt"
    sub.write_text("\n".join([names[8].split(":")[0], names[1], "# c", names[4], "nosuch#0#x", names[1]]) + "\n")
    exc = tmp_path / "exc.txt"
    exc.write_text(names[2] + "\n" + names[9] + "\n")
    grp_list = tmp_path / "grp.txt"
    grp_list.write_text("s3\ns0\n")  # sample names: groups under -S
    cases = [
        (hl.GROUP_PATHID, None, str(sub), None), (hl.GROUP_SAMPLE, None, str(sub), None),
        (hl.GROUP_PATHID, None, None, str(exc)), (hl.GROUP_HAPLOTYPE, None, None, str(exc)),
        (hl.GROUP_SAMPLE, None, str(grp_list), None), (hl.GROUP_SAMPLE, None, None, str(grp_list)),
        (hl.GROUP_SAMPLE, None, str(sub), str(exc)),
    ]
    for gm, of, sf, ef in cases:
        pa, ga, na = a.path_order(gm, None, of, sf, ef)
        pb, gb, nb = b.path_order(gm, None, of, sf, ef)
        assert na == nb and np.array_equal(pa, pb) and np.array_equal(ga, gb), (gm, sf, ef)
        if ef:
            for ct in (hl.NODE, hl.EDGE):
                fa = a.exclude_flags(ct, ef, gm)
                fb = b.exclude_flags(ct, ef)
                assert np.array_equal(fa, fb)
                assert fa.sum() > 0
    # subset by explicit paths keeps only those paths, in list order (duplicates ignored)
    pa, ga, na = a.path_order(hl.GROUP_PATHID, None, None, str(sub), None)
    assert pa.tolist() == [8, 1, 4]
    # whole-path lists through the interval machinery give the same flags, and the full steps of
    # exactly the listed paths
    for ct in (hl.NODE, hl.BP, hl.EDGE):
        items, pre = a.item_table(hl.EDGE if ct == hl.EDGE else hl.NODE)
        mi, mp, mf, ui, ub = a.masked_table(ct, str(sub), str(exc))
        oi, op, of_, oui, oub = b.masked_table(ct, str(sub), str(exc))
        assert np.array_equal(mi.astype(np.uint64), oi) and np.array_equal(mp, op) and np.array_equal(mf, of_)
        assert len(ui) == 0 and len(oui) == 0
        assert np.array_equal(mf, a.exclude_flags(hl.EDGE if ct == hl.EDGE else hl.NODE, str(exc)))
        for k in range(a.n_paths):
            want = items[pre[k]:pre[k + 1]] if k in (8, 1, 4) else items[:0]
            assert np.array_equal(mi[mp[k]:mp[k + 1]], want), (ct, k)
    # a -s / -e value that is not a file is a regular expression over the path names (abacus.rs:212-240)
    for pat in ("^s1#", "#1#", "s[02]#0", "s3#|s4#1", "nothing_matches_this"):
        for sf, ef in ((pat, None), (None, pat), (pat, str(exc))):
            pa, ga, na = a.path_order(hl.GROUP_SAMPLE, None, None, sf, ef)
            pb, gb, nb = b.path_order(hl.GROUP_SAMPLE, None, None, sf, ef)
            assert na == nb and np.array_equal(pa, pb) and np.array_equal(ga, gb), (sf, ef)
            x = b.masked_table(hl.NODE, sf, ef)
            y = a.masked_table(hl.NODE, sf, ef, hl.GROUP_SAMPLE)
            for u, v in zip(x, y):
                assert np.array_equal(np.asarray(u, dtype=np.uint64), np.asarray(v, dtype=np.uint64)), (sf, ef)
    import re
    pa, _, _ = a.path_order(hl.GROUP_PATHID, None, None, "^s1#", None)
    assert sorted(pa.tolist()) == [k for k, nm in enumerate(names) if re.search("^s1#", nm)] and len(pa) > 0
    # malformed lists: two columns; a group name with coordinates
    bed = tmp_path / "bed.txt"
    bed.write_text(names[0] + "\t100\n")
    for g in (a, b):
        with pytest.raises(ValueError):
            g.path_order(hl.GROUP_PATHID, None, None, str(bed), None)
    bed.write_text("s3\t0\t100\n")
    for g in (a, b):
        with pytest.raises(ValueError):
            g.path_order(hl.GROUP_SAMPLE, None, None, str(bed), None)
    b.path_order(hl.GROUP_SAMPLE)
    for g, kw in ((a, {"group_mode": hl.GROUP_SAMPLE}), (b, {})):
        with pytest.raises(ValueError):
            g.masked_table(hl.NODE, str(bed), None, **kw)


def _same_graph(a, b, edges):
    assert (a.n_nodes, a.n_paths) == (b.n_nodes, b.n_paths)
    assert np.array_equal(a.node_lens, b.node_lens)
    assert a.path_names() == b.path_names()
    for ct in (hl.NODE, hl.EDGE) if edges else (hl.NODE,):
        ia, pa = a.item_table(ct)
        ib, pb = b.item_table(ct)
        assert np.array_equal(pa, pb) and np.array_equal(ia, ib)
    for mode in (hl.GROUP_PATHID, hl.GROUP_SAMPLE, hl.GROUP_HAPLOTYPE):
        pa, ga, na = a.path_order(mode)
        pb, gb, nb = b.path_order(mode)
        assert na == nb and np.array_equal(pa, pb) and np.array_equal(ga, gb)


@pytest.mark.parametrize("name", FIX)
def test_pcsr_cache_roundtrip(golden_dir, tmp_path, name):
    """the binary cache gives back exactly the parsed graph, and is refused when it is stale"""
    import shutil
    gfa = str(tmp_path / name)
    shutil.copy(os.path.join(golden_dir, name), gfa)
    cache = gfa + ".pcsr"
    g = hl.GfaGraph(gfa, index_edges=True)
    assert hl.GfaGraph.from_cache(cache, gfa, True) is None  # no cache yet
    g.save_cache(cache, gfa)
    c = hl.GfaGraph.from_cache(cache, gfa, True)
    assert c is not None and c.n_edges == g.n_edges
    _same_graph(c, g, edges=True)
    # a cache written without the edge index does not serve an edge request
    g2 = hl.GfaGraph(gfa, index_edges=False)
    g2.save_cache(cache, gfa)
    assert hl.GfaGraph.from_cache(cache, gfa, True) is None
    c2 = hl.GfaGraph.from_cache(cache, gfa, False)
    _same_graph(c2, g2, edges=False)
    # the GFA changes (content and mtime): the cache is stale
    with open(gfa, "a") as f:
        f.write("S\textra\tACGT\n")
    os.utime(gfa, ns=(1, 1))
    assert hl.GfaGraph.from_cache(cache, gfa, False) is None
    # a truncated or foreign file is not a cache
    with open(cache, "wb") as f:
        f.write(b"PCSR0002" + b"\x00" * 11)
    assert hl.GfaGraph.from_cache(cache, gfa, False) is None


def test_pcsr_cache_cli_growth_from_cache(golden_dir, tmp_path):
    """`--cache` writes <gfa>.pcsr on the first run (checked here without a GPU through the graph API)"""
    import shutil
    gfa = str(tmp_path / "t_groups.gfa")
    shutil.copy(os.path.join(golden_dir, "t_groups.gfa"), gfa)
    g = hl.GfaGraph(gfa, index_edges=False)
    g.save_cache(gfa + ".pcsr", gfa)
    c = hl.GfaGraph.from_cache(gfa + ".pcsr", gfa, False)
    items, pre = c.item_table(hl.NODE)
    pi, gi, names = c.path_order(hl.GROUP_PATHID)
    cov = orc.coverage(items.astype(np.uint64), pre, pi.astype(np.uint64), gi.astype(np.uint64), c.n_nodes)
    assert orc.hist(cov, len(names)).tolist() == [5, 0, 10, 0, 0, 0, 0]


def test_edge_relabel_is_a_rank_by_canonical_ends(golden_dir, tmp_path):
    """GraphStorage::edge_relabel: a permutation of 1..E under which the edge steps of a path whose
    node ids rise (fall) rise (fall) too, whatever the order of the L lines."""
    path = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "3000", "--paths", "6", "--links", "-o", path])
    assert rc == 0, err
    lines = open(path).read().split("\n")
    links = [l for l in lines if l.startswith("L\t")]
    rng = np.random.default_rng(2)
    shuf = str(tmp_path / "shuf.gfa")
    with open(shuf, "w") as f:
        f.write("\n".join([l for l in lines if l and not l.startswith("L\t")] + [links[i] for i in rng.permutation(len(links))]) + "\n")
    for gfa in (path, shuf, os.path.join(golden_dir, "chrM_test.gfa")):
        g = hl.GfaGraph(gfa, index_edges=True)
        new_id = g.edge_relabel()
        assert new_id[0] == 0 and sorted(new_id[1:].tolist()) == list(range(1, g.n_edges + 1))
        nodes, npre = g.item_table(hl.NODE)
        edges, epre = g.item_table(hl.EDGE)
        for k in range(g.n_paths):
            nd = nodes[npre[k]:npre[k + 1]].astype(np.int64)
            ed = new_id[edges[epre[k]:epre[k + 1]]].astype(np.int64)
            if len(nd) > 2 and (np.diff(nd) > 0).all():
                assert (np.diff(ed) > 0).all()
            if len(nd) > 2 and (np.diff(nd) < 0).all():
                assert (np.diff(ed) < 0).all()
    # the two files hold the same graph: identical histograms need identical multisets of renumbered steps
    a, b = hl.GfaGraph(path, index_edges=True), hl.GfaGraph(shuf, index_edges=True)
    ea, pa = a.item_table(hl.EDGE)
    eb, pb = b.item_table(hl.EDGE)
    assert np.array_equal(pa, pb) and np.array_equal(a.edge_relabel()[ea], b.edge_relabel()[eb])


def _write_bgzf(data: bytes, path: str, block=60_000):
    """the block gzip of bgzip / htslib (SAM spec 4.1): gzip members with a BC extra field, then the empty EOF block"""
    import struct
    import zlib

    def member(chunk: bytes) -> bytes:
        comp = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = comp.compress(chunk) + comp.flush()
        bsize = len(payload) + 25  # header 18 + payload + trailer 8, minus 1
        head = b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
        return head + payload + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    with open(path, "wb") as f:
        for at in range(0, len(data), block):
            f.write(member(data[at:at + block]))
        f.write(member(b""))


def test_bgzf_input_is_inflated_block_by_block(tmp_path, golden_dir):
    """a bgzip'ed GFA gives the same graph as the plain file (blocks inflated in parallel), a plain .gz still works,
    a damaged block is reported"""
    import gzip
    plain = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--shape", "pggb", "--nodes", "20000", "--samples", "4", "-o", plain])
    assert rc == 0, err
    data = open(plain, "rb").read()
    bg = str(tmp_path / "syn.bgzf.gfa.gz")
    _write_bgzf(data, bg)
    assert gzip.open(bg, "rb").read() == data  # a valid multi-member gzip file
    gz = str(tmp_path / "syn.gfa.gz")
    with gzip.open(gz, "wb", compresslevel=1) as f:
        f.write(data)
    a = hl.GfaGraph(plain, index_edges=True)
    for other in (bg, gz):
        b = hl.GfaGraph(other, index_edges=True)
        assert (a.n_nodes, a.n_edges, a.n_paths) == (b.n_nodes, b.n_edges, b.n_paths)
        assert a.path_names() == b.path_names() and np.array_equal(a.node_lens, b.node_lens)
        for ct in (hl.NODE, hl.EDGE):
            x, y = a.item_table(ct), b.item_table(ct)
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    raw = bytearray(open(bg, "rb").read())
    raw[len(raw) // 2] ^= 0x55
    bad = str(tmp_path / "bad.gfa.gz")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        hl.GfaGraph(bad)


def test_duplicated_links_resolve_to_their_first_line(tmp_path):
    """the edge map is filled by all threads at once: an edge written several times -- also in its other spelling, b- a- for
    a+ b+ -- keeps the id of its FIRST line (duplicates are skipped, graph.rs:296), ids are ranks of the first occurrences in
    file order, exactly as the oracle's serial pass gives them"""
    path = str(tmp_path / "syn.gfa")
    rc, out, err = hl.run_cli(["synth", "--nodes", "30000", "--paths", "8", "--links", "-o", path])
    assert rc == 0, err
    lines = [l for l in open(path).read().split("\n") if l]
    links = [l for l in lines if l.startswith("L\t")]
    rng = np.random.default_rng(5)
    flip = {"+": "-", "-": "+"}
    extra = []
    for i in rng.integers(0, len(links), size=len(links) // 2):
        _, a, oa, b, ob, cg = links[int(i)].split("\t")
        extra.append(links[int(i)] if rng.random() < 0.5 else "\t".join(["L", b, flip[ob], a, flip[oa], cg]))
    mixed = links + extra
    order = rng.permutation(len(mixed))
    dup = str(tmp_path / "dup.gfa")
    with open(dup, "w") as f:
        f.write("\n".join([l for l in lines if not l.startswith("L\t")] + [mixed[i] for i in order]) + "\n")
    a = hl.GfaGraph(dup, index_edges=True)
    b = orc.Graph(dup, index_edges=True)
    assert a.n_edges == b.n_edges == len(links)
    ia, pa = a.item_table(hl.EDGE)
    ib, pb = b.item_table(hl.EDGE)
    assert np.array_equal(pa, pb) and np.array_equal(ia.astype(np.uint64), ib)


def test_how_segments_are_named(tmp_path):
    """what the S-line pass concludes about the names: plain numbers and `s12`-style names (the same <= 8 bytes in front of a
    number) go through the numeric tokeniser, anything else of <= 16 bytes is hashed, longer names stay on the host"""
    def kind(names, steps=None):
        p = str(tmp_path / "k.gfa")
        steps = steps or names
        with open(p, "w") as f:
            f.write("H\tVN:Z:1.0\n" + "".join(f"S\t{nm}\tACGT\n" for nm in names) + "P\tp#1#c\t" + ",".join(x + "+" for x in steps) + "\t*\n")
        g = hl.GfaGraph(p)
        items, _ = g.item_table(hl.NODE)
        assert items.tolist() == [names.index(x) + 1 for x in steps]   # (the host's own parser resolves them either way)
        return g.name_kind()
    assert kind(["1", "2", "3"]) == (1, "")
    assert kind(["s1", "s2", "s3"], ["s3", "s1", "s2", "s3"]) == (1, "s")
    assert kind(["chr22_1", "chr22_2"]) == (1, "chr22_")
    assert kind(["s2", "s1", "s7"]) == (2, "s")
    assert kind(["5", "9", "2"]) == (2, "")
    assert kind(["s1", "t2", "s3"])[0] == 3             # two prefixes
    assert kind(["s01", "s02"])[0] == 3                  # leading zeros: not numbers
    assert kind(["utg000001l", "utg000002l"])[0] == 3    # the number is not at the end
    assert kind(["s1", "2", "s3"])[0] == 3               # with and without
    assert kind(["abcdefghi1", "abcdefghi2"])[0] == 3    # nine bytes in front of the number
    assert kind(["a", "b"])[0] == 3
    assert kind(["NODE_1_length_1000_cov_12.5", "NODE_2_length_900_cov_3.25"])[0] == 0
