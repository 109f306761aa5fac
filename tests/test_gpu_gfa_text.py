"""pnx_set_csr_gfa: the step columns of P / W lines tokenised on the device (csrc/kernels_gfa.hip) against a plain
Python split of the same text -- what parse_path_seq_to_item_vec / parse_walk_seq_to_item_vec
(src/graph_broker/util.rs:1021-1091) produce on the host -- and the histogram on top against the oracle."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from panacus_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _gfa_text(rng, n_nodes, n_paths, names, walk_share=0.3, long_path=None):
    """-> (text bytes, col_begin, col_end, is_walk, expected steps per path); names[i] = decimal name of node id i + 1"""
    parts = [b"H\tVN:Z:1.1\n"]
    pos = len(parts[0])
    cb, ce, wk, exp = [], [], [], []
    for p in range(n_paths):
        ln = int(rng.integers(1, 400)) if p != long_path else 60_000   # one path far longer than a 16 KB piece
        ids = rng.integers(1, n_nodes + 1, size=ln)
        if p % 7 == 3:
            ids = np.sort(ids)
        back = rng.random(ln) < 0.3
        walk = rng.random() < walk_share
        if walk:
            head = f"W\ts{p}\t1\tctg\t0\t{ln}\t".encode()
            col = "".join(("<" if b else ">") + names[i - 1] for i, b in zip(ids, back)).encode()
            tail = b"\n"
        else:
            head = f"P\ts{p}#1#c{p}\t".encode()
            col = ",".join(names[i - 1] + ("-" if b else "+") for i, b in zip(ids, back)).encode()
            tail = b"\t*\n"
        cb.append(pos + len(head))
        ce.append(pos + len(head) + len(col))
        wk.append(1 if walk else 0)
        exp.append(ids.astype(np.uint32))
        parts += [head, col, tail]
        pos += len(head) + len(col) + len(tail)
    return b"".join(parts), np.array(cb, np.uint64), np.array(ce, np.uint64), np.array(wk, np.uint8), exp


@pytest.mark.parametrize("nice", [True, False])
@pytest.mark.parametrize("upload_first", [False, True])
def test_step_columns_tokenised_on_the_device(ctx, nice, upload_first):
    rng = np.random.default_rng(5 + nice)
    n, P = 9000, 60
    if nice:
        names = [str(i) for i in range(1, n + 1)]
        table = None
    else:   # numeric names in another order, with gaps: the table maps name -> id
        vals = rng.permutation(4 * n)[:n] + 1
        names = [str(int(v)) for v in vals]
        table = np.zeros(4 * n + 2, dtype=np.uint32)
        table[vals] = np.arange(1, n + 1, dtype=np.uint32)
    text, cb, ce, wk, exp = _gfa_text(rng, n, P, names, long_path=17)
    lens = rng.integers(1, 50, size=n + 1).astype(np.uint32)
    ctx.set_csr_gfa(text, cb, ce, wk, n, id_of_name=table, weights=lens, upload_first=upload_first)
    items, off, _ = ctx.get_csr()
    want = np.concatenate(exp)
    assert np.array_equal(off, np.concatenate([[0], np.cumsum([len(e) for e in exp])]).astype(np.uint64))
    assert np.array_equal(items, want)
    pi = np.arange(P, dtype=np.uint64)
    gi = (pi // 3).astype(np.uint64)
    ctx.set_order(pi, gi, P // 3)
    cnt, h = ctx.hist()
    ocov = orc.coverage(want.astype(np.uint64), off, pi, gi, n)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, orc.hist(ocov, P // 3, lens))


@pytest.mark.parametrize("bad", ["name", "unknown", "sign", "empty", "walk", "zero", "trailing", "walk_head", "leading_zero", "long", "walk_zero"])
def test_malformed_steps_fail_the_call(ctx, bad):
    from panacus_amd import capi
    n = 50
    col = {"name": b"1+,2-,s3+,4+", "unknown": b"1+,51+,2+", "sign": b"1+,2,3+", "empty": b"1+,,3+", "walk": b">1<2>x3", "zero": b"1+,0+",
           # what the reference's parser panics on as well (util.rs:1021-1091) and a lenient tokeniser would let through
           "trailing": b"1+,2+,", "walk_head": b"1>2<3", "leading_zero": b"1+,007+,3+", "long": b"1+,12345678901+", "walk_zero": b">1<02"}[bad]
    text = b"P\tp\t" + col + b"\t*\n"
    cb, ce = np.array([4], np.uint64), np.array([4 + len(col)], np.uint64)
    with pytest.raises(capi.PnxError) as e:
        ctx.set_csr_gfa(text, cb, ce, np.array([1 if bad.startswith("walk") else 0], np.uint8), n)
    assert e.value.code == capi.PNX_EINVAL
    with pytest.raises(capi.PnxError):
        ctx.hist()   # nothing is resident after a rejected upload
    # ... and a column outside the text is refused before anything is touched
    with pytest.raises(capi.PnxError):
        ctx.set_csr_gfa(text, cb, ce + np.uint64(100), np.array([0], np.uint8), n)


def _columns(rng, names_of, ids_per_path, walk, pad_char):
    parts, cb, ce = [b"H\tVN:Z:1.1\n"], [], []
    pos = len(parts[0])
    for ids in ids_per_path:
        back = rng.random(len(ids)) < 0.5
        pad = pad_char * int(rng.integers(0, 17))     # moves the column to every alignment of the 16-byte lanes
        if walk:
            head = b"W\ts\t1\t" + pad + b"\t0\t1\t"
            col = "".join(("<" if b else ">") + names_of(i) for i, b in zip(ids, back)).encode()
            tail = b"\n"
        else:
            head = b"P\t" + pad + b"p\t"
            col = ",".join(names_of(i) + ("-" if b else "+") for i, b in zip(ids, back)).encode()
            tail = b"\t*\n"
        cb.append(pos + len(head))
        ce.append(pos + len(head) + len(col))
        parts += [head, col, tail]
        pos += len(head) + len(col) + len(tail)
    return b"".join(parts), np.array(cb, np.uint64), np.array(ce, np.uint64)


@pytest.mark.parametrize("walk", [False, True])
def test_names_of_every_width_and_position(ctx, walk):
    """names of 1..10 digits (up to the largest id a u32 graph holds), starting at every byte offset of the tokeniser's 16-byte
    lanes and across its 1 KB chunks and 16 KB pieces; through a name table and as the ids themselves"""
    from panacus_amd import capi
    rng = np.random.default_rng(3 + walk)
    vals = np.unique(np.concatenate([rng.integers(10 ** (k - 1), 10 ** k, size=60) for k in range(1, 10)] +
                                    [np.array([1, 9, 10, 99, 100, 4294967292, 4294967291, 1000000000, 999999999, 4000000000])]))
    # names below 3 M through a table (ids = ranks)
    small = vals[vals < 3_000_000]
    tab = np.zeros(3_000_001, dtype=np.uint32)
    tab[small] = np.arange(1, len(small) + 1, dtype=np.uint32)
    paths = [rng.choice(small, size=int(rng.integers(1, 9000))) for _ in range(24)]
    text, cb, ce = _columns(rng, lambda i: str(int(i)), paths, walk, b"y")
    ctx.set_csr_gfa(text, cb, ce, np.full(24, 1 if walk else 0, np.uint8), len(small), id_of_name=tab)
    items, off, _ = ctx.get_csr()
    assert np.array_equal(items, np.concatenate([tab[q] for q in paths]))
    assert np.array_equal(off, np.concatenate([[0], np.cumsum([len(q) for q in paths])]).astype(np.uint64))
    # the names ARE the ids: every width up to 10 digits on a graph of 2^32 - 4 items (no rows are derived for it: the
    # one-shot route is asked for, so the upload only validates the ids)
    ctx.config(capi.CFG_COVER_ROUTE, 1)
    try:
        paths = [rng.choice(vals, size=int(rng.integers(1, 5000))) for _ in range(8)]
        text, cb, ce = _columns(rng, lambda i: str(int(i)), paths, walk, b"z")
        ctx.set_csr_gfa(text, cb, ce, np.full(8, 1 if walk else 0, np.uint8), 4294967292)
        items, off, _ = ctx.get_csr()
        assert np.array_equal(items, np.concatenate(paths).astype(np.uint32))
        # one item fewer: the largest name is now unknown
        with pytest.raises(capi.PnxError) as e:
            ctx.set_csr_gfa(text, cb, ce, np.full(8, 1 if walk else 0, np.uint8), 4294967291)
        assert e.value.code == capi.PNX_EINVAL
    finally:
        ctx.config(capi.CFG_COVER_ROUTE, 0)


def _canonical(u, o1, v, o2):
    """Edge::canonical (graph.rs:142-148)"""
    if u > v or (u == v and o1 == 1):
        return (v << 32) | u, ((o2 ^ 1) << 1) | (o1 ^ 1)
    return (u << 32) | v, (o1 << 1) | o2


@pytest.mark.parametrize("nice", [True, False])
def test_edge_item_table_from_walks_that_stay_on_the_device(ctx, nice):
    """pnx_set_csr_gfa with edge_uv / edge_oo: the walks are tokenised on the device and the edge of every consecutive step
    pair is looked up there (parse_path_seq_to_item_vec / parse_walk_seq_to_item_vec with edge2id, util.rs:1048-1091):
    the resident items are edge ids, a path of k steps has k - 1 of them; coverage and histogram against the oracle"""
    rng = np.random.default_rng(11 + nice)
    n, P = 3000, 40
    if nice:
        names = [str(i) for i in range(1, n + 1)]
        table = None
    else:
        vals = rng.permutation(3 * n)[:n] + 1
        names = [str(int(v)) for v in vals]
        table = np.zeros(3 * n + 2, dtype=np.uint32)
        table[vals] = np.arange(1, n + 1, dtype=np.uint32)
    parts, cb, ce, wk, walks = [b"H\tVN:Z:1.1\n"], [], [], [], []
    pos = len(parts[0])
    for p in range(P):
        ln = int(rng.integers(0, 300)) if p % 9 else (1 if p else 0)     # also an empty path and a path of one step
        ids = rng.integers(1, 60, size=ln).cumsum() % n + 1              # local walks: the same edges come back in other paths
        back = (rng.random(ln) < 0.3).astype(np.int64)
        walk = p % 3 == 0
        if walk:
            head = f"W\ts{p}\t1\tctg\t0\t{ln}\t".encode()
            col = "".join(("<" if b else ">") + names[i - 1] for i, b in zip(ids, back)).encode()
            tail = b"\n"
        else:
            head = f"P\ts{p}#1#c{p}\t".encode()
            col = ",".join(names[i - 1] + ("-" if b else "+") for i, b in zip(ids, back)).encode()
            tail = b"\t*\n"
        cb.append(pos + len(head))
        ce.append(pos + len(head) + len(col))
        wk.append(1 if walk else 0)
        walks.append((ids, back))
        parts += [head, col, tail]
        pos += len(head) + len(col) + len(tail)
    text = b"".join(parts)
    # the edges of the graph: every canonical pair once, ids in order of first appearance; plus edges no path uses
    edge_id, uv, oo = {}, [0], [0]
    for ids, back in walks:
        for a in range(len(ids) - 1):
            k = _canonical(int(ids[a]), int(back[a]), int(ids[a + 1]), int(back[a + 1]))
            if k not in edge_id:
                edge_id[k] = len(uv)
                uv.append(k[0])
                oo.append(k[1])
    for extra in range(50):
        k = _canonical(n - extra, 0, n - extra, 0)
        if k not in edge_id:
            edge_id[k] = len(uv)
            uv.append(k[0])
            oo.append(k[1])
    want = [np.array([edge_id[_canonical(int(ids[a]), int(back[a]), int(ids[a + 1]), int(back[a + 1]))] for a in range(len(ids) - 1)],
                     dtype=np.uint32) for ids, back in walks]
    E = len(uv) - 1
    ctx.set_csr_gfa(text, np.array(cb, np.uint64), np.array(ce, np.uint64), np.array(wk, np.uint8), n, id_of_name=table,
                    edge_uv=np.array(uv, np.uint64), edge_oo=np.array(oo, np.uint8))
    items, off, _ = ctx.get_csr()
    assert np.array_equal(off, np.concatenate([[0], np.cumsum([len(w) for w in want])]).astype(np.uint64))
    assert np.array_equal(items, np.concatenate(want)) and ctx.info().n_items == E
    pi = np.arange(P, dtype=np.uint64)
    gi = (pi // 2).astype(np.uint64)
    ctx.set_order(pi, gi, P // 2)
    cnt, h = ctx.hist()
    ocov = orc.coverage(np.concatenate(want).astype(np.uint64), off, pi, gi, E)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, orc.hist(ocov, P // 2))
    # `-c all`: the text tokenised once into walks that stay on the device, the node table and the edge table both made from them
    woff = ctx.gfa_walks(text, np.array(cb, np.uint64), np.array(ce, np.uint64), np.array(wk, np.uint8), n, id_of_name=table)
    assert np.array_equal(woff, np.concatenate([[0], np.cumsum([len(ids) for ids, _ in walks])]).astype(np.uint64))
    for _ in range(2):
        ctx.set_csr_walks(n)
        it_n, off_n, _ = ctx.get_csr()
        assert np.array_equal(it_n, np.concatenate([ids for ids, _ in walks]).astype(np.uint32)) and np.array_equal(off_n, woff)
        ctx.set_csr_walks(n, edge_uv=np.array(uv, np.uint64), edge_oo=np.array(oo, np.uint8))
        it_e, off_e, _ = ctx.get_csr()
        assert np.array_equal(it_e, np.concatenate(want)) and np.array_equal(off_e, off) and ctx.info().n_items == E
    ctx.set_order(pi, gi, P // 2)
    cnt2, h2 = ctx.hist()
    assert np.array_equal(cnt2, ocov) and np.array_equal(h2, orc.hist(ocov, P // 2))
    # a step pair the graph has no edge for fails the call (the reference panics, util.rs:1080); so does an edge that is not canonical
    from panacus_amd import capi
    with pytest.raises(capi.PnxError) as e:
        ctx.set_csr_gfa(text, np.array(cb, np.uint64), np.array(ce, np.uint64), np.array(wk, np.uint8), n, id_of_name=table,
                        edge_uv=np.array(uv[:-60], np.uint64), edge_oo=np.array(oo[:-60], np.uint8))
    assert e.value.code == capi.PNX_EINVAL and "not joined by an edge" in str(e.value)
    bad = np.array(uv, np.uint64)
    bad[1] = (np.uint64(5) << np.uint64(32)) | np.uint64(2)
    with pytest.raises(capi.PnxError):
        ctx.set_csr_gfa(text, np.array(cb, np.uint64), np.array(ce, np.uint64), np.array(wk, np.uint8), n, id_of_name=table,
                        edge_uv=bad, edge_oo=np.array(oo, np.uint8))
    with pytest.raises(capi.PnxError):
        ctx.hist()   # nothing is resident after a rejected upload


def _named_gfa(rng, n, P, style, walk_share=0.3, with_links=False, dup_links=0):
    """A whole GFA text with S (and L) lines in front of the paths.  style: how segments are named.
    -> text, name_off, name_len, col_begin, col_end, is_walk, walks [(ids, back)], link_off, links [(u, o1, v, o2)] in file order"""
    if style == "mixed":
        pool = [lambda i: f"s{i}", lambda i: f"utg{i:06d}l", lambda i: f"n{i}_x", lambda i: "ABCDEFGHIJKLMNOP"[: 1 + i % 16] + str(i)]
        names = [pool[i % 4](i)[:16] for i in range(1, n + 1)]
        seen_names = set()
        for k, nm in enumerate(names):  # (unique)
            if nm in seen_names:
                names[k] = nm = f"q{k}"
            seen_names.add(nm)
        assert len(set(names)) == n and max(map(len, names)) <= 16
    elif style == "nice":
        names = [str(i) for i in range(1, n + 1)]
    elif style == "prefix":        # minigraph-cactus: s1, s2, ...
        names = [f"s{i}" for i in range(1, n + 1)]
    elif style == "prefix_table":  # a longer prefix, numbers that are not the ranks
        vals = rng.permutation(4 * n)[:n] + 1
        names = [f"chr22_{int(v)}" for v in vals]
    else:  # numeric through a table
        vals = rng.permutation(4 * n)[:n] + 1
        names = [str(int(v)) for v in vals]
    parts, pos = [b"H\tVN:Z:1.1\n"], 11
    name_off, name_len = [], []
    for nm in names:
        head = b"S\t"
        seq = b"ACGT"[: 1 + len(nm) % 4]
        name_off.append(pos + len(head))
        name_len.append(len(nm))
        line = head + nm.encode() + b"\t" + seq + b"\n"
        parts.append(line)
        pos += len(line)
    walks, cb, ce, wk = [], [], [], []
    for p in range(P):
        ln = int(rng.integers(0, 300)) if p % 9 else (1 if p else 0)
        ids = rng.integers(1, 60, size=ln).cumsum() % n + 1
        back = (rng.random(ln) < 0.3).astype(np.int64)
        walks.append((ids, back))
    if with_links and P > 3:   # a hub: node 3 joined to 40 others, walked back and forth (its edge list is bisected, not scanned)
        others = rng.permutation(np.arange(4, n + 1))[:40]
        ids = np.empty(80, dtype=np.int64)
        ids[0::2], ids[1::2] = 3, others
        walks[3] = (ids, (rng.random(80) < 0.5).astype(np.int64))
    links, link_off = [], []
    if with_links:
        seen = []
        for ids, back in walks:
            for a in range(len(ids) - 1):
                seen.append((int(ids[a]), int(back[a]), int(ids[a + 1]), int(back[a + 1])))
        seen += [(n - k, 0, n - k, 0) for k in range(20)] + [(5, 1, 5, 1), (7, 0, 7, 1)]   # edges no path uses, self loops
        order = rng.permutation(len(seen))
        for k in order:
            u, o1, v, o2 = seen[k]
            if rng.random() < 0.5:   # the same edge written from its other end
                u, o1, v, o2 = v, o2 ^ 1, u, o1 ^ 1
            links.append((u, o1, v, o2))
        for _ in range(dup_links):
            links.insert(int(rng.integers(0, len(links))), links[int(rng.integers(0, len(links)))])
        for (u, o1, v, o2) in links:
            line = f"L\t{names[u - 1]}\t{'-' if o1 else '+'}\t{names[v - 1]}\t{'-' if o2 else '+'}\t0M\n".encode()
            link_off.append(pos)
            parts.append(line)
            pos += len(line)
    for p, (ids, back) in enumerate(walks):
        walk = rng.random() < walk_share
        if walk:
            head = f"W\ts{p}\t1\tctg\t0\t{len(ids)}\t".encode()
            col = "".join(("<" if b else ">") + names[i - 1] for i, b in zip(ids, back)).encode()
            tail = b"\n"
        else:
            head = f"P\tp{p}#1#c\t".encode()
            col = ",".join(names[i - 1] + ("-" if b else "+") for i, b in zip(ids, back)).encode()
            tail = b"\t*\n"
        cb.append(pos + len(head))
        ce.append(pos + len(head) + len(col))
        wk.append(1 if walk else 0)
        parts += [head, col, tail]
        pos += len(head) + len(col) + len(tail)
    text = b"".join(parts)
    assert len(text) == pos
    table = None
    if style in ("table", "prefix_table"):
        table = np.zeros(4 * n + 2, dtype=np.uint32)
        table[np.array([int(x.split("_")[-1]) for x in names])] = np.arange(1, n + 1, dtype=np.uint32)
    return dict(text=text, names=names, name_off=np.array(name_off, np.uint64), name_len=np.array(name_len, np.uint8),
                cb=np.array(cb, np.uint64), ce=np.array(ce, np.uint64), wk=np.array(wk, np.uint8), walks=walks,
                link_off=np.array(link_off, np.uint64), links=links, table=table)


def test_segment_names_that_are_not_numbers(ctx):
    """names like s12 / utg000012l / up to 16 bytes: node2id is a hash table in HBM keyed by the name bytes (graph.rs:308-375),
    the tokeniser looks every step up there (graph.rs:231) -- same ItemTable as a plain split of the text, histogram against the oracle"""
    from panacus_amd import capi
    rng = np.random.default_rng(21)
    n, P = 7000, 50
    g = _named_gfa(rng, n, P, "mixed")
    lens = rng.integers(1, 50, size=n + 1).astype(np.uint32)
    ctx.set_csr_gfa(g["text"], g["cb"], g["ce"], g["wk"], n, weights=lens, name_off=g["name_off"], name_len=g["name_len"])
    items, off, _ = ctx.get_csr()
    want = np.concatenate([ids for ids, _ in g["walks"]]).astype(np.uint32)
    assert np.array_equal(items, want)
    assert np.array_equal(off, np.concatenate([[0], np.cumsum([len(ids) for ids, _ in g["walks"]])]).astype(np.uint64))
    pi = np.arange(P, dtype=np.uint64)
    gi = (pi // 5).astype(np.uint64)
    ctx.set_order(pi, gi, P // 5)
    cnt, h = ctx.hist()
    ocov = orc.coverage(want.astype(np.uint64), off, pi, gi, n)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, orc.hist(ocov, P // 5, lens))
    # PNX_NAMES_FIND: the library finds the S lines itself -- in the whole text, and in the byte range that holds them
    s_lo, s_hi = int(g["name_off"][0]) - 2, int(g["name_off"][-1]) + 20
    for find in (True, (s_lo, s_hi), (0, s_hi)):
        ctx.set_csr_gfa(g["text"], g["cb"], g["ce"], g["wk"], n, find_names=find)
        it_f, off_f, _ = ctx.get_csr()
        assert np.array_equal(it_f, want) and np.array_equal(off_f, off), find
    with pytest.raises(capi.PnxError) as e:   # a range that misses S lines: their number is not n_nodes
        ctx.set_csr_gfa(g["text"], g["cb"], g["ce"], g["wk"], n, find_names=(s_lo, (s_lo + s_hi) // 2))
    assert e.value.code == capi.PNX_EINVAL and "S lines" in str(e.value)
    # a name that occurs twice (the reference panics, graph.rs:336), a name of 17 bytes, a step that names no segment, an empty name
    t = bytearray(g["text"])
    o0, o1 = int(g["name_off"][0]), int(g["name_off"][4])
    l0 = int(g["name_len"][0])
    dup_len = g["name_len"].copy()
    dup_off = g["name_off"].copy()
    dup_off[4], dup_len[4] = o0, l0          # segment 5 carries segment 1's name
    with pytest.raises(capi.PnxError) as e:
        ctx.set_csr_gfa(g["text"], g["cb"], g["ce"], g["wk"], n, name_off=dup_off, name_len=dup_len)
    assert e.value.code == capi.PNX_EINVAL and "more than once" in str(e.value)
    long_len = g["name_len"].copy()
    long_len[3] = 17
    with pytest.raises(capi.PnxError) as e:
        ctx.set_csr_gfa(g["text"], g["cb"], g["ce"], g["wk"], n, name_off=g["name_off"], name_len=long_len)
    assert e.value.code == capi.PNX_ELIMIT
    col = b"s1+,nosuchsegment+,s2+"
    text = g["text"] + b"P\tx\t" + col + b"\t*\n"
    cb = np.array([len(g["text"]) + 4], np.uint64)
    with pytest.raises(capi.PnxError) as e:
        ctx.set_csr_gfa(text, cb, cb + np.uint64(len(col)), np.array([0], np.uint8), n, name_off=g["name_off"], name_len=g["name_len"])
    assert e.value.code == capi.PNX_EINVAL
    col = b"s1+,+,s2+"
    text = g["text"] + b"P\tx\t" + col + b"\t*\n"
    with pytest.raises(capi.PnxError) as e:
        ctx.set_csr_gfa(text, cb, cb + np.uint64(len(col)), np.array([0], np.uint8), n, name_off=g["name_off"], name_len=g["name_len"])
    assert e.value.code == capi.PNX_EINVAL
    with pytest.raises(capi.PnxError):
        ctx.hist()   # nothing is resident after a rejected upload


@pytest.mark.parametrize("style", ["nice", "table", "mixed", "prefix", "prefix_table"])
def test_l_lines_parsed_on_the_device(ctx, style):
    """edge counts without the host's edge map: the library parses the L lines (graph.rs:276-306), numbers the distinct
    canonical edges by their first line -- duplicates and edges written from their other end are skipped like the reference
    skips them (graph.rs:296) -- and looks the step pairs up; the edge ItemTable must be the one the reference's ids give"""
    from panacus_amd import capi
    rng = np.random.default_rng(31 + len(style))
    n, P = 2500, 40
    g = _named_gfa(rng, n, P, style, with_links=True, dup_links=30)
    edge_id = {}
    for (u, o1, v, o2) in g["links"]:           # ids = ranks of the first occurrences, in file order
        k = _canonical(u, o1, v, o2)
        if k not in edge_id:
            edge_id[k] = len(edge_id) + 1
    E = len(edge_id)
    want = [np.array([edge_id[_canonical(int(ids[a]), int(back[a]), int(ids[a + 1]), int(back[a + 1]))] for a in range(len(ids) - 1)],
                     dtype=np.uint32) for ids, back in g["walks"]]
    kw = dict(link_off=g["link_off"])
    if style == "mixed":
        kw.update(name_off=g["name_off"], name_len=g["name_len"])
    elif style == "table":
        kw.update(id_of_name=g["table"])
    elif style == "prefix":          # the number behind the prefix IS the id
        kw.update(name_prefix=b"s")
    elif style == "prefix_table":
        kw.update(id_of_name=g["table"], name_prefix=b"chr22_")
    ctx.set_csr_gfa(g["text"], g["cb"], g["ce"], g["wk"], n, **kw)
    items, off, _ = ctx.get_csr()
    assert ctx.info().n_items == E
    assert np.array_equal(off, np.concatenate([[0], np.cumsum([len(w) for w in want])]).astype(np.uint64))
    assert np.array_equal(items, np.concatenate(want))
    pi = np.arange(P, dtype=np.uint64)
    gi = (pi // 2).astype(np.uint64)
    ctx.set_order(pi, gi, P // 2)
    cnt, h = ctx.hist()
    ocov = orc.coverage(np.concatenate(want).astype(np.uint64), off, pi, gi, E)
    assert np.array_equal(cnt, ocov) and np.array_equal(h, orc.hist(ocov, P // 2))
    # `-c all`: walks + links tokenised once, node table and edge table from them
    woff = ctx.gfa_walks(g["text"], g["cb"], g["ce"], g["wk"], n, **kw)
    ctx.set_csr_walks(n)
    it_n, _, _ = ctx.get_csr()
    assert np.array_equal(it_n, np.concatenate([ids for ids, _ in g["walks"]]).astype(np.uint32))
    ctx.set_csr_walks(n, edges_from_links=True)
    it_e, off_e, _ = ctx.get_csr()
    assert np.array_equal(it_e, np.concatenate(want)) and np.array_equal(off_e, off) and ctx.info().n_items == E
    # the library finds the L lines itself (PNX_LINKS_FIND): in the whole text, in the byte range that holds them, and in a
    # text whose L lines stand between other lines
    kf = {k: v for k, v in kw.items() if k != "link_off"}
    if style == "mixed":   # (and the S lines with them)
        kf = dict(find_names=(int(g["name_off"][0]) - 2, int(g["link_off"][0])))
    l_lo, l_hi = int(g["link_off"][0]), int(g["link_off"][-1]) + 8
    for find in (True, (l_lo, l_hi), (l_lo, len(g["text"])), (0, l_hi)):
        ctx.set_csr_gfa(g["text"], g["cb"], g["ce"], g["wk"], n, find_links=find, **kf)
        it_f, off_f, _ = ctx.get_csr()
        assert np.array_equal(it_f, np.concatenate(want)) and np.array_equal(off_f, off) and ctx.info().n_items == E, find
    with pytest.raises(capi.PnxError):   # a range that cuts the L lines short: steps without an edge
        ctx.set_csr_gfa(g["text"], g["cb"], g["ce"], g["wk"], n, find_links=(l_lo, (l_lo + l_hi) // 2), **kf)
    # an L line that names no segment; a malformed one
    pre = {"prefix": b"s", "prefix_table": b"chr22_"}.get(style, b"")
    if pre:   # a step whose name lacks the prefix, or carries another one: malformed, not segment 7
        col = b"7+," + g["names"][1].encode() + b"+"
        text = g["text"] + b"P\tx\t" + col + b"\t*\n"
        cb1 = np.array([len(g["text"]) + 4], np.uint64)
        for c in (col, col.replace(b"7+", b"t7+")):
            t2 = g["text"] + b"P\tx\t" + c + b"\t*\n"
            with pytest.raises(capi.PnxError) as e:
                ctx.set_csr_gfa(t2, cb1, cb1 + np.uint64(len(c)), np.array([0], np.uint8), n, **{k: v for k, v in kw.items() if k != "link_off"})
            assert e.value.code == capi.PNX_EINVAL
    bad = g["text"] + b"L\t" + (b"zz9" if style == "mixed" else pre + b"99999999") + b"\t+\t" + g["names"][0].encode() + b"\t+\t0M\n"
    lo = np.concatenate([g["link_off"], [len(g["text"])]]).astype(np.uint64)
    with pytest.raises(capi.PnxError) as e:
        ctx.set_csr_gfa(bad, g["cb"], g["ce"], g["wk"], n, **dict(kw, link_off=lo))
    assert e.value.code == capi.PNX_EINVAL
    bad = g["text"] + b"L\t" + g["names"][0].encode() + b"\n"
    with pytest.raises(capi.PnxError) as e:
        ctx.set_csr_gfa(bad, g["cb"], g["ce"], g["wk"], n, **dict(kw, link_off=lo))
    assert e.value.code == capi.PNX_EINVAL


def test_edges_among_leading_zero_length_nodes(tmp_path):
    """update_tables_edgecount (util.rs:723-795) includes an edge only where `include_coords[0].0 < p + l`: the edges among the
    leading zero-length nodes of a path that starts at 0 are not in the reference's table.  The CLI takes the cut route for
    edge counts on such a graph (the device's plain edge route has k - 1 edges for k steps); found by the CLI fuzz, seed 171"""
    from panacus_amd import hostlib as hl
    gfa = str(tmp_path / "z.gfa")
    with open(gfa, "w") as f:
        f.write("H\tVN:Z:1.0\n" "S\t1\t\n" "S\t2\t\n" "S\t3\tACGT\n" "S\t4\t\n" "S\t5\tAC\n"
                "L\t1\t+\t2\t+\t0M\n" "L\t2\t+\t3\t+\t0M\n" "L\t3\t+\t4\t+\t0M\n" "L\t4\t+\t5\t+\t0M\n" "L\t1\t+\t1\t+\t0M\n" "L\t2\t+\t4\t-\t0M\n"
                "P\ta#1#c\t1+,1+,2+,3+,4+,5+\t*\n"          # 1-1 and 1-2 are left out (p + l == 0), 2-3 on are in
                "P\tb#1#c:7-90\t1+,2+,3+\t*\n"               # starts at 7: nothing is left out
                "P\tc#1#c\t3+,4+,5+\t*\n"
                "W\td\t1\tc\t0\t9\t>2<4\n"                  # 2+ -> 4-: both of length 0, left out
                "P\te#1#c\t4+,2-\t*\n")                       # the same edge from its other end: left out too
    g = orc.Graph(gfa, index_edges=True)
    items, pre = g.item_table(orc.EDGE)
    assert np.diff(pre).tolist() == [3, 2, 2, 0, 0]
    pi, gi, names = g.path_order(orc.GROUP_PATHID)
    want = orc.hist(orc.coverage(items, pre, pi, gi, g.n_items(orc.EDGE)), len(names)).tolist()
    for args in (["hist", "-c", "edge", gfa], ["hist", "-c", "all", gfa]):
        rc, out, err = hl.run_cli(args)
        assert rc == 0, err
        rows = [l.split("\t") for l in out.split("\n") if l and not l.startswith("#")]
        j = [k for k in range(1, len(rows[0])) if rows[1][k] == "edge"][0]
        assert [int(r[j]) for r in rows[4:]] == want, args
