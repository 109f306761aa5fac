// gfa_graph.cpp -- see gfa_graph.hpp.
#include "gfa_graph.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <regex>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <stdexcept>

#include "thread_pool.hpp"

namespace pnh {

// ------------------------------------------------------------------------------------------
// PathSegment
// ------------------------------------------------------------------------------------------
namespace {

bool parse_u64(std::string_view s, uint64_t &out) {  // usize::from_str(..).ok()
    if (s.empty()) return false;
    uint64_t v = 0;
    for (char ch : s) {
        if (ch < '0' || ch > '9') return false;
        uint64_t d = (uint64_t)(ch - '0');
        if (v > (UINT64_MAX - d) / 10) return false;
        v = v * 10 + d;
    }
    out = v;
    return true;
}

// PATHID_COORDS = ^(.+):([0-9]+)-([0-9]+)$  (graph.rs:18); returns length of group 1 or npos
size_t match_coords(std::string_view s, bool &hs, uint64_t &st, bool &he, uint64_t &en) {
    size_t i = s.size();
    while (i > 0 && s[i - 1] >= '0' && s[i - 1] <= '9') --i;
    if (i == s.size() || i == 0 || s[i - 1] != '-') return std::string_view::npos;
    const size_t dash = i - 1;
    size_t j = dash;
    while (j > 0 && s[j - 1] >= '0' && s[j - 1] <= '9') --j;
    if (j == dash || j == 0 || s[j - 1] != ':') return std::string_view::npos;
    const size_t colon = j - 1;
    if (colon == 0) return std::string_view::npos;
    hs = parse_u64(s.substr(j, dash - j), st);
    he = parse_u64(s.substr(dash + 1), en);
    return colon;
}

}  // namespace

// PATHID_PANSN = ^([^#]+)(#[^#]+)?(#[^#].*)?$ with leftmost-first (Perl-like) semantics
PathSegment PathSegment::from_str(std::string_view s) {
    PathSegment r;
    r.sample = std::string(s);
    const size_t n = s.size();
    size_t a = 0;
    while (a < n && s[a] != '#') ++a;
    if (a == 0) return r;  // no match: whole string is the sample
    std::string_view g2, g3;
    bool matched = false;
    if (a == n) {
        matched = true;
    } else {
        size_t b = a + 1;
        while (b < n && s[b] != '#') ++b;
        if (b > a + 1) {  // group 2 = '#' + maximal run of non-'#'
            if (b == n) {
                g2 = s.substr(a, b - a);
                matched = true;
            } else if (b + 1 < n && s[b + 1] != '#') {
                g2 = s.substr(a, b - a);
                g3 = s.substr(b);
                matched = true;
            }
        }
        if (!matched && a + 1 < n && s[a + 1] != '#') {  // backtrack: no group 2, group 3 from a
            g3 = s.substr(a);
            matched = true;
        }
    }
    if (!matched) return r;
    const int nseg = 2 + (g2.empty() ? 0 : 1) + (g3.empty() ? 0 : 1);
    bool hs = false, he = false;
    uint64_t st = 0, en = 0;
    if (nseg == 4) {
        r.sample = std::string(s.substr(0, a));
        r.has_haplotype = true;
        r.haplotype = std::string(g2.substr(1));
        std::string_view rest = g3.substr(1);
        size_t g1 = match_coords(rest, hs, st, he, en);
        r.has_seqid = true;
        if (g1 == std::string_view::npos) {
            r.seqid = std::string(rest);
        } else {
            r.seqid = std::string(rest.substr(0, g1));
            r.has_start = hs; r.start = st; r.has_end = he; r.end = en;
        }
    } else if (nseg == 3) {
        std::string_view seg = (g2.empty() ? g3 : g2).substr(1);
        r.sample = std::string(s.substr(0, a));
        r.has_haplotype = true;
        size_t g1 = match_coords(seg, hs, st, he, en);
        if (g1 == std::string_view::npos) {
            r.haplotype = std::string(seg);
        } else {
            r.haplotype = std::string(seg.substr(0, g1));
            r.has_start = hs; r.start = st; r.has_end = he; r.end = en;
        }
    } else {
        size_t g1 = match_coords(s.substr(0, a), hs, st, he, en);
        if (g1 != std::string_view::npos) {
            r.sample = std::string(s.substr(0, g1));
            r.has_start = hs; r.start = st; r.has_end = he; r.end = en;
        }
    }
    return r;
}

std::string PathSegment::id() const {
    if (has_haplotype) return has_seqid ? sample + "#" + haplotype + "#" + seqid : sample + "#" + haplotype;
    if (has_seqid) return sample + "#*#" + seqid;
    return sample;
}

std::string PathSegment::display() const {
    if (has_start && has_end) return id() + ":" + std::to_string(start) + "-" + std::to_string(end);
    return id();
}

std::string PathSegment::clear_key() const {
    std::string k = sample;
    k += '\x01';
    k += has_haplotype ? '1' : '0';
    k += haplotype;
    k += '\x01';
    k += has_seqid ? '1' : '0';
    k += seqid;
    return k;
}

// ------------------------------------------------------------------------------------------
// implementation state: the file image and the id maps
// ------------------------------------------------------------------------------------------
namespace {

inline uint64_t hash_bytes(const char *p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xff51afd7ed558ccdULL);
    while (n >= 8) {
        uint64_t w;
        std::memcpy(&w, p, 8);
        h = (h ^ w) * 0x9FB21C651E98DF25ull;
        h ^= h >> 32;
        p += 8;
        n -= 8;
    }
    uint64_t w = 0;
    std::memcpy(&w, p, n);
    h = (h ^ w) * 0x9FB21C651E98DF25ull;
    return h ^ (h >> 29);
}

struct NameMap {  // segment name -> node id; keys are views into the file image
    struct Slot {
        const char *p;
        uint32_t len, id;
    };
    std::vector<Slot> tab;
    uint64_t mask = 0;
    void init(size_t n) {
        size_t cap = 64;
        while (cap < n * 2) cap <<= 1;
        tab.assign(cap, Slot{nullptr, 0, 0});
        mask = cap - 1;
    }
    bool insert(const char *p, uint32_t len, uint32_t id) {
        uint64_t j = hash_bytes(p, len) & mask;
        while (tab[j].p) {
            if (tab[j].len == len && std::memcmp(tab[j].p, p, len) == 0) return false;
            j = (j + 1) & mask;
        }
        tab[j] = Slot{p, len, id};
        return true;
    }
    uint32_t find(const char *p, uint32_t len) const {
        uint64_t j = hash_bytes(p, len) & mask;
        while (tab[j].p) {
            if (tab[j].len == len && std::memcmp(tab[j].p, p, len) == 0) return tab[j].id;
            j = (j + 1) & mask;
        }
        return 0;
    }
};

struct EdgeMap {  // canonical (u, o1, v, o2) -> edge id
    struct Slot {
        uint64_t uv;
        uint32_t id;
        uint8_t oo;
    };
    std::vector<Slot> tab;
    uint64_t mask = 0;
    size_t n = 0;
    void init(size_t want) {
        size_t cap = 64;
        while (cap < want * 2) cap <<= 1;
        tab.assign(cap, Slot{0, 0, 0});
        mask = cap - 1;
        n = 0;
    }
    // the same with the table cleared by the pool's threads: 268 MB of first-touch page faults for 6 M edges are 60-80 ms on
    // one thread
    void init_parallel(size_t want);
    static uint64_t h(uint64_t uv, uint8_t oo) {
        uint64_t x = (uv ^ ((uint64_t)oo << 62)) * 0x9FB21C651E98DF25ull;
        return x ^ (x >> 31);
    }
    uint32_t find(uint64_t uv, uint8_t oo) const {
        uint64_t j = h(uv, oo) & mask;
        while (tab[j].id) {
            if (tab[j].uv == uv && tab[j].oo == oo) return tab[j].id;
            j = (j + 1) & mask;
        }
        return 0;
    }
    bool insert(uint64_t uv, uint8_t oo, uint32_t id) {
        uint64_t j = h(uv, oo) & mask;
        while (tab[j].id) {
            if (tab[j].uv == uv && tab[j].oo == oo) return false;
            j = (j + 1) & mask;
        }
        tab[j] = Slot{uv, id, oo};
        ++n;
        return true;
    }
};

void EdgeMap::init_parallel(size_t want) {
    size_t cap = 64;
    while (cap < want * 2) cap <<= 1;
    std::vector<Slot> fresh;
    fresh.reserve(cap);  // untouched pages: the pool's threads fault them in, the assignment below then only writes
    char *base = reinterpret_cast<char *>(fresh.data());
    const size_t bytes = cap * sizeof(Slot), slice = (size_t)16 << 20;
    ThreadPool::instance().parallel_for((bytes + slice - 1) / slice, [&](size_t t) {
        const size_t e = std::min(bytes, (t + 1) * slice);
        std::memset(base + t * slice, 0, e - t * slice);
    });
    fresh.assign(cap, Slot{0, 0, 0});
    tab.swap(fresh);
    mask = cap - 1;
    n = 0;
}

// Edge::canonical (graph.rs:142-148); orientation 0 = Forward, 1 = Backward
inline void canonical(uint32_t u, uint8_t o1, uint32_t v, uint8_t o2, uint64_t &uv, uint8_t &oo) {
    if (u > v || (u == v && o1 == 1)) {
        uv = ((uint64_t)v << 32) | u;
        oo = (uint8_t)(((o2 ^ 1) << 1) | (o1 ^ 1));
    } else {
        uv = ((uint64_t)u << 32) | v;
        oo = (uint8_t)((o1 << 1) | o2);
    }
}

// The GFA as one contiguous read-only byte range: plain files are mmap'ed (no copy), gzip
// files are inflated into an owned buffer (bufreader_from_compressed_gfa, io.rs:23-33).
struct Image {
    const char *p = nullptr;
    size_t n = 0;
    std::string owned;
    void *map = nullptr;
    size_t map_len = 0;

    const char *data() const { return p; }
    size_t size() const { return n; }
    char operator[](size_t i) const { return p[i]; }
    std::string substr(size_t a, size_t len) const { return std::string(p + a, std::min(len, n - a)); }
    std::string_view view(size_t a, size_t len) const { return std::string_view(p + a, len); }

    void open(const std::string &path) {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        unsigned char magic[2] = {0, 0};
        ssize_t got = ::pread(fd, magic, 2, 0);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            if (open_bgzf(fd)) {  // block gzip (bgzip, htslib): the blocks are inflated in parallel
                ::close(fd);
                return;
            }
            ::close(fd);
            gzFile g = gzopen(path.c_str(), "rb");
            if (!g) throw std::runtime_error("cannot open " + path);
            gzbuffer(g, 1 << 20);
            std::vector<char> chunk(1 << 22);
            int r;
            while ((r = gzread(g, chunk.data(), (unsigned)chunk.size())) > 0) owned.append(chunk.data(), (size_t)r);
            gzclose(g);
            if (r < 0) throw std::runtime_error("error while decompressing " + path);
            p = owned.data();
            n = owned.size();
            return;
        }
        struct stat st;
        if (fstat(fd, &st) != 0) {
            ::close(fd);
            throw std::runtime_error("cannot stat " + path);
        }
        n = (size_t)st.st_size;
        if (n) {
            map = mmap(nullptr, n, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (map == MAP_FAILED) {
                map = nullptr;
                ::close(fd);
                throw std::runtime_error("cannot map " + path);
            }
            map_len = n;
            p = static_cast<const char *>(map);
        }
        ::close(fd);
    }
    // BGZF (the block gzip of bgzip / htslib, SAM specification section 4.1): a series of gzip members of at most 64 KiB,
    // each with an extra field "BC" that holds its compressed size, so the members can be found without inflating and
    // inflated independently -- one raw-deflate stream per block, all blocks in parallel over the worker pool.  A plain
    // gzip stream has no such index and is inflated serially (below).  Returns false if the file is not BGZF.
    bool open_bgzf(int fd) {
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size < 28) return false;
        const size_t zn = (size_t)st.st_size;
        void *zmap = mmap(nullptr, zn, PROT_READ, MAP_PRIVATE, fd, 0);
        if (zmap == MAP_FAILED) return false;
        const unsigned char *z = static_cast<const unsigned char *>(zmap);
        struct Block {
            size_t data, data_len, out;
            uint32_t isize, crc;
        };
        std::vector<Block> blocks;
        size_t at = 0, total = 0;
        bool ok = true;
        while (at < zn) {
            // ID1 ID2 CM FLG MTIME(4) XFL OS XLEN(2) | extra subfields | deflate data | CRC32 ISIZE
            if (at + 18 > zn || z[at] != 0x1f || z[at + 1] != 0x8b || z[at + 2] != 8 || !(z[at + 3] & 4)) {
                ok = false;
                break;
            }
            const size_t xlen = z[at + 10] | ((size_t)z[at + 11] << 8);
            size_t x = at + 12, xend = x + xlen, bsize = 0;
            if (xend > zn) {
                ok = false;
                break;
            }
            while (x + 4 <= xend) {
                const size_t slen = z[x + 2] | ((size_t)z[x + 3] << 8);
                if (z[x] == 'B' && z[x + 1] == 'C' && slen == 2 && x + 6 <= xend) bsize = (z[x + 4] | ((size_t)z[x + 5] << 8)) + 1;
                x += 4 + slen;
            }
            if (!bsize || at + bsize > zn || bsize < xlen + 20 || (z[at + 3] & ~4)) {  // other header flags: not what bgzip writes
                ok = false;
                break;
            }
            const unsigned char *tail = z + at + bsize - 8;
            Block b;
            b.data = at + 12 + xlen;
            b.data_len = bsize - xlen - 20;
            b.crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
            b.isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
            b.out = total;
            total += b.isize;
            blocks.push_back(b);
            at += bsize;
        }
        if (!ok || blocks.empty()) {
            munmap(zmap, zn);
            if (!blocks.empty()) throw std::runtime_error("damaged BGZF file (block " + std::to_string(blocks.size()) + ")");
            return false;
        }
        owned.resize(total);
        std::atomic<int64_t> bad{-1};
        const size_t GROUP = 16;  // blocks per task
        ThreadPool::instance().parallel_for((blocks.size() + GROUP - 1) / GROUP, [&](size_t t) {
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) {
                bad.store((int64_t)(t * GROUP));
                return;
            }
            const size_t hi = std::min(blocks.size(), (t + 1) * GROUP);
            for (size_t k = t * GROUP; k < hi; ++k) {
                const Block &b = blocks[k];
                if (!b.isize) continue;
                inflateReset(&zs);
                zs.next_in = const_cast<unsigned char *>(z + b.data);
                zs.avail_in = (uInt)b.data_len;
                zs.next_out = reinterpret_cast<unsigned char *>(&owned[b.out]);
                zs.avail_out = b.isize;
                const int r = inflate(&zs, Z_FINISH);
                if (r != Z_STREAM_END || zs.avail_out != 0 ||
                    crc32(0L, reinterpret_cast<const unsigned char *>(&owned[b.out]), b.isize) != b.crc) {
                    bad.store((int64_t)k);
                    break;
                }
            }
            inflateEnd(&zs);
        });
        munmap(zmap, zn);
        if (bad.load() >= 0) throw std::runtime_error("damaged BGZF block " + std::to_string(bad.load()));
        p = owned.data();
        n = owned.size();
        return true;
    }
    ~Image() {
        if (map) munmap(map, map_len);
    }
    Image() = default;
    Image(const Image &) = delete;
    Image &operator=(const Image &) = delete;
};

struct Span {
    size_t b, e;  // [b, e) in the file image
};

inline size_t field_end(const Image &s, size_t from, size_t line_end) {
    const void *p = std::memchr(s.data() + from, '\t', line_end - from);
    return p ? (size_t)((const char *)p - s.data()) : line_end;
}

}  // namespace

struct GraphStorage::Impl {
    Image image;                      // the whole GFA
    std::vector<Span> p_lines;        // P and W lines in file order
    std::vector<Span> step_fields;    // per path: the step / walk column
    std::vector<uint8_t> is_walk;     // per path
    mutable NameMap names;            // name -> id for names that are not the ranks: built on first use (ensure_names)
    mutable std::once_flag names_once;
    std::vector<Span> node_names;     // per node id - 1: the name field of its S line
    uint32_t max_name_len = 0;        // longest segment name in bytes
    std::vector<Span> l_lines;        // the L lines (from_gfa with index_edges; else collected when the index is asked for)
    bool links_only = false;          // the L lines were seen, not kept: the device finds and parses them (PNX_LINKS_FIND)
    uint64_t link_lo = 0, link_hi = 0;  // links_only: the bytes from the first L line to the end of the last one
    uint64_t seg_lo = 0, seg_hi = 0;    // the bytes from the first S line to the end of the last one
    bool nice = false;                // segment names are the integers 1..N in file order
    // every segment name is name_prefix (at most 8 bytes, the same for all; empty for pggb's plain numbers, "s" for
    // minigraph-cactus' s1, s2, ...) followed by a decimal number: the number IS the id (number_is_rank) or id_of_name maps it
    bool numeric_names = false;
    bool number_is_rank = false;      // the numbers are 1..N in file order (nice: and the prefix is empty)
    std::string name_prefix;
    std::vector<uint32_t> id_of_name; // numeric, not number_is_rank: name value -> node id (0 = no such segment)
    bool has_edges = false;
    EdgeMap edges;
    std::vector<uint64_t> edge_uv_by_id;  // the edges in id order ([0] unused), kept beside the map: what edge_ends hands out
    std::vector<uint8_t> edge_oo_by_id;

    // filled instead of the fields above when the graph comes from a .pcsr cache
    bool cached = false;                 // image = the mapped .pcsr file, the arrays below point into it
    const uint32_t *c_items[2] = {nullptr, nullptr};    // [0] node, [1] edge ItemTable
    const uint64_t *c_prefsum[2] = {nullptr, nullptr};
    uint64_t c_n_items[2] = {0, 0};
    const char *c_name_blob = nullptr;   // node names, concatenated
    const uint64_t *c_name_off = nullptr;  // node_count + 1
    const uint64_t *c_edge_uv = nullptr; // per edge id (0 unused): (u << 32) | v
    const uint8_t *c_edge_oo = nullptr;  // per edge id: (o1 << 1) | o2

    uint32_t node_id(const char *p, size_t len) const {
        if (number_is_rank) {
            const size_t pl = name_prefix.size();
            if (len <= pl || std::memcmp(p, name_prefix.data(), pl) != 0) return 0;
            uint64_t v = 0;
            for (size_t i = pl; i < len; ++i) {
                if (p[i] < '0' || p[i] > '9') return 0;
                v = v * 10 + (uint64_t)(p[i] - '0');
                if (v > 0xFFFFFFFFull) return 0;
            }
            return (uint32_t)v;
        }
        ensure_names();
        return names.find(p, (uint32_t)len);
    }
    // node2id for names that are not numbers in rank order (graph.rs:308-375).  The device routes never ask for it (they hash
    // the names themselves, or index a table by the number), so it is built when the first host lookup comes.
    void ensure_names() const {
        std::call_once(names_once, [this]() {
            names.init(node_names.size());
            for (size_t k = 0; k < node_names.size(); ++k)
                if (!names.insert(image.data() + node_names[k].b, (uint32_t)(node_names[k].e - node_names[k].b), (uint32_t)(k + 1)))
                    throw std::runtime_error("Segment with ID " + image.substr(node_names[k].b, node_names[k].e - node_names[k].b) +
                                             " occurs multiple times in GFA");
        });
    }
};

GraphStorage::GraphStorage() : impl_(std::make_shared<Impl>()) {}
GraphStorage::~GraphStorage() = default;

// GraphStorage::from_gfa (graph.rs:195-220): parse_nodes_gfa (308-375) + parse_edge_gfa (276-306)
std::unique_ptr<GraphStorage> GraphStorage::from_gfa(const std::string &gfa_file, bool index_edges, bool /*nice*/,
                                                     const TextHook &on_text, bool links_only) {
    std::unique_ptr<GraphStorage> g(new GraphStorage());
    Impl &im = *g->impl_;
    phase_mark("start from_gfa");
    im.image.open(gfa_file);
    phase_mark("image open (mmap / inflate)");
    if (on_text) on_text(im.image.data(), im.image.size(), std::static_pointer_cast<const void>(g->impl_));
    const Image &s = im.image;
    const size_t N = s.size();

    // line scan in parallel: every worker takes a byte range and starts at the first line that
    // begins inside it
    std::vector<Span> s_lines, l_lines;
    {
        const size_t PIECE = 8u << 20;
        const size_t n_pieces = (N + PIECE - 1) / PIECE;
        struct Found {
            std::vector<Span> s, l, p;
            size_t l_lo = ~(size_t)0, l_hi = 0;
        };
        std::vector<Found> found(n_pieces);
        ThreadPool::instance().parallel_for(n_pieces, [&](size_t k) {
            size_t b = k * PIECE;
            const size_t stop = std::min(N, b + PIECE);
            if (b > 0) {  // skip the tail of a line that started in the previous piece
                if (s[b - 1] != '\n') {
                    const void *nl = std::memchr(s.data() + b, '\n', N - b);
                    b = nl ? (size_t)((const char *)nl - s.data()) + 1 : N;
                }
            }
            Found &f = found[k];
            while (b < stop) {
                const void *nl = std::memchr(s.data() + b, '\n', N - b);
                const size_t e = nl ? (size_t)((const char *)nl - s.data()) : N;
                if (e > b) {
                    switch (s[b]) {
                        case 'S': f.s.push_back({b, e}); break;
                        case 'L':
                            if (index_edges) f.l.push_back({b, e});
                            else if (links_only) {
                                f.l_lo = std::min(f.l_lo, b);
                                f.l_hi = e;
                            }
                            break;
                        case 'P': case 'W': f.p.push_back({b, e}); break;
                        default: break;
                    }
                }
                b = e + 1;
            }
        });
        size_t ns = 0, nl = 0, np = 0;
        for (auto &f : found) {
            ns += f.s.size();
            nl += f.l.size();
            np += f.p.size();
            if (f.l_hi) {
                if (!im.link_hi) im.link_lo = f.l_lo;
                im.link_hi = f.l_hi;
            }
        }
        s_lines.reserve(ns);
        l_lines.reserve(nl);
        im.p_lines.reserve(np);
        for (auto &f : found) {
            s_lines.insert(s_lines.end(), f.s.begin(), f.s.end());
            l_lines.insert(l_lines.end(), f.l.begin(), f.l.end());
            im.p_lines.insert(im.p_lines.end(), f.p.begin(), f.p.end());
        }
    }
    phase_mark("line scan");
    if (s_lines.size() >= 0xFFFFFFFEull) throw std::runtime_error("more than 2^32-2 segments are not supported");
    if (!s_lines.empty()) {
        im.seg_lo = s_lines.front().b;
        im.seg_hi = s_lines.back().e;
    }

    // --- S lines: ids are 1-based ranks, node_lens[id] = length of the sequence column ---
    g->node_lens_.assign(s_lines.size() + 1, 0);
    std::vector<Span> name_of(s_lines.size());
    std::atomic<bool> nice_all{true}, malformed{false}, numeric_all{true};
    std::atomic<uint64_t> name_max{0}, len_max{0};
    std::vector<uint32_t> name_val(s_lines.size());  // the number at the end of the name (when there is one)
    const size_t BL = 1u << 16;
    const size_t nb = (s_lines.size() + BL - 1) / BL;
    std::vector<Span> block_head(nb, Span{0, 0});    // what stands in front of the number, per block (the same for all its names)
    {
        ThreadPool::instance().parallel_for(nb, [&](size_t blk) {
            bool nice = true, numeric = true;
            uint64_t vmax = 0, lmax = 0;
            const size_t k1 = std::min(s_lines.size(), (blk + 1) * BL);
            for (size_t k = blk * BL; k < k1; ++k) {
                const Span ln = s_lines[k];
                if (ln.e < ln.b + 3) {
                    malformed.store(true);
                    return;
                }
                size_t ne = field_end(s, ln.b + 2, ln.e);
                name_of[k] = {ln.b + 2, ne};
                lmax = std::max<uint64_t>(lmax, ne - (ln.b + 2));
                size_t q0 = ne < ln.e ? ne + 1 : ln.e, q1 = q0;
                const void *tb = q0 < ln.e ? std::memchr(s.data() + q0, '\t', ln.e - q0) : nullptr;
                q1 = tb ? (size_t)((const char *)tb - s.data()) : ln.e;
                if (q1 > q0 && s[q1 - 1] == '\r' && q1 == ln.e) --q1;
                g->node_lens_[k + 1] = (uint32_t)(q1 - q0);
                if (numeric) {
                    // the name as [head] + a decimal number (no sign, no leading zero, < 2^32); the head -- at most 8 bytes --
                    // must be the same for every segment; nice: the number is exactly k + 1 (and, below, the head is empty)
                    size_t ts = ne;
                    while (ts > ln.b + 2 && s[ts - 1] >= '0' && s[ts - 1] <= '9') --ts;
                    uint64_t v = 0;
                    bool ok = ne > ts && ne - ts <= 10 && (s[ts] != '0' || ne == ts + 1) && ts - (ln.b + 2) <= 8;
                    for (size_t i = ts; ok && i < ne; ++i) v = v * 10 + (uint64_t)(s[i] - '0');
                    ok = ok && v <= 0xFFFFFFFEull;
                    if (ok) {
                        const Span head{ln.b + 2, ts};
                        if (k == blk * BL) block_head[blk] = head;
                        else {
                            const Span h0 = block_head[blk];
                            ok = h0.e - h0.b == head.e - head.b && std::memcmp(s.data() + h0.b, s.data() + head.b, head.e - head.b) == 0;
                        }
                    }
                    nice = nice && ok && v == k + 1;
                    numeric = ok;
                    name_val[k] = ok ? (uint32_t)v : 0u;
                    if (ok && v > vmax) vmax = v;
                } else {
                    nice = false;
                }
            }
            if (!nice) nice_all.store(false);
            if (!numeric) numeric_all.store(false);
            uint64_t cur = name_max.load();
            while (vmax > cur && !name_max.compare_exchange_weak(cur, vmax)) {
            }
            cur = len_max.load();
            while (lmax > cur && !len_max.compare_exchange_weak(cur, lmax)) {
            }
        });
    }
    if (malformed.load()) throw std::runtime_error("malformed S line");
    bool same_head = numeric_all.load();
    for (size_t b = 1; same_head && b < nb; ++b)
        same_head = block_head[b].e - block_head[b].b == block_head[0].e - block_head[0].b &&
                    std::memcmp(s.data() + block_head[b].b, s.data() + block_head[0].b, block_head[0].e - block_head[0].b) == 0;
    if (!same_head) numeric_all.store(false);
    if (same_head && nb) im.name_prefix = s.substr(block_head[0].b, block_head[0].e - block_head[0].b);
    im.number_is_rank = same_head && nice_all.load() && !s_lines.empty();
    im.nice = im.number_is_rank && im.name_prefix.empty();
    im.max_name_len = (uint32_t)std::min<uint64_t>(len_max.load(), 0xFFFFFFFFull);
    g->node_count_ = s_lines.size();
    im.node_names = std::move(name_of);
    // (the name -> id map of names that are not the ranks is built on first use: Impl::ensure_names.  A name that occurs twice
    // -- the reference panics, graph.rs:336 -- is found there, by the check of the number table below, or by the device's name
    // table, whichever route the command takes)
    // numeric names: name -> id as a plain table (the device tokeniser's lookup), when the numbers are small enough
    im.numeric_names = !s_lines.empty() && numeric_all.load() && (im.number_is_rank || name_max.load() <= 16 * (uint64_t)s_lines.size() + 4096);
    if (!im.numeric_names) {
        im.name_prefix.clear();
        im.number_is_rank = false;
    }
    if (im.numeric_names && !im.number_is_rank) {
        im.id_of_name.assign(name_max.load() + 1, 0);
        ThreadPool::instance().parallel_for((s_lines.size() + (1u << 16) - 1) >> 16, [&](size_t blk) {
            const size_t k1 = std::min(s_lines.size(), (blk + 1) << 16);
            for (size_t k = blk << 16; k < k1; ++k) im.id_of_name[name_val[k]] = (uint32_t)(k + 1);
        });
        std::atomic<int64_t> dup{-1};
        ThreadPool::instance().parallel_for((s_lines.size() + (1u << 16) - 1) >> 16, [&](size_t blk) {
            const size_t k1 = std::min(s_lines.size(), (blk + 1) << 16);
            for (size_t k = blk << 16; k < k1; ++k)
                if (im.id_of_name[name_val[k]] != (uint32_t)(k + 1)) dup.store((int64_t)k);  // another S line wrote the same number
        });
        if (dup.load() >= 0) {
            const Span sp = im.node_names[(size_t)dup.load()];
            throw std::runtime_error("Segment with ID " + s.substr(sp.b, sp.e - sp.b) + " occurs multiple times in GFA");
        }
    }

    phase_mark("S lines (names, lengths)");
    // --- P / W lines: path identity + the span of the step column ---
    const size_t P = im.p_lines.size();
    g->paths_.resize(P);
    im.step_fields.resize(P);
    im.is_walk.resize(P);
    // (in parallel: finding the end of a step column is a scan over the column, 2 GB in all for a chr22-sized graph)
    ThreadPool::instance().parallel_for(P, [&](size_t k) {
        const Span ln = im.p_lines[k];
        size_t le = ln.e;
        if (le > ln.b && s[le - 1] == '\r') --le;
        if (s[ln.b] == 'P') {
            size_t n0 = field_end(s, ln.b, le) + 1;
            if (n0 > le) throw std::runtime_error("malformed P line");
            size_t n1 = field_end(s, n0, le);
            g->paths_[k] = PathSegment::from_str(s.view(n0, n1 - n0));
            size_t f0 = n1 < le ? n1 + 1 : le;
            im.step_fields[k] = {f0, field_end(s, f0, le)};
            im.is_walk[k] = 0;
        } else {  // W sample hap seqid start end walk  (graph.rs:381-412)
            size_t col[7];
            size_t pos = ln.b;
            for (int c = 0; c < 6; ++c) {
                size_t e = field_end(s, pos, le);
                if (e >= le) throw std::runtime_error("malformed W line");
                col[c] = pos;
                pos = e + 1;
            }
            col[6] = pos;
            PathSegment ps;
            ps.sample = s.substr(col[1], col[2] - col[1] - 1);
            ps.has_haplotype = true;
            ps.haplotype = s.substr(col[2], col[3] - col[2] - 1);
            ps.has_seqid = true;
            ps.seqid = s.substr(col[3], col[4] - col[3] - 1);
            std::string_view a = s.view(col[4], col[5] - col[4] - 1);
            std::string_view b = s.view(col[5], col[6] - col[5] - 1);
            if (a != "*") {
                if (!parse_u64(a, ps.start)) throw std::runtime_error("malformed W line (start)");
                ps.has_start = true;
            }
            if (b != "*") {
                if (!parse_u64(b, ps.end)) throw std::runtime_error("malformed W line (end)");
                ps.has_end = true;
            }
            g->paths_[k] = std::move(ps);
            im.step_fields[k] = {pos, field_end(s, pos, le)};
            im.is_walk[k] = 1;
        }
    });

    phase_mark("P/W headers");
    // (names that are not numbers are checked for duplicates -- the reference panics, graph.rs:336 -- where they are first looked
    // up: the host's map, or the device's name table.  A graph without a single P, W or L line never looks one up: check it here)
    if (!im.numeric_names && g->paths_.empty() && l_lines.empty() && !s_lines.empty()) im.ensure_names();
    im.l_lines = std::move(l_lines);
    im.links_only = links_only && !index_edges;
    if (index_edges) g->build_edge_index();
    return g;
}

// --- L lines: edge id = rank of the first occurrence of the canonical form (graph.rs:276-306) ---
void GraphStorage::build_edge_index() {
    GraphStorage *g = this;
    Impl &im = *impl_;
    if (im.has_edges) return;
    const Image &s = im.image;
    if (im.links_only && im.link_hi > im.link_lo) {  // the L lines were only seen so far: collect them now
        size_t b = im.link_lo;                        // (link_lo is the start of a line)
        while (b < im.link_hi) {
            const void *nl = std::memchr(s.data() + b, '\n', s.size() - b);
            const size_t e = nl ? (size_t)((const char *)nl - s.data()) : s.size();
            if (e > b && s[b] == 'L') im.l_lines.push_back({b, e});
            b = e + 1;
        }
    }
    const std::vector<Span> &l_lines = im.l_lines;
    {
        im.has_edges = true;
        im.edges.init_parallel(l_lines.size());
        // the lines are parsed in parallel (two name lookups each); ids are given in file order afterwards
        const size_t n_l = l_lines.size();
        std::vector<uint64_t> l_uv(n_l);
        std::vector<uint8_t> l_oo(n_l);
        std::atomic<int64_t> bad_line{-1};
        std::atomic<int> bad_kind{0};
        const size_t L_CHUNK = 8192;
        ThreadPool::instance().parallel_for((n_l + L_CHUNK - 1) / L_CHUNK, [&](size_t c) {
            const size_t hi = std::min(n_l, (c + 1) * L_CHUNK);
            for (size_t k = c * L_CHUNK; k < hi; ++k) {
                const Span ln = l_lines[k];
                size_t a0 = ln.b + 2, a1 = field_end(s, a0, ln.e);
                if (a1 + 2 >= ln.e) {
                    bad_kind.store(1);
                    bad_line.store((int64_t)k);
                    return;
                }
                const uint32_t u = im.node_id(s.data() + a0, a1 - a0);
                // (an orientation is '+' or '-': the reference panics on anything else, Orientation::from_pm, graph.rs:42-48)
                if (s[a1 + 1] != '+' && s[a1 + 1] != '-') {
                    bad_kind.store(1);
                    bad_line.store((int64_t)k);
                    return;
                }
                const uint8_t o1 = s[a1 + 1] == '+' ? 0 : 1;
                size_t b0 = a1 + 3, b1 = field_end(s, b0, ln.e);
                if (b1 + 1 >= ln.e) {
                    bad_kind.store(1);
                    bad_line.store((int64_t)k);
                    return;
                }
                const uint32_t v = im.node_id(s.data() + b0, b1 - b0);
                if (s[b1 + 1] != '+' && s[b1 + 1] != '-') {
                    bad_kind.store(1);
                    bad_line.store((int64_t)k);
                    return;
                }
                const uint8_t o2 = s[b1 + 1] == '+' ? 0 : 1;
                if (!u || u > g->node_count_ || !v || v > g->node_count_) {
                    bad_kind.store(!u || u > g->node_count_ ? 2 : 3);
                    bad_line.store((int64_t)k);
                    return;
                }
                canonical(u, o1, v, o2, l_uv[k], l_oo[k]);
            }
        });
        if (bad_line.load() >= 0) {
            const Span ln = l_lines[(size_t)bad_line.load()];
            if (bad_kind.load() == 1) throw std::runtime_error("malformed L line");
            size_t a0 = ln.b + 2, a1 = field_end(s, a0, ln.e);
            if (bad_kind.load() == 2) throw std::runtime_error("unknown node " + s.substr(a0, a1 - a0));
            size_t b0 = a1 + 3, b1 = field_end(s, b0, ln.e);
            throw std::runtime_error("unknown node " + s.substr(b0, b1 - b0));
        }
        // The map is filled by all threads at once.  A slot is claimed by a compare-and-swap on its ends; its id field
        // carries, while the map is built, the TAG of the earliest line with that edge -- (line + 1) << 2 | orientations, 0 =
        // not written yet -- lowered by an atomic minimum, so that duplicated edges (skipped by the reference, graph.rs:296)
        // resolve to their first occurrence whatever the threads' order.  Ids are then the ranks of the first occurrences
        // in file order, as a serial pass would give them.
        if (n_l >= ((size_t)1 << 30)) throw std::runtime_error("more than 2^30 L lines");
        std::vector<uint32_t> slot_of(n_l);
        auto &tab = im.edges.tab;
        const uint64_t mask = im.edges.mask;
        ThreadPool::instance().parallel_for((n_l + L_CHUNK - 1) / L_CHUNK, [&](size_t c) {
            const size_t hi = std::min(n_l, (c + 1) * L_CHUNK);
            for (size_t k = c * L_CHUNK; k < hi; ++k) {
                const uint64_t uv = l_uv[k];
                const uint32_t oo = l_oo[k], tag = (uint32_t)(((k + 1) << 2) | oo);
                uint64_t j = EdgeMap::h(uv, (uint8_t)oo) & mask;
                for (;;) {
                    uint64_t cur = __atomic_load_n(&tab[j].uv, __ATOMIC_ACQUIRE);
                    bool mine = false;
                    if (cur == 0) {
                        uint64_t expected = 0;
                        mine = __atomic_compare_exchange_n(&tab[j].uv, &expected, uv, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
                        cur = mine ? uv : expected;
                    }
                    if (cur == uv) {
                        uint32_t t = __atomic_load_n(&tab[j].id, __ATOMIC_ACQUIRE);
                        if (!mine)
                            while (t == 0) t = __atomic_load_n(&tab[j].id, __ATOMIC_ACQUIRE);  // the claimer writes it next
                        if (mine || (t & 3u) == oo) {
                            while ((t == 0 || tag < t) &&
                                   !__atomic_compare_exchange_n(&tab[j].id, &t, tag, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
                            }
                            slot_of[k] = (uint32_t)j;
                            break;
                        }
                    }
                    j = (j + 1) & mask;
                }
            }
        });
        std::vector<uint32_t> id_of_line(n_l);
        ThreadPool::instance().parallel_for((n_l + L_CHUNK - 1) / L_CHUNK, [&](size_t c) {  // (random reads of the map: not for one thread)
            const size_t hi = std::min(n_l, (c + 1) * L_CHUNK);
            for (size_t k = c * L_CHUNK; k < hi; ++k) id_of_line[k] = tab[slot_of[k]].id == (uint32_t)(((k + 1) << 2) | l_oo[k]) ? 1u : 0u;
        });
        uint32_t next = 1;
        for (size_t k = 0; k < n_l; ++k)
            if (id_of_line[k]) id_of_line[k] = next++;
        im.edge_uv_by_id.assign(next, 0);
        im.edge_oo_by_id.assign(next, 0);
        ThreadPool::instance().parallel_for((n_l + L_CHUNK - 1) / L_CHUNK, [&](size_t c) {
            const size_t hi = std::min(n_l, (c + 1) * L_CHUNK);
            for (size_t k = c * L_CHUNK; k < hi; ++k) {
                const uint32_t id = id_of_line[k];
                if (!id) continue;
                auto &sl = tab[slot_of[k]];
                sl.id = id;
                sl.oo = l_oo[k];
                im.edge_uv_by_id[id] = l_uv[k];
                im.edge_oo_by_id[id] = l_oo[k];
            }
        });
        im.edges.n = next - 1;
        g->edge_count_ = next - 1;
        phase_mark("L lines (edges)");
    }
    im.links_only = false;
}
void GraphStorage::ensure_edge_index() const {
    // (const accessors come here from any thread: one of them builds, the others wait)
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    const_cast<GraphStorage *>(this)->build_edge_index();
}
bool GraphStorage::has_zero_length_nodes() const {
    for (size_t i = 1; i < node_lens_.size(); ++i)
        if (node_lens_[i] == 0) return true;
    return false;
}
void GraphStorage::require_edges(const char *why) const {
    if (!impl_->has_edges && impl_->links_only) ensure_edge_index();
    if (!impl_->has_edges) throw std::runtime_error(why);
}
bool GraphStorage::has_edge_index() const { return impl_->has_edges; }
bool GraphStorage::links_for_device() const { return impl_->links_only && !impl_->has_edges && !impl_->cached; }
void GraphStorage::link_range(uint64_t &lo, uint64_t &hi) const {
    lo = impl_->link_lo;
    hi = impl_->link_hi;
}
bool GraphStorage::names_by_bytes_on_device() const {
    const Impl &im = *impl_;
    return !im.cached && !im.numeric_names && node_count_ > 0 && im.max_name_len >= 1 && im.max_name_len <= 16;
}
void GraphStorage::segment_range(uint64_t &lo, uint64_t &hi) const {
    lo = impl_->seg_lo;
    hi = impl_->seg_hi;
}
void GraphStorage::name_fields(std::vector<uint64_t> &off, std::vector<uint8_t> &len) const {
    const auto &nn = impl_->node_names;
    off.resize(nn.size());
    len.resize(nn.size());
    for (size_t k = 0; k < nn.size(); ++k) {
        off[k] = nn[k].b;
        len[k] = (uint8_t)(nn[k].e - nn[k].b);
    }
}


// ------------------------------------------------------------------------------------------
// ItemTable
// ------------------------------------------------------------------------------------------
namespace {

// chunks of a step column that start on a step boundary, ~64 KiB each
struct Chunk {
    uint32_t path;
    size_t b, e;
    uint64_t n_steps = 0, out = 0;
};

}  // namespace

ItemTableView GraphStorage::item_table_view(CountType count, ItemTable &storage) const {
    const Impl &im = *impl_;
    if (im.cached) {
        if (count == COUNT_EDGE) require_edges("graph was loaded without edge index");
        const int k = count == COUNT_EDGE ? 1 : 0;
        return ItemTableView{im.c_items[k], im.c_prefsum[k], im.c_n_items[k]};
    }
    storage = item_table(count);
    return ItemTableView{storage.items.data(), storage.id_prefsum.data(), storage.items.size()};
}

namespace {
struct Steps {  // every path's steps as node ids (+ orientations), file order
    std::vector<uint32_t> ids;
    std::vector<uint8_t> ori;  // 0 forward, 1 backward; empty unless asked for
    std::vector<uint64_t> pref;
};
}  // namespace

static void parse_all_steps(const GraphStorage::Impl &im, const std::vector<PathSegment> &paths_, uint64_t node_count_,
                            bool want_ori, Steps &out) {
    const Image &s = im.image;
    const size_t P = paths_.size();
    const CountType count = want_ori ? COUNT_EDGE : COUNT_NODE;
    constexpr size_t CHUNK = 64 * 1024;

    std::vector<Chunk> chunks;
    for (size_t k = 0; k < P; ++k) {
        const Span f = im.step_fields[k];
        size_t b = f.b;
        while (b < f.e) {
            size_t e = std::min(f.e, b + CHUNK);
            if (e < f.e) {  // move e forward to the start of the next step
                if (im.is_walk[k]) {
                    while (e < f.e && s[e] != '>' && s[e] != '<') ++e;
                } else {
                    const void *c = std::memchr(s.data() + e, ',', f.e - e);
                    e = c ? (size_t)((const char *)c - s.data()) + 1 : f.e;
                }
            }
            chunks.push_back(Chunk{(uint32_t)k, b, e});
            b = e;
        }
    }
    ThreadPool &pool = ThreadPool::instance();
    // pass 1: steps per chunk
    pool.parallel_for(chunks.size(), [&](size_t ci) {
        Chunk &c = chunks[ci];
        uint64_t n = 0;
        if (im.is_walk[c.path]) {
            for (size_t i = c.b; i < c.e; ++i) n += (s[i] == '>') | (s[i] == '<');
        } else {
            // steps are separated by ','; empty pieces (",,") carry no step
            size_t i = c.b;
            while (i < c.e) {
                const void *cm = std::memchr(s.data() + i, ',', c.e - i);
                size_t e = cm ? (size_t)((const char *)cm - s.data()) : c.e;
                if (e > i) ++n;
                i = e + 1;
            }
        }
        c.n_steps = n;
    });
    std::vector<uint64_t> node_pref(P + 1, 0);
    {
        uint64_t run = 0;
        size_t ci = 0;
        for (size_t k = 0; k < P; ++k) {
            node_pref[k] = run;
            while (ci < chunks.size() && chunks[ci].path == k) {
                chunks[ci].out = run;
                run += chunks[ci].n_steps;
                ++ci;
            }
        }
        node_pref[P] = run;
    }
    const uint64_t S = node_pref[P];
    std::vector<uint32_t> ids(S);
    std::vector<uint8_t> ori(count == COUNT_EDGE ? S : 0);
    std::atomic<size_t> bad_at{(size_t)-1};
    // pass 2: name -> id (get_segment_id / get_walk_segment_id, util.rs:1017-1046)
    pool.parallel_for(chunks.size(), [&](size_t ci) {
        const Chunk &c = chunks[ci];
        uint64_t o = c.out;
        if (im.is_walk[c.path]) {
            size_t i = c.b;
            while (i < c.e) {
                size_t e = i + 1;
                while (e < c.e && s[e] != '>' && s[e] != '<') ++e;
                uint32_t id = (s[i] == '>' || s[i] == '<') ? im.node_id(s.data() + i + 1, e - i - 1) : 0;
                if (!id || id > node_count_) { bad_at.store(i); return; }
                ids[o] = id;
                if (!ori.empty()) ori[o] = s[i] == '>' ? 0 : 1;
                ++o;
                i = e;
            }
        } else {
            size_t i = c.b;
            while (i < c.e) {
                const void *cm = std::memchr(s.data() + i, ',', c.e - i);
                size_t e = cm ? (size_t)((const char *)cm - s.data()) : c.e;
                if (e > i) {
                    char oc = s[e - 1];
                    uint32_t id = (oc == '+' || oc == '-') ? im.node_id(s.data() + i, e - 1 - i) : 0;
                    if (!id || id > node_count_) { bad_at.store(i); return; }
                    ids[o] = id;
                    if (!ori.empty()) ori[o] = oc == '+' ? 0 : 1;
                    ++o;
                }
                i = e + 1;
            }
        }
    });
    if (bad_at.load() != (size_t)-1) {
        size_t i = bad_at.load(), e = i;
        while (e < s.size() && s[e] != ',' && s[e] != '\t' && s[e] != '\n' && e - i < 64) ++e;
        throw std::runtime_error("unknown node " + s.substr(i, e - i));
    }

    out.ids = std::move(ids);
    out.ori = std::move(ori);
    out.pref = std::move(node_pref);
}

ItemTable GraphStorage::item_table(CountType count) const {
    const Impl &im = *impl_;
    const size_t P = paths_.size();
    if (count == COUNT_EDGE) require_edges("graph was loaded without edge index");
    if (im.cached) {  // parallel copy out of the mapping (item_table_view() avoids even that)
        const int k = count == COUNT_EDGE ? 1 : 0;
        ItemTable t;
        t.id_prefsum.assign(im.c_prefsum[k], im.c_prefsum[k] + P + 1);
        t.items.resize(im.c_n_items[k]);
        const size_t n = im.c_n_items[k], CH = 1 << 22;
        ThreadPool::instance().parallel_for((n + CH - 1) / CH, [&](size_t c) {
            const size_t b = c * CH, e = std::min(n, b + CH);
            std::memcpy(t.items.data() + b, im.c_items[k] + b, (e - b) * sizeof(uint32_t));
        });
        return t;
    }
    Steps steps;
    phase_mark("item_table start");
    parse_all_steps(im, paths_, node_count_, count == COUNT_EDGE, steps);
    phase_mark("parse_all_steps");
    std::vector<uint32_t> &ids = steps.ids;
    std::vector<uint8_t> &ori = steps.ori;
    std::vector<uint64_t> &node_pref = steps.pref;
    ThreadPool &pool = ThreadPool::instance();

    ItemTable t;
    if (count != COUNT_EDGE) {
        t.items = std::move(ids);
        t.id_prefsum = std::move(node_pref);
        return t;
    }
    // edge steps: update_tables_edgecount (util.rs:723-795) with include=[(0,MAX)], exclude=[]
    t.id_prefsum.assign(P + 1, 0);
    std::vector<uint64_t> slot(P + 1, 0);  // upper bound: len-1 edges per path
    for (size_t k = 0; k < P; ++k) {
        uint64_t len = node_pref[k + 1] - node_pref[k];
        slot[k + 1] = slot[k] + (len ? len - 1 : 0);
    }
    t.items.assign(slot[P], 0);
    std::vector<uint64_t> kept(P, 0);
    std::atomic<int64_t> bad_path{-1};
    pool.parallel_for(P, [&](size_t k) {
        const uint64_t b = node_pref[k], e = node_pref[k + 1];
        if (e - b < 2) return;
        uint64_t pcoord = (paths_[k].has_start && paths_[k].has_end) ? paths_[k].start : 0;
        pcoord += node_lens_[ids[b]];
        uint64_t w = slot[k];
        for (uint64_t j = b; j + 1 < e; ++j) {
            const uint64_t l = node_lens_[ids[j + 1]];
            uint64_t uv;
            uint8_t oo;
            canonical(ids[j], ori[j], ids[j + 1], ori[j + 1], uv, oo);
            uint32_t eid = im.edges.find(uv, oo);
            if (!eid) { bad_path.store((int64_t)k); return; }
            if (0 < pcoord + l) t.items[w++] = eid;  // include_coords[0].0 < p + l
            pcoord += l;
        }
        kept[k] = w - slot[k];
    });
    if (bad_path.load() >= 0)
        throw std::runtime_error("unknown edge in path " + paths_[(size_t)bad_path.load()].display());
    // compact (only needed if some edge was skipped, which requires zero-length segments)
    uint64_t run = 0;
    bool dense = true;
    for (size_t k = 0; k < P; ++k) {
        t.id_prefsum[k] = run;
        dense = dense && kept[k] == slot[k + 1] - slot[k];
        run += kept[k];
    }
    t.id_prefsum[P] = run;
    if (!dense) {
        std::vector<uint32_t> packed(run);
        for (size_t k = 0; k < P; ++k)
            std::copy(t.items.begin() + slot[k], t.items.begin() + slot[k] + kept[k], packed.begin() + t.id_prefsum[k]);
        t.items.swap(packed);
    }
    return t;
}

// ------------------------------------------------------------------------------------------
// groups + visiting order
// ------------------------------------------------------------------------------------------
namespace {
std::vector<std::string> read_lines(const std::string &file) {
    std::ifstream in(file, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + file);
    std::vector<std::string> out;
    std::string l;
    while (std::getline(in, l)) out.push_back(l);
    return out;
}
}  // namespace

namespace {

// groups of all paths: GraphMask::load_groups (abacus.rs:242-308)
std::vector<std::string> load_groups(const std::vector<PathSegment> &paths, const std::vector<std::string> &key,
                                     GroupMode mode, const std::string &group_file) {
    const size_t P = paths.size();
    std::vector<std::string> group(P);
    if (mode == GROUP_HAPLOTYPE) {
        for (size_t i = 0; i < P; ++i) group[i] = paths[i].sample + "#" + (paths[i].has_haplotype ? paths[i].haplotype : "");
    } else if (mode == GROUP_SAMPLE) {
        for (size_t i = 0; i < P; ++i) group[i] = paths[i].sample;
    } else if (mode == GROUP_FILE) {
        std::unordered_map<std::string, std::string> assigned;
        int lineno = 1;
        for (std::string l : read_lines(group_file)) {  // parse_groups, io.rs:121-151
            if (!l.empty() && l.back() == '\r') l.pop_back();
            size_t t0 = l.find('\t');
            if (t0 == std::string::npos || l.find('\t', t0 + 1) != std::string::npos)
                throw std::runtime_error("error in line " + std::to_string(lineno) + ": table must have exactly two columns");
            std::string k = PathSegment::from_str(std::string_view(l).substr(0, t0)).clear_key();
            std::string grp = l.substr(t0 + 1);
            auto it = assigned.find(k);
            if (it != assigned.end() && it->second != grp)
                throw std::runtime_error("error in line " + std::to_string(lineno) +
                                         ": path cannot be assigned to more than one group");
            assigned.emplace(k, grp);
            ++lineno;
        }
        for (size_t i = 0; i < P; ++i) {
            auto it = assigned.find(key[i]);
            group[i] = it != assigned.end() ? it->second : paths[i].id();
        }
    } else {
        for (size_t i = 0; i < P; ++i) group[i] = paths[i].id();
    }
    return group;
}

// usize::from_str: decimal digits with an optional leading '+'
bool parse_usize_field(std::string_view f, uint64_t &out) {
    if (!f.empty() && f[0] == '+') f.remove_prefix(1);
    return parse_u64(f, out);
}

// "1,2,3,".split(',').filter_map(|s| usize::from_str(s.trim()).ok())
std::vector<uint64_t> parse_usize_list(std::string_view f) {
    std::vector<uint64_t> out;
    size_t i = 0;
    for (;;) {
        size_t e = f.find(',', i);
        if (e == std::string_view::npos) e = f.size();
        std::string_view t = f.substr(i, e - i);
        while (!t.empty() && std::isspace((unsigned char)t.front())) t.remove_prefix(1);
        while (!t.empty() && std::isspace((unsigned char)t.back())) t.remove_suffix(1);
        uint64_t v;
        if (parse_usize_field(t, v)) out.push_back(v);
        if (e >= f.size()) break;
        i = e + 1;
    }
    return out;
}

// parse_bed_to_path_segments with block info (io.rs:35-119): a name alone, or name / start / end
// (the columns override coordinates in the name), or BED12 with one segment per block
std::vector<PathSegment> parse_bed(const std::string &file) {
    std::vector<PathSegment> out;
    int lineno = 0;
    for (std::string l : read_lines(file)) {
        ++lineno;
        if (!l.empty() && l.back() == '\r') l.pop_back();
        std::vector<std::string_view> fields;
        {
            std::string_view v(l);
            size_t i = 0;
            for (;;) {
                size_t e = v.find('\t', i);
                if (e == std::string_view::npos) {
                    fields.push_back(v.substr(i));
                    break;
                }
                fields.push_back(v.substr(i, e - i));
                i = e + 1;
            }
        }
        const std::string_view name = fields[0];
        if (name.rfind("browser ", 0) == 0 || name.rfind("track ", 0) == 0 || (!name.empty() && name[0] == '#')) continue;
        auto with_coords = [&](uint64_t st, uint64_t en) {
            PathSegment ps = PathSegment::from_str(name);
            ps.has_start = ps.has_end = true;
            ps.start = st;
            ps.end = en;
            out.push_back(std::move(ps));
        };
        if (fields.size() == 1) {
            out.push_back(PathSegment::from_str(name));
        } else if (fields.size() >= 3) {
            uint64_t st, en;
            if (!parse_usize_field(fields[1], st) || !parse_usize_field(fields[2], en))
                throw std::runtime_error("error line " + std::to_string(lineno) + ": start / end is not an usize");
            if (fields.size() == 12) {
                uint64_t bc = 0;
                if (!parse_usize_field(fields[9], bc)) bc = 0;
                const std::vector<uint64_t> sizes = parse_usize_list(fields[10]), starts = parse_usize_list(fields[11]);
                if (bc != sizes.size() || bc != starts.size())
                    throw std::runtime_error("error in block sizes/starts in line " + std::to_string(lineno) +
                                             ": counts do not match");
                for (size_t k = 0; k < sizes.size(); ++k) with_coords(st + starts[k], st + starts[k] + sizes[k]);
            } else {
                with_coords(st, en);
            }
        } else {
            throw std::runtime_error("error in line " + std::to_string(lineno) +
                                     ": row must have either 1, 3, or 12 columns, but has 2");
        }
    }
    return out;
}

// GraphMask::load_coord_list (abacus.rs:212-240): a BED file if `text` names one, otherwise a regular
// expression searched in the displayed path names; the matching paths themselves are the list.
// (Rust `regex` syntax in the reference, ECMAScript here: the same for literals, anchors, classes,
// alternation and repetition.)
std::vector<PathSegment> load_coord_list(const std::string &text, const std::vector<PathSegment> &paths) {
    struct stat st;
    if (::stat(text.c_str(), &st) == 0 && S_ISREG(st.st_mode)) return parse_bed(text);
    std::regex re;
    try {
        re = std::regex(text, std::regex::ECMAScript);
    } catch (const std::regex_error &) {
        throw std::runtime_error("string " + text + " is not valid! Neither as a file name nor as a regex");
    }
    std::vector<PathSegment> out;
    for (const PathSegment &p : paths)
        if (std::regex_search(p.display(), re)) out.push_back(p);
    return out;
}

// a BED list of paths / groups resolved like complement_with_group_assignments (abacus.rs:152-206).
// mark[i] = path i is named; visit = entries in file order as path indices (first path of a group).
// exact_coords: a path entry only matches graph paths with equal coordinates, and group members
// only match paths without coordinates (HashSet<&PathSegment> comparison, abacus.rs:329-336).
void read_path_list(const std::string &file, const std::vector<PathSegment> &paths, const std::vector<std::string> &key,
                    const std::vector<std::string> &group, bool exact_coords, std::vector<uint8_t> &mark,
                    std::vector<uint32_t> *visit) {
    const size_t P = paths.size();
    mark.assign(P, 0);
    std::unordered_map<std::string, std::vector<uint32_t>> by_key, by_group;
    for (size_t i = 0; i < P; ++i) {
        by_key[key[i]].push_back((uint32_t)i);
        by_group[group[i]].push_back((uint32_t)i);
    }
    for (const PathSegment &ps : load_coord_list(file, paths)) {
        auto pk = by_key.find(ps.clear_key());
        if (pk != by_key.end()) {
            for (uint32_t i : pk->second) {
                if (exact_coords && (paths[i].has_start != ps.has_start || paths[i].has_end != ps.has_end ||
                                     (ps.has_start && paths[i].start != ps.start) || (ps.has_end && paths[i].end != ps.end)))
                    continue;
                mark[i] = 1;
            }
            if (visit) visit->push_back(pk->second.front());
        } else {
            auto bg = by_group.find(ps.id());
            if (bg == by_group.end()) continue;  // unknown path/group: logged and skipped by the reference
            if (ps.has_start && ps.has_end)
                throw std::runtime_error("invalid coordinate \"" + ps.display() +
                                         "\": group identifiers are not allowed to have start/stop information!");
            bool first = true;
            for (uint32_t i : bg->second) {
                if (exact_coords && (paths[i].has_start || paths[i].has_end)) continue;
                mark[i] = 1;
                if (visit && first) visit->push_back(i);
                first = false;
            }
        }
    }
}

}  // namespace

// ---- .pcsr cache ---------------------------------------------------------------------------
namespace {
constexpr char PCSR_MAGIC[8] = {'P', 'C', 'S', 'R', '0', '0', '0', '2'};

struct GfaKey {
    uint64_t size = 0, mtime_ns = 0, hash = 0;
};

bool gfa_key(const std::string &gfa_file, GfaKey &k) {
    struct stat st;
    if (::stat(gfa_file.c_str(), &st) != 0) return false;
    k.size = (uint64_t)st.st_size;
    k.mtime_ns = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
    int fd = ::open(gfa_file.c_str(), O_RDONLY);
    if (fd < 0) return false;
    const size_t MiB = 1 << 20;
    std::vector<char> buf(MiB);
    uint64_t h = 0x9E3779B97F4A7C15ull ^ k.size;
    ssize_t got = ::pread(fd, buf.data(), MiB, 0);
    if (got > 0) h = hash_bytes(buf.data(), (size_t)got) ^ (h << 1);
    if (k.size > MiB) {
        got = ::pread(fd, buf.data(), MiB, (off_t)(k.size - MiB));
        if (got > 0) h ^= hash_bytes(buf.data(), (size_t)got) * 0xD6E8FEB86659FD93ull;
    }
    ::close(fd);
    k.hash = h;
    return true;
}

// every section is padded to 8 bytes, so that the arrays of a mapped cache are aligned
struct Writer {
    FILE *f;
    bool ok = true;
    void raw(const void *p, size_t n) {
        if (n && std::fwrite(p, 1, n, f) != n) ok = false;
        static const char zeros[8] = {0};
        const size_t pad = (8 - n % 8) % 8;
        if (pad && std::fwrite(zeros, 1, pad, f) != pad) ok = false;
    }
    void u64(uint64_t v) { raw(&v, 8); }
    template <typename T>
    void vec(const std::vector<T> &v) { arr(v.data(), v.size()); }
    template <typename T>
    void arr(const T *p, size_t n) {
        u64(n);
        raw(p, n * sizeof(T));
    }
    void str(const std::string &s) {
        u64(s.size());
        raw(s.data(), s.size());
    }
};

// cursor over the mapped cache file
struct Reader {
    const char *p;
    uint64_t remaining;
    bool ok = true;
    const char *take(uint64_t n) {
        const uint64_t padded = (n + 7) & ~7ull;
        if (!ok || padded > remaining) {
            ok = false;
            return nullptr;
        }
        const char *r = p;
        p += padded;
        remaining -= padded;
        return r;
    }
    uint64_t u64() {
        const char *q = take(8);
        uint64_t v = 0;
        if (q) std::memcpy(&v, q, 8);
        return v;
    }
    template <typename T>
    const T *view(uint64_t &n) {  // array that stays in the mapping
        n = u64();
        if (!ok || n > remaining / sizeof(T)) {
            ok = false;
            return nullptr;
        }
        return reinterpret_cast<const T *>(take(n * sizeof(T)));
    }
    template <typename T>
    void vec(std::vector<T> &v) {
        uint64_t n = 0;
        const T *q = view<T>(n);
        if (ok) v.assign(q, q + n);
    }
    void str(std::string &s) {
        uint64_t n = 0;
        const char *q = view<char>(n);
        if (ok) s.assign(q, n);
    }
};
}  // namespace

void GraphStorage::save_cache(const std::string &cache_file, const std::string &gfa_file) const {
    const Impl &im = *impl_;
    GfaKey key;
    if (!gfa_key(gfa_file, key)) throw std::runtime_error("cannot stat " + gfa_file);
    const std::string tmp = cache_file + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + tmp);
    Writer w{f};
    w.raw(PCSR_MAGIC, 8);
    w.u64(key.size);
    w.u64(key.mtime_ns);
    w.u64(key.hash);
    w.u64(node_count_);
    w.u64(im.has_edges ? 1 : 0);
    w.u64(edge_count_);
    w.vec(node_lens_);
    // node names
    std::string blob;
    std::vector<uint64_t> off(node_count_ + 1, 0);
    for (uint64_t i = 1; i <= node_count_; ++i) {
        blob += node_name((uint32_t)i);
        off[i] = blob.size();
    }
    w.str(blob);
    w.vec(off);
    // path segments, field by field
    w.u64(paths_.size());
    for (const auto &p : paths_) {
        w.str(p.sample);
        w.u64((p.has_haplotype ? 1u : 0u) | (p.has_seqid ? 2u : 0u) | (p.has_start ? 4u : 0u) | (p.has_end ? 8u : 0u));
        w.str(p.haplotype);
        w.str(p.seqid);
        w.u64(p.start);
        w.u64(p.end);
    }
    const ItemTable nt = item_table(COUNT_NODE);
    w.vec(nt.items);
    w.vec(nt.id_prefsum);
    if (im.has_edges) {
        const ItemTable et = item_table(COUNT_EDGE);
        w.vec(et.items);
        w.vec(et.id_prefsum);
        std::vector<uint64_t> uv(edge_count_ + 1, 0);
        std::vector<uint8_t> oo(edge_count_ + 1, 0);
        if (im.cached) {
            uv.assign(im.c_edge_uv, im.c_edge_uv + edge_count_ + 1);
            oo.assign(im.c_edge_oo, im.c_edge_oo + edge_count_ + 1);
        } else {
            for (const auto &sl : im.edges.tab)
                if (sl.id) {
                    uv[sl.id] = sl.uv;
                    oo[sl.id] = sl.oo;
                }
        }
        w.vec(uv);
        w.vec(oo);
    }
    const bool ok = w.ok && std::fclose(f) == 0;
    if (!ok || std::rename(tmp.c_str(), cache_file.c_str()) != 0) {
        std::remove(tmp.c_str());
        throw std::runtime_error("cannot write " + cache_file);
    }
}

std::unique_ptr<GraphStorage> GraphStorage::from_cache(const std::string &cache_file, const std::string &gfa_file,
                                                       bool need_edges) {
    GfaKey key;
    if (!gfa_key(gfa_file, key)) return nullptr;
    struct stat st;
    if (::stat(cache_file.c_str(), &st) != 0 || st.st_size < 64) return nullptr;
    {   // header check before the whole file is mapped
        FILE *f = std::fopen(cache_file.c_str(), "rb");
        if (!f) return nullptr;
        char head[32];
        const size_t got = std::fread(head, 1, 32, f);
        std::fclose(f);
        uint64_t k[3];
        std::memcpy(k, head + 8, 24);
        if (got != 32 || std::memcmp(head, PCSR_MAGIC, 8) != 0 || k[0] != key.size || k[1] != key.mtime_ns ||
            k[2] != key.hash)
            return nullptr;
    }
    std::unique_ptr<GraphStorage> g(new GraphStorage());
    Impl &im = *g->impl_;
    try {
        im.image.open(cache_file);  // mmap, populated
    } catch (const std::exception &) {
        return nullptr;
    }
    Reader r{im.image.data(), im.image.size()};
    r.take(32);
    im.cached = true;
    g->node_count_ = r.u64();
    im.has_edges = r.u64() != 0;
    g->edge_count_ = r.u64();
    if (!r.ok || (need_edges && !im.has_edges)) return nullptr;
    r.vec(g->node_lens_);
    uint64_t n_blob = 0, n_off = 0;
    im.c_name_blob = r.view<char>(n_blob);
    im.c_name_off = r.view<uint64_t>(n_off);
    const uint64_t P = r.u64();
    if (!r.ok || P > r.remaining / 8) return nullptr;
    g->paths_.resize(P);
    for (uint64_t k = 0; k < P && r.ok; ++k) {
        PathSegment &p = g->paths_[k];
        r.str(p.sample);
        const uint64_t fl = r.u64();
        p.has_haplotype = fl & 1;
        p.has_seqid = fl & 2;
        p.has_start = fl & 4;
        p.has_end = fl & 8;
        r.str(p.haplotype);
        r.str(p.seqid);
        p.start = r.u64();
        p.end = r.u64();
    }
    uint64_t n_pre[2] = {0, 0}, n_uv = 0, n_oo = 0;
    im.c_items[0] = r.view<uint32_t>(im.c_n_items[0]);
    im.c_prefsum[0] = r.view<uint64_t>(n_pre[0]);
    if (im.has_edges) {
        im.c_items[1] = r.view<uint32_t>(im.c_n_items[1]);
        im.c_prefsum[1] = r.view<uint64_t>(n_pre[1]);
        im.c_edge_uv = r.view<uint64_t>(n_uv);
        im.c_edge_oo = r.view<uint8_t>(n_oo);
    }
    bool shapes = r.ok && g->node_lens_.size() == g->node_count_ + 1 && n_off == g->node_count_ + 1 &&
                  n_pre[0] == P + 1 &&
                  (!im.has_edges || (n_pre[1] == P + 1 && n_uv == g->edge_count_ + 1 && n_oo == g->edge_count_ + 1));
    if (shapes) {
        shapes = im.c_name_off[g->node_count_] <= n_blob && im.c_prefsum[0][P] == im.c_n_items[0] &&
                 (!im.has_edges || im.c_prefsum[1][P] == im.c_n_items[1]);
    }
    if (!shapes) return nullptr;
    return g;
}

std::string GraphStorage::node_name(uint32_t id) const {
    const Impl &im = *impl_;
    if (im.cached) {
        if (id == 0 || id > node_count_) throw std::runtime_error("node id out of range");
        return std::string(im.c_name_blob + im.c_name_off[id - 1], im.c_name_off[id] - im.c_name_off[id - 1]);
    }
    if (id == 0 || id > im.node_names.size()) throw std::runtime_error("node id out of range");
    const Span sp = im.node_names[id - 1];
    return std::string(im.image.data() + sp.b, sp.e - sp.b);
}

std::vector<std::string> GraphStorage::edge_labels() const {
    const Impl &im = *impl_;
    require_edges("edge labels need the edge index");
    std::vector<std::string> out(edge_count_ + 1);
    if (im.cached) {
        for (uint64_t id = 1; id <= edge_count_; ++id) {
            const uint32_t u = (uint32_t)(im.c_edge_uv[id] >> 32), v = (uint32_t)im.c_edge_uv[id];
            const char o1 = (im.c_edge_oo[id] >> 1) & 1 ? '<' : '>', o2 = im.c_edge_oo[id] & 1 ? '<' : '>';
            out[id] = std::string(1, o1) + node_name(u) + std::string(1, o2) + node_name(v);
        }
        return out;
    }
    for (const auto &sl : im.edges.tab) {
        if (!sl.id) continue;
        const uint32_t u = (uint32_t)(sl.uv >> 32), v = (uint32_t)sl.uv;
        const char o1 = (sl.oo >> 1) & 1 ? '<' : '>', o2 = sl.oo & 1 ? '<' : '>';
        out[sl.id] = std::string(1, o1) + node_name(u) + std::string(1, o2) + node_name(v);
    }
    return out;
}

std::vector<uint64_t> GraphStorage::edge_keys() const {
    const Impl &im = *impl_;
    require_edges("edge keys need the edge index");
    std::vector<uint64_t> keys(edge_count_ + 1, 0);
    if (im.cached) {
        for (uint64_t id = 1; id <= edge_count_; ++id) keys[id] = im.c_edge_uv[id];
    } else {
        for (const auto &sl : im.edges.tab)
            if (sl.id) keys[sl.id] = sl.uv;
    }
    return keys;
}

void GraphStorage::edge_ends(std::vector<uint64_t> &uv, std::vector<uint8_t> &oo) const {
    const Impl &im = *impl_;
    require_edges("edge ends need the edge index");
    uv.assign(edge_count_ + 1, 0);
    oo.assign(edge_count_ + 1, 0);
    if (im.cached) {
        for (uint64_t id = 1; id <= edge_count_; ++id) {
            uv[id] = im.c_edge_uv[id];
            oo[id] = im.c_edge_oo[id];
        }
        return;
    }
    if (im.edge_uv_by_id.size() == edge_count_ + 1) {  // kept in id order when the L lines were read
        uv = im.edge_uv_by_id;
        oo = im.edge_oo_by_id;
        return;
    }
    // (every id is written by exactly one slot: the slices of the table go to the pool)
    const size_t n_slots = im.edges.tab.size(), slice = (size_t)1 << 18;
    ThreadPool::instance().parallel_for((n_slots + slice - 1) / slice, [&](size_t t) {
        const size_t e = std::min(n_slots, (t + 1) * slice);
        for (size_t k = t * slice; k < e; ++k) {
            const auto &sl = im.edges.tab[k];
            if (sl.id) {
                uv[sl.id] = sl.uv;
                oo[sl.id] = sl.oo;
            }
        }
    });
}

std::vector<uint32_t> GraphStorage::edge_relabel() const {
    const Impl &im = *impl_;
    require_edges("edge renumbering needs the edge index");
    struct Key {
        uint64_t uv;
        uint32_t id;
        uint8_t oo;
    };
    std::vector<Key> keys(edge_count_);
    if (im.cached) {
        for (uint64_t id = 1; id <= edge_count_; ++id) keys[id - 1] = Key{im.c_edge_uv[id], (uint32_t)id, im.c_edge_oo[id]};
    } else {
        for (const auto &sl : im.edges.tab)
            if (sl.id) keys[sl.id - 1] = Key{sl.uv, sl.id, sl.oo};
    }
    auto less = [](const Key &a, const Key &b) { return a.uv != b.uv ? a.uv < b.uv : a.oo < b.oo; };
    if (std::is_sorted(keys.begin(), keys.end(), less)) return {};  // the file's link order already is the rank
    // sorted pieces in parallel, then pairwise merges
    ThreadPool &pool = ThreadPool::instance();
    const size_t n = keys.size();
    size_t pieces = 1;
    while (pieces < pool.size() && n / (pieces * 2) >= (1u << 16)) pieces *= 2;
    auto bound = [&](size_t k) { return n / pieces * k + std::min(k, n % pieces); };
    pool.parallel_for(pieces, [&](size_t k) { std::sort(keys.begin() + (ptrdiff_t)bound(k), keys.begin() + (ptrdiff_t)bound(k + 1), less); });
    for (size_t width = 1; width < pieces; width *= 2) {
        const size_t n_merges = pieces / (2 * width);
        pool.parallel_for(n_merges, [&](size_t m) {
            const size_t a = bound(2 * width * m), mid = bound(2 * width * m + width), b = bound(2 * width * (m + 1));
            std::inplace_merge(keys.begin() + (ptrdiff_t)a, keys.begin() + (ptrdiff_t)mid, keys.begin() + (ptrdiff_t)b, less);
        });
    }
    std::vector<uint32_t> new_id(edge_count_ + 1, 0);
    pool.parallel_for((n + 65535) / 65536, [&](size_t c) {
        const size_t e = std::min(n, (c + 1) * 65536);
        for (size_t r = c * 65536; r < e; ++r) new_id[keys[r].id] = (uint32_t)r + 1;
    });
    return new_id;
}

std::vector<uint8_t> GraphStorage::exclude_flags(CountType count, const ItemTable &table, GroupMode mode,
                                                 const std::string &group_file, const std::string &exclude_file) const {
    const size_t P = paths_.size();
    std::vector<std::string> key(P);
    for (size_t i = 0; i < P; ++i) key[i] = paths_[i].clear_key();
    std::vector<std::string> group = load_groups(paths_, key, mode, group_file);
    std::vector<uint8_t> ex;
    read_path_list(exclude_file, paths_, key, group, false, ex, nullptr);
    std::vector<uint8_t> flags(number_of_items(count) + 1, 0);
    for (size_t p = 0; p < P; ++p)
        if (ex[p])
            for (uint64_t j = table.id_prefsum[p]; j < table.id_prefsum[p + 1]; ++j) flags[table.items[j]] = 1;
    return flags;
}

// ------------------------------------------------------------------------------------------
// subset / exclude lists with coordinates
// ------------------------------------------------------------------------------------------
namespace {

struct Iv {
    uint64_t s, e;  // [s, e), path or node coordinates
    bool operator<(const Iv &o) const { return s != o.s ? s < o.s : e < o.e; }
    bool operator==(const Iv &o) const { return s == o.s && e == o.e; }
};
constexpr uint64_t USIZE_MAX = ~0ull;
using IvMap = std::unordered_map<std::string, std::vector<Iv>>;

// the coordinate list of a BED file per path id: load_coord_list_file +
// complement_with_group_assignments (abacus.rs:152-210) + build_subpath_map (:354-382).
// Entries without coordinates and group members stand for the whole path, (0, usize::MAX).
IvMap load_subpath_map(const std::string &file, const std::vector<PathSegment> &paths, const std::vector<std::string> &key,
                       const std::vector<std::string> &group) {
    std::unordered_map<std::string, bool> known_key;
    std::unordered_map<std::string, std::vector<uint32_t>> by_group;
    for (size_t i = 0; i < paths.size(); ++i) {
        known_key.emplace(key[i], true);
        by_group[group[i]].push_back((uint32_t)i);
    }
    IvMap m;
    for (const PathSegment &ps : load_coord_list(file, paths)) {
        if (known_key.count(ps.clear_key())) {
            m[ps.id()].push_back(ps.has_start && ps.has_end ? Iv{ps.start, ps.end} : Iv{0, USIZE_MAX});
            continue;
        }
        auto bg = by_group.find(ps.id());
        if (bg == by_group.end()) continue;  // unknown path/group: logged and skipped
        if (ps.has_start && ps.has_end)
            throw std::runtime_error("invalid coordinate \"" + ps.display() +
                                     "\": group identifiers are not allowed to have start/stop information!");
        for (uint32_t i : bg->second) m[paths[i].id()].push_back(Iv{0, USIZE_MAX});
    }
    for (auto &kv : m) {  // a set of intervals, sorted, overlapping or touching ones joined
        std::vector<Iv> &v = kv.second;
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        std::vector<Iv> out;
        for (const Iv &x : v) {
            if (!out.empty() && out.back().e >= x.s)
                out.back().e = std::max(out.back().e, x.e);
            else
                out.push_back(x);
        }
        v.swap(out);
    }
    return m;
}

// intersects / is_contained (src/util.rs:370-398) on sorted disjoint intervals
bool any_touching(const std::vector<Iv> *v, Iv el) {
    if (!v) return false;
    for (const Iv &x : *v)
        if (x.s <= el.e && x.e >= el.s) return true;
    return false;
}
bool any_containing(const std::vector<Iv> *v, Iv el) {
    if (!v) return false;
    for (const Iv &x : *v)
        if (x.s <= el.s && x.e >= el.e) return true;
    return false;
}

// IntervalContainer (src/util.rs:209-310) for one node.  `add` follows the reference's own steps (util.rs:215-258:
// position by start, then one of three cases) instead of a generic interval union: for proper pieces the two are the
// same, but a BED row with start > end makes update_tables hand over a piece with a > b (a node that holds both ends
// of such a row), and what the container then holds is whatever these steps make of it.
struct Pieces {
    std::vector<Iv> v;
    void add(uint64_t start, uint64_t end) {
        if (v.empty()) {
            v.push_back(Iv{start, end});
            return;
        }
        size_t i = 0;  // binary_search_by_key(&start, |(y, _)| y).unwrap_or_else(|z| z): starts are unique and sorted
        while (i < v.size() && v[i].s < start) ++i;
        if (i > 0 && v[i - 1].e >= start) {
            if (v[i - 1].e < end) {
                uint64_t stop = end;
                while (i < v.size() && v[i].s <= end) {
                    stop = std::max(stop, v[i].e);
                    v.erase(v.begin() + (ptrdiff_t)i);
                }
                v[i - 1].e = stop;
            }  // else: enclosed in the previous piece
        } else if (i < v.size() && v[i].e >= start && v[i].s <= end) {
            v[i].s = std::min(v[i].s, start);
            uint64_t stop = std::max(v[i].e, end);
            while (i + 1 < v.size() && v[i + 1].s <= end) {
                stop = std::max(stop, v[i + 1].e);
                v.erase(v.begin() + (ptrdiff_t)i + 1);
            }
            v[i].e = stop;
        } else {
            v.insert(v.begin() + (ptrdiff_t)i, Iv{start, end});
        }
    }
};

// IntervalContainer::total_coverage (src/util.rs:272-305) in the reference's own arithmetic
// (usize, wrapping as in a release build): the bp of `v` outside the exclude pieces `ex`
uint64_t total_coverage(const std::vector<Iv> &v, const std::vector<Iv> *ex) {
    uint64_t res = 0;
    size_t i = 0;
    for (const Iv &x : v) {
        if (!ex) {
            res += x.e - x.s;
            continue;
        }
        while (i < ex->size() && (*ex)[i].e <= x.s) ++i;
        if (i < ex->size() && (*ex)[i].s < x.e) {
            res += std::min((*ex)[i].s - 1, x.e) - x.s;
            if ((*ex)[i].e < x.e) res += x.e - (*ex)[i].e + 1;
        } else {
            res += x.e - x.s;
        }
    }
    return res;
}

// The pieces of one node [p, p + l) that a sorted interval list selects, in node coordinates
// (mirrored for a backward step); `cur` walks the list along the path and stays on an interval
// that reaches the end of the node (update_tables, util.rs:626-660 / 662-704).
template <typename F>
inline void node_pieces(const std::vector<Iv> &list, size_t &cur, uint64_t p, uint64_t l, bool backward, F &&emit) {
    while (cur < list.size() && list[cur].s < p + l) {
        const Iv &x = list[cur];
        if (x.e <= p) {  // lies before the node
            ++cur;
            continue;
        }
        uint64_t a = x.s > p ? x.s - p : 0;
        const bool ends_inside = x.e < p + l;
        uint64_t b = ends_inside ? x.e - p : l;
        if (backward) {
            const uint64_t ma = l - b, mb = l - a;
            a = ma;
            b = mb;
        }
        emit(a, b);
        if (!ends_inside) return;
        ++cur;
    }
}

}  // namespace

bool GraphStorage::from_cache_file() const { return impl_->cached; }

bool GraphStorage::steps_tokenisable_on_device() const { return !impl_->cached && (impl_->numeric_names || names_by_bytes_on_device()); }
// `nice: true` of the YAML runner (graph.rs:224-229): are the segment names the integers 1..N in the order of the S lines?
// A parsed graph knows; a graph from the .pcsr cache holds its names, and is asked once.
bool GraphStorage::names_are_ranks() const {
    const Impl &im = *impl_;
    if (!im.cached) return im.nice;
    for (uint32_t id = 1; id <= node_count_; ++id) {
        const char *b = im.c_name_blob + im.c_name_off[id - 1], *e = im.c_name_blob + im.c_name_off[id];
        char buf[12];
        const int n = std::snprintf(buf, sizeof buf, "%u", id);
        if ((size_t)(e - b) != (size_t)n || std::memcmp(b, buf, (size_t)n) != 0) return false;
    }
    return node_count_ > 0;
}
const char *GraphStorage::text_data() const { return impl_->image.data(); }
size_t GraphStorage::text_size() const { return impl_->image.size(); }
const std::vector<uint32_t> &GraphStorage::id_of_name() const { return impl_->id_of_name; }
const std::string &GraphStorage::name_prefix() const { return impl_->name_prefix; }
void GraphStorage::step_columns(std::vector<uint64_t> &col_begin, std::vector<uint64_t> &col_end, std::vector<uint8_t> &is_walk) const {
    const size_t P = impl_->step_fields.size();
    col_begin.resize(P);
    col_end.resize(P);
    for (size_t k = 0; k < P; ++k) {
        col_begin[k] = impl_->step_fields[k].b;
        col_end[k] = impl_->step_fields[k].e;
    }
    is_walk = impl_->is_walk;
}

namespace {
// how each path is treated under the lists (parse_gfa_paths_walks, util.rs:240-300)
enum { SKIP = 0, WHOLE = 1, WALK = 2 };  // = PNX_WALK_SKIP / _WHOLE / _CUT of the device ABI
struct MaskSetup {
    bool have_inc = false, have_exc = false;
    IvMap inc, exc;
    std::vector<Iv> complete = {Iv{0, USIZE_MAX}}, none;
    std::vector<uint8_t> how;
    std::vector<const std::vector<Iv> *> ic, ec;
};
}  // namespace

static void mask_setup(MaskSetup &ms, const std::vector<PathSegment> &paths_, CountType count, GroupMode mode,
                       const std::string &group_file, const std::string &subset_file, const std::string &exclude_file) {
    const size_t P = paths_.size();
    std::vector<std::string> key(P);
    for (size_t i = 0; i < P; ++i) key[i] = paths_[i].clear_key();
    const std::vector<std::string> group = load_groups(paths_, key, mode, group_file);
    ms.have_inc = !subset_file.empty();
    ms.have_exc = !exclude_file.empty();
    if (ms.have_inc) ms.inc = load_subpath_map(subset_file, paths_, key, group);
    if (ms.have_exc) ms.exc = load_subpath_map(exclude_file, paths_, key, group);
    ms.how.assign(P, SKIP);
    ms.ic.assign(P, nullptr);
    ms.ec.assign(P, nullptr);
    for (size_t k = 0; k < P; ++k) {
        const std::string id = paths_[k].id();
        ms.ic[k] = &ms.complete;
        ms.ec[k] = &ms.none;
        if (ms.have_inc) {
            auto it = ms.inc.find(id);
            ms.ic[k] = it == ms.inc.end() ? &ms.none : &it->second;
        }
        if (ms.have_exc) {
            auto it = ms.exc.find(id);
            if (it != ms.exc.end()) ms.ec[k] = &it->second;
        }
        const Iv span = paths_[k].has_start && paths_[k].has_end ? Iv{paths_[k].start, paths_[k].end} : Iv{0, USIZE_MAX};
        if (ms.have_inc && !any_touching(ms.ic[k], span) && !any_touching(ms.ec[k], span))
            ms.how[k] = SKIP;
        else if (count != COUNT_EDGE && (!ms.have_inc || any_containing(ms.ic[k], span)) &&
                 (!ms.have_exc || any_containing(ms.ec[k], span)))
            ms.how[k] = WHOLE;
        else
            ms.how[k] = WALK;
    }
}

MaskedTable GraphStorage::masked_table(CountType count, GroupMode mode, const std::string &group_file,
                                       const std::string &subset_file, const std::string &exclude_file) const {
    const Impl &im = *impl_;
    if (im.cached) throw std::runtime_error("subset / exclude lists need the GFA text: load the graph without the cache");
    if (count == COUNT_EDGE) require_edges("graph was loaded without edge index");
    const size_t P = paths_.size();
    const uint64_t n_items = number_of_items(count);
    MaskSetup ms;
    mask_setup(ms, paths_, count, mode, group_file, subset_file, exclude_file);
    const bool have_inc = ms.have_inc, have_exc = ms.have_exc;
    std::vector<uint8_t> &how = ms.how;
    const std::vector<const std::vector<Iv> *> &ic = ms.ic, &ec = ms.ec;
    const std::vector<Iv> &complete = ms.complete;

    Steps steps;
    parse_all_steps(im, paths_, node_count_, true, steps);

    MaskedTable out;
    out.table.id_prefsum.assign(P + 1, 0);
    if (have_exc) out.exclude.assign(n_items + 1, 0);
    // bp only: partially covered / partially excluded nodes (load_optional_subsetting, abacus.rs:384-425)
    const bool track_cov = count == COUNT_BP && have_inc, annotate = count == COUNT_BP && have_exc;
    std::unordered_map<uint32_t, Pieces> covered, partly_excluded;

    // Paths that are walked node by node go first, one after the other in file order (their
    // bookkeeping of partly covered nodes depends on it); whole paths are plain copies.
    std::vector<std::vector<uint32_t>> walked(P);
    std::atomic<int64_t> bad_edge_path{-1};
    auto flag = [&](uint32_t id) { __atomic_store_n(&out.exclude[id], (uint8_t)1, __ATOMIC_RELAXED); };
    auto walk_path = [&](size_t k) {
        const uint64_t b = steps.pref[k], e = steps.pref[k + 1];
        const uint32_t *ids = steps.ids.data() + b;
        const uint8_t *ori = steps.ori.data() + b;
        const uint64_t len = e - b;
        std::vector<uint32_t> &items = walked[k];
        if (how[k] == WHOLE) {
            if (!ec[k]->empty())  // every node of an excluded path is excluded as a whole (util.rs:1171-1181)
                for (uint64_t j = 0; j < len; ++j) flag(ids[j]);
        } else if (how[k] == WALK && count != COUNT_EDGE) {
            size_t ci = 0, cj = 0;
            uint64_t p = paths_[k].has_start && paths_[k].has_end ? paths_[k].start : 0;
            for (uint64_t j = 0; j < len; ++j) {
                const uint32_t sid = ids[j];
                const uint64_t l = node_lens_[sid];
                node_pieces(*ic[k], ci, p, l, ori[j] != 0, [&](uint64_t a, uint64_t bb) {
                    items.push_back(sid);  // once per piece, like the reference
                    if (!track_cov) return;
                    if (bb - a == l)
                        covered.erase(sid);  // seen in full: earlier partial sightings are dropped
                    else
                        covered[sid].add(a, bb);
                });
                node_pieces(*ec[k], cj, p, l, ori[j] != 0, [&](uint64_t a, uint64_t bb) {
                    if (!annotate) {
                        flag(sid);
                        return;
                    }
                    // ActiveTable::activate_n_annotate (src/util.rs:147-181)
                    if (bb - a == l) {
                        flag(sid);
                        partly_excluded.erase(sid);
                        return;
                    }
                    // activate_n_annotate (src/util.rs:147-181): a piece with start > end is reported and not added
                    // (the reference then unwraps the node's list, which panics if nothing was ever added: not modelled)
                    auto it = a <= bb ? partly_excluded.try_emplace(sid).first : partly_excluded.find(sid);
                    if (it == partly_excluded.end()) return;
                    if (a <= bb) it->second.add(a, bb);
                    if (!it->second.v.empty() && it->second.v[0] == Iv{0, l}) {
                        partly_excluded.erase(it);
                        flag(sid);
                    }
                });
                if (ci >= ic[k]->size() && cj >= ec[k]->size()) break;  // nothing left to meet
                p += l;
            }
        } else if (how[k] == WALK && len > 0) {
            // update_tables_edgecount (util.rs:723-795): an edge sits at the start of its second node
            size_t ci = 0, cj = 0;
            uint64_t p = (paths_[k].has_start && paths_[k].has_end ? paths_[k].start : 0) + node_lens_[ids[0]];
            for (uint64_t j = 0; j + 1 < len; ++j) {
                while (ci < ic[k]->size() && (*ic[k])[ci].e <= p) ++ci;
                while (cj < ec[k]->size() && (*ec[k])[cj].e <= p) ++cj;
                const uint64_t l = node_lens_[ids[j + 1]];
                uint64_t uv;
                uint8_t oo;
                canonical(ids[j], ori[j], ids[j + 1], ori[j + 1], uv, oo);
                const uint32_t eid = im.edges.find(uv, oo);
                if (!eid) {
                    bad_edge_path.store((int64_t)k);
                    return;
                }
                if (ci < ic[k]->size() && (*ic[k])[ci].s < p + l) items.push_back(eid);
                if (have_exc && cj < ec[k]->size() && (*ec[k])[cj].s < p + l)
                    flag(eid);
                else if (ci >= ic[k]->size() && cj >= ec[k]->size())
                    break;
                p += l;
            }
        }
    };
    // bp counts keep per-node interval state across paths (file order matters); node and edge
    // counts only set flags, so their paths are independent
    if (count == COUNT_BP) {
        for (size_t k = 0; k < P; ++k) walk_path(k);
    } else {
        ThreadPool::instance().parallel_for(P, walk_path);
    }
    if (bad_edge_path.load() >= 0) throw std::runtime_error("unknown edge in path " + paths_[(size_t)bad_edge_path.load()].display());
    for (size_t k = 0; k < P; ++k)
        out.table.id_prefsum[k + 1] = out.table.id_prefsum[k] + (how[k] == WHOLE ? steps.pref[k + 1] - steps.pref[k] : walked[k].size());
    out.table.items.resize(out.table.id_prefsum[P]);
    ThreadPool::instance().parallel_for(P, [&](size_t k) {
        uint32_t *dst = out.table.items.data() + out.table.id_prefsum[k];
        if (how[k] == WHOLE)
            std::copy(steps.ids.begin() + (ptrdiff_t)steps.pref[k], steps.ids.begin() + (ptrdiff_t)steps.pref[k + 1], dst);
        else
            std::copy(walked[k].begin(), walked[k].end(), dst);
    });

    // quantify_uncovered_bps (abacus.rs:1187-1229)
    if (track_cov) {
        for (const auto &kv : covered) {
            const uint32_t sid = kv.first;
            if (have_exc && out.exclude[sid]) continue;  // excluded as a whole: never counted
            const uint64_t l = node_lens_[sid];
            std::vector<Iv> ex_pieces;
            if (have_exc) {
                auto it = partly_excluded.find(sid);
                if (it != partly_excluded.end()) ex_pieces = it->second.v;
            }
            const uint64_t cov = total_coverage(kv.second.v, have_exc ? &ex_pieces : nullptr);
            if (cov > l) continue;  // "oops, total coverage is larger than node length": left alone
            out.uncovered.emplace_back(sid, l - cov);
        }
        std::sort(out.uncovered.begin(), out.uncovered.end());
    }
    return out;
}

bool GraphStorage::mask_is_path_level(CountType count, GroupMode mode, const std::string &group_file, const std::string &subset_file,
                                      const std::string &exclude_file, std::vector<uint8_t> &take) const {
    take.clear();
    if (!exclude_file.empty() || subset_file.empty() || count == COUNT_EDGE) return false;  // (edges are always walked, see mask_setup)
    MaskSetup ms;
    mask_setup(ms, paths_, count, mode, group_file, subset_file, exclude_file);
    take.assign(paths_.size(), 0);
    for (size_t k = 0; k < paths_.size(); ++k) {
        if (ms.how[k] == WALK) return false;
        take[k] = ms.how[k] == WHOLE ? 1 : 0;
    }
    return true;
}

WalkCut GraphStorage::walk_cut(CountType count, GroupMode mode, const std::string &group_file, const std::string &subset_file,
                               const std::string &exclude_file, bool with_walks) const {
    const Impl &im = *impl_;
    if (im.cached) throw std::runtime_error("subset / exclude lists need the GFA text: load the graph without the cache");
    if (count == COUNT_EDGE) require_edges("graph was loaded without edge index");
    const size_t P = paths_.size();
    MaskSetup ms;
    mask_setup(ms, paths_, count, mode, group_file, subset_file, exclude_file);
    WalkCut w;
    w.count = count;
    if (with_walks) {
        Steps steps;
        parse_all_steps(im, paths_, node_count_, true, steps);
        w.walk_node.swap(steps.ids);
        w.walk_backward.swap(steps.ori);
        w.walk_off.swap(steps.pref);
    }
    if (count == COUNT_EDGE) edge_ends(w.edge_uv, w.edge_oo);
    w.path_start.assign(P, 0);
    w.path_mode.assign(P, SKIP);
    w.inc_off.assign(P + 1, 0);
    if (ms.have_exc) w.exc_off.assign(P + 1, 0);
    for (size_t k = 0; k < P; ++k) {
        w.path_start[k] = paths_[k].has_start && paths_[k].has_end ? paths_[k].start : 0;
        // (a walked path with the whole-path interval and nothing to exclude is NOT the same as a path taken whole:
        // the walk drops zero-length nodes at coordinate 0, update_tables' `include_coords[i].0 < p + l`, util.rs:625)
        const uint8_t how = ms.how[k];
        w.path_mode[k] = how;
        if (how != SKIP)
            for (const Iv &x : *ms.ic[k]) {
                w.inc_iv.push_back(x.s);
                w.inc_iv.push_back(x.e);
            }
        w.inc_off[k + 1] = w.inc_iv.size() / 2;
        if (ms.have_exc) {
            if (how != SKIP)
                for (const Iv &x : *ms.ec[k]) {
                    w.exc_iv.push_back(x.s);
                    w.exc_iv.push_back(x.e);
                }
            w.exc_off[k + 1] = w.exc_iv.size() / 2;
        }
    }
    w.track_covered = count == COUNT_BP && ms.have_inc;
    w.max_events = count == COUNT_BP ? 2 * (w.inc_iv.size() / 2 + w.exc_iv.size() / 2) + 16 : 0;
    return w;
}

void GraphStorage::replay_piece_events(const WalkCut &cut, std::vector<PieceEvent> events,
                                       std::vector<std::pair<uint32_t, uint64_t>> &uncovered,
                                       std::vector<uint32_t> &late_flags) const {
    uncovered.clear();
    late_flags.clear();
    if (cut.count != COUNT_BP || events.empty()) return;
    const bool have_exc = !cut.exc_off.empty();
    // the order of the reference's walk: path by path, step by step, piece by piece
    std::sort(events.begin(), events.end(), [](const PieceEvent &x, const PieceEvent &y) {
        return x.step != y.step ? x.step < y.step : (x.kind != y.kind ? x.kind < y.kind : x.piece < y.piece);
    });
    std::unordered_map<uint32_t, Pieces> covered, partly_excluded;
    std::unordered_map<uint32_t, bool> flagged;  // nodes of events: excluded as a whole by the device's walk
    for (const PieceEvent &e : events) {
        flagged[e.item] = e.flagged != 0;
        if (e.kind == 0) {
            // a later full sighting drops every earlier partial one (covered.remove, util.rs:640-643)
            if (e.step + 1 > e.last_full) covered[e.item].add(e.a, e.b);
        } else if (!e.flagged && e.a <= e.b) {  // activate_n_annotate drops a piece with start > end (src/util.rs:163-170)
            partly_excluded[e.item].add(e.a, e.b);
        }
    }
    // ActiveTable::activate_n_annotate (src/util.rs:147-181): pieces that join to the whole node exclude it
    for (auto it = partly_excluded.begin(); it != partly_excluded.end();) {
        const uint64_t l = node_lens_[it->first];
        if (!it->second.v.empty() && it->second.v[0] == Iv{0, l}) {
            late_flags.push_back(it->first);
            flagged[it->first] = true;
            it = partly_excluded.erase(it);
        } else {
            ++it;
        }
    }
    std::sort(late_flags.begin(), late_flags.end());
    // quantify_uncovered_bps (abacus.rs:1187-1229)
    for (const auto &kv : covered) {
        const uint32_t sid = kv.first;
        if (have_exc && flagged[sid]) continue;
        const uint64_t l = node_lens_[sid];
        std::vector<Iv> ex_pieces;
        if (have_exc) {
            auto it = partly_excluded.find(sid);
            if (it != partly_excluded.end()) ex_pieces = it->second.v;
        }
        const uint64_t cov = total_coverage(kv.second.v, have_exc ? &ex_pieces : nullptr);
        if (cov > l) continue;
        uncovered.emplace_back(sid, l - cov);
    }
    std::sort(uncovered.begin(), uncovered.end());
}

PathOrder GraphStorage::path_order(GroupMode mode, const std::string &group_file, const std::string &order_file,
                                   const std::string &subset_file, const std::string &exclude_file) const {
    const size_t P = paths_.size();
    std::vector<std::string> key(P);
    for (size_t i = 0; i < P; ++i) key[i] = paths_[i].clear_key();
    std::vector<std::string> group = load_groups(paths_, key, mode, group_file);

    // get_path_order (abacus.rs:310-347): buckets by group, emitted whole at the first visit
    std::unordered_map<std::string, uint32_t> bucket_of_group, path_of_key;
    std::vector<std::vector<uint32_t>> buckets;
    std::vector<uint32_t> bucket(P);
    for (size_t i = 0; i < P; ++i) {
        auto it = bucket_of_group.find(group[i]);
        if (it == bucket_of_group.end()) {
            it = bucket_of_group.emplace(group[i], (uint32_t)buckets.size()).first;
            buckets.emplace_back();
        }
        bucket[i] = it->second;
        buckets[it->second].push_back((uint32_t)i);
        path_of_key.emplace(key[i], (uint32_t)i);
    }
    std::vector<uint32_t> visit;
    if (!order_file.empty()) {
        for (std::string l : read_lines(order_file)) {  // parse_bed_to_path_segments, 1-column form
            if (!l.empty() && l.back() == '\r') l.pop_back();
            std::string name = l.substr(0, l.find('\t'));
            if (name.rfind("browser ", 0) == 0 || name.rfind("track ", 0) == 0 || (!name.empty() && name[0] == '#')) continue;
            PathSegment ps = PathSegment::from_str(name);
            auto pk = path_of_key.find(ps.clear_key());
            if (pk != path_of_key.end()) {  // complement_with_group_assignments, abacus.rs:152-206
                visit.push_back(bucket[pk->second]);
            } else {
                auto bg = bucket_of_group.find(ps.id());
                if (bg != bucket_of_group.end()) visit.push_back(bg->second);
                // unknown path/group: logged and skipped by the reference
            }
        }
    } else if (!subset_file.empty()) {  // the subset list is the order source (abacus.rs:326-327)
        std::vector<uint8_t> tmp;
        std::vector<uint32_t> entries;
        read_path_list(subset_file, paths_, key, group, false, tmp, &entries);
        for (uint32_t i : entries) visit.push_back(bucket[i]);
    } else {
        std::vector<uint8_t> ex(P, 0);
        if (!exclude_file.empty()) read_path_list(exclude_file, paths_, key, group, true, ex, nullptr);
        for (size_t i = 0; i < P; ++i)
            if (!ex[i]) visit.push_back(bucket[i]);
    }
    // paths outside the subset have an empty item-table entry in the reference (skipped during the
    // parse, util.rs:88-105); dropping them from the order gives the same countables and groups
    std::vector<uint8_t> in_subset;
    if (!subset_file.empty()) read_path_list(subset_file, paths_, key, group, false, in_subset, nullptr);
    PathOrder out;
    std::vector<uint8_t> done(buckets.size(), 0);
    for (uint32_t b : visit) {
        if (done[b]) continue;
        done[b] = 1;
        for (uint32_t i : buckets[b]) {
            if (out.groups.empty() || out.groups.back() != group[i]) out.groups.push_back(group[i]);  // abacus.rs:555-559
            if (!in_subset.empty() && !in_subset[i]) continue;
            out.path_idx.push_back(i);
            out.group_id.push_back((uint32_t)out.groups.size() - 1);
        }
    }
    return out;
}

}  // namespace pnh
