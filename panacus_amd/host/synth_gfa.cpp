#include "synth_gfa.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <set>
#include <stdexcept>
#include <vector>

#include "thread_pool.hpp"

namespace pnh {
namespace {

inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t key(uint64_t seed, uint64_t stream) { return splitmix64(seed ^ (0xA0761D6478BD642Full * (stream + 1))); }
constexpr uint64_t ONE53 = 1ull << 53;

uint32_t node_len(uint64_t seed, uint64_t i) {
    uint64_t t = splitmix64(key(seed, 1) + i) >> 11;
    if (t < ONE53 / 100 * 55) return 1;
    uint64_t e = splitmix64(key(seed, 2) + i);
    uint64_t k = e ? (uint64_t)__builtin_clzll(e) : 63;
    if (k > 63) k = 63;
    uint64_t l = 1 + ((180 * ((k << 16) + (e & 0xFFFF))) >> 16);
    return (uint32_t)(l > 50000 ? 50000 : l);
}
uint64_t node_thr(uint64_t seed, uint64_t i, uint64_t n_paths) {
    uint64_t t = splitmix64(key(seed, 3) + i) >> 11;
    if (t < ONE53 / 100 * 20) return ONE53;
    if (t < ONE53 / 100 * 65) return ONE53 / n_paths;
    return splitmix64(key(seed, 4) + i) >> 11;
}

void append_uint(std::string &s, uint64_t v) {
    char buf[24];
    int n = 0;
    do {
        buf[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) s.push_back(buf[--n]);
}

}  // namespace

uint64_t write_pansyn_gfa(const std::string &file, uint64_t seed, uint32_t n_nodes, uint32_t n_paths,
                          bool with_links, bool sequences) {
    FILE *f = std::fopen(file.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + file);
    std::fputs("H\tVN:Z:1.0\n", f);
    std::vector<uint64_t> thr((size_t)n_nodes + 1, 0);
    {
        std::string buf;
        static const char ACGT[4] = {'A', 'C', 'G', 'T'};
        for (uint32_t i = 1; i <= n_nodes; ++i) {
            thr[i] = node_thr(seed, i, n_paths);
            buf += "S\t";
            append_uint(buf, i);
            buf += '\t';
            const uint32_t len = node_len(seed, i);
            if (sequences) {
                uint64_t h = splitmix64(key(seed, 8) + i);
                for (uint32_t k = 0; k < len; ++k) {
                    if ((k & 31) == 31) h = splitmix64(h);
                    buf += ACGT[(h >> (2 * (k & 31))) & 3];
                }
            } else {
                buf.append(len, 'A');
            }
            buf += '\n';
            if (buf.size() > (1u << 22)) {
                std::fwrite(buf.data(), 1, buf.size(), f);
                buf.clear();
            }
        }
        std::fwrite(buf.data(), 1, buf.size(), f);
    }
    // paths: text built in parallel, written in file order
    const uint64_t k5 = key(seed, 5);
    uint64_t total_steps = 0;
    std::set<std::pair<uint32_t, uint32_t>> links;  // (u, v), all '+' orientation
    const uint32_t BATCH = 32;
    for (uint32_t p0 = 0; p0 < n_paths; p0 += BATCH) {
        const uint32_t pn = std::min(BATCH, n_paths - p0);
        std::vector<std::string> text(pn);
        std::vector<std::vector<uint32_t>> ids(pn);
        ThreadPool::instance().parallel_for(pn, [&](size_t k) {
            const uint32_t p = p0 + (uint32_t)k;
            const uint64_t kp = splitmix64(k5 + p);
            std::vector<uint32_t> &v = ids[k];
            for (uint32_t i = 1; i <= n_nodes; ++i) {
                const uint64_t h = splitmix64(kp + i);
                if ((h >> 11) < thr[i]) {
                    v.push_back(i);
                    if ((h & 63) == 0) v.push_back(i);
                }
            }
            if (p % 16 == 15) std::reverse(v.begin(), v.end());
            std::string &s = text[k];
            const bool walk = p % 4 == 3;  // exercise W lines too
            if (walk) {
                s += "W\ts";
                append_uint(s, p / 2);
                s += '\t';
                append_uint(s, p % 2);
                s += "\tctg\t0\t";
                append_uint(s, v.size());
                s += '\t';
                for (uint32_t id : v) {
                    s += '>';
                    append_uint(s, id);
                }
                s += '\n';
            } else {
                s += "P\ts";
                append_uint(s, p / 2);
                s += '#';
                append_uint(s, p % 2);
                s += "#ctg\t";
                for (size_t q = 0; q < v.size(); ++q) {
                    if (q) s += ',';
                    append_uint(s, v[q]);
                    s += '+';
                }
                s += "\t*\n";
            }
        });
        for (uint32_t k = 0; k < pn; ++k) {
            std::fwrite(text[k].data(), 1, text[k].size(), f);
            total_steps += ids[k].size();
            if (with_links)
                for (size_t q = 0; q + 1 < ids[k].size(); ++q) links.emplace(ids[k][q], ids[k][q + 1]);
        }
    }
    if (with_links) {
        std::string buf;
        for (const auto &l : links) {
            buf += "L\t";
            append_uint(buf, l.first);
            buf += "\t+\t";
            append_uint(buf, l.second);
            buf += "\t+\t0M\n";
            if (buf.size() > (1u << 22)) {
                std::fwrite(buf.data(), 1, buf.size(), f);
                buf.clear();
            }
        }
        std::fwrite(buf.data(), 1, buf.size(), f);
    }
    std::fclose(f);
    return total_steps;
}


// ---- pggb-shaped graph ------------------------------------------------------------------------
namespace {

struct EdgeSet {  // open addressing, concurrent inserts (keys are never 0)
    std::vector<std::atomic<uint64_t>> slot;
    uint64_t mask;
    explicit EdgeSet(uint64_t want) {
        uint64_t n = 1024;
        while (n < want * 2) n <<= 1;
        slot = std::vector<std::atomic<uint64_t>>(n);
        for (auto &x : slot) x.store(0, std::memory_order_relaxed);
        mask = n - 1;
    }
    void insert(uint64_t k) {
        uint64_t i = splitmix64(k) & mask;
        for (;;) {
            uint64_t cur = slot[i].load(std::memory_order_relaxed);
            if (cur == k) return;
            if (cur == 0) {
                if (slot[i].compare_exchange_strong(cur, k, std::memory_order_relaxed)) return;
                if (cur == k) return;
            }
            i = (i + 1) & mask;
        }
    }
};

// an oriented step: id << 1 | backward
inline uint64_t edge_key(uint32_t a, uint32_t b) {
    // (u, ou) -> (v, ov) and (v, !ov) -> (u, !ou) are the same edge: keep the smaller writing
    const uint32_t ra = b ^ 1u, rb = a ^ 1u;
    const uint64_t k1 = ((uint64_t)a << 32) | b, k2 = ((uint64_t)ra << 32) | rb;
    return k1 < k2 ? k1 : k2;
}

}  // namespace

uint64_t write_pggb_like_gfa(const std::string &file, uint64_t seed, uint32_t n_nodes, uint32_t n_samples, bool sequences,
                             uint32_t *n_paths_out, uint64_t *n_edges_out, const std::string &name_prefix) {
    auto append_name = [&name_prefix](std::string &to, uint32_t id) {  // `12`, or `s12` / `utg12` with a prefix
        to += name_prefix;
        append_uint(to, id);
    };
    if (n_nodes < 100) throw std::runtime_error("pggb-shaped graph: at least 100 nodes");
    FILE *f = std::fopen(file.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + file);
    std::fputs("H\tVN:Z:1.0\n", f);
    // node kinds: thr[i] = ONE53 for backbone segments, else the carrier share of a variant node
    std::vector<uint64_t> thr((size_t)n_nodes + 1, 0);
    {
        std::string buf;
        static const char ACGT[4] = {'A', 'C', 'G', 'T'};
        for (uint32_t i = 1; i <= n_nodes; ++i) {
            const uint64_t t = splitmix64(key(seed, 11) + i) >> 11;
            uint32_t len;
            if (t < ONE53 / 100 * 45) {
                thr[i] = ONE53;
                len = node_len(seed ^ 0x5bd1e995u, i);
                if (len == 1) len = 2 + (uint32_t)(splitmix64(key(seed, 12) + i) % 60);
            } else {
                const uint64_t u = splitmix64(key(seed, 13) + i) >> 11;  // U-shaped: most variants are rare or near-fixed
                const double x = (double)u / (double)ONE53, fshare = (t & 1) ? x * x * x : 1.0 - x * x * x;
                thr[i] = (uint64_t)(fshare * (double)ONE53);
                len = (t % 10) < 7 ? 1 : 1 + (uint32_t)(splitmix64(key(seed, 14) + i) % 50);
            }
            buf += "S\t";
            append_name(buf, i);
            buf += '\t';
            if (sequences) {
                uint64_t h = splitmix64(key(seed, 8) + i);
                for (uint32_t k = 0; k < len; ++k) {
                    if ((k & 31) == 31) h = splitmix64(h);
                    buf += ACGT[(h >> (2 * (k & 31))) & 3];
                }
            } else {
                buf.append(len, 'A');
            }
            buf += '\n';
            if (buf.size() > (1u << 22)) {
                std::fwrite(buf.data(), 1, buf.size(), f);
                buf.clear();
            }
        }
        std::fwrite(buf.data(), 1, buf.size(), f);
    }
    const uint32_t n_haps = 2 * n_samples + 2;  // two references first
    EdgeSet edges((uint64_t)n_nodes * 4 + 1024);
    uint64_t total_steps = 0;
    uint32_t n_paths = 0;
    const uint32_t BATCH = 16;
    for (uint32_t h0 = 0; h0 < n_haps; h0 += BATCH) {
        const uint32_t hn = std::min(BATCH, n_haps - h0);
        std::vector<std::string> text(hn);
        std::vector<uint64_t> steps(hn, 0);
        std::vector<uint32_t> paths(hn, 0);
        ThreadPool::instance().parallel_for(hn, [&](size_t k) {
            const uint32_t h = h0 + (uint32_t)k;
            const bool reference = h < 2;
            const uint64_t kh = splitmix64(key(seed, 15) + h);
            // the haplotype's walk over the whole chromosome, oriented steps
            std::vector<uint32_t> walk;
            walk.reserve((size_t)n_nodes * 3 / 4);
            for (uint32_t i = 1; i <= n_nodes; ++i) {
                const uint64_t r = splitmix64(kh + i);
                if ((r >> 11) >= thr[i] || (thr[i] == ONE53 && (r & 1023) < 3)) continue;  // not carried / a small deletion
                walk.push_back(i << 1);
            }
            if (!reference) {  // structural events: inversions and tandem duplications
                std::vector<uint32_t> w2;
                w2.reserve(walk.size() + walk.size() / 50);
                size_t q = 0;
                while (q < walk.size()) {
                    const uint64_t r = splitmix64(kh ^ (0x9E37ull * (q + 1)));
                    const uint64_t ev = r % 120000;
                    if (ev == 0 && q + 20 < walk.size()) {  // inversion
                        size_t l = 20 + (size_t)((r >> 20) % 3000);
                        l = std::min(l, walk.size() - q);
                        for (size_t x = 0; x < l; ++x) w2.push_back(walk[q + l - 1 - x] | 1u);
                        q += l;
                    } else if (ev == 1 && q + 5 < walk.size()) {  // tandem duplication
                        size_t l = 5 + (size_t)((r >> 20) % 500);
                        l = std::min(l, walk.size() - q);
                        for (int rep = 0; rep < 2; ++rep)
                            for (size_t x = 0; x < l; ++x) w2.push_back(walk[q + x]);
                        q += l;
                    } else {
                        w2.push_back(walk[q++]);
                    }
                }
                walk.swap(w2);
            }
            // contigs: references are one path, assembled haplotypes 3..26 pieces with gaps between them
            std::vector<std::pair<size_t, size_t>> pieces;
            if (reference) {
                pieces.emplace_back(0, walk.size());
            } else {
                const uint32_t nc = 3 + (uint32_t)(splitmix64(kh ^ 77) % 24);
                size_t at = (size_t)(splitmix64(kh ^ 78) % (walk.size() / 50 + 1));
                for (uint32_t c = 0; c < nc && at < walk.size(); ++c) {
                    const uint64_t r = splitmix64(kh ^ (1000 + c));
                    size_t len = walk.size() / nc / 2 + (size_t)(r % (walk.size() / nc + 1));
                    if (c + 1 == nc || at + len > walk.size()) len = walk.size() - at;
                    pieces.emplace_back(at, at + len);
                    at += len + (size_t)((r >> 32) % (walk.size() / 100 + 1));
                }
            }
            std::string &s = text[k];
            for (size_t c = 0; c < pieces.size(); ++c) {
                const size_t b = pieces[c].first, e = pieces[c].second;
                if (e <= b) continue;
                s += "P\t";
                if (reference) {
                    s += h == 0 ? "chm13#chr22" : "grch38#chr22";
                } else {
                    char nm[64];
                    std::snprintf(nm, sizeof nm, "HG%05u#%u#JA%04u%02zu.1", (h - 2) / 2 + 1, (h - 2) % 2 + 1, h, c);
                    s += nm;
                }
                s += '\t';
                for (size_t q = b; q < e; ++q) {
                    if (q > b) {
                        s += ',';
                        edges.insert(edge_key(walk[q - 1], walk[q]));
                    }
                    append_name(s, walk[q] >> 1);
                    s += (walk[q] & 1u) ? '-' : '+';
                }
                s += "\t*\n";
                steps[k] += e - b;
                paths[k] += 1;
            }
        });
        for (uint32_t k = 0; k < hn; ++k) {
            std::fwrite(text[k].data(), 1, text[k].size(), f);
            total_steps += steps[k];
            n_paths += paths[k];
        }
    }
    std::vector<uint64_t> all;
    for (auto &x : edges.slot) {
        const uint64_t k = x.load(std::memory_order_relaxed);
        if (k) all.push_back(k);
    }
    std::sort(all.begin(), all.end());
    {
        std::string buf;
        for (uint64_t k : all) {
            const uint32_t a = (uint32_t)(k >> 32), b = (uint32_t)k;
            buf += "L\t";
            append_name(buf, a >> 1);
            buf += (a & 1u) ? "\t-\t" : "\t+\t";
            append_name(buf, b >> 1);
            buf += (b & 1u) ? "\t-\t0M\n" : "\t+\t0M\n";
            if (buf.size() > (1u << 22)) {
                std::fwrite(buf.data(), 1, buf.size(), f);
                buf.clear();
            }
        }
        std::fwrite(buf.data(), 1, buf.size(), f);
    }
    std::fclose(f);
    if (n_paths_out) *n_paths_out = n_paths;
    if (n_edges_out) *n_edges_out = all.size();
    return total_steps;
}

}  // namespace pnh
