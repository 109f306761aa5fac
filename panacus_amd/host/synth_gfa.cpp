#include "synth_gfa.hpp"

#include <algorithm>
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

}  // namespace pnh
