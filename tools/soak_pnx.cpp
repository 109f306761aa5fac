// soak_pnx.cpp -- the device ABI alone (include/panacus_amd.h), no host library, no Python: context after context,
//   pnx_init -> pnx_set_csr -> pnx_set_order -> pnx_hist (-> pnx_ordered_growth -> pnx_group_intersections) -> pnx_free
// thousands of times, on one or several threads with a context each -- the way a host that binds the library in-process uses it
// (the reference calls the replaced functions from rayon workers, src/analyses/ordered_histgrowth.rs:174-188).  Splits
// libpanacus_hip.so from libpanacus_host.so in the hunt of round 5's intermittent abort; built plain / asan / tsan by
// tools/build_sanitized.sh.  Every result is compared with a serial count on the host (abacus.rs:719-787 in ten lines), so a
// race that changes a number shows as well as one that kills the process.
//
//   soak_pnx <iterations> [threads = 2] [contexts kept alive per thread = 1]
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "panacus_amd.h"

namespace {
std::atomic<uint64_t> g_fail{0}, g_ctx{0};

uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Graph {
    uint32_t n_items, n_paths, n_groups;
    std::vector<uint32_t> items, weights, path_idx, group_id;
    std::vector<uint64_t> off;
    std::vector<uint64_t> hist, hist_bp;  // expected
};

// a small random graph: paths mostly ascending with duplicates, some reversed, one shuffled; groups of 1-3 paths
Graph make_graph(uint64_t seed) {
    Graph g;
    g.n_items = 500 + (uint32_t)(splitmix(seed) % 6000);
    g.n_paths = 2 + (uint32_t)(splitmix(seed) % 12);
    g.off.push_back(0);
    for (uint32_t p = 0; p < g.n_paths; ++p) {
        std::vector<uint32_t> st;
        const uint64_t keep = 20 + splitmix(seed) % 80;
        for (uint32_t i = 1; i <= g.n_items; ++i)
            if (splitmix(seed) % 100 < keep) {
                st.push_back(i);
                if (splitmix(seed) % 64 == 0) st.push_back(i);
            }
        if (st.empty()) st.push_back(1);
        if (p % 5 == 3) std::reverse(st.begin(), st.end());
        if (p % 7 == 6)
            for (size_t k = st.size(); k > 1; --k) std::swap(st[k - 1], st[splitmix(seed) % k]);
        g.items.insert(g.items.end(), st.begin(), st.end());
        g.off.push_back(g.items.size());
    }
    g.weights.resize(g.n_items + 1);
    for (auto &w : g.weights) w = 1 + (uint32_t)(splitmix(seed) % 300);
    uint32_t grp = 0;
    for (uint32_t p = 0; p < g.n_paths; ++p) {
        g.path_idx.push_back(p);
        g.group_id.push_back(grp);
        if (splitmix(seed) % 3 != 0 && p + 1 < g.n_paths) ++grp;
    }
    g.n_groups = g.group_id.back() + 1;
    // AbacusByTotal::coverage + construct_hist(_bps) (abacus.rs:719-787), serially
    std::vector<uint32_t> cov(g.n_items + 1, 0), last(g.n_items + 1, UINT32_MAX);
    for (uint32_t k = 0; k < g.n_paths; ++k)
        for (uint64_t s = g.off[g.path_idx[k]]; s < g.off[g.path_idx[k] + 1]; ++s) {
            const uint32_t id = g.items[s];
            if (last[id] != g.group_id[k]) {
                last[id] = g.group_id[k];
                ++cov[id];
            }
        }
    g.hist.assign(g.n_groups + 1, 0);
    g.hist_bp.assign(g.n_groups + 1, 0);
    for (uint32_t i = 1; i <= g.n_items; ++i) {
        g.hist[cov[i]] += 1;
        g.hist_bp[cov[i]] += g.weights[i];
    }
    return g;
}

#define CK(call)                                                                                         \
    do {                                                                                                 \
        int rc_ = (call);                                                                                \
        if (rc_ != PNX_OK) {                                                                             \
            std::fprintf(stderr, "thread %d: %s -> %d: %s\n", tid, #call, rc_, pnx_last_error(ctx));     \
            g_fail.fetch_add(1);                                                                         \
            goto done;                                                                                   \
        }                                                                                                \
    } while (0)

void worker(int tid, int iters, int keep_alive) {
    uint64_t seed = 0x1234 + 7919ull * (uint64_t)tid;
    std::vector<pnx_ctx *> alive;
    for (int it = 0; it < iters; ++it) {
        const Graph g = make_graph(splitmix(seed));
        pnx_ctx *ctx = nullptr;
        const bool bp = it % 2 == 1;
        std::vector<uint64_t> hist(g.n_groups + 1, 0);
        std::vector<uint32_t> cnt(g.n_items + 1, 0);
        if (pnx_init_flags(&ctx, 0, (it % 3 == 0) ? PNX_INIT_ONE_SHOT : 0u) != PNX_OK) {
            std::fprintf(stderr, "thread %d: pnx_init: %s\n", tid, pnx_last_error(nullptr));
            g_fail.fetch_add(1);
            return;
        }
        g_ctx.fetch_add(1);
        CK(pnx_set_csr(ctx, g.items.data(), g.off.data(), g.n_paths, g.n_items, bp ? g.weights.data() : nullptr, nullptr));
        CK(pnx_set_order(ctx, g.path_idx.data(), g.group_id.data(), g.n_paths, g.n_groups));
        CK(pnx_hist(ctx, cnt.data(), hist.data()));
        if (hist != (bp ? g.hist_bp : g.hist)) {
            std::fprintf(stderr, "thread %d iteration %d: histogram differs from the serial count (n_items %u, paths %u, groups %u, bp %d)\n", tid,
                         it, g.n_items, g.n_paths, g.n_groups, (int)bp);
            g_fail.fetch_add(1);
        }
        if (it % 4 == 0) {  // a second sweep (the rows route) must repeat the first
            std::vector<uint64_t> h2(g.n_groups + 1, 0);
            CK(pnx_hist(ctx, nullptr, h2.data()));
            if (h2 != hist) {
                std::fprintf(stderr, "thread %d iteration %d: second sweep differs\n", tid, it);
                g_fail.fetch_add(1);
            }
        }
        if (it % 3 == 1) {  // ordered growth, identity order: the last value of the c = 1, q = 0 pair is the number (bp) of covered items
            const uint32_t cov_thr[2] = {1, 1};
            std::vector<uint32_t> qtab(2 * g.n_groups, 0);
            for (uint32_t j = 0; j < g.n_groups; ++j) qtab[g.n_groups + j] = (uint32_t)((j + 1.0) * 0.5 + 0.999999);
            std::vector<uint64_t> out(2 * (size_t)g.n_groups, 0);
            CK(pnx_ordered_growth(ctx, nullptr, 1, cov_thr, qtab.data(), 2, out.data()));
            uint64_t covered = 0;
            for (uint32_t c = 1; c <= g.n_groups; ++c) covered += (bp ? g.hist_bp : g.hist)[c];
            if (out[g.n_groups - 1] != covered) {
                std::fprintf(stderr, "thread %d iteration %d: ordered growth ends at %llu, covered %llu\n", tid, it,
                             (unsigned long long)out[g.n_groups - 1], (unsigned long long)covered);
                g_fail.fetch_add(1);
            }
        }
        if (it % 5 == 2) {
            std::vector<uint64_t> inter((size_t)g.n_groups * g.n_groups, 0);
            CK(pnx_group_intersections(ctx, inter.data()));
        }
    done:
        alive.push_back(ctx);
        if ((int)alive.size() > keep_alive - 1) {
            pnx_free(alive.front());
            alive.erase(alive.begin());
        }
        if (tid == 0 && (it + 1) % 200 == 0)
            std::fprintf(stderr, "iteration %d: %llu contexts, %llu failures\n", it + 1, (unsigned long long)g_ctx.load(), (unsigned long long)g_fail.load());
    }
    for (pnx_ctx *c : alive) pnx_free(c);
}
}  // namespace

int main(int argc, char **argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 100;
    const int threads = argc > 2 ? std::max(1, std::atoi(argv[2])) : 2;
    const int keep = argc > 3 ? std::max(1, std::atoi(argv[3])) : 1;
    std::vector<std::thread> ts;
    for (int k = 1; k < threads; ++k) ts.emplace_back(worker, k, iters, keep);
    worker(0, iters, keep);
    for (auto &t : ts) t.join();
    std::fprintf(stderr, "soak_pnx: %llu contexts, %llu failures\n", (unsigned long long)g_ctx.load(), (unsigned long long)g_fail.load());
    return g_fail.load() ? 1 : 0;
}
