// Stress of pnh::ThreadPool::parallel_for: many small jobs back to back on more threads than cores; every task of every job must
// run exactly once and be finished when the call returns (tests/test_host_thread_pool.py compiles and runs this).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "thread_pool.hpp"
int main(int argc, char **argv) {
    const long jobs = argc > 1 ? atol(argv[1]) : 200000;
    std::vector<std::atomic<int>> cnt(64);
    long bad = 0;
    for (long k = 0; k < jobs; ++k) {
        const size_t n = 2 + (size_t)((k * 7) % 13);
        for (auto &c : cnt) c.store(0, std::memory_order_relaxed);
        std::atomic<int> total{0};
        pnh::ThreadPool::instance().parallel_for(n, [&](size_t i) {
            cnt[i].fetch_add(1, std::memory_order_relaxed);
            total.fetch_add(1, std::memory_order_relaxed);
        });
        int t = total.load();
        bool ok = t == (int)n;
        for (size_t i = 0; i < n; ++i) ok = ok && cnt[i].load() == 1;
        if (!ok) {
            if (++bad < 5) std::fprintf(stderr, "job %ld: n = %zu, total = %d\n", k, n, t);
        }
    }
    std::printf("%ld jobs, %ld bad\n", jobs, bad);
    return bad ? 1 : 0;
}
