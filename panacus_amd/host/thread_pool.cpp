#include "thread_pool.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#if defined(__x86_64__)
#include <immintrin.h>
#define PNH_PAUSE() _mm_pause()
#else
#define PNH_PAUSE() ((void)0)
#endif

namespace pnh {

ThreadPool &ThreadPool::instance() {
    static ThreadPool pool;
    return pool;
}

// CPUs this process may actually use: the affinity mask, cut down to the CFS bandwidth quota of
// its cgroup (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1).  Running more busy
// threads than the quota gets the whole cgroup throttled for the rest of the 100 ms period --
// including the thread that drives the GPU.
unsigned ThreadPool::usable_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    auto read_two = [](const char *path, double &a, double &b) {
        FILE *f = std::fopen(path, "r");
        if (!f) return false;
        char x[64] = {0}, y[64] = {0};
        const int k = std::fscanf(f, "%63s %63s", x, y);
        std::fclose(f);
        if (k < 1 || std::strcmp(x, "max") == 0) return false;
        a = std::atof(x);
        b = k >= 2 ? std::atof(y) : 0.0;
        return a > 0;
    };
    double quota = 0, period = 0;
    if (read_two("/sys/fs/cgroup/cpu.max", quota, period) && period > 0) {
        n = std::min(n, std::max(1u, (unsigned)(quota / period)));
    } else {
        double q1 = 0, p1 = 0, dummy = 0;
        if (read_two("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q1, dummy) &&
            read_two("/sys/fs/cgroup/cpu/cpu.cfs_period_us", p1, dummy) && p1 > 0)
            n = std::min(n, std::max(1u, (unsigned)(q1 / p1)));
    }
    return n;
}

ThreadPool::ThreadPool() {
    unsigned n = usable_cpus();
    // under a CPU quota the pool leaves room for the threads beside it: the one that brings the GPU up and copies the
    // GFA text to it while the pool parses (commands.cpp), the HIP runtime's own.  One busy thread too many and the whole
    // cgroup -- that thread included -- is throttled for the rest of the period.
    if (n > 4 && n < std::max(1u, std::thread::hardware_concurrency())) n -= 2;
    if (const char *e = std::getenv("PANACUS_AMD_THREADS")) {
        int v = std::atoi(e);
        if (v > 0) n = (unsigned)v;
    }
    n = std::min(n, 64u);  // the host jobs here are small; more workers only add wake-up cost
    start(n - 1);
}

ThreadPool::~ThreadPool() { stop(); }

void ThreadPool::start(unsigned n_workers) {
    quit_.store(false);
    for (unsigned i = 0; i < n_workers; ++i) workers_.emplace_back([this, i]() { worker_loop(i); });
}

void ThreadPool::stop() {
    quit_.store(true);
    {
        std::lock_guard<std::mutex> lk(mu_);
        epoch_.fetch_add(1);
    }
    cv_work_.notify_all();
    for (auto &t : workers_) t.join();
    workers_.clear();
}

void ThreadPool::set_threads(unsigned n) {
    std::lock_guard<std::mutex> call(call_mu_);
    if (n == 0) n = usable_cpus();
    stop();
    start(n - 1);
}

// Tasks are handed out through one 64-bit ticket = (job id << 32) | next index, advanced by
// compare-exchange: a worker that is late for job k can never take (or run with a stale
// function pointer) a task of job k+1, because its job id no longer matches the ticket.
void ThreadPool::drain(const std::function<void(size_t)> &fn, size_t n_tasks, uint64_t job) {
    for (;;) {
        uint64_t t = ticket_.load(std::memory_order_acquire);
        if ((t >> 32) != (job & 0xFFFFFFFFull)) return;
        const size_t idx = (size_t)(t & 0xFFFFFFFFull);
        if (idx >= n_tasks) return;
        if (!ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel)) continue;
        try {
            fn(idx);
        } catch (...) {  // kept for the caller: a task must never unwind through the pool (the job would never complete)
            std::lock_guard<std::mutex> lk(err_mu_);
            if (!first_error_) first_error_ = std::current_exception();
        }
        done_.fetch_add(1, std::memory_order_release);
    }
}

void ThreadPool::worker_loop(unsigned id) {
    uint64_t seen = epoch_.load();
    for (;;) {
        // wait for a new epoch: spin for a short while (jobs arrive ~1 ms apart in the
        // pipelined histgrowth loop), then sleep on the condition variable
        const auto t0 = std::chrono::steady_clock::now();
        uint64_t e;
        int spins = 0;
        while ((e = epoch_.load(std::memory_order_acquire)) == seen) {
            PNH_PAUSE();
            if ((++spins & 1023) == 0 &&
                std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(60)) {
                std::unique_lock<std::mutex> lk(mu_);
                sleepers_.fetch_add(1);
                cv_work_.wait(lk, [&]() { return epoch_.load() != seen; });
                sleepers_.fetch_sub(1);
            }
        }
        seen = e;
        if (quit_.load()) return;
        // the job of this epoch, read inside the sequence lock: if anything else than job e stands there -- before or after
        // the fields were read -- the job is over (its caller has moved on) and there is nothing for this worker to do
        const uint64_t g1 = gen_.load(std::memory_order_acquire);
        const std::function<void(size_t)> *fn = fn_.load(std::memory_order_relaxed);
        const size_t n_tasks = n_tasks_.load(std::memory_order_relaxed);
        const unsigned limit = active_limit_.load(std::memory_order_relaxed);
        std::atomic_thread_fence(std::memory_order_acquire);
        const uint64_t g2 = gen_.load(std::memory_order_relaxed);
        if (g1 != 2 * e || g2 != 2 * e) continue;
        if (fn && id + 1 < limit) drain(*fn, n_tasks, e);
    }
}

void ThreadPool::parallel_for(size_t n_tasks, const std::function<void(size_t)> &fn, unsigned max_threads) {
    if (n_tasks == 0) return;
    std::lock_guard<std::mutex> call(call_mu_);
    unsigned limit = max_threads ? std::min(max_threads, size()) : size();
    if (limit <= 1 || n_tasks == 1 || workers_.empty()) {
        for (size_t t = 0; t < n_tasks; ++t) fn(t);
        return;
    }
    if (n_tasks >= 0xFFFFFFFFull) throw std::runtime_error("parallel_for: too many tasks");
    // Publish the job.  A worker may still be on its way into the PREVIOUS job (it saw that job's epoch, all of whose tasks
    // other threads have taken since): it must find nothing to do.  It used to be able to read the NEW function and task
    // count while the ticket still named the old job -- an index below the new count, a task of the new job run as part of
    // the old one: run twice, and counted into done_ so that this call returned one task early (seen once in ~50,000 CLI runs
    // of the fuzz soak: a table row missing, a crash).  Hence: (1) the ticket is closed under the NEW job's id first, so no
    // index can be taken from now on; (2) the fields change inside a sequence lock that the workers check on both sides of
    // their reads; (3) only then the ticket opens and the epoch moves.
    const uint64_t job = epoch_.load(std::memory_order_relaxed) + 1;
    ticket_.store(((job & 0xFFFFFFFFull) << 32) | 0xFFFFFFFFull, std::memory_order_seq_cst);
    gen_.store(2 * job + 1, std::memory_order_seq_cst);
    std::atomic_thread_fence(std::memory_order_seq_cst);
    fn_.store(&fn, std::memory_order_relaxed);
    n_tasks_.store(n_tasks, std::memory_order_relaxed);
    active_limit_.store(limit, std::memory_order_relaxed);
    done_.store(0, std::memory_order_relaxed);
    gen_.store(2 * job, std::memory_order_release);
    ticket_.store((job & 0xFFFFFFFFull) << 32, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lk(mu_);
        epoch_.store(job, std::memory_order_release);
    }
    if (sleepers_.load() > 0) cv_work_.notify_all();
    drain(fn, n_tasks, job);
    // the last tasks run on other threads: spin briefly (jobs here are sub-millisecond), then give the CPU away -- under a
    // CFS quota shared with other busy processes a spinning waiter takes the time its own workers need
    for (unsigned spins = 0; done_.load(std::memory_order_acquire) < n_tasks;) {
        if (++spins < 4096) PNH_PAUSE();
        else std::this_thread::yield();
    }
    std::exception_ptr err;
    {
        std::lock_guard<std::mutex> lk(err_mu_);
        std::swap(err, first_error_);
    }
    if (err) std::rethrow_exception(err);
}

void phase_mark(const char *what) {
    static const bool on = std::getenv("PANACUS_AMD_HOST_TIMING") != nullptr;
    if (!on) return;
    static std::mutex mu;
    static auto t_prev = std::chrono::steady_clock::now(), t0 = t_prev;
    std::lock_guard<std::mutex> g(mu);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[host phase] %-34s %9.3f ms  (at %9.3f ms)\n", what,
                 std::chrono::duration<double, std::milli>(now - t_prev).count(),
                 std::chrono::duration<double, std::milli>(now - t0).count());
    t_prev = now;
}

}  // namespace pnh
