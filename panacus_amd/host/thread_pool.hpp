// thread_pool.hpp -- small persistent worker pool for the host layer (closed-form growth,
// GFA parsing).  The reference uses one process-wide rayon pool (src/lib.rs:71-83); this is
// the C++ counterpart: workers are created once and reused, tasks are handed out through one
// atomic counter, and idle workers spin briefly before they sleep, so a sub-millisecond job
// (the closed-form growth of one histogram) does not pay thread-spawn or wake-up latency.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace pnh {

class ThreadPool {
public:
    static ThreadPool &instance();
    unsigned size() const { return (unsigned)workers_.size() + 1; }
    // runs fn(task) for task in [0, n_tasks) on up to max_threads threads (caller included);
    // returns when all tasks are done.  Not re-entrant; calls are serialised.
    void parallel_for(size_t n_tasks, const std::function<void(size_t)> &fn, unsigned max_threads = 0);
    void set_threads(unsigned n);  // like `panacus -t N` (src/lib.rs:110-127); 0 = all usable cores
    static unsigned usable_cpus(); // hardware threads, limited by the cgroup's CPU bandwidth quota
    ~ThreadPool();

private:
    ThreadPool();
    void start(unsigned n_workers);
    void stop();
    void worker_loop(unsigned id);
    void drain(const std::function<void(size_t)> &fn, size_t n_tasks, uint64_t job);

    std::vector<std::thread> workers_;
    std::mutex call_mu_;  // serialises parallel_for callers
    std::mutex mu_;
    std::condition_variable cv_work_;
    std::atomic<uint64_t> epoch_{0};
    std::atomic<uint64_t> ticket_{0};  // (job id << 32) | next task index
    std::atomic<size_t> done_{0};
    std::atomic<int> sleepers_{0};
    // the current job, written by parallel_for inside a sequence lock: gen_ is odd while the three fields change and
    // 2 * (job id) while they describe that job -- a worker that woke up late for job k must not pair job k's ticket with
    // the fields of job k + 1 (see parallel_for)
    std::atomic<uint64_t> gen_{0};
    std::atomic<const std::function<void(size_t)> *> fn_{nullptr};
    std::atomic<size_t> n_tasks_{0};
    std::atomic<unsigned> active_limit_{0};
    std::atomic<bool> quit_{false};
    std::mutex err_mu_;
    std::exception_ptr first_error_;  // the first exception a task of the current job threw; rethrown by parallel_for
};

// PANACUS_AMD_HOST_TIMING=1: one stderr line per phase of the host front end (ms since the previous mark)
void phase_mark(const char *what);

}  // namespace pnh
