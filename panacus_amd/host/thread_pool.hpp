// thread_pool.hpp -- small persistent worker pool for the host layer (closed-form growth,
// GFA parsing).  The reference uses one process-wide rayon pool (src/lib.rs:71-83); this is
// the C++ counterpart: workers are created once and reused, so a 1 ms job does not pay a
// thread-spawn per call.
#pragma once
#include <condition_variable>
#include <cstddef>
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
    // returns when all tasks are done.  Not re-entrant.
    void parallel_for(size_t n_tasks, const std::function<void(size_t)> &fn, unsigned max_threads = 0);
    void set_threads(unsigned n);  // like `panacus -t N` (src/lib.rs:110-127); 0 = all cores
    ~ThreadPool();

private:
    ThreadPool();
    void start(unsigned n_workers);
    void stop();
    void worker_loop(unsigned id);
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_tasks_ = 0, next_ = 0, pending_ = 0;
    unsigned active_limit_ = 0;
    uint64_t epoch_ = 0;
    bool quit_ = false;
};

}  // namespace pnh
