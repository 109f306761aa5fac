#include "thread_pool.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace pnh {

ThreadPool &ThreadPool::instance() {
    static ThreadPool pool;
    return pool;
}

ThreadPool::ThreadPool() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (const char *e = std::getenv("PANACUS_AMD_THREADS")) {
        int v = std::atoi(e);
        if (v > 0) n = (unsigned)v;
    }
    n = std::min(n, 64u);  // the host jobs here are small; more workers only add wake-up cost
    start(n - 1);
}

ThreadPool::~ThreadPool() { stop(); }

void ThreadPool::start(unsigned n_workers) {
    quit_ = false;
    for (unsigned i = 0; i < n_workers; ++i) workers_.emplace_back([this, i]() { worker_loop(i); });
}

void ThreadPool::stop() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        quit_ = true;
    }
    cv_work_.notify_all();
    for (auto &t : workers_) t.join();
    workers_.clear();
}

void ThreadPool::set_threads(unsigned n) {
    if (n == 0) n = std::max(1u, std::thread::hardware_concurrency());
    stop();
    start(n - 1);
}

void ThreadPool::worker_loop(unsigned id) {
    uint64_t seen = 0;
    for (;;) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_work_.wait(lk, [&]() { return quit_ || (epoch_ != seen && fn_ != nullptr); });
        if (quit_) return;
        seen = epoch_;
        if (id + 1 >= active_limit_) continue;  // this job wants fewer threads
        while (next_ < n_tasks_) {
            size_t t = next_++;
            lk.unlock();
            (*fn_)(t);
            lk.lock();
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }
}

void ThreadPool::parallel_for(size_t n_tasks, const std::function<void(size_t)> &fn, unsigned max_threads) {
    if (n_tasks == 0) return;
    unsigned limit = max_threads ? std::min(max_threads, size()) : size();
    if (limit <= 1 || n_tasks == 1 || workers_.empty()) {
        for (size_t t = 0; t < n_tasks; ++t) fn(t);
        return;
    }
    std::unique_lock<std::mutex> lk(mu_);
    fn_ = &fn;
    n_tasks_ = n_tasks;
    next_ = 0;
    pending_ = n_tasks;
    active_limit_ = limit;
    ++epoch_;
    cv_work_.notify_all();
    while (next_ < n_tasks_) {  // the caller works too
        size_t t = next_++;
        lk.unlock();
        fn(t);
        lk.lock();
        --pending_;
    }
    cv_done_.wait(lk, [&]() { return pending_ == 0; });
    fn_ = nullptr;
}

}  // namespace pnh
