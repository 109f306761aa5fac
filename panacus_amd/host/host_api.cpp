// host_api.cpp -- C entry points of libpanacus_host.so (host layer above the device ABI):
// closed-form growth, GFA front end, table writers.  Bound from Python with ctypes
// (panacus_amd/hostlib.py) and linked into the panacus-amd CLI.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "growth_closed_form.hpp"
#include "thread_pool.hpp"

extern "C" {

// Hist::calc_growth (src/graph_broker/hist.rs:51-66). out: hist_len-1 doubles. Returns n.
int64_t pnh_calc_growth(const uint64_t *hist, uint64_t hist_len, int cov_kind, double cov_val, int quo_kind,
                        double quo_val, unsigned n_threads, double *out) {
    if (!hist || hist_len < 2) return 0;
    std::vector<uint64_t> h(hist, hist + hist_len);
    std::vector<double> g = pnh::calc_growth(h, pnh::Threshold{cov_kind, cov_val}, pnh::Threshold{quo_kind, quo_val}, n_threads);
    std::memcpy(out, g.data(), g.size() * sizeof(double));
    return (int64_t)g.size();
}

// the three branches, callable directly (the reference's unit tests do the same, hist.rs:352-398)
int64_t pnh_calc_growth_branch(int branch, const uint64_t *hist, uint64_t hist_len, int cov_kind, double cov_val,
                               int quo_kind, double quo_val, unsigned n_threads, double *out) {
    if (!hist || hist_len < 2) return 0;
    std::vector<uint64_t> h(hist, hist + hist_len);
    std::vector<double> g;
    pnh::Threshold c{cov_kind, cov_val}, q{quo_kind, quo_val};
    if (branch == 0) g = pnh::calc_growth_union(h, c, n_threads);
    else if (branch == 1) g = pnh::calc_growth_core(h, c, n_threads);
    else g = pnh::calc_growth_quorum(h, c, q, n_threads);
    std::memcpy(out, g.data(), g.size() * sizeof(double));
    return (int64_t)g.size();
}

// several (coverage, quorum) pairs on one histogram: Hist::calc_all_growths (hist.rs:68-87,
// which runs the pairs on the rayon pool).  out: n_pairs x (hist_len-1) doubles, no NaN row.
int64_t pnh_calc_all_growths(const uint64_t *hist, uint64_t hist_len, const int *cov_kind, const double *cov_val,
                             const int *quo_kind, const double *quo_val, uint32_t n_pairs, unsigned n_threads,
                             double *out) {
    if (!hist || hist_len < 2) return 0;
    std::vector<uint64_t> h(hist, hist + hist_len);
    const uint64_t n = hist_len - 1;
    for (uint32_t t = 0; t < n_pairs; ++t) {
        std::vector<double> g = pnh::calc_growth(h, pnh::Threshold{cov_kind[t], cov_val[t]},
                                                 pnh::Threshold{quo_kind[t], quo_val[t]}, n_threads);
        std::memcpy(out + (size_t)t * n, g.data(), g.size() * sizeof(double));
    }
    return (int64_t)n;
}

void pnh_set_threads(unsigned n) { pnh::ThreadPool::instance().set_threads(n); }
unsigned pnh_get_threads(void) { return pnh::ThreadPool::instance().size(); }

double pnh_choose_log2(uint64_t n, uint64_t k) { return pnh::choose_log2(n, k); }

}  // extern "C"
