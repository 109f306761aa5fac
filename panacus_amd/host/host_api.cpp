// host_api.cpp -- C entry points of libpanacus_host.so (host layer above the device ABI):
// closed-form growth, GFA front end, table writers.  Bound from Python with ctypes
// (panacus_amd/hostlib.py) and linked into the panacus-amd CLI.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "growth_closed_form.hpp"

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
    if (branch == 0) g = pnh::calc_growth_union(h, c);
    else if (branch == 1) g = pnh::calc_growth_core(h, c);
    else g = pnh::calc_growth_quorum(h, c, q, n_threads);
    std::memcpy(out, g.data(), g.size() * sizeof(double));
    return (int64_t)g.size();
}

double pnh_choose_log2(uint64_t n, uint64_t k) { return pnh::choose_log2(n, k); }

}  // extern "C"
