// growth_closed_form.hpp -- exact expected pangenome growth from a coverage histogram.
//
// Host-side mirror of Hist::calc_growth and its three branches
// (src/graph_broker/hist.rs:51-187 of the reference).  The values are f64 and the
// reference prints floor() of them, so the arithmetic here keeps the reference's operation
// order and uses the platform libm (glibc log2/exp2) exactly like Rust's f64::log2/exp2 on
// linux-gnu: results are bit-identical.  Work that does not affect rounding is shared
// (log2 of small integers is tabulated; the O(n^3) quorum branch is spread over threads
// along the independent histogram index i and summed in the reference's order).
#pragma once
#include <cstdint>
#include <vector>

namespace pnh {

enum ThresholdKind { THR_ABSOLUTE = 0, THR_RELATIVE = 1 };
struct Threshold {
    int kind;
    double value;
    uint64_t to_absolute(uint64_t n) const;  // src/util.rs:350-355
    double to_relative(uint64_t n) const;    // src/util.rs:357-362
};

double choose_log2(uint64_t n, uint64_t k);  // hist.rs:21-36

// Quorum branch with n >= min_n: the O(n^3) inner sums run on the GPU of that pnx_ctx
// (pnx_quorum_sums, bit-identical by construction and guarded by a run-time check of the exp2
// restatement against libm).  nullptr switches it off.  The context must outlive the calls.
void set_quorum_offload(void *pnx_context, uint64_t min_n = 256);
// forget the context if it is the registered one (a context that goes away while a newer one is already registered)
void release_quorum_offload(void *pnx_context);
// the same for the CALLING THREAD only, ahead of the process-wide context: what the in-process CLI binds (one context per command,
// commands on several threads at once must not evaluate on each other's context); unbind clears it if it is still this context
void bind_thread_offload(void *pnx_context);
void unbind_thread_offload(void *pnx_context);
bool quorum_offload_usable();

// Hist::calc_all_growths (hist.rs:68-87) without the NaN row: one curve per (coverage, quorum)
// pair, all pairs evaluated in one parallel region
std::vector<std::vector<double>> calc_all_growths(const std::vector<uint64_t> &hist,
                                                  const std::vector<Threshold> &coverage,
                                                  const std::vector<Threshold> &quorum, unsigned n_threads = 0);

// The same in two halves: _begin sets the jobs up and enqueues the device part (if any) and
// returns; _end waits for it and does the host part.  A host that pipelines passes enqueues its next
// device pass in between.  Every handle must be passed to _end exactly once.
struct GrowthRun;
GrowthRun *calc_all_growths_begin(const std::vector<uint64_t> &hist, const std::vector<Threshold> &coverage,
                                  const std::vector<Threshold> &quorum, unsigned n_threads = 0);
std::vector<std::vector<double>> calc_all_growths_end(GrowthRun *run);
// the curves of the histogram of the coverage pass enqueued last on the offload context, computed without the histogram
// visiting the host (nullptr: not available -- fetch the histogram and use calc_all_growths_begin)
// only_ctx: take the device path only when THIS context is the offload context (a caller that just enqueued its pass on it)
GrowthRun *calc_all_growths_begin_on_device(uint64_t n_groups, const std::vector<Threshold> &coverage, const std::vector<Threshold> &quorum,
                                            const void *only_ctx = nullptr);
// before the coverage pass is enqueued: start the first part of the device tables of (n_groups, pairs) (false: not on the device)
bool growth_tables_begin(uint64_t n_groups, const std::vector<Threshold> &coverage, const std::vector<Threshold> &quorum, const void *only_ctx = nullptr);
// true iff the restated log2 AND exp2 reproduce this platform's libm bit for bit: whole closed forms may run on the device
bool device_growth_usable();
void log2_restated(const double *x, double *y, uint64_t n);

// n = hist.size() - 1 values (the caller prepends the NaN row, hist.rs:83-85)
std::vector<double> calc_growth(const std::vector<uint64_t> &hist, Threshold coverage, Threshold quorum,
                                unsigned n_threads = 0);
std::vector<double> calc_growth_union(const std::vector<uint64_t> &hist, Threshold coverage, unsigned n_threads = 0);
std::vector<double> calc_growth_core(const std::vector<uint64_t> &hist, Threshold coverage, unsigned n_threads = 0);
std::vector<double> calc_growth_quorum(const std::vector<uint64_t> &hist, Threshold coverage, Threshold quorum,
                                       unsigned n_threads = 0);

}  // namespace pnh
