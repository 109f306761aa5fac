// growth_closed_form.cpp -- see growth_closed_form.hpp.
//
// Structure: every (histogram, coverage, quorum) evaluation is a "job" = cheap serial
// prologue + independent row tasks + cheap serial epilogue.  The rows are the histogram
// index i: all state the reference carries from one m to the next (perc_mult[i], Q[i][*])
// lives inside a row, so rows can run on any thread in any order; the sums over i are then
// taken serially in ascending i, exactly like the reference's loops, which keeps every
// result bit-identical to the serial Rust code (same libm calls on the same arguments, same
// addition order).  All jobs of a calc_all_growths call are flattened into ONE parallel
// region so the pool is entered once per histogram, not once per branch.
#include "growth_closed_form.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>

#include "../csrc/exp2_exact.hpp"
#include "thread_pool.hpp"
#include "../csrc/log2_exact.hpp"

// libpanacus_hip (include/panacus_amd.h); declared here so that this file needs no HIP headers
struct pnx_ctx;
extern "C" int pnx_quorum_sums_async(pnx_ctx *ctx, uint32_t n, uint32_t c, const uint32_t *m_quorum,
                                     const double *log2_tab, const double *m_fact, const double *n_fall);
extern "C" int pnx_quorum_sums_fetch(pnx_ctx *ctx, const double **sum_q);
extern "C" int pnx_growth_tables_begin(pnx_ctx *ctx, uint32_t n, uint32_t n_pairs, const uint32_t *branch, const uint32_t *cov_abs,
                                       const double *quorum_rel);
extern "C" int pnx_growth_closed_form_async(pnx_ctx *ctx, const uint64_t *hist, uint32_t n, uint32_t n_pairs, const uint32_t *branch,
                                            const uint32_t *cov_abs, const double *quorum_rel);
extern "C" int pnx_growth_closed_form_fetch(pnx_ctx *ctx, double *out);

namespace pnh {

uint64_t Threshold::to_absolute(uint64_t n) const {
    if (kind == THR_ABSOLUTE) return (uint64_t)value;
    return (uint64_t)std::ceil((double)n * value);
}
double Threshold::to_relative(uint64_t n) const {
    if (kind == THR_RELATIVE) return value;
    return (double)(uint64_t)value / (double)n;
}

double choose_log2(uint64_t n, uint64_t k) {
    if (k > n) return 0.0;
    if (k > n - k) k = n - k;
    double res = 0.0;
    const double nf = (double)n;
    for (uint64_t i = 0; i < k; ++i) {
        res += std::log2(nf - (double)i);
        res -= std::log2((double)i + 1.0);
    }
    return res;
}

namespace {

// log2 of the integers 0..m as f64: the same libm call on the same argument the reference
// makes, hoisted out of the loops (log2(0) = -inf, which the formulas rely on).
struct Log2Table {
    std::vector<double> v, rev;  // rev[t] = v[max - t]: a descending walk over v becomes an ascending one
    explicit Log2Table(uint64_t m) : v(m + 1), rev(m + 1) {
        for (uint64_t i = 0; i <= m; ++i) v[i] = std::log2((double)i);
        for (uint64_t i = 0; i <= m; ++i) rev[i] = v[m - i];
    }
    double operator()(uint64_t i) const { return v[i]; }
};

// One step m of row i of the quorum recurrences for all admissible j at once (hist.rs:167-175
// without the exp2): q[j] is seeded with choose(i, j) where it is still 0.0, advanced by the two
// log2 terms, and x[j] = (q[j] + m_fact) - n_fall is left for the summation.  Element-wise IEEE
// additions in the reference's order -- the compiler may use any vector width, never an FMA.
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("avx2", "default"), optimize("O3")))
#endif
void quorum_step(double *__restrict q, const double *__restrict ch, const double *__restrict la,
                 const double *__restrict lb, double mf, double nf, double *__restrict x, uint64_t len) {
    // (every array begins at the first admissible j: a base "la - something" in front of the table it points into is undefined
    // behaviour even if only in-range elements are read -- UBSan, round 6)
    for (uint64_t j = 0; j < len; ++j) {
        double qj = q[j];
        qj = qj == 0.0 ? ch[j] : qj;
        qj = qj + la[j];
        qj = qj - lb[j];
        q[j] = qj;
        x[j] = (qj + mf) - nf;
    }
}

enum Branch { UNION, CORE, QUORUM };

// ---- device offload of the quorum branch's inner sums ------------------------------------------
// pnx_quorum_sums evaluates the O(n^3) exp2 terms with a restatement of the platform libm's exp2
// (csrc/exp2_exact.hpp).  It is only used after that restatement has reproduced std::exp2 bit for
// bit on a sample of arguments of the kind the closed form produces -- on a libm that computes
// exp2 differently the test fails and everything stays on the host.
const uint64_t EXP2_TAB[256] = {
#include "../csrc/exp2_table.inc"
};

bool exp2_restatement_matches_libm() {
    static const bool ok = []() {
        uint64_t s = 0x2545F4914F6CDD1Dull;
        auto next = [&]() {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            return (double)(s >> 11) * (1.0 / 9007199254740992.0);
        };
        for (int k = 0; k < 400000; ++k) {
            const double u = next();
            double x;
            switch (k & 3) {
                case 0: x = -1100.0 * u; break;   // the whole range of the terms, incl. subnormal results
                case 1: x = -60.0 * u; break;     // where the sums are decided
                case 2: x = -1080.0 + 10.0 * u; break;
                default: x = 40.0 * u - 20.0; break;
            }
            const double a = pnx_exp2::exp2_exact(x, EXP2_TAB), b = std::exp2(x);
            if (std::memcmp(&a, &b, sizeof a) != 0) return false;
        }
        return true;
    }();
    return ok;
}

const uint64_t LOG2_TAB[274] = {
#include "../csrc/log2_table.inc"
};

// the same for log2 (csrc/log2_exact.hpp; its table comes out of the image's libm, tools/gen_log2_table.py): with both
// restatements confirmed the WHOLE closed form can run on the device (pnx_growth_closed_form_*)
bool log2_restatement_matches_libm() {
    static const bool ok = []() {
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (int k = 0; k < 400000; ++k) {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            double x;
            uint64_t u;
            switch (k & 7) {
                case 0: u = s & 0x7FFFFFFFFFFFFFFFull; std::memcpy(&x, &u, 8); break;        // any non-negative bit pattern
                case 1: x = (double)(s >> 40); break;                                          // histogram bins, small integers
                case 2: x = (double)(s >> 11); break;
                case 3: x = 1.0 + ((double)(int64_t)(s >> 11) - 4503599627370496.0) * 0x1p-57; break;  // around 1
                case 4: u = (s & 0x000FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 60 + (s >> 58) * 2) << 52); std::memcpy(&x, &u, 8); break;
                case 5: u = s & 0x000FFFFFFFFFFFFFull; std::memcpy(&x, &u, 8); break;          // subnormal sums
                case 6: x = (double)(k >> 3); break;
                default: u = (s & 0x000FFFFFFFFFFFFFull) | 0x3FE0000000000000ull; std::memcpy(&x, &u, 8); break;
            }
            const double a = pnx_exp2::log2_exact(x, LOG2_TAB), b = std::log2(x);
            if (std::memcmp(&a, &b, sizeof a) != 0 && !(a != a && b != b)) return false;
        }
        return true;
    }();
    return ok;
}

std::mutex g_offload_mu;
pnx_ctx *g_offload_ctx = nullptr;
uint64_t g_offload_min_n = 256;
// a context bound by THIS thread (bind_thread_offload: the in-process CLI, one context per command) goes before the process-wide
// one: two commands on two threads must not evaluate their closed forms on each other's context (a pnx_ctx serves one thread at a time)
thread_local pnx_ctx *t_offload_ctx = nullptr;
// -> the context the calling thread offloads to (nullptr: none) and its threshold
pnx_ctx *offload_context(uint64_t &min_n) {
    if (t_offload_ctx) {
        min_n = 256;
        return t_offload_ctx;
    }
    std::lock_guard<std::mutex> g(g_offload_mu);
    min_n = g_offload_min_n;
    return g_offload_ctx;
}

// The (n+1)^2 term arrays are recycled across calls: fresh 8 MB allocations are mmap'ed by
// malloc and first touched by all workers at once, and those page faults (plus the munmap
// shoot-downs) cost several times the arithmetic at n = 1024.
class ScratchPool {
public:
    struct Buf {
        double *p = nullptr;
        size_t cap = 0;
    };
    static ScratchPool &instance() {
        static ScratchPool pool;
        return pool;
    }
    Buf take(size_t n) {
        {
            std::lock_guard<std::mutex> g(mu_);
            for (size_t k = 0; k < free_.size(); ++k)
                if (free_[k].cap >= n && free_[k].cap <= 2 * n + 1024) {
                    Buf b = free_[k];
                    free_.erase(free_.begin() + (long)k);
                    return b;
                }
        }
        Buf b;
        b.p = new double[n];
        b.cap = n;
        return b;
    }
    void give(Buf b) {
        if (!b.p) return;
        std::lock_guard<std::mutex> g(mu_);
        if (free_.size() >= 16) {  // bounded: drop the smallest
            size_t s = 0;
            for (size_t k = 1; k < free_.size(); ++k)
                if (free_[k].cap < free_[s].cap) s = k;
            if (free_[s].cap < b.cap) std::swap(free_[s], b);
            delete[] b.p;
            return;
        }
        free_.push_back(b);
    }

    ~ScratchPool() {
        for (Buf &b : free_) delete[] b.p;
    }

private:
    std::mutex mu_;
    std::vector<Buf> free_;
};

struct Job {
    Branch branch;
    uint64_t n, c;
    double quorum = 0.0;
    const std::vector<uint64_t> &hist;
    std::shared_ptr<Log2Table> lg;
    std::vector<double> lh, n_fall, m_fact;
    std::vector<uint64_t> m_quorum;
    double tot = 0.0;
    // term1[i][m]: the perc_mult-type term (union / core / quorum's "100 %" part)
    // term2[i][m]: the quorum branch's [m_quorum, 100 %) part; NaN = "no admissible j" (add == false)
    const double *sumq = nullptr;  // from the device (pinned, owned by the context): (n+1) x (n+1), NaN = no admissible j
    std::vector<double> sumq_own;  // ... or a copy of it, when several jobs of a region are offloaded
    ScratchPool::Buf buf1, buf2;
    double *term1 = nullptr, *term2 = nullptr;  // left uninitialised: every entry that is read is written by its row
    std::vector<double> out;

    Job(Branch b, const std::vector<uint64_t> &h, Threshold cov, Threshold quo, std::shared_ptr<Log2Table> tab)
        : branch(b), n(h.size() - 1), hist(h), lg(std::move(tab)) {
        // hist.rs:91, :118, :142 -- core converts the coverage threshold against n + 1
        c = std::max<uint64_t>(1, cov.to_absolute(b == CORE ? n + 1 : n));
        if (b == QUORUM) quorum = quo.to_relative(n);
        lh.resize(n + 1);
        for (uint64_t i = 0; i <= n; ++i) lh[i] = std::log2((double)hist[i]);
        n_fall.assign(n + 1, 0.0);
        m_fact.assign(n + 1, 0.0);
        m_quorum.assign(n + 1, 0);
        const Log2Table &L = *lg;
        for (uint64_t m = 1; m <= n; ++m) {
            n_fall[m] = n_fall[m - 1] + L(n - m + 1);
            if (b == QUORUM) {
                m_fact[m] = m_fact[m - 1] + L(m);
                m_quorum[m] = (uint64_t)std::ceil((double)m * quorum);
            }
        }
        if (b == UNION) {
            uint64_t t = 0;
            for (uint64_t i = c; i <= n; ++i) t += hist[i];
            tot = (double)t;
        }
        buf1 = ScratchPool::instance().take((n + 1) * (n + 1));
        term1 = buf1.p;
        if (b == QUORUM) {
            buf2 = ScratchPool::instance().take((n + 1) * (n + 1));
            term2 = buf2.p;
        }
        out.assign(n, 0.0);
    }
    ~Job() {
        ScratchPool::instance().give(buf1);
        ScratchPool::instance().give(buf2);
    }
    Job(const Job &) = delete;
    Job &operator=(const Job &) = delete;

    uint64_t n_rows() const { return n + 1; }

    // everything the reference computes for histogram index i, for all m
    void run_row(uint64_t i, std::vector<double> &q) const {
        const Log2Table &L = *lg;
        double *t1 = term1 + i * (n + 1);
        double pm = 0.0;
        if (branch == UNION) {
            // hist.rs:102-111: for i in c..n-m+1 { perc_mult[i] += log2(n-m-i+1); y += exp2(..) }
            if (i >= c)
                for (uint64_t m = 1; i + m <= n; ++m) {
                    pm += L(n - m - i + 1);
                    t1[m] = std::exp2(lh[i] + pm - n_fall[m]);
                }
            return;
        }
        // hist.rs:127-135 / :157-160: for i in max(m,c)..n+1 { perc_mult[i] += log2(i-m+1); .. }
        if (i >= c)
            for (uint64_t m = 1; m <= std::min(i, n); ++m) {
                pm += L(i - m + 1);
                t1[m] = std::exp2(lh[i] + pm - n_fall[m]);
            }
        if (branch != QUORUM || i >= n) return;
        // hist.rs:163-183, row i of Q
        double *t2 = term2 + i * (n + 1);
        const double nan = std::nan("");
        if (sumq) {  // inner sums came from the device; hist.rs:178-180 stays here
            for (uint64_t m = 1; m <= n; ++m) {
                const double sq = sumq[i * (n + 1) + m];
                t2[m] = sq == sq ? std::exp2(lh[i] + std::log2(sq)) : nan;
            }
            return;
        }
        // q | choose(i, .) | x of the current step
        q.assign(3 * (n + 1), 0.0);
        double *qq = q.data(), *ch = qq + (n + 1), *xs = ch + (n + 1);
        const uint64_t K = L.v.size() - 1;
        uint64_t ch_hi = 0;  // choose(i, j) is known for the js that were admissible so far
        for (uint64_t m = 1; m <= n; ++m) {
            t2[m] = nan;
            if (i < m_quorum[m]) continue;  // the reference loops "for i in m_quorum..n"
            // for j in max(m_quorum, c)..m { if n + j + 1 > i + m && j <= i {..} }  (hist.rs:164-166)
            uint64_t jlo = std::max(m_quorum[m], c);
            if (i + m > n + jlo) jlo = i + m - n;
            const uint64_t jhi = std::min(m, i + 1);
            if (jlo >= jhi) continue;  // add stays false
            for (uint64_t j = std::max(ch_hi, jlo); j < jhi; ++j) {  // choose(i, j), hist.rs:21-36
                const uint64_t k = j > i - j ? i - j : j;
                double res = 0.0;
                for (uint64_t a = 0; a < k; ++a) {
                    res += L(i - a);
                    res -= L(a + 1);
                }
                ch[j] = res;
            }
            ch_hi = std::max(ch_hi, jhi);
            // q[j] += log2(n - i - m + 1 + j); q[j] -= log2(m - j); x = q[j] + m_fact - n_fall
            quorum_step(qq + jlo, ch + jlo, L.v.data() + ((n + 1 + jlo) - (i + m)), L.rev.data() + (K - m) + jlo, m_fact[m], n_fall[m], xs + jlo,
                        jhi - jlo);
            double sum_q = 0.0;
            for (uint64_t j = jlo; j < jhi; ++j) {
                // sum_q += exp2(x).  The libm call is skipped where its result provably cannot change
                // the running sum: exp2(x) <= 2^(floor(x)+1) (+1 ulp), so for x + 56 <= exponent(sum_q)
                // the addend is below half an ulp of a normal sum_q and the rounded sum is sum_q itself;
                // far below the subnormals it is +0.
                const double x = xs[j];
                if (x < -1100.0) continue;
                uint64_t sb;
                std::memcpy(&sb, &sum_q, sizeof sb);
                const int64_t es = (int64_t)((sb >> 52) & 0x7FF);  // biased exponent, 0 = zero / subnormal
                if (es > 0 && x + 56.0 <= (double)(es - 1023)) continue;
                sum_q += std::exp2(x);
            }
            t2[m] = std::exp2(lh[i] + std::log2(sum_q));
        }
    }

    // the sum over i for one m, in ascending i like the reference (different m are independent)
    void finish_m(uint64_t m) {
        double y = 0.0;
        if (branch == UNION) {
            for (uint64_t i = c; i + m <= n; ++i) y += term1[i * (n + 1) + m];
            out[m - 1] = tot - y;
            return;
        }
        for (uint64_t i = std::max(m, c); i <= n; ++i) y += term1[i * (n + 1) + m];
        if (branch == CORE) {
            out[m - 1] = y;
            return;
        }
        double yr = 0.0;
        for (uint64_t i = m_quorum[m]; i < n; ++i) {
            const double t = term2[i * (n + 1) + m];
            if (t == t) yr += t;  // not NaN <=> add == true
        }
        out[m - 1] = y + yr;
    }
};

Branch dispatch(uint64_t n, Threshold quorum) {  // Hist::calc_growth, hist.rs:51-66
    const uint64_t q_abs = std::max<uint64_t>(1, quorum.to_absolute(n));
    if (q_abs == 1) return UNION;
    if (q_abs >= n) return CORE;
    return QUORUM;
}

// Quorum jobs that are large enough: enqueue the inner sums on the GPU.  The context keeps one
// result buffer, so one job per region is offloaded; the others (and any failure) take the host path.
struct Offload {
    pnx_ctx *ctx = nullptr;
    Job *job = nullptr;
};

bool offload_eligible(const Job &j, uint64_t min_n) {
    return j.branch == QUORUM && j.n >= min_n && j.n <= 8192 && exp2_restatement_matches_libm();
}

Offload start_offload(std::vector<std::unique_ptr<Job>> &jobs) {
    Offload o;
    uint64_t min_n = 0;
    o.ctx = offload_context(min_n);
    if (!o.ctx) return o;
    for (auto &j : jobs) {
        if (!offload_eligible(*j, min_n)) continue;
        std::vector<uint32_t> mq(j->n + 1);
        for (uint64_t m = 0; m <= j->n; ++m) mq[m] = (uint32_t)j->m_quorum[m];
        if (pnx_quorum_sums_async(o.ctx, (uint32_t)j->n, (uint32_t)j->c, mq.data(), j->lg->v.data(), j->m_fact.data(),
                                  j->n_fall.data()) == 0)
            o.job = j.get();
        break;
    }
    return o;
}


// waits for the enqueued job; further eligible jobs of the region follow one at a time (the
// context has one result buffer, so results are copied out when there is more than one)
void finish_offload(const Offload &o, std::vector<std::unique_ptr<Job>> &jobs) {
    if (!o.job) return;
    uint64_t min_n = 0;
    (void)offload_context(min_n);
    std::vector<Job *> more;
    for (auto &j : jobs)
        if (j.get() != o.job && offload_eligible(*j, min_n)) more.push_back(j.get());
    const double *res = nullptr;
    if (pnx_quorum_sums_fetch(o.ctx, &res) != 0) return;  // any device error: the host path
    auto keep = [&](Job *j) {
        if (more.empty()) {
            j->sumq = res;
        } else {
            j->sumq_own.assign(res, res + (j->n + 1) * (j->n + 1));
            j->sumq = j->sumq_own.data();
        }
    };
    keep(o.job);
    for (Job *j : more) {
        std::vector<uint32_t> mq(j->n + 1);
        for (uint64_t m = 0; m <= j->n; ++m) mq[m] = (uint32_t)j->m_quorum[m];
        if (pnx_quorum_sums_async(o.ctx, (uint32_t)j->n, (uint32_t)j->c, mq.data(), j->lg->v.data(), j->m_fact.data(),
                                  j->n_fall.data()) != 0 ||
            pnx_quorum_sums_fetch(o.ctx, &res) != 0)
            continue;
        keep(j);
    }
}

std::vector<std::vector<double>> run_jobs(std::vector<std::unique_ptr<Job>> &jobs, unsigned n_threads) {
    static const bool timing = std::getenv("PANACUS_AMD_HOST_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_begin = now();
    // flatten (job, row-chunk) into one task list; quorum rows cost ~i^2, so they are cut finer
    struct Task {
        Job *job;
        uint64_t lo, hi;
    };
    std::vector<Task> tasks;
    for (auto &j : jobs) {
        const uint64_t rows = j->n_rows();
        const uint64_t chunks = std::min<uint64_t>(rows, j->branch == QUORUM ? 192 : 32);
        for (uint64_t k = 0; k < chunks; ++k) {
            uint64_t lo = rows * k / chunks, hi = rows * (k + 1) / chunks;
            if (hi > lo) tasks.push_back(Task{j.get(), lo, hi});
        }
    }
    // heavy tasks first (quorum rows with large i), so the tail of the region is short
    std::stable_sort(tasks.begin(), tasks.end(), [](const Task &a, const Task &b) {
        auto w = [](const Task &t) { return t.job->branch == QUORUM ? (double)t.hi * (double)t.hi : 1.0; };
        return w(a) > w(b);
    });
    auto body = [&](size_t k) {
        thread_local std::vector<double> q;
        const Task &t = tasks[k];
        for (uint64_t i = t.lo; i < t.hi; ++i) t.job->run_row(i, q);
    };
    uint64_t work = 0;
    for (auto &j : jobs) work += j->n * j->n * (j->branch == QUORUM ? j->n / 8 + 1 : 1);
    if (work < 20000 || n_threads == 1) {
        for (size_t k = 0; k < tasks.size(); ++k) body(k);
    } else {
        ThreadPool::instance().parallel_for(tasks.size(), body, n_threads);
    }
    const auto t_rows = now();
    // epilogue: one task per (job, block of m)
    struct Fin {
        Job *job;
        uint64_t lo, hi;
    };
    std::vector<Fin> fins;
    for (auto &j : jobs)
        for (uint64_t lo = 1; lo <= j->n; lo += 16) fins.push_back(Fin{j.get(), lo, std::min(j->n, lo + 15)});
    auto fin_body = [&](size_t k) {
        for (uint64_t m = fins[k].lo; m <= fins[k].hi; ++m) fins[k].job->finish_m(m);
    };
    if (work < 20000 || n_threads == 1) {
        for (size_t k = 0; k < fins.size(); ++k) fin_body(k);
    } else {
        ThreadPool::instance().parallel_for(fins.size(), fin_body, n_threads);
    }
    if (timing)
        std::fprintf(stderr, "[host growth] rows %.3f ms (%zu tasks), finish %.3f ms (%zu tasks)\n",
                     ms(t_begin, t_rows), tasks.size(), ms(t_rows, now()), fins.size());
    std::vector<std::vector<double>> out;
    for (auto &j : jobs) out.push_back(std::move(j->out));
    return out;
}

}  // namespace

void set_quorum_offload(void *pnx_context, uint64_t min_n) {
    std::lock_guard<std::mutex> g(g_offload_mu);
    g_offload_ctx = static_cast<pnx_ctx *>(pnx_context);
    g_offload_min_n = min_n;
}

void release_quorum_offload(void *pnx_context) {
    std::lock_guard<std::mutex> g(g_offload_mu);
    if (g_offload_ctx == static_cast<pnx_ctx *>(pnx_context)) g_offload_ctx = nullptr;
}

void bind_thread_offload(void *pnx_context) { t_offload_ctx = static_cast<pnx_ctx *>(pnx_context); }
void unbind_thread_offload(void *pnx_context) {
    if (t_offload_ctx == static_cast<pnx_ctx *>(pnx_context)) t_offload_ctx = nullptr;
}

bool quorum_offload_usable() { return exp2_restatement_matches_libm(); }
bool device_growth_usable() { return exp2_restatement_matches_libm() && log2_restatement_matches_libm(); }
void log2_restated(const double *x, double *y, uint64_t n) {
    for (uint64_t k = 0; k < n; ++k) y[k] = pnx_exp2::log2_exact(x[k], LOG2_TAB);
}

struct GrowthRun {
    std::vector<uint64_t> hist;  // the jobs refer to it
    std::vector<Threshold> coverage, quorum;
    std::vector<std::unique_ptr<Job>> jobs;
    size_t n_pairs = 0;
    uint64_t n = 0;
    unsigned n_threads = 0;
    Offload off;
    pnx_ctx *dev = nullptr;  // the whole closed form was enqueued on this context (pnx_growth_closed_form_async)
};

namespace {
void host_jobs(GrowthRun &run) {
    auto tab = std::make_shared<Log2Table>(2 * run.n + 2);
    for (size_t t = 0; t < run.n_pairs; ++t)
        run.jobs.emplace_back(new Job(dispatch(run.n, run.quorum[t]), run.hist, run.coverage[t], run.quorum[t], tab));
}

// Whole closed forms on the device: with a context set (set_quorum_offload), both libm restatements confirmed, and n
// in the range the device path takes.  hist == nullptr: the counters of the context's last coverage pass.
bool start_device_growth(GrowthRun &run, const uint64_t *hist, const void *only_ctx = nullptr) {
    uint64_t min_n = 0;
    pnx_ctx *ctx = offload_context(min_n);
    if (only_ctx && ctx != only_ctx) return false;  // "the counters of the last pass" are those of the offload context and of no other
    const uint64_t n = run.n;
    if (!ctx || n < min_n || n < 2 || n > 2048 || run.n_pairs == 0 || run.n_pairs > 16 || !device_growth_usable()) return false;
    std::vector<uint32_t> br(run.n_pairs), cv(run.n_pairs);
    std::vector<double> qr(run.n_pairs, 0.0);
    for (size_t t = 0; t < run.n_pairs; ++t) {
        const Branch b = dispatch(n, run.quorum[t]);  // Hist::calc_growth, hist.rs:51-66
        br[t] = b == UNION ? 0u : (b == CORE ? 1u : 2u);
        cv[t] = (uint32_t)std::max<uint64_t>(1, run.coverage[t].to_absolute(b == CORE ? n + 1 : n));  // hist.rs:91, :118, :142
        if (b == QUORUM) qr[t] = run.quorum[t].to_relative(n);
    }
    if (pnx_growth_closed_form_async(ctx, hist, (uint32_t)n, (uint32_t)run.n_pairs, br.data(), cv.data(), qr.data()) != 0) return false;
    run.dev = ctx;
    return true;
}
}  // namespace

// The thresholds are known before the coverage pass is enqueued: the first part of the device tables of (n, pairs) -- two small
// kernels -- is started now and runs while the pass's kernels are being launched (pnx_growth_tables_begin); false: the device path
// would not take these arguments anyway.
bool growth_tables_begin(uint64_t n, const std::vector<Threshold> &coverage, const std::vector<Threshold> &quorum, const void *only_ctx) {
    uint64_t min_n = 0;
    pnx_ctx *ctx = offload_context(min_n);
    if (only_ctx && ctx != only_ctx) return false;
    const size_t n_pairs = coverage.size();
    if (!ctx || n < min_n || n < 2 || n > 2048 || n_pairs == 0 || n_pairs > 16 || quorum.size() != n_pairs || !device_growth_usable()) return false;
    std::vector<uint32_t> br(n_pairs), cv(n_pairs);
    std::vector<double> qr(n_pairs, 0.0);
    for (size_t t = 0; t < n_pairs; ++t) {
        const Branch b = dispatch(n, quorum[t]);  // Hist::calc_growth, hist.rs:51-66
        br[t] = b == UNION ? 0u : (b == CORE ? 1u : 2u);
        cv[t] = (uint32_t)std::max<uint64_t>(1, coverage[t].to_absolute(b == CORE ? n + 1 : n));  // hist.rs:91, :118, :142
        if (b == QUORUM) qr[t] = quorum[t].to_relative(n);
    }
    return pnx_growth_tables_begin(ctx, (uint32_t)n, (uint32_t)n_pairs, br.data(), cv.data(), qr.data()) == 0;
}

GrowthRun *calc_all_growths_begin(const std::vector<uint64_t> &hist, const std::vector<Threshold> &coverage,
                                  const std::vector<Threshold> &quorum, unsigned n_threads) {
    std::unique_ptr<GrowthRun> run(new GrowthRun);
    run->hist = hist;
    run->coverage = coverage;
    run->quorum = quorum;
    run->n_pairs = coverage.size();
    run->n_threads = n_threads;
    if (hist.size() >= 2) {
        run->n = hist.size() - 1;
        if (!start_device_growth(*run, run->hist.data())) {
            host_jobs(*run);
            run->off = start_offload(run->jobs);
        }
    }
    return run.release();
}

// the curves of the histogram of the coverage pass enqueued LAST on the offload context (n groups), without the histogram
// visiting the host; nullptr when the device path cannot take it (the caller then fetches the histogram and uses _begin)
GrowthRun *calc_all_growths_begin_on_device(uint64_t n, const std::vector<Threshold> &coverage, const std::vector<Threshold> &quorum,
                                            const void *only_ctx) {
    std::unique_ptr<GrowthRun> run(new GrowthRun);
    run->coverage = coverage;
    run->quorum = quorum;
    run->n_pairs = coverage.size();
    run->n = n;
    if (!start_device_growth(*run, nullptr, only_ctx)) return nullptr;
    return run.release();
}

std::vector<std::vector<double>> calc_all_growths_end(GrowthRun *handle) {
    std::unique_ptr<GrowthRun> run(handle);
    if (!run) return {};
    if (run->n < 1) return std::vector<std::vector<double>>(run->n_pairs);
    if (run->dev) {
        std::vector<double> flat(run->n_pairs * run->n);
        if (pnx_growth_closed_form_fetch(run->dev, flat.data()) == 0) {
            std::vector<std::vector<double>> out(run->n_pairs);
            for (size_t t = 0; t < run->n_pairs; ++t) out[t].assign(flat.begin() + t * run->n, flat.begin() + (t + 1) * run->n);
            return out;
        }
        if (run->hist.size() < 2) return std::vector<std::vector<double>>(run->n_pairs);  // no histogram on the host to fall back on
        host_jobs(*run);  // a device error: the host path
    }
    finish_offload(run->off, run->jobs);
    return run_jobs(run->jobs, run->n_threads);
}

std::vector<std::vector<double>> calc_all_growths(const std::vector<uint64_t> &hist,
                                                  const std::vector<Threshold> &coverage,
                                                  const std::vector<Threshold> &quorum, unsigned n_threads) {
    return calc_all_growths_end(calc_all_growths_begin(hist, coverage, quorum, n_threads));
}

static std::vector<double> one(Branch b, const std::vector<uint64_t> &hist, Threshold c, Threshold q, unsigned n_threads) {
    if (hist.size() < 2) return {};
    auto tab = std::make_shared<Log2Table>(2 * (hist.size() - 1) + 2);
    std::vector<std::unique_ptr<Job>> jobs;
    jobs.emplace_back(new Job(b, hist, c, q, tab));
    finish_offload(start_offload(jobs), jobs);
    return run_jobs(jobs, n_threads)[0];
}

std::vector<double> calc_growth_union(const std::vector<uint64_t> &hist, Threshold coverage, unsigned n_threads) {
    return one(UNION, hist, coverage, Threshold{THR_RELATIVE, 0.0}, n_threads);
}
std::vector<double> calc_growth_core(const std::vector<uint64_t> &hist, Threshold coverage, unsigned n_threads) {
    return one(CORE, hist, coverage, Threshold{THR_RELATIVE, 1.0}, n_threads);
}
std::vector<double> calc_growth_quorum(const std::vector<uint64_t> &hist, Threshold coverage, Threshold quorum,
                                       unsigned n_threads) {
    return one(QUORUM, hist, coverage, quorum, n_threads);
}
std::vector<double> calc_growth(const std::vector<uint64_t> &hist, Threshold coverage, Threshold quorum,
                                unsigned n_threads) {
    if (hist.size() < 2) return {};
    return one(dispatch(hist.size() - 1, quorum), hist, coverage, quorum, n_threads);
}

}  // namespace pnh
