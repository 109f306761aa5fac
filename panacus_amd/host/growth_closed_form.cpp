// growth_closed_form.cpp -- see growth_closed_form.hpp.
#include "growth_closed_form.hpp"

#include <algorithm>
#include <cmath>
#include <thread>

#include "thread_pool.hpp"

namespace pnh {

uint64_t Threshold::to_absolute(uint64_t n) const {
    if (kind == THR_ABSOLUTE) return (uint64_t)value;
    return (uint64_t)std::ceil((double)n * value);
}
double Threshold::to_relative(uint64_t n) const {
    if (kind == THR_RELATIVE) return value;
    return (double)(uint64_t)value / (double)n;
}

namespace {
// log2 of the integers 0..m as f64: the same libm call on the same argument the reference
// makes, hoisted out of the loops (log2(0) = -inf, which the formulas rely on).
struct Log2Table {
    std::vector<double> v;
    explicit Log2Table(uint64_t m) : v(m + 1) {
        for (uint64_t i = 0; i <= m; ++i) v[i] = std::log2((double)i);
    }
    double operator()(uint64_t i) const { return v[i]; }
};
}  // namespace

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

// Both O(n^2) branches share one shape: per histogram index i a running perc_mult[i]
// (sequential in m), per m a sum over i in ascending order.  Rows i are independent, so
// they are cut into chunks for the pool; each chunk fills term[m][i], and the sums over i
// are then taken serially in the reference's order -- bit-identical to the serial loops.
namespace {
template <class RowRange, class Incr>
void two_level(uint64_t n, const std::vector<double> &lh, const std::vector<double> &n_fall, RowRange row_active,
               Incr incr, unsigned n_threads, std::vector<double> &term) {
    // term is (n+1) x (n+1), stored [i][m]: a thread owns whole rows, so no cache line is shared
    auto work = [&](uint64_t i_lo, uint64_t i_hi) {
        for (uint64_t i = i_lo; i < i_hi; ++i) {
            double pm = 0.0;
            for (uint64_t m = 1; m <= n; ++m) {
                if (!row_active(i, m)) continue;
                pm += incr(i, m);
                term[i * (n + 1) + m] = std::exp2(lh[i] + pm - n_fall[m]);
            }
        }
    };
    const uint64_t rows = n + 1;
    if (n < 96) {
        work(0, rows);
        return;
    }
    const size_t chunks = 64;
    ThreadPool::instance().parallel_for(
        chunks, [&](size_t k) { work(rows * k / chunks, rows * (k + 1) / chunks); }, n_threads);
}
}  // namespace

std::vector<double> calc_growth_union(const std::vector<uint64_t> &hist, Threshold coverage, unsigned n_threads) {
    const uint64_t n = hist.size() - 1;
    const uint64_t c = std::max<uint64_t>(1, coverage.to_absolute(n));
    std::vector<double> out(n, 0.0), lh(n + 1), n_fall(n + 1, 0.0);
    Log2Table lg(n + 1);
    for (uint64_t i = 0; i <= n; ++i) lh[i] = std::log2((double)hist[i]);
    uint64_t tot_i = 0;
    for (uint64_t i = c; i <= n; ++i) tot_i += hist[i];
    const double tot = (double)tot_i;
    for (uint64_t m = 1; m <= n; ++m) n_fall[m] = n_fall[m - 1] + lg(n - m + 1);
    std::vector<double> term((n + 1) * (n + 1), 0.0);
    // hist.rs:102-111: for i in c..n-m+1 { perc_mult[i] += log2(n-m-i+1); y += exp2(..) }
    two_level(
        n, lh, n_fall, [&](uint64_t i, uint64_t m) { return i >= c && i + m <= n; },
        [&](uint64_t i, uint64_t m) { return lg(n - m - i + 1); }, n_threads, term);
    for (uint64_t m = 1; m <= n; ++m) {
        double y = 0.0;
        for (uint64_t i = c; i + m <= n; ++i) y += term[i * (n + 1) + m];
        out[m - 1] = tot - y;
    }
    return out;
}

std::vector<double> calc_growth_core(const std::vector<uint64_t> &hist, Threshold coverage, unsigned n_threads) {
    const uint64_t n = hist.size() - 1;
    const uint64_t c = std::max<uint64_t>(1, coverage.to_absolute(n + 1));
    std::vector<double> out(n, 0.0), lh(n + 1), n_fall(n + 1, 0.0);
    Log2Table lg(n + 1);
    for (uint64_t i = 0; i <= n; ++i) lh[i] = std::log2((double)hist[i]);
    for (uint64_t m = 1; m <= n; ++m) n_fall[m] = n_fall[m - 1] + lg(n - m + 1);
    std::vector<double> term((n + 1) * (n + 1), 0.0);
    // hist.rs:127-135: for i in max(m,c)..n+1 { perc_mult[i] += log2(i-m+1); y += exp2(..) }
    two_level(
        n, lh, n_fall, [&](uint64_t i, uint64_t m) { return i >= std::max(m, c); },
        [&](uint64_t i, uint64_t m) { return lg(i - m + 1); }, n_threads, term);
    for (uint64_t m = 1; m <= n; ++m) {
        double y = 0.0;
        for (uint64_t i = std::max(m, c); i <= n; ++i) y += term[i * (n + 1) + m];
        out[m - 1] = y;
    }
    return out;
}

std::vector<double> calc_growth_quorum(const std::vector<uint64_t> &hist, Threshold coverage, Threshold quorum_t,
                                       unsigned n_threads) {
    const uint64_t n = hist.size() - 1;
    const uint64_t c = std::max<uint64_t>(1, coverage.to_absolute(n));
    const double quorum = quorum_t.to_relative(n);
    std::vector<double> out(n, 0.0), lh(n + 1);
    Log2Table lg(2 * n + 2);
    for (uint64_t i = 0; i <= n; ++i) lh[i] = std::log2((double)hist[i]);

    // scalars of the m-recurrence, one value per m (prefix sums in the reference's order)
    std::vector<double> n_fall(n + 1, 0.0), m_fact(n + 1, 0.0);
    std::vector<uint64_t> m_quorum(n + 1, 0);
    for (uint64_t m = 1; m <= n; ++m) {
        m_fact[m] = m_fact[m - 1] + lg(m);
        n_fall[m] = n_fall[m - 1] + lg(n - m + 1);
        m_quorum[m] = (uint64_t)std::ceil((double)m * quorum);
    }

    // yl[m]: the "100 % quorum" part (hist.rs:157-160), same shape as the core branch
    std::vector<double> yl(n + 1, 0.0);
    {
        std::vector<double> t2((n + 1) * (n + 1), 0.0);
        two_level(
            n, lh, n_fall, [&](uint64_t i, uint64_t m) { return i >= std::max(m, c); },
            [&](uint64_t i, uint64_t m) { return lg(i - m + 1); }, n_threads, t2);
        for (uint64_t m = 1; m <= n; ++m) {
            double y = 0.0;
            for (uint64_t i = std::max(m, c); i <= n; ++i) y += t2[i * (n + 1) + m];
            yl[m] = y;
        }
    }

    // term[m][i] = exp2(log2 h[i] + log2 sum_q(i, m)) or "absent": rows i are independent
    // (Q[i][*] only ever touches row i), so they go to threads; the sum over i is then done
    // serially in ascending i like the reference.
    std::vector<double> term((n + 1) * (n + 1), 0.0);
    std::vector<uint8_t> has((n + 1) * (n + 1), 0);
    // choose(i, j) of hist.rs:21-36 with its log2 calls served from the table (same values)
    auto choose_tab = [&](uint64_t nn, uint64_t k) -> double {
        if (k > nn) return 0.0;
        if (k > nn - k) k = nn - k;
        double res = 0.0;
        for (uint64_t a = 0; a < k; ++a) {
            res += lg(nn - a);
            res -= lg(a + 1);
        }
        return res;
    };
    auto work = [&](uint64_t i_lo, uint64_t i_hi) {
        std::vector<double> q(n + 1);
        for (uint64_t i = i_lo; i < i_hi; ++i) {
            std::fill(q.begin(), q.end(), 0.0);
            for (uint64_t m = 1; m <= n; ++m) {
                if (i < m_quorum[m]) continue;  // loop is "for i in m_quorum..n"
                double sum_q = 0.0;
                bool add = false;
                for (uint64_t j = std::max(m_quorum[m], c); j < m; ++j) {
                    if (n + j + 1 > i + m && j <= i) {
                        if (q[j] == 0.0) q[j] = choose_tab(i, j);
                        q[j] += lg(n - i - m + 1 + j);
                        q[j] -= lg(m - j);
                        sum_q += std::exp2(q[j] + m_fact[m] - n_fall[m]);
                        add = true;
                    }
                }
                if (add) {
                    term[i * (n + 1) + m] = std::exp2(lh[i] + std::log2(sum_q));
                    has[i * (n + 1) + m] = 1;
                }
            }
        }
    };
    if (n < 48) {
        work(0, n);
    } else {
        // rows get more expensive with i (more admissible j), so cut finely and let the pool balance
        const size_t chunks = std::min<uint64_t>(n, 256);
        ThreadPool::instance().parallel_for(
            chunks, [&](size_t k) { work(n * k / chunks, n * (k + 1) / chunks); }, n_threads);
    }
    for (uint64_t m = 1; m <= n; ++m) {
        double yr = 0.0;
        for (uint64_t i = m_quorum[m]; i < n; ++i)
            if (has[i * (n + 1) + m]) yr += term[i * (n + 1) + m];
        out[m - 1] = yl[m] + yr;
    }
    return out;
}

std::vector<double> calc_growth(const std::vector<uint64_t> &hist, Threshold coverage, Threshold quorum,
                                unsigned n_threads) {
    if (hist.size() < 2) return {};
    const uint64_t n = hist.size() - 1;
    const uint64_t q_abs = std::max<uint64_t>(1, quorum.to_absolute(n));
    if (q_abs == 1) return calc_growth_union(hist, coverage, n_threads);
    if (q_abs >= n) return calc_growth_core(hist, coverage, n_threads);
    return calc_growth_quorum(hist, coverage, quorum, n_threads);
}

}  // namespace pnh
