// growth_closed_form.cpp -- see growth_closed_form.hpp.
#include "growth_closed_form.hpp"

#include <algorithm>
#include <cmath>
#include <thread>

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

std::vector<double> calc_growth_union(const std::vector<uint64_t> &hist, Threshold coverage) {
    const uint64_t n = hist.size() - 1;
    const uint64_t c = std::max<uint64_t>(1, coverage.to_absolute(n));
    std::vector<double> out(n, 0.0), perc_mult(n + 1, 0.0), lh(n + 1);
    Log2Table lg(n + 1);
    for (uint64_t i = 0; i <= n; ++i) lh[i] = std::log2((double)hist[i]);
    uint64_t tot_i = 0;
    for (uint64_t i = c; i <= n; ++i) tot_i += hist[i];
    const double tot = (double)tot_i;
    double n_fall_m = 0.0;
    for (uint64_t m = 1; m <= n; ++m) {
        double y = 0.0;
        n_fall_m += lg(n - m + 1);
        for (uint64_t i = c; i + m <= n; ++i) {
            perc_mult[i] += lg(n - m - i + 1);
            y += std::exp2(lh[i] + perc_mult[i] - n_fall_m);
        }
        out[m - 1] = tot - y;
    }
    return out;
}

std::vector<double> calc_growth_core(const std::vector<uint64_t> &hist, Threshold coverage) {
    const uint64_t n = hist.size() - 1;
    const uint64_t c = std::max<uint64_t>(1, coverage.to_absolute(n + 1));
    std::vector<double> out(n, 0.0), perc_mult(n + 1, 0.0), lh(n + 1);
    Log2Table lg(n + 1);
    for (uint64_t i = 0; i <= n; ++i) lh[i] = std::log2((double)hist[i]);
    double n_fall_m = 0.0;
    for (uint64_t m = 1; m <= n; ++m) {
        double y = 0.0;
        n_fall_m += lg(n - m + 1);
        for (uint64_t i = std::max(m, c); i <= n; ++i) {
            perc_mult[i] += lg(i - m + 1);
            y += std::exp2(lh[i] + perc_mult[i] - n_fall_m);
        }
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

    // yl[m]: the "100 % quorum" part -- sequential in m through perc_mult, cheap (O(n^2))
    std::vector<double> yl(n + 1, 0.0);
    {
        std::vector<double> perc_mult(n + 1, 0.0);
        for (uint64_t m = 1; m <= n; ++m) {
            double y = 0.0;
            for (uint64_t i = std::max(m, c); i <= n; ++i) {
                perc_mult[i] += lg(i - m + 1);
                y += std::exp2(lh[i] + perc_mult[i] - n_fall[m]);
            }
            yl[m] = y;
        }
    }

    // term[m][i] = exp2(log2 h[i] + log2 sum_q(i, m)) or "absent": rows i are independent
    // (Q[i][*] only ever touches row i), so they go to threads; the sum over i is then done
    // serially in ascending i like the reference.
    std::vector<double> term((n + 1) * (n + 1), 0.0);
    std::vector<uint8_t> has((n + 1) * (n + 1), 0);
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
                        if (q[j] == 0.0) q[j] = choose_log2(i, j);
                        q[j] += lg(n - i - m + 1 + j);
                        q[j] -= lg(m - j);
                        sum_q += std::exp2(q[j] + m_fact[m] - n_fall[m]);
                        add = true;
                    }
                }
                if (add) {
                    term[m * (n + 1) + i] = std::exp2(lh[i] + std::log2(sum_q));
                    has[m * (n + 1) + i] = 1;
                }
            }
        }
    };
    unsigned nt = n_threads ? n_threads : std::max(1u, std::thread::hardware_concurrency());
    if (n < 64) nt = 1;
    nt = (unsigned)std::min<uint64_t>(nt, n);
    if (nt <= 1) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        // interleaved blocks: the cost of row i is roughly proportional to i
        const uint64_t blocks = (uint64_t)nt * 4;
        std::vector<std::pair<uint64_t, uint64_t>> ranges;
        for (uint64_t b = 0; b < blocks; ++b) {
            uint64_t lo = n * b / blocks, hi = n * (b + 1) / blocks;
            if (hi > lo) ranges.emplace_back(lo, hi);
        }
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t]() {
                for (size_t k = t; k < ranges.size(); k += nt) work(ranges[k].first, ranges[k].second);
            });
        for (auto &x : th) x.join();
    }
    for (uint64_t m = 1; m <= n; ++m) {
        double yr = 0.0;
        for (uint64_t i = m_quorum[m]; i < n; ++i)
            if (has[m * (n + 1) + i]) yr += term[m * (n + 1) + i];
        out[m - 1] = yl[m] + yr;
    }
    return out;
}

std::vector<double> calc_growth(const std::vector<uint64_t> &hist, Threshold coverage, Threshold quorum,
                                unsigned n_threads) {
    if (hist.size() < 2) return {};
    const uint64_t n = hist.size() - 1;
    const uint64_t q_abs = std::max<uint64_t>(1, quorum.to_absolute(n));
    if (q_abs == 1) return calc_growth_union(hist, coverage);
    if (q_abs >= n) return calc_growth_core(hist, coverage);
    return calc_growth_quorum(hist, coverage, quorum, n_threads);
}

}  // namespace pnh
