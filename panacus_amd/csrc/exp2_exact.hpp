// exp2_exact.hpp -- the exp2 of the platform libm, restated so that it can run on the GPU.
//
// The closed-form growth (Hist::calc_growth_quorum, src/graph_broker/hist.rs:138-187) spends
// O(n^3) calls of f64::exp2, which Rust forwards to the platform libm.  glibc >= 2.28 computes
// exp2 with the table-driven algorithm of S. Nagy (ARM optimized routines): x = k/128 + r,
// 2^(k/128) from a 128-entry table split into H (1 + T), 2^r - 1 from a degree-5 polynomial,
// result = scale + scale * tmp.  Every step is plain IEEE double arithmetic, so the same
// sequence of operations gives the same bits on any IEEE machine -- PROVIDED nothing is
// contracted into FMAs: the x86-64 libm of this platform evaluates it without FMA (checked
// against 4*10^7 arguments here and on the MI355X hosts, and re-checked at run time before the
// device path is used: quorum_offload_selftest in growth_closed_form.cpp).
// The coefficients are the published ones of that algorithm; the table is regenerated from
// first principles by tools/gen_exp2_table.py.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define PNX_HD __host__ __device__
#else
#define PNX_HD
#endif

namespace pnx_exp2 {

#if defined(__HIP_DEVICE_COMPILE__)
// round-to-nearest primitives that the compiler never fuses
PNX_HD static inline double mul(double a, double b) { return __dmul_rn(a, b); }
PNX_HD static inline double add(double a, double b) { return __dadd_rn(a, b); }
#else
// host: the translation units that include this header are built with -ffp-contract=off
PNX_HD static inline double mul(double a, double b) { return a * b; }
PNX_HD static inline double add(double a, double b) { return a + b; }
#endif
PNX_HD static inline double sub(double a, double b) { return add(a, -b); }

PNX_HD static inline uint64_t as_u64(double x) {
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return u;
}
PNX_HD static inline double as_f64(uint64_t u) {
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
}

// `tab` = the 256 words of exp2_table.inc
PNX_HD static inline double exp2_exact(double x, const uint64_t *tab) {
    constexpr double C1 = 0x1.62e42fefa39efp-1, C2 = 0x1.ebfbdff82c424p-3, C3 = 0x1.c6b08d70cf4b5p-5,
                     C4 = 0x1.3b2abd24650ccp-7, C5 = 0x1.5d7e09b4e3a84p-10;
    constexpr double SHIFT = 0x1.8p45;  // 0x1.8p52 / 128
    uint32_t abstop = (uint32_t)(as_u64(x) >> 52) & 0x7ffu;
    bool special = false;
    if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {        // |x| < 2^-54 or |x| >= 512 or not finite
        if (abstop - 0x3c9u >= 0x80000000u) return add(1.0, x);
        if (abstop >= 0x409u) {                        // |x| >= 1024
            if (as_u64(x) == 0xfff0000000000000ull) return 0.0;
            if (abstop >= 0x7ffu) return add(1.0, x);  // NaN, +inf
            if (!(as_u64(x) >> 63)) return as_f64(0x7ff0000000000000ull);  // overflow
            if (as_u64(x) >= 0xc090cc0000000000ull) return 0.0;            // x <= -1075: underflow to +0
        }
        if (2 * as_u64(x) > 2 * 0x408d000000000000ull) special = true;     // |x| > 928
    }
    double kd = add(x, SHIFT);
    const uint64_t ki = as_u64(kd);
    kd = sub(kd, SHIFT);
    const double r = sub(x, kd);
    const uint64_t idx = 2 * (ki % 128);
    const uint64_t top = ki << 45;
    const double tail = as_f64(tab[idx]);
    uint64_t sbits = tab[idx + 1] + top;
    const double r2 = mul(r, r);
    // tmp = tail + r*C1 + r2*(C2 + r*C3) + r2*r2*(C4 + r*C5), left to right
    double tmp = add(tail, mul(r, C1));
    tmp = add(tmp, mul(r2, add(C2, mul(r, C3))));
    tmp = add(tmp, mul(mul(r2, r2), add(C4, mul(r, C5))));
    if (!special) {
        const double scale = as_f64(sbits);
        return add(scale, mul(scale, tmp));
    }
    // results near the ends of the exponent range
    if ((ki & 0x80000000ull) == 0) {  // k > 0: the exponent of scale may have overflowed by one
        sbits -= 1ull << 52;
        const double scale = as_f64(sbits);
        return mul(2.0, add(scale, mul(scale, tmp)));
    }
    sbits += 1022ull << 52;  // k < 0: care in the subnormal range
    const double scale = as_f64(sbits);
    double y = add(scale, mul(scale, tmp));
    if (y < 1.0) {
        double lo = add(sub(scale, y), mul(scale, tmp));
        const double hi = add(1.0, y);
        lo = add(add(sub(1.0, hi), y), lo);
        y = sub(add(hi, lo), 1.0);
        if (y == 0.0) y = 0.0;
    }
    return mul(0x1p-1022, y);
}

}  // namespace pnx_exp2
