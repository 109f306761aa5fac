// log2_exact.hpp -- the log2 of the platform libm, restated so that it can run on the GPU.
//
// The closed-form growth (Hist::calc_growth*, src/graph_broker/hist.rs:89-187) takes f64::log2 of the histogram
// bins, of small integers and -- in the quorum branch -- of the inner sums (:178-180); Rust forwards log2 to the
// platform libm.  glibc >= 2.28 computes it with the table-driven algorithm of the ARM optimized routines
// (e_log2.c): x = 2^k z, z in [0x1.6p-1, 0x1.6p0) falls into one of 64 subintervals with centre c;
// r = (z - chi - clo) * invc ~ z/c - 1 (the form for builds WITHOUT a fused multiply-add, which is what the x86-64
// libm runs), log2(x) = k + logc + r/ln2 (hi/lo split) + r^2 p(r); a longer polynomial near x = 1.  Every step is
// plain IEEE double arithmetic in a fixed order, so the same sequence gives the same bits on any IEEE machine as
// long as nothing is contracted into FMAs.  The words of `tab` come from the image's libm (log2_table.inc,
// tools/gen_log2_table.py); this restatement reproduced libm on 2*10^8 arguments of every kind (any bit pattern,
// integers, subnormals, the neighbourhood of 1) when it was written, and the host re-checks it against the running
// libm before a device result is ever used (growth_closed_form.cpp).
#pragma once
#include "exp2_exact.hpp"

namespace pnx_exp2 {

// `tab` = the 274 words of log2_table.inc
PNX_HD static inline double log2_exact(double x, const uint64_t *tab) {
    const double inv_hi = as_f64(tab[0]), inv_lo = as_f64(tab[1]);
    const uint64_t *A = tab + 2, *B = tab + 8, *T = tab + 18, *T2 = tab + 18 + 128;
    uint64_t ix = as_u64(x);
    const uint32_t top = (uint32_t)(ix >> 48);
    const uint64_t LO = 0x3feea4af00000000ull /* 1 - 0x1.5b51p-5 */, HI = 0x3ff0b55900000000ull /* 1 + 0x1.6ab2p-5 */;
    if (ix - LO < HI - LO) {  // close to 1
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = sub(x, 1.0);
        const double rhi = as_f64(as_u64(r) & 0xffffffff00000000ull), rlo = sub(r, rhi);
        const double hi = mul(rhi, inv_hi);
        double lo = add(mul(rlo, inv_hi), mul(r, inv_lo));
        const double r2 = mul(r, r), r4 = mul(r2, r2);
        const double p = mul(r2, add(as_f64(B[0]), mul(r, as_f64(B[1]))));
        double y = add(hi, p);
        lo = add(lo, add(sub(hi, y), p));
        const double q0 = add(add(as_f64(B[2]), mul(r, as_f64(B[3]))), mul(r2, add(as_f64(B[4]), mul(r, as_f64(B[5])))));
        const double q1 = add(add(as_f64(B[6]), mul(r, as_f64(B[7]))), mul(r2, add(as_f64(B[8]), mul(r, as_f64(B[9])))));
        lo = add(lo, mul(r4, add(q0, mul(r4, q1))));
        y = add(y, lo);
        return y;
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {  // x < 2^-1022, infinite or not a number
        if (ix * 2 == 0) return as_f64(0xfff0000000000000ull);       // log2(+-0) = -inf
        if (ix == 0x7ff0000000000000ull) return x;                   // log2(inf) = inf
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return as_f64(0x7ff8000000000000ull);  // negative, NaN
        ix = as_u64(mul(x, 0x1p52));  // subnormal: normalise
        ix -= 52ull << 52;
    }
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const uint32_t i = (uint32_t)(tmp >> (52 - 6)) % 64u;
    const int64_t k = (int64_t)tmp >> 52;  // arithmetic shift
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double invc = as_f64(T[2 * i]), logc = as_f64(T[2 * i + 1]);
    const double z = as_f64(iz), kd = (double)k;
    const double r = mul(sub(sub(z, as_f64(T2[2 * i])), as_f64(T2[2 * i + 1])), invc);
    const double rhi = as_f64(as_u64(r) & 0xffffffff00000000ull), rlo = sub(r, rhi);
    const double t1 = mul(rhi, inv_hi);
    const double t2 = add(mul(rlo, inv_hi), mul(r, inv_lo));
    const double t3 = add(kd, logc);
    const double hi = add(t3, t1);
    const double lo = add(add(sub(t3, hi), t1), t2);
    const double r2 = mul(r, r), r4 = mul(r2, r2);
    const double p = add(add(add(as_f64(A[0]), mul(r, as_f64(A[1]))), mul(r2, add(as_f64(A[2]), mul(r, as_f64(A[3]))))),
                         mul(r4, add(as_f64(A[4]), mul(r, as_f64(A[5])))));
    return add(add(lo, mul(r2, p)), hi);
}

}  // namespace pnx_exp2
