// kernels_closed_form.hip -- the O(n^3) inner sums of the quorum growth closed form on the GPU.
//
// Hist::calc_growth_quorum (src/graph_broker/hist.rs:138-187) evaluates, for every m and every
// histogram index i, sum_q = sum_j exp2(q[i][j] + m_fact - n_fall_m) over the admissible j
// (:164-176), where q[i][j] is a running sum of log2 terms that only depends on (n, c, quorum).
// That is n^3/6 libm exp2 calls -- 1.8e8 for n = 1024, which is what bounds `histgrowth` on a
// 1000-path graph once the histogram itself takes 3 ms.  Everything in it is plain IEEE double
// arithmetic in a fixed order, so it can run here bit for bit:
//   K7a  one thread per (i, j): walks m = 1..n, keeps q[i][j] exactly like the reference (same
//        additions in the same order, choose(i, j) seeded from the log2 table the HOST computed
//        with libm), and stores the term exp2(...) of every admissible m  (exp2_exact.hpp)
//   K7b  one wave per (i, m): adds the terms of its j range in ascending j with a single
//        lane, sequentially -- the reference's order of additions
// The host then finishes each (i, m) with libm: exp2(log2(h[i]) + log2(sum_q)) (:178-180).
#include <algorithm>
#include <cstring>

#include "exp2_exact.hpp"
#include "pnx_context.hpp"

namespace pnx {

__device__ __constant__ uint64_t c_exp2_tab[256] = {
#include "exp2_table.inc"
};

// admissible j range of (i, m): hist.rs:164-166
//   for j in max(m_quorum, c)..m { if n + j + 1 > i + m && j <= i { ... } }   with i in m_quorum..n
__host__ __device__ static inline void j_range(uint32_t n, uint32_t c, uint32_t mq, uint32_t i, uint32_t m,
                                               uint32_t &jlo, uint32_t &jhi) {
    jlo = mq > c ? mq : c;
    if ((uint64_t)i + m > (uint64_t)n + jlo) jlo = i + m - n;  // n + j + 1 > i + m  <=>  j >= i + m - n
    jhi = m < i + 1 ? m : i + 1;                                  // j < m and j <= i
    if (i < mq || i >= n || jlo > jhi) jhi = jlo;                 // empty
}

// terms[(i_local * (n + 1) + m) * (n + 1) + j]
__global__ __launch_bounds__(64) void k_quorum_terms(uint32_t n, uint32_t c, uint32_t i0, uint32_t i1,
                                                      const uint32_t *__restrict__ m_quorum,
                                                      const double *__restrict__ L, const double *__restrict__ m_fact,
                                                      const double *__restrict__ n_fall, double *__restrict__ terms) {
    const uint32_t i = i0 + blockIdx.x;
    const uint32_t j = blockIdx.y * 64 + threadIdx.x;
    if (i >= i1 || j > i || j >= n) return;
    double q = 0.0;
    double *row = terms + (size_t)(i - i0) * (n + 1) * (n + 1);
    for (uint32_t m = j + 1; m <= n; ++m) {
        uint32_t jlo, jhi;
        j_range(n, c, m_quorum[m], i, m, jlo, jhi);
        // the lower bound only rises with m (m_quorum and i + m - n do), and j < m, j <= i hold in
        // this loop: once j falls below it, no later m is admissible
        if (j < jlo || j >= jhi) break;
        if (q == 0.0) {  // choose(i, j), hist.rs:21-36: res += log2(i - a); res -= log2(a + 1)
            const uint32_t k = j > i - j ? i - j : j;
            double res = 0.0;
            for (uint32_t a = 0; a < k; ++a) {
                res = pnx_exp2::add(res, L[i - a]);
                res = pnx_exp2::sub(res, L[a + 1]);
            }
            q = res;
        }
        q = pnx_exp2::add(q, L[n - i - m + 1 + j]);  // hist.rs:171
        q = pnx_exp2::sub(q, L[m - j]);              // hist.rs:172
        const double x = pnx_exp2::sub(pnx_exp2::add(q, m_fact[m]), n_fall[m]);
        row[(size_t)m * (n + 1) + j] = pnx_exp2::exp2_exact(x, c_exp2_tab);
    }
}

// sum_q[i * (n + 1) + m], NaN where no j is admissible (add == false)
__global__ __launch_bounds__(256) void k_quorum_sums(uint32_t n, uint32_t c, uint32_t i0, uint32_t i1,
                                                      const uint32_t *__restrict__ m_quorum,
                                                      const double *__restrict__ terms, double *__restrict__ sum_q) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wid = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t n_pairs = (uint64_t)(i1 - i0) * n;  // m = 1..n
    if (wid >= n_pairs) return;
    const uint32_t i = i0 + (uint32_t)(wid / n), m = (uint32_t)(wid % n) + 1;
    uint32_t jlo, jhi;
    j_range(n, c, m_quorum[m], i, m, jlo, jhi);
    const double *row = terms + ((size_t)(i - i0) * (n + 1) + m) * (n + 1);
    double s = 0.0;
    for (uint32_t b = jlo; b < jhi; b += 64) {
        const uint32_t j = b + lane;
        const double t = j < jhi ? row[j] : 0.0;
        const uint32_t cnt = jhi - b < 64 ? jhi - b : 64;
        // one lane adds the 64 terms in ascending j: the reference's order
        for (uint32_t u = 0; u < cnt; ++u) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pnx_exp2::as_u64(t), u);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pnx_exp2::as_u64(t) >> 32), u);
            s = pnx_exp2::add(s, pnx_exp2::as_f64(((uint64_t)hi << 32) | lo));
        }
    }
    if (lane == 0) sum_q[(size_t)i * (n + 1) + m] = jlo < jhi ? s : pnx_exp2::as_f64(0x7ff8000000000000ull);
}

__global__ void k_exp2_exact(const double *__restrict__ x, double *__restrict__ y, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = pnx_exp2::exp2_exact(x[i], c_exp2_tab);
}

}  // namespace pnx

using namespace pnx;

extern "C" {

int pnx_quorum_sums_async(pnx_ctx *ctx, uint32_t n, uint32_t c, const uint32_t *m_quorum, const double *log2_tab,
                          const double *m_fact, const double *n_fall) {
    if (!ctx) return PNX_EINVAL;
    if (!m_quorum || !log2_tab || !m_fact || !n_fall || n == 0 || n > 8192)
        return ctx->fail(PNX_EINVAL, "pnx_quorum_sums: bad arguments");
    if (ctx->cf_pending) return ctx->fail(PNX_EINVAL, "pnx_quorum_sums_async: the previous result has not been fetched");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    const size_t np1 = (size_t)n + 1;
    // scratch of the closed form lives in the context: a fresh hipMalloc + hipFree of gigabytes per
    // call costs more than the kernels
    DevBuf &d_in = ctx->d_cf[0], &d_terms = ctx->d_cf[4], &d_sum = ctx->d_cf[5];
    // rows per slab: up to 12 GiB of terms at a time (n = 1024: all rows in one launch, 16 K waves)
    uint32_t slab = (uint32_t)std::max<size_t>(1, ((size_t)12 << 30) / (np1 * np1 * sizeof(double)));
    if (slab > n) slab = n;
    // inputs: [log2 table 2(n+1) | m_fact n+1 | n_fall n+1 | m_quorum n+1 (u32)] staged in pinned memory
    const size_t in_bytes = (4 * np1) * sizeof(double) + np1 * sizeof(uint32_t);
    const size_t out_bytes = np1 * np1 * sizeof(double);
    int rc;
    if ((rc = ensure(ctx, d_in, in_bytes)) || (rc = ensure(ctx, d_terms, (size_t)slab * np1 * np1 * sizeof(double))) ||
        (rc = ensure(ctx, d_sum, out_bytes)))
        return rc;
    if (ctx->h_cf_cap < out_bytes + in_bytes) {
        if (ctx->h_cf) (void)hipHostFree(ctx->h_cf);
        ctx->h_cf = nullptr;
        ctx->h_cf_cap = 0;
        PNX_HIP(ctx, hipHostMalloc(&ctx->h_cf, out_bytes + in_bytes, hipHostMallocDefault));
        ctx->h_cf_cap = out_bytes + in_bytes;
    }
    if (!ctx->ev_cf) PNX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_cf, hipEventDisableTiming));
    char *h_in = (char *)ctx->h_cf + out_bytes;
    std::memcpy(h_in, log2_tab, 2 * np1 * sizeof(double));
    std::memcpy(h_in + 2 * np1 * sizeof(double), m_fact, np1 * sizeof(double));
    std::memcpy(h_in + 3 * np1 * sizeof(double), n_fall, np1 * sizeof(double));
    std::memcpy(h_in + 4 * np1 * sizeof(double), m_quorum, np1 * sizeof(uint32_t));
    PNX_HIP(ctx, hipMemcpyAsync(d_in.p, h_in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    const double *d_L = (const double *)d_in.p, *d_mf = d_L + 2 * np1, *d_nf = d_L + 3 * np1;
    const uint32_t *d_mq = (const uint32_t *)(d_L + 4 * np1);
    PNX_HIP(ctx, hipMemsetAsync(d_sum.p, 0xFF, out_bytes, ctx->stream));  // NaN everywhere
    for (uint32_t i0 = 0; i0 < n; i0 += slab) {
        const uint32_t i1 = std::min(n, i0 + slab);
        hipLaunchKernelGGL(k_quorum_terms, dim3(i1 - i0, (n + 63) / 64), dim3(64), 0, ctx->stream, n, c, i0, i1, d_mq,
                           d_L, d_mf, d_nf, (double *)d_terms.p);
        const uint64_t n_pairs = (uint64_t)(i1 - i0) * n;
        hipLaunchKernelGGL(k_quorum_sums, dim3((unsigned)((n_pairs + 3) / 4)), dim3(256), 0, ctx->stream, n, c, i0, i1,
                           d_mq, (const double *)d_terms.p, (double *)d_sum.p);
        PNX_HIP(ctx, hipGetLastError());
    }
    // results go to pinned host memory owned by the context (8 MB at n = 1024: a pageable copy
    // would cost as much as the kernels)
    PNX_HIP(ctx, hipMemcpyAsync(ctx->h_cf, d_sum.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    PNX_HIP(ctx, hipEventRecord(ctx->ev_cf, ctx->stream));
    ctx->cf_pending = true;
    return PNX_OK;
}

int pnx_quorum_sums_fetch(pnx_ctx *ctx, const double **sum_q) {
    if (!ctx || !sum_q) return PNX_EINVAL;
    if (!ctx->cf_pending) return ctx->fail(PNX_EINVAL, "pnx_quorum_sums_fetch: nothing was enqueued");
    ctx->cf_pending = false;
    PNX_HIP(ctx, hipEventSynchronize(ctx->ev_cf));
    *sum_q = (const double *)ctx->h_cf;
    return PNX_OK;
}

int pnx_quorum_sums(pnx_ctx *ctx, uint32_t n, uint32_t c, const uint32_t *m_quorum, const double *log2_tab,
                    const double *m_fact, const double *n_fall, const double **sum_q) {
    int rc = pnx_quorum_sums_async(ctx, n, c, m_quorum, log2_tab, m_fact, n_fall);
    if (rc) return rc;
    return pnx_quorum_sums_fetch(ctx, sum_q);
}

int pnx_exp2_exact(pnx_ctx *ctx, const double *x, double *y, uint64_t n) {
    if (!ctx || !x || !y) return PNX_EINVAL;
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf dx, dy;
    int rc;
    if ((rc = ensure(ctx, dx, (n ? n : 1) * sizeof(double))) || (rc = ensure(ctx, dy, (n ? n : 1) * sizeof(double)))) {
        release(dx);
        release(dy);
        return rc;
    }
    hipError_t e = hipMemcpyAsync(dx.p, x, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n) {
        hipLaunchKernelGGL(k_exp2_exact, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const double *)dx.p, (double *)dy.p, n);
        e = hipMemcpyAsync(y, dy.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    release(dx);
    release(dy);
    if (e != hipSuccess) return ctx->fail(PNX_EHIP, "pnx_exp2_exact: %s", hipGetErrorString(e));
    return PNX_OK;
}

}  // extern "C"
