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

// row pitch of the terms: n + 1 (odd for the usual even n).  Rounding it up to whole 64-byte lines puts
// consecutive m of a column a multiple of 8 KiB apart -- on the same HBM channels: K7a 0.59 -> 0.78 ms.
__host__ __device__ static inline size_t term_ld(uint32_t n) { return (size_t)n + 1; }

// admissible j range of (i, m): hist.rs:164-166
//   for j in max(m_quorum, c)..m { if n + j + 1 > i + m && j <= i { ... } }   with i in m_quorum..n
__host__ __device__ static inline void j_range(uint32_t n, uint32_t c, uint32_t mq, uint32_t i, uint32_t m,
                                               uint32_t &jlo, uint32_t &jhi) {
    jlo = mq > c ? mq : c;
    if ((uint64_t)i + m > (uint64_t)n + jlo) jlo = i + m - n;  // n + j + 1 > i + m  <=>  j >= i + m - n
    jhi = m < i + 1 ? m : i + 1;                                  // j < m and j <= i
    if (i < mq || i >= n || jlo > jhi) jhi = jlo;                 // empty
}

// terms[(i_local * (n + 1) + m) * ld + j], ld = term_ld(n).  One workgroup = one i and 256 consecutive j; the
// log2 table, which every step of the m loop reads at two lane-dependent indices, is staged in LDS
// first when it fits (LDS_TABLES): from global memory the loop ran at the latency of those reads
// (1.26 ms for n = 1024).
template <bool LDS_TABLES>
__global__ __launch_bounds__(256) void k_quorum_terms(uint32_t n, uint32_t c, uint32_t i0, uint32_t i1,
                                                       const uint32_t *__restrict__ g_mq,
                                                       const double *__restrict__ g_L, const double *__restrict__ g_mf,
                                                       const double *__restrict__ g_nf, double *__restrict__ terms) {
    extern __shared__ double sh_tab[];
    __shared__ uint64_t s_exp2[256];  // the exp2 table is read twice per term at a lane-dependent index
    s_exp2[threadIdx.x] = c_exp2_tab[threadIdx.x];
    if (!LDS_TABLES) __syncthreads();
    const uint32_t i = i0 + blockIdx.x;
    const uint32_t j = blockIdx.y * 256 + threadIdx.x;
    const double *L = g_L, *m_fact = g_mf, *n_fall = g_nf;
    const uint32_t *m_quorum = g_mq;
    if (LDS_TABLES) {
        // only the log2 table is read at lane-dependent indices; m_fact[m], n_fall[m] and m_quorum[m]
        // are wave-uniform reads (see the loop) and come through the scalar cache
        double *sL = sh_tab;
        for (uint32_t k = threadIdx.x; k < 2 * (n + 1); k += 256) sL[k] = g_L[k];
        __syncthreads();
        L = sL;
    }
    // The m loop is wave-uniform (it starts at the first j of the wave + 1 and a lane joins at its
    // own j + 1): i and m then live in scalar registers, and the admissible range, m_quorum[m],
    // m_fact[m] and n_fall[m] are computed / read once per wave and step instead of once per lane.
    const uint32_t wave_j0 = blockIdx.y * 256 + ((uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.x) & ~63u);
    bool alive = i < i1 && j <= i && j < n;
    double q = 0.0;
    const size_t ld = term_ld(n);
    double *row = terms + (size_t)(i < i1 ? i - i0 : 0) * (n + 1) * ld;
    // choose(i, j), hist.rs:21-36 (res += log2(i - a); res -= log2(a + 1)): what q is seeded with
    // whenever it is 0.0 (hist.rs:167-169).  All lanes compute theirs together, before the walk.
    double seed = 0.0;
    if (alive) {
        const uint32_t k = j > i - j ? i - j : j;
        for (uint32_t a = 0; a < k; ++a) {
            seed = pnx_exp2::add(seed, L[i - a]);
            seed = pnx_exp2::sub(seed, L[a + 1]);
        }
    }
    for (uint32_t m = wave_j0 + 1; m <= n; ++m) {
        if (!__builtin_amdgcn_ballot_w64(alive)) break;
        uint32_t jlo, jhi;
        j_range(n, c, m_quorum[m], i, m, jlo, jhi);
        const double mf = m_fact[m], nf = n_fall[m];
        if (alive && m > j) {
            // the lower bound only rises with m (m_quorum and i + m - n do), and j < m, j <= i hold
            // here: once j falls below it, no later m is admissible
            if (j < jlo || j >= jhi) {
                alive = false;
            } else {
                if (q == 0.0) q = seed;
                q = pnx_exp2::add(q, L[n - i - m + 1 + j]);  // hist.rs:171
                q = pnx_exp2::sub(q, L[m - j]);              // hist.rs:172
                const double x = pnx_exp2::sub(pnx_exp2::add(q, mf), nf);
                row[(size_t)m * ld + j] = pnx_exp2::exp2_exact(x, s_exp2);
            }
        }
    }
}

// sum_q[i * (n + 1) + m], NaN where no j is admissible (add == false).  One wave = one i and 64
// consecutive m, one lane per m: every lane adds the terms of its own m in ascending j -- the
// reference's order -- but the terms come in through LDS in tiles of 64 m x 32 j, so that the
// global reads are whole 256-byte row pieces (a lane walking its own row would touch 8 bytes of
// every 64-byte sector, and one wave per (i, m) spends three instructions per term).
constexpr int QS_J = 32;
__global__ __launch_bounds__(64) void k_quorum_sums(uint32_t n, uint32_t c, uint32_t i0, uint32_t i1,
                                                     const uint32_t *__restrict__ m_quorum,
                                                     const double *__restrict__ terms, double *__restrict__ sum_q) {
    __shared__ double tile[64][QS_J + 1];
    const uint32_t lane = threadIdx.x;
    const uint32_t i = i0 + blockIdx.x;
    const uint32_t m0 = blockIdx.y * 64 + 1;  // m = m0 .. m0 + 63
    if (i >= i1) return;
    const uint32_t m = m0 + lane;
    uint32_t jlo = 0, jhi = 0;
    if (m <= n) j_range(n, c, m_quorum[m], i, m, jlo, jhi);
    // union of the j ranges of the 64 lanes
    uint32_t lo_all = jlo < jhi ? jlo : 0xFFFFFFFFu, hi_all = jlo < jhi ? jhi : 0u;
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = __shfl_xor(lo_all, o), b = __shfl_xor(hi_all, o);
        lo_all = a < lo_all ? a : lo_all;
        hi_all = b > hi_all ? b : hi_all;
    }
    const size_t ld = term_ld(n);
    const double *base = terms + (size_t)(i - i0) * (n + 1) * ld;
    double s = 0.0;
    const uint32_t half = lane >> 5, jj = lane & 31u;
    for (uint32_t jb = lo_all & ~(uint32_t)(QS_J - 1); jb < hi_all; jb += QS_J) {
        // rows r and r + 1 per step: lanes 0..31 / 32..63 read 32 consecutive j of one row each.
        // All 32 loads of a tile are issued before the first one is used.
        double v[32];
#pragma unroll
        for (uint32_t r = 0; r < 64; r += 2) {
            const uint32_t rr = r + half, mr = m0 + rr;
            // only the entries K7a wrote are read: the range of row rr, known to lane rr
            const uint32_t rlo = __shfl(jlo, rr), rhi = __shfl(jhi, rr);
            const uint32_t j = jb + jj;
            const bool ok = mr <= n && j >= rlo && j < rhi;
            const double *src = base + (ok ? (size_t)mr * ld + j : 0);  // clamped: the load itself is unconditional
            const double x = *src;
            v[r / 2] = ok ? x : 0.0;
        }
#pragma unroll
        for (uint32_t r = 0; r < 64; r += 2) tile[r + half][jj] = v[r / 2];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 8
        for (uint32_t k = 0; k < (uint32_t)QS_J; ++k) {
            const uint32_t j = jb + k;
            if (j >= jlo && j < jhi) s = pnx_exp2::add(s, tile[lane][k]);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (m <= n) sum_q[(size_t)i * (n + 1) + m] = jlo < jhi ? s : pnx_exp2::as_f64(0x7ff8000000000000ull);
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
    uint32_t slab = (uint32_t)std::max<size_t>(1, ((size_t)12 << 30) / (np1 * term_ld(n) * sizeof(double)));
    if (slab > n) slab = n;
    // inputs: [log2 table 2(n+1) | m_fact n+1 | n_fall n+1 | m_quorum n+1 (u32)] staged in pinned memory
    const size_t in_bytes = (4 * np1) * sizeof(double) + np1 * sizeof(uint32_t);
    const size_t out_bytes = np1 * np1 * sizeof(double);
    int rc;
    if ((rc = ensure(ctx, d_in, in_bytes)) || (rc = ensure(ctx, d_terms, (size_t)slab * np1 * term_ld(n) * sizeof(double))) ||
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
    // The closed form has no data dependence on the coverage passes (its inputs come from the host,
    // its scratch is its own), and it is arithmetic-bound where they are HBM-bound: on a stream of
    // its own it shares the CUs with a running pass instead of queueing behind it.
    // (a low stream priority changes nothing measurable: 4.2 ms per 1 k-path step either way)
    if (!ctx->stream_cf) PNX_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_cf, hipStreamNonBlocking));
    hipStream_t st = ctx->stream_cf;
    char *h_in = (char *)ctx->h_cf + out_bytes;
    std::memcpy(h_in, log2_tab, 2 * np1 * sizeof(double));
    std::memcpy(h_in + 2 * np1 * sizeof(double), m_fact, np1 * sizeof(double));
    std::memcpy(h_in + 3 * np1 * sizeof(double), n_fall, np1 * sizeof(double));
    std::memcpy(h_in + 4 * np1 * sizeof(double), m_quorum, np1 * sizeof(uint32_t));
    PNX_HIP(ctx, hipMemcpyAsync(d_in.p, h_in, in_bytes, hipMemcpyHostToDevice, st));
    const double *d_L = (const double *)d_in.p, *d_mf = d_L + 2 * np1, *d_nf = d_L + 3 * np1;
    const uint32_t *d_mq = (const uint32_t *)(d_L + 4 * np1);
    PNX_HIP(ctx, hipMemsetAsync(d_sum.p, 0xFF, out_bytes, st));  // NaN everywhere
    for (uint32_t i0 = 0; i0 < n; i0 += slab) {
        const uint32_t i1 = std::min(n, i0 + slab);
        const size_t tab_bytes = (2 * np1) * sizeof(double);  // the log2 table
        if (tab_bytes <= 96 * 1024) {
            if (tab_bytes > 64 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_quorum_terms<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_bytes);
            hipLaunchKernelGGL(k_quorum_terms<true>, dim3(i1 - i0, (n + 255) / 256), dim3(256), tab_bytes, st, n,
                               c, i0, i1, d_mq, d_L, d_mf, d_nf, (double *)d_terms.p);
        } else {
            hipLaunchKernelGGL(k_quorum_terms<false>, dim3(i1 - i0, (n + 255) / 256), dim3(256), 0, st, n, c, i0,
                               i1, d_mq, d_L, d_mf, d_nf, (double *)d_terms.p);
        }
        hipLaunchKernelGGL(k_quorum_sums, dim3(i1 - i0, (n + 63) / 64), dim3(64), 0, st, n, c, i0, i1, d_mq,
                           (const double *)d_terms.p, (double *)d_sum.p);
        PNX_HIP(ctx, hipGetLastError());
    }
    // results go to pinned host memory owned by the context (8 MB at n = 1024: a pageable copy
    // would cost as much as the kernels)
    PNX_HIP(ctx, hipMemcpyAsync(ctx->h_cf, d_sum.p, out_bytes, hipMemcpyDeviceToHost, st));
    PNX_HIP(ctx, hipEventRecord(ctx->ev_cf, st));
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
