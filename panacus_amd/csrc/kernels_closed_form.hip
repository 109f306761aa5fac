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
#include "log2_exact.hpp"
#include "pnx_context.hpp"

namespace pnx {

__device__ __constant__ uint64_t c_exp2_tab[256] = {
#include "exp2_table.inc"
};
__device__ __constant__ uint64_t c_log2_tab[274] = {
#include "log2_table.inc"
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

// sum_q[(n+1)^2] (NaN where no j is admissible) from inputs that are already on the device; scratch of the terms lives
// in the context (a fresh hipMalloc + hipFree of gigabytes per call costs more than the kernels)
static int launch_quorum_sums(pnx_ctx *ctx, hipStream_t st, DevBuf &d_terms, uint32_t n, uint32_t c, const uint32_t *d_mq, const double *d_L,
                              const double *d_mf, const double *d_nf, double *d_sum) {
    const size_t np1 = (size_t)n + 1;
    // rows per slab: up to 12 GiB of terms at a time (n = 1024: all rows in one launch, 16 K waves)
    uint32_t slab = (uint32_t)std::max<size_t>(1, ((size_t)12 << 30) / (np1 * term_ld(n) * sizeof(double)));
    if (slab > n) slab = n;
    int rc;
    if ((rc = ensure(ctx, d_terms, (size_t)slab * np1 * term_ld(n) * sizeof(double)))) return rc;
    PNX_HIP(ctx, hipMemsetAsync(d_sum, 0xFF, np1 * np1 * sizeof(double), st));  // NaN everywhere
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
                           (const double *)d_terms.p, d_sum);
        PNX_HIP(ctx, hipGetLastError());
    }
    return PNX_OK;
}

// ------------------------------------------------------------------------------------------
// K9: the closed forms themselves -- Hist::calc_growth_union / _core / _quorum (hist.rs:89-187) -- on the device,
// from the histogram to the growth values, in the reference's order of operations with the restated log2 / exp2.
//   k_cf_setup   L[v] = log2(v) for v = 0 .. 2n+1 (:21-36, :104, :129, :171-172), lh[i] = log2(hist[i]) (:106); per threshold
//                pair, one lane: n_fall, m_fact (running sums :100, :125, :148-149), m_quorum (:150), tot (:95-97)
//   k_cf_rows    one lane per histogram index i walks m: perc_mult and the ARGUMENT of the term's exp2 (:102-108, :127-133, :157-160)
//   k_cf_exp2    the exp2 of every term, one lane per (i, m)
//   (quorum pairs: K7a, K7b above for the inner sums, then)
//   k_cf_term2   exp2(log2(hist[i]) + log2(sum_q)) per (i, m) (:178-180)
//   k_cf_finish  one lane per m adds its terms in ascending i (:107, :132, :159, :180) and closes the value (:110, :135, :183)
// Term arrays are [i][m]: every kernel reads and writes whole lines (the row walk through an LDS tile).
// ------------------------------------------------------------------------------------------
enum { CF_UNION = PNX_GROWTH_UNION, CF_CORE = PNX_GROWTH_CORE, CF_QUORUM = PNX_GROWTH_QUORUM };

// one workgroup: the log2 tables, then the two running sums -- the only sequential part: lane 0 walks n_fall, lane 1
// m_fact, nothing but one LDS read, one addition and one LDS write per step (they depend on n alone, so every pair gets
// a copy) -- then everything per (pair, m) in parallel again
__global__ __launch_bounds__(256) void k_cf_setup(uint32_t n, uint32_t n_pairs, const uint64_t *__restrict__ hist, double *__restrict__ L,
                                                  double *__restrict__ lh, const uint32_t *__restrict__ branch,
                                                  const uint32_t *__restrict__ cov, const double *__restrict__ quorum,
                                                  double *__restrict__ n_fall, double *__restrict__ m_fact,
                                                  uint32_t *__restrict__ m_quorum, double *__restrict__ tot) {
    extern __shared__ double s_dyn[];  // L: 2 (n + 1) values | n_fall: n + 1 | m_fact: n + 1
    __shared__ uint64_t s_log2[274];
    const uint32_t np1 = n + 1, nl = 2 * np1;
    double *s_L = s_dyn, *s_nf = s_dyn + nl, *s_mf = s_nf + np1;
    for (uint32_t k = threadIdx.x; k < 274; k += 256) s_log2[k] = c_log2_tab[k];
    __syncthreads();
    for (uint32_t v = threadIdx.x; v < nl; v += 256) {
        const double x = pnx_exp2::log2_exact((double)v, s_log2);
        s_L[v] = x;
        L[v] = x;
    }
    for (uint32_t i = threadIdx.x; i <= n; i += 256) lh[i] = pnx_exp2::log2_exact((double)hist[i], s_log2);
    __syncthreads();
    if (threadIdx.x < 2) {
        // hist.rs:100 / :125 / :148: n_fall += log2(n - m + 1); :149: m_fact += log2(m).  Sixteen table values at a time
        // go to registers first: the running sum is a chain of dependent additions, the LDS reads must not be part of it
        const bool fall = threadIdx.x == 0;
        double *dst = fall ? s_nf : s_mf;
        double a = 0.0;
        dst[0] = 0.0;
        for (uint32_t m0 = 1; m0 <= n; m0 += 16) {
            double v[16];
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) {
                const uint32_t m = m0 + k < n ? m0 + k : n;
                v[k] = s_L[fall ? n - m + 1 : m];
            }
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) {
                a = pnx_exp2::add(a, v[k]);
                v[k] = a;
            }
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k)
                if (m0 + k <= n) dst[m0 + k] = v[k];
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_pairs * np1; k += 256) {
        const uint32_t t = k / np1, m = k % np1;
        const bool quo = branch[t] == CF_QUORUM;
        n_fall[k] = s_nf[m];
        m_fact[k] = quo ? s_mf[m] : 0.0;
        m_quorum[k] = quo && m ? (uint32_t)ceil(pnx_exp2::mul((double)m, quorum[t])) : 0u;  // hist.rs:150
    }
    // hist.rs:95-97: tot = sum of hist[c..] as integers (exact in any order), converted once
    __shared__ unsigned long long s_tot[PNX_GROWTH_MAX_PAIRS];
    if (threadIdx.x < n_pairs) s_tot[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t t = 0; t < n_pairs; ++t) {
        if (branch[t] != CF_UNION) continue;
        unsigned long long part = 0;
        for (uint32_t i = cov[t] + threadIdx.x; i <= n; i += 256) part += hist[i];
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if ((threadIdx.x & 63) == 0 && part) atomicAdd(&s_tot[t], part);
    }
    __syncthreads();
    if (threadIdx.x < n_pairs) tot[threadIdx.x] = (double)s_tot[threadIdx.x];
}

// One wave = 64 histogram indices i of one pair.  A lane walks m and keeps perc_mult -- the one sequential quantity: a
// running sum of table values (hist.rs:104, :129, :158), read from a copy of the table in LDS, one addition per step --
// and hands it, through a 64 x 64 LDS tile, to term1[i][m] as whole lines.  Everything else of the term is done by
// k_cf_exp2, one lane per (i, m): inside this walk it would be n dependent-latency evaluations per lane.
__global__ __launch_bounds__(64) void k_cf_rows(uint32_t n, const double *__restrict__ L, const uint32_t *__restrict__ branch,
                                                const uint32_t *__restrict__ cov, double *__restrict__ term1) {
    extern __shared__ double s_L[];  // 2 (n + 1) values
    __shared__ double tile[64][65];
    const uint32_t lane = threadIdx.x, i0 = blockIdx.x * 64, i = i0 + lane, t = blockIdx.y;
    const uint32_t nl = 2 * (n + 1);
    const size_t np1 = (size_t)n + 1;
    for (uint32_t k = lane; k < nl; k += 64) s_L[k] = L[k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    double *t1 = term1 + t * np1 * np1;
    const uint32_t c = cov[t];
    const bool uni = branch[t] == CF_UNION;
    const bool live = i <= n && i >= c;
    // m range of lane i: union 1 .. n - i, otherwise 1 .. min(i, n); of the whole wave: up to the longest
    const uint32_t my_top = !live ? 0u : (uni ? n - i : (i < n ? i : n));
    uint32_t top = my_top;
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t x = __shfl_xor(top, o);
        top = x > top ? x : top;
    }
    // table index of step m: union n - m - i + 1, else i - m + 1 (both fall by one per step)
    const int32_t idx0 = uni ? (int32_t)n - (int32_t)i + 1 : (int32_t)i + 1;
    double pm = 0.0;
    for (uint32_t m0 = 1; m0 <= top; m0 += 64) {
#pragma unroll
        for (uint32_t kb = 0; kb < 64; kb += 16) {
            double v[16];
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) {
                const uint32_t m = m0 + kb + k;
                // beyond the lane's range the table value is replaced by +0.0: perc_mult + 0.0 is perc_mult
                const double lv = s_L[m <= my_top ? idx0 - (int32_t)m : 0];
                v[k] = m <= my_top ? lv : 0.0;
            }
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) {
                pm = pnx_exp2::add(pm, v[k]);
                tile[lane][kb + k] = pm;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 8
        for (uint32_t r = 0; r < 64; ++r) {
            const uint32_t ir = i0 + r, m = m0 + lane;
            if (ir <= n && m <= n) t1[(size_t)ir * np1 + m] = tile[r][lane];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// term1[t][i][m] <- exp2((log2 h[i] + perc_mult[i][m]) - n_fall[m]) where (i, m) is an entry the sums will read
// (hist.rs:105-106, :130-131, :159)
__global__ __launch_bounds__(256) void k_cf_exp2(uint32_t n, const uint32_t *__restrict__ branch, const uint32_t *__restrict__ cov,
                                                 const double *__restrict__ lh, const double *__restrict__ n_fall,
                                                 double *__restrict__ term1) {
    __shared__ uint64_t s_exp2[256];
    s_exp2[threadIdx.x] = c_exp2_tab[threadIdx.x];
    __syncthreads();
    const uint32_t m = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y, t = blockIdx.z;
    const size_t np1 = (size_t)n + 1;
    if (m < 1 || m > n || i < cov[t]) return;
    if (branch[t] == CF_UNION ? i + m > n : m > i) return;
    double *p = term1 + t * np1 * np1 + (size_t)i * np1 + m;
    *p = pnx_exp2::exp2_exact(pnx_exp2::sub(pnx_exp2::add(lh[i], *p), n_fall[t * np1 + m]), s_exp2);
}

// term2[i][m] = exp2(log2 h[i] + log2 sum_q[i][m]), NaN where no j was admissible
__global__ __launch_bounds__(256) void k_cf_term2(uint32_t n, const double *__restrict__ lh, const double *__restrict__ sum_q,
                                                  double *__restrict__ term2) {
    __shared__ uint64_t s_exp2[256];
    __shared__ uint64_t s_log2[274];
    s_exp2[threadIdx.x] = c_exp2_tab[threadIdx.x];
    for (uint32_t k = threadIdx.x; k < 274; k += 256) s_log2[k] = c_log2_tab[k];
    __syncthreads();
    const uint32_t m = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    const size_t np1 = (size_t)n + 1;
    if (m < 1 || m > n || i >= n) return;
    const double sq = sum_q[i * np1 + m];
    term2[i * np1 + m] = sq == sq ? pnx_exp2::exp2_exact(pnx_exp2::add(lh[i], pnx_exp2::log2_exact(sq, s_log2)), s_exp2)
                                  : pnx_exp2::as_f64(0x7ff8000000000000ull);
}

// One lane per m adds its terms in ascending i -- the reference's order.  The sum is a chain of dependent additions, so
// the memory latency must stay off it: a workgroup of 4 waves serves 64 values of m; all four fetch the next 128 rows
// of the column strip ([i][m] layout: whole 512-byte lines) into registers and park them in LDS, while the first wave
// walks the previous 128 rows out of LDS, one addition per step.
constexpr int CF_CHUNK = 128;  // rows per LDS buffer (2 buffers x 128 x 64 x 8 B = 128 KB)

__device__ static inline double cf_column_sum(const double *__restrict__ col, size_t ld, uint32_t lo, uint32_t hi, uint32_t n_rows,
                                              bool skip_nan, double *buf /* 2 x CF_CHUNK x 64 */) {
    __shared__ uint32_t s_range[2];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // [lo, hi) of this lane (wave 0 holds the lanes' ranges); the workgroup walks the union
    __syncthreads();  // a previous call of the workgroup is over: its range and its buffers are free
    if (threadIdx.x == 0) {
        s_range[0] = 0xFFFFFFFFu;
        s_range[1] = 0u;
    }
    __syncthreads();
    if (wave == 0 && lo < hi) {
        atomicMin(&s_range[0], lo);
        atomicMax(&s_range[1], hi);
    }
    __syncthreads();
    const uint32_t wlo = s_range[0], whi = s_range[1];
    double y = 0.0;
    if (wlo >= whi) return y;
    constexpr int RPW = CF_CHUNK / 4;  // rows per wave and chunk
    double v[RPW];
    auto fetch = [&](uint32_t i0) {
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const uint32_t i = i0 + wave * RPW + (uint32_t)k;
            v[k] = col[(size_t)(i < n_rows ? i : n_rows - 1) * ld];  // unconditional, at a clamped row
        }
    };
    // parked already masked (outside the lane's range, or NaN = "no admissible j": +0.0, and y + 0.0 is y -- y is never
    // -0.0: it starts at +0.0 and the terms are >= +0.0), so that the walk is one LDS read and one addition per step
    auto park = [&](double *b, uint32_t i0) {
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const uint32_t i = i0 + wave * RPW + (uint32_t)k;
            const bool on = i >= lo && i < hi && (!skip_nan || v[k] == v[k]);
            b[(wave * RPW + (uint32_t)k) * 64 + lane] = on ? v[k] : 0.0;
        }
    };
    fetch(wlo);
    park(buf, wlo);
    __syncthreads();
    uint32_t which = 0;
    for (uint32_t i0 = wlo; i0 < whi; i0 += CF_CHUNK, which ^= 1u) {
        const bool more = i0 + CF_CHUNK < whi;
        if (more) fetch(i0 + CF_CHUNK);
        if (wave == 0) {
            const double *b = buf + which * (CF_CHUNK * 64);
#pragma unroll 32
            for (uint32_t k = 0; k < (uint32_t)CF_CHUNK; ++k) y = pnx_exp2::add(y, b[k * 64 + lane]);
        }
        if (more) park(buf + (which ^ 1u) * (CF_CHUNK * 64), i0 + CF_CHUNK);
        __syncthreads();
    }
    return y;
}

__global__ __launch_bounds__(256) void k_cf_finish(uint32_t n, const uint32_t *__restrict__ branch, const uint32_t *__restrict__ cov,
                                                   const uint32_t *__restrict__ m_quorum, const double *__restrict__ tot,
                                                   const double *__restrict__ term1, const double *__restrict__ term2,
                                                   double *__restrict__ out) {
    extern __shared__ double s_buf[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t m_raw = blockIdx.x * 64 + lane + 1, t = blockIdx.y;
    const bool in = m_raw <= n;
    const uint32_t m = in ? m_raw : n;
    const size_t np1 = (size_t)n + 1;
    const double *t1 = term1 + t * np1 * np1 + m;
    const uint32_t c = cov[t], br = branch[t];
    // union: i in c .. n - m; core / quorum: i in max(m, c) .. n
    const uint32_t lo = br == CF_UNION ? c : (m > c ? m : c);
    const uint32_t hi = br == CF_UNION ? (n >= m ? n - m + 1 : 0u) : n + 1;
    double y = cf_column_sum(t1, np1, in ? lo : 1u, in ? hi : 0u, n + 1, false, s_buf);
    if (br == CF_UNION) {
        y = pnx_exp2::sub(tot[t], y);
    } else if (br == CF_QUORUM) {
        const double *t2 = term2 + t * np1 * np1 + m;
        // hist.rs:163, :180-182: i in m_quorum .. n - 1, only where a j was admissible (NaN = add stayed false)
        const double yr = cf_column_sum(t2, np1, in ? m_quorum[t * np1 + m] : 1u, in ? n : 0u, n + 1, true, s_buf);
        y = pnx_exp2::add(y, yr);
    }
    if (in && threadIdx.x < 64) out[(size_t)t * n + m - 1] = y;
}

__global__ void k_log2_exact(const double *__restrict__ x, double *__restrict__ y, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = pnx_exp2::log2_exact(x[i], c_log2_tab);
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
    DevBuf &d_in = ctx->d_cf[0], &d_sum = ctx->d_cf[5];
    // inputs: [log2 table 2(n+1) | m_fact n+1 | n_fall n+1 | m_quorum n+1 (u32)] staged in pinned memory
    const size_t in_bytes = (4 * np1) * sizeof(double) + np1 * sizeof(uint32_t);
    const size_t out_bytes = np1 * np1 * sizeof(double);
    int rc;
    if ((rc = ensure(ctx, d_in, in_bytes)) || (rc = ensure(ctx, d_sum, out_bytes))) return rc;
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
    if ((rc = launch_quorum_sums(ctx, st, ctx->d_cf[4], n, c, d_mq, d_L, d_mf, d_nf, (double *)d_sum.p))) return rc;
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

int pnx_log2_exact(pnx_ctx *ctx, const double *x, double *y, uint64_t n) {
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
        hipLaunchKernelGGL(k_log2_exact, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const double *)dx.p, (double *)dy.p, n);
        e = hipMemcpyAsync(y, dy.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    release(dx);
    release(dy);
    if (e != hipSuccess) return ctx->fail(PNX_EHIP, "pnx_log2_exact: %s", hipGetErrorString(e));
    return PNX_OK;
}

int pnx_growth_closed_form_async(pnx_ctx *ctx, const uint64_t *hist, uint32_t n, uint32_t n_pairs, const uint32_t *branch,
                                 const uint32_t *cov_abs, const double *quorum_rel) {
    if (!ctx) return PNX_EINVAL;
    if (!branch || !cov_abs || !quorum_rel || n == 0 || n > PNX_GROWTH_MAX_N || n_pairs == 0 || n_pairs > PNX_GROWTH_MAX_PAIRS)
        return ctx->fail(PNX_EINVAL, "pnx_growth_closed_form: bad arguments (1 <= n <= %d, 1 <= pairs <= %d)", PNX_GROWTH_MAX_N,
                         PNX_GROWTH_MAX_PAIRS);
    for (uint32_t t = 0; t < n_pairs; ++t)
        if (branch[t] > PNX_GROWTH_QUORUM || cov_abs[t] == 0) return ctx->fail(PNX_EINVAL, "pnx_growth_closed_form: bad threshold pair %u", t);
    // (two slots once a slot's scratch passes 1 GiB: the terms of the quorum branch are (n + 1)^3 doubles per slot)
    const int slot_cap = ((size_t)n + 1) * (n + 1) * (n + 1) * 8 > ((size_t)1 << 30) ? std::min(2, ctx->max_in_flight) : ctx->max_in_flight;
    if (ctx->gslot_count && slot_cap != ctx->gslot_cap)
        return ctx->fail(PNX_EINVAL, "closed-form calls of very different sizes cannot be in flight together; fetch the pending ones first");
    if (ctx->gslot_count >= slot_cap) return ctx->fail(PNX_EINVAL, "%d closed-form calls are already in flight; fetch one first", slot_cap);
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    Ticket *src = nullptr;
    if (!hist) {  // the counters of the pass enqueued last, straight from the device
        if (ctx->tk_count == 0 || n != ctx->n_groups) return ctx->fail(PNX_EINVAL, "pnx_growth_closed_form: hist == NULL needs a coverage pass in flight with n groups");
        src = &ctx->tk[ctx->tk_last()];
    }
    const size_t np1 = (size_t)n + 1, T = n_pairs;
    if (ctx->gslot_count == 0) ctx->gslot_next = ctx->gslot_oldest = 0;  // (keeps the low slots in use when the cap is below the ring)
    pnx_ctx::GrowthSlot &g = ctx->gslot[ctx->gslot_next];
    if (!g.stream) PNX_HIP(ctx, hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    hipStream_t st = g.stream;
    // slot: [quorum f64 T | hist u64 n+1 | branch u32 T | cov u32 T] padded to 8, then out f64 T x n
    const size_t in_bytes = (T * 8 + np1 * 8 + T * 4 + T * 4 + 7) & ~(size_t)7, out_bytes = T * n * 8;
    int rc;
    if ((rc = ensure(ctx, g.d_io, in_bytes + out_bytes))) return rc;
    if (g.h_cap < in_bytes + out_bytes) {
        if (g.h_io) (void)hipHostFree(g.h_io);
        g.h_io = nullptr;
        g.h_cap = 0;
        PNX_HIP(ctx, hipHostMalloc(&g.h_io, in_bytes + out_bytes, hipHostMallocDefault));
        g.h_cap = in_bytes + out_bytes;
    }
    if (!g.done) PNX_HIP(ctx, hipEventCreateWithFlags(&g.done, hipEventDisableTiming | (ctx->blocking_sync ? hipEventBlockingSync : 0)));
    // the slot's own scratch: the chains of the two calls in flight run beside each other
    DevBuf &d_L = g.d_gc[0], &d_lh = g.d_gc[1], &d_nf = g.d_gc[2], &d_mf = g.d_gc[3], &d_mq = g.d_gc[4], &d_tot = g.d_gc[5],
           &d_t1 = g.d_gc[6], &d_t2 = g.d_gc[7], &d_sum = g.d_sum;
    bool any_quorum = false;
    for (uint32_t t = 0; t < n_pairs; ++t) any_quorum |= branch[t] == PNX_GROWTH_QUORUM;
    if ((rc = ensure(ctx, d_L, 2 * np1 * 8)) || (rc = ensure(ctx, d_lh, np1 * 8)) || (rc = ensure(ctx, d_nf, T * np1 * 8)) ||
        (rc = ensure(ctx, d_mf, T * np1 * 8)) || (rc = ensure(ctx, d_mq, T * np1 * 4)) || (rc = ensure(ctx, d_tot, T * 8)) ||
        (rc = ensure(ctx, d_t1, T * np1 * np1 * 8)) || (any_quorum && ((rc = ensure(ctx, d_t2, T * np1 * np1 * 8)) || (rc = ensure(ctx, d_sum, np1 * np1 * 8)))))
        return rc;
    char *h = (char *)g.h_io;
    double *h_q = (double *)h;
    uint64_t *h_hist = (uint64_t *)(h + T * 8);
    uint32_t *h_br = (uint32_t *)(h + T * 8 + np1 * 8), *h_cov = h_br + T;
    for (uint32_t t = 0; t < n_pairs; ++t) {
        h_q[t] = quorum_rel[t];
        h_br[t] = branch[t];
        h_cov[t] = cov_abs[t];
    }
    if (hist) std::memcpy(h_hist, hist, np1 * 8);
    char *d = (char *)g.d_io.p;
    const double *d_q = (const double *)d;
    const uint64_t *d_hist = (const uint64_t *)(d + T * 8);
    const uint32_t *d_br = (const uint32_t *)(d + T * 8 + np1 * 8), *d_cov = d_br + T;
    double *d_out = (double *)(d + in_bytes);
    PNX_HIP(ctx, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, st));
    if (src) {
        // the counters are final once the pass's own copy to the host was enqueued behind them (and behind the all-reduce of
        // a multi-GPU context): stream_cf waits for exactly that point
        PNX_HIP(ctx, hipStreamWaitEvent(st, src->done, 0));
        PNX_HIP(ctx, hipMemcpyAsync(d + T * 8, src->d_hist, np1 * 8, hipMemcpyDeviceToDevice, st));  // the slot's own copy
        if (!src->ev_reader) PNX_HIP(ctx, hipEventCreateWithFlags(&src->ev_reader, hipEventDisableTiming));
        PNX_HIP(ctx, hipEventRecord(src->ev_reader, st));
        src->has_reader = true;
    }
    const size_t lds_setup = 4 * np1 * sizeof(double), lds_rows = 2 * np1 * sizeof(double);  // tables staged in LDS
    if (n > 1000) {  // beyond the default 64 KB per workgroup (k_cf_rows also holds a 33 KB tile)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_cf_setup), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_setup);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_cf_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows);
    }
    hipLaunchKernelGGL(k_cf_setup, dim3(1), dim3(256), lds_setup, st, n, n_pairs, d_hist, (double *)d_L.p, (double *)d_lh.p, d_br,
                       d_cov, d_q, (double *)d_nf.p, (double *)d_mf.p, (uint32_t *)d_mq.p, (double *)d_tot.p);
    hipLaunchKernelGGL(k_cf_rows, dim3((unsigned)((np1 + 63) / 64), n_pairs), dim3(64), lds_rows, st, n, (const double *)d_L.p, d_br, d_cov,
                       (double *)d_t1.p);
    hipLaunchKernelGGL(k_cf_exp2, dim3((unsigned)((np1 + 255) / 256), (unsigned)np1, n_pairs), dim3(256), 0, st, n, d_br, d_cov,
                       (const double *)d_lh.p, (const double *)d_nf.p, (double *)d_t1.p);
    PNX_HIP(ctx, hipGetLastError());
    for (uint32_t t = 0; t < n_pairs; ++t) {
        if (branch[t] != PNX_GROWTH_QUORUM) continue;
        if ((rc = launch_quorum_sums(ctx, st, g.d_terms, n, cov_abs[t], (const uint32_t *)d_mq.p + t * np1, (const double *)d_L.p,
                                     (const double *)d_mf.p + t * np1, (const double *)d_nf.p + t * np1, (double *)d_sum.p)))
            return rc;
        hipLaunchKernelGGL(k_cf_term2, dim3((unsigned)((np1 + 255) / 256), n), dim3(256), 0, st, n, (const double *)d_lh.p, (const double *)d_sum.p,
                           (double *)d_t2.p + t * np1 * np1);
    }
    {
        static bool once = false;  // 128 KB of LDS per workgroup: beyond the default limit
        if (!once) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_cf_finish), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CF_CHUNK * 64 * 8);
            once = true;
        }
    }
    hipLaunchKernelGGL(k_cf_finish, dim3((n + 63) / 64, n_pairs), dim3(256), 2 * CF_CHUNK * 64 * 8, st, n, d_br, d_cov, (const uint32_t *)d_mq.p,
                       (const double *)d_tot.p, (const double *)d_t1.p, (const double *)d_t2.p, d_out);
    PNX_HIP(ctx, hipGetLastError());
    PNX_HIP(ctx, hipMemcpyAsync(h + in_bytes, d_out, out_bytes, hipMemcpyDeviceToHost, st));
    PNX_HIP(ctx, hipEventRecord(g.done, st));
    g.pending = true;
    g.n = n;
    g.n_pairs = n_pairs;
    g.out_off = in_bytes;
    ctx->gslot_next = (ctx->gslot_next + 1) % slot_cap;
    ctx->gslot_count += 1;
    ctx->gslot_cap = slot_cap;
    return PNX_OK;
}

int pnx_growth_closed_form_fetch(pnx_ctx *ctx, double *out) {
    if (!ctx || !out) return PNX_EINVAL;
    if (ctx->gslot_count == 0) return ctx->fail(PNX_EINVAL, "pnx_growth_closed_form_fetch: nothing was enqueued");
    pnx_ctx::GrowthSlot &g = ctx->gslot[ctx->gslot_oldest];
    PNX_HIP(ctx, hipEventSynchronize(g.done));
    std::memcpy(out, (const char *)g.h_io + g.out_off, (size_t)g.n_pairs * g.n * 8);
    g.pending = false;
    ctx->gslot_oldest = (ctx->gslot_oldest + 1) % ctx->gslot_cap;
    ctx->gslot_count -= 1;
    return PNX_OK;
}

}  // extern "C"
