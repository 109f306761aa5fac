// kernels_pairs_mfma.hip -- K5 on the matrix cores: group x group intersections as an int8 product.
//
//      inter[a][b] = sum_i w_i * [i in group a] * [i in group b]        (similarity.rs:119-150)
// is  M · diag(w) · M^T  for the 0/1 presence matrix M (G x N): the one genuinely GEMM-shaped, compute-
// bound piece of this path (G^2 · N multiply-adds on 640 MB of input: 2.6e12 for 512 groups x 10 M items).
// Round 1 ran it on the vector ALUs -- AND + popcount per 32 items and pair, and for bp counts once per
// BIT PLANE of the weights (16 planes: 80 ms against 4.4 ms unweighted).  On the MFMA units the weights
// ride along for free: A = presence bits expanded to bytes 0/1, B = the same bytes times one 7-bit
// DIGIT of the weight (signed i8 holds 0..127), one v_mfma_i32_32x32x32_i8 per 32 items, 32 x 32 pairs
// and digit; a 16-bit weight is 3 digits instead of 16 planes, and the unweighted product is 1 "digit"
// that is always 1.  Exact: products <= 127, i32 accumulators are flushed into u64 partial sums per chunk
// of <= 2^17 words (2^17 * 32 * 127 < 2^31).
//
// Operand layout.  One k-step = one WORD position of the presence rows = 32 items (bit t of word w of a
// row = item (w / 64) * 2048 + w % 64 + 64 t).  Lane l of a wave holds row l % 32 and the
// 16 items of half l / 32 of the word: A = bits [16 h, 16 h + 16) of the word of group a0 + l % 32 as 16
// bytes, B = the same of group b0 + l % 32, each byte ANDed with 0xFF * bit and the item's digit.  The
// hardware pairs byte e of A's half h with byte e of B's half h whatever its internal k numbering is, so
// the sum over the 32 items is right by construction.  C/D: col = lane & 31, row = (reg & 3) + 8 (reg >> 2)
// + 4 (lane >> 5) (cdna_hip_programming.md "Fragment layout").
//
// Tiling: a workgroup of 4 waves owns a 128 x 128 tile of group pairs (only tiles on or above the diagonal)
// over a chunk of words, each wave a 64 x 64 quarter = 2 x 2 MFMA tiles x PL digits (<= 3: 192 accumulator
// registers); 32 words of the 128 + 128 rows and their digits are staged in LDS per step, double buffered.
// Per word a wave does 8 table lookups in LDS (8 bits -> 8 bytes each: 0 / 1 bytes for A, 0x00 / 0xFF masks for B) and
// 8 ANDs per digit -- against 4 PL MFMAs of 32 cycles, so with two workgroups per CU the matrix pipe of one wave runs
// beside the lookups of the other.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "pnx_context.hpp"

namespace pnx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MF_T = 128;    // groups per tile side
constexpr int MF_KS = 32;    // words per staging step
constexpr int MF_LD = 132;   // LDS stride of one word slice ([word][row])
constexpr uint32_t MF_CHUNK_MAX = 1u << 17;

// digits[p][w][t] = (weight of item (w / 64) * 2048 + w % 64 + 64 t  >> 7 p) & 127, one byte each
__global__ void k_weight_digits(const uint32_t *__restrict__ weights, uint32_t n_items, uint64_t row_words, uint32_t n_digits,
                                uint32_t *__restrict__ digits) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one dword = 4 items of one word position
    if (q >= row_words * 8) return;
    const uint64_t w = q >> 3;
    const uint32_t t0 = (uint32_t)(q & 7u) * 4u;
    uint32_t wt[4];
#pragma unroll
    for (uint32_t e = 0; e < 4; ++e) {
        const uint64_t item = (w >> 6) * BLOCK_ITEMS + (w & 63u) + 64ull * (t0 + e);
        wt[e] = item >= 1 && item <= n_items ? weights[item] : 0u;
    }
    for (uint32_t p = 0; p < n_digits; ++p) {
        uint32_t d = 0;
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) d |= ((wt[e] >> (7u * p)) & 127u) << (8u * e);
        digits[(uint64_t)p * row_words * 8 + q] = d;
    }
}

// 16 bits -> 16 bytes through a 256-entry table in LDS (8 bits -> 8 bytes per ds_read_b64): two lookups per operand
// instead of 12 (20) vector instructions -- the expansion, not the matrix pipe, bounded the first version of this kernel
__device__ static inline v4i expand16(const uint2 *__restrict__ tab, uint32_t word, uint32_t shift) {
    const uint2 lo = tab[(word >> shift) & 0xFFu], hi = tab[(word >> (shift + 8u)) & 0xFFu];
    v4i r;
    r[0] = (int)lo.x;
    r[1] = (int)lo.y;
    r[2] = (int)hi.x;
    r[3] = (int)hi.y;
    return r;
}
// entry b of the table: byte i = bit i of b times `one` (1 for the A operand, 0xFF for the masks of B)
__device__ static inline uint2 spread8(uint32_t b, uint32_t one) {
    const uint32_t lo = ((b & 0xFu) * 0x204081u) & 0x01010101u, hi = (((b >> 4) & 0xFu) * 0x204081u) & 0x01010101u;
    return make_uint2(lo * one, hi * one);
}

// PL digits [plane_base, plane_base + PL) of the weights (WEIGHTED), or the plain product (PL = 1)
// KS: words per staging step (32; 16 for the plain product, whose 39 KB of LDS per workgroup then leave room for three
// workgroups per CU instead of two -- the lookup -> MFMA chain of one wave hides behind more neighbours)
template <bool WEIGHTED, int PL, bool ACCUM, int KS>
__global__ __launch_bounds__(256, KS == 32 ? 2 : 3) void k_pair_mfma(const uint32_t *__restrict__ M, uint64_t row_words, uint32_t G, uint32_t n_side,
                                                      uint32_t chunk_words, const uint32_t *__restrict__ digits, uint32_t plane_base,
                                                      unsigned long long *__restrict__ partial) {
    const uint32_t ti = blockIdx.x / n_side, tj = blockIdx.x % n_side;
    if (ti > tj) return;
    const bool diag = ti == tj;
    const uint32_t pair = ti * n_side - ti * (ti + 1) / 2 + tj;
    const uint32_t n_pairs = n_side * (n_side + 1) / 2;
    const uint64_t w_begin = (uint64_t)blockIdx.y * chunk_words;
    const uint64_t w_end = w_begin + chunk_words < row_words ? w_begin + chunk_words : row_words;

    extern __shared__ __attribute__((aligned(16))) uint32_t lds_pairs[];  // > 64 KB: dynamic (mfma_lds_bytes)
    auto sA = [&](int b) { return lds_pairs + b * (KS * MF_LD); };
    auto sB = [&](int b) { return lds_pairs + (2 + b) * (KS * MF_LD); };
    auto sW = [&](int b) { return lds_pairs + 4 * KS * MF_LD + b * (PL * KS * 8); };
    uint2 *tab01 = reinterpret_cast<uint2 *>(lds_pairs + 4 * KS * MF_LD + 2 * PL * KS * 8);
    uint2 *tabff = tab01 + 256;
    tab01[threadIdx.x] = spread8(threadIdx.x, 1u);      // 256 threads, 256 entries each
    tabff[threadIdx.x] = spread8(threadIdx.x, 0xFFu);

    const uint32_t t = threadIdx.x;
    constexpr uint32_t TPR = KS / 4, RPP = 256 / TPR, NQ = MF_T / RPP;  // threads per row, rows per pass, passes
    const uint32_t ld_row = t / TPR, ld_k = (t % TPR) * 4u;  // staging: rows ld_row + RPP q, words ld_k .. ld_k + 3
    const uint32_t wave = t >> 6, lane = t & 63u, r = lane & 31u, h = lane >> 5;
    const uint32_t ra = (wave >> 1) * 64u + r, rb = (wave & 1u) * 64u + r;

    const uint32_t ga0 = ti * MF_T, gb0 = tj * MF_T;
    const uint4 *rowA[NQ], *rowB[NQ];
    uint32_t mA[NQ], mB[NQ];
#pragma unroll
    for (int q = 0; q < (int)NQ; ++q) {  // rows past G: clamped to a valid row, masked to zero after the load
        const uint32_t ga = ga0 + ld_row + RPP * q, gb = gb0 + ld_row + RPP * q;
        rowA[q] = reinterpret_cast<const uint4 *>(M + (uint64_t)(ga < G ? ga : G - 1) * row_words + ld_k);
        rowB[q] = reinterpret_cast<const uint4 *>(M + (uint64_t)(gb < G ? gb : G - 1) * row_words + ld_k);
        mA[q] = ga < G ? ~0u : 0u;
        mB[q] = gb < G ? ~0u : 0u;
    }

    v16i acc[PL][2][2];
#pragma unroll
    for (int p = 0; p < PL; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[p][i][j][e] = 0;

    uint4 va[NQ], vb[NQ];
    uint32_t vw[WEIGHTED ? PL : 1];
    static_assert(!WEIGHTED || KS == 32, "the digits of a staging step are fetched one dword per thread");
    auto fetch = [&](uint64_t w) {
#pragma unroll
        for (int q = 0; q < (int)NQ; ++q) {
            va[q] = rowA[q][w >> 2];
            if (!diag) vb[q] = rowB[q][w >> 2];
        }
        if (WEIGHTED) {
#pragma unroll
            for (int p = 0; p < PL; ++p) vw[p] = digits[((uint64_t)(plane_base + p) * row_words + w) * 8 + t];  // 32 words x 8 dwords
        }
    };
    auto put4 = [&](uint32_t *s, uint32_t row, const uint4 &v, uint32_t m) {
        s[(ld_k + 0) * MF_LD + row] = v.x & m;
        s[(ld_k + 1) * MF_LD + row] = v.y & m;
        s[(ld_k + 2) * MF_LD + row] = v.z & m;
        s[(ld_k + 3) * MF_LD + row] = v.w & m;
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < (int)NQ; ++q) {
            put4(sA(buf), ld_row + RPP * q, va[q], mA[q]);
            if (!diag) put4(sB(buf), ld_row + RPP * q, vb[q], mB[q]);
        }
        if (WEIGHTED) {
#pragma unroll
            for (int p = 0; p < PL; ++p) sW(buf)[p * KS * 8 + t] = vw[p];
        }
    };

    if (w_begin < w_end) {
        fetch(w_begin);
        stage(0);
    }
    __syncthreads();
    int buf = 0;
    for (uint64_t w = w_begin; w < w_end; w += KS, buf ^= 1) {
        const bool more = w + KS < w_end;
        if (more) fetch(w + KS);
        const uint32_t *a_s = sA(buf) + ra;
        const uint32_t *b_s = (diag ? sA(buf) : sB(buf)) + rb;
#pragma unroll 4
        for (int k = 0; k < KS; ++k) {
            const uint32_t a0 = a_s[k * MF_LD], a1 = a_s[k * MF_LD + 32];
            const uint32_t b0 = b_s[k * MF_LD], b1 = b_s[k * MF_LD + 32];
            const v4i fa[2] = {expand16(tab01, a0, 16u * h), expand16(tab01, a1, 16u * h)};
            if (!WEIGHTED) {
                const v4i mb[2] = {expand16(tab01, b0, 16u * h), expand16(tab01, b1, 16u * h)};
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[0][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], mb[j], acc[0][i][j], 0, 0, 0);
            } else {
                const v4i ff[2] = {expand16(tabff, b0, 16u * h), expand16(tabff, b1, 16u * h)};
#pragma unroll
                for (int p = 0; p < PL; ++p) {
                    const v4i dg = *reinterpret_cast<const v4i *>(sW(buf) + p * KS * 8 + k * 8 + 4 * h);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const v4i fb = ff[j] & dg;
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[p][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb, acc[p][i][j], 0, 0, 0);
                    }
                }
            }
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
    }

    // partial[(chunk * n_pairs + pair) * 128 * 128 + row * 128 + col]
    unsigned long long *out = partial + ((uint64_t)blockIdx.y * n_pairs + pair) * (MF_T * MF_T);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t row = (wave >> 1) * 64u + 32u * i + (e & 3) + 8u * (e >> 2) + 4u * h;
                const uint32_t col = (wave & 1u) * 64u + 32u * j + r;
                unsigned long long v = 0;
#pragma unroll
                for (int p = 0; p < PL; ++p) v += (unsigned long long)(uint32_t)acc[p][i][j][e] << (7u * p);
                v <<= 7u * plane_base;
                unsigned long long *dst = out + (uint64_t)row * MF_T + col;
                *dst = ACCUM ? *dst + v : v;
            }
}

// sum the chunks of one tile pair and write inter[a][b] (and inter[b][a] for off-diagonal tiles)
__global__ __launch_bounds__(256) void k_pair_mfma_reduce(const unsigned long long *__restrict__ partial, uint32_t n_chunks, uint32_t G,
                                                          uint32_t n_side, unsigned long long *__restrict__ inter) {
    const uint32_t ti = blockIdx.x / n_side, tj = blockIdx.x % n_side;
    if (ti > tj) return;
    const uint32_t pair = ti * n_side - ti * (ti + 1) / 2 + tj;
    const uint32_t n_pairs = n_side * (n_side + 1) / 2;
    const uint32_t el = blockIdx.y * 256 + threadIdx.x;  // 0 .. 128 * 128
    const unsigned long long *src = partial + (uint64_t)pair * (MF_T * MF_T) + el;
    const uint64_t stride = (uint64_t)n_pairs * (MF_T * MF_T);
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    uint32_t c = 0;
    for (; c + 4 <= n_chunks; c += 4) {
        s0 += src[(uint64_t)c * stride];
        s1 += src[(uint64_t)(c + 1) * stride];
        s2 += src[(uint64_t)(c + 2) * stride];
        s3 += src[(uint64_t)(c + 3) * stride];
    }
    for (; c < n_chunks; ++c) s0 += src[(uint64_t)c * stride];
    const unsigned long long s = (s0 + s1) + (s2 + s3);
    const uint32_t ga = ti * MF_T + el / MF_T, gb = tj * MF_T + el % MF_T;
    if (ga < G && gb < G) {
        if (ti != tj || ga <= gb) inter[(uint64_t)ga * G + gb] = s;
        if (ti != tj || ga < gb) inter[(uint64_t)gb * G + ga] = s;
    }
}

static size_t mfma_lds_bytes(uint32_t pl, uint32_t ks = MF_KS) { return (size_t)(4 * ks * MF_LD + 2 * pl * ks * 8 + 2 * 256 * 2) * sizeof(uint32_t); }

int launch_pair_intersections_mfma(pnx_ctx *ctx) {
    const uint32_t G = ctx->n_groups, NB = ctx->n_blocks;
    const uint64_t row_words = (uint64_t)NB * BLOCK_WORDS;
    const uint32_t n_side = (G + MF_T - 1) / MF_T;
    if ((uint64_t)n_side * n_side > 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "too many groups for the pair kernel");
    const uint64_t n_pairs = (uint64_t)n_side * (n_side + 1) / 2;
    int rc;
    uint32_t n_digits = 1;
    if (ctx->weighted) {
        if ((rc = ensure_weight_planes(ctx))) return rc;  // also finds the widest weight
        n_digits = (ctx->n_wplanes + 6) / 7;
        if (!ctx->wdigits_valid) {
            if ((rc = ensure(ctx, ctx->d_wdigits, (size_t)n_digits * row_words * 32))) return rc;
            hipLaunchKernelGGL(k_weight_digits, dim3((unsigned)((row_words * 8 + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const uint32_t *)ctx->d_weights.p, ctx->n_items, row_words, n_digits, (uint32_t *)ctx->d_wdigits.p);
            PNX_HIP(ctx, hipGetLastError());
            ctx->wdigits_valid = true;
        }
    }
    // chunks: enough workgroups for two rounds over the chip, every chunk a multiple of the staging step
    uint64_t n_chunks = (2048 + n_pairs - 1) / n_pairs;
    const uint64_t max_chunks = (row_words + 255) / 256;
    if (n_chunks > max_chunks) n_chunks = max_chunks;
    if (n_chunks < 1) n_chunks = 1;
    uint64_t chunk_words = ((row_words + n_chunks - 1) / n_chunks + MF_KS - 1) / MF_KS * MF_KS;
    if (chunk_words > MF_CHUNK_MAX) chunk_words = MF_CHUNK_MAX;
    n_chunks = (row_words + chunk_words - 1) / chunk_words;
    if (n_chunks > 65535) return ctx->fail(PNX_ELIMIT, "pair kernel: %llu word chunks exceed the grid", (unsigned long long)n_chunks);
    const size_t part_bytes = (size_t)n_chunks * n_pairs * MF_T * MF_T * sizeof(uint64_t);
    if ((rc = ensure(ctx, ctx->d_pair_partial, part_bytes))) return rc;
    prof_begin(ctx, PNX_K_PAIRS);
    const dim3 grid(n_side * n_side, (unsigned)n_chunks);
    const uint32_t *M = (const uint32_t *)ctx->d_M.p, *dg = (const uint32_t *)ctx->d_wdigits.p;
    unsigned long long *part = (unsigned long long *)ctx->d_pair_partial.p;
    if (!ctx->weighted) {
        static const bool ks32 = std::getenv("PNX_PAIRS_KS32") != nullptr;  // experiments: round 2's staging depth
        auto go = [&](auto kern, uint32_t ks) {
            const size_t lds = mfma_lds_bytes(1, ks);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, M, row_words, G, n_side, (uint32_t)chunk_words,
                               (const uint32_t *)nullptr, 0u, part);
        };
        if (ks32) go(k_pair_mfma<false, 1, false, 32>, 32);
        else go(k_pair_mfma<false, 1, false, 16>, 16);
    } else {
        for (uint32_t base = 0; base < n_digits; base += 3) {
            const uint32_t pl = n_digits - base < 3 ? n_digits - base : 3;
            const bool first = base == 0;
            auto go = [&](auto kern) {
                const size_t lds = mfma_lds_bytes(pl);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, M, row_words, G, n_side, (uint32_t)chunk_words, dg, base, part);
            };
            if (pl == 3) first ? go(k_pair_mfma<true, 3, false, 32>) : go(k_pair_mfma<true, 3, true, 32>);
            else if (pl == 2) first ? go(k_pair_mfma<true, 2, false, 32>) : go(k_pair_mfma<true, 2, true, 32>);
            else first ? go(k_pair_mfma<true, 1, false, 32>) : go(k_pair_mfma<true, 1, true, 32>);
        }
    }
    hipLaunchKernelGGL(k_pair_mfma_reduce, dim3(n_side * n_side, MF_T * MF_T / 256), dim3(256), 0, ctx->stream,
                       (const unsigned long long *)part, (uint32_t)n_chunks, G, n_side, (unsigned long long *)ctx->d_inter.p);
    prof_end(ctx);
    PNX_HIP(ctx, hipGetLastError());
    return PNX_OK;
}

}  // namespace pnx

namespace pnx {
// pnx_preload (see kernels_gfa.hip): touching one kernel loads the code object of this translation unit
void preload_pairs_mfma(unsigned what) {
    hipFuncAttributes a;
    auto touch = [&a](const void *k) { (void)hipFuncGetAttributes(&a, k); };
    if (what & PNX_PRELOAD_PAIRS) {
        touch((const void *)k_pair_mfma_reduce);
        touch((const void *)k_weight_digits);
    }
}
}  // namespace pnx
