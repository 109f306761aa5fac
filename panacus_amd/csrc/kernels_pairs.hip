// kernels_pairs.hip -- pairwise group intersections and the plain export of the presence matrix.
//
// K5 replaces the accumulation loop of Similarity::set_table
// (src/analyses/similarity.rs:119-150): for every item the reference walks its group slice
// c[r[i]..r[i+1]] and bumps path_lens[x] and path_similarities[(x, y)] for every ordered pair
// of the slice, i.e.
//      inter[a][b] = sum_i w_i * [i in group a] * [i in group b],   inter[a][a] = path_lens[a]
// with w_i = 1 (node / edge) or node_lens[i] (bp).  On the presence bit matrix M this is
//      inter[a][b] = sum_w popc(M[a][w] & M[b][w])                          (w_i = 1)
//                  = sum_p 2^p * sum_w popc(M[a][w] & M[b][w] & W_p[w])      (bp, weight planes)
// Integer set work: AND + popcount on the vector ALUs (2 instructions per pair and 32 items);
// no MFMA.  The kernel is register/LDS tiled like a matrix product because every row is used
// by G pairs: a workgroup owns a 64 x 64 tile of group pairs over a chunk of words, stages 32
// words of its 64 + 64 rows per step in LDS as [word][row] and every thread accumulates a 4 x 4
// micro-tile from two 16-byte LDS reads per word.  Only tiles on or above the diagonal are
// computed; partial sums go to a [chunk][tile pair] buffer with plain coalesced stores and a
// second kernel adds the chunks and writes both triangles (no atomics, deterministic).
//
// K6 turns the lane-interleaved rows of M into plain bit rows (bit n % 64 of u64 word n / 64
// = item n) for consumers above the ABI (the `table` writer, abacus.rs:1056-1178).
#include "pnx_context.hpp"

namespace pnx {

constexpr int PAIR_T = 64;    // groups per tile side
constexpr int PAIR_KS = 32;   // words per staging step
constexpr int PAIR_LD = 68;   // LDS stride of one word slice ([word][row]); 272 B keeps b128 reads aligned
constexpr uint32_t PAIR_WCHUNK_MAX = 1024;  // weighted: 16 planes * 32 items * 2^15 * 1024 words < 2^32

// popc(x) + acc in one instruction (the compiler otherwise pairs popcounts into v_add3_u32)
__device__ static inline uint32_t popc_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ static inline uint32_t lshl_add(uint32_t v, uint32_t s, uint32_t acc) { return (v << s) + acc; }

template <bool WEIGHTED>
__global__ __launch_bounds__(256) void k_pair_intersect(const uint32_t *__restrict__ M, uint64_t row_words,
                                                        uint32_t G, uint32_t n_side, uint32_t chunk_words,
                                                        const uint32_t *__restrict__ wplanes, uint32_t n_planes,
                                                        unsigned long long *__restrict__ partial) {
    const uint32_t ti = blockIdx.x / n_side, tj = blockIdx.x % n_side;
    if (ti > tj) return;
    const bool diag = ti == tj;
    const uint32_t pair = ti * n_side - ti * (ti + 1) / 2 + tj;  // rank of (ti, tj) among ti <= tj, row-major
    const uint32_t n_pairs = n_side * (n_side + 1) / 2;
    const uint64_t w_begin = (uint64_t)blockIdx.y * chunk_words;
    const uint64_t w_end = w_begin + chunk_words < row_words ? w_begin + chunk_words : row_words;

    __shared__ __attribute__((aligned(16))) uint32_t sA[2][PAIR_KS * PAIR_LD];
    __shared__ __attribute__((aligned(16))) uint32_t sB[2][PAIR_KS * PAIR_LD];
    __shared__ uint32_t sW[WEIGHTED ? PAIR_KS * 32 : 1];

    const uint32_t t = threadIdx.x;
    const uint32_t ld_row = t >> 3, ld_k = (t & 7u) * 4u;  // staging: row 0..31 (+32), words ld_k..ld_k+3
    const uint32_t wave = t >> 6, lane = t & 63u;
    const uint32_t ra = (wave >> 1) * 32u + (lane >> 3) * 4u;  // first A row of the micro-tile
    const uint32_t rb = (wave & 1u) * 32u + (lane & 7u) * 4u;  // first B row

    // rows past G are clamped to a valid row and masked to zero after the (unconditional,
    // 16-byte) load
    const uint32_t ga0 = ti * PAIR_T, gb0 = tj * PAIR_T;
    auto row_ptr = [&](uint32_t g) {
        return reinterpret_cast<const uint4 *>(M + (uint64_t)(g < G ? g : G - 1) * row_words + ld_k);
    };
    const uint4 *rowA0 = row_ptr(ga0 + ld_row), *rowA1 = row_ptr(ga0 + ld_row + 32);
    const uint4 *rowB0 = row_ptr(gb0 + ld_row), *rowB1 = row_ptr(gb0 + ld_row + 32);
    const uint32_t mA0 = ga0 + ld_row < G ? ~0u : 0u, mA1 = ga0 + ld_row + 32 < G ? ~0u : 0u;
    const uint32_t mB0 = gb0 + ld_row < G ? ~0u : 0u, mB1 = gb0 + ld_row + 32 < G ? ~0u : 0u;

    uint32_t acc[4][4], acc_hi[WEIGHTED ? 4 : 1][WEIGHTED ? 4 : 1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = 0;
            if (WEIGHTED) acc_hi[i][j] = 0;
        }

    uint4 va0, va1, vb0 = make_uint4(0, 0, 0, 0), vb1 = vb0;
    auto masked = [](uint4 v, uint32_t m) { return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m); };
    auto fetch = [&](uint64_t w) {
        va0 = rowA0[w >> 2];
        va1 = rowA1[w >> 2];
        if (!diag) {
            vb0 = rowB0[w >> 2];
            vb1 = rowB1[w >> 2];
        }
    };
    auto put4 = [&](uint32_t *s, uint32_t row, const uint4 &v) {
        s[(ld_k + 0) * PAIR_LD + row] = v.x;
        s[(ld_k + 1) * PAIR_LD + row] = v.y;
        s[(ld_k + 2) * PAIR_LD + row] = v.z;
        s[(ld_k + 3) * PAIR_LD + row] = v.w;
    };
    auto stage = [&](int buf) {
        put4(sA[buf], ld_row, masked(va0, mA0));
        put4(sA[buf], ld_row + 32, masked(va1, mA1));
        if (!diag) {
            put4(sB[buf], ld_row, masked(vb0, mB0));
            put4(sB[buf], ld_row + 32, masked(vb1, mB1));
        }
    };

    if (w_begin < w_end) {
        fetch(w_begin);
        stage(0);
    }
    __syncthreads();
    int buf = 0;
    for (uint64_t w = w_begin; w < w_end; w += PAIR_KS, buf ^= 1) {
        const bool more = w + PAIR_KS < w_end;
        if (more) fetch(w + PAIR_KS);
        if (WEIGHTED) {
            // planes of this step: [word][plane]; the previous step's readers passed the barrier
            for (uint32_t idx = t; idx < PAIR_KS * n_planes; idx += 256) {
                const uint32_t p = idx / PAIR_KS, k = idx % PAIR_KS;
                sW[k * 32 + p] = wplanes[(uint64_t)p * row_words + w + k];
            }
            __syncthreads();
        }
        const uint32_t *a_s = sA[buf] + ra;
        const uint32_t *b_s = (diag ? sA[buf] : sB[buf]) + rb;
#pragma unroll 4
        for (int k = 0; k < PAIR_KS; ++k) {
            const uint4 a4 = *reinterpret_cast<const uint4 *>(a_s + k * PAIR_LD);
            const uint4 b4 = *reinterpret_cast<const uint4 *>(b_s + k * PAIR_LD);
            const uint32_t a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
            if (!WEIGHTED) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = popc_acc(a[i] & b[j], acc[i][j]);
            } else {
                uint32_t x[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[i][j] = a[i] & b[j];
                const uint32_t n_lo = n_planes < 16 ? n_planes : 16;
                for (uint32_t p = 0; p < n_lo; ++p) {
                    const uint32_t wp = sW[k * 32 + p];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = lshl_add(__popc(x[i][j] & wp), p, acc[i][j]);
                }
                for (uint32_t p = 16; p < n_planes; ++p) {
                    const uint32_t wp = sW[k * 32 + p];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc_hi[i][j] = lshl_add(__popc(x[i][j] & wp), p - 16, acc_hi[i][j]);
                }
            }
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
    }

    // partial[(chunk * n_pairs + pair) * 4096 + e * 256 + t], e = 4 i + j
    unsigned long long *out = partial + ((uint64_t)blockIdx.y * n_pairs + pair) * (PAIR_T * PAIR_T) + t;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned long long v = acc[i][j];
            if (WEIGHTED) v += (unsigned long long)acc_hi[i][j] << 16;
            out[(i * 4 + j) * 256] = v;
        }
}

// sum the chunks of one tile pair (one of its 16 micro-tile slots per workgroup) and write
// inter[a][b] (and inter[b][a] for off-diagonal tiles)
__global__ __launch_bounds__(256) void k_pair_reduce(const unsigned long long *__restrict__ partial,
                                                     uint32_t n_chunks, uint32_t G, uint32_t n_side,
                                                     unsigned long long *__restrict__ inter) {
    const uint32_t ti = blockIdx.x / n_side, tj = blockIdx.x % n_side;
    if (ti > tj) return;
    const uint32_t pair = ti * n_side - ti * (ti + 1) / 2 + tj;
    const uint32_t n_pairs = n_side * (n_side + 1) / 2;
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63u, e = blockIdx.y;
    const uint32_t ra = (wave >> 1) * 32u + (lane >> 3) * 4u, rb = (wave & 1u) * 32u + (lane & 7u) * 4u;
    const unsigned long long *src = partial + (uint64_t)pair * (PAIR_T * PAIR_T) + e * 256 + t;
    const uint64_t stride = (uint64_t)n_pairs * (PAIR_T * PAIR_T);
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
    const uint32_t ga = ti * PAIR_T + ra + (e >> 2), gb = tj * PAIR_T + rb + (e & 3);
    if (ga < G && gb < G) {
        inter[(uint64_t)ga * G + gb] = s;
        if (ti != tj) inter[(uint64_t)gb * G + ga] = s;
    }
}

// K6: one wave per (group, block): out[g][blk * 32 + b] bit l = bit b of M[g][blk][l]
__global__ __launch_bounds__(256) void k_presence_plain(const uint32_t *__restrict__ M, uint32_t n_blocks,
                                                        uint32_t G, unsigned long long *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wid = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (uint64_t)G * n_blocks) return;
    const uint32_t m = M[wid * BLOCK_WORDS + lane];
    unsigned long long mine = 0;
#pragma unroll
    for (uint32_t b = 0; b < 32; ++b) {
        const unsigned long long v = __ballot((m >> b) & 1u);
        if (lane == b) mine = v;
    }
    if (lane < 32) out[wid * 32 + lane] = mine;
}

int launch_pair_intersections(pnx_ctx *ctx) {
    const uint32_t G = ctx->n_groups, NB = ctx->n_blocks;
    int rc;
    if ((rc = ensure(ctx, ctx->d_inter, ((size_t)G * G ? (size_t)G * G : 1) * sizeof(uint64_t)))) return rc;
    if (G == 0) return PNX_OK;
    if (NB == 0) {
        PNX_HIP(ctx, hipMemsetAsync(ctx->d_inter.p, 0, (size_t)G * G * sizeof(uint64_t), ctx->stream));
        return PNX_OK;
    }
    if (ctx->pairs_variant == 1) return launch_pair_intersections_mfma(ctx);
    const uint64_t row_words = (uint64_t)NB * BLOCK_WORDS;
    const uint32_t n_side = (G + PAIR_T - 1) / PAIR_T;
    if ((uint64_t)n_side * n_side > 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "too many groups for the pair kernel");
    const uint64_t n_pairs = (uint64_t)n_side * (n_side + 1) / 2;
    if (ctx->weighted && (rc = ensure_weight_planes(ctx))) return rc;
    // chunks: enough workgroups to fill the chip, every chunk a multiple of the staging step
    uint64_t n_chunks = (4096 + n_pairs - 1) / n_pairs;
    const uint64_t max_chunks = (row_words + 255) / 256;
    if (n_chunks > max_chunks) n_chunks = max_chunks;
    if (n_chunks < 1) n_chunks = 1;
    uint64_t chunk_words = ((row_words + n_chunks - 1) / n_chunks + PAIR_KS - 1) / PAIR_KS * PAIR_KS;
    if (ctx->weighted && chunk_words > PAIR_WCHUNK_MAX) chunk_words = PAIR_WCHUNK_MAX;
    n_chunks = (row_words + chunk_words - 1) / chunk_words;
    if (n_chunks > 65535) return ctx->fail(PNX_ELIMIT, "pair kernel: %llu word chunks exceed the grid", (unsigned long long)n_chunks);
    const size_t part_bytes = (size_t)n_chunks * n_pairs * PAIR_T * PAIR_T * sizeof(uint64_t);
    if ((rc = ensure(ctx, ctx->d_pair_partial, part_bytes))) return rc;
    prof_begin(ctx, PNX_K_PAIRS);
    const dim3 grid(n_side * n_side, (unsigned)n_chunks);
    if (ctx->weighted)
        hipLaunchKernelGGL(k_pair_intersect<true>, grid, dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_M.p,
                           row_words, G, n_side, (uint32_t)chunk_words, (const uint32_t *)ctx->d_wplanes.p,
                           ctx->n_wplanes, (unsigned long long *)ctx->d_pair_partial.p);
    else
        hipLaunchKernelGGL(k_pair_intersect<false>, grid, dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_M.p,
                           row_words, G, n_side, (uint32_t)chunk_words, (const uint32_t *)nullptr, 0u,
                           (unsigned long long *)ctx->d_pair_partial.p);
    hipLaunchKernelGGL(k_pair_reduce, dim3(n_side * n_side, 16), dim3(256), 0, ctx->stream,
                       (const unsigned long long *)ctx->d_pair_partial.p, (uint32_t)n_chunks, G, n_side,
                       (unsigned long long *)ctx->d_inter.p);
    prof_end(ctx);
    PNX_HIP(ctx, hipGetLastError());
    return PNX_OK;
}

// ------------------------------------------------------------------------------------------
// K8: per-group visit counts of an item range -- AbacusByGroup.v (abacus.rs:901-986) laid out
// densely.  v[k] of the reference is the number of steps that the paths of group c[k] take on the
// item (the "pointer game" of compute_column_values adds one per step to the slot of the step's
// group).  One thread per step; the path of a step comes from a search in path_off per workgroup
// and a short forward walk per thread.  Output-bound command (`table`): plain global atomics.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_visit_counts(const uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off,
                                                      uint32_t n_paths, const uint32_t *__restrict__ path_group,
                                                      const uint8_t *__restrict__ exclude,
                                                      const uint32_t *__restrict__ old_of_new /* internal -> caller id, or nullptr */,
                                                      uint32_t lo, uint32_t hi, uint32_t *__restrict__ out) {
    const uint64_t S = path_off[n_paths];
    const uint64_t base = (uint64_t)blockIdx.x * 1024;
    __shared__ uint32_t p0;
    if (threadIdx.x == 0) {  // last path whose first step is <= base
        uint32_t a = 0, b = n_paths;
        while (b - a > 1) {
            const uint32_t m = a + (b - a) / 2;
            if (path_off[m] <= base) a = m; else b = m;
        }
        p0 = a;
    }
    __syncthreads();
    uint32_t p = p0;
    const uint32_t width = hi - lo;
    for (int r = 0; r < 4; ++r) {
        const uint64_t s = base + (uint64_t)r * 256 + threadIdx.x;
        if (s >= S) return;
        while (p + 1 < n_paths && path_off[p + 1] <= s) ++p;  // empty paths are stepped over
        const uint32_t g = path_group[p];
        if (g == 0xFFFFFFFFu) continue;  // the path is not in the visiting order
        const uint32_t id = items[s];  // internal numbering, like the exclusion flags
        if (exclude && exclude[id]) continue;
        const uint32_t cid = old_of_new ? old_of_new[id] : id;  // the slice [lo, hi) is in the caller's ids
        if (cid < lo || cid >= hi) continue;
        atomicAdd(&out[(uint64_t)g * width + (cid - lo)], 1u);
    }
}

int launch_visit_counts(pnx_ctx *ctx, uint32_t lo, uint32_t hi, DevBuf &d_path_group, DevBuf &out) {
    const uint64_t cells = (uint64_t)ctx->n_groups * (hi - lo);
    int rc;
    if ((rc = ensure(ctx, out, (cells ? cells : 1) * sizeof(uint32_t)))) return rc;
    if (!cells) return PNX_OK;
    PNX_HIP(ctx, hipMemsetAsync(out.p, 0, cells * sizeof(uint32_t), ctx->stream));
    const uint64_t blocks = (ctx->n_steps + 1023) / 1024;
    if (blocks > 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "visit counts: too many steps for one launch");
    if (blocks)
        hipLaunchKernelGGL(k_visit_counts, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_items.p,
                           (const uint64_t *)ctx->d_path_off.p, ctx->n_paths, (const uint32_t *)d_path_group.p,
                           ctx->have_exclude ? (const uint8_t *)ctx->d_exclude.p : (const uint8_t *)nullptr,
                           ctx->relabeled ? (const uint32_t *)ctx->d_old_of_new.p : (const uint32_t *)nullptr, lo, hi,
                           (uint32_t *)out.p);
    PNX_HIP(ctx, hipGetLastError());
    return PNX_OK;
}

int launch_presence_plain(pnx_ctx *ctx, DevBuf &out) {
    const uint32_t G = ctx->n_groups, NB = ctx->n_blocks;
    const size_t words = (size_t)G * NB * 32;
    int rc;
    if ((rc = ensure(ctx, out, (words ? words : 1) * sizeof(uint64_t)))) return rc;
    if (!words) return PNX_OK;
    const uint64_t waves = (uint64_t)G * NB;
    if ((waves + 3) / 4 > 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "presence matrix too large for one export");
    hipLaunchKernelGGL(k_presence_plain, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, ctx->stream,
                       (const uint32_t *)ctx->d_M.p, NB, G, (unsigned long long *)out.p);
    PNX_HIP(ctx, hipGetLastError());
    return PNX_OK;
}

}  // namespace pnx

namespace pnx {
// pnx_preload (see kernels_gfa.hip): touching one kernel loads the code object of this translation unit
void preload_pairs(unsigned what) {
    hipFuncAttributes a;
    auto touch = [&a](const void *k) { (void)hipFuncGetAttributes(&a, k); };
    if (what & PNX_PRELOAD_TABLES) {
        touch((const void *)k_presence_plain);
        touch((const void *)k_visit_counts);
    }
}
}  // namespace pnx
