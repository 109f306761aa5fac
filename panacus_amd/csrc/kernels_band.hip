// kernels_band.hip -- the one-shot coverage pass: steps -> coverage vector + histogram in ONE read of the ItemTable (gfx950).
//
// AbacusByTotal::coverage (src/graph_broker/abacus.rs:719-744) sweeps the ItemTable once per abacus, and nothing in the
// reference ever sweeps the same table twice with the same grouping (graph_broker.rs:389-432): the unit of work is ONE
// sweep.  The path rows of kernels_rows.hip pay for a derived table (steps read once, rows written, rows read) that only
// a second sweep gets anything out of.  This file is the route for the first sweep of a graph whose paths run through the
// ids in order (pangenome paths of graphs with sorted ids; ascending or descending):
//
//   k_band_index   per (path of the visiting order, band edge): the step position where the path crosses the edge of a
//                  BAND of BT item tiles -- an interpolation search with sqrt-steps, a handful of dependent probes.
//   k_band_cover   one WORKGROUP owns one band (BT tiles = BT waves, wave w keeps the bit-sliced counters of tile w).
//                  The visiting order is taken CW entries at a time: wave w streams the whole band segment of entry
//                  k0 + w -- contiguous steps, 16-byte non-temporal loads, BAND_D of them in flight per lane, the next
//                  segment's first loads issued before the current one is folded -- and ORs presence bits into the
//                  entry's band bitmap in LDS (ds_or_b32: visiting an item twice is idempotent, which is the
//                  reference's last[] array); after ONE barrier per CW entries every wave reads its tile's slice of the
//                  CW bitmaps in visiting order and folds finished groups into its counters (tile_counters.hpp).  The
//                  tail of a tile -- coverage vector, histogram bins -- is the one the rows kernel has.
//
// Exactness does not rest on the search: the segments of a path partition its steps (edge positions are checked to be
// monotone), every step is checked against the band it was dealt to, and a single step outside -- a path that is not
// sorted by id -- raises flags[5]: the pass is void and the host runs it again over path rows, which serve any path
// (pnx_api.hip: settle_oldest).  Steps ORed from positions of the SAME path beyond a segment's ends (16-byte alignment)
// are never foreign facts: positions outside the segment are masked.
#include <hip/hip_runtime.h>

#include "pnx_context.hpp"
#include "tile_counters.hpp"

namespace pnx {

constexpr int BAND_D = 4;              // 16-byte loads in flight per lane
constexpr uint64_t BAND_DESC = 1ull << 63;  // index entry: the path runs through the ids downwards

// first j in [0, len] with key(j) >= X, key = id (ascending path) or ~id (descending); keys are non-decreasing on a
// sorted path -- on any other the result is some position in [0, len] and the coverage kernel finds out
template <bool DESC>
__device__ static inline uint64_t band_edge_search(const uint32_t *__restrict__ it, uint64_t len, uint32_t X, uint32_t ka, uint32_t kz) {
    auto key = [&](uint64_t j) { const uint32_t v = it[j]; return DESC ? ~v : v; };
    if (ka >= X) return 0;
    if (kz < X) return len;
    uint64_t lo = 0, hi = len - 1;  // key(lo) < X <= key(hi)
    uint32_t klo = ka, khi = kz;
    for (int iter = 0; iter < 4 && hi - lo > 32; ++iter) {
        const uint64_t range = hi - lo;
        const double f = (double)(X - klo) / (double)(khi - klo);
        uint64_t g = lo + (uint64_t)(f * (double)range);
        g = g <= lo ? lo + 1 : (g >= hi ? hi - 1 : g);
        uint64_t step = (uint64_t)sqrt((double)range);
        step = step < 8 ? 8 : step;
        const uint32_t kg = key(g);
        if (kg >= X) {  // walk down until a key below X
            hi = g;
            khi = kg;
            for (int s = 0; s < 8 && hi - lo > step; ++s, step += step >> 1) {
                const uint64_t c = hi - step;
                const uint32_t kc = key(c);
                if (kc >= X) {
                    hi = c;
                    khi = kc;
                } else {
                    lo = c;
                    klo = kc;
                    break;
                }
            }
        } else {  // walk up
            lo = g;
            klo = kg;
            for (int s = 0; s < 8 && hi - lo > step; ++s, step += step >> 1) {
                const uint64_t c = lo + step;
                const uint32_t kc = key(c);
                if (kc < X) {
                    lo = c;
                    klo = kc;
                } else {
                    hi = c;
                    khi = kc;
                    break;
                }
            }
        }
    }
    while (hi - lo > 1) {
        const uint64_t m = lo + ((hi - lo) >> 1);
        if (key(m) >= X) hi = m;
        else lo = m;
    }
    return hi;
}

// bidx[e * n_ordered + k] = absolute step position where entry k's path crosses band edge e (id e * band_items), e = 0 ..
// n_bands, | BAND_DESC for a descending path: band b of an ascending path is [bidx[b], bidx[b+1]), of a descending one
// [bidx[b+1], bidx[b]).  (Also clears the pass's counter block: a memset in front is one more kernel in the chain.)
__global__ __launch_bounds__(256) void k_band_index(const uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off,
                                                    const uint32_t *__restrict__ ord_path, uint32_t n_ordered, uint32_t n_bands,
                                                    uint32_t band_items, unsigned long long *__restrict__ bidx,
                                                    uint4 *__restrict__ block16, uint32_t n_block16) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t q = tid; q < n_block16; q += (uint64_t)gridDim.x * blockDim.x) block16[q] = make_uint4(0, 0, 0, 0);
    const uint64_t total = (uint64_t)(n_bands + 1) * n_ordered;
    if (tid >= total) return;
    const uint32_t k = (uint32_t)(tid % n_ordered), e = (uint32_t)(tid / n_ordered);
    const uint32_t p = ord_path[k];
    const uint64_t ps = path_off[p], pe = path_off[p + 1], len = pe - ps;
    if (len == 0) {
        bidx[tid] = ps;
        return;
    }
    const uint32_t a = items[ps], z = items[pe - 1];
    const bool desc = a > z;
    uint64_t j;
    if (e == 0) j = desc ? len : 0;
    else if (e == n_bands) j = desc ? 0 : len;
    else {
        const uint32_t x = e * band_items;  // 1 <= x <= n_items: inner edges only
        j = desc ? band_edge_search<true>(items + ps, len, ~(x - 1u), ~a, ~z) : band_edge_search<false>(items + ps, len, x, a, z);
    }
    bidx[tid] = (ps + j) | (desc ? BAND_DESC : 0ull);
}

// The coverage kernel.  flags[5] |= 1 when a step was found outside the band it was dealt to (or an index entry is
// inconsistent): the result of the pass is void.
template <int NPL, int CW, bool WRITE_M>
__global__ __launch_bounds__(CW * 64) void k_band_cover(const uint32_t *__restrict__ items, const unsigned long long *__restrict__ bidx,
                                                        const uint32_t *__restrict__ ord_group, uint32_t n_ordered,
                                                        const uint8_t *__restrict__ exclude, uint32_t n_items, uint32_t n_tiles,
                                                        uint32_t *__restrict__ M, uint64_t row_words, uint32_t *__restrict__ countable,
                                                        RowHist hs, uint32_t *__restrict__ flags) {
    constexpr int BT = CW;  // tiles per band = waves per workgroup
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __shared__ uint32_t bm[2][CW][BT * 64];  // two generations of CW band bitmaps (one per entry of a batch)
    extern __shared__ unsigned long long sh_hist[];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t band = blockIdx.x;
    const uint32_t tile = band * BT + wave;
    const bool active = tile < n_tiles;  // the last band may hold fewer tiles; its spare waves still stream segments
    for (uint32_t i = threadIdx.x; i < 2 * CW * BT * 64; i += CW * 64) (&bm[0][0][0])[i] = 0;
    if (hs.rep)
        for (uint32_t b = threadIdx.x; b <= hs.n_groups; b += CW * 64) sh_hist[b] = 0;
    __syncthreads();
    const uint32_t excl = active ? tile_exclusion_word(exclude, tile, lane, n_items) : 0u;
    // ids of this band: [lo_id, lo_id + width); item 0 is no item, ids beyond n_items are none either
    const uint32_t lo_id = band ? band * (uint32_t)(BT * BLOCK_ITEMS) : 1u;
    const uint64_t hi64 = (uint64_t)(band + 1) * (BT * BLOCK_ITEMS);
    const uint32_t hi_id = hi64 > (uint64_t)n_items + 1 ? n_items + 1u : (uint32_t)hi64;
    const uint32_t width = hi_id - lo_id;

    TileCounters<NPL> tc;
    uint32_t acc = 0, cur_g = NONE;
    bool bad = false;
    auto flush = [&](uint32_t g) {
        const uint32_t x = acc & ~excl;
        acc = 0;
        if (WRITE_M && active) M[(uint64_t)g * row_words + (uint64_t)tile * BLOCK_WORDS + lane] = x;
        tc.add_group(x);
    };

    // ---- 64 entries of the visiting order in the lanes: their segments on this band (issue side) ----
    uint64_t w_lo = 0;
    uint32_t w_len = 0, swin = NONE;  // swin: which window of 64 entries
    const unsigned long long *e0 = bidx + (uint64_t)band * n_ordered, *e1 = e0 + n_ordered;
    auto load_swin = [&](uint32_t win) {
        swin = win;
        const uint32_t k = win * 64u + lane;
        uint64_t a = 0, b = 0;
        if (k < n_ordered) {
            a = e0[k];
            b = e1[k];
        }
        const bool desc = (a & BAND_DESC) != 0;
        if (((a ^ b) & BAND_DESC) != 0) bad = true;
        a &= ~BAND_DESC;
        b &= ~BAND_DESC;
        uint64_t lo = desc ? b : a, hi = desc ? a : b;
        if (hi < lo) {  // the searches of a path that is not sorted
            bad = true;
            hi = lo;
        }
        uint64_t len = hi - lo;
        if (len >= (1ull << 31)) {
            bad = true;
            len = 0;
        }
        w_lo = lo;
        w_len = (uint32_t)len;
    };
    // ---- ... and their groups (fold side: the issue side may already be one window ahead) ----
    uint32_t f_g = NONE, fwin = NONE;

    struct Seg {
        const uint32_t *base;  // 16-byte aligned, <= first step
        uint32_t head, nal;    // steps of the first load before the segment; head + length
    };
    auto seg_of = [&](uint32_t batch) {
        Seg s{items, 0u, 0u};
        const uint32_t k = batch * CW + wave;
        if (k < n_ordered) {
            if ((k >> 6) != swin) load_swin(k >> 6);
            const uint32_t l = k & 63u;
            const uint32_t lo_l = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w_lo, l);
            const uint32_t lo_h = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w_lo >> 32), l);
            const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)w_len, l);
            const uint64_t lo = ((uint64_t)lo_h << 32) | lo_l;
            if (len) {
                s.base = items + (lo & ~3ull);
                s.head = (uint32_t)(lo & 3ull);
                s.nal = s.head + len;
            }
        }
        return s;
    };

    const uint32_t n_batches = (n_ordered + CW - 1) / CW;
    u32x4 buf[BAND_D];
    auto issue = [&](const Seg &s, uint32_t r0, int u) {
        const uint32_t r = r0 + (uint32_t)u * 256u + lane * 4u;
        u32x4 v = u32x4{0, 0, 0, 0};
        if (r < s.nal) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s.base + r));
        return v;
    };
    auto process = [&](const u32x4 &v, const Seg &s, uint32_t r0, int u, uint32_t *map) {
        const uint32_t q = r0 + (uint32_t)u * 256u + lane * 4u - s.head;  // position in the segment (wraps before it)
        const uint32_t len = s.nal - s.head;
        const uint32_t ids[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t id = ids[e];
            const bool valid = q + (uint32_t)e < len;
            const bool inb = id - lo_id < width;
            if (valid && inb) atomicOr(&map[((id >> 11) & (uint32_t)(BT - 1)) * 64u + (id & 63u)], 1u << ((id >> 6) & 31u));
            bad |= valid && !inb;
        }
    };
    auto fold = [&](uint32_t batch) {
        const uint32_t k0 = batch * CW;
        if ((k0 >> 6) != fwin) {
            fwin = k0 >> 6;
            const uint32_t k = fwin * 64u + lane;
            f_g = k < n_ordered ? ord_group[k] : NONE;
        }
        uint32_t x[CW];
#pragma unroll
        for (int i = 0; i < CW; ++i) {
            uint32_t *w = &bm[batch & 1u][i][wave * 64u + lane];
            x[i] = *w;
            *w = 0;
        }
#pragma unroll
        for (int i = 0; i < CW; ++i) {
            if (k0 + (uint32_t)i < n_ordered) {
                const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)f_g, (k0 + (uint32_t)i) & 63u);
                if (g != cur_g) {
                    if (cur_g != NONE) flush(cur_g);
                    cur_g = g;
                }
                acc |= x[i];
            }
        }
    };

    if (n_batches) {
        Seg cur = seg_of(0);
        uint32_t c_r0 = 0, c_batch = 0;
#pragma unroll
        for (int u = 0; u < BAND_D; ++u) buf[u] = issue(cur, 0, u);
        while (c_batch < n_batches) {
            Seg nxt = cur;
            uint32_t n_r0 = c_r0 + 256u * BAND_D, n_batch = c_batch;
            if (n_r0 >= cur.nal) {  // the segment ends with this group of loads
                n_batch = c_batch + 1;
                n_r0 = 0;
                nxt = n_batch < n_batches ? seg_of(n_batch) : Seg{items, 0u, 0u};
            }
            uint32_t *map = &bm[c_batch & 1u][wave][0];
#pragma unroll
            for (int u = 0; u < BAND_D; ++u) {
                const u32x4 v = buf[u];
                buf[u] = issue(nxt, n_r0, u);
                process(v, cur, c_r0, u, map);
            }
            if (n_batch != c_batch) {
                __syncthreads();
                fold(c_batch);
            }
            cur = nxt;
            c_r0 = n_r0;
            c_batch = n_batch;
        }
        if (cur_g != NONE) flush(cur_g);
    }
    tc.settle();
    if (__ballot(bad) && lane == 0) atomicOr(flags + 5, 1u);
    if (active) tile_tail<NPL>(tc.cnt, tile, lane, n_items, countable, hs, sh_hist, 0u, 32u, 0xFFFFFFFFu);
    if (hs.rep) {
        __syncthreads();
        hist_bins_flush(hs, sh_hist, CW * 64);
    }
}

// The shapes the band route is worth it for: enough bands to fill the chip, segments long enough to stream, an index
// of reasonable size.  (Anything else -- and any graph whose paths turn out not to be sorted -- takes the path rows.)
bool band_route_fits(const pnx_ctx *ctx, uint32_t n_entries) {
    constexpr uint32_t BT = 4;
    if (!n_entries || !ctx->n_steps || !ctx->n_paths) return false;
    const uint64_t n_bands = (ctx->n_blocks + BT - 1) / BT;
    if (n_bands < 2ull * (uint64_t)ctx->prop.multiProcessorCount) return false;
    const uint64_t cells = n_bands * ctx->n_paths;
    if (ctx->n_steps / cells < 512) return false;
    if ((n_bands + 1) * n_entries * 8 > (256ull << 20)) return false;
    return true;
}

template <int NPL>
static void launch_band_cover_t(pnx_ctx *ctx, bool write_m, uint32_t n_bands) {
    Ticket *tk = ctx->cur;
    const RowHist hs{tk->hist_fused ? (unsigned long long *)tk->d_hist_rep : nullptr,
                     ctx->weighted ? (const uint32_t *)ctx->d_weights.p : (const uint32_t *)nullptr, ctx->n_groups};
    const size_t lds_hist = tk->hist_fused ? ((size_t)ctx->n_groups + 1) * sizeof(unsigned long long) : 0;
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(n_bands), dim3(4 * 64), lds_hist, ctx->s_main, (const uint32_t *)ctx->d_items.p,
                           (const unsigned long long *)tk->d_tile_idx_own.p, (const uint32_t *)ctx->d_ord_group.p, ctx->n_ordered,
                           ctx->have_exclude ? (const uint8_t *)ctx->d_exclude.p : (const uint8_t *)nullptr, ctx->n_items, ctx->n_blocks,
                           (uint32_t *)ctx->d_M.p, (uint64_t)ctx->n_blocks * BLOCK_WORDS, (uint32_t *)tk->d_countable.p, hs, tk->d_flags);
    };
    if (write_m) go(k_band_cover<NPL, 4, true>);
    else go(k_band_cover<NPL, 4, false>);
}

// phases 1 + 2 of a one-shot pass over the steps (the histogram phase is shared: launch_cover_pass)
int launch_band_phases(pnx_ctx *ctx, bool write_m) {
    constexpr uint32_t BT = 4;
    Ticket *tk = ctx->cur;
    int rc;
    const uint32_t n_bands = (ctx->n_blocks + BT - 1) / BT;
    const uint64_t cells = (uint64_t)(n_bands + 1) * ctx->n_ordered;
    if ((rc = ensure(ctx, tk->d_tile_idx_own, cells * 8))) return rc;
    const bool phased = ctx->s_pre != ctx->s_main;
    prof_begin(ctx, PNX_K_INDEX, ctx->s_pre);
    hipLaunchKernelGGL(k_band_index, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->s_pre, (const uint32_t *)ctx->d_items.p,
                       (const uint64_t *)ctx->d_path_off.p, (const uint32_t *)ctx->d_ord_path.p, ctx->n_ordered, n_bands,
                       BT * BLOCK_ITEMS, (unsigned long long *)tk->d_tile_idx_own.p, (uint4 *)tk->d_block.p, (uint32_t)(tk->block_bytes / 16));
    prof_end(ctx);
    PNX_HIP(ctx, hipGetLastError());
    if (phased) {
        PNX_HIP(ctx, hipEventRecord(tk->ev_pre, ctx->s_pre));
        PNX_HIP(ctx, hipStreamWaitEvent(ctx->s_main, tk->ev_pre, 0));
    }
    uint32_t bits = 1;  // planes needed to count up to n_groups inclusive
    while (bits < 32 && (ctx->n_groups >> bits) != 0) ++bits;
    prof_begin(ctx, PNX_K_COVER, ctx->s_main);
    if (bits <= 12) launch_band_cover_t<12>(ctx, write_m, n_bands);
    else if (bits <= 24) launch_band_cover_t<24>(ctx, write_m, n_bands);
    else {
        prof_end(ctx);
        return ctx->fail(PNX_ELIMIT, "more than 2^24-1 groups are not supported (got %u)", ctx->n_groups);
    }
    prof_end(ctx);
    PNX_HIP(ctx, hipGetLastError());
    return PNX_OK;
}

}  // namespace pnx
