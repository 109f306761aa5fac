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

constexpr int BAND_CW = 4;                  // waves per workgroup = item tiles per band
constexpr uint64_t BAND_UNSORTED = 1ull << 62;  // index entry: a probe of the search met steps out of order -- the path is not sorted
constexpr uint64_t BAND_DESC = 1ull << 63;  // index entry: the path runs through the ids downwards

// first j in [0, len] with key(j) >= X, key = id (ascending path) or ~id (descending); keys are non-decreasing on a
// sorted path -- on any other the result is some position in [0, len] and the coverage kernel finds out.
// A probe is one aligned 64-byte sector = 16 steps (four 16-byte loads, one latency): either the sector holds the
// crossing, or its nearest step becomes one of the two points of the next secant guess (the path's ends at first: ids
// along a pangenome path are close to evenly spread, so the first guess is off by a few thousand steps of millions, the
// second by tens, the third lands in the sector) and tightens the bracket; after 8 probes the guess is the midpoint.
// The chain of dependent reads is what this kernel costs: ~4 probes here against ~14 single-step probes of a plain
// interpolation + binary search (0.058 -> 0.03 ms on 10 M items x 256 paths).
template <bool DESC>
__device__ static inline uint64_t band_edge_search(const uint32_t *__restrict__ items, uint64_t ps, uint64_t len, uint32_t X, uint32_t ka,
                                                   uint32_t kz, bool &unsorted) {
    if (ka >= X) return 0;
    if (kz < X) return len;
    uint64_t lo = 0, hi = len - 1;  // key(lo) < X <= key(hi)
    double pa = 0.0, va = (double)ka, pb = (double)(len - 1), vb = (double)kz;  // the two points of the secant
    for (int iter = 0; hi - lo > 1; ++iter) {
        uint64_t g = lo + ((hi - lo) >> 1);
        if (iter < 8 && vb != va) {
            const double t = pb + ((double)X - vb) * (pb - pa) / (vb - va);
            g = t <= (double)(lo + 1) ? lo + 1 : (t >= (double)(hi - 1) ? hi - 1 : (uint64_t)t);
        }
        const uint64_t a0 = (ps + g) & ~15ull;  // the sector of step g (absolute index of its first step)
        const uint4 *src = reinterpret_cast<const uint4 *>(items + a0);
        const uint4 w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
        const uint32_t w[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
        // steps of the sector that belong to the path and lie inside the bracket: [i0, i1]
        const uint64_t first = ps + lo + 1, last = ps + hi - 1;  // absolute; first <= ps + g <= last
        const uint32_t i0 = first > a0 ? (uint32_t)(first - a0) : 0u, i1 = last - a0 < 15 ? (uint32_t)(last - a0) : 15u;
        uint32_t below = 0, k_first = 0, k_last = 0, prev = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i) {
            const uint32_t key = DESC ? ~w[i] : w[i];
            if (i == i0) k_first = key;
            if (i == i1) k_last = key;
            below += (i >= i0 && i <= i1 && key < X) ? 1u : 0u;
            unsorted |= i > i0 && i <= i1 && key < prev;  // the sector in hand says the path is not sorted: no need to read it all to find out
            prev = key;
        }
        const uint32_t n = i1 - i0 + 1;
        if (below != 0 && below != n) return a0 + i0 + below - ps;  // the crossing lies in the sector
        pa = pb;
        va = vb;
        if (below == 0) {  // all at or above X
            hi = a0 + i0 - ps;
            pb = (double)hi;
            vb = (double)k_first;
        } else {
            lo = a0 + i1 - ps;
            pb = (double)lo;
            vb = (double)k_last;
        }
    }
    return hi;
}

// bidx[k * (n_bands + 1) + e] = absolute step position where entry k's path crosses band edge e (id e * band_items), e = 0 ..
// n_bands, | BAND_DESC for a descending path: band b of an ascending path is [bidx[k][b], bidx[k][b+1]), of a descending one
// [bidx[k][b+1], bidx[k][b]).  The lanes of a wave take consecutive edges of ONE path: their probes stay within a few
// hundred KB of each other (one or two translation entries per round of probes instead of 64).  (Also clears the pass's counter block: a memset in front is one more kernel in the chain.)
__global__ __launch_bounds__(256) void k_band_index(const uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off,
                                                    const uint32_t *__restrict__ ord_path, uint32_t n_ordered, uint32_t n_bands,
                                                    uint32_t band_items, unsigned long long *__restrict__ bidx,
                                                    uint4 *__restrict__ block16, uint32_t n_block16) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t q = tid; q < n_block16; q += (uint64_t)gridDim.x * blockDim.x) block16[q] = make_uint4(0, 0, 0, 0);
    const uint64_t total = (uint64_t)(n_bands + 1) * n_ordered;
    if (tid >= total) return;
    const uint32_t e = (uint32_t)(tid % (n_bands + 1)), k = (uint32_t)(tid / (n_bands + 1));
    const uint32_t p = ord_path[k];
    const uint64_t ps = path_off[p], pe = path_off[p + 1], len = pe - ps;
    if (len == 0) {
        bidx[tid] = ps;
        return;
    }
    const uint32_t a = items[ps], z = items[pe - 1];
    const bool desc = a > z;
    uint64_t j;
    bool unsorted = false;
    if (e == 0) j = desc ? len : 0;
    else if (e == n_bands) j = desc ? 0 : len;
    else {
        const uint32_t x = e * band_items;  // 1 <= x <= n_items: inner edges only
        j = desc ? band_edge_search<true>(items, ps, len, ~(x - 1u), ~a, ~z, unsorted) : band_edge_search<false>(items, ps, len, x, a, z, unsorted);
    }
    bidx[tid] = (ps + j) | (desc ? BAND_DESC : 0ull) | (unsorted ? BAND_UNSORTED : 0ull);
}

// The coverage kernel.  flags[5] |= 1 when a step was found outside the band it was dealt to (or an index entry is
// inconsistent): the result of the pass is void.
template <int NPL, int CW, bool WRITE_M, int BAND_D>
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
    bool bad = false, dead = false;
    auto flush = [&](uint32_t g) {
        const uint32_t x = acc & ~excl;
        acc = 0;
        if (WRITE_M && active) M[(uint64_t)g * row_words + (uint64_t)tile * BLOCK_WORDS + lane] = x;
        tc.add_group(x);
    };

    // ---- 64 entries of the visiting order in the lanes: their segments on this band (issue side) ----
    uint64_t w_lo = 0;
    uint32_t w_len = 0, swin = NONE;  // swin: which window of 64 entries
    uint32_t w_g = NONE;              // ... and their groups: handed to the fold side when it reaches the window
    const uint32_t n_edges = gridDim.x + 1;  // per entry: n_bands + 1 edge positions
    auto load_swin = [&](uint32_t win) {
        swin = win;
        const uint32_t k = win * 64u + lane;
        uint64_t a = 0, b = 0;
        w_g = NONE;
        if (k < n_ordered) {
            const unsigned long long *ek = bidx + (uint64_t)k * n_edges + band;
            a = ek[0];
            b = ek[1];
            w_g = ord_group[k];
        }
        const bool desc = (a & BAND_DESC) != 0;
        if (((a ^ b) & BAND_DESC) != 0) bad = true;
        // the index already knows of a path that is not sorted: the pass is void, and this workgroup stops reading (every wave
        // of the workgroup loads the same window and leaves after the same fold)
        if (__ballot(((a | b) & BAND_UNSORTED) != 0)) bad = dead = true;
        a &= ~(BAND_DESC | BAND_UNSORTED);
        b &= ~(BAND_DESC | BAND_UNSORTED);
        uint64_t lo = desc ? b : a, hi = desc ? a : b;
        if (hi < lo) {  // the searches of a path that is not sorted
            bad = true;
            hi = lo;
        }
        uint64_t len = hi - lo;
        if (len >= (1ull << 29)) {  // (a buffer descriptor holds 2^32 bytes)
            bad = true;
            len = 0;
        }
        w_lo = lo;
        w_len = (uint32_t)len;
    };
    // ---- the groups of the window being folded (the issue side may already be one window ahead; it has loaded this
    // window before the first fold in it, and moves on only after that fold: no load on the fold side, whose wait would
    // drain the next segment's loads in flight) ----
    uint32_t f_g = NONE, fwin = NONE;

    // A segment is read through a buffer descriptor (base = its 16-byte aligned start, size = its bytes): loads beyond its
    // end return zeros without a branch, so every load is issued unconditionally and the compiler counts them exactly
    // (s_waitcnt vmcnt(n) per slot instead of a drain in front of every use).
    struct Seg {
        __amdgpu_buffer_rsrc_t rs;
        const uint32_t *base;
        uint32_t head, nal;  // steps of the first load before the segment; head + length
    };
    auto make_seg = [&](uint64_t lo, uint32_t len) {
        const uint32_t *base = items + (lo & ~3ull);
        const uint32_t head = len ? (uint32_t)(lo & 3ull) : 0u, nal = head + len;
        const uint32_t b_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)base);
        const uint32_t b_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)base >> 32));
        void *bp = (void *)(((uintptr_t)b_hi << 32) | b_lo);
        return Seg{__builtin_amdgcn_make_buffer_rsrc(bp, 0, (int)__builtin_amdgcn_readfirstlane(nal * 4u), 0x00020000), (const uint32_t *)bp, head, nal};
    };
    auto seg_of = [&](uint32_t batch) {
        const uint32_t k = batch * CW + wave;
        if ((k >> 6) != swin) load_swin(k >> 6);  // (every wave: the fold needs the window's groups whether or not this wave has an entry)
        if (k >= n_ordered) return make_seg(0, 0);
        const uint32_t l = k & 63u;
        const uint32_t lo_l = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w_lo, l);
        const uint32_t lo_h = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w_lo >> 32), l);
        const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)w_len, l);
        return make_seg(((uint64_t)lo_h << 32) | lo_l, len);
    };

    const uint32_t n_batches = (n_ordered + CW - 1) / CW;
    auto issue = [&](const Seg &s, uint32_t r0, int u) {
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(s.rs, lane * 16u + (uint32_t)u * 1024u, r0 * 4u, /*nt*/ 2));
    };
    // the steps of one 16-byte load: presence bits into the band bitmap of the entry.  (Under the execution mask, not as an OR
    // of zero: the lanes past the end of a segment would all meet on one word, and same-address LDS atomics serialise.)
    auto process = [&](const u32x4 &v, const Seg &s, uint32_t r0, int u, uint32_t *map) {
        const uint32_t q = r0 + (uint32_t)u * 256u + lane * 4u - s.head;  // position in the segment (wraps before it)
        const uint32_t len = s.nal - s.head;
        const uint32_t ids[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t id = ids[e];
            const bool valid = q + (uint32_t)e < len;
            const bool inb = id - lo_id < width;
            if (valid && inb) atomicOr(&map[(((id >> 11) & (uint32_t)(BT - 1)) << 6) | (id & 63u)], 1u << ((id >> 6) & 31u));
            bad |= valid && !inb;
        }
    };
    auto fold = [&](uint32_t batch) {
        const uint32_t k0 = batch * CW;
        if ((k0 >> 6) != fwin) {  // the first batch of a window: w_g still holds this window's groups
            fwin = k0 >> 6;
            f_g = w_g;
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
        // one group of BAND_D loads per lane: consume `in` slot by slot, the next group's loads going out into `out` as the
        // slots are taken -- BAND_D loads in flight throughout.  (Two register sets that swap roles: with one set refilled
        // in place the compiler parks the new loads elsewhere and copies them back at the loop head, i.e. waits for them.)
        auto group = [&](const u32x4(&in)[BAND_D], u32x4(&out)[BAND_D]) {
            Seg nxt = cur;
            uint32_t n_r0 = c_r0 + 256u * BAND_D, n_batch = c_batch;
            if (n_r0 >= cur.nal) {  // the segment ends with this group of loads
                n_batch = c_batch + 1;
                n_r0 = 0;
                nxt = n_batch < n_batches ? seg_of(n_batch) : make_seg(0, 0);
            }
            uint32_t *map = &bm[c_batch & 1u][wave][0];
#pragma unroll
            for (int u = 0; u < BAND_D; ++u) {
                out[u] = issue(nxt, n_r0, u);
                if (c_r0 + (uint32_t)u * 256u < cur.nal) process(in[u], cur, c_r0, u, map);  // (wave-uniform: the slot holds steps)
                __builtin_amdgcn_sched_barrier(0);  // slot by slot: the other slots' loads stay in flight meanwhile
            }
            if (n_batch != c_batch) {
                __syncthreads();
                fold(c_batch);
            }
            cur = nxt;
            c_r0 = n_r0;
            c_batch = n_batch;
        };
        u32x4 bufA[BAND_D], bufB[BAND_D];
#pragma unroll
        for (int u = 0; u < BAND_D; ++u) bufA[u] = issue(cur, 0, u);
        // (`dead` changes only where a wave loads a window: before the loop -- every wave at once -- and in the group that ends a
        // segment, i.e. the one with the barrier and the fold of that batch: the waves of a workgroup leave after the same barrier)
        while (!dead) {
            group(bufA, bufB);
            if (c_batch >= n_batches || dead) break;
            group(bufB, bufA);
            if (c_batch >= n_batches || dead) break;
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
    constexpr uint32_t BT = BAND_CW;
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
        hipLaunchKernelGGL(kern, dim3(n_bands), dim3(BAND_CW * 64), lds_hist, ctx->s_main, (const uint32_t *)ctx->d_items.p,
                           (const unsigned long long *)tk->d_tile_idx_own.p, (const uint32_t *)ctx->d_ord_group.p, ctx->n_ordered,
                           ctx->have_exclude ? (const uint8_t *)ctx->d_exclude.p : (const uint8_t *)nullptr, ctx->n_items, ctx->n_blocks,
                           (uint32_t *)ctx->d_M.p, (uint64_t)ctx->n_blocks * BLOCK_WORDS, (uint32_t *)tk->d_countable.p, hs, tk->d_flags);
    };
    // 4 waves per band, 2 loads in flight per lane: measured best on 10 M items x 256 and x 1024 paths (0.64 / 2.45 ms; 4 in flight
    // 0.67 / 2.56, 8 in flight 0.72; 8 waves per band 0.70, 2 waves 0.71) -- with every workgroup resident at once the chip holds
    // ~19 waves per CU whatever the register count, and a deeper pipeline only adds loads beyond the ends of the segments
    if (write_m) go(k_band_cover<NPL, BAND_CW, true, 2>);
    else go(k_band_cover<NPL, BAND_CW, false, 2>);
}

// phases 1 + 2 of a one-shot pass over the steps (the histogram phase is shared: launch_cover_pass)
int launch_band_phases(pnx_ctx *ctx, bool write_m) {
    constexpr uint32_t BT = BAND_CW;
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
    // (recorded on one stream as well where the closed forms' tables are derived by the two-kernel route: that derivation starts
    // behind it -- kernels_closed_form.hip.  Otherwise nothing waits for it, and an event between the two kernels of one stream
    // costs the second one 6-15 us: index end -> coverage start 8-17 us with it, as seen in the kernel trace)
    tk->pre_recorded = phased || !quorum_route_fused(ctx->n_groups);
    if (tk->pre_recorded) PNX_HIP(ctx, hipEventRecord(tk->ev_pre, ctx->s_pre));
    if (phased) PNX_HIP(ctx, hipStreamWaitEvent(ctx->s_main, tk->ev_pre, 0));
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

namespace pnx {
// pnx_preload: the first launch of a kernel loads the code object of its translation unit (tens of ms) and builds the
// kernel's function object; asking for a kernel's attributes does the same, without a launch -- and can be done by a host
// thread that has nothing else to do while the GFA text travels to HBM
void preload_band(unsigned what) {
    hipFuncAttributes a;
    auto touch = [&a](const void *k) { (void)hipFuncGetAttributes(&a, k); };
    if (what & PNX_PRELOAD_PASS) {
        touch((const void *)k_band_index);
        touch((const void *)k_band_cover<12, BAND_CW, false, 2>);
        touch((const void *)k_band_cover<24, BAND_CW, false, 2>);
    }
}
}  // namespace pnx
