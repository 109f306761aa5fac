// kernels_rows.hip -- path rows: the path x item presence table, and the coverage pass over it (gfx950).
//
// Every result of the hot path -- AbacusByTotal::coverage (src/graph_broker/abacus.rs:719-744), the
// histograms built from it (abacus.rs:746-787), AbacusByGroup's (r, c) (abacus.rs:859-986) -- depends on
// the SET of items a path visits, never on the order or the multiplicity of its steps.  So the steps of a
// graph are turned, ONCE per upload and in ONE streaming read of the u32 ItemTable, into "path rows":
//
//   row (p, t) = 256 bytes = the 2048 presence bits of path p on item tile t, in the block layout of the
//                presence matrix (pnx_context.hpp: item n -> word n % 64, bit (n % 2048) / 64),
//                for every tile t between the smallest and the largest id on the path.
//
// A coverage pass then never touches the steps again: a wave owns one item tile, walks the visiting order,
// loads one coalesced 256-byte row per (path, tile), ORs the rows of a group in a register and folds the
// group into bit-sliced counters -- no LDS atomics, no boundary search, no per-step work, and no special
// routes: a path whose ids jump around (edge ids in L-line order, shuffled ids) has rows like any other.
// cfg3 (10 M items x 256 paths, 0.98 G steps): 0.32 GB of rows against 3.9 GB of steps.
//
//   k_rows_build   one wave per 4096-step chunk of a path: 16-byte loads, presence bits ORed into an LDS
//                  window of 8 tiles anchored at the chunk's own ids, flushed with one global atomic OR per
//                  non-empty row word; a step outside the window (paths that are not sorted by id) goes
//                  straight to its row with a global atomic.  Also validates every id (the reference
//                  panics on unknown nodes, graph_broker/util.rs:1021) and records the id range of every
//                  path -- so an upload costs exactly one read of its steps.
//   k_rows_order   per pass: row base / tile span of the paths laid out in visiting order (+ the bands of
//                  tiles reached by every 64 entries, for window skipping).
//   k_rows_cover   the coverage kernel (K1) described above.
#include <algorithm>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>

#include "pnx_context.hpp"
#include "step_chunks.hpp"
#include "tile_counters.hpp"

namespace pnx {

constexpr int ROWS_WIN = 8;  // tiles in the LDS window of a build wave
constexpr int ROWS_U = 4;    // 16-byte loads in flight per lane (build)
constexpr uint32_t NO_ROW = 0xFFFFFFFFu;

// ------------------------------------------------------------------------------------------
// build: steps -> rows (SCAN_ONLY: only take the id range of every path)
// ------------------------------------------------------------------------------------------
// The ids are validated through the id range: the smallest id of a path must be >= 1, the largest <= n_items
// (ensure_rows).  A row write can still never leave the table: the window flush skips tiles past the last one,
// the direct route checks the id.
constexpr uint32_t ROWS_CHUNKS_PER_WAVE = 4;  // a wave takes 4 consecutive 4096-step chunks (they may belong to several paths)

template <bool SCAN_ONLY>
__global__ __launch_bounds__(256) void k_rows_build(const uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off,
                                                    const uint64_t *__restrict__ chunk_off, uint32_t n_paths, uint64_t n_chunks,
                                                    uint32_t n_items, uint32_t n_tiles, uint32_t *__restrict__ rows,
                                                    const uint32_t *__restrict__ row_base, uint32_t tstride,
                                                    uint32_t *__restrict__ id_min, uint32_t *__restrict__ id_max) {
    __shared__ uint32_t win_all[4][ROWS_WIN * 64];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t c_first = ((uint64_t)blockIdx.x * 4 + wave) * ROWS_CHUNKS_PER_WAVE;
    if (c_first >= n_chunks) return;  // whole wave; no workgroup barrier is used
    const uint64_t c_end = c_first + ROWS_CHUNKS_PER_WAVE < n_chunks ? c_first + ROWS_CHUNKS_PER_WAVE : n_chunks;
    uint32_t *win = win_all[wave];
    if (!SCAN_ONLY) {
#pragma unroll
        for (int w = 0; w < ROWS_WIN; ++w) win[w * 64 + lane] = 0;
    }
    RunChunk ch = chunk_of(c_first, chunk_off, path_off, n_paths);
    uint32_t path = __builtin_amdgcn_readfirstlane(ch.path);
    uint64_t path_c_end = chunk_off[path + 1];  // first chunk of the next path
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    auto close_path = [&]() {  // id range of the steps seen on `path`
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        if (lane == 0) {
            atomicMin(id_min + path, mn);
            atomicMax(id_max + path, mx);
        }
        mn = 0xFFFFFFFFu;
        mx = 0u;
    };
    for (uint64_t c = c_first; c < c_end; ++c) {
        if (c != c_first) {
            if (c >= path_c_end) {  // the next path that has steps
                close_path();
                do {
                    ++path;
                    path_c_end = chunk_off[path + 1];
                } while (c >= path_c_end);
                ch.pstart = path_off[path];
                ch.start = ch.pstart;
            } else {
                ch.start += RUN_CHUNK;
            }
            const uint64_t left = path_off[path + 1] - ch.start;
            ch.len = (uint32_t)(left < RUN_CHUNK ? left : RUN_CHUNK);
        }
        const uint32_t base = SCAN_ONLY ? 0u : row_base[path];
        // how many steps fit one window: the chunk's own id range says how many ids a step advances
        uint32_t sub = ch.len;
        uint32_t a = 0, b = 0;
        if (!SCAN_ONLY) {
            a = __builtin_amdgcn_readfirstlane(items[ch.start]);
            b = __builtin_amdgcn_readfirstlane(items[ch.start + ch.len - 1]);
            const float span = (float)(a > b ? a - b : b - a) + 1.0f;
            const float room = (float)((ROWS_WIN - 2) * BLOCK_ITEMS);
            if (span > room) {
                const uint32_t s = (uint32_t)((float)ch.len * (room / span)) & ~255u;
                sub = s < 256u ? 256u : s;
            }
            sub = __builtin_amdgcn_readfirstlane(sub);
        }
        for (uint32_t s0 = 0; s0 < ch.len; s0 += sub) {
            const uint32_t sl = ch.len - s0 < sub ? ch.len - s0 : sub;
            const uint64_t first = ch.start + s0;
            uint32_t t0 = 0;
            if (!SCAN_ONLY) {
                // anchor: the tile of the smaller end of the piece, one tile of margin below it
                uint32_t e0 = a, e1 = b;
                if (sub != ch.len) {
                    e0 = __builtin_amdgcn_readfirstlane(items[first]);
                    e1 = __builtin_amdgcn_readfirstlane(items[first + sl - 1]);
                }
                t0 = (e0 < e1 ? e0 : e1) / BLOCK_ITEMS;
                t0 = t0 ? t0 - 1u : 0u;
            }
            const uint32_t id0 = t0 * BLOCK_ITEMS;
            const uint64_t j_al = first & ~3ull;
            const uint32_t head = (uint32_t)(first - j_al);  // steps of the first load that belong to the predecessor
            const uint32_t n_al = head + sl;                 // steps from j_al to the end of the piece
            for (uint32_t r0 = 0; r0 < n_al; r0 += 256u * ROWS_U) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                u32x4 v[ROWS_U];
#pragma unroll
                for (int u = 0; u < ROWS_U; ++u) {
                    const uint32_t r = r0 + (uint32_t)u * 256u + lane * 4u;
                    v[u] = u32x4{0, 0, 0, 0};
                    if (r < n_al) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(items + j_al + r));
                }
#pragma unroll
                for (int u = 0; u < ROWS_U; ++u) {
                    const uint32_t r = r0 + (uint32_t)u * 256u + lane * 4u;
                    const uint32_t ids[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (r + (uint32_t)e - head < sl) {  // unsigned: also false for the steps before the piece
                            const uint32_t id = ids[e];
                            mn = id < mn ? id : mn;
                            mx = id > mx ? id : mx;
                            if (!SCAN_ONLY) {
                                const uint32_t n = id - id0;
                                const uint32_t bit = 1u << ((id >> 6) & 31u);
                                if (n < (uint32_t)ROWS_WIN * BLOCK_ITEMS)
                                    atomicOr(&win[(n >> 11) * 64u + (n & 63u)], bit);
                                else if (id - 1u < n_items)  // outside the window: a path that is not sorted by id
                                    atomicOr(rows + (uint64_t)(uint32_t)(base + (id >> 11) * tstride) * 64u + (id & 63u), bit);
                            }
                        }
                    }
                }
            }
            if (!SCAN_ONLY) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int w = 0; w < ROWS_WIN; ++w) {
                    const uint32_t x = win[w * 64 + lane];
                    if (x) {
                        win[w * 64 + lane] = 0;
                        if (t0 + (uint32_t)w < n_tiles)
                            atomicOr(rows + (uint64_t)(uint32_t)(base + (t0 + (uint32_t)w) * tstride) * 64u + lane, x);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
    }
    close_path();
}

// id range -> (first tile, tiles spanned); an empty path spans nothing
__global__ void k_rows_spans(const uint32_t *__restrict__ id_min, const uint32_t *__restrict__ id_max, uint32_t n_paths, uint32_t n_items,
                             uint32_t *__restrict__ tfirst, uint32_t *__restrict__ tspan) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_paths) return;
    const uint32_t b = id_max[p] < n_items ? id_max[p] : n_items, a = id_min[p] < b ? id_min[p] : b;  // (a rejected upload may hold anything)
    const bool some = id_max[p] != 0u;
    tfirst[p] = some ? a / BLOCK_ITEMS : 0u;
    tspan[p] = some ? b / BLOCK_ITEMS - a / BLOCK_ITEMS + 1u : 0u;
}

__global__ void k_rows_iota(uint32_t *__restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// The rows of the resident graph (no-op when they exist).  validate: fail with PNX_EINVAL on ids outside 1..n_items.
int ensure_rows(pnx_ctx *ctx, bool validate) {
    if (ctx->rows_valid) return PNX_OK;
    const uint32_t P = ctx->n_paths;
    const uint64_t S = ctx->n_steps;
    const uint32_t n_tiles = ctx->n_blocks;  // a row is one block of 2048 items
    int rc;
    const size_t p1 = P ? P : 1;
    if ((rc = ensure(ctx, ctx->d_id_minmax, 2 * p1 * 4)) || (rc = ensure(ctx, ctx->d_row_base, p1 * 4)) ||
        (rc = ensure(ctx, ctx->d_rt_first, p1 * 4)) || (rc = ensure(ctx, ctx->d_rt_span, p1 * 4)))
        return rc;
    uint32_t *d_min = (uint32_t *)ctx->d_id_minmax.p, *d_max = d_min + p1;
    hipStream_t st = ctx->stream;
    PNX_HIP(ctx, hipMemsetAsync(d_min, 0xFF, p1 * 4, st));
    PNX_HIP(ctx, hipMemsetAsync(d_max, 0, p1 * 4, st));
    ctx->h_id_minmax.assign(2 * p1, 0);
    ctx->h_rt_first.assign(P, 0);
    ctx->h_rt_span.assign(P, 0);
    uint64_t n_chunks = 0;
    if (P && S) {
        if ((rc = ensure_chunk_off(ctx))) return rc;
        n_chunks = ctx->h_chunk_off[P];
        if ((n_chunks + 3) / 4 > 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "too many path chunks");
    }
    const uint32_t *items = (const uint32_t *)ctx->d_items.p;
    const uint64_t *path_off = (const uint64_t *)ctx->d_path_off.p, *chunk_off = (const uint64_t *)ctx->d_chunk_off.p;
    const uint64_t n_waves = (n_chunks + ROWS_CHUNKS_PER_WAVE - 1) / ROWS_CHUNKS_PER_WAVE;
    const unsigned grid = (unsigned)((n_waves + 3) / 4);
    auto spans_from_minmax = [&]() {
        uint64_t total = 0;
        uint32_t mx = 0;
        for (uint32_t p = 0; p < P; ++p) {
            const uint32_t hi = ctx->h_id_minmax[p1 + p], b = std::min(hi, ctx->n_items), a = std::min(ctx->h_id_minmax[p], b);
            ctx->h_rt_first[p] = hi ? a / BLOCK_ITEMS : 0;
            ctx->h_rt_span[p] = hi ? b / BLOCK_ITEMS - a / BLOCK_ITEMS + 1 : 0;
            total += ctx->h_rt_span[p];
            mx = std::max(mx, ctx->h_rt_span[p]);
        }
        ctx->rows_max_span = mx;
        return total;
    };
    auto check_ids = [&]() {  // every id in 1..n_items (the id ranges were taken over ALL steps)
        if (!validate) return PNX_OK;
        for (uint32_t p = 0; p < P; ++p)
            if (ctx->h_id_minmax[p] != 0xFFFFFFFFu && (ctx->h_id_minmax[p] == 0 || ctx->h_id_minmax[p1 + p] > ctx->n_items))
                return ctx->fail(PNX_EINVAL, "items contains ids outside 1..n_items");
        return PNX_OK;
    };

    // layout: tile-major over all (tile, path) pairs when that table is no larger than the steps it replaces,
    // else path-major over the tiles every path really spans (thousands of contig paths, each on a few tiles)
    const uint64_t dense_rows = (uint64_t)P * n_tiles;
    bool dense = ctx->rows_layout == 1 || (ctx->rows_layout == 0 && dense_rows * 256 <= std::max<uint64_t>(4 * S, 64ull << 20));
    if (dense_rows >= 0xFFFFFFFFull) dense = false;
    prof_begin(ctx, PNX_K_INDEX);
    if (!dense && n_chunks) {
        // the spans decide the allocation: one extra read of the steps (ids validated, id ranges taken)
        hipLaunchKernelGGL(k_rows_build<true>, dim3(grid), dim3(256), 0, st, items, path_off, chunk_off, P, n_chunks, ctx->n_items,
                           n_tiles, (uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, d_min, d_max);
        PNX_HIP(ctx, hipMemcpyAsync(ctx->h_id_minmax.data(), d_min, 2 * p1 * 4, hipMemcpyDeviceToHost, st));
        PNX_HIP(ctx, hipStreamSynchronize(st));
        if ((rc = check_ids())) {
            prof_end(ctx);
            return rc;
        }
    }
    uint64_t n_rows;
    if (dense) {
        n_rows = dense_rows;
        ctx->row_tstride = P;
        if (P) hipLaunchKernelGGL(k_rows_iota, dim3((P + 255) / 256), dim3(256), 0, st, (uint32_t *)ctx->d_row_base.p, P);
    } else {
        n_rows = spans_from_minmax();
        if (n_rows >= 0xFFFFFFFFull) {
            prof_end(ctx);
            return ctx->fail(PNX_ELIMIT, "the paths span %llu item tiles in all; at most 2^32-2 are supported", (unsigned long long)n_rows);
        }
        ctx->row_tstride = 1;
        ctx->h_row_base.resize(p1);
        uint32_t off = 0;
        for (uint32_t p = 0; p < P; ++p) {
            ctx->h_row_base[p] = off - ctx->h_rt_first[p];  // row(p, t) = base + t, modulo 2^32
            off += ctx->h_rt_span[p];
        }
        if (P) PNX_HIP(ctx, hipMemcpyAsync(ctx->d_row_base.p, ctx->h_row_base.data(), (size_t)P * 4, hipMemcpyHostToDevice, st));
    }
    if ((rc = ensure(ctx, ctx->d_rows, n_rows * 256 + 256))) {
        prof_end(ctx);
        return rc;
    }
    ctx->n_rows = n_rows;
    if (n_rows) PNX_HIP(ctx, hipMemsetAsync(ctx->d_rows.p, 0, n_rows * 256, st));
    if (n_chunks) {
        if (!dense) {  // the scan has filled the id ranges already
            PNX_HIP(ctx, hipMemsetAsync(d_min, 0xFF, p1 * 4, st));
            PNX_HIP(ctx, hipMemsetAsync(d_max, 0, p1 * 4, st));
        }
        hipLaunchKernelGGL(k_rows_build<false>, dim3(grid), dim3(256), 0, st, items, path_off, chunk_off, P, n_chunks, ctx->n_items,
                           n_tiles, (uint32_t *)ctx->d_rows.p, (const uint32_t *)ctx->d_row_base.p, ctx->row_tstride, d_min, d_max);
    }
    if (P)
        hipLaunchKernelGGL(k_rows_spans, dim3((P + 255) / 256), dim3(256), 0, st, d_min, d_max, P, ctx->n_items, (uint32_t *)ctx->d_rt_first.p,
                           (uint32_t *)ctx->d_rt_span.p);
    prof_end(ctx);
    PNX_HIP(ctx, hipGetLastError());
    PNX_HIP(ctx, hipMemcpyAsync(ctx->h_id_minmax.data(), d_min, 2 * p1 * 4, hipMemcpyDeviceToHost, st));
    PNX_HIP(ctx, hipStreamSynchronize(st));  // once per upload; the passes read the rows from other streams too
    if ((rc = check_ids())) return rc;
    spans_from_minmax();
    ctx->rows_tile_major = dense;
    ctx->rows_valid = true;
    ctx->order_normalized = false;
    return PNX_OK;
}

// ------------------------------------------------------------------------------------------
// per pass: the rows of the ordered paths laid out in visiting order
// ------------------------------------------------------------------------------------------
struct RowOrd {
    uint32_t *tfirst, *tspan, *base;  // n_ordered each
    uint32_t *win_lo, *win_hi;        // per 64 entries: the tiles [lo, hi) their paths reach
};

// (also clears the pass's counter block -- flags, histogram, its replicas --: a memset in front of this kernel is two more
// short kernels in the chain of a pass)
__global__ void k_rows_order(const uint32_t *__restrict__ ord_path, uint32_t n_ordered, const uint32_t *__restrict__ tfirst,
                             const uint32_t *__restrict__ tspan, const uint32_t *__restrict__ row_base, RowOrd oi,
                             uint4 *__restrict__ block16, uint32_t n_block16) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t q = k; q < n_block16; q += gridDim.x * blockDim.x) block16[q] = make_uint4(0, 0, 0, 0);
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    if (k < n_ordered) {
        const uint32_t p = ord_path[k];
        const uint32_t f = tfirst[p], s = tspan[p];
        oi.tfirst[k] = f;
        oi.tspan[k] = s;
        oi.base[k] = row_base[p];
        if (s) {
            lo = f;
            hi = f + s;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = __shfl_xor(lo, o), b = __shfl_xor(hi, o);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & 63) == 0 && (k >> 6) < (n_ordered + 63) / 64) {
        oi.win_lo[k >> 6] = lo;
        oi.win_hi[k >> 6] = hi;
    }
}

// ------------------------------------------------------------------------------------------
// K1 over rows
// ------------------------------------------------------------------------------------------
struct RowSplit {  // cut points of the visiting order for the split kernel
    uint32_t k[9];
};

constexpr int ROWS_D = 8;  // rows in flight per wave and buffer (two buffers; 16 makes the compiler keep 232 registers: 0.24 ms)

// One wave owns one item tile (SPLIT > 1: one group-aligned part of the visiting order on one tile; the parts
// add their counters through LDS at the end).  64 entries of the order sit in the lanes (row index, group);
// only the entries that have a row in this tile -- or open a group -- are visited (SKIP: only rows, and the
// 64-entry windows whose band of tiles misses the tile are never loaded).  Rows are fetched ROWS_D at a time
// into one register buffer while the other is consumed.
template <int NPL, bool WRITE_M, int CW, int SPLIT, bool SKIP>
__global__ __launch_bounds__(CW * 64) void k_rows_cover(const uint32_t *__restrict__ rows, uint32_t tstride, RowOrd oi,
                                                        const uint32_t *__restrict__ ord_group, uint32_t n_ordered,
                                                        const uint8_t *__restrict__ exclude, uint32_t n_items, uint32_t n_tiles,
                                                        uint32_t *__restrict__ M, uint64_t row_words,
                                                        uint32_t *__restrict__ countable, RowSplit sp, RowHist hs) {
    constexpr int TPW = CW / SPLIT;  // tiles per workgroup
    static_assert(CW % SPLIT == 0, "waves per workgroup must be a multiple of the split");
    static_assert(!(SKIP && WRITE_M), "a pass that writes the presence matrix visits every group");
    __shared__ uint32_t xch[SPLIT > 1 ? TPW * SPLIT * NPL * 64 : 1];
    extern __shared__ unsigned long long sh_hist[];  // n_groups + 1 bins when the kernel adds the histogram

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t part = SPLIT > 1 ? wave % SPLIT : 0;
    const uint32_t tile_raw = blockIdx.x * TPW + wave / SPLIT;
    const bool active = tile_raw < n_tiles;  // idle waves still meet the barriers
    const uint32_t tile = active ? tile_raw : n_tiles - 1;
    const uint32_t k_lo = SPLIT > 1 ? (active ? sp.k[part] : 0u) : 0u;
    const uint32_t k_hi = SPLIT > 1 ? (active ? sp.k[part + 1] : 0u) : (active ? n_ordered : 0u);
    // The tail of a tile -- counters unpacked into the coverage vector, histogram bins -- is shared by the SPLIT waves of the
    // tile: every part adds up all parts' counters (they meet in LDS) and then owns 32 / SPLIT of the 32 bit positions.
    // (With part 0 alone doing it, one wave ran ~1000 instructions while the others had left.)
    const bool fin = active;
    constexpr uint32_t BITS = 32 / SPLIT;                  // bit positions per part
    const uint32_t b_lo = SPLIT > 1 ? part * BITS : 0u;   // this wave's positions: [b_lo, b_lo + BITS)
    const uint32_t own = SPLIT > 1 ? (BITS == 32 ? 0xFFFFFFFFu : ((1u << BITS) - 1u) << b_lo) : 0xFFFFFFFFu;
    if (hs.rep) {
        for (uint32_t b = threadIdx.x; b <= hs.n_groups; b += CW * 64) sh_hist[b] = 0;
        if (SPLIT == 1) __syncthreads();  // (SPLIT > 1: the barrier of the exchange below orders this before the first add)
    }
    const uint32_t excl = tile_exclusion_word(exclude, tile, lane, n_items);
    TileCounters<NPL> tc;  // bit-sliced counters of the tile (tile_counters.hpp)
    uint32_t(&cnt)[NPL] = tc.cnt;
    uint32_t acc = 0;  // OR of the rows of the current group
    auto flush = [&](uint32_t g) {
        const uint32_t x = acc & ~excl;
        acc = 0;
        if (WRITE_M) M[(uint64_t)g * row_words + (uint64_t)tile * BLOCK_WORDS + lane] = x;
        tc.add_group(x);
    };
    auto settle = [&]() { tc.settle(); };

    // ---- window of 64 order entries, one per lane ----
    uint32_t w_row = NO_ROW, w_g = 0xFFFFFFFFu;
    uint32_t win_base = k_lo;
    uint64_t todo = 0;
    auto load_window = [&](uint32_t base, uint32_t prev_group) {
        win_base = base;
        const uint32_t k = base + lane;
        const bool in = k < k_hi && (!SKIP || k >= k_lo);
        w_g = in ? ord_group[k] : 0xFFFFFFFFu;
        w_row = NO_ROW;
        if (in) {
            const uint32_t jt = tile - oi.tfirst[k];  // wraps for tiles before the path's first one
            if (jt < oi.tspan[k]) w_row = oi.base[k] + tile * tstride;
        }
        if (SKIP) {
            todo = __ballot(w_row != NO_ROW);
        } else {
            uint32_t pg = __shfl_up(w_g, 1);
            if (lane == 0) pg = prev_group;
            todo = __ballot(in && (w_row != NO_ROW || w_g != pg));
        }
    };
    auto seek_window = [&](uint32_t w_from, uint32_t &w_out) {
        const uint32_t w_end = (k_hi + 63) >> 6;
        for (uint32_t w = w_from; w < w_end; w += 64) {
            const uint32_t wi = w + lane;
            const bool hit = wi < w_end && oi.win_lo[wi] <= tile && tile < oi.win_hi[wi];
            const unsigned long long hm = __ballot(hit);
            if (hm) {
                w_out = w + (uint32_t)__builtin_ctzll(hm);
                return true;
            }
        }
        return false;
    };
    // makes the window hold an entry to visit; false when the part is exhausted
    auto refill = [&]() {
        while (todo == 0) {
            if (SKIP) {
                uint32_t w;
                if (!seek_window((win_base >> 6) + 1, w)) return false;
                load_window(w << 6, 0u);
                continue;
            }
            const uint32_t nb = win_base + 64;
            if (nb >= k_hi || nb < win_base) return false;
            load_window(nb, (uint32_t)__builtin_amdgcn_readlane((int)w_g, 63));
        }
        return true;
    };
    bool more = k_lo < k_hi;
    if (more) {
        if (SKIP) {
            uint32_t w;
            if (seek_window(k_lo >> 6, w)) load_window(w << 6, 0u);
            else more = false;
        } else {
            load_window(k_lo, 0xFFFFFFFFu);  // the first entry of the part always opens a group
        }
    }
    uint32_t cur_g = 0xFFFFFFFFu;
    // up to ROWS_D entries of the current window into (v, g): group (0xFFFFFFFF: none) and row words (0: the entry
    // only opens a group)
    auto fetch = [&](uint32_t (&v)[ROWS_D], uint32_t (&g)[ROWS_D]) {
#pragma unroll
        for (int j = 0; j < ROWS_D; ++j) {
            v[j] = 0;
            g[j] = 0xFFFFFFFFu;
            if (todo) {
                const uint32_t i = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1;
                const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)w_row, i);
                g[j] = (uint32_t)__builtin_amdgcn_readlane((int)w_g, i);
                if (row != NO_ROW) v[j] = __builtin_nontemporal_load(rows + (uint64_t)row * 64u + lane);
            }
        }
    };
    auto consume = [&](const uint32_t (&v)[ROWS_D], const uint32_t (&g)[ROWS_D]) {
#pragma unroll
        for (int j = 0; j < ROWS_D; ++j) {
            if (g[j] != 0xFFFFFFFFu) {
                if (g[j] != cur_g) {
                    if (cur_g != 0xFFFFFFFFu) flush(cur_g);
                    cur_g = g[j];
                }
                acc |= v[j];
            }
        }
    };
    {
        uint32_t va[ROWS_D], ga[ROWS_D], vb[ROWS_D], gb[ROWS_D];
        bool have_a = more && refill();
        if (have_a) fetch(va, ga);
        while (have_a) {
            const bool have_b = refill();
            if (have_b) fetch(vb, gb);
            consume(va, ga);
            if (!have_b) break;
            have_a = refill();
            if (have_a) fetch(va, ga);
            consume(vb, gb);
        }
        if (cur_g != 0xFFFFFFFFu) flush(cur_g);
        settle();
    }

    if (SPLIT > 1) {
        // every part leaves its counters in LDS and adds the others' to its own
        uint32_t *xt = xch + (size_t)(wave / SPLIT) * SPLIT * NPL * 64;
#pragma unroll
        for (int k = 0; k < NPL; ++k) xt[(part * NPL + k) * 64 + lane] = cnt[k];
        __syncthreads();
        if (fin) {
            for (int q = 1; q < SPLIT; ++q) {
                const uint32_t src = (part + (uint32_t)q) % SPLIT;
                uint32_t carry = 0;
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    const uint32_t a = cnt[k], b = xt[(src * NPL + k) * 64 + lane];
                    cnt[k] = a ^ b ^ carry;
                    carry = (a & b) | (carry & (a ^ b));
                }
            }
        }
    }
    if (fin) tile_tail<NPL>(cnt, tile, lane, n_items, countable, hs, sh_hist, b_lo, BITS, own);
    if (hs.rep) {
        __syncthreads();
        hist_bins_flush(hs, sh_hist, CW * 64);
    }
}

// ------------------------------------------------------------------------------------------
// K1 over rows, four rows per load (verdict r3 #7): for plain histogram passes over DENSE tile-major rows.
// One wave owns one item tile.  Its four 16-lane quarters walk the four group-aligned parts of the visiting order side by
// side: one 16-byte load per lane fetches FOUR rows (one per quarter, four presence words of it per lane), so a wave needs a
// quarter of the load instructions, address computations and scalar bookkeeping per row of k_rows_cover.  What a part's
// entry is on this tile -- its row, and whether it ends its group -- is worked out once per wave into LDS (4 bytes per
// entry); a slot of the main loop is then: one LDS read, one load, four ORs.  The quarters' groups end at different slots,
// but the counters are only touched when ALL quarters have finished a group (a quarter that is done idles: its loads go to
// row 0 and are masked), so that the carry-save tree of the bit-sliced counters sees wave-uniform steps.  At the end the four
// quarters' counters are added up by a reduce-scatter over lane ^ 32 and lane ^ 16, which leaves every lane with ONE word of
// the tile, and the tail of k_rows_cover takes over.
// ------------------------------------------------------------------------------------------
constexpr int ROWQ_D = 4;                        // slots (of four rows) in flight per register buffer, two buffers
constexpr uint32_t ROWQ_NONE = 0x7FFFFFFFu;      // meta: no row on this tile
constexpr uint32_t ROWQ_LAST = 0x80000000u;      // meta: the entry ends its group
constexpr uint32_t ROWQ_MAX_ORDER = 2048;        // entries of the order a wave keeps in LDS (8 KB)

typedef uint32_t rowq_u32x4 __attribute__((ext_vector_type(4)));

template <int NPL>
__global__ __launch_bounds__(256) void k_rows_cover_q(const uint32_t *__restrict__ rows, uint32_t tstride, RowOrd oi,
                                                      const uint32_t *__restrict__ ord_group, uint32_t n_ordered,
                                                      const uint8_t *__restrict__ exclude, uint32_t n_items, uint32_t n_tiles,
                                                      uint32_t *__restrict__ countable, RowSplit sp, RowHist hs) {
    extern __shared__ unsigned long long sh_hist[];  // [n_groups + 1 bins when the kernel adds the histogram | meta of 4 waves]
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t q = lane >> 4, j = lane & 15u;
    const uint32_t tile_raw = blockIdx.x * 4 + wave;
    const bool active = tile_raw < n_tiles;  // idle waves still meet the barriers
    const uint32_t tile = active ? tile_raw : n_tiles - 1;
    const size_t hist_words = hs.rep ? ((size_t)hs.n_groups + 1) : 0;
    uint32_t *meta = reinterpret_cast<uint32_t *>(sh_hist + hist_words) + (size_t)wave * n_ordered;
    if (hs.rep) {
        for (uint32_t b = threadIdx.x; b <= hs.n_groups; b += 256) sh_hist[b] = 0;
    }
    // ---- what every entry of the order is on this tile ----
    for (uint32_t k = lane; k < n_ordered; k += 64) {
        const uint32_t jt = tile - oi.tfirst[k];  // wraps for tiles before the path's first one
        uint32_t m = jt < oi.tspan[k] ? oi.base[k] + tile * tstride : ROWQ_NONE;
        const bool last = k + 1 == n_ordered || k + 1 == sp.k[1] || k + 1 == sp.k[2] || k + 1 == sp.k[3] || ord_group[k + 1] != ord_group[k];
        meta[k] = m | (last ? ROWQ_LAST : 0u);
    }
    __syncthreads();  // (also: the zeroed histogram bins)
    uint32_t kq = active ? sp.k[q] : 0u;
    const uint32_t khi = active ? sp.k[q + 1] : 0u;
    uint32_t excl[4];
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) excl[c] = tile_exclusion_word(exclude, tile, 4 * j + c, n_items);
    TileCounters<NPL> tc[4];
    uint32_t acc[4] = {0, 0, 0, 0};
    bool done = false;  // this quarter has finished the group of the current round
    // issue stage of one slot: -> the loaded words, whether they count, whether the round ends behind this slot
    bool any_live = false;  // (wave-uniform) the last fill held at least one entry
    auto issue = [&](rowq_u32x4 &v, bool &keep, bool &round_end) {
        const bool live = !done && kq < khi;
        any_live = any_live || __builtin_amdgcn_ballot_w64(live) != 0ull;
        const uint32_t m = live ? meta[kq] : ROWQ_NONE;
        const uint32_t row = m & 0x7FFFFFFFu;
        keep = live && row != ROWQ_NONE;
        v = __builtin_nontemporal_load(reinterpret_cast<const rowq_u32x4 *>(rows + (uint64_t)(keep ? row : 0u) * 64u) + j);
        if (live) {
            ++kq;
            done = (m & ROWQ_LAST) != 0u;
        }
        round_end = __builtin_amdgcn_ballot_w64(done || kq >= khi) == ~0ull;
        if (round_end) done = false;
    };
    auto consume = [&](const rowq_u32x4 &v, bool keep, bool round_end) {
        const uint32_t km = keep ? 0xFFFFFFFFu : 0u;
        acc[0] |= v.x & km;
        acc[1] |= v.y & km;
        acc[2] |= v.z & km;
        acc[3] |= v.w & km;
        if (round_end) {  // (wave-uniform)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                tc[c].add_group(acc[c] & ~excl[c]);
                acc[c] = 0;
            }
        }
    };
    {
        // Two register buffers of ROWQ_D slots that swap roles (one set refilled in place makes the compiler park the new loads
        // elsewhere and copy them back at the loop head, i.e. wait for them); a group consumes `in` slot by slot while the loads
        // of `out` go out, ROWQ_D loads in flight throughout.  Every group issues ALL its loads, whatever is left of the order
        // (a slot behind the end loads row 0 and counts nothing, its "round" adds an empty group): with the number of loads
        // in flight known at every point the compiler waits for exactly the slot it consumes.
        rowq_u32x4 va[ROWQ_D], vb[ROWQ_D];
        bool ka[ROWQ_D], kb[ROWQ_D], ea[ROWQ_D], eb[ROWQ_D];
        auto group = [&](const rowq_u32x4 (&vi)[ROWQ_D], const bool (&ki)[ROWQ_D], const bool (&ei)[ROWQ_D], rowq_u32x4 (&vo)[ROWQ_D],
                         bool (&ko)[ROWQ_D], bool (&eo)[ROWQ_D]) {
            any_live = false;
#pragma unroll
            for (int t = 0; t < ROWQ_D; ++t) {
                issue(vo[t], ko[t], eo[t]);
                consume(vi[t], ki[t], ei[t]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma unroll
        for (int t = 0; t < ROWQ_D; ++t) issue(va[t], ka[t], ea[t]);
        while (any_live) {  // `in` holds entries
            group(va, ka, ea, vb, kb, eb);
            if (!any_live) break;  // nothing went into vb
            group(vb, kb, eb, va, ka, ea);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) tc[c].settle();
    }
    // ---- the four quarters' counters -> one word per lane: reduce-scatter over lane ^ 32, then lane ^ 16 ----
    uint32_t two[2][NPL], one[NPL];
    {
        const bool up = (lane & 32u) != 0u;  // keeps words 2, 3 of its four; the lower half keeps 0, 1
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t carry = 0;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const uint32_t mine = up ? tc[2 + h].cnt[k] : tc[h].cnt[k], give = up ? tc[h].cnt[k] : tc[2 + h].cnt[k];
                const uint32_t got = (uint32_t)__shfl_xor((int)give, 32);
                two[h][k] = mine ^ got ^ carry;
                carry = (mine & got) | (carry & (mine ^ got));
            }
        }
        const bool up2 = (lane & 16u) != 0u;  // keeps the second of its two
        uint32_t carry = 0;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const uint32_t mine = up2 ? two[1][k] : two[0][k], give = up2 ? two[0][k] : two[1][k];
            const uint32_t got = (uint32_t)__shfl_xor((int)give, 16);
            one[k] = mine ^ got ^ carry;
            carry = (mine & got) | (carry & (mine ^ got));
        }
    }
    const uint32_t word = 4 * j + ((lane >> 5) & 1u) * 2u + ((lane >> 4) & 1u);
    if (active) tile_tail<NPL>(one, tile, lane, n_items, countable, hs, sh_hist, 0u, 32u, 0xFFFFFFFFu, word);
    if (hs.rep) {
        __syncthreads();
        hist_bins_flush(hs, sh_hist, 256);
    }
}

// Within a group the order of the paths does not matter for any result, so the paths of every group are put
// in the order of their first tile once per (graph, order): 64 consecutive entries then reach a narrow band of
// tiles, and a coverage wave skips the windows whose band misses its tile.
static int normalize_order_rows(pnx_ctx *ctx) {
    if (ctx->order_normalized) return PNX_OK;
    const size_t n = ctx->h_ord_path.size();
    if (n == ctx->n_ordered && n > 1 && ctx->h_rt_first.size() == ctx->n_paths) {
        bool changed = false;
        size_t a = 0;
        const auto &tf = ctx->h_rt_first;
        while (a < n) {
            size_t b = a + 1;
            while (b < n && ctx->h_ord_group[b] == ctx->h_ord_group[a]) ++b;
            auto key_less = [&](uint32_t x, uint32_t y) { return tf[x] != tf[y] ? tf[x] < tf[y] : x < y; };
            if (b - a > 1 && !std::is_sorted(ctx->h_ord_path.begin() + a, ctx->h_ord_path.begin() + b, key_less)) {
                std::sort(ctx->h_ord_path.begin() + a, ctx->h_ord_path.begin() + b, key_less);
                changed = true;
            }
            a = b;
        }
        if (changed) {
            PNX_HIP(ctx, hipMemcpyAsync(ctx->d_ord_path.p, ctx->h_ord_path.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    ctx->order_normalized = true;
    return PNX_OK;
}

template <int NPL>
static void launch_rows_cover_t(pnx_ctx *ctx, bool write_m) {
    Ticket *tk = ctx->cur;
    const RowOrd oi{(uint32_t *)tk->d_ord_tfirst.p, (uint32_t *)tk->d_ord_tspan.p, (uint32_t *)tk->d_ord_off.p,
                    (uint32_t *)tk->d_win_lo.p, (uint32_t *)tk->d_win_hi.p};
    const uint64_t row_words = (uint64_t)ctx->n_blocks * BLOCK_WORDS;
    const uint32_t n_tiles = ctx->n_blocks;
    RowSplit sp{};
    auto launch = [&](auto kern, int cw, int split) {
        const unsigned tpw = (unsigned)(cw / split);
        const unsigned grid = (n_tiles + tpw - 1) / tpw;
        for (int j = 0; j <= split; ++j) {  // group-aligned cut points of the visiting order
            uint64_t t = (uint64_t)ctx->n_ordered * j / split;
            while (t > 0 && t < ctx->n_ordered && ctx->h_ord_group[t] == ctx->h_ord_group[t - 1]) ++t;
            sp.k[j] = (uint32_t)t;
        }
        const RowHist hs{tk->hist_fused ? (unsigned long long *)tk->d_hist_rep : nullptr,
                         ctx->weighted ? (const uint32_t *)ctx->d_weights.p : (const uint32_t *)nullptr, ctx->n_groups};
        const size_t lds_hist = tk->hist_fused ? ((size_t)ctx->n_groups + 1) * sizeof(unsigned long long) : 0;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(cw * 64), lds_hist, ctx->s_main, (const uint32_t *)ctx->d_rows.p, ctx->row_tstride, oi,
                           (const uint32_t *)ctx->d_ord_group.p, ctx->n_ordered,
                           ctx->have_exclude ? (const uint8_t *)ctx->d_exclude.p : (const uint8_t *)nullptr, ctx->n_items, n_tiles,
                           (uint32_t *)ctx->d_M.p, row_words, (uint32_t *)tk->d_countable.p, sp, hs);
    };
    const bool skip = !write_m && (ctx->cover_skip == 1 || (ctx->cover_skip == 0 && ctx->n_ordered >= 4096));
    // four rows per load (k_rows_cover_q): plain histogram passes over dense tile-major rows, orders that a wave can keep in
    // LDS and that split into four group-aligned parts; PNX_CFG_ROWS_KERNEL 1 keeps k_rows_cover, 2 asks for the new one
    // wherever it is legal
    {
        const bool legal = !write_m && ctx->rows_tile_major && ctx->n_ordered <= ROWQ_MAX_ORDER && ctx->n_groups >= 4 && ctx->n_ordered >= 4 &&
                           !(ctx->cover_skip == 1);
        uint64_t rows_in_order = 0;
        if (legal && ctx->h_rt_span.size() == ctx->n_paths)
            for (uint32_t k = 0; k < ctx->n_ordered; ++k) rows_in_order += ctx->h_rt_span[ctx->h_ord_path[k]];
        const bool dense = rows_in_order * 10 >= (uint64_t)n_tiles * ctx->n_ordered * 7;  // idle slots are loads of row 0
        // (not the default yet: 0.105 ms per launch on cfg3 against k_rows_cover's 0.092 -- 104 registers, 4 waves per SIMD, but
        // the compiler reuses the destination registers of loads in flight as temporaries at the loop head and waits for them;
        // DESIGN section 8)
        (void)dense;
        if (legal && ctx->rows_kernel == 2) {
            for (int jq = 0; jq <= 4; ++jq) {  // group-aligned cut points of the visiting order
                uint64_t t = (uint64_t)ctx->n_ordered * jq / 4;
                while (t > 0 && t < ctx->n_ordered && ctx->h_ord_group[t] == ctx->h_ord_group[t - 1]) ++t;
                sp.k[jq] = (uint32_t)t;
            }
            const RowHist hs{tk->hist_fused ? (unsigned long long *)tk->d_hist_rep : nullptr,
                             ctx->weighted ? (const uint32_t *)ctx->d_weights.p : (const uint32_t *)nullptr, ctx->n_groups};
            const size_t lds = (tk->hist_fused ? ((size_t)ctx->n_groups + 1) * sizeof(unsigned long long) : 0) + (size_t)4 * ctx->n_ordered * 4;
            hipLaunchKernelGGL(k_rows_cover_q<NPL>, dim3((n_tiles + 3) / 4), dim3(256), lds, ctx->s_main, (const uint32_t *)ctx->d_rows.p,
                               ctx->row_tstride, oi, (const uint32_t *)ctx->d_ord_group.p, ctx->n_ordered,
                               ctx->have_exclude ? (const uint8_t *)ctx->d_exclude.p : (const uint8_t *)nullptr, ctx->n_items, n_tiles,
                               (uint32_t *)tk->d_countable.p, sp, hs);
            ctx->n_rows_q_passes += 1;
            return;
        }
    }
    int split = ctx->cover_split;
    if (split == 0) {  // enough waves to fill the chip a few times over, and no part shorter than 32 entries
        const uint64_t want = (uint64_t)ctx->prop.multiProcessorCount * 4 * 8 * 2;
        split = 1;
        while (split < 8 && (uint64_t)n_tiles * split < want && (uint32_t)split * 2 <= ctx->n_groups &&
               ctx->n_ordered / (uint32_t)(split * 2) >= 32)
            split *= 2;
    }
    if ((uint32_t)split > ctx->n_groups) split = 1;
    auto go = [&](auto k1, auto k2, auto k4, auto k8) {
        if (split == 2) launch(k2, 4, 2);
        else if (split == 4) launch(k4, 4, 4);
        else if (split == 8) launch(k8, 8, 8);
        else launch(k1, 4, 1);
    };
    if (write_m) go(k_rows_cover<NPL, true, 4, 1, false>, k_rows_cover<NPL, true, 4, 2, false>, k_rows_cover<NPL, true, 4, 4, false>, k_rows_cover<NPL, true, 8, 8, false>);
    else if (skip) go(k_rows_cover<NPL, false, 4, 1, true>, k_rows_cover<NPL, false, 4, 2, true>, k_rows_cover<NPL, false, 4, 4, true>, k_rows_cover<NPL, false, 8, 8, true>);
    else go(k_rows_cover<NPL, false, 4, 1, false>, k_rows_cover<NPL, false, 4, 2, false>, k_rows_cover<NPL, false, 4, 4, false>, k_rows_cover<NPL, false, 8, 8, false>);
}

// phases 1 + 2 of a pass over rows (the histogram phase is shared with the step routes: launch_cover_pass)
int launch_rows_phases(pnx_ctx *ctx, bool write_m) {
    Ticket *tk = ctx->cur;
    int rc;
    if ((rc = normalize_order_rows(ctx))) return rc;
    const bool phased = ctx->s_pre != ctx->s_main;
    if (ctx->n_ordered) {
        prof_begin(ctx, PNX_K_SCATTER, ctx->s_pre);
        const RowOrd oi{(uint32_t *)tk->d_ord_tfirst.p, (uint32_t *)tk->d_ord_tspan.p, (uint32_t *)tk->d_ord_off.p,
                        (uint32_t *)tk->d_win_lo.p, (uint32_t *)tk->d_win_hi.p};
        hipLaunchKernelGGL(k_rows_order, dim3((ctx->n_ordered + 255) / 256), dim3(256), 0, ctx->s_pre, (const uint32_t *)ctx->d_ord_path.p,
                           ctx->n_ordered, (const uint32_t *)ctx->d_rt_first.p, (const uint32_t *)ctx->d_rt_span.p,
                           (const uint32_t *)ctx->d_row_base.p, oi, (uint4 *)tk->d_block.p, (uint32_t)(tk->block_bytes / 16));
        prof_end(ctx);
        PNX_HIP(ctx, hipGetLastError());
    }
    if (phased) {
        PNX_HIP(ctx, hipEventRecord(tk->ev_pre, ctx->s_pre));
        PNX_HIP(ctx, hipStreamWaitEvent(ctx->s_main, tk->ev_pre, 0));
    }
    uint32_t bits = 1;  // planes needed to count up to n_groups inclusive
    while (bits < 32 && (ctx->n_groups >> bits) != 0) ++bits;
    prof_begin(ctx, PNX_K_COVER, ctx->s_main);
    // (two widths only: with the carry-save tree in front, unused planes cost a fraction of an instruction per group,
    // and every instance less is code the first pass of a process does not have to load)
    if (bits <= 12) launch_rows_cover_t<12>(ctx, write_m);
    else if (bits <= 24) launch_rows_cover_t<24>(ctx, write_m);
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
void preload_rows(unsigned what) {
    hipFuncAttributes a;
    auto touch = [&a](const void *k) { (void)hipFuncGetAttributes(&a, k); };
    if (what & PNX_PRELOAD_PASS) {
        touch((const void *)k_rows_build<true>);
        touch((const void *)k_rows_build<false>);
        touch((const void *)k_rows_spans);
        touch((const void *)k_rows_order);
        touch((const void *)k_rows_cover<12, false, 8, 8, false>);
    }
}
}  // namespace pnx
