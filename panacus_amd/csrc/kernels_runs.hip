// kernels_runs.hip -- the run index: tile route for paths that are NOT tile-monotone.
//
// K0's binary-search index is only valid for tile-monotone paths.  Real pangenome paths are
// usually *nearly* monotone (ids follow the walk order with local back-steps around bubbles),
// so one back-step across a tile border would push a whole path onto the atomic scatter route.
// Instead such a path is cut into RUNS -- maximal stretches of consecutive steps that stay in
// one tile -- with one streaming read of its steps; the runs are sorted by (tile, group) and
// each tile's wave consumes the runs of group g right before it folds g's bitmap.  No global
// atomics, every step is read once more than on the fast path (for the index) and the result
// is exact for ANY path.  Only paths whose runs are very short (edge-id paths: ids are random
// along the walk) stay on the scatter route, where sorting one record per step would cost more
// than the atomics.
//
// The build is host-driven (it needs sizes on the host for allocation and for rocPRIM's sort);
// it only happens when a pass has flagged unclassified general paths, and it is cached with
// the tile index (the sort by group is redone when the visiting order changes).
#include <cstring>

#include <hip/hip_runtime.h>

#include <algorithm>

#include <rocprim/rocprim.hpp>

#include "pnx_context.hpp"

namespace pnx {

constexpr uint32_t RUN_CHUNK = 4096;      // steps per chunk (one wave walks one chunk)
constexpr uint32_t RUN_MIN_AVG_LEN = 16;  // paths with shorter average runs stay on the scatter route

struct RunChunk {
    uint64_t start;   // first step (absolute index into items)
    uint64_t pstart;  // first step of the path
    uint32_t len;     // steps in this chunk
    uint32_t path;
};

// flag = this step starts a run (first step of the path, or a different tile than the step before)
__device__ static inline bool run_starts(const uint32_t *__restrict__ items, uint64_t j, uint64_t pstart,
                                         uint32_t cur, uint32_t lane, uint32_t tile_shift) {
    uint32_t prev = __shfl_up(cur, 1);
    if (lane == 0 && j > pstart) prev = items[j - 1];
    return j == pstart || (cur >> tile_shift) != (prev >> tile_shift);
}

__global__ __launch_bounds__(256) void k_runs_count(const uint32_t *__restrict__ items,
                                                    const RunChunk *__restrict__ chunks, uint32_t n_chunks,
                                                    uint32_t tile_shift, uint32_t *__restrict__ counts) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const RunChunk ch = chunks[c];
    uint32_t total = 0;
    for (uint32_t it = 0; it < ch.len; it += 64) {
        const uint64_t j = ch.start + it + lane;
        const bool in = it + lane < ch.len;
        const uint32_t cur = in ? items[j] : 0;
        const bool f = run_starts(items, j, ch.pstart, cur, lane, tile_shift) && in;
        total += (uint32_t)__popcll(__ballot(f));
    }
    if (lane == 0) counts[c] = total;
}

__global__ __launch_bounds__(256) void k_runs_emit(const uint32_t *__restrict__ items,
                                                   const RunChunk *__restrict__ chunks, uint32_t n_chunks,
                                                   const uint64_t *__restrict__ offs /* UINT64_MAX = skip */,
                                                   uint32_t tile_shift, uint64_t *__restrict__ r_start,
                                                   uint32_t *__restrict__ r_tile, uint32_t *__restrict__ r_path) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    uint64_t off = offs[c];
    if (off == 0xFFFFFFFFFFFFFFFFull) return;
    const RunChunk ch = chunks[c];
    for (uint32_t it = 0; it < ch.len; it += 64) {
        const uint64_t j = ch.start + it + lane;
        const bool in = it + lane < ch.len;
        const uint32_t cur = in ? items[j] : 0;
        const bool f = run_starts(items, j, ch.pstart, cur, lane, tile_shift) && in;
        const unsigned long long b = __ballot(f);
        if (f) {
            const uint64_t idx = off + (uint64_t)__popcll(b & ((1ull << lane) - 1ull));
            r_start[idx] = j;
            r_tile[idx] = cur >> tile_shift;
            r_path[idx] = ch.path;
        }
        off += (uint64_t)__popcll(b);
    }
}

// run length = distance to the next run of the same path (runs are emitted in path order)
__global__ void k_runs_len(const uint64_t *__restrict__ r_start, const uint32_t *__restrict__ r_path,
                           const uint64_t *__restrict__ path_off, uint64_t n_runs, uint32_t *__restrict__ r_len) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_runs) return;
    const uint32_t p = r_path[i];
    const uint64_t end = (i + 1 < n_runs && r_path[i + 1] == p) ? r_start[i + 1] : path_off[p + 1];
    r_len[i] = (uint32_t)(end - r_start[i]);
}

// sort key: tile major, then the group of the path in the current visiting order
__global__ void k_runs_keys(const uint32_t *__restrict__ r_tile, const uint32_t *__restrict__ r_path,
                            const uint32_t *__restrict__ group_of_path /* UINT32_MAX = not visited */,
                            uint64_t n_runs, uint64_t *__restrict__ keys, uint32_t *__restrict__ idx) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_runs) return;
    const uint32_t g = group_of_path[r_path[i]];
    keys[i] = g == 0xFFFFFFFFu ? 0xFFFFFFFFFFFFFFFFull : (((uint64_t)r_tile[i] << 32) | g);
    idx[i] = (uint32_t)i;
}

__global__ void k_runs_gather(const uint64_t *__restrict__ keys_sorted, const uint32_t *__restrict__ idx_sorted,
                              const uint64_t *__restrict__ r_start, const uint32_t *__restrict__ r_len,
                              uint64_t n_runs, uint64_t *__restrict__ s_start, uint32_t *__restrict__ s_len,
                              uint32_t *__restrict__ s_group) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_runs) return;
    const uint32_t k = idx_sorted[i];
    s_start[i] = r_start[k];
    s_len[i] = r_len[k];
    s_group[i] = (uint32_t)(keys_sorted[i] & 0xFFFFFFFFull);
}

// tile_off[t] = first sorted run whose tile is >= t (runs of unvisited paths sort behind every tile)
__global__ void k_runs_tile_off(const uint64_t *__restrict__ keys_sorted, uint64_t n_runs, uint32_t n_tiles,
                                uint64_t *__restrict__ tile_off) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    const uint64_t key = (uint64_t)t << 32;
    uint64_t lo = 0, hi = n_runs;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (keys_sorted[mid] < key) lo = mid + 1; else hi = mid;
    }
    tile_off[t] = lo;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int sync_copy(pnx_ctx *ctx, void *dst, const void *src, size_t n, hipMemcpyKind kind) {
    PNX_HIP(ctx, hipMemcpyAsync(dst, src, n, kind, ctx->stream));
    PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PNX_OK;
}

// (re)sort the runs by (tile, group) for the current visiting order
int sort_run_index(pnx_ctx *ctx) {
    ctx->runs_sorted = false;
    const uint64_t n = ctx->n_runs;
    int rc;
    if ((rc = ensure(ctx, ctx->d_run_tile_off, ((size_t)ctx->n_tiles + 1) * sizeof(uint64_t)))) return rc;
    if (n == 0) {
        PNX_HIP(ctx, hipMemsetAsync(ctx->d_run_tile_off.p, 0, ((size_t)ctx->n_tiles + 1) * sizeof(uint64_t), ctx->stream));
        ctx->runs_sorted = true;
        return PNX_OK;
    }
    // path -> group of the current order
    std::vector<uint32_t> gop(ctx->n_paths, 0xFFFFFFFFu);
    for (size_t k = 0; k < ctx->h_ord_path.size(); ++k) gop[ctx->h_ord_path[k]] = ctx->h_ord_group[k];
    DevBuf d_gop, d_keys, d_keys2, d_idx, d_idx2, d_tmp;
    auto cleanup = [&]() {
        for (DevBuf *b : {&d_gop, &d_keys, &d_keys2, &d_idx, &d_idx2, &d_tmp}) release(*b);
    };
    if ((rc = ensure(ctx, d_gop, gop.size() * 4)) || (rc = ensure(ctx, d_keys, n * 8)) || (rc = ensure(ctx, d_keys2, n * 8)) ||
        (rc = ensure(ctx, d_idx, n * 4)) || (rc = ensure(ctx, d_idx2, n * 4)) || (rc = ensure(ctx, ctx->d_srun_start, n * 8)) ||
        (rc = ensure(ctx, ctx->d_srun_len, n * 4)) || (rc = ensure(ctx, ctx->d_srun_group, n * 4))) {
        cleanup();
        return rc;
    }
    if ((rc = sync_copy(ctx, d_gop.p, gop.data(), gop.size() * 4, hipMemcpyHostToDevice))) {
        cleanup();
        return rc;
    }
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_runs_keys, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_run_tile.p,
                       (const uint32_t *)ctx->d_run_path.p, (const uint32_t *)d_gop.p, n, (uint64_t *)d_keys.p,
                       (uint32_t *)d_idx.p);
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint64_t *)d_keys.p, (uint64_t *)d_keys2.p,
                                             (uint32_t *)d_idx.p, (uint32_t *)d_idx2.p, n, 0, 64, ctx->stream);
    if (e == hipSuccess && (rc = ensure(ctx, d_tmp, tmp_bytes)) == PNX_OK)
        e = rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, (uint64_t *)d_keys.p, (uint64_t *)d_keys2.p,
                                      (uint32_t *)d_idx.p, (uint32_t *)d_idx2.p, n, 0, 64, ctx->stream);
    if (e != hipSuccess || rc) {
        cleanup();
        return rc ? rc : ctx->fail(PNX_EHIP, "run index sort failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k_runs_gather, dim3(grid), dim3(256), 0, ctx->stream, (const uint64_t *)d_keys2.p,
                       (const uint32_t *)d_idx2.p, (const uint64_t *)ctx->d_run_start.p,
                       (const uint32_t *)ctx->d_run_len.p, n, (uint64_t *)ctx->d_srun_start.p,
                       (uint32_t *)ctx->d_srun_len.p, (uint32_t *)ctx->d_srun_group.p);
    hipLaunchKernelGGL(k_runs_tile_off, dim3((ctx->n_tiles + 1 + 255) / 256), dim3(256), 0, ctx->stream,
                       (const uint64_t *)d_keys2.p, n, ctx->n_tiles, (uint64_t *)ctx->d_run_tile_off.p);
    e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return ctx->fail(PNX_EHIP, "run index sort failed: %s", hipGetErrorString(e));
    ctx->runs_sorted = true;
    return PNX_OK;
}

// classify every path that is not tile-monotone (path_class 1, 2 or 3 on the device) into
// run route (2) or scatter route (3) and build the run list of the run-route paths
int build_run_index(pnx_ctx *ctx) {
    const uint32_t P = ctx->n_paths;
    const uint32_t tile_items = ctx->tile_blocks * BLOCK_ITEMS;
    uint32_t tile_shift = 0;
    while ((1u << tile_shift) < tile_items) ++tile_shift;
    int rc;
    std::vector<uint8_t> cls(P ? P : 1, 0);
    if (P && (rc = sync_copy(ctx, cls.data(), ctx->d_path_class.p, P, hipMemcpyDeviceToHost))) return rc;

    // chunks over all general paths
    std::vector<RunChunk> chunks;
    std::vector<uint32_t> general;
    for (uint32_t p = 0; p < P; ++p) {
        if (!cls[p]) continue;
        general.push_back(p);
        const uint64_t s = ctx->h_path_off[p], e = ctx->h_path_off[p + 1];
        for (uint64_t b = s; b < e; b += RUN_CHUNK)
            chunks.push_back(RunChunk{b, s, (uint32_t)std::min<uint64_t>(RUN_CHUNK, e - b), p});
        if (e == s) cls[p] = 2;  // empty path: nothing to do on either route
    }
    ctx->n_runs = 0;
    ctx->n_scatter_paths = 0;
    if (chunks.size() >= 0xFFFFFFFFull) return ctx->fail(PNX_ELIMIT, "too many general path chunks");
    const uint32_t n_chunks = (uint32_t)chunks.size();
    std::vector<uint32_t> counts(n_chunks ? n_chunks : 1, 0);
    DevBuf d_chunks, d_counts, d_offs;
    auto cleanup = [&]() {
        release(d_chunks);
        release(d_counts);
        release(d_offs);
    };
    if (n_chunks) {
        if ((rc = ensure(ctx, d_chunks, (size_t)n_chunks * sizeof(RunChunk))) || (rc = ensure(ctx, d_counts, (size_t)n_chunks * 4)) ||
            (rc = ensure(ctx, d_offs, (size_t)n_chunks * 8)) ||
            (rc = sync_copy(ctx, d_chunks.p, chunks.data(), (size_t)n_chunks * sizeof(RunChunk), hipMemcpyHostToDevice))) {
            cleanup();
            return rc;
        }
        prof_begin(ctx, PNX_K_SCATTER);
        hipLaunchKernelGGL(k_runs_count, dim3((n_chunks + 3) / 4), dim3(256), 0, ctx->stream,
                           (const uint32_t *)ctx->d_items.p, (const RunChunk *)d_chunks.p, n_chunks, tile_shift,
                           (uint32_t *)d_counts.p);
        prof_end(ctx);
        if ((rc = sync_copy(ctx, counts.data(), d_counts.p, (size_t)n_chunks * 4, hipMemcpyDeviceToHost))) {
            cleanup();
            return rc;
        }
    }
    // per-path totals -> class; offsets of the run-route chunks
    std::vector<uint64_t> runs_of(P, 0);
    for (uint32_t c = 0; c < n_chunks; ++c) runs_of[chunks[c].path] += counts[c];
    for (uint32_t p : general) {
        const uint64_t len = ctx->h_path_off[p + 1] - ctx->h_path_off[p];
        if (len == 0) continue;
        cls[p] = (runs_of[p] * RUN_MIN_AVG_LEN <= len || runs_of[p] <= 64) ? 2 : 3;
        if (cls[p] == 3) ctx->n_scatter_paths += 1;
    }
    std::vector<uint64_t> offs(n_chunks ? n_chunks : 1, 0xFFFFFFFFFFFFFFFFull);
    uint64_t n_runs = 0;
    for (uint32_t c = 0; c < n_chunks; ++c)
        if (cls[chunks[c].path] == 2) {
            offs[c] = n_runs;
            n_runs += counts[c];
        }
    if (n_runs >= 0xFFFFFFFFull) {
        cleanup();
        return ctx->fail(PNX_ELIMIT, "more than 2^32-1 runs in the run index");
    }
    if (P && (rc = sync_copy(ctx, ctx->d_path_class.p, cls.data(), P, hipMemcpyHostToDevice))) {
        cleanup();
        return rc;
    }
    if (n_runs) {
        if ((rc = ensure(ctx, ctx->d_run_start, n_runs * 8)) || (rc = ensure(ctx, ctx->d_run_len, n_runs * 4)) ||
            (rc = ensure(ctx, ctx->d_run_tile, n_runs * 4)) || (rc = ensure(ctx, ctx->d_run_path, n_runs * 4)) ||
            (rc = sync_copy(ctx, d_offs.p, offs.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice))) {
            cleanup();
            return rc;
        }
        prof_begin(ctx, PNX_K_SCATTER);
        hipLaunchKernelGGL(k_runs_emit, dim3((n_chunks + 3) / 4), dim3(256), 0, ctx->stream,
                           (const uint32_t *)ctx->d_items.p, (const RunChunk *)d_chunks.p, n_chunks,
                           (const uint64_t *)d_offs.p, tile_shift, (uint64_t *)ctx->d_run_start.p,
                           (uint32_t *)ctx->d_run_tile.p, (uint32_t *)ctx->d_run_path.p);
        hipLaunchKernelGGL(k_runs_len, dim3((unsigned)((n_runs + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t *)ctx->d_run_start.p, (const uint32_t *)ctx->d_run_path.p,
                           (const uint64_t *)ctx->d_path_off.p, n_runs, (uint32_t *)ctx->d_run_len.p);
        prof_end(ctx);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            cleanup();
            return ctx->fail(PNX_EHIP, "run index build failed: %s", hipGetErrorString(e));
        }
    }
    cleanup();
    ctx->n_runs = n_runs;
    ctx->n_run_paths = 0;
    for (uint32_t p : general)
        if (cls[p] == 2) ctx->n_run_paths += 1;
    return sort_run_index(ctx);
}

}  // namespace pnx
