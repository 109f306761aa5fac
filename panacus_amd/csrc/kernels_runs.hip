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
// The build runs on the device: counting, classification (run route vs scatter route), a scan for
// the offsets, emission and the sort are a chain of launches on the context's stream; the host
// reads back one number (how many runs) to size their arrays.  It only happens when a pass has
// flagged unclassified general paths, and it is cached with the tile index (the sort by group is
// redone when the visiting order changes).
#include <cstring>

#include <hip/hip_runtime.h>

#include <algorithm>

#include <rocprim/rocprim.hpp>

#include "pnx_context.hpp"
#include "step_chunks.hpp"

namespace pnx {


// flag = this step starts a run (first step of the path, or a different tile than the step before)
__device__ static inline bool run_starts(const uint32_t *__restrict__ items, uint64_t j, uint64_t pstart,
                                         uint32_t cur, uint32_t lane, uint32_t tile_shift) {
    uint32_t prev = __shfl_up(cur, 1);
    if (lane == 0 && j > pstart) prev = items[j - 1];
    return j == pstart || (cur >> tile_shift) != (prev >> tile_shift);
}

// counts[c] = runs that start in chunk c (0 for chunks of paths that need no classification);
// runs_of_path[p] += the same
__global__ __launch_bounds__(256) void k_runs_count(const uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off,
                                                    const uint64_t *__restrict__ chunk_off, uint32_t n_paths, uint64_t n_chunks,
                                                    const uint8_t *__restrict__ path_class, uint32_t tile_shift,
                                                    uint32_t *__restrict__ counts, unsigned long long *__restrict__ runs_of_path) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const RunChunk ch = chunk_of(c, chunk_off, path_off, n_paths);
    if (path_class[ch.path] == 0) {  // tile-monotone: served by the boundary index
        if (lane == 0) counts[c] = 0;
        return;
    }
    uint32_t total = 0;
    for (uint32_t it = 0; it < ch.len; it += 64) {
        const uint64_t j = ch.start + it + lane;
        const bool in = it + lane < ch.len;
        const uint32_t cur = in ? items[j] : 0;
        const bool f = run_starts(items, j, ch.pstart, cur, lane, tile_shift) && in;
        total += (uint32_t)__popcll(__ballot(f));
    }
    if (lane == 0) {
        counts[c] = total;
        atomicAdd(&runs_of_path[ch.path], (unsigned long long)total);
    }
}

// every path that is not tile-monotone: run route (2) if its runs are long enough, else scatter route (3);
// meta[0] = scatter paths, meta[1] = run-route paths
__global__ void k_runs_classify(const uint64_t *__restrict__ path_off, uint32_t n_paths,
                                const unsigned long long *__restrict__ runs_of_path, uint8_t *__restrict__ path_class,
                                unsigned long long *__restrict__ meta) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_paths || path_class[p] == 0) return;
    const uint64_t len = path_off[p + 1] - path_off[p];
    const unsigned long long runs = runs_of_path[p];
    // an empty path has nothing to do on either route
    const uint8_t cls = (len == 0 || runs * RUN_MIN_AVG_LEN <= len || runs <= 64) ? 2 : 3;
    path_class[p] = cls;
    atomicAdd(&meta[cls == 3 ? 0 : 1], 1ull);
}

// counts of the chunks whose path did not end up on the run route are dropped; the rest widen to u64 for the scan
__global__ void k_runs_mask_counts(const uint32_t *__restrict__ counts, const uint64_t *__restrict__ path_off,
                                   const uint64_t *__restrict__ chunk_off, uint32_t n_paths, uint64_t n_chunks,
                                   const uint8_t *__restrict__ path_class, uint64_t *__restrict__ wide) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const RunChunk ch = chunk_of(c, chunk_off, path_off, n_paths);
    wide[c] = path_class[ch.path] == 2 ? counts[c] : 0;
}

// meta[2] = total number of runs = offs[last] + wide[last]
__global__ void k_runs_total(const uint64_t *__restrict__ offs, const uint64_t *__restrict__ wide, uint64_t n_chunks,
                             unsigned long long *__restrict__ meta) {
    if (threadIdx.x == 0 && blockIdx.x == 0) meta[2] = n_chunks ? offs[n_chunks - 1] + wide[n_chunks - 1] : 0;
}

__global__ __launch_bounds__(256) void k_runs_emit(const uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off,
                                                   const uint64_t *__restrict__ chunk_off, uint32_t n_paths, uint64_t n_chunks,
                                                   const uint8_t *__restrict__ path_class, const uint64_t *__restrict__ offs,
                                                   uint32_t tile_shift, uint64_t *__restrict__ r_start,
                                                   uint32_t *__restrict__ r_tile, uint32_t *__restrict__ r_path) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const RunChunk ch = chunk_of(c, chunk_off, path_off, n_paths);
    if (path_class[ch.path] != 2) return;
    uint64_t off = offs[c];
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

// group of every path in the current visiting order (filled over a 0xFF memset = "not visited")
__global__ void k_group_of_path(const uint32_t *__restrict__ ord_path, const uint32_t *__restrict__ ord_group,
                                uint32_t n_ordered, uint32_t *__restrict__ gop) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_ordered) gop[ord_path[k]] = ord_group[k];
}

// ------------------------------------------------------------------------------------------
// host side: the build is a chain of launches on the context's stream with ONE read-back (the
// number of runs, to size their arrays); every scratch buffer lives in the context and only grows
// ------------------------------------------------------------------------------------------
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
    const size_t P = ctx->n_paths ? ctx->n_paths : 1;
    DevBuf &d_gop = ctx->d_rs[0], &d_keys = ctx->d_rs[1], &d_keys2 = ctx->d_rs[2], &d_idx = ctx->d_rs[3], &d_idx2 = ctx->d_rs[4],
           &d_tmp = ctx->d_rs[5];
    if ((rc = ensure(ctx, d_gop, P * 4)) || (rc = ensure(ctx, d_keys, n * 8)) || (rc = ensure(ctx, d_keys2, n * 8)) ||
        (rc = ensure(ctx, d_idx, n * 4)) || (rc = ensure(ctx, d_idx2, n * 4)) || (rc = ensure(ctx, ctx->d_srun_start, n * 8)) ||
        (rc = ensure(ctx, ctx->d_srun_len, n * 4)) || (rc = ensure(ctx, ctx->d_srun_group, n * 4)))
        return rc;
    PNX_HIP(ctx, hipMemsetAsync(d_gop.p, 0xFF, P * 4, ctx->stream));
    if (ctx->n_ordered)
        hipLaunchKernelGGL(k_group_of_path, dim3((ctx->n_ordered + 255) / 256), dim3(256), 0, ctx->stream,
                           (const uint32_t *)ctx->d_ord_path.p, (const uint32_t *)ctx->d_ord_group.p, ctx->n_ordered,
                           (uint32_t *)d_gop.p);
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
    if (rc) return rc;
    if (e != hipSuccess) return ctx->fail(PNX_EHIP, "run index sort failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_runs_gather, dim3(grid), dim3(256), 0, ctx->stream, (const uint64_t *)d_keys2.p,
                       (const uint32_t *)d_idx2.p, (const uint64_t *)ctx->d_run_start.p,
                       (const uint32_t *)ctx->d_run_len.p, n, (uint64_t *)ctx->d_srun_start.p,
                       (uint32_t *)ctx->d_srun_len.p, (uint32_t *)ctx->d_srun_group.p);
    hipLaunchKernelGGL(k_runs_tile_off, dim3((ctx->n_tiles + 1 + 255) / 256), dim3(256), 0, ctx->stream,
                       (const uint64_t *)d_keys2.p, n, ctx->n_tiles, (uint64_t *)ctx->d_run_tile_off.p);
    PNX_HIP(ctx, hipGetLastError());
    PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the passes read the sorted runs from other streams too
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
    ctx->n_runs = 0;
    ctx->n_scatter_paths = 0;
    ctx->n_run_paths = 0;
    if (P == 0) return sort_run_index(ctx);
    if ((rc = ensure_chunk_off(ctx))) return rc;
    const uint64_t n_chunks = ctx->h_chunk_off[P];
    if ((n_chunks + 3) / 4 > 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "too many path chunks for the run index");
    DevBuf &d_counts = ctx->d_rb[0], &d_wide = ctx->d_rb[1], &d_offs = ctx->d_rb[2], &d_rop = ctx->d_rb[3], &d_meta = ctx->d_rb[4],
           &d_scan_tmp = ctx->d_rb[5];
    const size_t nc = n_chunks ? n_chunks : 1;
    if ((rc = ensure(ctx, d_counts, nc * 4)) || (rc = ensure(ctx, d_wide, nc * 8)) || (rc = ensure(ctx, d_offs, nc * 8)) ||
        (rc = ensure(ctx, d_rop, (size_t)P * 8)) || (rc = ensure(ctx, d_meta, 4 * 8)))
        return rc;
    PNX_HIP(ctx, hipMemsetAsync(d_rop.p, 0, (size_t)P * 8, ctx->stream));
    PNX_HIP(ctx, hipMemsetAsync(d_meta.p, 0, 4 * 8, ctx->stream));
    const uint32_t *items = (const uint32_t *)ctx->d_items.p;
    const uint64_t *path_off = (const uint64_t *)ctx->d_path_off.p, *chunk_off = (const uint64_t *)ctx->d_chunk_off.p;
    uint8_t *cls = (uint8_t *)ctx->d_path_class.p;
    prof_begin(ctx, PNX_K_SCATTER);
    if (n_chunks)
        hipLaunchKernelGGL(k_runs_count, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, ctx->stream, items, path_off, chunk_off, P,
                           n_chunks, (const uint8_t *)cls, tile_shift, (uint32_t *)d_counts.p, (unsigned long long *)d_rop.p);
    hipLaunchKernelGGL(k_runs_classify, dim3((P + 255) / 256), dim3(256), 0, ctx->stream, path_off, P,
                       (const unsigned long long *)d_rop.p, cls, (unsigned long long *)d_meta.p);
    if (n_chunks) {
        hipLaunchKernelGGL(k_runs_mask_counts, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const uint32_t *)d_counts.p, path_off, chunk_off, P, n_chunks, (const uint8_t *)cls, (uint64_t *)d_wide.p);
        size_t tmp_bytes = 0;
        hipError_t e = rocprim::exclusive_scan(nullptr, tmp_bytes, (uint64_t *)d_wide.p, (uint64_t *)d_offs.p, (uint64_t)0, (size_t)n_chunks,
                                               rocprim::plus<uint64_t>(), ctx->stream);
        if (e == hipSuccess && (rc = ensure(ctx, d_scan_tmp, tmp_bytes)) == PNX_OK)
            e = rocprim::exclusive_scan(d_scan_tmp.p, tmp_bytes, (uint64_t *)d_wide.p, (uint64_t *)d_offs.p, (uint64_t)0, (size_t)n_chunks,
                                        rocprim::plus<uint64_t>(), ctx->stream);
        if (rc) return rc;
        if (e != hipSuccess) return ctx->fail(PNX_EHIP, "run index scan failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k_runs_total, dim3(1), dim3(64), 0, ctx->stream, (const uint64_t *)d_offs.p, (const uint64_t *)d_wide.p, n_chunks,
                       (unsigned long long *)d_meta.p);
    prof_end(ctx);
    // the one read-back: how many runs (their arrays must be sized), how many paths on either route
    unsigned long long meta[4] = {0, 0, 0, 0};
    PNX_HIP(ctx, hipMemcpyAsync(meta, d_meta.p, sizeof meta, hipMemcpyDeviceToHost, ctx->stream));
    PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t n_runs = meta[2];
    if (n_runs >= 0xFFFFFFFFull) return ctx->fail(PNX_ELIMIT, "more than 2^32-1 runs in the run index");
    ctx->n_scatter_paths = (uint32_t)meta[0];
    ctx->n_run_paths = (uint32_t)meta[1];
    if (n_runs) {
        if ((rc = ensure(ctx, ctx->d_run_start, n_runs * 8)) || (rc = ensure(ctx, ctx->d_run_len, n_runs * 4)) ||
            (rc = ensure(ctx, ctx->d_run_tile, n_runs * 4)) || (rc = ensure(ctx, ctx->d_run_path, n_runs * 4)))
            return rc;
        prof_begin(ctx, PNX_K_SCATTER);
        hipLaunchKernelGGL(k_runs_emit, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, ctx->stream, items, path_off, chunk_off, P,
                           n_chunks, (const uint8_t *)cls, (const uint64_t *)d_offs.p, tile_shift, (uint64_t *)ctx->d_run_start.p,
                           (uint32_t *)ctx->d_run_tile.p, (uint32_t *)ctx->d_run_path.p);
        hipLaunchKernelGGL(k_runs_len, dim3((unsigned)((n_runs + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t *)ctx->d_run_start.p, (const uint32_t *)ctx->d_run_path.p, path_off, n_runs,
                           (uint32_t *)ctx->d_run_len.p);
        prof_end(ctx);
        PNX_HIP(ctx, hipGetLastError());
    }
    ctx->n_runs = n_runs;
    return sort_run_index(ctx);
}

}  // namespace pnx
