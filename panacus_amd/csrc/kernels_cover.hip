// kernels_cover.hip -- the STEP ROUTES: coverage counting straight over (packed) steps with a tile boundary index, round 2's
// kernels.  Since round 4 a CROSS-CHECK MODULE (libpanacus_hip_steps.so), not part of the product library: the product computes
// coverage over path rows (kernels_rows.hip) or in one read of the steps (kernels_band.hip); these kernels compute the same
// results in a completely different way and the tests hold the product against them (PNX_CFG_COVER_VARIANT 0 / 1 / 2 loads
// the module on demand; a system without it gets a clear error).
//
// Replaces, on the device, the reference's serial loops
//   AbacusByTotal::coverage            src/graph_broker/abacus.rs:719-744
//   AbacusByTotal::construct_hist      src/graph_broker/abacus.rs:746-762
//   AbacusByTotal::construct_hist_bps  src/graph_broker/abacus.rs:764-787
// and produces the bit-packed presence matrix that stands in for AbacusByGroup's (r, c)
// (abacus.rs:859-986).
//
// Design (HBM-bound integer set work; no MFMA):
//   * the item id space is cut into tiles of WT*2048 ids; a tile is owned by ONE WAVE (or, split,
//     by the SPLIT waves of one workgroup, each with a group-aligned part of the visiting order)
//     for the whole kernel, so no global atomics and no inter-workgroup traffic exist on the fast
//     path; the split waves meet once, at the very end, to add their counters;
//   * K0 finds, for every path and every tile boundary inside the path's id range, where the
//     path's steps cross the boundary (value-interpolating sector search; exact for
//     tile-monotone paths, verified by K1);
//   * K1 keeps 64 entries of the visiting order per wave (segment start, length, group), visits
//     the non-empty segments and the group changes only, streams each (path, tile) segment of the
//     CSR exactly once with 16 B/lane non-temporal loads, ORs presence bits into a per-wave 256 B
//     LDS bitmap (ds_or_b32; dedupes repeated visits inside a group for free), and at every
//     group change folds the bitmap into bit-sliced vertical counters held in registers
//     (carry-save ripple adder: NPL planes);
//   * at the end the counters are unpacked to the u32 coverage vector with coalesced
//     stores; K2 turns it into the (optionally bp-weighted) histogram in LDS;
//   * paths that are not tile-monotone are detected, not assumed away: nearly monotone ones are
//     cut into per-tile runs (kernels_runs.hip) that the tile's wave consumes at flush time,
//     short-run ones (edge ids) take the scatter route (global atomicOr into the presence
//     matrix, merged by K1 at flush time); the pass is re-run when a violation is first seen.
#include <algorithm>
#include <cstdlib>
#include <cstring>  // rocprim's texture iterator calls memset
#include <type_traits>
#include <vector>

#include <hip/hip_runtime.h>

#include <rocprim/rocprim.hpp>

#include "pnx_context.hpp"
#include "step_chunks.hpp"

namespace pnx {

// ------------------------------------------------------------------------------------------
// step preparation: once per upload, one streaming pass over the steps
//   * steps12: the step ids modulo 4096, 12 bits each, 8 steps per 12 bytes (k_pack12).  A coverage wave owns one
//     tile of 2048 (or 4096) ids, so inside its tile a step needs 11 (12) bits; the coverage kernel streams these
//     packed steps -- 1.5 bytes per step against the 4 of the u32 ItemTable, which stays resident for everything
//     that needs whole ids (the index search, the run index, the scatter route, the read-back);
//   * an EXACT classification of every path: tile-monotone (the sequence of 2048-id tiles of its
//     steps never turns around; any disorder inside a tile is fine) or not.  A packed step cannot
//     tell which tile it came from, so the coverage kernel can no longer verify the boundary index
//     step by step as it did in round 1 -- instead nothing is left to verify: for a tile-monotone
//     path the boundary search is exact by construction, and every other path goes to the run /
//     scatter routes before the first pass looks at it.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prepare_steps(const uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off,
                                                       const uint64_t *__restrict__ chunk_off, uint32_t n_paths, uint64_t n_chunks,
                                                       uint8_t *__restrict__ path_dir, uint32_t *__restrict__ path_changes) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const RunChunk ch = chunk_of(c, chunk_off, path_off, n_paths);
    uint32_t dir = 0;  // bit 0: a step enters a higher tile than its predecessor, bit 1: a lower one
    uint32_t changes = 0;  // steps that lie in another tile than their predecessor
    for (uint32_t it = 0; it < ch.len; it += 64) {
        const uint64_t j = ch.start + it + lane;
        const bool in = it + lane < ch.len;
        const uint32_t cur = in ? items[j] : 0;
        uint32_t prev = __shfl_up(cur, 1);
        if (lane == 0 && in && j > ch.pstart) prev = items[j - 1];
        if (in) {
            if (j > ch.pstart) {
                const uint32_t tc = cur / BLOCK_ITEMS, tp = prev / BLOCK_ITEMS;
                dir |= (tc > tp ? 1u : 0u) | (tc < tp ? 2u : 0u);
                changes += tc != tp ? 1u : 0u;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        dir |= __shfl_xor(dir, o);
        changes += __shfl_xor(changes, o);
    }
    if (lane == 0 && dir) {
        atomicOr(reinterpret_cast<unsigned int *>(path_dir) + (ch.path >> 2), dir << ((ch.path & 3u) * 8u));
        atomicAdd(path_changes + ch.path, changes);
    }
}

// ---- shuffled paths: sorted by id, once (PNX_CFG_SORT_SHUFFLED) ---------------------------------
// slot s = the s-th sorted path; compact index i runs over the steps of all of them
__device__ static inline uint32_t slot_of(const uint64_t *__restrict__ coff, uint32_t n_slots, uint64_t i) {
    uint32_t lo = 0, hi = n_slots;  // last s with coff[s] <= i
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (coff[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}
__global__ void k_sort_keys(const uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off, const uint64_t *__restrict__ coff,
                            const uint32_t *__restrict__ slot_path, uint32_t n_slots, uint64_t total, uint32_t id_bits,
                            uint64_t *__restrict__ keys, uint32_t *__restrict__ unsorted) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t s = slot_of(coff, n_slots, i);
    const uint32_t v = items[path_off[slot_path[s]] + (i - coff[s])];
    unsorted[i] = v;
    keys[i] = ((uint64_t)s << id_bits) | v;
}
// sorted slot-major, every slot keeps its range: position i still belongs to slot_of(i)
__global__ void k_sort_write_back(uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off, const uint64_t *__restrict__ coff,
                                  const uint32_t *__restrict__ slot_path, uint32_t n_slots, uint64_t total, uint32_t id_bits,
                                  const uint64_t *__restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t s = slot_of(coff, n_slots, i);
    items[path_off[slot_path[s]] + (i - coff[s])] = (uint32_t)(keys[i] & ((1ull << id_bits) - 1ull));
}
__global__ void k_restore_order(uint32_t *__restrict__ items_copy, const uint64_t *__restrict__ path_off, const uint64_t *__restrict__ coff,
                                const uint32_t *__restrict__ slot_path, uint32_t n_slots, uint64_t total,
                                const uint32_t *__restrict__ unsorted) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t s = slot_of(coff, n_slots, i);
    items_copy[path_off[slot_path[s]] + (i - coff[s])] = unsorted[i];
}

int restore_step_order(pnx_ctx *ctx, uint32_t *d_items_copy) {
    if (!ctx->n_sorted_paths || !ctx->n_unsorted) return PNX_OK;
    hipLaunchKernelGGL(k_restore_order, dim3((unsigned)((ctx->n_unsorted + 255) / 256)), dim3(256), 0, ctx->stream, d_items_copy,
                       (const uint64_t *)ctx->d_path_off.p, (const uint64_t *)ctx->d_sorted_coff.p, (const uint32_t *)ctx->d_sorted_path.p,
                       ctx->n_sorted_paths, ctx->n_unsorted, (const uint32_t *)ctx->d_unsorted.p);
    PNX_HIP(ctx, hipGetLastError());
    return PNX_OK;
}

// sorts the steps of the listed paths in place (ids ascending), keeping their original order in d_unsorted
static int sort_paths_in_place(pnx_ctx *ctx, const std::vector<uint32_t> &paths) {
    const uint32_t n_slots = (uint32_t)paths.size();
    std::vector<uint64_t> coff((size_t)n_slots + 1, 0);
    for (uint32_t s = 0; s < n_slots; ++s) coff[s + 1] = coff[s] + (ctx->h_path_off[paths[s] + 1] - ctx->h_path_off[paths[s]]);
    const uint64_t total = coff[n_slots];
    if ((total + 255) / 256 > 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "too many steps on shuffled paths to sort them");
    uint32_t id_bits = 1, slot_bits = 1;
    while ((1ull << id_bits) <= (uint64_t)ctx->n_items) ++id_bits;
    while ((1ull << slot_bits) < (uint64_t)n_slots) ++slot_bits;
    int rc;
    if ((rc = ensure(ctx, ctx->d_unsorted, total * 4 + 16)) || (rc = ensure(ctx, ctx->d_sorted_coff, ((size_t)n_slots + 1) * 8)) ||
        (rc = ensure(ctx, ctx->d_sorted_path, (size_t)n_slots * 4)))
        return rc;
    PNX_HIP(ctx, hipMemcpyAsync(ctx->d_sorted_coff.p, coff.data(), coff.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    PNX_HIP(ctx, hipMemcpyAsync(ctx->d_sorted_path.p, paths.data(), (size_t)n_slots * 4, hipMemcpyHostToDevice, ctx->stream));
    DevBuf k_in, k_out, tmp;
    auto done = [&](int r) {
        release(k_in);
        release(k_out);
        release(tmp);
        return r;
    };
    if ((rc = ensure(ctx, k_in, total * 8 + 16)) || (rc = ensure(ctx, k_out, total * 8 + 16))) return done(rc);
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(k_sort_keys, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_items.p, (const uint64_t *)ctx->d_path_off.p,
                       (const uint64_t *)ctx->d_sorted_coff.p, (const uint32_t *)ctx->d_sorted_path.p, n_slots, total, id_bits,
                       (uint64_t *)k_in.p, (uint32_t *)ctx->d_unsorted.p);
    size_t bytes = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, bytes, (const uint64_t *)k_in.p, (uint64_t *)k_out.p, (size_t)total, 0u,
                                            id_bits + slot_bits, ctx->stream);
    if (e == hipSuccess && !(rc = ensure(ctx, tmp, bytes ? bytes : 8)))
        e = rocprim::radix_sort_keys(tmp.p, bytes, (const uint64_t *)k_in.p, (uint64_t *)k_out.p, (size_t)total, 0u, id_bits + slot_bits,
                                     ctx->stream);
    if (rc) return done(rc);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_sort_write_back, dim3(grid), dim3(256), 0, ctx->stream, (uint32_t *)ctx->d_items.p,
                           (const uint64_t *)ctx->d_path_off.p, (const uint64_t *)ctx->d_sorted_coff.p,
                           (const uint32_t *)ctx->d_sorted_path.p, n_slots, total, id_bits, (const uint64_t *)k_out.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // coff / paths are the caller's, the key buffers go away
    if (e != hipSuccess) return done(ctx->fail(PNX_EHIP, "sorting the shuffled paths failed: %s", hipGetErrorString(e)));
    ctx->n_sorted_paths = n_slots;
    ctx->n_unsorted = total;
    ctx->spans_valid = false;        // first / last step of those paths changed
    ctx->order_normalized = false;
    return done(PNX_OK);
}

// steps12: 8 consecutive steps (by global step index) = 8 x 12 bits = three dwords; step e of a group sits at bit 12 e.
// One thread per group.  The last group of the table is padded with zeros.
__global__ void k_pack12(const uint32_t *__restrict__ items, uint64_t n_steps, uint32_t *__restrict__ steps12) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t j0 = g * 8;
    if (j0 >= n_steps) return;
    uint32_t v[8];
    if (j0 + 8 <= n_steps) {
        const uint4 a = *reinterpret_cast<const uint4 *>(items + j0), b = *reinterpret_cast<const uint4 *>(items + j0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = j0 + e < n_steps ? items[j0 + e] : 0u;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] &= 4095u;
    uint32_t *out = steps12 + g * 3;
    out[0] = v[0] | (v[1] << 12) | (v[2] << 24);
    out[1] = (v[2] >> 8) | (v[3] << 4) | (v[4] << 16) | (v[5] << 28);
    out[2] = (v[5] >> 4) | (v[6] << 8) | (v[7] << 20);
}

__global__ void k_mono_class(uint8_t *__restrict__ path_dir, uint32_t n_paths) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_paths) path_dir[p] = path_dir[p] == 3 ? 1 : 0;  // both directions seen: not tile-monotone
}

int prepare_steps(pnx_ctx *ctx) {
    if (ctx->steps_prepared) return PNX_OK;
    const uint32_t P = ctx->n_paths;
    int rc;
    const size_t mono_bytes = ((size_t)(P ? P : 1) + 3) / 4 * 4;
    const uint64_t n_groups8 = (ctx->n_steps + 7) / 8;
    if ((rc = ensure(ctx, ctx->d_steps12, n_groups8 * 12 + 64)) || (rc = ensure(ctx, ctx->d_path_mono, mono_bytes)))
        return rc;
    DevBuf d_changes;
    if ((rc = ensure(ctx, d_changes, (size_t)(P ? P : 1) * 4))) return rc;
    // a graph that is lent out (pnx_share_csr) is prepared by its owner before the first borrower reads it
    bool may_sort = ctx->sort_shuffled && !ctx->d_items.borrowed && ctx->n_sorted_paths == 0;
    for (int round = 0; round < 2; ++round) {
        hipError_t e = hipMemsetAsync(ctx->d_path_mono.p, 0, mono_bytes, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_changes.p, 0, (size_t)(P ? P : 1) * 4, ctx->stream);
        if (e != hipSuccess) {
            release(d_changes);
            return ctx->fail(PNX_EHIP, "prepare_steps: %s", hipGetErrorString(e));
        }
        if (!P || !ctx->n_steps) break;
        if ((rc = ensure_chunk_off(ctx))) {
            release(d_changes);
            return rc;
        }
        const uint64_t n_chunks = ctx->h_chunk_off[P];
        if ((n_chunks + 3) / 4 > 0x7FFFFFFFull) {
            release(d_changes);
            return ctx->fail(PNX_ELIMIT, "too many path chunks");
        }
        hipLaunchKernelGGL(k_prepare_steps, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, ctx->stream,
                           (const uint32_t *)ctx->d_items.p, (const uint64_t *)ctx->d_path_off.p,
                           (const uint64_t *)ctx->d_chunk_off.p, P, n_chunks,
                           (uint8_t *)ctx->d_path_mono.p, (uint32_t *)d_changes.p);
        hipLaunchKernelGGL(k_mono_class, dim3((P + 255) / 256), dim3(256), 0, ctx->stream, (uint8_t *)ctx->d_path_mono.p, P);
        if (!may_sort) break;
        // paths that jump between tiles at random: more than one tile change per RUN_MIN_AVG_LEN steps (what the run
        // index would refuse, kernels_runs.hip).  They are sorted once, then this pass runs again over the new order.
        std::vector<uint8_t> mono(P);
        std::vector<uint32_t> changes(P), shuffled;
        e = hipMemcpyAsync(mono.data(), ctx->d_path_mono.p, P, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(changes.data(), d_changes.p, (size_t)P * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            release(d_changes);
            return ctx->fail(PNX_EHIP, "prepare_steps: %s", hipGetErrorString(e));
        }
        for (uint32_t p = 0; p < P; ++p) {
            const uint64_t len = ctx->h_path_off[p + 1] - ctx->h_path_off[p], runs = (uint64_t)changes[p] + 1;
            if (mono[p] && runs * RUN_MIN_AVG_LEN > len && runs > 64) shuffled.push_back(p);
        }
        may_sort = false;
        if (shuffled.empty()) break;
        if ((rc = sort_paths_in_place(ctx, shuffled))) {
            release(d_changes);
            return rc;
        }
    }
    release(d_changes);
    if (n_groups8) {  // the packed steps of the FINAL order (after a sort of shuffled paths)
        if ((n_groups8 + 255) / 256 > 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "too many steps to pack");
        hipLaunchKernelGGL(k_pack12, dim3((unsigned)((n_groups8 + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_items.p,
                           ctx->n_steps, (uint32_t *)ctx->d_steps12.p);
    }
    PNX_HIP(ctx, hipGetLastError());
    PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // once per upload; the passes read this from other streams too
    ctx->steps_prepared = true;
    return PNX_OK;
}

// ------------------------------------------------------------------------------------------
// K0: tile boundary index (two passes, see k_tile_index_coarse / k_tile_index_fine).  The
// boundary of tile t in path p = path_off[p] + (#steps of p "before" tile t), where "before"
// means id < t*tile_items for an ascending path and id >= t*tile_items for a descending one
// (direction = first step vs last step).  For a tile-monotone path the steps of tile t are
// exactly [min(B[t],B[t+1]), max(B[t],B[t+1])).
// ------------------------------------------------------------------------------------------
template <bool ASC>
__device__ static inline bool before_key(uint32_t v, uint64_t key) {
    return ASC ? ((uint64_t)v < key) : ((uint64_t)v >= key);
}

// number of leading elements of a[lo..hi) (absolute indices) that are "before" key, + lo
template <bool ASC>
__device__ static inline uint64_t bsearch_before(const uint32_t *__restrict__ a, uint64_t lo, uint64_t hi, uint64_t key) {
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (before_key<ASC>(a[mid], key)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Boundary of one tile inside the bracket [lo, hi) of a path (absolute step indices): the number
// of steps "before" key, + lo.  `frac` is the share of the bracket that lies before the boundary.
// First guess by position.  Every probe reads one aligned window of W ids.  If the boundary lies
// inside the window we are done; otherwise the smallest / largest id seen tells how many ids are
// still missing, and the bracket's own density (steps per id) turns that into the next guess -- the
// error shrinks from ~sqrt(bracket) to ~sqrt(error) per probe.
// Measured, not guessed (DESIGN_DEADENDS.md section 8): the index is bound by the RATE of L1->L2 requests --
// 44 G requests/s against the ~50 G/s that benchmarks/micro/random_sector_rate.hip reaches with wave-
// local 64-byte probes -- not by round trips: 32-id windows for the later probes (fewer rounds per
// wave, more lines per probe) and windows centred on the guess (fewer probes, two lines each) both
// lost to plain aligned 16-id sectors.
// Whatever the data, [lo, hi) only ever shrinks around the answer of a monotone path and the
// search ends in a binary search of what is left, so the result is always inside the bracket.
template <int W, bool CENTERED>
__device__ static inline bool probe_window(const uint32_t *__restrict__ items, uint64_t &lo, uint64_t &hi, uint64_t &pos,
                                           bool asc, uint32_t key32, float dens, uint64_t &found) {
    if (pos >= hi) pos = hi - 1;
    if (pos < lo) pos = lo;
    // the positional first guess is off by a few sectors anyway: an aligned window; the later guesses
    // are good to a few steps, and an aligned window would still lose the boundary whenever the guess
    // sits near a window edge: those windows are centred on the guess (16-byte aligned start)
    const uint64_t s0 = CENTERED ? (pos > W / 2 ? (pos - W / 2) & ~(uint64_t)3 : 0) : pos & ~(uint64_t)(W - 1);
    const uint64_t a0 = s0 > lo ? s0 : lo, a1 = s0 + W < hi ? s0 + W : hi;  // [a0, a1) of the window
    const uint32_t o0 = (uint32_t)(a0 - s0), o1 = (uint32_t)(a1 - s0);      // the same, relative to s0
    const uint4 *win = reinterpret_cast<const uint4 *>(items + s0);
    uint4 x[W / 4];
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
        // only the quads that overlap [a0, a1) are read (the window may stick out of the path)
        x[q] = make_uint4(0, 0, 0, 0);
        if ((uint32_t)(4 * q + 4) > o0 && (uint32_t)(4 * q) < o1) x[q] = win[q];
    }
    uint32_t cnt = 0, vmin = 0xFFFFFFFFu, vmax = 0;
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
        const uint32_t v[4] = {x[q].x, x[q].y, x[q].z, x[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t i = (uint32_t)(4 * q + e);
            const bool in = i >= o0 && i < o1;
            const bool bef = asc ? v[e] < key32 : v[e] >= key32;
            cnt += (in && bef) ? 1u : 0u;
            vmin = in && v[e] < vmin ? v[e] : vmin;
            vmax = in && v[e] > vmax ? v[e] : vmax;
        }
    }
    if (cnt == 0) {  // everything here is at or past the boundary: it lies at or left of a0
        hi = a0;
        // asc: all ids >= key, the nearest is the smallest; desc: all ids < key, the nearest is the largest
        const float gap = asc ? (float)vmin - (float)key32 : (float)key32 - (float)vmax;
        const uint64_t back = (uint64_t)(gap > 0.f ? gap * dens : 0.f) + 1;
        pos = a0 > lo + back ? a0 - back : lo;
        return false;
    }
    if (cnt == o1 - o0) {  // everything here is before it
        lo = a1;
        const float gap = asc ? (float)key32 - (float)vmax : (float)vmin - (float)key32;
        pos = a1 + (uint64_t)(gap > 0.f ? gap * dens : 0.f);
        return false;
    }
    found = a0 + cnt;
    return true;
}

template <int W1, int W2, bool CENTER2 = false>
__device__ static inline uint64_t locate_boundary(const uint32_t *__restrict__ items, uint64_t lo, uint64_t hi,
                                                  bool asc, uint64_t key, double frac /* of the bracket before the boundary */,
                                                  float dens /* steps per id inside the bracket */, int max_probes) {
    if (lo >= hi) return lo;
    uint64_t pos = lo + (uint64_t)((double)(hi - lo) * frac);
    const uint32_t key32 = key > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)key;  // ids are < 2^32 - 1
    uint64_t found = 0;
    if (probe_window<W1, false>(items, lo, hi, pos, asc, key32, dens, found)) return found;
    for (int it = 1; it < max_probes && lo < hi; ++it)
        if (probe_window<W2, CENTER2>(items, lo, hi, pos, asc, key32, dens, found)) return found;
    if (lo < hi) lo = asc ? bsearch_before<true>(items, lo, hi, key) : bsearch_before<false>(items, lo, hi, key);
    return lo;
}

// The index is SPARSE: a path only has boundaries for the tiles between its first and its last
// id (TileIdx: first tile, number of tiles spanned, offset of its row).  An assembly-shaped
// pangenome has thousands of contig paths that each span a few per cent of the id space; a dense
// paths x tiles table would be 20-100 times larger there, and as slow to fill.  Row of path p:
// off[p] + j, j = 0..span, = boundary of tile first + j; the two ends are the ends of the path, so
// the segments of a path always partition it, whatever its ids do in between.
struct TileIdx {
    const uint64_t *B;
    const uint64_t *off;     // n_paths + 1
    const uint32_t *tfirst;  // n_paths
    const uint32_t *tspan;   // n_paths, 0 = empty path
};

// the same per entry of the visiting order (filled by k_count_general for every pass)
struct OrdIdx {
    uint32_t *tfirst, *tspan;  // n_ordered each
    uint64_t *off;             // n_ordered
    uint32_t *win_lo, *win_hi; // per 64 entries: the tiles [lo, hi) reached by their tile-route paths
};

__global__ void k_path_spans(const uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off,
                             uint32_t n_paths, uint32_t tile_items, uint32_t *__restrict__ tfirst,
                             uint32_t *__restrict__ tspan) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_paths) return;
    const uint64_t s = path_off[p], e = path_off[p + 1];
    if (e == s) {
        tfirst[p] = 0;
        tspan[p] = 0;
        return;
    }
    const uint32_t a = items[s], b = items[e - 1];
    const uint32_t t0 = (a < b ? a : b) / tile_items, t1 = (a < b ? b : a) / tile_items;
    tfirst[p] = t0;
    tspan[p] = t1 - t0 + 1;
}

// Which (path, row position) a thread of the index kernels works on.  Dense graphs (every path
// spans about as many tiles as the longest one) number their workgroups path-major: blockIdx =
// p * bpp + chunk.  Skewed graphs (a few paths span the whole id space, thousands span a few
// tiles) would launch paths x longest-span threads that way, so there (bpp == 0) a thread is one
// entry of the sparse index and finds its path by binary search in the row offsets.
__device__ static inline bool index_slot(const TileIdx &ix, uint32_t bpp, uint32_t n_paths, uint64_t n_entries,
                                         uint32_t &p, uint32_t &j) {
    if (bpp) {
        p = blockIdx.x / bpp;
        j = (blockIdx.x % bpp) * blockDim.x + threadIdx.x;
        return true;
    }
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return false;
    uint32_t lo = 0, hi = n_paths;  // last p with off[p] <= e
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ix.off[mid] <= e) lo = mid; else hi = mid;
    }
    p = lo;
    j = (uint32_t)(e - ix.off[lo]);
    return true;
}

// K0 pass A: every `coarse`-th boundary of a path's row (and the last one), located inside the
// whole path.  Workgroups are numbered path-major: blockIdx = p * bpp + chunk.
template <int W1, int W2>
__global__ void k_tile_index_coarse(const uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off,
                                    uint32_t bpp, uint32_t n_paths, uint64_t n_entries, uint32_t tile_items,
                                    uint32_t coarse, uint64_t *__restrict__ B, TileIdx ix,
                                    uint8_t *__restrict__ path_class, const uint8_t *__restrict__ path_mono) {
    uint32_t p, c;
    if (!index_slot(ix, bpp, n_paths, n_entries, p, c)) return;
    const uint32_t span = ix.tspan[p];
    if (c == 0) path_class[p] = path_mono[p];  // 0 tile-monotone (exact, prepare_steps), 1 = to be put on the run / scatter route
    uint32_t j;
    if (bpp) {  // slot c = the c-th coarse boundary
        const uint32_t n_coarse = (span + coarse - 1) / coarse + 1;  // j = 0, c, 2c, ..., span
        if (c >= n_coarse) return;
        j = c * coarse;
        if (j > span) j = span;
    } else {    // slot c = row position: only the coarse ones are filled here
        j = c;
        if (j % coarse != 0 && j != span) return;
    }
    const uint64_t s = path_off[p], e = path_off[p + 1];
    uint64_t *out = B + ix.off[p] + j;
    if (e == s) {
        *out = s;
        return;
    }
    const bool asc = items[s] <= items[e - 1];
    if (j == 0) *out = asc ? s : e;
    else if (j == span) *out = asc ? e : s;
    else {
        // the path's own id range gives the first guess and the density
        const uint64_t key = (uint64_t)(ix.tfirst[p] + j) * tile_items;
        const uint64_t id_a = items[s], id_b = items[e - 1];  // first and last id (asc: smallest, largest)
        const uint64_t id_min = asc ? id_a : id_b, id_max = asc ? id_b : id_a;
        if (key <= id_min) *out = asc ? s : e;       // no step lies before the tile (asc) / all do (desc)
        else if (key > id_max) *out = asc ? e : s;
        else {
            const double range = (double)(id_max - id_min + 1);
            const double below = (double)(key - id_min) / range;  // share of the ids that are < key
            *out = locate_boundary<W1, W2>(items, s, e, asc, key, asc ? below : 1.0 - below, (float)((double)(e - s) / range), 6);
        }
    }
}

// K0 pass B: the boundaries in between, inside the bracket of the two enclosing coarse
// boundaries.  For a path that is not tile-monotone the result is still a monotone sequence
// inside the bracket, so the (path, tile) segments always partition the path; K1's per-step
// in-tile check then catches every misplaced step.
template <int W1, int W2>
__global__ void k_tile_index_fine(const uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off,
                                  uint32_t bpp, uint32_t n_paths, uint64_t n_entries, uint32_t tile_items,
                                  uint32_t coarse, uint64_t *__restrict__ B, TileIdx ix, uint8_t *path_class) {
    uint32_t p, j;
    if (!index_slot(ix, bpp, n_paths, n_entries, p, j)) return;
    const uint32_t span = ix.tspan[p];
    if (j % coarse == 0 || j >= span) return;  // done by the coarse level
    const uint32_t j0 = j - j % coarse;
    const uint32_t j1 = j0 + coarse < span ? j0 + coarse : span;
    const uint64_t s = path_off[p], e = path_off[p + 1];
    uint64_t *row = B + ix.off[p];
    if (e == s) {
        row[j] = s;
        return;
    }
    const bool asc = items[s] <= items[e - 1];
    uint64_t lo = row[j0], hi = row[j1];  // ascending: lo <= hi ; descending: lo >= hi
    if (!asc) {
        uint64_t tmp = lo;
        lo = hi;
        hi = tmp;
    }
    if (lo > hi) {  // coarse boundaries out of order: not tile-monotone
        path_class[p] = 1;
        row[j] = asc ? lo : hi;
        return;
    }
    row[j] = locate_boundary<W1, W2>(items, lo, hi, asc, (uint64_t)(ix.tfirst[p] + j) * tile_items,
                             (double)(asc ? j - j0 : j1 - j) / (double)(j1 - j0),
                             (float)(hi - lo) / ((float)(j1 - j0) * (float)tile_items), 4);
}

// No separate monotonicity check of the finished rows is needed: the two ends of a row are the
// ends of the path, so the union of [min, max) over consecutive boundaries covers every step at
// least once, and K1 verifies every step it consumes against its tile -- a path whose boundaries
// are out of order is caught there (a step lands in a foreign tile) or is harmlessly OR-ed twice.

// tile spans of the paths and the row offsets of the sparse index: once per graph / tile size
static int ensure_path_spans(pnx_ctx *ctx) {
    if (ctx->spans_valid) return PNX_OK;
    const uint32_t P = ctx->n_paths;
    const uint32_t tile_items = ctx->tile_blocks * BLOCK_ITEMS;
    int rc;
    if ((rc = ensure(ctx, ctx->d_tfirst, (P ? P : 1) * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->d_tspan, (P ? P : 1) * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->d_idx_off, ((size_t)P + 1) * sizeof(uint64_t)))) return rc;
    std::vector<uint32_t> span(P);
    std::vector<uint64_t> off((size_t)P + 1, 0);
    if (P) {
        hipLaunchKernelGGL(k_path_spans, dim3((P + 255) / 256), dim3(256), 0, ctx->stream,
                           (const uint32_t *)ctx->d_items.p, (const uint64_t *)ctx->d_path_off.p, P, tile_items,
                           (uint32_t *)ctx->d_tfirst.p, (uint32_t *)ctx->d_tspan.p);
        ctx->h_tfirst.resize(P);
        PNX_HIP(ctx, hipMemcpyAsync(span.data(), ctx->d_tspan.p, (size_t)P * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        PNX_HIP(ctx, hipMemcpyAsync(ctx->h_tfirst.data(), ctx->d_tfirst.p, (size_t)P * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    uint32_t mx = 0;
    for (uint32_t p = 0; p < P; ++p) {
        off[p + 1] = off[p] + span[p] + 1;
        mx = span[p] > mx ? span[p] : mx;
    }
    ctx->max_span = mx;
    ctx->idx_entries = off[P];
    PNX_HIP(ctx, hipMemcpyAsync(ctx->d_idx_off.p, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `off` is a local
    ctx->spans_valid = true;
    return PNX_OK;
}

// Within a group the order of the paths does not matter for any result (coverage counts groups,
// the presence matrix and the runs are per group), so the paths of every group are put in the
// order of their first tile once per (graph, order): 64 consecutive entries then reach a narrow
// band of tiles, and a coverage wave can skip the windows whose band misses its tile.
static int normalize_order(pnx_ctx *ctx) {
    if (ctx->order_normalized) return PNX_OK;
    const size_t n = ctx->h_ord_path.size();
    if (n == ctx->n_ordered && n > 1 && ctx->h_tfirst.size() == ctx->n_paths) {
        bool changed = false;
        size_t a = 0;
        while (a < n) {
            size_t b = a + 1;
            while (b < n && ctx->h_ord_group[b] == ctx->h_ord_group[a]) ++b;
            auto key_less = [&](uint32_t x, uint32_t y) {
                return ctx->h_tfirst[x] != ctx->h_tfirst[y] ? ctx->h_tfirst[x] < ctx->h_tfirst[y] : x < y;
            };
            if (b - a > 1 && !std::is_sorted(ctx->h_ord_path.begin() + a, ctx->h_ord_path.begin() + b, key_less)) {
                std::sort(ctx->h_ord_path.begin() + a, ctx->h_ord_path.begin() + b, key_less);
                changed = true;
            }
            a = b;
        }
        if (changed) {
            PNX_HIP(ctx, hipMemcpyAsync(ctx->d_ord_path.p, ctx->h_ord_path.data(), n * sizeof(uint32_t),
                                        hipMemcpyHostToDevice, ctx->stream));
            PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    ctx->order_normalized = true;
    return PNX_OK;
}

static OrdIdx ord_idx_view(const pnx_ctx *ctx) {
    const Ticket *t = ctx->cur;
    return OrdIdx{(uint32_t *)t->d_ord_tfirst.p, (uint32_t *)t->d_ord_tspan.p, (uint64_t *)t->d_ord_off.p,
                  (uint32_t *)t->d_win_lo.p, (uint32_t *)t->d_win_hi.p};
}

// the boundary table of the pass being enqueued: the context's when it is kept across passes, the
// pass's own when every pass rebuilds it (the next pass then builds its table beside this pass's K1)
static DevBuf &tile_idx_buf(pnx_ctx *ctx) { return ctx->cache_index ? ctx->d_tile_idx : ctx->cur->d_tile_idx_own; }

static TileIdx tile_idx_view(pnx_ctx *ctx) {
    return TileIdx{(const uint64_t *)tile_idx_buf(ctx).p, (const uint64_t *)ctx->d_idx_off.p,
                   (const uint32_t *)ctx->d_tfirst.p, (const uint32_t *)ctx->d_tspan.p};
}

int launch_tile_index(pnx_ctx *ctx) {
    const uint32_t tile_items = ctx->tile_blocks * BLOCK_ITEMS;
    int rc;
    if ((rc = prepare_steps(ctx))) return rc;
    if ((rc = ensure_path_spans(ctx))) return rc;
    if ((rc = ensure(ctx, tile_idx_buf(ctx), (ctx->idx_entries ? ctx->idx_entries : 1) * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->d_path_class, ctx->n_paths ? ctx->n_paths : 1))) return rc;
    if (ctx->n_paths == 0) return PNX_OK;
    const TileIdx ix = tile_idx_view(ctx);
    uint64_t *d_B = (uint64_t *)tile_idx_buf(ctx).p;
    prof_begin(ctx, PNX_K_INDEX, ctx->s_pre);
    {
        // two levels: every coarse-th boundary of a row is located inside the whole path, the
        // rest inside those brackets
        const uint32_t coarse = ctx->index_coarse ? ctx->index_coarse : 1;
        const uint32_t n_coarse_max = (ctx->max_span + coarse - 1) / coarse + 1;
        uint32_t bpp_c = (n_coarse_max + 255) / 256;
        uint32_t bpp_f = (ctx->max_span + 1 + 255) / 256;
        // path-major numbering launches paths x longest-row threads: fine when the rows are about
        // equally long, hopeless when a few paths span everything and most span little
        const uint64_t entry_blocks = (ctx->idx_entries + 255) / 256;
        const bool by_entry = ctx->index_by_entry == 1 ||
                              (ctx->index_by_entry == 0 && (uint64_t)bpp_f * ctx->n_paths > 4 * entry_blocks + 1024);
        if (by_entry) bpp_c = bpp_f = 0;
        const uint64_t grid_c = by_entry ? entry_blocks : (uint64_t)ctx->n_paths * bpp_c;
        const uint64_t grid_f = by_entry ? entry_blocks : (uint64_t)ctx->n_paths * bpp_f;
        if (grid_f > 0x7FFFFFFFull || grid_c > 0x7FFFFFFFull)
            return ctx->fail(PNX_ELIMIT, "tile index: %u paths x %u tiles exceed the grid", ctx->n_paths, ctx->max_span);
        auto go = [&](auto k_coarse, auto k_fine) {
            hipLaunchKernelGGL(k_coarse, dim3((unsigned)grid_c), dim3(256), 0, ctx->s_pre,
                               (const uint32_t *)ctx->d_items.p, (const uint64_t *)ctx->d_path_off.p, bpp_c, ctx->n_paths,
                               ctx->idx_entries, tile_items, coarse, d_B, ix,
                               (uint8_t *)ctx->d_path_class.p, (const uint8_t *)ctx->d_path_mono.p);
            if (coarse > 1)
                hipLaunchKernelGGL(k_fine, dim3((unsigned)grid_f), dim3(256), 0, ctx->s_pre,
                                   (const uint32_t *)ctx->d_items.p, (const uint64_t *)ctx->d_path_off.p, bpp_f, ctx->n_paths,
                                   ctx->idx_entries, tile_items, coarse, d_B, ix,
                                   (uint8_t *)ctx->d_path_class.p);
        };
        // ids per probe, first / later probes (PNX_CFG_INDEX_PROBE): 16 = 16/16 [default], 32 = 16/32
        if (ctx->index_probe_ids == 32) go(k_tile_index_coarse<16, 32>, k_tile_index_fine<16, 32>);
        else go(k_tile_index_coarse<16, 16>, k_tile_index_fine<16, 16>);
    }
    prof_end(ctx);
    PNX_HIP(ctx, hipGetLastError());
    return PNX_OK;
}

// ------------------------------------------------------------------------------------------
// general (non tile-monotone) paths: bookkeeping + scatter route
// ------------------------------------------------------------------------------------------
// Also lays the index of the ordered paths out in VISITING order (first tile, tiles spanned --
// 0 unless the path takes the tile route -- and row offset): the coverage kernel scans the order
// 64 entries at a time, and reading these per entry through the path id would be three dependent
// scattered gathers per window.

__global__ void k_count_general(const uint8_t *__restrict__ path_class,
                                const uint32_t *__restrict__ ord_path,
                                const uint32_t *__restrict__ ord_group, uint32_t n_ordered,
                                uint8_t *grp_general, uint32_t *flags, TileIdx ix, OrdIdx oi) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    {   // band of tiles of this wave's 64 entries (tile-route paths only)
        uint32_t lo = 0xFFFFFFFFu, hi = 0;
        if (k < n_ordered) {
            const uint32_t p0 = ord_path[k];
            if (path_class[p0] == 0 && ix.tspan[p0]) {
                lo = ix.tfirst[p0];
                hi = lo + ix.tspan[p0];
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
    if (k >= n_ordered) return;
    const uint32_t p = ord_path[k];
    const uint8_t cls = path_class[p];
    oi.tfirst[k] = ix.tfirst[p];
    oi.tspan[k] = cls == 0 ? ix.tspan[p] : 0u;
    oi.off[k] = ix.off[p];
    if (cls == 1) atomicAdd(&flags[2], 1u);        // not classified yet: this pass cannot be valid
    else if (cls == 2) atomicAdd(&flags[3], 1u);   // run route
    else if (cls == 3) {                           // scatter route: K1 merges the row of M
        grp_general[ord_group[k]] = 1;
        atomicAdd(&flags[1], 1u);
    }
}

__global__ void k_zero_if_general(uint4 *__restrict__ M, uint64_t n_vec4,
                                  const uint32_t *__restrict__ flags) {
    if (flags[1] == 0) return;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_vec4; i += stride) M[i] = make_uint4(0, 0, 0, 0);
}

__global__ void k_scatter_general(const uint32_t *__restrict__ items,
                                  const uint64_t *__restrict__ path_off,
                                  const uint32_t *__restrict__ ord_path,
                                  const uint32_t *__restrict__ ord_group, uint32_t n_ordered,
                                  const uint8_t *__restrict__ path_class, uint32_t *M,
                                  uint64_t row_words, const uint32_t *__restrict__ flags) {
    if (flags[1] == 0) return;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint32_t k = 0; k < n_ordered; ++k) {
        uint32_t p = ord_path[k];
        if (path_class[p] != 3) continue;
        uint32_t *row = M + (uint64_t)ord_group[k] * row_words;
        uint64_t e = path_off[p + 1];
        for (uint64_t j = path_off[p] + tid; j < e; j += stride) {
            uint32_t id = items[j];
            atomicOr(&row[(uint64_t)(id >> 11) * BLOCK_WORDS + (id & 63u)], 1u << ((id >> 6) & 31u));
        }
    }
}

// ------------------------------------------------------------------------------------------
// K1: tile coverage kernel (the dominant kernel of hist / histgrowth)
// ------------------------------------------------------------------------------------------
constexpr int COVER_WAVES = 4;   // waves (= tiles) per workgroup
constexpr int COVER_UNROLL = 4;  // 16-byte loads in flight per lane (u32 steps: the plain cross-check kernel)
constexpr int COVER_UNROLL12 = 2;  // the same for the packed steps: 2 x 64 lanes x 8 steps = 1024 steps per batch

// runs of the run-route paths, sorted by (tile, group); see kernels_runs.hip
struct SplitPts {  // cut points of the visiting order for the split coverage kernel
    uint32_t k[9];
};

struct RunView {
    const uint64_t *start;
    const uint32_t *len;
    const uint32_t *group;
    const uint64_t *tile_off;  // n_tiles + 1, nullptr = no run index
};

// OR the presence bits of up to four steps (positions j..j+3, valid inside [lo, hi)) into the
// wave's LDS bitmap; a step outside the tile is reported, never mis-counted
template <uint32_t TILE>
__device__ static inline void fold_steps(uint32_t *bm, const uint4 &v, uint64_t j, uint64_t lo, uint64_t hi,
                                         uint32_t tile_lo, uint32_t &viol) {
    const uint32_t ids[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint64_t idx = j + e;
        if (idx >= lo && idx < hi) {
            const uint32_t n = ids[e] - tile_lo;
            if (n < TILE)
                atomicOr(&bm[(n & 63u) + ((n >> 11) << 6)], 1u << ((n >> 6) & 31u));
            else
                viol = 1;
        }
    }
}

// stream the runs of group g that fall into this wave's tile (called right before g is folded)
// 64 run descriptors held one per lane, so that walking the runs of a tile costs one batch of
// loads per 64 runs instead of three dependent loads per run
struct RunWindow {
    uint64_t base;   // cursor of lane 0's descriptor (wave-uniform)
    uint64_t start;  // per lane
    uint32_t len, group;
};

__device__ static inline void run_window_load(const RunView &rv, RunWindow &w, uint64_t base, uint64_t end,
                                              uint32_t lane) {
    w.base = base;
    const uint64_t i = base + lane;
    const bool ok = i < end;
    w.group = ok ? rv.group[i] : 0xFFFFFFFFu;
    w.len = ok ? rv.len[i] : 0u;
    w.start = ok ? rv.start[i] : 0ull;
}

// the packed steps: a distinct pointer type, so that a kernel's first parameter tells which table it streams
struct step12_word {
    uint32_t w;
};
struct Steps8 {  // 8 steps = 96 bits
    uint32_t d0, d1, d2;
};

// 12 bytes = 8 packed steps (group j / 8 of the table; j is a multiple of 8)
template <bool NT>
__device__ static inline Steps8 load_steps12(const step12_word *tab, uint64_t j) {
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    const u32x3 *p = reinterpret_cast<const u32x3 *>(reinterpret_cast<const uint32_t *>(tab) + (j >> 3) * 3);
    u32x3 v;
    if (NT) v = __builtin_nontemporal_load(p);
    else v = *p;
    return Steps8{v.x, v.y, v.z};
}

// OR the presence bits of the 8 steps of one 12-byte load into the wave's LDS bitmap.  The load holds the
// positions jrel .. jrel + 7 of a segment whose valid steps are [rlo, rhi) (all relative to the 8-step
// aligned start of the segment's first load).  A step is 12 bits: word (id % 64), bit (id / 64 % 32) and,
// for two-block tiles, the block (id / 2048 % 2); which tile it lies in is known from the segment, not
// from the step (prepare_steps).  Steps outside [rlo, rhi) -- the neighbours of the segment's two ends in
// their groups -- OR a zero.  Steps 2 and 5 straddle two dwords (v_alignbit).
template <int WT>
__device__ static inline void fold8(uint32_t *bm, const Steps8 &v, uint32_t jrel, uint32_t rlo, uint32_t rhi) {
    const int a = (int)rlo - (int)jrel, b = (int)rhi - (int)jrel;
    const uint32_t na = a <= 0 ? 0u : (a >= 8 ? 8u : (uint32_t)a), nb = b <= 0 ? 0u : (b >= 8 ? 8u : (uint32_t)b);
    const uint32_t valid = ((1u << nb) - 1u) & ~((1u << na) - 1u);
    if (valid == 0) return;
    const uint32_t x2 = __builtin_amdgcn_alignbit(v.d1, v.d0, 24), x5 = __builtin_amdgcn_alignbit(v.d2, v.d1, 28);
    // (dword, bit offset) of the 8 steps
    const uint32_t src[8] = {v.d0, v.d0, x2, v.d1, v.d1, x5, v.d2, v.d2};
    constexpr uint32_t off[8] = {0, 12, 0, 4, 16, 0, 8, 20};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t word = __builtin_amdgcn_ubfe(src[e], off[e], 6u) +
                              (WT == 2 ? __builtin_amdgcn_ubfe(src[e], off[e] + 11u, 1u) << 6 : 0u);
        const uint32_t bit = __builtin_amdgcn_ubfe(src[e], off[e] + 6u, 5u);
        atomicOr(&bm[word], ((valid >> e) & 1u) << bit);
    }
}

// the plain cross-check kernel streams the u32 steps and checks every one against its tile
template <uint32_t TILE>
__device__ static inline void consume_runs(const RunView &rv, RunWindow &w, uint64_t &cursor, uint64_t cursor_end,
                                           uint32_t g, const uint32_t *__restrict__ items, uint32_t *bm, uint32_t lane,
                                           uint32_t tile_lo, uint32_t *flags) {
    while (cursor < cursor_end) {
        if (cursor - w.base >= 64) run_window_load(rv, w, cursor, cursor_end, lane);
        const uint32_t k = __builtin_amdgcn_readfirstlane((uint32_t)(cursor - w.base));
        if ((uint32_t)__builtin_amdgcn_readlane((int)w.group, k) != g) break;
        const uint64_t lo = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w.start >> 32), k) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w.start, k);
        const uint64_t hi = lo + (uint32_t)__builtin_amdgcn_readlane((int)w.len, k);
        uint32_t viol = 0;
        for (uint64_t base = lo & ~3ull; base < hi; base += 256ull * COVER_UNROLL) {
            uint4 v[COVER_UNROLL];
#pragma unroll
            for (int u = 0; u < COVER_UNROLL; ++u) {
                const uint64_t j = base + (uint64_t)u * 256 + lane * 4u;
                v[u] = make_uint4(0, 0, 0, 0);
                if (j < hi) v[u] = *reinterpret_cast<const uint4 *>(items + j);
            }
#pragma unroll
            for (int u = 0; u < COVER_UNROLL; ++u)
                fold_steps<TILE>(bm, v[u], base + (uint64_t)u * 256 + lane * 4u, lo, hi, tile_lo, viol);
        }
        if (__any(viol) && lane == 0) atomicAdd(&flags[4], 1u);  // cannot happen: runs are built per tile
        ++cursor;
    }
}

// the runs of group g in this wave's tile, as packed steps (a run lies inside one tile by construction)
template <int WT>
__device__ static inline void consume_runs12(const RunView &rv, RunWindow &w, uint64_t &cursor, uint64_t cursor_end,
                                             uint32_t g, const step12_word *__restrict__ steps12, uint32_t *bm, uint32_t lane) {
    constexpr int U = COVER_UNROLL12;
    while (cursor < cursor_end) {
        if (cursor - w.base >= 64) run_window_load(rv, w, cursor, cursor_end, lane);
        const uint32_t k = __builtin_amdgcn_readfirstlane((uint32_t)(cursor - w.base));
        if ((uint32_t)__builtin_amdgcn_readlane((int)w.group, k) != g) break;
        const uint64_t lo = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w.start >> 32), k) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w.start, k);
        const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)w.len, k);
        const uint64_t base = lo & ~7ull;
        const uint32_t rlo = (uint32_t)(lo - base), rhi = rlo + len;
        for (uint32_t off = 0; off < rhi; off += 512u * U) {
            Steps8 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t jrel = off + (uint32_t)u * 512u + lane * 8u;
                v[u] = Steps8{0, 0, 0};
                if (jrel < rhi) v[u] = load_steps12<false>(steps12, base + jrel);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) fold8<WT>(bm, v[u], off + (uint32_t)u * 512u + lane * 8u, rlo, rhi);
        }
        ++cursor;
    }
}

template <int NPL, int WT, bool WRITE_M>
__global__ __launch_bounds__(COVER_WAVES * 64) void k_tile_cover(
    const uint32_t *__restrict__ items, TileIdx ix, OrdIdx,
    const uint32_t *__restrict__ ord_path, const uint32_t *__restrict__ ord_group,
    uint32_t n_ordered, uint8_t *path_class, const uint8_t *__restrict__ grp_general,
    const uint8_t *__restrict__ exclude, uint32_t n_items, uint32_t n_tiles, uint32_t n_blocks,
    uint32_t *M, uint64_t row_words, uint32_t *__restrict__ countable, uint32_t *flags, RunView rv, SplitPts) {
    constexpr uint32_t TILE = WT * BLOCK_ITEMS;
    __shared__ uint32_t bm_all[COVER_WAVES][WT * BLOCK_WORDS];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile = blockIdx.x * COVER_WAVES + wave;
    if (tile >= n_tiles) return;  // whole wave leaves; no workgroup barrier is ever used
    uint32_t *bm = bm_all[wave];
    const uint32_t tile_lo = tile * TILE;
    uint64_t run_c = rv.tile_off ? rv.tile_off[tile] : 0;
    const uint64_t run_end = rv.tile_off ? rv.tile_off[tile + 1] : 0;
    RunWindow run_w;
    run_w.base = run_c;
    run_w.start = 0;
    run_w.len = 0;
    run_w.group = 0xFFFFFFFFu;
    if (run_c < run_end) run_window_load(rv, run_w, run_c, run_end, lane);

#pragma unroll
    for (int w = 0; w < WT; ++w) bm[w * BLOCK_WORDS + lane] = 0;

    // exclusion words in presence layout (ActiveTable, src/util.rs:118-124)
    uint32_t excl[WT];
#pragma unroll
    for (int w = 0; w < WT; ++w) {
        excl[w] = 0;
        if (exclude) {
            for (uint32_t b = 0; b < 32; ++b) {
                uint64_t node = (uint64_t)tile_lo + (uint32_t)w * BLOCK_ITEMS + b * 64u + lane;
                if (node <= n_items && exclude[node]) excl[w] |= 1u << b;
            }
        }
    }

    uint32_t cnt[NPL][WT];
#pragma unroll
    for (int k = 0; k < NPL; ++k)
#pragma unroll
        for (int w = 0; w < WT; ++w) cnt[k][w] = 0;

    // fold the LDS bitmap of the finished group into the bit-sliced counters
    auto flush = [&](uint32_t g) {
        if (run_c < run_end) consume_runs<TILE>(rv, run_w, run_c, run_end, g, items, bm, lane, tile_lo, flags);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool merge = grp_general != nullptr && grp_general[g] != 0;
#pragma unroll
        for (int w = 0; w < WT; ++w) {
            uint32_t x = bm[w * BLOCK_WORDS + lane];
            bm[w * BLOCK_WORDS + lane] = 0;
            const uint32_t blk = tile * WT + w;
            if (blk < n_blocks) {
                uint32_t *mw = M + (uint64_t)g * row_words + (uint64_t)blk * BLOCK_WORDS + lane;
                if (merge) x |= *mw;
                x &= ~excl[w];
                if (WRITE_M) *mw = x;
            }
            uint32_t carry = x;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                uint32_t t = cnt[k][w] & carry;
                cnt[k][w] ^= carry;
                carry = t;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };

    uint32_t cur_g = n_ordered ? ord_group[0] : 0;
    for (uint32_t k = 0; k < n_ordered; ++k) {
        const uint32_t p = ord_path[k];
        const uint32_t g = ord_group[k];
        if (g != cur_g) {
            flush(cur_g);
            cur_g = g;
        }
        if (path_class[p]) continue;  // scatter route
        const uint32_t jt = tile - ix.tfirst[p];  // wraps for tiles before the path's first one
        if (jt >= ix.tspan[p]) continue;          // the path does not reach this tile
        const uint64_t ba = ix.B[ix.off[p] + jt], bb = ix.B[ix.off[p] + jt + 1];
        const uint64_t lo = ba < bb ? ba : bb, hi = ba < bb ? bb : ba;
        uint32_t viol = 0;
        for (uint64_t base = lo & ~3ull; base < hi; base += 256ull * COVER_UNROLL) {
            uint4 v[COVER_UNROLL];
#pragma unroll
            for (int u = 0; u < COVER_UNROLL; ++u) {
                const uint64_t j = base + (uint64_t)u * 256 + lane * 4u;
                v[u] = make_uint4(0, 0, 0, 0);
                if (j < hi) v[u] = *reinterpret_cast<const uint4 *>(items + j);
            }
#pragma unroll
            for (int u = 0; u < COVER_UNROLL; ++u) {
                const uint64_t j = base + (uint64_t)u * 256 + lane * 4u;
                const uint32_t ids[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint64_t idx = j + e;
                    if (idx >= lo && idx < hi) {
                        const uint32_t n = ids[e] - tile_lo;
                        if (n < TILE)
                            atomicOr(&bm[(n & 63u) + ((n >> 11) << 6)], 1u << ((n >> 6) & 31u));
                        else
                            viol = 1;
                    }
                }
            }
        }
        if (__any(viol)) {
            // the path is not tile-monotone after all: send it down the scatter route and
            // invalidate this pass (the host re-runs it)
            if (lane == 0) {
                path_class[p] = 1;
                atomicAdd(&flags[0], 1u);
            }
        }
    }
    if (n_ordered) flush(cur_g);

    // unpack the bit-sliced counters: one coalesced 256 B store per bit position
#pragma unroll
    for (int w = 0; w < WT; ++w) {
        const uint32_t blk = tile * WT + w;
        if (blk >= n_blocks) break;
        for (uint32_t b = 0; b < 32; ++b) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < NPL; ++k) v |= ((cnt[k][w] >> b) & 1u) << k;
            const uint64_t node = (uint64_t)blk * BLOCK_ITEMS + b * 64u + lane;
            // countable[0] is the reference's reserved element (abacus.rs:549-551)
            if (node <= n_items) countable[node] = node ? v : 0xFFFFFFFFu;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K1, software-pipelined form: the 16-byte loads of segment k+1 (and its boundary pair) are
// issued before the steps of segment k are folded into the LDS bitmap, so every wave keeps
// two segments in flight and the HBM latency of one segment hides behind the VALU/LDS work
// of the previous one.  NT selects non-temporal loads for the CSR stream (read exactly once).
// ------------------------------------------------------------------------------------------
// RUNS = false compiles the run consumption out: the kernel of graphs whose paths are all
// tile-monotone keeps its low register count (occupancy 5+ waves/SIMD, all tiles resident).
// SPLIT > 1 gives every tile SPLIT waves of one workgroup; each takes a contiguous, group-aligned
// part of the visiting order (SplitPts, cut by the host) with private counters, and the first one
// adds the others' bit-sliced counters (ripple-carry through LDS) before the write-out.  The
// parallelism of the kernel is then tiles x SPLIT instead of tiles: 10 M items are only 4883
// tiles for 6144 wave slots, and one wave per tile is a long serial chain of segments.
// SKIP (plain histogram passes only: no presence matrix, no runs, no scatter merge): a group that has
// no segment in the tile needs no visit at all, so only non-empty segments are entries, and the
// windows (aligned to 64 entries of the order) whose band of tiles misses the tile are never loaded
// -- 64 bands are tested per step.  With the paths of a group sorted by their first tile
// (normalize_order) a wave of a 100 000-contig graph loads a handful of windows per group.
template <int NPL, int WT, bool WRITE_M, bool NT, int CW, bool RUNS, int SPLIT = 1, bool SKIP = false>
__global__ __launch_bounds__(CW * 64, (!RUNS && !SKIP && NPL <= 12 && SPLIT > 1) ? 6 : 1) void k_tile_cover_pipe(
    const step12_word *__restrict__ steps12, TileIdx ix, OrdIdx oi,
    const uint32_t *__restrict__ ord_path, const uint32_t *__restrict__ ord_group,
    uint32_t n_ordered, uint8_t *path_class, const uint8_t *__restrict__ grp_general,
    const uint8_t *__restrict__ exclude, uint32_t n_items, uint32_t n_tiles, uint32_t n_blocks,
    uint32_t *M, uint64_t row_words, uint32_t *__restrict__ countable, uint32_t *flags, RunView rv,
    SplitPts sp) {
    constexpr uint32_t TILE = WT * BLOCK_ITEMS;
    constexpr int U = COVER_UNROLL12;
    constexpr int TPW = CW / SPLIT;  // tiles per workgroup
    static_assert(CW % SPLIT == 0, "waves per workgroup must be a multiple of the split");
    __shared__ uint32_t bm_all[CW][WT * BLOCK_WORDS];
    __shared__ uint32_t xch[SPLIT > 1 ? TPW * (SPLIT - 1) * NPL * WT * 64 : 1];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t part = SPLIT > 1 ? wave % SPLIT : 0;
    const uint32_t tile_raw = blockIdx.x * TPW + wave / SPLIT;
    if (SPLIT == 1 && tile_raw >= n_tiles) return;
    const bool active = tile_raw < n_tiles;  // SPLIT > 1: idle waves still meet the barrier
    const uint32_t tile = active ? tile_raw : n_tiles - 1;
    const uint32_t k_lo = SPLIT > 1 ? (active ? sp.k[part] : 0u) : 0u;
    const uint32_t k_hi = SPLIT > 1 ? (active ? sp.k[part + 1] : 0u) : n_ordered;
    uint32_t *bm = bm_all[wave];
    const uint32_t tile_lo = tile * TILE;
    uint64_t run_c = RUNS && rv.tile_off ? rv.tile_off[tile] : 0;
    const uint64_t run_end = RUNS && rv.tile_off ? rv.tile_off[tile + 1] : 0;
    if (RUNS && SPLIT > 1 && part > 0 && k_lo < k_hi && run_c < run_end) {
        // the tile's runs are sorted by group: start at the first run of this part's groups
        const uint32_t g_first = ord_group[k_lo];
        uint64_t a = run_c, b = run_end;
        while (a < b) {
            const uint64_t mid = (a + b) >> 1;
            if (rv.group[mid] < g_first) a = mid + 1; else b = mid;
        }
        run_c = a;
    }
    RunWindow run_w;
    run_w.base = run_c;
    run_w.start = 0;
    run_w.len = 0;
    run_w.group = 0xFFFFFFFFu;
    if (RUNS && run_c < run_end) run_window_load(rv, run_w, run_c, run_end, lane);

#pragma unroll
    for (int w = 0; w < WT; ++w) bm[w * BLOCK_WORDS + lane] = 0;

    uint32_t excl[WT];
#pragma unroll
    for (int w = 0; w < WT; ++w) {
        excl[w] = 0;
        if (exclude) {
            for (uint32_t b = 0; b < 32; ++b) {
                uint64_t node = (uint64_t)tile_lo + (uint32_t)w * BLOCK_ITEMS + b * 64u + lane;
                if (node <= n_items && exclude[node]) excl[w] |= 1u << b;
            }
        }
    }

    uint32_t cnt[NPL][WT];
#pragma unroll
    for (int k = 0; k < NPL; ++k)
#pragma unroll
        for (int w = 0; w < WT; ++w) cnt[k][w] = 0;

    auto flush = [&](uint32_t g) {
        if (RUNS && run_c < run_end) consume_runs12<WT>(rv, run_w, run_c, run_end, g, steps12, bm, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool merge = grp_general != nullptr && grp_general[g] != 0;
#pragma unroll
        for (int w = 0; w < WT; ++w) {
            uint32_t x = bm[w * BLOCK_WORDS + lane];
            bm[w * BLOCK_WORDS + lane] = 0;
            const uint32_t blk = tile * WT + w;
            if (blk < n_blocks) {
                uint32_t *mw = M + (uint64_t)g * row_words + (uint64_t)blk * BLOCK_WORDS + lane;
                if (merge) x |= *mw;
                x &= ~excl[w];
                if (WRITE_M) *mw = x;
            }
            uint32_t carry = x;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                uint32_t t = cnt[k][w] & carry;
                cnt[k][w] ^= carry;
                carry = t;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };

    // ---- window of 64 order entries, one per lane -------------------------------------------
    // Entry k = win_base + lane: its segment [w_lo, w_lo + w_len) of this tile and its group.
    // The dependent loads (order -> path -> boundary pair) are paid once per 64 entries instead of
    // once per entry, and only the INTERESTING entries -- a non-empty segment or the first entry
    // of a group -- are visited at all: in an assembly-shaped pangenome (thousands of contig
    // paths, each touching a few per cent of the tiles) almost every (path, tile) pair is empty.
    uint64_t w_lo = 0;
    uint32_t w_len = 0, w_g = 0xFFFFFFFFu;
    uint32_t win_base = k_lo;
    uint64_t todo = 0;  // interesting entries of the window that are still to be visited
    auto load_window = [&](uint32_t base, uint32_t prev_group) {
        win_base = base;
        const uint32_t k = base + lane;
        const bool in = k < k_hi && (!SKIP || k >= k_lo);
        w_g = in ? ord_group[k] : 0xFFFFFFFFu;
        uint64_t ba = 0, bb = 0;
        if (in) {
            const uint32_t jt = tile - oi.tfirst[k];  // wraps for tiles before the path's first one
            if (jt < oi.tspan[k]) {                   // 0 for paths that are not on the tile route
                const uint64_t *row = ix.B + oi.off[k] + jt;
                ba = row[0];
                bb = row[1];
            }
        }
        w_lo = ba < bb ? ba : bb;
        const uint64_t len = (ba < bb ? bb : ba) - w_lo;
        w_len = (uint32_t)len;
        if (len > 0x7FFFFFF0ull) {  // not a segment this kernel can stream (32-bit positions): hand the path to the general routes
            w_len = 0;
            path_class[ord_path[k]] = 1;
            atomicAdd(&flags[0], 1u);
        }
        if (SKIP) {
            todo = __ballot(in && w_len != 0);
        } else {
            uint32_t pg = __shfl_up(w_g, 1);
            if (lane == 0) pg = prev_group;
            todo = __ballot(in && (w_len != 0 || w_g != pg));
        }
    };
    // SKIP: first window at or after w_from whose band of tiles contains this tile
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
    // scalars of one entry out of the window
    auto entry = [&](uint32_t k, uint64_t &lo, uint64_t &hi, uint32_t &g) {
        const uint32_t i = __builtin_amdgcn_readfirstlane(k - win_base);
        lo = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w_lo >> 32), i) << 32) |
             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w_lo, i);
        hi = lo + (uint32_t)__builtin_amdgcn_readlane((int)w_len, i);
        g = (uint32_t)__builtin_amdgcn_readlane((int)w_g, i);
    };
    // next interesting entry at or after the window position; false when the part is exhausted
    auto next_entry = [&](uint32_t &k) {
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
        const uint32_t i = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        k = win_base + i;
        return true;
    };

    // prefetch state of the NEXT interesting entry
    Steps8 nxt[U];
    auto issue = [&](uint64_t lo, uint64_t hi) {
        const uint64_t base = lo & ~7ull;  // 8 steps = 12 bytes
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t j = base + (uint64_t)u * 512 + lane * 8u;
            nxt[u] = Steps8{0, 0, 0};
            if (j < hi) nxt[u] = load_steps12<NT>(steps12, j);
        }
    };

    // One extra (sentinel) iteration closes the last group, so the flush code -- and the run
    // consumption inlined in it -- exists once in the kernel (register pressure).
    const bool some = k_lo < k_hi;
    uint32_t k = k_lo, cur_g = 0;
    uint64_t e_lo = 0, e_hi = 0;
    uint32_t e_g = 0xFFFFFFFFu;
    bool have = false;
    if (some) {
        if (SKIP) {
            uint32_t w;
            if (seek_window(k_lo >> 6, w)) {
                load_window(w << 6, 0u);
            } else {
                win_base = ((k_hi + 63) >> 6) << 6;  // nothing reaches this tile
                todo = 0;
            }
        } else {
            load_window(k_lo, 0xFFFFFFFFu);  // the first entry of the part always counts as a group change
        }
        have = next_entry(k);
        if (have) {
            entry(k, e_lo, e_hi, e_g);
            cur_g = e_g;
            issue(e_lo, e_hi);
        }
    }
    for (;;) {
        const bool last = !have;
        const uint32_t g = last ? 0xFFFFFFFFu : e_g;
        // With runs the group change comes first: while the previous group is folded and its
        // runs are streamed only the prefetched segment is live, not a second copy of it.
        // Without runs the next segment is issued first, so its loads also cover the fold.
        if (RUNS && g != cur_g && some) {
            flush(cur_g);
            cur_g = g;
        }
        Steps8 cur[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        const uint64_t lo = last ? 0 : e_lo, hi = last ? 0 : e_hi;
        if (!last) {
            have = next_entry(k);
            if (have) {
                entry(k, e_lo, e_hi, e_g);
                issue(e_lo, e_hi);
            }
        }
        if (!RUNS && g != cur_g && some) {
            flush(cur_g);
            cur_g = g;
        }
        if (hi > lo) {
            const uint64_t base = lo & ~7ull;
            const uint32_t rlo = (uint32_t)(lo - base), rhi = (uint32_t)(hi - base);
#pragma unroll
            for (int u = 0; u < U; ++u) fold8<WT>(bm, cur[u], (uint32_t)u * 512u + lane * 8u, rlo, rhi);
            // long segments (> U*512 steps): the tail is streamed directly
            for (uint32_t off = 512u * U; off < rhi; off += 512u * U) {
                Steps8 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t jrel = off + (uint32_t)u * 512u + lane * 8u;
                    v[u] = Steps8{0, 0, 0};
                    if (jrel < rhi) v[u] = load_steps12<NT>(steps12, base + jrel);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) fold8<WT>(bm, v[u], off + (uint32_t)u * 512u + lane * 8u, rlo, rhi);
            }
        }
        if (last) break;
    }

    if (SPLIT > 1) {
        // parts 1.. hand their counters to part 0 of the tile
        uint32_t *xt = xch + (size_t)(wave / SPLIT) * (SPLIT - 1) * NPL * WT * 64;
        if (part > 0) {
#pragma unroll
            for (int k = 0; k < NPL; ++k)
#pragma unroll
                for (int w = 0; w < WT; ++w) xt[((part - 1) * NPL * WT + k * WT + w) * 64 + lane] = cnt[k][w];
        }
        __syncthreads();
        if (part > 0 || !active) return;
        for (int q = 0; q < SPLIT - 1; ++q) {
#pragma unroll
            for (int w = 0; w < WT; ++w) {
                uint32_t carry = 0;
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    const uint32_t a = cnt[k][w], b = xt[(q * NPL * WT + k * WT + w) * 64 + lane];
                    cnt[k][w] = a ^ b ^ carry;
                    carry = (a & b) | (carry & (a ^ b));
                }
            }
        }
    }
#pragma unroll
    for (int w = 0; w < WT; ++w) {
        const uint32_t blk = tile * WT + w;
        if (blk >= n_blocks) break;
        for (uint32_t b = 0; b < 32; ++b) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < NPL; ++k) v |= ((cnt[k][w] >> b) & 1u) << k;
            const uint64_t node = (uint64_t)blk * BLOCK_ITEMS + b * 64u + lane;
            if (node <= n_items) countable[node] = node ? v : 0xFFFFFFFFu;
        }
    }
}

// ------------------------------------------------------------------------------------------
// one full pass for the current order: general bookkeeping -> scatter -> K1 -> K2
// ------------------------------------------------------------------------------------------
template <typename F>
struct first_param;
template <typename A0, typename... As>
struct first_param<void (*)(A0, As...)> {
    using type = A0;
};

template <int NPL, int WT>
static void launch_cover_t(pnx_ctx *ctx, bool write_m, bool use_m) {
    const uint64_t row_words = (uint64_t)ctx->n_blocks * BLOCK_WORDS;
    RunView rv{nullptr, nullptr, nullptr, nullptr};
    if (ctx->n_runs && ctx->runs_sorted)
        rv = RunView{(const uint64_t *)ctx->d_srun_start.p, (const uint32_t *)ctx->d_srun_len.p,
                     (const uint32_t *)ctx->d_srun_group.p, (const uint64_t *)ctx->d_run_tile_off.p};
    const bool has_runs = rv.tile_off != nullptr;
    SplitPts sp{};
    auto launch = [&](auto kern, int cw, int split = 1) {
        // the pipelined kernels stream the packed 12-bit steps, the plain cross-check kernel the u32 ItemTable
        using items_ptr_t = typename first_param<decltype(kern)>::type;
        const void *items_ptr = std::is_same<items_ptr_t, const step12_word *>::value ? ctx->d_steps12.p : ctx->d_items.p;
        const unsigned tpw = (unsigned)(cw / split);
        const unsigned grid = (ctx->n_tiles + tpw - 1) / tpw;
        // group-aligned cut points of the visiting order
        for (int j = 0; j <= split; ++j) {
            uint64_t t = (uint64_t)ctx->n_ordered * j / split;
            while (t > 0 && t < ctx->n_ordered && ctx->h_ord_group[t] == ctx->h_ord_group[t - 1]) ++t;
            sp.k[j] = (uint32_t)t;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(cw * 64), 0, ctx->s_main,
                           static_cast<items_ptr_t>(items_ptr), tile_idx_view(ctx), ord_idx_view(ctx),
                           (const uint32_t *)ctx->d_ord_path.p, (const uint32_t *)ctx->d_ord_group.p,
                           ctx->n_ordered, (uint8_t *)ctx->d_path_class.p,
                           use_m ? (const uint8_t *)ctx->cur->d_grp_general : (const uint8_t *)nullptr,
                           ctx->have_exclude ? (const uint8_t *)ctx->d_exclude.p : (const uint8_t *)nullptr,
                           ctx->n_items, ctx->n_tiles, ctx->n_blocks, (uint32_t *)ctx->d_M.p, row_words,
                           (uint32_t *)ctx->cur->d_countable.p, ctx->cur->d_flags, rv, sp);
    };
    // window skipping pays when the order is long (thousands of paths); with a few hundred dense
    // paths every window is needed and the plain form keeps its lower register count
    const bool skip = !write_m && !use_m && !has_runs &&
                      (ctx->cover_skip == 1 || (ctx->cover_skip == 0 && ctx->n_ordered >= 4096));
    // waves per tile: enough waves to fill the chip about six times over
    int split = 1;
    if (WT == 1 && ctx->cover_variant == 2) {
        split = ctx->cover_split;
        if (split == 0) {
            const uint64_t want = 6ull * (uint64_t)ctx->prop.multiProcessorCount * (has_runs ? 20 : 24);
            split = 1;
            while (split < 8 && (uint64_t)ctx->n_tiles * split < want && (uint32_t)split * 2 <= ctx->n_groups) split *= 2;
        }
    }
    switch (ctx->cover_variant) {
        case 1:
            if (write_m) launch(k_tile_cover_pipe<NPL, WT, true, false, COVER_WAVES, true>, COVER_WAVES);
            else launch(k_tile_cover_pipe<NPL, WT, false, false, COVER_WAVES, true>, COVER_WAVES);
            break;
        case 2:
            if constexpr (WT == 1) {
                if (split > 1) {
                    auto go = [&](auto k2, auto k4, auto k8) {
                        if (split == 2) launch(k2, 4, 2);
                        else if (split == 4) launch(k4, 4, 4);
                        else launch(k8, 8, 8);
                    };
                    if (has_runs) {
                        if (write_m) go(k_tile_cover_pipe<NPL, 1, true, true, 4, true, 2>, k_tile_cover_pipe<NPL, 1, true, true, 4, true, 4>, k_tile_cover_pipe<NPL, 1, true, true, 8, true, 8>);
                        else go(k_tile_cover_pipe<NPL, 1, false, true, 4, true, 2>, k_tile_cover_pipe<NPL, 1, false, true, 4, true, 4>, k_tile_cover_pipe<NPL, 1, false, true, 8, true, 8>);
                    } else {
                        if (write_m) go(k_tile_cover_pipe<NPL, 1, true, true, 4, false, 2>, k_tile_cover_pipe<NPL, 1, true, true, 4, false, 4>, k_tile_cover_pipe<NPL, 1, true, true, 8, false, 8>);
                        else if (skip) go(k_tile_cover_pipe<NPL, 1, false, true, 4, false, 2, true>, k_tile_cover_pipe<NPL, 1, false, true, 4, false, 4, true>, k_tile_cover_pipe<NPL, 1, false, true, 8, false, 8, true>);
                        else go(k_tile_cover_pipe<NPL, 1, false, true, 4, false, 2>, k_tile_cover_pipe<NPL, 1, false, true, 4, false, 4>, k_tile_cover_pipe<NPL, 1, false, true, 8, false, 8>);
                    }
                    break;
                }
            }
            if (has_runs) {
                if (write_m) launch(k_tile_cover_pipe<NPL, WT, true, true, 4, true>, 4);
                else launch(k_tile_cover_pipe<NPL, WT, false, true, 4, true>, 4);
                break;
            }
            switch (ctx->cover_waves) {
                case 1: if (write_m) launch(k_tile_cover_pipe<NPL, WT, true, true, 1, false>, 1); else launch(k_tile_cover_pipe<NPL, WT, false, true, 1, false>, 1); break;
                case 2: if (write_m) launch(k_tile_cover_pipe<NPL, WT, true, true, 2, false>, 2); else launch(k_tile_cover_pipe<NPL, WT, false, true, 2, false>, 2); break;
                case 8: if (write_m) launch(k_tile_cover_pipe<NPL, WT, true, true, 8, false>, 8); else launch(k_tile_cover_pipe<NPL, WT, false, true, 8, false>, 8); break;
                default:
                    if (write_m) launch(k_tile_cover_pipe<NPL, WT, true, true, 4, false>, 4);
                    else if (skip) launch(k_tile_cover_pipe<NPL, WT, false, true, 4, false, 1, true>, 4);
                    else launch(k_tile_cover_pipe<NPL, WT, false, true, 4, false>, 4);
                    break;
            }
            break;
        default:
            if (write_m) launch(k_tile_cover<NPL, WT, true>, COVER_WAVES); else launch(k_tile_cover<NPL, WT, false>, COVER_WAVES);
    }
}

template <int WT>
static int launch_cover_wt(pnx_ctx *ctx, bool write_m, bool use_m) {
    // planes needed to count up to n_groups inclusive
    uint32_t bits = 1;
    while (bits < 32 && (ctx->n_groups >> bits) != 0) ++bits;
    if (bits <= 8) launch_cover_t<8, WT>(ctx, write_m, use_m);
    else if (bits <= 12) launch_cover_t<12, WT>(ctx, write_m, use_m);
    else if (bits <= 16) launch_cover_t<16, WT>(ctx, write_m, use_m);
    else if (bits <= 24) launch_cover_t<24, WT>(ctx, write_m, use_m);
    else return ctx->fail(PNX_ELIMIT, "more than 2^24-1 groups are not supported (got %u)", ctx->n_groups);
    return PNX_OK;
}

// phases 1 + 2 of a pass over the steps themselves (the histogram phase is shared: launch_cover_pass, pass_pipeline.hip)
static int launch_step_phases(pnx_ctx *ctx, bool use_m, uint64_t m_words) {
    Ticket *tk = ctx->cur;
    int rc;
    if ((rc = ensure_path_spans(ctx)) || (rc = normalize_order(ctx))) return rc;
    const bool phased = ctx->s_pre != ctx->s_main;
    if (ctx->n_ordered) {
        prof_begin(ctx, PNX_K_SCATTER, ctx->s_pre);
        hipLaunchKernelGGL(k_count_general, dim3((ctx->n_ordered + 255) / 256), dim3(256), 0, ctx->s_pre,
                           (const uint8_t *)ctx->d_path_class.p, (const uint32_t *)ctx->d_ord_path.p,
                           (const uint32_t *)ctx->d_ord_group.p, ctx->n_ordered,
                           ctx->cur->d_grp_general, ctx->cur->d_flags, tile_idx_view(ctx), ord_idx_view(ctx));
        if (use_m && m_words) {  // never phased: s_pre == s_main
            hipLaunchKernelGGL(k_zero_if_general, dim3(2048), dim3(256), 0, ctx->s_pre,
                               (uint4 *)ctx->d_M.p, m_words / 4, (const uint32_t *)ctx->cur->d_flags);
            hipLaunchKernelGGL(k_scatter_general, dim3(2048), dim3(256), 0, ctx->s_pre,
                               (const uint32_t *)ctx->d_items.p, (const uint64_t *)ctx->d_path_off.p,
                               (const uint32_t *)ctx->d_ord_path.p, (const uint32_t *)ctx->d_ord_group.p,
                               ctx->n_ordered, (const uint8_t *)ctx->d_path_class.p, (uint32_t *)ctx->d_M.p,
                               (uint64_t)ctx->n_blocks * BLOCK_WORDS, (const uint32_t *)ctx->cur->d_flags);
        }
        prof_end(ctx);
        PNX_HIP(ctx, hipGetLastError());
    }
    if (phased) {
        PNX_HIP(ctx, hipEventRecord(tk->ev_pre, ctx->s_pre));
        PNX_HIP(ctx, hipStreamWaitEvent(ctx->s_main, tk->ev_pre, 0));
    }

    // ---- phase 2 (s_main): the coverage kernel
    prof_begin(ctx, PNX_K_COVER, ctx->s_main);
    if (ctx->tile_blocks == 2) rc = launch_cover_wt<2>(ctx, ctx->want_M, use_m);
    else rc = launch_cover_wt<1>(ctx, ctx->want_M, use_m);
    prof_end(ctx);
    if (rc) return rc;
    PNX_HIP(ctx, hipGetLastError());
    return PNX_OK;
}

static int sort_runs_if_needed(pnx_ctx *ctx) {
    if (ctx->n_runs && !ctx->runs_sorted) return sort_run_index(ctx);
    return PNX_OK;
}

}  // namespace pnx

// the table the product library asks for when a step route is configured (pass_pipeline.hip: step_routes)
extern "C" const pnx::StepRoutes *pnx_step_routes_table() {
    static const pnx::StepRoutes t{pnx::prepare_steps, pnx::restore_step_order, pnx::launch_tile_index, pnx::build_run_index,
                                   pnx::sort_runs_if_needed, pnx::launch_step_phases};
    return &t;
}
