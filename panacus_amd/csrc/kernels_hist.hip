// kernels_hist.hip -- K2: histogram of the coverage vector on gfx950 (MI355X).
//
// Replaces AbacusByTotal::construct_hist / construct_hist_bps (src/graph_broker/abacus.rs:746-787).  A file (and
// therefore a code object) of its own: together with kernels_rows.hip it is all a default `hist` run loads.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "pnx_context.hpp"

namespace pnx {

// ------------------------------------------------------------------------------------------
// K2: histogram of the coverage vector (construct_hist / construct_hist_bps)
// ------------------------------------------------------------------------------------------
constexpr uint32_t HIST_LDS_BINS = 4096;

template <bool WEIGHTED, bool USE_LDS>
__global__ __launch_bounds__(1024) void k_hist(const uint32_t *__restrict__ countable,
                                              const uint32_t *__restrict__ weights,
                                              uint32_t n_items, uint32_t n_groups,
                                              unsigned long long *hist) {
    __shared__ unsigned long long sh[USE_LDS ? HIST_LDS_BINS : 1];
    if (USE_LDS) {
        for (uint32_t i = threadIdx.x; i <= n_groups; i += blockDim.x) sh[i] = 0;
        __syncthreads();
    }
    // The bins 0, 1 and n_groups (uncovered, private and core items) hold most items of a
    // pangenome; adding them through LDS atomics would serialise up to 64 lanes on one address,
    // so each lane keeps them in registers and the wave folds them once at the end.
    unsigned long long hot0 = 0, hot1 = 0, hotg = 0;
    auto add = [&](uint32_t c, unsigned long long w) {
        if (c > n_groups) return;  // abacus.rs:752 / :771: coverage beyond #groups is ignored
        if (c == 0) hot0 += w;
        else if (c == 1) hot1 += w;
        else if (c == n_groups) hotg += w;
        else if (USE_LDS) atomicAdd(&sh[c], w);
        else atomicAdd(&hist[c], w);
    };
    // items 1..n_items, four per lane and step (16-byte loads); quad q covers items 4q..4q+3
    const uint64_t n_quads = ((uint64_t)n_items + 4) / 4;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += stride) {
        const uint64_t i0 = 4 * q;
        if (i0 + 3 <= n_items) {
            const uint4 c4 = *reinterpret_cast<const uint4 *>(countable + i0);
            uint4 w4 = make_uint4(1, 1, 1, 1);
            if (WEIGHTED) w4 = *reinterpret_cast<const uint4 *>(weights + i0);
            if (i0 != 0) add(c4.x, w4.x);  // item 0 is the sentinel
            add(c4.y, w4.y);
            add(c4.z, w4.z);
            add(c4.w, w4.w);
        } else {
            for (uint64_t i = i0 ? i0 : 1; i <= n_items; ++i) add(countable[i], WEIGHTED ? weights[i] : 1u);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        hot0 += __shfl_down(hot0, o);
        hot1 += __shfl_down(hot1, o);
        hotg += __shfl_down(hotg, o);
    }
    if ((threadIdx.x & 63) == 0) {
        // n_groups == 1 (or 0): the hot bins coincide, every item went to exactly one of them
        if (USE_LDS) {
            if (hot0) atomicAdd(&sh[0], hot0);
            if (hot1) atomicAdd(&sh[1], hot1);
            if (hotg) atomicAdd(&sh[n_groups], hotg);
        } else {
            if (hot0) atomicAdd(&hist[0], hot0);
            if (hot1) atomicAdd(&hist[1], hot1);
            if (hotg) atomicAdd(&hist[n_groups], hotg);
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (uint32_t b = threadIdx.x; b <= n_groups; b += blockDim.x)
            if (sh[b]) atomicAdd(&hist[b], sh[b]);
    }
}

// hist[b] = sum of the replicas the coverage kernel added to (kernels_rows.hip); host_block: the pass's result block
// [flags u32[8] | hist] in the ticket's pinned memory, written here so that no copy has to follow
__global__ __launch_bounds__(1024) void k_hist_publish(const unsigned long long *__restrict__ rep, uint32_t bins,
                                                       unsigned long long *__restrict__ hist, const uint32_t *__restrict__ flags,
                                                       uint32_t *__restrict__ host_block) {
    for (uint32_t b = threadIdx.x; b < bins; b += blockDim.x) {
        unsigned long long v[HIST_REPLICAS], s = 0;  // all loads in flight before the first addition
#pragma unroll
        for (uint32_t r = 0; r < HIST_REPLICAS; ++r) v[r] = rep[(size_t)r * bins + b];
#pragma unroll
        for (uint32_t r = 0; r < HIST_REPLICAS; ++r) s += v[r];
        hist[b] = s;
        if (host_block) {
            host_block[8 + 2 * b] = (uint32_t)s;
            host_block[8 + 2 * b + 1] = (uint32_t)(s >> 32);
        }
    }
    if (host_block && threadIdx.x < 8) host_block[threadIdx.x] = flags[threadIdx.x];
}

int launch_hist(pnx_ctx *ctx, Ticket *tk) {
    if (tk->hist_fused) {
        // the coverage kernel left HIST_REPLICAS partial histograms: add them up, and hand [flags | hist] to the host
        // (multi-GPU: the all-reduce of the block follows this kernel, the copy to the host follows that)
        const bool to_host = tk->h_block_mapped && !(ctx->comm && ctx->comm_reduce_hist);
        tk->host_written = to_host;
        prof_begin(ctx, PNX_K_HIST, ctx->s_post);
        hipLaunchKernelGGL(k_hist_publish, dim3(1), dim3(1024), 0, ctx->s_post, (const unsigned long long *)tk->d_hist_rep, ctx->n_groups + 1,
                           (unsigned long long *)tk->d_hist, (const uint32_t *)tk->d_flags, to_host ? (uint32_t *)tk->h_block_mapped : (uint32_t *)nullptr);
        prof_end(ctx);
        PNX_HIP(ctx, hipGetLastError());
        return PNX_OK;
    }
    tk->host_written = false;
    prof_begin(ctx, PNX_K_HIST, ctx->s_post);
    {
        // one workgroup of 16 waves per CU at most: every workgroup ends with one global atomic
        // per non-empty bin, and those serialise per address
        uint64_t want = ((uint64_t)ctx->n_items / 4 + 1024) / 1024;
        unsigned grid = (unsigned)(want < 1 ? 1 : (want > 256 ? 256 : want));
        const bool lds = ctx->n_groups + 1 <= HIST_LDS_BINS;
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, ctx->s_post,
                               (const uint32_t *)tk->d_countable.p, (const uint32_t *)ctx->d_weights.p,
                               ctx->n_items, ctx->n_groups, (unsigned long long *)tk->d_hist);
        };
        if (ctx->weighted) { if (lds) go(k_hist<true, true>); else go(k_hist<true, false>); }
        else { if (lds) go(k_hist<false, true>); else go(k_hist<false, false>); }
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
void preload_hist(unsigned what) {
    hipFuncAttributes a;
    auto touch = [&a](const void *k) { (void)hipFuncGetAttributes(&a, k); };
    if (what & PNX_PRELOAD_PASS) touch((const void *)k_hist_publish);
}
}  // namespace pnx
