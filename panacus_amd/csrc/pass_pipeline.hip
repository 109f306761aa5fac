// pass_pipeline.hip -- what every coverage pass shares, whatever reads the steps: buffers of the pass's ticket, the three
// phases and their streams, the histogram phase behind them; the upload's id check; the chunk prefix of the graph.
//
//   phases 1 + 2   over path rows (kernels_rows.hip), in one read of the steps (kernels_band.hip), or -- cross-check only --
//                  by the step routes of round 2, which live in a module of their own (libpanacus_hip_steps.so,
//                  kernels_cover.hip + kernels_runs.hip) that is opened when PNX_CFG_COVER_VARIANT asks for one
//   phase 3        launch_hist (kernels_hist.hip)
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>

#include <hip/hip_runtime.h>

#include "pnx_context.hpp"
#include "step_chunks.hpp"

namespace pnx {

// ------------------------------------------------------------------------------------------
// upload validation: every step id must be in 1..n_items
// ------------------------------------------------------------------------------------------
__global__ void k_validate_items(const uint32_t *__restrict__ items, uint64_t n_steps,
                                 uint32_t n_items, uint32_t *bad) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t b = 0;
    for (; i < n_steps; i += stride) {
        uint32_t id = items[i];
        b |= (id == 0u) | (id > n_items);
    }
    if (b) atomicOr(bad, 1u);
}

int launch_validate_items(pnx_ctx *ctx, uint32_t *d_bad) {
    if (ctx->n_steps == 0) return PNX_OK;
    uint64_t want = (ctx->n_steps + 255) / 256;
    int grid = (int)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(k_validate_items, dim3(grid), dim3(256), 0, ctx->stream,
                       (const uint32_t *)ctx->d_items.p, ctx->n_steps, ctx->n_items, d_bad);
    PNX_HIP(ctx, hipGetLastError());
    return PNX_OK;
}

// chunk prefix of the graph (the host knows path_off): once per upload
int ensure_chunk_off(pnx_ctx *ctx) {
    if (ctx->chunk_off_valid) return PNX_OK;
    const uint32_t P = ctx->n_paths;
    ctx->h_chunk_off.assign((size_t)P + 1, 0);
    for (uint32_t p = 0; p < P; ++p)
        ctx->h_chunk_off[p + 1] = ctx->h_chunk_off[p] + (ctx->h_path_off[p + 1] - ctx->h_path_off[p] + RUN_CHUNK - 1) / RUN_CHUNK;
    int rc = ensure(ctx, ctx->d_chunk_off, ((size_t)P + 1) * 8);
    if (rc) return rc;
    PNX_HIP(ctx, hipMemcpyAsync(ctx->d_chunk_off.p, ctx->h_chunk_off.data(), ((size_t)P + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    ctx->chunk_off_valid = true;  // h_chunk_off stays alive with the context
    return PNX_OK;
}

// The step routes are a cross-check module beside the product library: opened on first use, from the library's own directory.
const StepRoutes *step_routes(pnx_ctx *ctx) {
    // (contexts of several host threads may ask at once: one lock around the lookup and its cached outcome.  A module that was
    // not there is looked for again the next time -- it may have been installed since.)
    static std::mutex mu;
    static const StepRoutes *table = nullptr;
    std::string why;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!table) {
            Dl_info info{};
            std::string dir = ".";
            if (dladdr(reinterpret_cast<const void *>(&step_routes), &info) && info.dli_fname) {
                dir = info.dli_fname;
                const size_t slash = dir.rfind('/');
                dir = slash == std::string::npos ? "." : dir.substr(0, slash);
            }
            const std::string path = dir + "/libpanacus_hip_steps.so";
            void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!h) {
                why = std::string("the step routes (PNX_CFG_COVER_VARIANT 0 / 1 / 2) are a cross-check module that this installation does not have: ") + dlerror();
            } else {
                using fn_t = const StepRoutes *(*)();
                fn_t fn = reinterpret_cast<fn_t>(dlsym(h, "pnx_step_routes_table"));
                if (fn) table = fn();
                else why = "libpanacus_hip_steps.so does not export pnx_step_routes_table";
            }
        }
        if (table) return table;
    }
    ctx->fail(PNX_EINVAL, "%s", why.c_str());
    return nullptr;
}

int launch_cover_pass(pnx_ctx *ctx) {
    int rc;
    const bool rows = use_rows(ctx);
    if (!rows) {
        const StepRoutes *sr = step_routes(ctx);
        if (!sr) return PNX_EINVAL;
        if ((rc = sr->sort_run_index(ctx))) return rc;
    }
    const bool use_m = ctx->want_M || (!rows && ctx->last_general_paths > 0);
    const uint64_t m_words = (uint64_t)ctx->n_groups * ctx->n_blocks * BLOCK_WORDS;
    Ticket *tk = ctx->cur;
    tk->used_m = use_m;
    tk->wrote_m = ctx->want_M;
    const size_t hist_bytes = ((size_t)ctx->n_groups + 1) * sizeof(uint64_t);
    tk->block_bytes = 8 * sizeof(uint32_t) + hist_bytes + (((size_t)ctx->n_groups + 15) & ~(size_t)15) + 16;
    tk->block_bytes = (tk->block_bytes + 15) & ~(size_t)15;
    const size_t scratch_off = tk->block_bytes;  // 16 words the workgroups of k_band_tail share
    tk->block_bytes += 64;
    tk->hist_fused = rows && ctx->hist_in_cover && (size_t)ctx->n_groups + 1 <= HIST_FUSED_MAX_BINS;
    ctx->band_splits = 1;
    if (rows && ctx->pass_band) {
        // a small graph: several workgroups share a band, their counters meet in the coverage vector and K2 takes the histogram
        ctx->band_splits = band_route_splits(ctx, ctx->n_groups);
        if (ctx->band_splits > 1) tk->hist_fused = false;
    }
    const size_t rep_off = (tk->block_bytes + 255) & ~(size_t)255;
    if (tk->hist_fused) tk->block_bytes = rep_off + (size_t)HIST_REPLICAS * hist_bytes;
    tk->block_bytes = (tk->block_bytes + 15) & ~(size_t)15;
    if ((rc = ensure(ctx, tk->d_block, tk->block_bytes))) return rc;
    tk->d_hist_rep = tk->hist_fused ? (uint64_t *)((char *)tk->d_block.p + rep_off) : nullptr;
    tk->d_flags = (uint32_t *)tk->d_block.p;
    tk->d_hist = (uint64_t *)((char *)tk->d_block.p + 8 * sizeof(uint32_t));
    tk->d_grp_general = (uint8_t *)tk->d_block.p + 8 * sizeof(uint32_t) + hist_bytes;
    tk->d_band_scratch = (uint32_t *)((char *)tk->d_block.p + scratch_off);
    if ((rc = ensure(ctx, tk->d_countable, ((size_t)ctx->n_items + 1) * sizeof(uint32_t)))) return rc;
    if (use_m && (rc = ensure(ctx, ctx->d_M, (m_words ? m_words : 1) * sizeof(uint32_t)))) return rc;
    const size_t no = ctx->n_ordered ? ctx->n_ordered : 1;
    if ((rc = ensure(ctx, tk->d_ord_tfirst, no * sizeof(uint32_t))) || (rc = ensure(ctx, tk->d_ord_tspan, no * sizeof(uint32_t))) ||
        (rc = ensure(ctx, tk->d_ord_off, no * sizeof(uint64_t))) ||
        (rc = ensure(ctx, tk->d_win_lo, ((no + 63) / 64) * sizeof(uint32_t))) ||
        (rc = ensure(ctx, tk->d_win_hi, ((no + 63) / 64) * sizeof(uint32_t))))
        return rc;
    if (!tk->ev_pre) PNX_HIP(ctx, hipEventCreateWithFlags(&tk->ev_pre, hipEventDisableTiming));
    if (!tk->ev_cov) PNX_HIP(ctx, hipEventCreateWithFlags(&tk->ev_cov, hipEventDisableTiming));
    const bool phased = ctx->s_pre != ctx->s_main;  // three streams chained by events (see pnx_context.hpp)

    // ---- phase 1 (s_pre, behind this pass's K0 if it has one): counters cleared, the index of the
    // ordered paths laid out in visiting order
    // flags, histogram and per-group "general" marks of this pass: one clear
    if (tk->has_reader) {
        PNX_HIP(ctx, hipStreamWaitEvent(ctx->s_pre, tk->ev_reader, 0));
        tk->has_reader = false;
    }
    // (a pass over rows with paths in the order: the kernel that lays the order out clears the block)
    if (!(rows && ctx->n_ordered)) PNX_HIP(ctx, hipMemsetAsync(tk->d_block.p, 0, tk->block_bytes, ctx->s_pre));

    tk->band = rows && ctx->pass_band;
    if (tk->band) {
        // ---- phases 1 + 2 straight over the steps, one read (kernels_band.hip): the first sweep of a graph with sorted paths
        if ((rc = launch_band_phases(ctx, ctx->want_M))) return rc;
    } else if (rows) {
        // ---- phases 1 + 2 over path rows (kernels_rows.hip): no boundary index, no routes
        if ((rc = launch_rows_phases(ctx, ctx->want_M))) return rc;
    } else {
    {
        const StepRoutes *sr = step_routes(ctx);
        if (!sr) return PNX_EINVAL;
        if ((rc = sr->launch_step_phases(ctx, use_m, m_words))) return rc;
    }
    }
    if (phased) {
        PNX_HIP(ctx, hipEventRecord(tk->ev_cov, ctx->s_main));
        PNX_HIP(ctx, hipStreamWaitEvent(ctx->s_post, tk->ev_cov, 0));
    }

    // ---- phase 3 (s_post): the histogram of the coverage vector (K2, kernels_hist.hip); a one-shot pass first adds the steps
    // it spilled, and hands the histogram over in the same kernel if it added one itself
    if (tk->band && (rc = launch_band_tail(ctx, tk, ctx->want_M))) return rc;
    if (!(tk->band && tk->hist_fused) && (rc = launch_hist(ctx, tk))) return rc;
    ctx->M_valid = false;  // settled by the verification in pnx_api
    return PNX_OK;
}

}  // namespace pnx

namespace pnx {
// pnx_preload: the first launch of a kernel loads the code object of its translation unit (tens of ms) and builds the
// kernel's function object; asking for a kernel's attributes does the same, without a launch -- and can be done by a host
// thread that has nothing else to do while the GFA text travels to HBM
void preload_pass(unsigned what) {
    hipFuncAttributes a;
    auto touch = [&a](const void *k) { (void)hipFuncGetAttributes(&a, k); };
    if (what & PNX_PRELOAD_PASS) touch((const void *)k_validate_items);
}
}  // namespace pnx
