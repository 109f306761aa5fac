// pass_pipeline.hip -- what every coverage pass shares, whatever reads the steps: buffers of the pass's ticket, the three
// phases and their streams, the histogram phase behind them; the upload's id check; the chunk prefix of the graph.
//
//   phases 1 + 2   over path rows (kernels_rows.hip), in one read of the steps (kernels_band.hip), or -- cross-check only --
//                  by the step routes of round 2, which live in a module of their own (libpanacus_hip_steps.so,
//                  kernels_cover.hip + kernels_runs.hip) that is opened when PNX_CFG_COVER_VARIANT asks for one
//   phase 3        launch_hist (kernels_hist.hip)
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>
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

// ------------------------------------------------------------------------------------------
// upload validation of a graph the one-shot route may take: the same one read of the steps also says where its paths
// turn round or jump back
// ------------------------------------------------------------------------------------------
// The paths of a pangenome run through the ids of a sorted graph in order -- except at LARGE rearrangements: an inversion
// (the path runs downwards for a stretch), a duplication or a translocation (the ids jump back and run on).  No single
// position per band edge deals the steps of such a path to bands (kernels_band.hip: a whole inverted stretch would fall to the
// bands at its two ends and be spilled), but every PIECE between two such breaks is a path that follows the ids.  So the
// read of the steps that validates the ids at upload (the reference panics on unknown nodes, graph_broker/util.rs:1021) also
// leaves a summary of every chunk of 4096 steps -- five evenly spaced ids, and how many steps go up / down --, and the host
// cuts every path where the summaries turn round or jump back by more than two bands for at least two chunks
// (path_cuts_from_chunks; refine_path_cuts then moves every cut to the step where the ids jump).  The one-shot pass then takes the pieces as entries of their own in the visiting order, under
// their path's group: AbacusByTotal::coverage (abacus.rs:727-742) counts a group once per item however its steps are split.
// Small disorder (pansyn-v1r's 64-step blocks) never makes a cut: it is spilled as before.
__global__ __launch_bounds__(256) void k_chunk_summaries(const uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off,
                                                         const uint64_t *__restrict__ chunk_off, uint32_t n_paths, uint64_t n_chunks,
                                                         uint32_t n_items, ChunkSummary *__restrict__ out, uint32_t *__restrict__ bad) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const RunChunk ch = chunk_of(c, chunk_off, path_off, n_paths);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t a0 = ch.start & ~3ull;
    const uint32_t head = (uint32_t)(ch.start - a0), nal = head + ch.len;
    uint32_t up = 0, down = 0;
    bool b = false;
    for (uint32_t r0 = 0; r0 < nal; r0 += 1024u) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t q = r0 + (uint32_t)u * 256u + lane * 4u;
            v[u] = q < nal ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(items + a0 + q)) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t q = r0 + (uint32_t)u * 256u + lane * 4u;
            const uint32_t ids[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = q + (uint32_t)e - head < ch.len;  // (unsigned: false before the chunk too)
                if (in && ids[e] - 1u >= n_items) b = true;
                if (e < 3 && in && q + (uint32_t)e + 1u - head < ch.len) {  // the steps inside one 16-byte load: three of every four
                    up += ids[e + 1] > ids[e] ? 1u : 0u;
                    down += ids[e + 1] < ids[e] ? 1u : 0u;
                }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        up += __shfl_down(up, o);
        down += __shfl_down(down, o);
    }
    if (__ballot(b) && lane == 0) atomicOr(bad, 1u);
    if (lane < 5) out[c].s[lane] = items[ch.start + (uint64_t)(ch.len - 1u) * lane / 4u];
    if (lane == 0) {
        out[c].up = up;
        out[c].down = down;
    }
}

// A cut found from the summaries lies at a chunk boundary; the break itself -- the step where the ids jump -- is somewhere in
// the chunk before it or the one behind.  One wave per cut looks at those 8192 steps and moves the cut to the largest jump
// between two consecutive steps (a path that follows the ids moves by a few ids per step; at a break by more than two bands).
__global__ __launch_bounds__(64) void k_refine_cuts(const uint32_t *__restrict__ items, const uint64_t *__restrict__ lo, const uint64_t *__restrict__ hi,
                                                    uint64_t *__restrict__ cuts) {
    const uint32_t lane = threadIdx.x;
    const uint64_t a = lo[blockIdx.x], z = hi[blockIdx.x];  // pairs (j - 1, j) for j in [a + 1, z)
    uint32_t best = 0;
    uint64_t at = cuts[blockIdx.x];
    for (uint64_t j = a + 1 + lane; j < z; j += 64) {
        const uint32_t x = items[j - 1], y = items[j];
        const uint32_t d = x > y ? x - y : y - x;
        if (d > best) {
            best = d;
            at = j;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t ob = __shfl_xor(best, o);
        const uint64_t oa = __shfl_xor(at, o);
        if (ob > best || (ob == best && oa < at)) {
            best = ob;
            at = oa;
        }
    }
    if (lane == 0) cuts[blockIdx.x] = at;
}

int refine_path_cuts(pnx_ctx *ctx) {
    const size_t n = ctx->h_cuts.size();
    if (!n) return PNX_OK;
    std::vector<uint64_t> lo(n), hi(n);
    for (uint32_t p = 0; p < ctx->n_paths; ++p)
        for (uint32_t c = ctx->h_cut_off[p]; c < ctx->h_cut_off[p + 1]; ++c) {
            const uint64_t cut = ctx->h_cuts[c], ps = ctx->h_path_off[p], pe = ctx->h_path_off[p + 1];
            lo[c] = cut - ps > RUN_CHUNK ? cut - RUN_CHUNK : ps;
            hi[c] = pe - cut > RUN_CHUNK ? cut + RUN_CHUNK : pe;
        }
    int rc;
    if ((rc = ensure(ctx, ctx->d_chunk_sum, 3 * n * 8))) return rc;  // (the summaries are on the host: their buffer serves)
    uint64_t *d = (uint64_t *)ctx->d_chunk_sum.p;
    PNX_HIP(ctx, hipMemcpyAsync(d, lo.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    PNX_HIP(ctx, hipMemcpyAsync(d + n, hi.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    PNX_HIP(ctx, hipMemcpyAsync(d + 2 * n, ctx->h_cuts.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_refine_cuts, dim3((unsigned)n), dim3(64), 0, ctx->stream, (const uint32_t *)ctx->d_items.p, (const uint64_t *)d,
                       (const uint64_t *)(d + n), d + 2 * n);
    PNX_HIP(ctx, hipGetLastError());
    PNX_HIP(ctx, hipMemcpyAsync(ctx->h_cuts.data(), d + 2 * n, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PNX_OK;
}

int launch_chunk_summaries(pnx_ctx *ctx, uint32_t *d_bad) {
    if (ctx->n_steps == 0) return PNX_OK;
    int rc;
    if ((rc = ensure_chunk_off(ctx))) return rc;
    const uint64_t n_chunks = ctx->h_chunk_off[ctx->n_paths];
    if ((n_chunks + 3) / 4 > 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "too many path chunks");
    if ((rc = ensure(ctx, ctx->d_chunk_sum, n_chunks * sizeof(ChunkSummary)))) return rc;
    hipLaunchKernelGGL(k_chunk_summaries, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_items.p,
                       (const uint64_t *)ctx->d_path_off.p, (const uint64_t *)ctx->d_chunk_off.p, ctx->n_paths, n_chunks, ctx->n_items,
                       (ChunkSummary *)ctx->d_chunk_sum.p, d_bad);
    PNX_HIP(ctx, hipGetLastError());
    ctx->h_chunk_sum.resize(n_chunks);
    PNX_HIP(ctx, hipMemcpyAsync(ctx->h_chunk_sum.data(), ctx->d_chunk_sum.p, n_chunks * sizeof(ChunkSummary), hipMemcpyDeviceToHost, ctx->stream));
    return PNX_OK;
}

// Where the paths are cut (absolute step positions, chunk boundaries): h_cut_off[p] .. h_cut_off[p + 1] into h_cuts.
// A chunk lies between the second smallest and the second largest of its five ids (one id from elsewhere does not move
// it) and runs the way most of its steps go; a chunk whose steps go both ways (a quarter against the rest) says nothing.
// A cut in front of chunk c: c and the chunk behind it run on from each other, and c does not run on from the
// chunk before it -- it runs the other way, or it begins more than two bands behind where that chunk began.
void path_cuts_from_chunks(pnx_ctx *ctx) {
    constexpr uint32_t SLACK = 2u << 13, MIN_RUN = 2, MAX_CUTS = 62;
    const uint32_t P = ctx->n_paths;
    ctx->h_cut_off.assign((size_t)P + 1, 0);
    ctx->h_cuts.clear();
    if (ctx->h_chunk_sum.empty() || ctx->h_chunk_off.size() != (size_t)P + 1) return;
    struct Ck {
        uint32_t lo, hi;
        int dir;
    };
    std::vector<Ck> ck;
    for (uint32_t p = 0; p < P; ++p) {
        const uint64_t c0 = ctx->h_chunk_off[p], c1 = ctx->h_chunk_off[p + 1];
        const size_t first_cut = ctx->h_cuts.size();
        if (c1 - c0 >= 2 * MIN_RUN) {
            ck.resize(c1 - c0);
            for (uint64_t c = c0; c < c1; ++c) {
                const ChunkSummary &s = ctx->h_chunk_sum[c];
                uint32_t v[5] = {s.s[0], s.s[1], s.s[2], s.s[3], s.s[4]};
                std::sort(v, v + 5);
                const uint32_t mn = s.up < s.down ? s.up : s.down, all = s.up + s.down;
                const int dir = (all > 64 && (uint64_t)mn * 4 > all) ? 0 : (s.up > s.down ? 1 : (s.down > s.up ? -1 : 0));
                ck[c - c0] = Ck{v[1], v[3], dir};
            }
            auto runs_on = [&](const Ck &a, const Ck &b) {  // b continues a
                if (a.dir == 0 || b.dir == 0) return true;
                if (a.dir != b.dir) return false;
                return a.dir > 0 ? (uint64_t)b.lo + SLACK >= a.lo : (uint64_t)a.hi + SLACK >= b.hi;
            };
            const size_t n = ck.size();
            size_t last = 0;  // the chunk the current piece was last seen to run on from
            for (size_t i = 1; i + MIN_RUN <= n; ++i) {
                if (ck[i].dir == 0) continue;
                if (runs_on(ck[last], ck[i])) {
                    last = i;
                    continue;
                }
                bool fresh = true;  // a new piece: it holds together for MIN_RUN chunks
                for (size_t j = i + 1; j < i + MIN_RUN && fresh; ++j) fresh = ck[j].dir != 0 && runs_on(ck[j - 1], ck[j]) && !runs_on(ck[last], ck[j]);
                if (fresh) {
                    ctx->h_cuts.push_back(ctx->h_path_off[p] + (uint64_t)i * RUN_CHUNK);
                    last = i;
                }
            }
            if (ctx->h_cuts.size() - first_cut > MAX_CUTS) ctx->h_cuts.resize(first_cut);  // (a path in pieces all over: no pieces)
        }
        ctx->h_cut_off[p + 1] = (uint32_t)ctx->h_cuts.size();
    }
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
