// pnx_api.hip -- extern "C" entry points of libpanacus_hip.so (see include/panacus_amd.h).
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "pnx_context.hpp"

static std::string g_init_err;

namespace pnx {

// PNX_GUARD_ALLOC=1 (a debugging aid, never the default): every device buffer is a mapping of its own (HIP's virtual memory
// management calls) that ENDS where the buffer and its 256 bytes of over-read allowance end, with an unmapped page behind it, and
// the address range of a released buffer is never handed out again -- a kernel that reads or writes past a buffer, or through a
// stale pointer, raises a GPU memory fault at once instead of touching whatever the allocator placed there (the intermittent
// abort of whole sessions at the end of round 5, DESIGN.md section 2, was hunted with this).
namespace {
struct GuardRec {
    void *base;
    size_t mapped;
    hipMemGenericAllocationHandle_t h;
};
std::mutex g_guard_mu;
std::unordered_map<void *, GuardRec> g_guard;
bool guard_mode() {
    static const bool on = [] {
        const char *e = getenv("PNX_GUARD_ALLOC");
        return e && e[0] && e[0] != '0';
    }();
    return on;
}
hipError_t guard_malloc(void **out, size_t bytes, int device) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran < 4096) gran = 4096;
    const size_t mapped = (bytes + gran - 1) / gran * gran;
    void *base = nullptr;
    if ((e = hipMemAddressReserve(&base, mapped + gran, gran, nullptr, 0)) != hipSuccess) return e;
    hipMemGenericAllocationHandle_t h;
    if ((e = hipMemCreate(&h, mapped, &prop, 0)) != hipSuccess) return e;
    if ((e = hipMemMap(base, mapped, 0, h, 0)) != hipSuccess) return e;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(base, mapped, &acc, 1)) != hipSuccess) return e;
    void *p = (char *)base + (mapped - bytes);
    {
        std::lock_guard<std::mutex> g(g_guard_mu);
        g_guard[p] = GuardRec{base, mapped, h};
    }
    *out = p;
    return hipSuccess;
}
void guard_free(void *p) {
    GuardRec r;
    {
        std::lock_guard<std::mutex> g(g_guard_mu);
        auto it = g_guard.find(p);
        if (it == g_guard.end()) {
            (void)hipFree(p);
            return;
        }
        r = it->second;
        g_guard.erase(it);
    }
    (void)hipDeviceSynchronize();  // (hipFree waits for the device as well)
    (void)hipMemUnmap(r.base, r.mapped);
    (void)hipMemRelease(r.h);
    // the reservation stays: the range is never mapped again, so a stale pointer faults for the rest of the process
}
// (buffers beyond PNX_GUARD_MAX_MB [256] keep hipMalloc: multi-GB mappings made of minimum-granularity pages raised faults INSIDE
// their own range on this stack -- k_pansyn_fill writing a 15.6 GB mapping, found with AMD_LOG_LEVEL=3 -- and the over-reads this
// mode hunts are at the ends of small and medium buffers)
size_t guard_max_bytes() {
    static const size_t v = [] {
        const char *e = getenv("PNX_GUARD_MAX_MB");
        return (size_t)(e ? strtoull(e, nullptr, 10) : 256ull) << 20;
    }();
    return v;
}
hipError_t dev_malloc(void **out, size_t bytes, int device) {
    return guard_mode() && bytes <= guard_max_bytes() ? guard_malloc(out, bytes, device) : hipMalloc(out, bytes);
}
void dev_free(void *p) {
    if (guard_mode()) guard_free(p);
    else (void)hipFree(p);
}
}  // namespace

int ensure(pnx_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~(size_t)255;
    if (!b.borrowed && b.cap >= bytes) return PNX_OK;
    if (b.p) {
        if (!b.borrowed) dev_free(b.p);
        b.p = nullptr;
        b.cap = 0;
        b.borrowed = false;
    }
    hipError_t e = dev_malloc(&b.p, bytes + 256, ctx->device);  // +256: vector loads may over-read a tail
    if (e != hipSuccess) {
        b.p = nullptr;
        return ctx->fail(PNX_ENOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    b.cap = bytes;
    return PNX_OK;
}

void release(DevBuf &b) {
    if (b.p && !b.borrowed) dev_free(b.p);
    b.p = nullptr;
    b.cap = 0;
    b.borrowed = false;
}

static hipEvent_t prof_event(pnx_ctx *ctx) {
    if (!ctx->prof.pool.empty()) {
        hipEvent_t e = ctx->prof.pool.back();
        ctx->prof.pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void prof_begin(pnx_ctx *ctx, int slot, hipStream_t stream) {
    ctx->prof.open = ctx->prof.on && ((ctx->prof.mask >> slot) & 1u);
    if (ctx->prof.open && ctx->prof.every > 1) ctx->prof.open = (ctx->prof.seen[slot]++ % ctx->prof.every) == 0;
    if (!ctx->prof.open) return;
    ctx->prof.open_stream = stream ? stream : ctx->stream;
    Profile::Pending pd{prof_event(ctx), prof_event(ctx), slot};
    (void)hipEventRecord(pd.a, ctx->prof.open_stream);
    ctx->prof.pending.push_back(pd);
}

void prof_end(pnx_ctx *ctx) {
    if (!ctx->prof.open || ctx->prof.pending.empty()) return;
    ctx->prof.open = false;
    (void)hipEventRecord(ctx->prof.pending.back().b, ctx->prof.open_stream);
}

int drain_streams(pnx_ctx *ctx) {
    for (hipStream_t st : {ctx->stream_pre, ctx->stream, ctx->stream_post})
        if (st) PNX_HIP(ctx, hipStreamSynchronize(st));
    return PNX_OK;
}

// which streams the pass being enqueued uses (pnx_context.hpp): three chained ones for a plain
// histogram pass, one when the pass also writes or merges the presence matrix
static int choose_pass_streams(pnx_ctx *ctx) {
    const bool use_m = ctx->want_M || (!use_rows(ctx) && ctx->last_general_paths > 0);
    // (a one-shot pass over the steps is one chain of three kernels: on one stream a dependent kernel follows within ~2 us, across
    // streams the event hand-over costs ~25 us each -- nothing to overlap it with either)
    // (the three-stream arrangement pays when passes overlap each other.  A context of pnx_init has the streams from the start;
    // a PNX_INIT_ONE_SHOT context makes them the first time a pass is enqueued while another is still in flight -- a command that
    // waits for each histogram before it asks for the next never does, and its passes run on one stream without event
    // hand-overs)
    const bool phased = ctx->overlap_phases && ctx->cover_variant >= 2 && !use_m && !ctx->pass_band &&
                        (ctx->stream_pre != nullptr || ctx->tk_count > 0);
    if (ctx->tk_count && phased != ctx->last_pass_phased) {
        int rc = drain_streams(ctx);  // a pass in flight took the other arrangement: let it finish on the device
        if (rc) return rc;
    }
    ctx->last_pass_phased = phased;
    if (phased && !ctx->stream_pre) PNX_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_pre, hipStreamNonBlocking));
    if (phased && !ctx->stream_post) PNX_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_post, hipStreamNonBlocking));
    ctx->s_main = ctx->stream;
    ctx->s_pre = phased ? ctx->stream_pre : ctx->stream;
    ctx->s_post = phased ? ctx->stream_post : ctx->stream;
    return PNX_OK;
}

int prof_resolve(pnx_ctx *ctx, bool wait) {
    std::vector<Profile::Pending> keep;
    for (auto &pd : ctx->prof.pending) {
        if (!wait && hipEventQuery(pd.b) != hipSuccess) {
            keep.push_back(pd);
            continue;
        }
        float ms = 0.f;
        if (hipEventSynchronize(pd.b) == hipSuccess && hipEventElapsedTime(&ms, pd.a, pd.b) == hipSuccess) {
            ctx->prof.ms[pd.slot] += ms;
            ctx->prof.launches[pd.slot] += 1;
        }
        ctx->prof.pool.push_back(pd.a);
        ctx->prof.pool.push_back(pd.b);
    }
    ctx->prof.pending.swap(keep);
    return PNX_OK;
}

static void invalidate_results(pnx_ctx *ctx) {
    if (ctx->tk_count) (void)drain_streams(ctx);
    ctx->d_countable_done = nullptr;
    for (auto &t : ctx->tk) t.in_flight = false;
    ctx->tk_count = 0;
    ctx->tk_oldest = ctx->tk_next;
    ctx->last_done = nullptr;
    ctx->hist_valid = false;
    ctx->M_valid = false;
    ctx->growth_pending = false;
}

static void drop_run_index(pnx_ctx *ctx) {
    ctx->n_runs = 0;
    ctx->n_run_paths = 0;
    ctx->n_scatter_paths = 0;
    ctx->runs_sorted = false;
}

static void set_geometry(pnx_ctx *ctx) {
    drop_run_index(ctx);
    ctx->chunk_off_valid = false;
    ctx->n_blocks = (uint32_t)(((uint64_t)ctx->n_items + 1 + BLOCK_ITEMS - 1) / BLOCK_ITEMS);
    ctx->n_tiles = (ctx->n_blocks + ctx->tile_blocks - 1) / ctx->tile_blocks;
    ctx->index_valid = false;
    ctx->spans_valid = false;
    ctx->order_normalized = false;
    ctx->wplanes_valid = false;
    ctx->wdigits_valid = false;
    ctx->last_general_paths = 0;
    ctx->rows_valid = false;
    ctx->band_failed = false;
    ctx->n_band_passes = 0;
    ctx->n_spilled_total = 0;
    ctx->n_spilled_last = 0;
}

// the ticket's pinned result block [flags u32[8] | hist (G+1) u64], before the pass is launched: its publishing kernel may
// write it directly
static int ensure_host_block(pnx_ctx *ctx, Ticket *t) {
    const size_t bytes = 8 * sizeof(uint32_t) + ((size_t)ctx->n_groups + 1) * sizeof(uint64_t);
    if (t->h_cap < bytes) {
        if (t->h_block) (void)hipHostFree(t->h_block);
        t->h_block = nullptr;
        t->h_block_mapped = nullptr;
        t->h_cap = 0;
        PNX_HIP(ctx, hipHostMalloc(&t->h_block, bytes, hipHostMallocDefault));
        t->h_cap = bytes;
        PNX_HIP(ctx, hipHostGetDevicePointer(&t->h_block_mapped, t->h_block, 0));
    }
    t->h_flags = (uint32_t *)t->h_block;
    t->h_hist = (uint64_t *)((char *)t->h_block + 8 * sizeof(uint32_t));
    return PNX_OK;
}

static int stage_results(pnx_ctx *ctx, Ticket *t) {
    const size_t bytes = 8 * sizeof(uint32_t) + ((size_t)ctx->n_groups + 1) * sizeof(uint64_t);
    if (t->done && t->done_blocking != ctx->blocking_sync) {
        (void)hipEventDestroy(t->done);
        t->done = nullptr;
    }
    if (!t->done) {
        PNX_HIP(ctx, hipEventCreateWithFlags(&t->done, hipEventDisableTiming | (ctx->blocking_sync ? hipEventBlockingSync : 0)));
        t->done_blocking = ctx->blocking_sync;
    }
    if (!t->host_written) PNX_HIP(ctx, hipMemcpyAsync(t->h_block, t->d_block.p, bytes, hipMemcpyDeviceToHost, ctx->s_post));
    PNX_HIP(ctx, hipEventRecord(t->done, ctx->s_post));
    return PNX_OK;
}

// enqueue one pass into the next free ticket
static int enqueue_pass(pnx_ctx *ctx) {
    Ticket *t = ctx->cur;  // chosen by pnx_hist_async (its index may already be in the making)
    int rc = ensure_host_block(ctx, t);
    if (rc) return rc;
    if ((rc = launch_cover_pass(ctx))) return rc;
    if ((rc = comm_reduce_pass(ctx, t))) return rc;  // multi-GPU: global flags + histogram (no-op without a communicator)
    if ((rc = stage_results(ctx, t))) return rc;
    t->in_flight = true;
    ctx->tk_next = (ctx->tk_next + 1) % pnx_ctx::N_TICKETS;
    ctx->tk_count += 1;
    return PNX_OK;
}

// wait for the oldest pass and verify it (no tile-monotonicity violation, no unserved general
// path); a failed pass is re-run with the paths it flagged sent down the scatter route
static int settle_oldest(pnx_ctx *ctx) {
    if (ctx->tk_count == 0) return PNX_OK;
    Ticket *t = &ctx->tk[ctx->tk_oldest];
    for (int attempt = 0; attempt < 4; ++attempt) {
        PNX_HIP(ctx, hipEventSynchronize(t->done));
        prof_resolve(ctx, false);
        // flags[5]: a one-shot pass (of this rank, or -- the flags are reduced with the histogram -- of any rank of the communicator)
        // met paths that stray from the order of the ids by more than its spill list or its scan budget hold: the pass is void.
        // Path rows serve any path (and validate the ids on the way: an id outside 1..n_items also ends up here).  Every rank
        // sees the same reduced flag and runs again, whichever route its own pass took: the collectives stay matched.
        if ((t->band || (ctx->comm && ctx->comm_reduce_hist)) && t->h_flags[5] != 0) {
            int rc = drain_streams(ctx);
            if (rc) return rc;
            ctx->band_failed = true;
            ctx->pass_band = false;
            ctx->n_reruns += 1;
            // (under a communicator flags[5] is the SUM over the ranks of a set of bits: which bits cannot be told -- two ranks
            // with bit 64 give 128 -- so any void pass there leaves the bitmaps to be cleared)
            if ((t->h_flags[5] & 64u) || (ctx->comm && ctx->comm_reduce_hist)) ctx->loose_dirty = true;
            // (why: 1 inconsistent index entry, 2 spill list full, 4 scan volume, 8 the index's probes met steps from elsewhere,
            // 16 more loose groups (or steps in them) than a pass takes in, 32 an id that is no item, 64 the marking workgroups did not meet)
            if (getenv("PNX_BAND_DEBUG"))
                fprintf(stderr, "[panacus_amd] one-shot pass void: flags[5] = %u, spilled %u in %u bursts, loose groups %u\n", t->h_flags[5],
                        t->h_flags[6], t->h_flags[7], t->h_flags[3]);
            if ((rc = ensure_rows(ctx, true))) return rc;
            ctx->cur = t;
            ctx->want_M = t->wrote_m;
            if ((rc = choose_pass_streams(ctx)) || (rc = ensure_host_block(ctx, t)) || (rc = launch_cover_pass(ctx)) ||
                (rc = comm_reduce_pass(ctx, t)) || (rc = stage_results(ctx, t)))
                return rc;
            continue;
        }
        const bool used_m = t->used_m;  // as launched, not as the context stands now
        // [0] tile-monotonicity violations found by K1, [1] scatter-route paths in the order,
        // [2] non-monotone paths that are not classified yet, [4] internal run-index check
        const bool need_build = t->h_flags[0] != 0 || t->h_flags[2] != 0;
        const bool bad = need_build || (t->h_flags[1] != 0 && !used_m);
        if (t->h_flags[4] != 0) return ctx->fail(PNX_EHIP, "run index inconsistency (internal error)");
        if (!bad) {
            static const bool band_debug = getenv("PNX_BAND_DEBUG") != nullptr;
            if (t->band && band_debug && ctx->d_entry_loose.p && ctx->d_group_loose.p) {  // (what the pass's tail must have set back)
                std::vector<uint32_t> ef(ctx->n_entries), gf(ctx->n_groups), st(20);
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(ef.data(), ctx->d_entry_loose.p, ef.size() * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(gf.data(), ctx->d_group_loose.p, gf.size() * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(st.data(), ctx->d_band_probe.p, 80, hipMemcpyDeviceToHost);
                size_t ne = 0, ng = 0;
                for (uint32_t x : ef) ne += x != 0;
                for (uint32_t x : gf) ng += x != 0;
                if (ne || ng || st[0] || st[1] || st[2] || st[3])
                    fprintf(stderr, "[panacus_amd] LEFT BEHIND by a one-shot pass: %zu entry flags, %zu group flags, state %u %u %u %u (loose groups of the pass: %u)\n", ne, ng,
                            st[0], st[1], st[2], st[3], t->h_flags[3]);
            }
            if (t->band) {
                ctx->n_spilled_last = t->h_flags[6];
                ctx->n_spill_bursts_last = t->h_flags[7];
                ctx->n_loose_last = t->h_flags[3];
                ctx->n_spilled_total += t->h_flags[6];
            }
            ctx->last_general_paths = t->h_flags[1];
            t->in_flight = false;
            ctx->tk_oldest = (ctx->tk_oldest + 1) % pnx_ctx::N_TICKETS;
            ctx->tk_count -= 1;
            ctx->last_done = t;
            ctx->d_countable_done = &t->d_countable;
            ctx->hist_valid = true;
            ctx->M_valid = t->wrote_m && ctx->tk_count == 0;
            return PNX_OK;
        }
        // a younger pass (if any) ran with the same stale classification; it fails and is
        // re-run at its own settle
        int drc = drain_streams(ctx);
        if (drc) return drc;
        ctx->n_reruns += 1;
        int rc;
        if (need_build) {
            // cut the non-monotone paths into runs (tile route) or leave them to the scatter route
            const StepRoutes *sr = step_routes(ctx);
            if (!sr) return PNX_EINVAL;
            if ((rc = sr->build_run_index(ctx))) return rc;
            ctx->last_general_paths = ctx->n_scatter_paths;
        } else {
            ctx->last_general_paths = t->h_flags[1];  // > 0: the re-run allocates and merges M
        }
        ctx->cur = t;
        if ((rc = choose_pass_streams(ctx))) return rc;
        if ((rc = ensure_host_block(ctx, t))) return rc;
        if ((rc = launch_cover_pass(ctx))) return rc;
        // the flags were reduced over all ranks, so every rank is here: the collectives stay matched
        if ((rc = comm_reduce_pass(ctx, t))) return rc;
        if ((rc = stage_results(ctx, t))) return rc;
    }
    return ctx->fail(PNX_EHIP, "coverage pass did not converge (internal error)");
}

static int settle_all(pnx_ctx *ctx) {
    while (ctx->tk_count) {
        int rc = settle_oldest(ctx);
        if (rc) return rc;
    }
    return PNX_OK;
}

}  // namespace pnx

using namespace pnx;

extern "C" {

const char *pnx_version(void) { return "panacus_amd 0.5.0 (gfx950)"; }

int pnx_abi_version(void) { return PNX_ABI_VERSION; }

const char *pnx_last_error(const pnx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_init_err.c_str(); }

int pnx_preload(int device, uint32_t what) {
    if (hipSetDevice(device) != hipSuccess) return PNX_ENODEV;
    pnx::preload_gfa(what);
    pnx::preload_cut(what);
    pnx::preload_relabel(what);
    pnx::preload_pass(what);
    pnx::preload_band(what);
    pnx::preload_rows(what);
    pnx::preload_hist(what);
    pnx::preload_closed_form(what);
    pnx::preload_growth(what);
    pnx::preload_pairs(what);
    pnx::preload_pairs_mfma(what);
    return PNX_OK;
}

int pnx_init(pnx_ctx **out, int device) { return pnx_init_flags(out, device, 0u); }

int pnx_init_flags(pnx_ctx **out, int device, uint32_t flags) {
    if (!out) return PNX_EINVAL;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_init_err = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                     " -- panacus_amd has no CPU fallback";
        return PNX_ENODEV;
    }
    if (device < 0 || device >= n) {
        g_init_err = "device ordinal out of range";
        return PNX_EINVAL;
    }
    pnx_ctx *ctx = new (std::nothrow) pnx_ctx();
    if (!ctx) return PNX_ENOMEM;
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&ctx->prop, device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        // stream_pre / stream_post cost 9 ms each to create and a one-shot caller never uses them (PNX_INIT_ONE_SHOT: made when
        // passes first overlap).  Made HERE they are the process's first streams and get hardware queues of their own; made
        // late, behind the streams of the closed forms, they share queues and the phases of overlapping passes no longer
        // run beside each other (0.18 instead of 0.10 ms per pipelined pass on cfg3).
        (!(flags & PNX_INIT_ONE_SHOT) && ((e = hipStreamCreateWithFlags(&ctx->stream_pre, hipStreamNonBlocking)) != hipSuccess ||
                                          (e = hipStreamCreateWithFlags(&ctx->stream_post, hipStreamNonBlocking)) != hipSuccess))) {
        g_init_err = std::string("device initialisation failed: ") + hipGetErrorString(e);
        delete ctx;
        return PNX_EHIP;
    }
    if (const char *v = getenv("PNX_COVER_VARIANT")) {  // default of PNX_CFG_COVER_VARIANT (cross-check runs of a whole host)
        if (v[0] >= '0' && v[0] <= '3' && v[1] == 0) ctx->cover_variant = v[0] - '0';
    }
    if (const char *v = getenv("PNX_ROWS_KERNEL")) {  // default of PNX_CFG_ROWS_KERNEL
        if (v[0] >= '0' && v[0] <= '2' && v[1] == 0) ctx->rows_kernel = v[0] - '0';
    }
    if (const char *v = getenv("PNX_HIST_IN_COVER")) {  // default of PNX_CFG_HIST_IN_COVER
        if ((v[0] == '0' || v[0] == '1') && v[1] == 0) ctx->hist_in_cover = v[0] == '1';
    }
    *out = ctx;
    return PNX_OK;
}

void pnx_free(pnx_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)pnx_comm_free(ctx);
    (void)drain_streams(ctx);
    if (ctx->stream_cf) (void)hipStreamSynchronize(ctx->stream_cf);
    prof_resolve(ctx);
    for (auto e : ctx->prof.pool) (void)hipEventDestroy(e);
    for (DevBuf *b : {&ctx->d_items, &ctx->d_path_off, &ctx->d_weights, &ctx->d_exclude, &ctx->d_ord_path,
                      &ctx->d_ord_group, &ctx->d_tile_idx, &ctx->d_tfirst, &ctx->d_tspan, &ctx->d_idx_off, &ctx->d_path_class, &ctx->d_wdigits, &ctx->d_unsorted, &ctx->d_sorted_coff, &ctx->d_sorted_path, &ctx->d_flags,
                      &ctx->d_M, &ctx->d_perms, &ctx->d_cov_thr, &ctx->d_qtab,
                      &ctx->d_cmask, &ctx->d_wplanes, &ctx->d_growth_out, &ctx->d_thr_meta, &ctx->d_run_start,
                      &ctx->d_run_len, &ctx->d_run_tile, &ctx->d_run_path, &ctx->d_srun_start, &ctx->d_srun_len,
                      &ctx->d_srun_group, &ctx->d_run_tile_off, &ctx->d_inter, &ctx->d_pair_partial, &ctx->d_plain, &ctx->d_new_of_old, &ctx->d_old_of_new, &ctx->d_countable_ext, &ctx->d_steps12, &ctx->d_path_mono, &ctx->d_rows, &ctx->d_row_base, &ctx->d_id_minmax, &ctx->d_rt_first, &ctx->d_rt_span, &ctx->d_chunk_off, &ctx->d_rb[0], &ctx->d_rb[1], &ctx->d_rb[2], &ctx->d_rb[3], &ctx->d_rb[4], &ctx->d_rb[5], &ctx->d_rs[0], &ctx->d_rs[1], &ctx->d_rs[2], &ctx->d_rs[3], &ctx->d_rs[4], &ctx->d_rs[5], &ctx->d_cf[0], &ctx->d_cf[1], &ctx->d_cf[2],
                      &ctx->d_cf[3], &ctx->d_cf[4], &ctx->d_cf[5], &ctx->d_comm_word, &ctx->d_gfa_text, &ctx->d_walk_node, &ctx->d_walk_back, &ctx->d_name_tab, &ctx->d_link_uv, &ctx->d_link_oo, &ctx->d_spill, &ctx->d_spill_dir, &ctx->d_spill_set, &ctx->d_band_probe, &ctx->d_group_loose, &ctx->d_entry_loose, &ctx->d_loose_bits, &ctx->d_chunk_sum, &ctx->d_ent_start, &ctx->d_ent_len, &ctx->d_ent_group})
        release(*b);
    if (ctx->h_cf) (void)hipHostFree(ctx->h_cf);
    if (ctx->ev_cf) (void)hipEventDestroy(ctx->ev_cf);
    for (auto &g : ctx->gslot) {
        if (g.stream) (void)hipStreamSynchronize(g.stream);
        if (g.h_io) (void)hipHostFree(g.h_io);
        if (g.done) (void)hipEventDestroy(g.done);
        if (g.stream) (void)hipStreamDestroy(g.stream);
    }
    {
        auto &t = ctx->gtab;
        for (DevBuf *b : {&t.d_par, &t.d_L, &t.d_nf, &t.d_mf, &t.d_mq, &t.d_pm, &t.d_lsq, &t.d_terms, &t.d_sum}) release(*b);
        if (t.ready) (void)hipEventDestroy(t.ready);
    }
    for (auto &t : ctx->tk) {
        if (t.h_block) (void)hipHostFree(t.h_block);
        if (t.done) (void)hipEventDestroy(t.done);
        if (t.ev_pre) (void)hipEventDestroy(t.ev_pre);
        if (t.ev_cov) (void)hipEventDestroy(t.ev_cov);
        if (t.ev_reader) (void)hipEventDestroy(t.ev_reader);
        for (DevBuf *b : {&t.d_block, &t.d_ord_tfirst, &t.d_ord_tspan, &t.d_ord_off, &t.d_win_lo, &t.d_win_hi, &t.d_countable, &t.d_tile_idx_own, &t.d_group_first, &t.d_band_clist, &t.d_band_ccnt})
            release(*b);
    }
    if (ctx->stream_pre) (void)hipStreamDestroy(ctx->stream_pre);
    if (ctx->stream_post) (void)hipStreamDestroy(ctx->stream_post);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream_cf) (void)hipStreamDestroy(ctx->stream_cf);
    delete ctx;
}

// an upload that does not tokenise: a copy of the GFA text that pnx_gfa_text_upload left in HBM (a host that offers its text
// before it knows the segment names are not decimal) is of no further use -- on a multi-GB graph it is the memory the item
// table, the rows and the coverage vectors are about to ask for
static void drop_gfa_text(pnx_ctx *ctx) {
    release(ctx->d_gfa_text);
    ctx->gfa_text_host = nullptr;
    ctx->gfa_text_bytes = 0;
}

// what every upload starts with: results, order and derived step data of the old graph are void
static void begin_upload(pnx_ctx *ctx) {
    invalidate_results(ctx);
    ctx->h_chunk_sum.clear();
    ctx->h_cuts.clear();
    ctx->h_cut_off.clear();
    ctx->h_jumbled.clear();
    ctx->h_sorted_at.clear();
    ctx->n_sorted_copies = 0;
    ctx->entries_valid = false;
    ctx->have_csr = false;
    ctx->have_order = false;
    ctx->steps_prepared = false;
    if (ctx->d_steps12.borrowed) release(ctx->d_steps12);
    if (ctx->d_path_mono.borrowed) release(ctx->d_path_mono);
    for (pnx::DevBuf *b : {&ctx->d_rows, &ctx->d_row_base, &ctx->d_id_minmax, &ctx->d_rt_first, &ctx->d_rt_span})
        if (b->borrowed) release(*b);
    if (ctx->d_unsorted.borrowed) {
        release(ctx->d_unsorted);
        release(ctx->d_sorted_coff);
        release(ctx->d_sorted_path);
    }
    ctx->n_sorted_paths = 0;
    ctx->n_unsorted = 0;
}

// ... and ends with: d_items / d_path_off / h_path_off (and d_exclude when exclude_resident) are in place
static int finish_upload(pnx_ctx *ctx, uint64_t S, uint32_t n_paths, uint32_t n_items, const uint32_t *weights,
                         const uint8_t *exclude, bool exclude_resident, const uint64_t *item_key) {
    int rc;
    ctx->weighted = ctx->have_weights = weights != nullptr;
    if (weights) {
        if ((rc = ensure(ctx, ctx->d_weights, ((size_t)n_items + 1) * sizeof(uint32_t)))) return rc;
        PNX_HIP(ctx, hipMemcpyAsync(ctx->d_weights.p, weights, ((size_t)n_items + 1) * sizeof(uint32_t),
                                    hipMemcpyHostToDevice, ctx->stream));
    }
    ctx->have_exclude = exclude != nullptr || exclude_resident;
    if (exclude) {
        if ((rc = ensure(ctx, ctx->d_exclude, (size_t)n_items + 1))) return rc;
        PNX_HIP(ctx, hipMemcpyAsync(ctx->d_exclude.p, exclude, (size_t)n_items + 1, hipMemcpyHostToDevice, ctx->stream));
    }
    ctx->n_items = n_items;
    ctx->n_paths = n_paths;
    ctx->n_steps = S;
    set_geometry(ctx);

    // every step id must be a valid item (the reference panics on unknown nodes, util.rs:1021).  Over path rows the
    // check rides on the one read of the steps that builds the rows; a renumbering needs it before it starts.
    if ((rc = ensure(ctx, ctx->d_flags, 8 * sizeof(uint32_t)))) return rc;
    // a shape the one-shot route may take (kernels_band.hip) derives no rows at upload: its first sweep reads the steps once
    const bool defer_rows = use_rows(ctx) && !item_key && ctx->cover_route != 2 && (ctx->cover_route == 1 || band_route_fits(ctx, n_paths));
    ctx->h_chunk_sum.clear();
    ctx->h_cuts.clear();
    ctx->h_cut_off.clear();
    ctx->h_jumbled.clear();
    ctx->h_sorted_at.clear();
    ctx->n_sorted_copies = 0;
    ctx->entries_valid = false;
    if (item_key || !use_rows(ctx) || defer_rows) {
        PNX_HIP(ctx, hipMemsetAsync(ctx->d_flags.p, 0, 8 * sizeof(uint32_t), ctx->stream));
        // (a graph the one-shot route may take: the same read says where its paths turn round or jump back)
        const bool summarise = defer_rows && !item_key;
        if ((rc = summarise ? launch_chunk_summaries(ctx, (uint32_t *)ctx->d_flags.p) : launch_validate_items(ctx, (uint32_t *)ctx->d_flags.p))) return rc;
        uint32_t bad = 0;
        PNX_HIP(ctx, hipMemcpyAsync(&bad, ctx->d_flags.p, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
        PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (bad) return ctx->fail(PNX_EINVAL, "items contains ids outside 1..n_items");
        if (summarise) {
            path_cuts_from_chunks(ctx);
            if ((rc = refine_path_cuts(ctx))) return rc;
            if ((rc = sort_jumbled_paths(ctx))) return rc;
            std::vector<pnx::ChunkSummary>().swap(ctx->h_chunk_sum);  // (28 bytes per 4096 steps: not kept beyond the cuts)
            if (getenv("PNX_BAND_DEBUG"))
                for (uint32_t p = 0; p < n_paths; ++p) {
                    fprintf(stderr, "[panacus_amd] path %u (%llu steps): cuts at", p, (unsigned long long)(ctx->h_path_off[p + 1] - ctx->h_path_off[p]));
                    for (uint32_t c = ctx->h_cut_off[p]; c < ctx->h_cut_off[p + 1]; ++c) fprintf(stderr, " %llu", (unsigned long long)(ctx->h_cuts[c] - ctx->h_path_off[p]));
                    fprintf(stderr, "\n");
                }
        }
    }
    ctx->relabeled = false;
    if (item_key && (rc = relabel_by_keys(ctx, item_key))) return rc;
    if (use_rows(ctx) && !defer_rows && (rc = ensure_rows(ctx, true))) return rc;
    ctx->have_csr = true;
    return PNX_OK;
}

static int set_csr_impl(pnx_ctx *ctx, const uint32_t *items, const uint64_t *path_off, uint32_t n_paths,
                        uint32_t n_items, const uint32_t *weights, const uint8_t *exclude, const uint64_t *item_key) {
    if (!ctx) return PNX_EINVAL;
    if (!path_off) return ctx->fail(PNX_EINVAL, "path_off is NULL");
    if (n_items >= 0xFFFFFFFEu || n_paths >= 0xFFFFFFFEu)
        return ctx->fail(PNX_ELIMIT, "n_items and n_paths must be < 2^32-2");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    if (path_off[0] != 0) return ctx->fail(PNX_EINVAL, "path_off[0] must be 0");
    for (uint32_t p = 0; p < n_paths; ++p)
        if (path_off[p + 1] < path_off[p]) return ctx->fail(PNX_EINVAL, "path_off is not non-decreasing at path %u", p);
    const uint64_t S = path_off[n_paths];
    if (S && !items) return ctx->fail(PNX_EINVAL, "items is NULL");
    begin_upload(ctx);
    drop_gfa_text(ctx);
    int rc;
    if ((rc = ensure(ctx, ctx->d_items, S * sizeof(uint32_t) + 64))) return rc;
    if ((rc = ensure(ctx, ctx->d_path_off, ((size_t)n_paths + 1) * sizeof(uint64_t)))) return rc;
    if (S) PNX_HIP(ctx, hipMemcpyAsync(ctx->d_items.p, items, S * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    PNX_HIP(ctx, hipMemcpyAsync(ctx->d_path_off.p, path_off, ((size_t)n_paths + 1) * sizeof(uint64_t),
                                hipMemcpyHostToDevice, ctx->stream));
    ctx->h_path_off.assign(path_off, path_off + n_paths + 1);
    return finish_upload(ctx, S, n_paths, n_items, weights, exclude, false, item_key);
}

// subset / exclude intervals cut on the device (kernels_cut.hip), then the upload is finished as usual
int pnx_set_csr_cut(pnx_ctx *ctx, const pnx_walks *w, const uint32_t *weights, const uint64_t *item_key,
                    pnx_piece_event *events, uint64_t cap, uint64_t *n_events) {
    if (!ctx) return PNX_EINVAL;
    if (n_events) *n_events = 0;
    if (!w || !w->walk_off || !w->path_start || !w->path_mode || !w->node_len || !w->inc_off)
        return ctx->fail(PNX_EINVAL, "pnx_set_csr_cut: NULL argument");
    if (w->count_type < 0 || w->count_type > 2) return ctx->fail(PNX_EINVAL, "pnx_set_csr_cut: count_type must be 0, 1 or 2");
    if (w->n_items >= 0xFFFFFFFEu || w->n_paths >= 0xFFFFFFFEu || w->n_nodes >= 0xFFFFFFFEu)
        return ctx->fail(PNX_ELIMIT, "n_items, n_nodes and n_paths must be < 2^32-2");
    const bool edge_lookup = w->count_type == 2 && !w->edge_item && w->edge_uv && w->edge_oo;
    if (w->count_type == 2 ? (!edge_lookup && (!w->edge_off || (w->edge_off[w->n_paths] && !w->edge_item))) : w->n_items != w->n_nodes)
        return ctx->fail(PNX_EINVAL, w->count_type == 2 ? "pnx_set_csr_cut: edge counts need the edge ItemTable"
                                                          : "pnx_set_csr_cut: n_items must equal n_nodes for node / bp counts");
    if (cap && !events) return ctx->fail(PNX_EINVAL, "pnx_set_csr_cut: events is NULL");
    if (w->walk_off[0] != 0 || w->inc_off[0] != 0 || (w->exc_off && w->exc_off[0] != 0))
        return ctx->fail(PNX_EINVAL, "pnx_set_csr_cut: offsets must start at 0");
    for (uint32_t p = 0; p < w->n_paths; ++p) {
        if (w->walk_off[p + 1] < w->walk_off[p] || w->inc_off[p + 1] < w->inc_off[p] || (w->exc_off && w->exc_off[p + 1] < w->exc_off[p]))
            return ctx->fail(PNX_EINVAL, "pnx_set_csr_cut: offsets are not non-decreasing at path %u", p);
        if (w->path_mode[p] > PNX_WALK_CUT) return ctx->fail(PNX_EINVAL, "pnx_set_csr_cut: path_mode[%u] = %u", p, w->path_mode[p]);
        const uint64_t len = w->walk_off[p + 1] - w->walk_off[p];
        if (w->count_type == 2 && !edge_lookup && w->edge_off[p + 1] - w->edge_off[p] != (len ? len - 1 : 0))
            return ctx->fail(PNX_EINVAL, "pnx_set_csr_cut: path %u has %llu steps but %llu edges", p, (unsigned long long)len,
                             (unsigned long long)(w->edge_off[p + 1] - w->edge_off[p]));
        for (int which = 0; which < 2; ++which) {  // sorted by start, each starts beyond its predecessor's end
            const uint64_t *off = which ? w->exc_off : w->inc_off, *iv = which ? w->exc_iv : w->inc_iv;
            if (!off) continue;
            for (uint64_t i = off[p]; i < off[p + 1]; ++i)
                if (i > off[p] && (iv[2 * i] < iv[2 * i - 2] || iv[2 * i] <= iv[2 * i - 1]))
                    return ctx->fail(PNX_EINVAL, "pnx_set_csr_cut: the %s intervals of path %u are not sorted and disjoint",
                                     which ? "exclude" : "include", p);
        }
    }
    if (w->walk_off[w->n_paths] && !w->walk_node &&
        !(ctx->walks_valid && ctx->h_walk_off.size() == (size_t)w->n_paths + 1 &&
          std::memcmp(ctx->h_walk_off.data(), w->walk_off, ctx->h_walk_off.size() * sizeof(uint64_t)) == 0))
        return ctx->fail(PNX_EINVAL, "pnx_set_csr_cut: walk_node is NULL and the context holds no walks with these offsets (pnx_gfa_walks)");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    begin_upload(ctx);
    drop_gfa_text(ctx);  // (walks tokenised on the device: pnx_gfa_walks has released it already)
    int rc = pnx::cut_walks(ctx, w, events, cap, n_events);
    if (rc) return rc;
    return finish_upload(ctx, ctx->n_steps, w->n_paths, w->n_items, weights, nullptr, w->exc_off != nullptr,
                         item_key ? item_key : (edge_lookup ? w->edge_uv : nullptr));
}

int pnx_gfa_text_upload(pnx_ctx *ctx, const char *text, uint64_t text_bytes) {
    if (!ctx) return PNX_EINVAL;
    if (!text && text_bytes) return ctx->fail(PNX_EINVAL, "pnx_gfa_text_upload: text is NULL");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    return gfa_text_upload(ctx, text, text_bytes);
}

// the arguments of pnx_set_csr_gfa / pnx_gfa_walks that say how segments and edges are named
static int check_gfa_naming(pnx_ctx *ctx, const pnx_gfa_steps *g, const char *who) {
    if (g->name_off && (!g->name_len || g->id_of_name)) return ctx->fail(PNX_EINVAL, "%s: name_off comes with name_len and without id_of_name", who);
    if (g->name_prefix_len > 8) return ctx->fail(PNX_EINVAL, "%s: name_prefix_len is at most 8", who);
    if (names_found_on_device(g) && g->name_hi && (g->name_lo > g->name_hi || g->name_hi > (g->text ? g->text_bytes : ctx->gfa_text_bytes)))
        return ctx->fail(PNX_EINVAL, "%s: name_lo .. name_hi is not a range of the text", who);
    if (g->link_off && (g->edge_uv || g->edge_oo)) return ctx->fail(PNX_EINVAL, "%s: link_off (the L lines parsed on the device) and edge_uv / edge_oo are two ways to hand over the edges: pass one", who);
    const uint64_t bytes = g->text ? g->text_bytes : ctx->gfa_text_bytes;
    const bool find_links = !g->link_off && g->n_links == PNX_LINKS_FIND;
    if (find_links && (g->edge_uv || g->edge_oo)) return ctx->fail(PNX_EINVAL, "%s: PNX_LINKS_FIND and edge_uv / edge_oo are two ways to hand over the edges: pass one", who);
    if (g->n_links && !g->link_off && !find_links) return ctx->fail(PNX_EINVAL, "%s: n_links without link_off", who);
    if (find_links && g->link_hi && (g->link_lo > g->link_hi || g->link_hi > bytes)) return ctx->fail(PNX_EINVAL, "%s: link_lo .. link_hi is not a range of the text", who);
    for (uint64_t k = 0; g->link_off && k < g->n_links; ++k)
        if (g->link_off[k] >= bytes) return ctx->fail(PNX_EINVAL, "%s: L line %llu lies outside the text", who, (unsigned long long)k);
    for (uint32_t i = 0; g->name_off && i < g->n_nodes; ++i)
        if (g->name_off[i] + g->name_len[i] > bytes) return ctx->fail(PNX_EINVAL, "%s: the name of segment %u lies outside the text", who, i + 1);
    return PNX_OK;
}

int pnx_set_csr_gfa(pnx_ctx *ctx, const pnx_gfa_steps *g, const uint32_t *weights, const uint8_t *exclude) {
    if (!ctx) return PNX_EINVAL;
    if (!g || (g->n_paths && (!g->col_begin || !g->col_end || !g->is_walk)))
        return ctx->fail(PNX_EINVAL, "pnx_set_csr_gfa: NULL argument");
    if (g->n_nodes >= 0xFFFFFFFEu || g->n_paths >= 0xFFFFFFFEu) return ctx->fail(PNX_ELIMIT, "n_nodes and n_paths must be < 2^32-2");
    const uint64_t bytes = g->text ? g->text_bytes : ctx->gfa_text_bytes;
    if (!g->text && !ctx->d_gfa_text.p) return ctx->fail(PNX_EINVAL, "pnx_set_csr_gfa: no text (pass it, or call pnx_gfa_text_upload first)");
    for (uint32_t p = 0; p < g->n_paths; ++p)
        if (g->col_begin[p] > g->col_end[p] || g->col_end[p] > bytes)
            return ctx->fail(PNX_EINVAL, "pnx_set_csr_gfa: the step column of path %u lies outside the text", p);
    int rc;
    if ((rc = check_gfa_naming(ctx, g, "pnx_set_csr_gfa"))) return rc;
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    begin_upload(ctx);
    if (g->text && !(ctx->d_gfa_text.p && ctx->gfa_text_host == g->text && ctx->gfa_text_bytes == g->text_bytes) &&
        (rc = gfa_text_upload(ctx, g->text, g->text_bytes)))
        return rc;
    ctx->walks_valid = false;
    release(ctx->d_walk_node);
    release(ctx->d_walk_back);
    const bool links = g->link_off != nullptr || g->n_links == PNX_LINKS_FIND, edges = g->edge_uv != nullptr || links;
    if (edges && ((!links && !g->edge_oo) || weights)) return ctx->fail(PNX_EINVAL, "pnx_set_csr_gfa: edge counts take edge_uv AND edge_oo (or link_off), and no weights");
    DevBuf d_backward, d_e_uv, d_e_oo;
    uint32_t n_edges = g->n_edges;
    rc = gfa_tokenise(ctx, g, edges ? &d_backward : nullptr);
    if (rc == PNX_OK && links) rc = gfa_links_to_edges(ctx, g, d_e_uv, d_e_oo, n_edges);  // (while the text is still in HBM)
    drop_gfa_text(ctx);
    release(ctx->d_name_tab);
    if (rc == PNX_OK && edges)
        rc = links ? gfa_edge_items(ctx, g->n_paths, g->n_nodes, d_backward, (const uint64_t *)d_e_uv.p, (const uint8_t *)d_e_oo.p, n_edges, true)
                   : gfa_edge_items(ctx, g->n_paths, g->n_nodes, d_backward, g->edge_uv, g->edge_oo, n_edges);
    release(d_backward);
    release(d_e_uv);
    release(d_e_oo);
    if (rc) return rc;
    return finish_upload(ctx, ctx->n_steps, g->n_paths, edges ? n_edges : g->n_nodes, weights, exclude, false, nullptr);
}

// a caller's (possibly older, shorter) pnx_gfa_steps as this library's: the fields it does not have are zero / NULL
static bool widen_gfa_steps(pnx_gfa_steps *full, const void *steps, size_t steps_bytes) {
    std::memset(full, 0, sizeof *full);
    if (!steps || steps_bytes < offsetof(pnx_gfa_steps, id_of_name)) return false;  // (text .. is_walk: what every version has had)
    std::memcpy(full, steps, std::min(steps_bytes, sizeof *full));
    return true;
}

int pnx_set_csr_gfa_sized(pnx_ctx *ctx, const void *steps, size_t steps_bytes, const uint32_t *weights, const uint8_t *exclude) {
    if (!ctx) return PNX_EINVAL;
    pnx_gfa_steps full;
    if (!widen_gfa_steps(&full, steps, steps_bytes)) return ctx->fail(PNX_EINVAL, "pnx_set_csr_gfa_sized: steps is NULL or shorter than any pnx_gfa_steps has been");
    return pnx_set_csr_gfa(ctx, &full, weights, exclude);
}

int pnx_gfa_walks_sized(pnx_ctx *ctx, const void *steps, size_t steps_bytes, uint64_t *walk_off) {
    if (!ctx) return PNX_EINVAL;
    pnx_gfa_steps full;
    if (!widen_gfa_steps(&full, steps, steps_bytes)) return ctx->fail(PNX_EINVAL, "pnx_gfa_walks_sized: steps is NULL or shorter than any pnx_gfa_steps has been");
    return pnx_gfa_walks(ctx, &full, walk_off);
}

int pnx_gfa_walks(pnx_ctx *ctx, const pnx_gfa_steps *g, uint64_t *walk_off) {
    if (!ctx) return PNX_EINVAL;
    if (!g || !walk_off || (g->n_paths && (!g->col_begin || !g->col_end || !g->is_walk)))
        return ctx->fail(PNX_EINVAL, "pnx_gfa_walks: NULL argument");
    if (g->n_nodes >= 0xFFFFFFFEu || g->n_paths >= 0xFFFFFFFEu) return ctx->fail(PNX_ELIMIT, "n_nodes and n_paths must be < 2^32-2");
    const uint64_t bytes = g->text ? g->text_bytes : ctx->gfa_text_bytes;
    if (!g->text && !ctx->d_gfa_text.p) return ctx->fail(PNX_EINVAL, "pnx_gfa_walks: no text (pass it, or call pnx_gfa_text_upload first)");
    for (uint32_t p = 0; p < g->n_paths; ++p)
        if (g->col_begin[p] > g->col_end[p] || g->col_end[p] > bytes)
            return ctx->fail(PNX_EINVAL, "pnx_gfa_walks: the step column of path %u lies outside the text", p);
    int rc;
    if ((rc = check_gfa_naming(ctx, g, "pnx_gfa_walks"))) return rc;
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    begin_upload(ctx);  // (the tokeniser writes through the resident graph's buffers)
    ctx->walks_valid = false;
    ctx->n_link_edges = 0;
    release(ctx->d_link_uv);
    release(ctx->d_link_oo);
    if (g->text && !(ctx->d_gfa_text.p && ctx->gfa_text_host == g->text && ctx->gfa_text_bytes == g->text_bytes) &&
        (rc = gfa_text_upload(ctx, g->text, g->text_bytes)))
        return rc;
    rc = gfa_tokenise(ctx, g, &ctx->d_walk_back);
    // the L lines, if handed over: parsed now, while the text is in HBM, and kept beside the walks (pnx_set_csr_walks)
    const bool with_links = g->link_off != nullptr || g->n_links == PNX_LINKS_FIND;
    if (rc == PNX_OK && with_links) rc = gfa_links_to_edges(ctx, g, ctx->d_link_uv, ctx->d_link_oo, ctx->n_link_edges);
    ctx->links_valid = rc == PNX_OK && with_links;
    drop_gfa_text(ctx);
    release(ctx->d_name_tab);
    if (rc) return rc;
    std::swap(ctx->d_items, ctx->d_walk_node);
    ctx->h_walk_off = ctx->h_path_off;
    std::memcpy(walk_off, ctx->h_walk_off.data(), ((size_t)g->n_paths + 1) * sizeof(uint64_t));
    ctx->walks_valid = true;
    return PNX_OK;
}

int pnx_set_csr_walks(pnx_ctx *ctx, uint32_t n_nodes, const uint32_t *weights, const uint8_t *exclude, const uint64_t *edge_uv,
                      const uint8_t *edge_oo, uint32_t n_edges) {
    if (!ctx) return PNX_EINVAL;
    if (!ctx->walks_valid || ctx->h_walk_off.empty()) return ctx->fail(PNX_EINVAL, "pnx_set_csr_walks: the context holds no walks (pnx_gfa_walks)");
    const bool from_links = edge_uv == nullptr && n_edges == PNX_EDGES_FROM_LINKS;  // the L lines pnx_gfa_walks parsed
    if (from_links && !ctx->links_valid) return ctx->fail(PNX_EINVAL, "pnx_set_csr_walks: pnx_gfa_walks was not given the L lines (link_off)");
    if (from_links) n_edges = ctx->n_link_edges;
    const bool edges = edge_uv != nullptr || from_links;
    if (edges && ((!from_links && !edge_oo) || weights)) return ctx->fail(PNX_EINVAL, "pnx_set_csr_walks: edge counts take edge_uv AND edge_oo, and no weights");
    if (n_nodes >= 0xFFFFFFFEu || (edges && n_edges >= 0xFFFFFFFEu)) return ctx->fail(PNX_ELIMIT, "n_nodes and n_edges must be < 2^32-2");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    begin_upload(ctx);
    const uint32_t n_paths = (uint32_t)(ctx->h_walk_off.size() - 1);
    const uint64_t S = ctx->h_walk_off[n_paths];
    int rc;
    if ((rc = ensure(ctx, ctx->d_items, S * sizeof(uint32_t) + 64)) || (rc = ensure(ctx, ctx->d_path_off, ((size_t)n_paths + 1) * sizeof(uint64_t))))
        return rc;
    if (S) PNX_HIP(ctx, hipMemcpyAsync(ctx->d_items.p, ctx->d_walk_node.p, S * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    ctx->h_path_off = ctx->h_walk_off;
    PNX_HIP(ctx, hipMemcpyAsync(ctx->d_path_off.p, ctx->h_path_off.data(), ((size_t)n_paths + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    ctx->n_steps = S;
    if (edges && (rc = from_links ? gfa_edge_items(ctx, n_paths, n_nodes, ctx->d_walk_back, (const uint64_t *)ctx->d_link_uv.p, (const uint8_t *)ctx->d_link_oo.p, n_edges, true)
                                  : gfa_edge_items(ctx, n_paths, n_nodes, ctx->d_walk_back, edge_uv, edge_oo, n_edges)))
        return rc;
    return finish_upload(ctx, ctx->n_steps, n_paths, edges ? n_edges : n_nodes, weights, exclude, false, nullptr);
}

int pnx_set_weights(pnx_ctx *ctx, const uint32_t *weights) {
    if (!ctx) return PNX_EINVAL;
    if (!ctx->have_csr) return ctx->fail(PNX_EINVAL, "pnx_set_weights before a graph is resident");
    if (!weights) return ctx->fail(PNX_EINVAL, "pnx_set_weights: weights is NULL");
    if (ctx->d_weights.borrowed) return ctx->fail(PNX_EINVAL, "pnx_set_weights: this context borrows its graph (pnx_share_csr)");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    invalidate_results(ctx);
    const size_t bytes = ((size_t)ctx->n_items + 1) * sizeof(uint32_t);
    int rc = ensure(ctx, ctx->d_weights, bytes);
    if (rc) return rc;
    if (ctx->relabeled) {
        DevBuf tmp;
        if ((rc = ensure(ctx, tmp, bytes))) return rc;
        hipError_t e = hipMemcpyAsync(tmp.p, weights, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) rc = to_internal_ids_u32(ctx, (const uint32_t *)tmp.p, (uint32_t *)ctx->d_weights.p);
        const hipError_t e2 = hipStreamSynchronize(ctx->stream);
        release(tmp);
        if (e != hipSuccess || e2 != hipSuccess) return ctx->fail(PNX_EHIP, "pnx_set_weights: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        if (rc) return rc;
    } else {
        PNX_HIP(ctx, hipMemcpyAsync(ctx->d_weights.p, weights, bytes, hipMemcpyHostToDevice, ctx->stream));
        PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->weighted = ctx->have_weights = true;
    ctx->wplanes_valid = false;  // bit planes (K4) and 7-bit digits (K5) are derived from the weights
    ctx->wdigits_valid = false;
    return PNX_OK;
}

int pnx_exclude_items(pnx_ctx *ctx, const uint32_t *ids, uint32_t n) {
    if (!ctx) return PNX_EINVAL;
    if (!ctx->have_csr) return ctx->fail(PNX_EINVAL, "pnx_exclude_items before a graph is resident");
    if (n && !ids) return ctx->fail(PNX_EINVAL, "pnx_exclude_items: ids is NULL");
    if (ctx->d_exclude.borrowed) return ctx->fail(PNX_EINVAL, "pnx_exclude_items: this context borrows its graph (pnx_share_csr)");
    if (!n) return PNX_OK;
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    invalidate_results(ctx);
    if (!ctx->have_exclude) {
        int rc = ensure(ctx, ctx->d_exclude, (size_t)ctx->n_items + 1);
        if (rc) return rc;
        PNX_HIP(ctx, hipMemsetAsync(ctx->d_exclude.p, 0, (size_t)ctx->n_items + 1, ctx->stream));
        ctx->have_exclude = true;
    }
    return pnx::flag_items(ctx, ids, n);
}

int pnx_set_csr(pnx_ctx *ctx, const uint32_t *items, const uint64_t *path_off, uint32_t n_paths,
                uint32_t n_items, const uint32_t *weights, const uint8_t *exclude) {
    return set_csr_impl(ctx, items, path_off, n_paths, n_items, weights, exclude, nullptr);
}

int pnx_set_csr_keyed(pnx_ctx *ctx, const uint32_t *items, const uint64_t *path_off, uint32_t n_paths,
                      uint32_t n_items, const uint32_t *weights, const uint8_t *exclude, const uint64_t *item_key) {
    return set_csr_impl(ctx, items, path_off, n_paths, n_items, weights, exclude, item_key);
}

int pnx_set_exclude(pnx_ctx *ctx, const uint8_t *exclude) {
    if (!ctx) return PNX_EINVAL;
    if (!ctx->have_csr) return ctx->fail(PNX_EINVAL, "pnx_set_exclude before a graph is resident");
    if (ctx->d_exclude.borrowed) return ctx->fail(PNX_EINVAL, "pnx_set_exclude: this context borrows its graph (pnx_share_csr)");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    invalidate_results(ctx);  // coverage, histogram and presence matrix all change
    if (exclude) {
        int rc = ensure(ctx, ctx->d_exclude, (size_t)ctx->n_items + 1);
        if (rc) return rc;
        if (ctx->relabeled) {  // the resident flags live in the internal numbering
            DevBuf tmp;
            if ((rc = ensure(ctx, tmp, (size_t)ctx->n_items + 1))) return rc;
            hipError_t e = hipMemcpyAsync(tmp.p, exclude, (size_t)ctx->n_items + 1, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) rc = to_internal_ids_u8(ctx, (const uint8_t *)tmp.p, (uint8_t *)ctx->d_exclude.p);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            release(tmp);
            if (e != hipSuccess) return ctx->fail(PNX_EHIP, "pnx_set_exclude: %s", hipGetErrorString(e));
            if (rc) return rc;
        } else {
            PNX_HIP(ctx, hipMemcpyAsync(ctx->d_exclude.p, exclude, (size_t)ctx->n_items + 1, hipMemcpyHostToDevice, ctx->stream));
            PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the flags are caller-owned
        }
    }
    ctx->have_exclude = exclude != nullptr;
    return PNX_OK;
}

int pnx_get_exclude(pnx_ctx *ctx, uint8_t *exclude) {
    if (!ctx) return PNX_EINVAL;
    if (!ctx->have_csr) return ctx->fail(PNX_EINVAL, "no graph is resident");
    if (!exclude) return ctx->fail(PNX_EINVAL, "pnx_get_exclude: exclude is NULL");
    const size_t n = (size_t)ctx->n_items + 1;
    if (!ctx->have_exclude) {
        std::fill(exclude, exclude + n, (uint8_t)0);
        return PNX_OK;
    }
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->relabeled) {
        PNX_HIP(ctx, hipMemcpyAsync(exclude, ctx->d_exclude.p, n, hipMemcpyDeviceToHost, ctx->stream));
        PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return PNX_OK;
    }
    DevBuf tmp;  // internal -> caller ids: caller[i] = internal[new_of_old[i]]
    int rc = ensure(ctx, tmp, n);
    if (rc) return rc;
    rc = to_caller_ids_u8(ctx, (const uint8_t *)ctx->d_exclude.p, (uint8_t *)tmp.p);
    hipError_t e = rc ? hipSuccess : hipMemcpyAsync(exclude, tmp.p, n, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    release(tmp);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return ctx->fail(PNX_EHIP, "pnx_get_exclude: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return PNX_OK;
}

int pnx_set_csr_pansyn(pnx_ctx *ctx, uint64_t seed, uint32_t n_nodes, uint32_t n_paths, int with_weights) {
    return pnx_set_csr_pansyn_shard(ctx, seed, 0, n_nodes, n_paths, with_weights);
}

int pnx_set_csr_pansyn_shard(pnx_ctx *ctx, uint64_t seed, uint64_t node_lo, uint32_t n_nodes, uint32_t n_paths, int with_weights) {
    if (!ctx) return PNX_EINVAL;
    if (n_nodes == 0 || n_paths == 0) return ctx->fail(PNX_EINVAL, "n_nodes and n_paths must be > 0");
    if (n_nodes >= 0xFFFFFFFEu) return ctx->fail(PNX_ELIMIT, "n_nodes must be < 2^32-2");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    begin_upload(ctx);
    drop_gfa_text(ctx);
    int rc = pansyn_generate_device(ctx, seed, n_nodes, n_paths, with_weights, node_lo);
    if (rc) return rc;
    ctx->have_exclude = false;
    ctx->relabeled = false;
    set_geometry(ctx);
    ctx->have_csr = true;
    return PNX_OK;
}

int pnx_set_csr_pansyn_rearranged(pnx_ctx *ctx, uint64_t seed, uint32_t n_nodes, uint32_t n_paths, int with_weights) {
    if (!ctx) return PNX_EINVAL;
    if (n_nodes == 0 || n_paths == 0) return ctx->fail(PNX_EINVAL, "n_nodes and n_paths must be > 0");
    if (n_nodes >= 0xFFFFFFFEu) return ctx->fail(PNX_ELIMIT, "n_nodes must be < 2^32-2");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    begin_upload(ctx);
    drop_gfa_text(ctx);
    int rc = pansyn_generate_device(ctx, seed, n_nodes, n_paths, with_weights, 0);
    if (rc == PNX_OK) rc = pansyn_rearrange_device(ctx, seed);
    if (rc) return rc;
    ctx->have_exclude = false;
    ctx->relabeled = false;
    set_geometry(ctx);
    ctx->have_csr = true;
    return PNX_OK;
}

int pnx_share_csr(pnx_ctx *dst, pnx_ctx *src) {
    if (!dst) return PNX_EINVAL;
    if (!src || src == dst) return dst->fail(PNX_EINVAL, "pnx_share_csr: needs another context as the source");
    if (!src->have_csr) return dst->fail(PNX_EINVAL, "pnx_share_csr: the source holds no graph");
    if (src->device != dst->device) return dst->fail(PNX_EINVAL, "pnx_share_csr: the contexts are on different devices");
    if (src->d_items.borrowed) return dst->fail(PNX_EINVAL, "pnx_share_csr: the source itself borrows its graph");
    PNX_HIP(dst, hipSetDevice(dst->device));
    // the packed steps and the path classes are derived data of the graph: made once, by the owner
    if (!use_rows(src) && !step_routes(src)) return dst->fail(PNX_EINVAL, "pnx_share_csr: %s", src->err.c_str());
    if (int prc = use_rows(src) ? ensure_rows(src, false) : step_routes(src)->prepare_steps(src)) return dst->fail(prc, "pnx_share_csr: %s", src->err.c_str());
    PNX_HIP(dst, hipStreamSynchronize(src->stream));  // its upload is complete
    invalidate_results(dst);
    dst->have_csr = false;
    dst->have_order = false;
    auto borrow = [](DevBuf &d, const DevBuf &s) {
        release(d);
        d.p = s.p;
        d.cap = 0;
        d.borrowed = s.p != nullptr;
    };
    borrow(dst->d_items, src->d_items);
    borrow(dst->d_path_off, src->d_path_off);
    borrow(dst->d_weights, src->d_weights);
    borrow(dst->d_exclude, src->d_exclude);
    borrow(dst->d_steps12, src->d_steps12);
    borrow(dst->d_path_mono, src->d_path_mono);
    dst->steps_prepared = src->steps_prepared;
    borrow(dst->d_unsorted, src->d_unsorted);
    borrow(dst->d_sorted_coff, src->d_sorted_coff);
    borrow(dst->d_sorted_path, src->d_sorted_path);
    dst->n_sorted_paths = src->n_sorted_paths;
    dst->n_unsorted = src->n_unsorted;
    borrow(dst->d_new_of_old, src->d_new_of_old);
    borrow(dst->d_old_of_new, src->d_old_of_new);
    dst->relabeled = src->relabeled;
    dst->h_path_off = src->h_path_off;
    dst->h_cuts = src->h_cuts;  // (where the paths turn round or jump back: found at the owner's upload)
    dst->h_cut_off = src->h_cut_off;
    dst->h_sorted_at = src->h_sorted_at;  // (the copies lie in the owner's d_items, behind the steps)
    dst->n_sorted_copies = src->n_sorted_copies;
    dst->entries_valid = false;
    dst->weighted = src->weighted;
    dst->have_weights = src->have_weights;
    dst->have_exclude = src->have_exclude;
    dst->n_items = src->n_items;
    dst->n_paths = src->n_paths;
    dst->n_steps = src->n_steps;
    set_geometry(dst);
    if (src->rows_valid) {  // the path rows are derived data of the graph as well
        borrow(dst->d_rows, src->d_rows);
        borrow(dst->d_row_base, src->d_row_base);
        borrow(dst->d_id_minmax, src->d_id_minmax);
        borrow(dst->d_rt_first, src->d_rt_first);
        borrow(dst->d_rt_span, src->d_rt_span);
        dst->h_id_minmax = src->h_id_minmax;
        dst->h_rt_first = src->h_rt_first;
        dst->h_rt_span = src->h_rt_span;
        dst->row_tstride = src->row_tstride;
        dst->rows_max_span = src->rows_max_span;
        dst->n_rows = src->n_rows;
        dst->rows_tile_major = src->rows_tile_major;
        dst->rows_valid = true;
    }
    dst->have_csr = true;
    return PNX_OK;
}

int pnx_prepare(pnx_ctx *ctx) {
    if (!ctx) return PNX_EINVAL;
    if (!ctx->have_csr) return ctx->fail(PNX_EINVAL, "pnx_prepare before a graph is resident");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    if (!use_rows(ctx) && !step_routes(ctx)) return PNX_EINVAL;
    int rc = use_rows(ctx) ? ensure_rows(ctx, false) : step_routes(ctx)->prepare_steps(ctx);
    if (rc) return rc;
    PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PNX_OK;
}

int pnx_get_csr(pnx_ctx *ctx, uint64_t *n_steps, uint32_t *items, uint64_t *path_off, uint32_t *weights) {
    if (!ctx) return PNX_EINVAL;
    if (!ctx->have_csr) return ctx->fail(PNX_EINVAL, "no graph is resident");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    if (n_steps) *n_steps = ctx->n_steps;
    DevBuf tmp_items, tmp_w;  // a relabelled graph is handed back in the caller's ids
    int rc = PNX_OK;
    hipError_t e = hipSuccess;
    if (items && ctx->n_steps) {
        const void *src = ctx->d_items.p;
        if (ctx->relabeled || ctx->n_sorted_paths) {  // the caller's ids, the caller's order
            if ((rc = ensure(ctx, tmp_items, ctx->n_steps * sizeof(uint32_t) + 64))) return rc;
            e = hipMemcpyAsync(tmp_items.p, ctx->d_items.p, ctx->n_steps * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream);
            if (e == hipSuccess && ctx->n_sorted_paths) rc = step_routes(ctx) ? step_routes(ctx)->restore_step_order(ctx, (uint32_t *)tmp_items.p) : PNX_EINVAL;
            if (e == hipSuccess && !rc && ctx->relabeled) rc = steps_to_caller_ids(ctx, (uint32_t *)tmp_items.p, ctx->n_steps);
            src = tmp_items.p;
        }
        if (e == hipSuccess && !rc) e = hipMemcpyAsync(items, src, ctx->n_steps * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (path_off) std::copy(ctx->h_path_off.begin(), ctx->h_path_off.end(), path_off);
    if (weights && !rc && e == hipSuccess) {
        if (!ctx->weighted) rc = ctx->fail(PNX_EINVAL, "no weights are resident");
        const void *src = ctx->d_weights.p;
        if (!rc && ctx->relabeled) {
            if (!(rc = ensure(ctx, tmp_w, ((size_t)ctx->n_items + 1) * sizeof(uint32_t))))
                rc = to_caller_ids_u32(ctx, (const uint32_t *)ctx->d_weights.p, (uint32_t *)tmp_w.p);
            src = tmp_w.p;
        }
        if (!rc) e = hipMemcpyAsync(weights, src, ((size_t)ctx->n_items + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    }
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    release(tmp_items);
    release(tmp_w);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return ctx->fail(PNX_EHIP, "pnx_get_csr: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return PNX_OK;
}

int pnx_set_order(pnx_ctx *ctx, const uint32_t *path_idx, const uint32_t *group_id, uint32_t n_ordered,
                  uint32_t n_groups) {
    if (!ctx) return PNX_EINVAL;
    if (!ctx->have_csr) return ctx->fail(PNX_EINVAL, "pnx_set_order before pnx_set_csr");
    if (n_ordered && (!path_idx || !group_id)) return ctx->fail(PNX_EINVAL, "NULL order arrays");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t expect = 0;
    for (uint32_t k = 0; k < n_ordered; ++k) {
        if (path_idx[k] >= ctx->n_paths) return ctx->fail(PNX_EINVAL, "path_idx[%u] = %u out of range", k, path_idx[k]);
        if (k == 0 ? group_id[0] != 0 : (group_id[k] != group_id[k - 1] && group_id[k] != group_id[k - 1] + 1))
            return ctx->fail(PNX_EINVAL, "group_id must start at 0 and be non-decreasing and dense (entry %u)", k);
        expect = group_id[k] + 1;
    }
    if (expect != n_groups) return ctx->fail(PNX_EINVAL, "n_groups = %u but the order holds %u groups", n_groups, expect);
    invalidate_results(ctx);
    int rc;
    if ((rc = ensure(ctx, ctx->d_ord_path, (size_t)n_ordered * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->d_ord_group, (size_t)n_ordered * sizeof(uint32_t)))) return rc;
    if (n_ordered) {
        PNX_HIP(ctx, hipMemcpyAsync(ctx->d_ord_path.p, path_idx, (size_t)n_ordered * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        PNX_HIP(ctx, hipMemcpyAsync(ctx->d_ord_group.p, group_id, (size_t)n_ordered * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host arrays are caller-owned
    }
    ctx->order_normalized = false;
    ctx->h_ord_path.assign(path_idx, path_idx + n_ordered);
    ctx->h_ord_group.assign(group_id, group_id + n_ordered);
    ctx->runs_sorted = false;  // run keys carry the group of the path
    ctx->entries_valid = false;
    ctx->n_ordered = n_ordered;
    ctx->n_groups = n_groups;
    ctx->have_order = true;
    // (the entries of a one-shot pass follow from the order and the graph alone: made here, not inside the first pass)
    // (the entries serve the one-shot route alone: not built where no such pass can follow)
    if (use_rows(ctx) && ctx->have_csr && ctx->h_path_off.size() == (size_t)ctx->n_paths + 1 && !ctx->rows_valid && !ctx->band_failed && ctx->cover_route != 2 &&
        (rc = ensure_band_entries(ctx)))
        return rc;
    return PNX_OK;
}

int pnx_hist_async(pnx_ctx *ctx) {
    if (!ctx) return PNX_EINVAL;
    if (!ctx->have_csr || !ctx->have_order) return ctx->fail(PNX_EINVAL, "pnx_hist needs pnx_set_csr and pnx_set_order first");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->tk_count >= ctx->max_in_flight)
        return ctx->fail(PNX_EINVAL, "%d coverage passes are already in flight; fetch one first", ctx->max_in_flight);
    int rc;
    // the presence matrix is only written when asked for (config) or when a growth call needs it
    ctx->want_M = ctx->keep_M_user || ctx->growth_needs_M;
    ctx->growth_needs_M = false;
    ctx->cur = &ctx->tk[ctx->tk_next];
    ctx->pass_band = false;
    if (use_rows(ctx)) {
        // the first sweep of a graph takes the steps themselves when its shape suits the one-shot route (one read, nothing
        // derived); a second sweep is a caller that keeps sweeping: it derives the path rows, and every later pass is 8x cheaper
        ctx->pass_band = !ctx->rows_valid && !ctx->band_failed && ctx->n_ordered && ctx->n_steps &&
                         (ctx->cover_route == 1 || (ctx->cover_route == 0 && ctx->n_band_passes == 0 && band_route_fits(ctx, ctx->entries_valid ? ctx->n_entries : ctx->n_ordered)));  // (the pieces of cut paths are entries too)
        if (ctx->pass_band) ctx->n_band_passes += 1;
        else if ((rc = ensure_rows(ctx, false))) return rc;  // once per upload
    }
    if ((rc = choose_pass_streams(ctx))) return rc;
    if (use_rows(ctx)) {
    } else if (!ctx->index_valid || !ctx->cache_index) {
        // a kept index is shared by the passes: nothing may still be reading it while it is rebuilt
        if (ctx->cache_index && ctx->tk_count && (rc = drain_streams(ctx))) return rc;
        drop_run_index(ctx);  // path classes are reset with the index
        const StepRoutes *sr = step_routes(ctx);
        if (!sr) return PNX_EINVAL;
        if ((rc = sr->launch_tile_index(ctx))) return rc;
        ctx->index_valid = true;
    }
    ctx->hist_valid = false;
    return enqueue_pass(ctx);
}

int pnx_sync(pnx_ctx *ctx) {
    if (!ctx) return PNX_EINVAL;
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = settle_all(ctx);
    if (rc) return rc;
    if ((rc = drain_streams(ctx))) return rc;
    prof_resolve(ctx);
    ctx->growth_pending = false;
    return PNX_OK;
}

int pnx_hist_enqueued(pnx_ctx *ctx, uint64_t **d_hist) {
    if (!ctx || !d_hist) return PNX_EINVAL;
    if (ctx->tk_count == 0) return ctx->fail(PNX_EINVAL, "no coverage pass is in flight");
    Ticket *t = &ctx->tk[ctx->tk_last()];
    *d_hist = t->d_hist;
    // the counters are written on an internal stream: whatever the caller now enqueues on pnx_stream() waits for them
    if (ctx->last_pass_phased) PNX_HIP(ctx, hipStreamWaitEvent(ctx->stream, t->done, 0));
    return PNX_OK;
}

int pnx_hist_device(pnx_ctx *ctx, uint64_t **d_hist, uint32_t **d_countable) {
    if (!ctx) return PNX_EINVAL;
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = settle_oldest(ctx);  // the oldest pass in flight (a younger one keeps running)
    if (rc) return rc;
    if (!ctx->hist_valid || !ctx->last_done) return ctx->fail(PNX_EINVAL, "no histogram has been computed");
    if (d_hist) *d_hist = ctx->last_done->d_hist;
    if (d_countable) {
        *d_countable = (uint32_t *)ctx->last_done->d_countable.p;  // the settled pass's own vector
        if (ctx->relabeled) {  // the caller's ids: a gathered copy
            if ((rc = ensure(ctx, ctx->d_countable_ext, ((size_t)ctx->n_items + 1) * sizeof(uint32_t)))) return rc;
            if ((rc = to_caller_ids_u32(ctx, (const uint32_t *)ctx->last_done->d_countable.p, (uint32_t *)ctx->d_countable_ext.p))) return rc;
            PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            *d_countable = (uint32_t *)ctx->d_countable_ext.p;
        }
    }
    return PNX_OK;
}

int pnx_hist_fetch(pnx_ctx *ctx, uint32_t *countable, uint64_t *hist) {
    if (!ctx) return PNX_EINVAL;
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = settle_oldest(ctx);
    if (rc) return rc;
    if (!ctx->hist_valid || !ctx->last_done) return ctx->fail(PNX_EINVAL, "no histogram has been computed");
    if (hist) std::memcpy(hist, ctx->last_done->h_hist, ((size_t)ctx->n_groups + 1) * sizeof(uint64_t));
    if (countable) {
        // every pass has its own coverage vector; this is the settled pass's
        const void *src = ctx->last_done->d_countable.p;
        if (ctx->relabeled) {
            if ((rc = ensure(ctx, ctx->d_countable_ext, ((size_t)ctx->n_items + 1) * sizeof(uint32_t)))) return rc;
            if ((rc = to_caller_ids_u32(ctx, (const uint32_t *)ctx->last_done->d_countable.p, (uint32_t *)ctx->d_countable_ext.p))) return rc;
            PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            src = ctx->d_countable_ext.p;
        }
        PNX_HIP(ctx, hipMemcpy(countable, src, ((size_t)ctx->n_items + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return PNX_OK;
}

int pnx_hist(pnx_ctx *ctx, uint32_t *countable, uint64_t *hist) {
    int rc = pnx_hist_async(ctx);
    if (rc) return rc;
    if ((rc = settle_all(ctx))) return rc;  // the result of THIS pass, not of an older one in flight
    return pnx_hist_fetch(ctx, countable, hist);
}

void *pnx_stream(pnx_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int pnx_hist_enqueued_on(pnx_ctx *ctx, uint64_t **d_hist, void **stream) {
    if (!ctx || !d_hist || !stream) return PNX_EINVAL;
    if (ctx->tk_count == 0) return ctx->fail(PNX_EINVAL, "no coverage pass is in flight");
    *d_hist = ctx->tk[ctx->tk_last()].d_hist;
    *stream = (void *)(ctx->last_pass_phased ? ctx->stream_post : ctx->stream);
    return PNX_OK;
}

int pnx_ordered_growth_async(pnx_ctx *ctx, const uint32_t *perms, uint32_t n_perms, const uint32_t *cov_thr,
                             const uint32_t *quorum_tab, uint32_t n_thr) {
    if (!ctx) return PNX_EINVAL;
    if (!ctx->have_csr || !ctx->have_order) return ctx->fail(PNX_EINVAL, "pnx_ordered_growth needs pnx_set_csr and pnx_set_order first");
    if (n_perms == 0 || n_thr == 0 || !cov_thr || !quorum_tab) return ctx->fail(PNX_EINVAL, "bad growth arguments");
    if (!perms && n_perms != 1) return ctx->fail(PNX_EINVAL, "perms == NULL (identity) requires n_perms == 1");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t G = ctx->n_groups;
    if (perms) {
        std::vector<uint8_t> seen(G);
        for (uint32_t r = 0; r < n_perms; ++r) {
            std::fill(seen.begin(), seen.end(), 0);
            for (uint32_t j = 0; j < G; ++j) {
                uint32_t g = perms[(size_t)r * G + j];
                if (g >= G || seen[g]) return ctx->fail(PNX_EINVAL, "perms[%u] is not a permutation of 0..%u", r, G ? G - 1 : 0);
                seen[g] = 1;
            }
        }
    }
    int rc;
    // a growth call that is still running owns the staging vectors and the result buffer
    if (ctx->growth_pending) {
        PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->growth_pending = false;
    }
    // the presence matrix must exist for the current order
    if ((rc = settle_all(ctx))) return rc;
    if (!(ctx->hist_valid && ctx->M_valid)) {
        ctx->growth_needs_M = true;
        if ((rc = pnx_hist_async(ctx))) return rc;
        if ((rc = settle_all(ctx))) return rc;
    }
    const size_t RG = (size_t)n_perms * G, TG = (size_t)n_thr * G;
    if ((rc = ensure(ctx, ctx->d_cov_thr, n_thr * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->d_qtab, (TG ? TG : 1) * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->d_growth_out, (RG ? RG : 1) * n_thr * sizeof(uint64_t)))) return rc;
    // host copies: the launch code derives its device tables from these (nothing is read back)
    if (perms) ctx->h_perms.assign(perms, perms + RG);
    else {
        ctx->h_perms.resize(G);
        for (uint32_t j = 0; j < G; ++j) ctx->h_perms[j] = j;
    }
    ctx->h_qtab.assign(quorum_tab, quorum_tab + TG);
    ctx->g_R = n_perms;
    ctx->g_T = n_thr;
    ctx->h_thr_meta.assign(cov_thr, cov_thr + n_thr);
    for (uint32_t t = 0; t < n_thr; ++t) {
        bool q0 = true;
        for (uint32_t j = 0; j < G && q0; ++j) q0 = quorum_tab[(size_t)t * G + j] <= 1;
        ctx->h_thr_meta.push_back(q0 ? 1u : 0u);
    }
    if ((rc = launch_growth(ctx, perms == nullptr))) return rc;
    ctx->growth_pending = true;
    return PNX_OK;
}

int pnx_ordered_growth_enqueued(pnx_ctx *ctx, uint64_t **d_out) {
    if (!ctx || !d_out) return PNX_EINVAL;
    if (!ctx->g_R || !ctx->growth_pending) return ctx->fail(PNX_EINVAL, "no growth call is in flight");
    *d_out = (uint64_t *)ctx->d_growth_out.p;
    return PNX_OK;
}

int pnx_ordered_growth_device(pnx_ctx *ctx, uint64_t **d_out) {
    if (!ctx) return PNX_EINVAL;
    int rc = pnx_sync(ctx);
    if (rc) return rc;
    if (!ctx->g_R) return ctx->fail(PNX_EINVAL, "no growth has been computed");
    if (d_out) *d_out = (uint64_t *)ctx->d_growth_out.p;
    return PNX_OK;
}

int pnx_ordered_growth_fetch(pnx_ctx *ctx, uint64_t *out) {
    if (!ctx) return PNX_EINVAL;
    int rc = pnx_sync(ctx);
    if (rc) return rc;
    if (!ctx->g_R) return ctx->fail(PNX_EINVAL, "no growth has been computed");
    const size_t n = (size_t)ctx->g_R * ctx->g_T * ctx->n_groups;
    if (out && n) {
        PNX_HIP(ctx, hipMemcpyAsync(out, ctx->d_growth_out.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PNX_OK;
}

int pnx_ordered_growth(pnx_ctx *ctx, const uint32_t *perms, uint32_t n_perms, const uint32_t *cov_thr,
                       const uint32_t *quorum_tab, uint32_t n_thr, uint64_t *out) {
    int rc = pnx_ordered_growth_async(ctx, perms, n_perms, cov_thr, quorum_tab, n_thr);
    if (rc) return rc;
    return pnx_ordered_growth_fetch(ctx, out);
}

// the presence matrix of the current order, resident and verified
static int ensure_presence(pnx_ctx *ctx, const char *who) {
    if (!ctx->have_csr || !ctx->have_order) return ctx->fail(PNX_EINVAL, "%s needs pnx_set_csr and pnx_set_order first", who);
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = settle_all(ctx))) return rc;
    if (!(ctx->hist_valid && ctx->M_valid)) {
        ctx->growth_needs_M = true;
        if ((rc = pnx_hist_async(ctx))) return rc;
        if ((rc = settle_all(ctx))) return rc;
    }
    return PNX_OK;
}

int pnx_group_intersections_device(pnx_ctx *ctx, uint64_t **d_inter) {
    if (!ctx) return PNX_EINVAL;
    int rc = ensure_presence(ctx, "pnx_group_intersections");
    if (rc) return rc;
    if ((rc = launch_pair_intersections(ctx))) return rc;
    PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (d_inter) *d_inter = (uint64_t *)ctx->d_inter.p;
    return PNX_OK;
}

int pnx_group_intersections(pnx_ctx *ctx, uint64_t *inter) {
    if (!ctx) return PNX_EINVAL;
    uint64_t *d = nullptr;
    int rc = pnx_group_intersections_device(ctx, &d);
    if (rc) return rc;
    const size_t n = (size_t)ctx->n_groups * ctx->n_groups;
    if (inter && n) {
        PNX_HIP(ctx, hipMemcpyAsync(inter, d, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PNX_OK;
}

uint64_t pnx_presence_row_words(pnx_ctx *ctx) { return ctx ? (uint64_t)ctx->n_blocks * 32u : 0; }

int pnx_presence(pnx_ctx *ctx, uint64_t *bits) {
    if (!ctx) return PNX_EINVAL;
    if (!bits) return ctx->fail(PNX_EINVAL, "pnx_presence: bits is NULL");
    int rc = ensure_presence(ctx, "pnx_presence");
    if (rc) return rc;
    if ((rc = launch_presence_plain(ctx, ctx->d_plain))) return rc;
    const size_t n = (size_t)ctx->n_groups * ctx->n_blocks * 32;
    DevBuf d_perm;
    const void *src = ctx->d_plain.p;
    if (ctx->relabeled && n) {
        if ((rc = presence_to_caller_ids(ctx, ctx->d_plain, d_perm))) return rc;
        src = d_perm.p;
    }
    hipError_t e = n ? hipMemcpyAsync(bits, src, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    release(ctx->d_plain);
    release(d_perm);
    if (e != hipSuccess || e2 != hipSuccess) return ctx->fail(PNX_EHIP, "pnx_presence: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return PNX_OK;
}

int pnx_group_visit_counts(pnx_ctx *ctx, uint32_t item_lo, uint32_t item_hi, uint32_t *out) {
    if (!ctx) return PNX_EINVAL;
    if (!ctx->have_csr || !ctx->have_order) return ctx->fail(PNX_EINVAL, "pnx_group_visit_counts needs pnx_set_csr and pnx_set_order first");
    if (!out) return ctx->fail(PNX_EINVAL, "pnx_group_visit_counts: out is NULL");
    if (item_lo > item_hi || (uint64_t)item_hi > (uint64_t)ctx->n_items + 1)
        return ctx->fail(PNX_EINVAL, "pnx_group_visit_counts: item range [%u, %u) outside 0..%u", item_lo, item_hi, ctx->n_items + 1);
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = settle_all(ctx);
    if (rc) return rc;
    // group of every path (0xFFFFFFFF: not in the visiting order)
    std::vector<uint32_t> pg(ctx->n_paths ? ctx->n_paths : 1, 0xFFFFFFFFu);
    for (uint32_t k = 0; k < ctx->n_ordered; ++k) pg[ctx->h_ord_path[k]] = ctx->h_ord_group[k];
    DevBuf d_pg, d_out;
    if ((rc = ensure(ctx, d_pg, pg.size() * sizeof(uint32_t)))) return rc;
    PNX_HIP(ctx, hipMemcpyAsync(d_pg.p, pg.data(), pg.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    rc = launch_visit_counts(ctx, item_lo, item_hi, d_pg, d_out);
    const size_t cells = (size_t)ctx->n_groups * (item_hi - item_lo);
    if (!rc && cells) {
        hipError_t e = hipMemcpyAsync(out, d_out.p, cells * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) rc = ctx->fail(PNX_EHIP, "hipMemcpyAsync: %s", hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(ctx->stream);
    release(d_pg);
    release(d_out);
    return rc;
}

int pnx_profile_enable(pnx_ctx *ctx, int on) {
    if (!ctx) return PNX_EINVAL;
    ctx->prof.on = on != 0;
    return PNX_OK;
}

int pnx_profile_select(pnx_ctx *ctx, uint32_t slot_mask) {
    if (!ctx) return PNX_EINVAL;
    ctx->prof.mask = slot_mask;
    return PNX_OK;
}

int pnx_profile_sample(pnx_ctx *ctx, uint32_t every) {
    if (!ctx) return PNX_EINVAL;
    if (every == 0) return ctx->fail(PNX_EINVAL, "pnx_profile_sample: every must be at least 1");
    ctx->prof.every = every;
    for (auto &x : ctx->prof.seen) x = 0;
    return PNX_OK;
}

int pnx_profile_read(pnx_ctx *ctx, double ms[PNX_K_COUNT], uint64_t launches[PNX_K_COUNT]) {
    if (!ctx) return PNX_EINVAL;
    int rc = pnx_sync(ctx);
    if (rc) return rc;
    for (int i = 0; i < PNX_K_COUNT; ++i) {
        if (ms) ms[i] = ctx->prof.ms[i];
        if (launches) launches[i] = ctx->prof.launches[i];
    }
    return PNX_OK;
}

int pnx_profile_reset(pnx_ctx *ctx) {
    if (!ctx) return PNX_EINVAL;
    int rc = pnx_sync(ctx);
    if (rc) return rc;
    for (int i = 0; i < PNX_K_COUNT; ++i) {
        ctx->prof.ms[i] = 0;
        ctx->prof.launches[i] = 0;
    }
    return PNX_OK;
}

int pnx_info(pnx_ctx *ctx, pnx_info_t *out) {
    if (!ctx || !out) return PNX_EINVAL;
    out->n_steps = ctx->n_steps;
    out->n_items = ctx->n_items;
    out->n_paths = ctx->n_paths;
    out->n_ordered = ctx->n_ordered;
    out->n_groups = ctx->n_groups;
    out->n_tiles = ctx->n_tiles;
    out->tile_items = ctx->tile_blocks * BLOCK_ITEMS;
    out->n_general_paths = ctx->n_run_paths + ctx->n_scatter_paths;
    out->n_run_paths = ctx->n_run_paths;
    out->n_scatter_paths = ctx->n_scatter_paths;
    out->n_sorted_paths = ctx->n_sorted_paths;
    out->rows_tile_major = ctx->rows_valid && ctx->rows_tile_major ? 1 : 0;
    out->n_rows = ctx->rows_valid ? ctx->n_rows : 0;
    out->n_growth_table_builds = ctx->gtab.n_builds;
    out->n_rows_in_order = 0;
    if (ctx->rows_valid && ctx->have_order && ctx->h_rt_span.size() == ctx->n_paths)
        for (uint32_t k = 0; k < ctx->n_ordered; ++k) out->n_rows_in_order += ctx->h_rt_span[ctx->h_ord_path[k]];
    if (use_rows(ctx)) {  // a row is one block of 2048 items, whatever PNX_CFG_TILE_BLOCKS says
        out->n_tiles = ctx->n_blocks;
        out->tile_items = BLOCK_ITEMS;
    }
    out->n_runs = ctx->n_runs;
    out->n_reruns = ctx->n_reruns;
    out->weighted = ctx->weighted ? 1 : 0;
    out->n_band_passes = ctx->n_band_passes;
    out->band_route_failed = ctx->band_failed ? 1 : 0;
    out->n_rows_q_passes = ctx->n_rows_q_passes;
    out->n_spilled_last = ctx->n_spilled_last;
    out->band_splits = ctx->band_splits;
    out->n_spilled_total = ctx->n_spilled_total;
    out->n_spill_bursts_last = ctx->n_spill_bursts_last;
    out->n_loose_groups_last = ctx->n_loose_last;
    out->n_path_cuts = (uint32_t)ctx->h_cuts.size();
    out->n_band_entries = ctx->entries_valid ? ctx->n_entries : 0u;
    out->n_sorted_copies = ctx->n_sorted_copies;
    out->reserved1 = 0;
    return PNX_OK;
}

int pnx_info_sized(pnx_ctx *ctx, void *out, size_t out_bytes, size_t *lib_bytes) {
    if (lib_bytes) *lib_bytes = sizeof(pnx_info_t);
    if (!ctx || !out) return PNX_EINVAL;
    pnx_info_t full;
    std::memset(&full, 0, sizeof full);
    const int rc = pnx_info(ctx, &full);
    if (rc) return rc;
    std::memcpy(out, &full, std::min(out_bytes, sizeof full));
    return PNX_OK;
}

int pnx_config(pnx_ctx *ctx, int key, int64_t value) {
    if (!ctx) return PNX_EINVAL;
    switch (key) {
        case PNX_CFG_CACHE_INDEX:
            ctx->cache_index = value != 0;
            return PNX_OK;
        case PNX_CFG_TILE_BLOCKS:
            if (value != 1 && value != 2) return ctx->fail(PNX_EINVAL, "tile_blocks must be 1 or 2");
            ctx->tile_blocks = (uint32_t)value;
            if (ctx->have_csr) {
                invalidate_results(ctx);
                uint32_t keep = ctx->last_general_paths;
                set_geometry(ctx);
                ctx->last_general_paths = keep;
            }
            return PNX_OK;
        case PNX_CFG_USE_WEIGHTS:
            if (value && !ctx->have_weights) return ctx->fail(PNX_EINVAL, "no weights are resident");
            if (ctx->weighted != (value != 0)) {
                if (ctx->tk_count) (void)drain_streams(ctx);
                for (auto &t : ctx->tk) t.in_flight = false;
                ctx->tk_count = 0;
                ctx->tk_oldest = ctx->tk_next;
                ctx->hist_valid = false;  // the histogram changes meaning; coverage and M do not
            }
            ctx->weighted = value != 0;
            return PNX_OK;
        case PNX_CFG_COVER_WAVES:
            if (value != 1 && value != 2 && value != 4 && value != 8) return ctx->fail(PNX_EINVAL, "cover waves must be 1, 2, 4 or 8");
            ctx->cover_waves = (int)value;
            return PNX_OK;
        case PNX_CFG_COVER_VARIANT:
            if (value < 0 || value > 3) return ctx->fail(PNX_EINVAL, "cover variant must be 0, 1, 2 or 3");
            if (value != 3 && !step_routes(ctx)) return PNX_EINVAL;  // the cross-check module is not installed: the error says so
            if ((int)value != ctx->cover_variant && ctx->have_csr) {
                invalidate_results(ctx);
                ctx->index_valid = false;
            }
            ctx->cover_variant = (int)value;
            return PNX_OK;
        case PNX_CFG_MAX_IN_FLIGHT:
            if (value < 1 || value > PNX_MAX_IN_FLIGHT) return ctx->fail(PNX_EINVAL, "PNX_CFG_MAX_IN_FLIGHT must be in 1..%d", PNX_MAX_IN_FLIGHT);
            if (ctx->tk_count || ctx->gslot_count) return ctx->fail(PNX_EINVAL, "PNX_CFG_MAX_IN_FLIGHT cannot change while work is in flight");
            ctx->max_in_flight = (int)value;
            return PNX_OK;
        case PNX_CFG_ROWS_LAYOUT:
            if (value < 0 || value > 2) return ctx->fail(PNX_EINVAL, "rows layout must be 0 (auto), 1 (tile-major) or 2 (path-major)");
            ctx->rows_layout = (int)value;
            return PNX_OK;
        case PNX_CFG_COVER_ROUTE:
            if (value < 0 || value > 2) return ctx->fail(PNX_EINVAL, "cover route must be 0 (auto), 1 (one-shot over the steps) or 2 (path rows)");
            ctx->cover_route = (int)value;
            return PNX_OK;
        case PNX_CFG_HIST_IN_COVER:
            if (value > 1) return ctx->fail(PNX_EINVAL, "hist_in_cover must be 0 or 1");
            if (ctx->tk_count) return ctx->fail(PNX_EINVAL, "PNX_CFG_HIST_IN_COVER cannot change while passes are in flight");
            ctx->hist_in_cover = value != 0;
            return PNX_OK;
        case PNX_CFG_DROP_GROWTH_TABLES:
            for (auto &sl : ctx->gslot)
                if (sl.pending && sl.done) PNX_HIP(ctx, hipEventSynchronize(sl.done));
            ctx->gtab.valid = false;
            ctx->gtab.first_part_done = false;
            return PNX_OK;
        case PNX_CFG_DROP_DERIVED:
            if (ctx->d_rows.borrowed || ctx->d_steps12.borrowed) return ctx->fail(PNX_EINVAL, "this context borrows its graph (pnx_share_csr)");
            if (ctx->have_csr) invalidate_results(ctx);
            ctx->rows_valid = false;
            ctx->index_valid = false;
            ctx->band_failed = false;
            ctx->n_band_passes = 0;
            if (ctx->n_sorted_paths == 0) ctx->steps_prepared = false;  // sorted paths stay sorted: their caller order is kept once
            return PNX_OK;
        case PNX_CFG_INDEX_COARSE:
            if (value < 1 || value > 4096) return ctx->fail(PNX_EINVAL, "index_coarse must be in 1..4096");
            ctx->index_coarse = (uint32_t)value;
            ctx->index_valid = false;
            return PNX_OK;
        case PNX_CFG_COVER_SPLIT:
            if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return ctx->fail(PNX_EINVAL, "cover split must be 0 (auto), 1, 2, 4 or 8");
            ctx->cover_split = (int)value;
            return PNX_OK;
        case PNX_CFG_ROWS_KERNEL:
            if (value < 0 || value > 2) return ctx->fail(PNX_EINVAL, "rows_kernel must be 0 (auto), 1 or 2");
            ctx->rows_kernel = (int)value;
            return PNX_OK;
        case PNX_CFG_COVER_SKIP:
            if (value < 0 || value > 2) return ctx->fail(PNX_EINVAL, "cover_skip must be 0 (auto), 1 or 2");
            ctx->cover_skip = (int)value;
            return PNX_OK;
        case PNX_CFG_INDEX_BY_ENTRY:
            if (value < 0 || value > 2) return ctx->fail(PNX_EINVAL, "index_by_entry must be 0 (auto), 1 or 2");
            ctx->index_by_entry = (int)value;
            ctx->index_valid = false;
            return PNX_OK;
        case PNX_CFG_INDEX_PROBE:
            if (value != 16 && value != 32) return ctx->fail(PNX_EINVAL, "index probe must be 16 or 32 ids");
            ctx->index_probe_ids = (int)value;
            ctx->index_valid = false;
            return PNX_OK;
        case PNX_CFG_BLOCKING_SYNC:
            ctx->blocking_sync = value != 0;
            return PNX_OK;
        case PNX_CFG_PAIRS_VARIANT:
            if (value != 0 && value != 1) return ctx->fail(PNX_EINVAL, "PNX_CFG_PAIRS_VARIANT must be 0 or 1");
            ctx->pairs_variant = (int)value;
            return PNX_OK;
        case PNX_CFG_SORT_SHUFFLED:
            ctx->sort_shuffled = value != 0;
            return PNX_OK;
        case PNX_CFG_OVERLAP_PHASES:
            ctx->overlap_phases = value != 0;
            return PNX_OK;
        case PNX_CFG_COMM_REDUCE_HIST:
            if (ctx->tk_count) return ctx->fail(PNX_EINVAL, "PNX_CFG_COMM_REDUCE_HIST cannot change while a pass is in flight");
            ctx->comm_reduce_hist = value != 0;
            return PNX_OK;
        case PNX_CFG_KEEP_PRESENCE:
            ctx->keep_M_user = value != 0;
            return PNX_OK;
        default:
            return ctx->fail(PNX_EINVAL, "unknown config key %d", key);
    }
}

}  // extern "C"
