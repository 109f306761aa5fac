// pnx_context.hpp -- device context of the MI355X hist/growth engine (internal).
//
// One pnx_ctx owns one HIP device, one stream, the resident graph (CSR of path steps in
// HBM), the visiting order, and the result buffers.  See include/panacus_amd.h for the ABI
// and DESIGN.md for the data layout.
#pragma once

#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/panacus_amd.h"

namespace pnx {

// Route of the quorum pair's inner sums (kernels_closed_form.hip, K7): fused (terms in LDS) up to 384 groups, two kernels
// through HBM above; PNX_QUORUM_ROUTE = 0 / 1 forces one.  The one-shot pass reads it too: the two-kernel route's many
// workgroups must start BEHIND the pass's index kernel (an event between index and coverage kernel), the fused route's need not.
inline bool quorum_route_fused(uint32_t n) {
    const char *e = getenv("PNX_QUORUM_ROUTE");
    return e && (e[0] == '0' || e[0] == '1') ? e[0] == '1' : n <= 384u;
}


// ---- geometry of the presence bit matrix ------------------------------------------------
// A "block" is 2048 consecutive item ids held as 64 u32 words, one per lane of a wave:
// item n  ->  block n / 2048, word (lane) n % 64, bit (n % 2048) / 64.
// Lane-interleaving makes both the coverage write-out (one coalesced 256 B store per bit)
// and the per-rank row reads of the growth kernel (256 B per wave) fully coalesced, and
// makes consecutive (sorted) step ids hit consecutive LDS banks.
constexpr uint32_t BLOCK_ITEMS = 2048;
constexpr uint32_t BLOCK_WORDS = 64;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool borrowed = false;  // p belongs to another context (pnx_share_csr): never freed here
};

// One enqueued coverage pass.  Several tickets exist so that pass k+1 (.. k+3) can be enqueued before the
// host has looked at pass k: each owns its result counters in HBM, a pinned staging copy and
// the event that marks "results of this pass are on the host".
constexpr uint32_t HIST_REPLICAS = 64;
constexpr uint32_t HIST_FUSED_MAX_BINS = 4096;  // 32 KB of LDS per workgroup

struct Ticket {
    // one device block [flags: u32[8] | hist: (G+1) u64 | group flags: G u8] so that a pass needs
    // one memset and one device-to-host copy; flags[0] violations, [1] #general paths in the order
    DevBuf d_block;
    uint32_t *d_flags = nullptr;
    uint64_t *d_hist = nullptr;
    uint8_t *d_grp_general = nullptr;
    uint32_t *d_band_scratch = nullptr;  // 16 words behind them, cleared with them: what the workgroups of k_band_tail share
    size_t block_bytes = 0;
    void *h_block = nullptr;  // pinned copy of flags + hist
    size_t h_cap = 0;         // bytes of h_block
    uint32_t *h_flags = nullptr;
    uint64_t *h_hist = nullptr;
    hipEvent_t done = nullptr;
    bool done_blocking = false;
    bool in_flight = false;
    // per-pass working set (so that the index of pass k+1 can be built, and the histogram of pass k-1
    // taken, while the coverage kernel of pass k runs): the order-aligned index arrays, the coverage
    // vector, and -- when the tile index is rebuilt in every pass -- the boundary table itself
    DevBuf d_ord_tfirst, d_ord_tspan, d_ord_off, d_win_lo, d_win_hi, d_countable, d_tile_idx_own;
    DevBuf d_group_first;  // one-shot route: n_groups + 1 u32, where the entries of every group begin in the visiting order
    DevBuf d_band_clist, d_band_ccnt;  // ... of a graph of many short paths: per band the entries that have steps there, and how many
    hipEvent_t ev_pre = nullptr, ev_cov = nullptr;  // index ready / coverage vector ready
    bool pre_recorded = false;                       // ev_pre was recorded by this pass (a one-shot pass leaves it out where nothing waits for it)
    // rows route, up to HIST_FUSED_MAX_BINS bins: the coverage kernel adds the histogram itself, into HIST_REPLICAS copies at the
    // end of d_block (cleared with it); k_hist_publish adds them up and writes flags + hist into h_block directly
    uint64_t *d_hist_rep = nullptr;
    bool hist_fused = false;
    void *h_block_mapped = nullptr;  // h_block as the device addresses it
    bool host_written = false;       // the publishing kernel of the pass wrote flags + hist to h_block itself
    hipEvent_t ev_reader = nullptr;  // a closed-form call reads the counters in d_block (pnx_growth_closed_form_async):
    bool has_reader = false;         // the next pass on this ticket clears the block only after that kernel
    // how the pass was LAUNCHED (the context's want_M / last_general_paths may have changed by the
    // time the pass is settled): it merged the scatter rows of M / it wrote the presence matrix
    bool used_m = false, wrote_m = false;
    bool band = false;  // the pass took the one-shot route over the steps (kernels_band.hip); flags[5] says whether it held
};

// what the upload's read of the steps leaves of every chunk of 4096 steps (pass_pipeline.hip: k_chunk_summaries)
struct ChunkSummary {
    uint32_t s[5];      // the ids at 0, 1/4, 1/2, 3/4 and the end of the chunk
    uint32_t up, down;  // steps (three of every four) that go to a larger / a smaller id
};

struct Profile {
    bool on = false;
    bool open = false;          // a prof_begin of a selected slot awaits its prof_end
    hipStream_t open_stream = nullptr;
    uint32_t mask = 0xFFFFFFFFu;  // slots that are timed (pnx_profile_select)
    uint32_t every = 1;           // ... every `every`-th launch of them (pnx_profile_sample)
    uint64_t seen[PNX_K_COUNT] = {0};
    double ms[PNX_K_COUNT] = {0};
    uint64_t launches[PNX_K_COUNT] = {0};
    // pending (start, stop, slot) event triples, resolved at the next sync
    struct Pending {
        hipEvent_t a, b;
        int slot;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
};

}  // namespace pnx

struct pnx_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t stream_cf = nullptr;  // closed-form kernels (K7): independent of the coverage passes
    // A plain histogram pass is three phases on three streams chained by events: the index (K0, the
    // order-aligned copy) on stream_pre, the coverage kernel on `stream`, the histogram + the copy of the
    // counters to the host on stream_post -- so K0 of pass k+1 and K2 of pass k-1 run beside K1 of pass k,
    // and the K1s follow each other without the short kernels in between.  s_* = the streams of the
    // pass being enqueued (all three = `stream` when the pass also writes / merges the presence matrix).
    hipStream_t stream_pre = nullptr, stream_post = nullptr;  // made when a pass is first enqueued behind one still in flight
    hipStream_t s_pre = nullptr, s_main = nullptr, s_post = nullptr;
    bool overlap_phases = true;  // PNX_CFG_OVERLAP_PHASES
    bool last_pass_phased = false;
    std::string err;
    hipDeviceProp_t prop;

    // ---- resident graph (a1: ItemTable) ----
    uint32_t n_items = 0, n_paths = 0;
    uint64_t n_steps = 0;
    bool have_csr = false, weighted = false, have_exclude = false;
    bool have_weights = false;  // weights are resident (weighted = resident AND enabled)
    pnx::DevBuf d_items, d_path_off, d_weights, d_exclude;
    // derived once per upload by one streaming pass (prepare_steps, kernels_cover.hip), shared by borrowers:
    //   d_steps12   ceil(S / 8) x 12 bytes: the step ids modulo 4096, 12 bits each, 8 steps per three dwords --
    //               inside its tile a step needs no more, and the coverage kernel streams these 1.5 bytes per
    //               step instead of 4;
    //   d_path_mono P x u8: 1 = the path is NOT tile-monotone (its 2048-id tile sequence goes both up
    //               and down), decided exactly; 0 = the boundary index serves it
    pnx::DevBuf d_steps12, d_path_mono;
    bool steps_prepared = false;
    std::vector<uint64_t> h_path_off;
    // internal renumbering of the items by a caller key (kernels_relabel.hip): resident steps, weights and
    // flags are in INTERNAL ids; per-item results are mapped back to the caller's ids on the way out
    bool relabeled = false;
    pnx::DevBuf d_new_of_old, d_old_of_new, d_countable_ext;  // n_items + 1 u32 each

    // ---- visiting order (a2) ----
    uint32_t n_ordered = 0, n_groups = 0;
    bool have_order = false;
    pnx::DevBuf d_ord_path, d_ord_group;
    std::vector<uint32_t> h_ord_path, h_ord_group;

    // ---- tile index over the CSR (K0) ----
    uint32_t tile_blocks = 1;  // blocks per coverage tile (WT)
    uint32_t index_coarse = 8; // every index_coarse-th tile boundary is searched exactly (K0 pass A)
    int cover_waves = 4;       // waves (= tiles) per workgroup of the pipelined coverage kernel
    int index_probe_ids = 16;  // ids per later probe of the index search: 16 (one 64-byte sector) or 32
    int index_by_entry = 0;    // K0 thread numbering: 0 = automatic, 1 = one thread per index entry, 2 = path-major
    int cover_skip = 0;        // window skipping of the coverage kernel: 0 = automatic, 1 = whenever legal, 2 = never
    int cover_split = 0;       // waves per tile of the coverage kernel: 0 = automatic, 1, 2, 4, 8
    uint64_t n_rows_q_passes = 0;  // passes that took k_rows_cover_q
    int rows_kernel = 0;       // PNX_CFG_ROWS_KERNEL: 0 = automatic, 1 = k_rows_cover, 2 = k_rows_cover_q wherever legal
    bool hist_in_cover = true;  // PNX_CFG_HIST_IN_COVER: the coverage kernel over rows adds the histogram itself (0: K2 reads the coverage vector)
    int cover_variant = 3;     // 0 = plain, 1 = software-pipelined, 2 = pipelined + non-temporal loads (all three over the
                               // steps), 3 = over path rows (kernels_rows.hip)
    uint32_t n_blocks = 0, n_tiles = 0;
    bool index_valid = false;
    bool cache_index = true;
    pnx::DevBuf d_tile_idx;    // sparse: sum over paths of (tiles spanned + 1) u64 (kept across passes: PNX_CFG_CACHE_INDEX)
    pnx::DevBuf d_tfirst, d_tspan, d_idx_off;  // per path: first tile, tiles spanned, row offset (n_paths + 1)
    // (the same per entry of the visiting order, and the window bands, are rebuilt every pass: Ticket)
    std::vector<uint32_t> h_tfirst;  // host copy (order normalisation)
    bool order_normalized = false;   // the paths of every group are sorted by their first tile
    bool spans_valid = false;  // the three arrays above match the resident CSR and tile size
    uint32_t max_span = 0;
    uint64_t idx_entries = 0;
    // paths sorted by id at preparation (PNX_CFG_SORT_SHUFFLED): their steps in the caller's order, kept for pnx_get_csr
    pnx::DevBuf d_unsorted;      // n_unsorted u32, the sorted paths one after the other
    pnx::DevBuf d_sorted_coff;   // n_sorted_paths + 1 u64: offsets into d_unsorted
    pnx::DevBuf d_sorted_path;   // n_sorted_paths u32: which path
    uint32_t n_sorted_paths = 0;
    uint64_t n_unsorted = 0;
    bool sort_shuffled = true;
    pnx::DevBuf d_path_class;  // n_paths u8: 0 = tile-monotone, 1 = general (scatter route)
    pnx::DevBuf d_flags;       // scratch flag block (upload validation)
    uint32_t last_general_paths = 0;  // scatter-route paths known to be in the order (=> M is needed)

    // ---- GFA text in HBM (kernels_gfa.hip): the bytes whose step columns pnx_set_csr_gfa tokenises ----
    pnx::DevBuf d_gfa_text;
    pnx::DevBuf d_name_tab;  // node2id of an upload by name (name_table.hpp): lives from the tokeniser to the L lines of the same upload
    const char *gfa_text_host = nullptr;  // what was uploaded (pnx_gfa_text_upload), to recognise it again
    uint64_t gfa_text_bytes = 0;

    // ---- path rows: the path x item presence table (kernels_rows.hip), derived once per upload ----
    pnx::DevBuf d_rows;       // n_rows x 256 bytes: row (p, t) = the presence bits of path p on item tile t (block layout)
    pnx::DevBuf d_row_base;   // n_paths u32: row (p, t) sits at row_base[p] + t * row_tstride (modulo 2^32)
    pnx::DevBuf d_id_minmax;  // 2 x n_paths u32: smallest / largest id on every path
    pnx::DevBuf d_rt_first, d_rt_span;  // n_paths u32 each: first tile / tiles spanned, from the id range
    std::vector<uint32_t> h_id_minmax, h_rt_first, h_rt_span, h_row_base;
    uint32_t row_tstride = 0;   // n_paths: tile-major over all (tile, path) pairs; 1: path-major over the spans
    uint32_t rows_max_span = 0;
    uint64_t n_rows = 0;
    bool rows_valid = false, rows_tile_major = false;
    int rows_layout = 0;        // PNX_CFG_ROWS_LAYOUT: 0 = chosen from the shape, 1 = tile-major, 2 = path-major

    // ---- one-shot route over the steps (kernels_band.hip): the first sweep of a graph whose paths are sorted by id ----
    int cover_route = 0;         // PNX_CFG_COVER_ROUTE: 0 = chosen per pass, 1 = band route whenever possible, 2 = path rows only
    bool band_failed = false;    // a band pass found a path that is not sorted by id: this upload takes the rows from now on
    uint32_t n_band_passes = 0;  // band passes enqueued on this upload (the second sweep of a graph derives the rows)
    bool pass_band = false;      // the pass being enqueued takes the band route
    uint32_t band_splits = 1;    // ... with this many workgroups per band (each takes a range of the visiting order)
    // steps found outside the band they were dealt to (paths that are not sorted by id): the list of a pass, and the set of
    // (group, id) pairs its tail has added -- slots carry the generation of the pass that wrote them, so no pass clears the set
    pnx::DevBuf d_spill, d_spill_dir, d_spill_set, d_band_probe;
    // the entries of a one-shot pass: the visiting order with every path cut where it turns round or jumps back (pieces of a path
    // are entries of their own under the path's group; an order over paths without such breaks is its own entry list)
    pnx::DevBuf d_chunk_sum, d_ent_start, d_ent_len, d_ent_group;
    std::vector<pnx::ChunkSummary> h_chunk_sum;
    std::vector<uint64_t> h_cuts;      // absolute step positions, path by path
    std::vector<uint32_t> h_cut_off;   // n_paths + 1: the cuts of path p are h_cuts[h_cut_off[p] .. h_cut_off[p + 1])
    std::vector<uint8_t> h_jumbled;    // n_paths: most chunks of the path go both ways
    std::vector<uint64_t> h_sorted_at; // n_paths: where in d_items (behind the graph's steps) the path is stored once more in the order of the ids, or 0
    uint32_t n_sorted_copies = 0;
    std::vector<uint32_t> h_ent_group;
    uint32_t n_entries = 0;
    bool entries_valid = false;        // (of the order and the graph as they stand)
    pnx::DevBuf d_group_loose, d_entry_loose, d_loose_bits;  // groups with a path that does not follow the ids at all: their flags, their presence bitmaps (kernels_band.hip: BandLoose)
    uint32_t n_loose_last = 0;                // ... how many the pass settled last took in
    bool loose_dirty = false;                 // a pass gave up waiting for its marking workgroups: the bitmaps are cleared before the next one
    uint32_t spill_cap = 0, spill_gen = 0;
    uint64_t spill_slots = 0;
    uint64_t n_spilled_total = 0;  // spilled steps of all settled passes of this upload
    uint32_t n_spilled_last = 0;   // ... of the pass settled last
    uint32_t n_spill_bursts_last = 0;  // ... in this many bursts (a burst: what one wave found in one load)

    // ---- run index: tile route for non-monotone paths (kernels_runs.hip) ----
    // path_class: 0 tile-monotone (K0 index), 1 not monotone & unclassified, 2 run route, 3 scatter route
    pnx::DevBuf d_run_start, d_run_len, d_run_tile, d_run_path;     // runs in path order
    pnx::DevBuf d_srun_start, d_srun_len, d_srun_group, d_run_tile_off;  // sorted by (tile, group)
    uint64_t n_runs = 0;
    uint32_t n_run_paths = 0, n_scatter_paths = 0;
    bool runs_sorted = false;
    pnx::DevBuf d_chunk_off;            // n_paths + 1 u64: chunks (4096 steps) of the paths before p
    std::vector<uint64_t> h_chunk_off;  // its host copy (staging of the upload; chunk total)
    bool chunk_off_valid = false;
    pnx::DevBuf d_rb[6], d_rs[6];       // scratch of the build / of the sort: kept, only ever grown

    // ---- results ----
    pnx::DevBuf *d_countable_done = nullptr;  // coverage vector (n_items + 1 u32) of the pass settled last
    static constexpr int N_TICKETS = PNX_MAX_IN_FLIGHT;
    pnx::Ticket tk[N_TICKETS];  // in-flight / finished passes (ring)
    int tk_next = 0, tk_oldest = 0, tk_count = 0;
    int max_in_flight = 2;      // PNX_CFG_MAX_IN_FLIGHT: passes (and closed-form calls) the caller may keep in flight
    int tk_last() const { return (tk_next + N_TICKETS - 1) % N_TICKETS; }  // the pass enqueued last
    pnx::Ticket *cur = nullptr;        // the ticket the launch functions write to
    pnx::Ticket *last_done = nullptr;  // holds the last verified histogram
    pnx::DevBuf d_M;          // n_groups * n_blocks * 64 u32 presence matrix
    bool hist_valid = false;
    bool M_valid = false;
    bool want_M = false;        // the pass being enqueued also writes the presence matrix
    bool keep_M_user = false;   // PNX_CFG_KEEP_PRESENCE
    bool growth_needs_M = false;
    uint64_t n_reruns = 0;
    bool blocking_sync = false;  // PNX_CFG_BLOCKING_SYNC: wait for a pass asleep instead of spinning

    // ---- growth ----
    pnx::DevBuf d_perms, d_cov_thr, d_qtab, d_cmask, d_wplanes, d_growth_out, d_thr_meta;
    uint32_t g_R = 0, g_T = 0;
    std::vector<uint32_t> h_thr_meta;  // cov_thr[T] then is_q0[T]
    // host copies of the call's tables (the launch code needs them; nothing is read back from the
    // device) and the staging vectors of its uploads -- they live until the next growth call, which
    // first waits for this one, so no upload ever needs its own synchronisation
    std::vector<uint32_t> h_perms, h_qtab, h_growth_tabs, h_growth_aux;
    uint32_t n_wplanes = 0;
    bool wplanes_valid = false;
    pnx::DevBuf d_wdigits;       // 7-bit digits of the weights, one byte per item in presence order (kernels_pairs_mfma.hip)
    bool wdigits_valid = false;
    int pairs_variant = 1;       // PNX_CFG_PAIRS_VARIANT: 0 vector ALUs (AND + popcount), 1 matrix cores (int8 MFMA)
    bool growth_pending = false;

    // ---- pairwise intersections / plain presence export (kernels_pairs.hip) ----
    pnx::DevBuf d_inter, d_pair_partial, d_plain;

    // ---- walks kept on the device for pnx_set_csr_cut (pnx_gfa_walks) ----
    pnx::DevBuf d_walk_node, d_walk_back;
    std::vector<uint64_t> h_walk_off;
    bool walks_valid = false;
    pnx::DevBuf d_link_uv, d_link_oo;  // the distinct edges of the L lines pnx_gfa_walks parsed, by id (pnx_set_csr_walks with PNX_EDGES_FROM_LINKS)
    uint32_t n_link_edges = 0;
    bool links_valid = false;

    // ---- closed-form quorum sums (kernels_closed_form.hip): scratch kept across calls ----
    pnx::DevBuf d_cf[6];
    void *h_cf = nullptr;  // pinned: the (n+1)^2 sums handed back to the caller, then the staged inputs
    size_t h_cf_cap = 0;
    hipEvent_t ev_cf = nullptr;
    bool cf_pending = false;
    // ---- whole closed forms on the device (pnx_growth_closed_form_*).  What depends on (n, threshold pairs) alone is kept
    // as tables shared by all calls; a call has a slot with its own stream, so that calls in flight run beside each other
    struct GrowthTables {
        bool valid = false;
        bool first_part_done = false;  // setup + perc_mult rows of (n, pairs) are enqueued (pnx_growth_tables_begin); the quorum part is not
        uint32_t n = 0, T = 0;
        uint32_t branch[PNX_GROWTH_MAX_PAIRS] = {}, cov[PNX_GROWTH_MAX_PAIRS] = {};
        double quorum[PNX_GROWTH_MAX_PAIRS] = {};
        pnx::DevBuf d_par;               // [quorum f64 T | branch u32 T | cov u32 T]
        pnx::DevBuf d_L, d_nf, d_mf, d_mq;  // log2 table, per-pair running sums, m_quorum
        pnx::DevBuf d_pm, d_lsq;         // perc_mult[t][i][m]; log2 of the quorum branch's inner sums [t][i][m]
        pnx::DevBuf d_terms, d_sum;      // scratch of a build: terms of the inner sums, the sums
        uint32_t lds_attr_n = 0;         // the n the kernels' LDS attributes were set for
        hipEvent_t ready = nullptr;
        uint64_t gen = 0, n_builds = 0;
    } gtab;
    struct GrowthSlot {
        hipStream_t stream = nullptr;
        void *h_io = nullptr;      // pinned: hist in, out[T][n] behind it
        void *d_io_mapped = nullptr;  // the same memory as the device addresses it
        size_t h_cap = 0;
        hipEvent_t done = nullptr;
        bool pending = false;
        uint32_t n = 0, n_pairs = 0;
        size_t out_off = 0;
        uint64_t tab_gen = 0;      // generation of the tables this slot's stream has waited for
    } gslot[PNX_MAX_IN_FLIGHT + 2];
    int gslot_next = 0, gslot_oldest = 0, gslot_count = 0, gslot_cap = 2;  // ring over the first gslot_cap slots

    // ---- multi-GPU (pnx_comm.hip): RCCL communicator, opened with dlopen on first use ----
    void *comm = nullptr;          // ncclComm_t
    pnx::DevBuf d_comm_word;       // one word for pnx_comm_barrier
    int comm_rank = 0, comm_world = 1;
    bool comm_reduce_hist = true;  // with a communicator: every coverage pass is followed by an all-reduce of its flags + histogram

    pnx::Profile prof;

    int fail(int code, const char *fmt, ...) {
        char buf[768];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define PNX_HIP(ctx, call)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return (ctx)->fail(e_ == hipErrorOutOfMemory ? PNX_ENOMEM : PNX_EHIP,           \
                               "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),       \
                               __FILE__, __LINE__);                                         \
    } while (0)

namespace pnx {

int ensure(pnx_ctx *ctx, DevBuf &b, size_t bytes);
void release(DevBuf &b);

// profiling brackets around a kernel launch (on `stream`, default ctx->stream)
void prof_begin(pnx_ctx *ctx, int slot, hipStream_t stream = nullptr);
void prof_end(pnx_ctx *ctx);
int drain_streams(pnx_ctx *ctx);  // waits for everything enqueued on the context's pass streams
int prof_resolve(pnx_ctx *ctx, bool wait = true);

// pass_pipeline.hip
int launch_validate_items(pnx_ctx *ctx, uint32_t *d_bad);
int launch_chunk_summaries(pnx_ctx *ctx, uint32_t *d_bad);  // ... of a graph the one-shot route may take: the ids validated, the chunks summarised
int refine_path_cuts(pnx_ctx *ctx);                        // ... every cut moved from its chunk boundary to the step where the ids jump
void path_cuts_from_chunks(pnx_ctx *ctx);
int sort_jumbled_paths(pnx_ctx *ctx);                      // ... and the paths that follow the ids nowhere stored once more, sorted (upload_scan.hip)                   // ... and the paths cut where the summaries turn round or jump back
int launch_cover_pass(pnx_ctx *ctx);  // phases 1 + 2 (rows / band / step routes) + the histogram phase, for the current order
int ensure_chunk_off(pnx_ctx *ctx);
// The step routes (kernels_cover.hip, kernels_runs.hip): a cross-check module, libpanacus_hip_steps.so, opened on demand.
struct StepRoutes {
    int (*prepare_steps)(pnx_ctx *ctx);                                    // d_steps12 + d_path_mono (no-op when done)
    int (*restore_step_order)(pnx_ctx *ctx, uint32_t *d_items_copy);       // sorted paths back in the caller's order (pnx_get_csr)
    int (*launch_tile_index)(pnx_ctx *ctx);
    int (*build_run_index)(pnx_ctx *ctx);
    int (*sort_run_index)(pnx_ctx *ctx);                                    // (no-op when sorted)
    int (*launch_step_phases)(pnx_ctx *ctx, bool use_m, uint64_t m_words);  // phases 1 + 2 of a pass over the steps
};
const StepRoutes *step_routes(pnx_ctx *ctx);  // nullptr (and ctx->err) when the module is not installed
// kernels_gfa.hip
int gfa_text_upload(pnx_ctx *ctx, const char *text, uint64_t n_bytes);
int gfa_tokenise(pnx_ctx *ctx, const pnx_gfa_steps *g, DevBuf *d_backward);  // -> d_items, d_path_off, h_path_off, n_steps
int gfa_links_to_edges(pnx_ctx *ctx, const pnx_gfa_steps *g, DevBuf &d_e_uv, DevBuf &d_e_oo, uint32_t &n_edges);  // the L lines -> distinct edges by id
// kernels_hist.hip
int launch_hist(pnx_ctx *ctx, Ticket *tk);  // K2 of the pass in `tk`, on the stream of its histogram phase
// kernels_rows.hip
inline bool use_rows(const pnx_ctx *ctx) { return ctx->cover_variant == 3; }
int ensure_rows(pnx_ctx *ctx, bool validate);          // d_rows & co. (no-op when they exist)
int launch_rows_phases(pnx_ctx *ctx, bool write_m);    // phases 1 + 2 of a pass over rows
// kernels_band.hip
bool band_route_fits(const pnx_ctx *ctx, uint32_t n_entries);  // is the one-shot route worth it for this shape?
int ensure_band_entries(pnx_ctx *ctx);                         // the visiting order with the paths cut where they turn round or jump back: once per (graph, order)
int launch_band_phases(pnx_ctx *ctx, bool write_m);            // phases 1 + 2 of a one-shot pass over the steps
int launch_band_tail(pnx_ctx *ctx, Ticket *tk, bool write_m);  // phase 3: spilled steps added, histogram handed over
uint32_t band_route_splits(const pnx_ctx *ctx, uint32_t n_groups);  // workgroups per band for this shape
// kernels_runs.hip (step-route module)
int build_run_index(pnx_ctx *ctx);
int sort_run_index(pnx_ctx *ctx);
// kernels_cover.hip (step-route module)
int prepare_steps(pnx_ctx *ctx);
int restore_step_order(pnx_ctx *ctx, uint32_t *d_items_copy);
int launch_tile_index(pnx_ctx *ctx);
// kernels_growth.hip
int launch_growth(pnx_ctx *ctx, bool identity_perm);
int ensure_weight_planes(pnx_ctx *ctx, uint32_t *d_scratch = nullptr);  // W_p in presence layout
// kernels_pairs.hip
int launch_pair_intersections(pnx_ctx *ctx);  // -> ctx->d_inter (G x G u64)
int launch_pair_intersections_mfma(pnx_ctx *ctx);  // kernels_pairs_mfma.hip
int launch_presence_plain(pnx_ctx *ctx, DevBuf &out);
int launch_visit_counts(pnx_ctx *ctx, uint32_t lo, uint32_t hi, DevBuf &d_path_group, DevBuf &out);  // lo, hi: caller ids  // -> n_groups x (hi - lo) u32
// pnx_comm.hip
int comm_allreduce_u64(pnx_ctx *ctx, uint64_t *d_buf, size_t n, hipStream_t stream = nullptr);
int comm_reduce_pass(pnx_ctx *ctx, Ticket *t);
// kernels_relabel.hip
int relabel_by_keys(pnx_ctx *ctx, const uint64_t *h_keys);
int to_caller_ids_u32(pnx_ctx *ctx, const uint32_t *d_internal, uint32_t *d_caller);
int to_internal_ids_u8(pnx_ctx *ctx, const uint8_t *d_caller, uint8_t *d_internal);
int to_caller_ids_u8(pnx_ctx *ctx, const uint8_t *d_internal, uint8_t *d_caller);
int to_internal_ids_u32(pnx_ctx *ctx, const uint32_t *d_caller, uint32_t *d_internal);
// pnx_gfa_steps: the segments are named by bytes (a name hash on the device), and whether the library finds the S lines itself
static inline bool names_found_on_device(const pnx_gfa_steps *g) { return !g->name_off && !g->id_of_name && g->n_names == PNX_NAMES_FIND; }
static inline bool names_by_bytes(const pnx_gfa_steps *g) { return g->name_off != nullptr || names_found_on_device(g); }

// pnx_preload: one function per translation unit
void preload_gfa(unsigned what);
void preload_cut(unsigned what);
void preload_relabel(unsigned what);
void preload_pass(unsigned what);
void preload_band(unsigned what);
void preload_rows(unsigned what);
void preload_hist(unsigned what);
void preload_closed_form(unsigned what);
void preload_growth(unsigned what);
void preload_pairs(unsigned what);
void preload_pairs_mfma(unsigned what);

// kernels_cut.hip
int cut_walks(pnx_ctx *ctx, const pnx_walks *w, pnx_piece_event *events, uint64_t cap, uint64_t *n_events);
int flag_items(pnx_ctx *ctx, const uint32_t *h_ids, uint32_t n);
int gfa_edge_items(pnx_ctx *ctx, uint32_t n_paths, uint32_t n_nodes, const DevBuf &d_backward, const uint64_t *edge_uv, const uint8_t *edge_oo,
                   uint32_t n_edges, bool edges_on_device = false);  // edges_on_device: edge_uv / edge_oo are device arrays (gfa_links_to_edges)
int steps_to_caller_ids(pnx_ctx *ctx, uint32_t *d_items_copy, uint64_t n_steps);
int presence_to_caller_ids(pnx_ctx *ctx, const DevBuf &in, DevBuf &out);
// pansyn.hip
int pansyn_generate_device(pnx_ctx *ctx, uint64_t seed, uint32_t n_nodes, uint32_t n_paths,
                           int with_weights, uint64_t node_lo = 0);
int pansyn_rearrange_device(pnx_ctx *ctx, uint64_t seed);  // pansyn-v1r: the resident pansyn-v1 paths rearranged in place

}  // namespace pnx
