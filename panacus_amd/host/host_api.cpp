// host_api.cpp -- C entry points of libpanacus_host.so (host layer above the device ABI):
// closed-form growth, GFA front end, table writers.  Bound from Python with ctypes
// (panacus_amd/hostlib.py) and linked into the panacus-amd CLI.
#include <cstdint>
#include <algorithm>
#include <atomic>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <execinfo.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdexcept>
#include <string>
#include <vector>

#include "commands.hpp"
#include "commands_internal.hpp"
#include "gfa_graph.hpp"
#include "tables.hpp"
#include "growth_closed_form.hpp"
#include "linkage.hpp"
#include "thread_pool.hpp"

static thread_local std::string g_host_err;

extern "C" {

const char *pnh_last_error(void) { return g_host_err.c_str(); }

// Debugging aid (soak runs): on SIGSEGV / SIGBUS / SIGABRT / SIGFPE the C stack of the faulting thread goes to `path`
// (appended) before the default action takes place.
static char g_crash_path[512];
static void crash_handler(int sig) {
    int fd = ::open(g_crash_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) {
        char head[64];
        const int n = std::snprintf(head, sizeof head, "--- signal %d, pid %d\n", sig, (int)getpid());
        if (n > 0 && ::write(fd, head, (size_t)n) < 0) {
        }
        void *frames[64];
        const int k = backtrace(frames, 64);
        backtrace_symbols_fd(frames, k, fd);
        // What the process last wrote to stderr / stdout, when those are regular files: under pytest's descriptor capture they
        // are unlinked temporary files that die with the process -- and the ROCm runtime says WHY it aborts ("Memory access
        // fault by GPU node ...", "HSA_STATUS_ERROR_...") on stderr just before it does.  The tail of each goes to the crash log.
        for (int cap = 2; cap >= 1; --cap) {
            struct stat st;
            if (::fstat(cap, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) continue;
            static char tail[8192];
            const off_t from = st.st_size > (off_t)sizeof tail ? st.st_size - (off_t)sizeof tail : 0;
            const ssize_t got = ::pread(cap, tail, sizeof tail, from);
            if (got > 0) {
                const int m = std::snprintf(head, sizeof head, "--- last %d bytes of descriptor %d\n", (int)got, cap);
                if (m > 0 && ::write(fd, head, (size_t)m) < 0) {
                }
                if (::write(fd, tail, (size_t)got) < 0) {
                }
                if (::write(fd, "\n", 1) < 0) {
                }
            }
        }
        ::close(fd);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
void pnh_install_crash_handler(const char *path) {
    std::snprintf(g_crash_path, sizeof g_crash_path, "%s", path ? path : "/tmp/panacus_amd_crash.txt");
    void *warm[4];
    (void)backtrace(warm, 4);  // (loads libgcc now, not inside the handler)
    static char alt_stack[1 << 16];
    stack_t ss{};
    ss.ss_sp = alt_stack;
    ss.ss_size = sizeof alt_stack;
    sigaltstack(&ss, nullptr);
    struct sigaction sa{};
    sa.sa_handler = crash_handler;
    sa.sa_flags = SA_ONSTACK | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    for (int sig : {SIGSEGV, SIGBUS, SIGABRT, SIGFPE}) sigaction(sig, &sa, nullptr);
}

// ---- GFA front end (gfa_graph.hpp) ----
void *pnh_graph_load(const char *gfa_file, int index_edges) {
    try {
        return pnh::GraphStorage::from_gfa(gfa_file, index_edges != 0).release();
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return nullptr;
    }
}
// .pcsr cache (gfa_graph.hpp): 0 on success
int pnh_graph_save_cache(const void *g, const char *cache_file, const char *gfa_file) {
    try {
        static_cast<const pnh::GraphStorage *>(g)->save_cache(cache_file, gfa_file);
        return 0;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return 1;
    }
}
// NULL when the cache is missing, stale or lacks the edge index asked for
void *pnh_graph_from_cache(const char *cache_file, const char *gfa_file, int need_edges) {
    try {
        return pnh::GraphStorage::from_cache(cache_file, gfa_file, need_edges != 0).release();
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return nullptr;
    }
}
// (every free advances the epoch: a per-thread cache keyed by a graph's ADDRESS must not serve the next graph the allocator puts there)
static std::atomic<uint64_t> g_graph_epoch{0};
void pnh_graph_free(void *g) {
    g_graph_epoch.fetch_add(1);
    delete static_cast<pnh::GraphStorage *>(g);
}
uint64_t pnh_graph_n_nodes(const void *g) { return static_cast<const pnh::GraphStorage *>(g)->node_count(); }
// how the segments are named, as the device routes see it: 0 names the device does not take (longer than 16 bytes), 1 a
// number that is the rank of the S line, 2 a number that a table maps, 3 up to 16 bytes that are hashed; prefix8 receives
// what stands in front of the number (kinds 1 and 2; NUL-terminated, at most 8 bytes)
int pnh_graph_name_kind(const void *g, char *prefix8) {
    const auto *gs = static_cast<const pnh::GraphStorage *>(g);
    if (prefix8) {
        std::memset(prefix8, 0, 9);
        std::memcpy(prefix8, gs->name_prefix().data(), std::min<size_t>(8, gs->name_prefix().size()));
    }
    if (gs->names_by_bytes_on_device()) return 3;
    if (!gs->steps_tokenisable_on_device()) return 0;
    return gs->id_of_name().empty() ? 1 : 2;
}
uint64_t pnh_graph_n_edges(const void *g) { return static_cast<const pnh::GraphStorage *>(g)->edge_count(); }
uint64_t pnh_graph_n_paths(const void *g) { return static_cast<const pnh::GraphStorage *>(g)->path_segments().size(); }
const uint32_t *pnh_graph_node_lens(const void *g) { return static_cast<const pnh::GraphStorage *>(g)->node_lens().data(); }
// copies the display name of path i into buf (NUL-terminated, truncated to cap); returns its length
uint64_t pnh_graph_path_name(const void *g, uint64_t i, char *buf, uint64_t cap) {
    std::string s = static_cast<const pnh::GraphStorage *>(g)->path_segments().at(i).display();
    if (cap) {
        size_t n = std::min<size_t>(s.size(), cap - 1);
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size();
}

// ItemTable: first call with items == NULL returns the number of steps (prefsum filled), second fills items
int64_t pnh_graph_item_table(const void *g, int count_type, uint32_t *items, uint64_t *prefsum) {
    static thread_local pnh::ItemTable cache;
    static thread_local const void *cache_g = nullptr;
    static thread_local int cache_c = -1;
    static thread_local uint64_t cache_epoch = 0;
    try {
        if (cache_g != g || cache_c != count_type || cache_epoch != g_graph_epoch.load()) {
            cache_epoch = g_graph_epoch.load();
            cache = static_cast<const pnh::GraphStorage *>(g)->item_table((pnh::CountType)count_type);
            cache_g = g;
            cache_c = count_type;
        }
        if (prefsum) std::copy(cache.id_prefsum.begin(), cache.id_prefsum.end(), prefsum);
        int64_t n = (int64_t)cache.items.size();
        if (items) {
            std::copy(cache.items.begin(), cache.items.end(), items);
            cache = pnh::ItemTable();
            cache_g = nullptr;
        }
        return n;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return -1;
    }
}

// ActiveTable flags (n_items + 1 bytes) of a whole-path exclude list
int pnh_graph_exclude_flags(const void *g, int count_type, int group_mode, const char *group_file,
                            const char *exclude_file, uint8_t *flags) {
    try {
        const pnh::GraphStorage *gs = static_cast<const pnh::GraphStorage *>(g);
        pnh::ItemTable t = gs->item_table((pnh::CountType)count_type);
        std::vector<uint8_t> f = gs->exclude_flags((pnh::CountType)count_type, t, (pnh::GroupMode)group_mode,
                                                   group_file ? group_file : "", exclude_file);
        std::copy(f.begin(), f.end(), flags);
        return 0;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return -1;
    }
}

// ItemTable / exclude flags / uncovered bps under BED subset and exclude lists (GraphStorage::masked_table).
// Two calls: with items == NULL the sizes are returned (*n_steps, *n_uncovered); then the caller's arrays are
// filled (prefsum n_paths + 1, exclude n_items + 1 -- zeros without an exclude list, uncovered as id / bp arrays).
int pnh_graph_masked_table(const void *g, int count_type, int group_mode, const char *group_file, const char *subset_file,
                           const char *exclude_file, uint64_t *n_steps, uint64_t *n_uncovered, uint32_t *items,
                           uint64_t *prefsum, uint8_t *exclude, uint32_t *uncov_ids, uint64_t *uncov_bps) {
    try {
        const pnh::GraphStorage *gs = static_cast<const pnh::GraphStorage *>(g);
        pnh::MaskedTable m = gs->masked_table((pnh::CountType)count_type, (pnh::GroupMode)group_mode, group_file ? group_file : "",
                                              subset_file ? subset_file : "", exclude_file ? exclude_file : "");
        if (!items) {
            *n_steps = m.table.items.size();
            *n_uncovered = m.uncovered.size();
            return 0;
        }
        if (*n_steps != m.table.items.size() || *n_uncovered != m.uncovered.size()) throw std::runtime_error("size mismatch");
        std::copy(m.table.items.begin(), m.table.items.end(), items);
        std::copy(m.table.id_prefsum.begin(), m.table.id_prefsum.end(), prefsum);
        const uint64_t n = gs->number_of_items((pnh::CountType)count_type);
        std::fill(exclude, exclude + n + 1, (uint8_t)0);
        std::copy(m.exclude.begin(), m.exclude.end(), exclude);
        for (size_t k = 0; k < m.uncovered.size(); ++k) {
            uncov_ids[k] = m.uncovered[k].first;
            uncov_bps[k] = m.uncovered[k].second;
        }
        return 0;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return -1;
    }
}

// The device-side cut (pnx_set_csr_cut through upload_cut of commands.cpp) into the caller's context `ctx`
// (a pnx_ctx*): the cut table becomes that context's resident graph; uncovered bps come back as id / bp arrays
// (at most *n_uncovered entries on entry; the count on return).
int pnh_graph_cut_upload(const void *g, void *ctx, int count_type, int group_mode, const char *group_file, const char *subset_file,
                         const char *exclude_file, int growth_weights, uint64_t *n_uncovered, uint32_t *uncov_ids, uint64_t *uncov_bps) {
    try {
        const pnh::GraphStorage *gs = static_cast<const pnh::GraphStorage *>(g);
        pnh::cli::Masking mk;
        mk.mode = (pnh::GroupMode)group_mode;
        mk.group_file = group_file ? group_file : "";
        mk.subset_file = subset_file ? subset_file : "";
        mk.exclude_file = exclude_file ? exclude_file : "";
        const pnh::cli::Uncovered u = pnh::cli::upload_cut([ctx]() { return static_cast<pnx_ctx *>(ctx); }, *gs, (pnh::CountType)count_type, mk, growth_weights != 0);
        if (u.size() > *n_uncovered) throw std::runtime_error("room for fewer uncovered entries than found");
        *n_uncovered = u.size();
        for (size_t k = 0; k < u.size(); ++k) {
            uncov_ids[k] = u[k].first;
            uncov_bps[k] = u[k].second;
        }
        return 0;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return -1;
    }
}

// GraphStorage::edge_keys: n_edges + 1 entries (the item_key of pnx_set_csr_keyed)
int pnh_graph_edge_keys(const void *g, uint64_t *keys) {
    try {
        std::vector<uint64_t> k = static_cast<const pnh::GraphStorage *>(g)->edge_keys();
        std::copy(k.begin(), k.end(), keys);
        return 0;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return -1;
    }
}

// GraphStorage::edge_relabel: new_id has n_edges + 1 entries
int pnh_graph_edge_relabel(const void *g, uint32_t *new_id) {
    try {
        const pnh::GraphStorage *gs = static_cast<const pnh::GraphStorage *>(g);
        std::vector<uint32_t> r = gs->edge_relabel();
        if (r.empty())  // already ranked: the identity
            for (uint64_t id = 0; id <= gs->edge_count(); ++id) new_id[id] = (uint32_t)id;
        std::copy(r.begin(), r.end(), new_id);
        return 0;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return -1;
    }
}

// visiting order; group names are returned '\n'-joined in names_buf. Returns #groups or -1.
int64_t pnh_graph_path_order(const void *g, int group_mode, const char *group_file, const char *order_file,
                             const char *subset_file, const char *exclude_file, uint32_t *path_idx,
                             uint32_t *group_id, uint64_t *n_out, char *names_buf, uint64_t names_cap) {
    try {
        pnh::PathOrder o = static_cast<const pnh::GraphStorage *>(g)->path_order(
            (pnh::GroupMode)group_mode, group_file ? group_file : "", order_file ? order_file : "",
            subset_file ? subset_file : "", exclude_file ? exclude_file : "");
        std::copy(o.path_idx.begin(), o.path_idx.end(), path_idx);
        std::copy(o.group_id.begin(), o.group_id.end(), group_id);
        *n_out = o.path_idx.size();
        std::string joined;
        for (size_t i = 0; i < o.groups.size(); ++i) {
            if (i) joined += '\n';
            joined += o.groups[i];
        }
        if (joined.size() + 1 > names_cap) {
            g_host_err = "group name buffer too small";
            return -1;
        }
        std::memcpy(names_buf, joined.c_str(), joined.size() + 1);
        return (int64_t)o.groups.size();
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return -1;
    }
}

// Hist::calc_growth (src/graph_broker/hist.rs:51-66). out: hist_len-1 doubles. Returns n.
int64_t pnh_calc_growth(const uint64_t *hist, uint64_t hist_len, int cov_kind, double cov_val, int quo_kind,
                        double quo_val, unsigned n_threads, double *out) {
    if (!hist || hist_len < 2) return 0;
    std::vector<uint64_t> h(hist, hist + hist_len);
    std::vector<double> g = pnh::calc_growth(h, pnh::Threshold{cov_kind, cov_val}, pnh::Threshold{quo_kind, quo_val}, n_threads);
    std::memcpy(out, g.data(), g.size() * sizeof(double));
    return (int64_t)g.size();
}

// the three branches, callable directly (the reference's unit tests do the same, hist.rs:352-398)
int64_t pnh_calc_growth_branch(int branch, const uint64_t *hist, uint64_t hist_len, int cov_kind, double cov_val,
                               int quo_kind, double quo_val, unsigned n_threads, double *out) {
    if (!hist || hist_len < 2) return 0;
    std::vector<uint64_t> h(hist, hist + hist_len);
    std::vector<double> g;
    pnh::Threshold c{cov_kind, cov_val}, q{quo_kind, quo_val};
    if (branch == 0) g = pnh::calc_growth_union(h, c, n_threads);
    else if (branch == 1) g = pnh::calc_growth_core(h, c, n_threads);
    else g = pnh::calc_growth_quorum(h, c, q, n_threads);
    std::memcpy(out, g.data(), g.size() * sizeof(double));
    return (int64_t)g.size();
}

// several (coverage, quorum) pairs on one histogram: Hist::calc_all_growths (hist.rs:68-87,
// which runs the pairs on the rayon pool).  out: n_pairs x (hist_len-1) doubles, no NaN row.
int64_t pnh_calc_all_growths(const uint64_t *hist, uint64_t hist_len, const int *cov_kind, const double *cov_val,
                             const int *quo_kind, const double *quo_val, uint32_t n_pairs, unsigned n_threads,
                             double *out) {
    if (!hist || hist_len < 2) return 0;
    std::vector<uint64_t> h(hist, hist + hist_len);
    const uint64_t n = hist_len - 1;
    std::vector<pnh::Threshold> cov, quo;
    for (uint32_t t = 0; t < n_pairs; ++t) {
        cov.push_back(pnh::Threshold{cov_kind[t], cov_val[t]});
        quo.push_back(pnh::Threshold{quo_kind[t], quo_val[t]});
    }
    std::vector<std::vector<double>> g = pnh::calc_all_growths(h, cov, quo, n_threads);
    for (uint32_t t = 0; t < n_pairs; ++t) std::memcpy(out + (size_t)t * n, g[t].data(), g[t].size() * sizeof(double));
    return (int64_t)n;
}

// the same in two halves (growth_closed_form.hpp): _begin returns a handle after the device part (if
// any) has been enqueued, _end waits for it, does the host part and releases the handle
void *pnh_calc_all_growths_begin(const uint64_t *hist, uint64_t hist_len, const int *cov_kind, const double *cov_val,
                                 const int *quo_kind, const double *quo_val, uint32_t n_pairs, unsigned n_threads) {
    std::vector<uint64_t> h(hist, hist + (hist ? hist_len : 0));
    std::vector<pnh::Threshold> cov, quo;
    for (uint32_t t = 0; t < n_pairs; ++t) {
        cov.push_back(pnh::Threshold{cov_kind[t], cov_val[t]});
        quo.push_back(pnh::Threshold{quo_kind[t], quo_val[t]});
    }
    if (!hist && hist_len >= 2) return pnh::calc_all_growths_begin_on_device(hist_len - 1, cov, quo);  // the device's own counters
    return pnh::calc_all_growths_begin(h, cov, quo, n_threads);
}
int pnh_growth_tables_begin(uint64_t n_groups, const int *cov_kind, const double *cov_val, const int *quo_kind, const double *quo_val, uint32_t n_pairs) {
    std::vector<pnh::Threshold> cov, quo;
    for (uint32_t t = 0; t < n_pairs; ++t) {
        cov.push_back(pnh::Threshold{cov_kind[t], cov_val[t]});
        quo.push_back(pnh::Threshold{quo_kind[t], quo_val[t]});
    }
    return pnh::growth_tables_begin(n_groups, cov, quo) ? 1 : 0;
}
int64_t pnh_calc_all_growths_end(void *handle, uint64_t n, uint32_t n_pairs, double *out) {
    std::vector<std::vector<double>> g = pnh::calc_all_growths_end(static_cast<pnh::GrowthRun *>(handle));
    for (uint32_t t = 0; t < n_pairs && t < g.size(); ++t)
        std::memcpy(out + (size_t)t * n, g[t].data(), std::min<size_t>(g[t].size(), n) * sizeof(double));
    return (int64_t)n;
}

// One complete `histgrowth` from the RESIDENT steps of a context whose order is set, in one native call: the coverage pass, its
// histogram, the closed-form curves of every (coverage, quorum) pair -- what `panacus histgrowth` computes after the parse
// (hist.rs:68-87 behind graph_broker.rs:353-362), with nothing but library calls in between (a host language between them costs
// tens of microseconds of a sub-millisecond call).  flags: 1 = everything derived from the steps is dropped first, 2 = the
// closed forms' tables too (a cold call).  The curves come from the device when the context is the offload context
// (pnh_set_quorum_offload) and takes n_groups, else from the host threads.  hist_out: n_groups + 1, growth_out: n_pairs x n_groups.
// Returns 0, or the library's error code (pnx_last_error has the text).
int pnh_histgrowth_resident(void *pnx_context, uint64_t n_groups, const int *cov_kind, const double *cov_val, const int *quo_kind,
                            const double *quo_val, uint32_t n_pairs, uint32_t flags, uint64_t *hist_out, double *growth_out) {
    pnx_ctx *ctx = static_cast<pnx_ctx *>(pnx_context);
    if (!ctx || !hist_out || !growth_out || n_groups == 0) return PNX_EINVAL;
    try {
        std::vector<pnh::Threshold> cov, quo;
        for (uint32_t t = 0; t < n_pairs; ++t) {
            cov.push_back(pnh::Threshold{cov_kind[t], cov_val[t]});
            quo.push_back(pnh::Threshold{quo_kind[t], quo_val[t]});
        }
        int rc;
        if ((flags & 1u) && (rc = pnx_config(ctx, PNX_CFG_DROP_DERIVED, 0))) return rc;
        if ((flags & 2u) && (rc = pnx_config(ctx, PNX_CFG_DROP_GROWTH_TABLES, 0))) return rc;
        // the first part of the tables before the pass is enqueued (two small kernels that must not run beside it), the curves behind it
        // (the device curves are those of the OFFLOAD context's last pass: taken only when `ctx` is that context)
        const bool tables = pnh::growth_tables_begin(n_groups, cov, quo, ctx);
        if ((rc = pnx_hist_async(ctx))) return rc;
        pnh::GrowthRun *run = tables ? pnh::calc_all_growths_begin_on_device(n_groups, cov, quo, ctx) : nullptr;
        if ((rc = pnx_hist_fetch(ctx, nullptr, hist_out))) {
            if (run) (void)pnh::calc_all_growths_end(run);
            return rc;
        }
        const std::vector<uint64_t> hist(hist_out, hist_out + n_groups + 1);
        if (!run) run = pnh::calc_all_growths_begin(hist, cov, quo, 0);
        std::vector<std::vector<double>> g = pnh::calc_all_growths_end(run);
        auto complete = [&] {
            if (g.size() != n_pairs) return false;
            for (const auto &c : g)
                if (c.size() != n_groups) return false;
            return true;
        };
        // the device curves failed at the fetch (there is no histogram in that run to fall back on): from the histogram that IS here
        if (!complete()) g = pnh::calc_all_growths_end(pnh::calc_all_growths_begin(hist, cov, quo, 0));
        if (!complete()) {
            g_host_err = "histgrowth: the closed-form curves could not be computed";
            return PNX_EHIP;
        }
        for (uint32_t t = 0; t < n_pairs; ++t) std::memcpy(growth_out + (size_t)t * n_groups, g[t].data(), n_groups * sizeof(double));
        return 0;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return PNX_EHIP;
    }
}

void pnh_set_threads(unsigned n) { pnh::ThreadPool::instance().set_threads(n); }
unsigned pnh_get_threads(void) { return pnh::ThreadPool::instance().size(); }

// run the CLI in-process: argv joined by '\n'. Output / error text are copied into the buffers
// (NUL-terminated, truncated); returns the exit code, *out_len / *err_len get the full lengths.
int pnh_run_cli(const char *argv_joined, char *out_buf, uint64_t out_cap, uint64_t *out_len, char *err_buf,
                uint64_t err_cap, uint64_t *err_len) {
    std::vector<std::string> argv;
    std::string cur;
    for (const char *p = argv_joined; *p; ++p) {
        if (*p == '\n') {
            argv.push_back(cur);
            cur.clear();
        } else {
            cur += *p;
        }
    }
    argv.push_back(cur);
    // PNX_TRACE_CLI=<file>: every in-process command is appended to the file BEFORE it runs (one line, flushed), so that a
    // session that dies inside the native libraries names the command it died in
    if (const char *tr = std::getenv("PNX_TRACE_CLI")) {
        if (FILE *f = std::fopen(tr, "a")) {
            for (size_t i = 0; i < argv.size(); ++i) std::fprintf(f, "%s%s", i ? " " : "", argv[i].c_str());
            std::fputc('\n', f);
            std::fclose(f);
        }
    }
    std::string out, err;
    int rc = pnh::run_cli(argv, out, err);
    auto put = [](const std::string &s, char *buf, uint64_t cap, uint64_t *len) {
        if (len) *len = s.size();
        if (buf && cap) {
            size_t n = std::min<size_t>(s.size(), cap - 1);
            std::memcpy(buf, s.data(), n);
            buf[n] = 0;
        }
    };
    put(out, out_buf, out_cap, out_len);
    put(err, err_buf, err_cap, err_len);
    return rc;
}

// Rust `{}` formatting of an f64 (tables.cpp), for tests
uint64_t pnh_format_f64(double x, char *buf, uint64_t cap) {
    std::string s = pnh::format_f64(x);
    if (cap) {
        size_t n = std::min<size_t>(s.size(), cap - 1);
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size();
}

uint64_t pnh_format_f32(float x, char *buf, uint64_t cap) {
    std::string s = pnh::format_f32(x);
    if (cap) {
        size_t n = std::min<size_t>(s.size(), cap - 1);
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size();
}

// threads of the host pool (caller included) and the CPUs the cgroup quota leaves to this process
uint32_t pnh_pool_threads(void) { return pnh::ThreadPool::instance().size(); }
uint32_t pnh_usable_cpus(void) { return pnh::ThreadPool::usable_cpus(); }

// quorum closed form: inner sums on the GPU of `pnx_context` for n >= min_n (NULL = host only)
void pnh_set_quorum_offload(void *pnx_context, uint64_t min_n) { pnh::set_quorum_offload(pnx_context, min_n); }
int pnh_quorum_offload_usable(void) { return pnh::quorum_offload_usable() ? 1 : 0; }
int pnh_device_growth_usable(void) { return pnh::device_growth_usable() ? 1 : 0; }
// y[k] = log2(x[k]) by the restatement of libm's log2 (csrc/log2_exact.hpp), on the host: what the self-test compares
void pnh_log2_restated(const double *x, double *y, uint64_t n) { pnh::log2_restated(x, y, n); }

double pnh_choose_log2(uint64_t n, uint64_t k) { return pnh::choose_log2(n, k); }

// ---- similarity: dendrogram order of the Jaccard table (linkage.hpp) ----
// perm[k] = input index of the group in row / column k; method as pnh::ClusterMethod (0..6); 0 on success
int pnh_similarity_order(const float *table, uint64_t n, int method, uint64_t *perm) {
    try {
        if (method < 0 || method > 6) throw std::runtime_error("unknown cluster method");
        std::vector<float> t(table, table + n * n);
        std::vector<size_t> p = pnh::similarity_order(t, (size_t)n, (pnh::ClusterMethod)method);
        for (size_t k = 0; k < p.size(); ++k) perm[k] = p[k];
        return 0;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return 1;
    }
}
// kodama::linkage on a condensed matrix: n - 1 steps (c1, c2 as SciPy labels, dissimilarity); 0 on success
int pnh_linkage(const float *condensed, uint64_t n, int method, uint64_t *c1, uint64_t *c2, float *dissimilarity) {
    try {
        if (method < 0 || method > 6) throw std::runtime_error("unknown cluster method");
        std::vector<float> d(condensed, condensed + n * (n ? n - 1 : 0) / 2);
        std::vector<pnh::LinkStep> st = pnh::linkage(d, (size_t)n, (pnh::ClusterMethod)method);
        for (size_t k = 0; k < st.size(); ++k) {
            c1[k] = st[k].c1;
            c2[k] = st[k].c2;
            dissimilarity[k] = st[k].dissimilarity;
        }
        return 0;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return 1;
    }
}

// a float as the report JSON prints it (serde_json / ryu); returns the length
uint64_t pnh_json_f64(double x, char *buf, uint64_t cap) {
    const std::string s = pnh::cli::json_number_f64(x);
    if (cap) {
        const size_t n = std::min<size_t>(s.size(), cap - 1);
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size();
}
uint64_t pnh_json_f32(float x, char *buf, uint64_t cap) {
    const std::string s = pnh::cli::json_number_f32(x);
    if (cap) {
        const size_t n = std::min<size_t>(s.size(), cap - 1);
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size();
}

}  // extern "C"
