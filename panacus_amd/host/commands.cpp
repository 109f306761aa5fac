// commands.cpp -- see commands.hpp.  Plays the role of the reference's GraphBroker
// (src/graph_broker.rs:96-247) for the four hot-path commands: build the graph once, hand the
// ItemTable and the visiting order to the GPU through the C ABI, format the results.
#include "commands.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <stdexcept>

#include "../../include/panacus_amd.h"
#include "gfa_graph.hpp"
#include "growth_closed_form.hpp"
#include "linkage.hpp"
#include "synth_gfa.hpp"
#include "commands_internal.hpp"
#include "tables.hpp"
#include "thread_pool.hpp"

#include <atomic>
#include <condition_variable>
#include <mutex>

namespace pnh {
namespace cli {

const char *USAGE =
    "panacus-amd -- MI355X-native hist / growth / histgrowth / ordered-histgrowth\n"
    "usage: panacus-amd <hist|growth|histgrowth|ordered-histgrowth|similarity|table> [options] <GFA_FILE | HIST.tsv>\n"
    "       panacus-amd report [-j|--json] [-d|--dry-run] <CONFIG.yaml>   run the analyses of a YAML config (one graph\n"
    "                                        upload per run and count type); --json prints the reference's report sections\n"
    "  -c, --count <node|bp|edge|all>   graph quantity to be counted [node]\n"
    "  -l, --coverage <LIST>            coverage thresholds, e.g. 1,2 [1]\n"
    "  -q, --quorum <LIST>              quorum thresholds in [0,1], e.g. 0,0.5 [0]\n"
    "  -a, --hist                       also include the histogram (growth, histgrowth)\n"
    "  -a, --total                      table: one column with the number of groups per item instead of one per group\n"
    "  -g, --groupby <FILE>             path-to-group mapping (2-column TSV)\n"
    "  -H, --groupby-haplotype          merge paths of the same haplotype\n"
    "  -S, --groupby-sample             merge paths of the same sample\n"
    "  -O, --order <FILE>               order of paths/groups (ordered-histgrowth)\n"
    "  -s, --subset <FILE|REGEX>        count only the listed paths/groups or path intervals (BED: 1, 3 or 12 columns;\n"
    "                                   a value that is not a file is a regular expression over the path names)\n"
    "  -e, --exclude <FILE|REGEX>       drop the listed paths/groups/intervals and every node/edge/bp they touch\n"
    "      --cache                      keep / reuse the parsed graph in <GFA_FILE>.pcsr (checked against the\n"
    "                                   GFA's size, mtime and a content hash)\n"
    "  -j, --json                       print the analysis as the reference's report JSON (AnalysisSection list)\n"
    "  -t, --threads <N>                host threads (0 = all) [0]\n"
    "      --device <N>                 GPU ordinal [0]\n"
    "  -m, --method <single|complete|average|weighted|ward|centroid|median>\n"
    "                                   similarity: linkage that orders rows and columns of the table [centroid]\n"
    "       panacus-amd synth --nodes N --paths P [--seed S] [--links] [--sequences] -o FILE.gfa\n"
    "                                        write a pansyn-v1 synthetic pangenome as GFA\n"
    "       panacus-amd synth --shape pggb --nodes N --samples M [--seed S] [--sequences] [--name-prefix s] -o FILE.gfa\n"
    "                                        write a pggb-shaped pangenome (contig paths, inversions, duplications)\n";

struct Device::TextSlot {
    std::mutex mu;
    std::condition_variable cv;
    bool decided = false, uploaded = false;
    const char *data = nullptr;
    size_t size = 0;
    std::shared_ptr<const void> keep;
};

Device::Device(int ordinal, bool expect_text) : text_(std::make_shared<TextSlot>()), ordinal_(ordinal) {
    if (!expect_text) text_->decided = true;
    std::shared_ptr<TextSlot> slot = text_;
    init_ = std::async(std::launch::async, [ordinal, slot]() -> pnx_ctx * {
        pnx_ctx *c = nullptr;
        // (every command of this CLI waits for a histogram before it asks for the next: no overlapping passes)
        const int rc = pnx_init_flags(&c, ordinal, PNX_INIT_ONE_SHOT);
        if (rc != PNX_OK) {
            std::lock_guard<std::mutex> g(slot->mu);
            slot->keep.reset();
            throw std::runtime_error(std::string("GPU initialisation failed: ") + pnx_last_error(nullptr));
        }
        phase_mark("GPU context up");
        // the bytes of the GFA, as soon as the parser has them: copied to HBM beside the parse
        std::unique_lock<std::mutex> lk(slot->mu);
        slot->cv.wait(lk, [&] { return slot->decided; });
        if (slot->data) {
            const char *d = slot->data;
            const size_t n = slot->size;
            lk.unlock();
            const bool ok = pnx_gfa_text_upload(c, d, n) == PNX_OK;  // (a failure only means: the main thread uploads again)
            lk.lock();
            slot->uploaded = ok;
            slot->keep.reset();
            phase_mark("GFA text in HBM");
        }
        return c;
    });
}
void Device::preload(uint32_t what) const {
    if (std::getenv("PANACUS_AMD_NO_PRELOAD")) return;
    (void)pnx_preload(ordinal_, what);
    phase_mark("device code preloaded");
}
void Device::offer_text(const char *data, size_t size, std::shared_ptr<const void> keep) const {
    std::lock_guard<std::mutex> g(text_->mu);
    if (text_->decided) return;
    text_->data = data;
    text_->size = size;
    text_->keep = std::move(keep);
    text_->decided = true;
    text_->cv.notify_all();
}
void Device::no_text() const {
    std::lock_guard<std::mutex> g(text_->mu);
    if (text_->decided) return;
    text_->decided = true;
    text_->cv.notify_all();
}
bool Device::text_uploaded() const {
    (void)ctx();
    std::lock_guard<std::mutex> g(text_->mu);
    const bool there = text_->uploaded;
    text_->uploaded = false;  // one use: pnx_set_csr_gfa frees the library's copy; a second upload of the command passes the text again
    return there;
}
pnx_ctx *Device::ctx() const {
    if (init_.valid()) {
        ctx_ = init_.get();  // throws what the initialisation threw
        // quorum closed form with >= 256 groups: inner sums on this GPU (PANACUS_AMD_HOST_GROWTH=1 keeps the whole closed form on
        // the host threads; the results are the same bits either way).  Bound for the command's THREAD, not for the process: a
        // host that runs commands on several threads gives each its own context.
        if (ctx_ && !std::getenv("PANACUS_AMD_HOST_GROWTH")) bind_thread_offload(ctx_);
    }
    return ctx_;
}
namespace {
std::atomic<bool> g_exit_after_command{false};
}
bool process_exits_after_command() { return g_exit_after_command.load(); }
void set_process_exits_after_command(bool on) { g_exit_after_command.store(on); }
void finish_command(std::unique_ptr<GraphStorage> &g, const Device &dev) {
    phase_mark("table ready");
    if (!process_exits_after_command()) return;
    (void)g.release();
    dev.leak();
}

Device::~Device() {
    no_text();  // (a command that failed before it offered any: the thread must not wait for ever)
    try {
        (void)ctx();
    } catch (...) {
    }
    if (!ctx_ || leaked_) return;
    unbind_thread_offload(ctx_);  // (only if it is still this one: the next run's context may be bound already -- report runner)
    pnx_free(ctx_);
}
void Device::check(int rc) const {
    if (rc != PNX_OK) throw std::runtime_error(pnx_last_error(ctx()));
}

// GraphStorage::from_gfa, or the .pcsr cache next to the GFA when --cache is given
bool wants_device_tokeniser(const Options &o, const std::vector<CountType> &) {
    // (with a -s list the text is not sent ahead: whether the list needs the walks is only known once the paths are)
    return !(o.cache || !o.subset_file.empty() || !o.exclude_file.empty() || std::getenv("PANACUS_AMD_HOST_PARSE"));
}

std::unique_ptr<GraphStorage> load_graph(const Options &o, bool index_edges, const Device *dev, bool links_on_device) {
    struct Decide {  // whatever happens below, the device thread learns that no (further) text is coming
        const Device *d;
        ~Decide() {
            if (d) d->no_text();
        }
    } decide{dev};
    GraphStorage::TextHook hook = nullptr;
    if (dev) hook = [dev](const char *p, size_t n, std::shared_ptr<const void> keep) { dev->offer_text(p, n, std::move(keep)); };
    // subset / exclude lists are applied while walking the path lines: they need the GFA text
    if (!o.cache || !o.subset_file.empty() || !o.exclude_file.empty()) {
        // hist / histgrowth need no edge id on the host: the L lines are found, and parsed on the device (or here, later, if
        // the graph turns out not to be one the device tokenises)
        const bool links_only = index_edges && links_on_device && dev && wants_device_tokeniser(o, {});
        return GraphStorage::from_gfa(o.file, index_edges && !links_only, false, hook, links_only);
    }
    const std::string cache_file = o.file + ".pcsr";
    if (auto g = GraphStorage::from_cache(cache_file, o.file, index_edges)) return g;
    auto g = GraphStorage::from_gfa(o.file, index_edges);
    g->save_cache(cache_file, o.file);
    return g;
}

std::vector<CountType> count_types(const std::string &c, bool allow_all) {
    std::string l;
    for (char ch : c) l += (char)std::tolower((unsigned char)ch);
    if (l == "all") {
        if (!allow_all) throw std::runtime_error("count type 'all' is not admissible here");
        return {COUNT_NODE, COUNT_BP, COUNT_EDGE};
    }
    CountType t;
    if (!parse_count_name(l, t)) throw std::runtime_error("invalid value '" + c + "' for '--count <count>'");
    return {t};
}

// what pnx_set_csr_gfa / pnx_gfa_walks are told about a graph whose steps the device tokenises: the step columns, how a
// segment name becomes a node id (a table indexed by the number, or -- names that are not numbers -- the byte range that
// holds the S lines: the device finds them and hashes their name fields), and -- edge counts -- the edges: the host's index of the L lines if it was built, else
// the byte range that holds the L lines (the device finds, parses and ranks them; no edge map on the host at all)
struct GfaStepArgs {
    std::vector<uint64_t> cb, ce, euv;
    std::vector<uint8_t> wk, eoo;
    pnx_gfa_steps st{};
    bool links = false;
    GfaStepArgs(const GraphStorage &g, bool text_there) {
        g.step_columns(cb, ce, wk);
        st.text = text_there ? nullptr : g.text_data();
        st.text_bytes = text_there ? 0 : g.text_size();
        st.n_paths = (uint32_t)cb.size();
        st.n_nodes = (uint32_t)g.node_count();
        if (g.names_by_bytes_on_device()) {  // the device finds the S lines and hashes their name fields itself
            st.n_names = PNX_NAMES_FIND;
            g.segment_range(st.name_lo, st.name_hi);
        } else {
            st.id_of_name = g.id_of_name().empty() ? nullptr : g.id_of_name().data();
            st.n_names = g.id_of_name().size();
            st.name_prefix_len = (uint32_t)g.name_prefix().size();  // (`s12`: at most 8 bytes in front of the number)
            std::memcpy(st.name_prefix, g.name_prefix().data(), g.name_prefix().size());
        }
    }
    void columns() {  // (after the caller emptied the columns it does not want)
        st.col_begin = cb.data();
        st.col_end = ce.data();
        st.is_walk = wk.data();
    }
    void edges(const GraphStorage &g, bool into_steps) {
        links = g.links_for_device();
        if (links) {
            st.n_links = PNX_LINKS_FIND;
            g.link_range(st.link_lo, st.link_hi);
            if (!st.link_hi) st.link_lo = st.link_hi = 1;  // (no L line was seen: an empty range, not "the whole text")
        } else {
            g.edge_ends(euv, eoo);
            if (into_steps) {
                st.edge_uv = euv.data();
                st.edge_oo = eoo.data();
                st.n_edges = (uint32_t)g.number_of_items(COUNT_EDGE);
            }
        }
    }
};

// upload graph + order for one count type

// Returns the uncovered bp of partially covered nodes (bp counts under a subset list, else empty).
// growth_weights: upload node_len - uncovered as the bp weight of such nodes, which is what
// AbacusByGroup::calc_growth adds (abacus.rs:1013-1023); the histogram instead takes plain node
// lengths and is corrected afterwards (construct_hist_bps, abacus.rs:779-785).
// Edge tables go up with one sort key per edge (its canonical ends): the reference numbers edges in the
// order of the L lines, and the library renumbers them internally on the device when that order does
// not follow the paths (pnx_set_csr_keyed); every per-item result still comes back in the reference's
// ids, so `table -c edge` needs no special case.
// -s / -e lists: the walks are cut ON THE DEVICE (pnx_set_csr_cut; graph_broker/util.rs:412-795) and stay
// there as the resident graph; the host replays the few partial pieces (bp counts) and hands the late
// exclusion flags and -- for growth -- the weights of partly covered nodes back to the library.
Uncovered upload_cut(const std::function<pnx_ctx *()> &get_ctx, const GraphStorage &g, CountType ct, const Masking &mk,
                     bool growth_weights) {
    pnx_ctx *ctx = nullptr;  // asked for after the host's share of the work (the GPU may still be coming up beside it)
    auto check = [&](int rc) {
        if (rc != PNX_OK) throw std::runtime_error(pnx_last_error(ctx));
    };
    const uint64_t n_items = g.number_of_items(ct);
    // segment names the device can resolve (numbers, or at most 16 bytes): the walks are made from the text on the device
    // (pnx_gfa_walks) and cut where they are; the paths no interval touches get an empty step column.  Otherwise the host's step parser makes them and they go up.
    const bool dev_walks = g.steps_tokenisable_on_device() && !g.from_cache_file() && !std::getenv("PANACUS_AMD_HOST_PARSE");
    WalkCut cut = g.walk_cut(ct, mk.mode, mk.group_file, mk.subset_file, mk.exclude_file, !dev_walks);
    const uint32_t none32 = 0;
    const uint8_t none8 = 0;
    const uint64_t none64 = 0;
    pnx_walks w{};
    if (dev_walks) {
        GfaStepArgs a(g, false);
        for (size_t k = 0; k < cut.path_mode.size(); ++k)
            if (cut.path_mode[k] == PNX_WALK_SKIP) a.ce[k] = a.cb[k];
        a.columns();
        pnx_gfa_steps &st = a.st;
        cut.walk_off.assign(cut.path_mode.size() + 1, 0);
        ctx = get_ctx();
        check(pnx_gfa_walks(ctx, &st, cut.walk_off.data()));
        phase_mark("pnx_gfa_walks (walks tokenised on the device)");
        w.walk_node = nullptr;
        w.walk_backward = nullptr;
    } else {
        w.walk_node = cut.walk_node.empty() ? &none32 : cut.walk_node.data();
        w.walk_backward = cut.walk_backward.empty() ? nullptr : cut.walk_backward.data();
    }
    w.walk_off = cut.walk_off.data();
    w.path_start = cut.path_start.empty() ? &none64 : cut.path_start.data();
    w.path_mode = cut.path_mode.empty() ? &none8 : cut.path_mode.data();
    w.n_paths = (uint32_t)cut.path_mode.size();
    w.n_nodes = (uint32_t)g.node_count();
    w.node_len = g.node_lens().data();
    if (ct == COUNT_EDGE) {  // the library looks the edge of every step pair up itself and ranks the edges by their ends
        w.edge_uv = cut.edge_uv.data();
        w.edge_oo = cut.edge_oo.data();
    }
    w.n_items = (uint32_t)n_items;
    w.count_type = (int)ct;
    w.track_covered = cut.track_covered ? 1 : 0;
    w.inc_off = cut.inc_off.data();
    w.inc_iv = cut.inc_iv.empty() ? &none64 : cut.inc_iv.data();
    if (!cut.exc_off.empty()) {
        w.exc_off = cut.exc_off.data();
        w.exc_iv = cut.exc_iv.empty() ? &none64 : cut.exc_iv.data();
    }
    static_assert(sizeof(PieceEvent) == sizeof(pnx_piece_event), "PieceEvent mirrors pnx_piece_event");
    std::vector<PieceEvent> events(cut.max_events);
    uint64_t n_events = 0;
    ctx = get_ctx();
    check(pnx_set_csr_cut(ctx, &w, ct == COUNT_BP ? g.node_lens().data() : nullptr, nullptr,
                          reinterpret_cast<pnx_piece_event *>(events.data()), events.size(), &n_events));
    events.resize(n_events);
    Uncovered uncovered;
    std::vector<uint32_t> late_flags;
    g.replay_piece_events(cut, std::move(events), uncovered, late_flags);
    if (!late_flags.empty()) check(pnx_exclude_items(ctx, late_flags.data(), (uint32_t)late_flags.size()));
    if (ct == COUNT_BP && growth_weights && !uncovered.empty()) {
        std::vector<uint32_t> wts = g.node_lens();
        for (const auto &u : uncovered) wts[u.first] = u.second > wts[u.first] ? 0 : (uint32_t)(wts[u.first] - u.second);
        check(pnx_set_weights(ctx, wts.data()));
    }
    return uncovered;
}

Uncovered upload(const Device &dev, const GraphStorage &g, CountType ct, const PathOrder &order, const Masking &mk,
                 bool growth_weights, bool /*per_item_output*/) {
    const uint32_t n_paths = (uint32_t)g.path_segments().size();
    Uncovered uncovered;
    // -s / -e lists: the walks are cut on the device.  Edge counts take the same entry even without lists (every path
    // "cut" by the whole-path interval): the node walks go up and the library finds the edge of every step pair in a hash
    // table in HBM, instead of one edge2id lookup per step on the host (a graph from the .pcsr cache has its edge table)
    // (edge counts on a graph with zero-length nodes: the reference drops the edges among the leading zero-length nodes of a
    // path, GraphStorage::has_zero_length_nodes -- the cut route below reproduces that, the device's plain edge route does not)
    const bool on_device = g.steps_tokenisable_on_device() && !g.from_cache_file() && !std::getenv("PANACUS_AMD_HOST_PARSE") &&
                           !(ct == COUNT_EDGE && g.has_zero_length_nodes());
    // a -s list that only picks whole paths (names of paths, samples, haplotypes; intervals that contain a path) needs no walk:
    // the paths it leaves out get an empty step column
    std::vector<uint8_t> take;
    const bool path_level = mk.any() && on_device && g.mask_is_path_level(ct, mk.mode, mk.group_file, mk.subset_file, mk.exclude_file, take);
    if ((mk.any() && !path_level) || (ct == COUNT_EDGE && !g.from_cache_file() && !on_device)) {
        uncovered = upload_cut([&dev]() { return dev.ctx(); }, g, ct, mk, growth_weights);
    } else if (on_device) {
        // the ItemTable is made from the raw text ON THE DEVICE (pnx_set_csr_gfa) -- no step is parsed on the host, no
        // ItemTable crosses PCIe; the text is in HBM already if the device thread was offered it.  Edge counts: the walks stay
        // on the device too; the L lines are parsed and ranked there as well (hist / histgrowth: GfaStepArgs::edges), or the
        // host hands over its index of them, and the library looks the edge of every step pair up itself
        const bool there = dev.text_uploaded();
        GfaStepArgs a(g, there);
        if (path_level)
            for (size_t k = 0; k < take.size(); ++k)
                if (!take[k]) a.ce[k] = a.cb[k];
        a.columns();
        if (ct == COUNT_EDGE) a.edges(g, true);
        pnx_gfa_steps &st = a.st;
        phase_mark(there ? "columns ready (text was uploaded beside the parse)" : "columns ready");
        dev.check(pnx_set_csr_gfa(dev.ctx(), &st, ct == COUNT_BP ? g.node_lens().data() : nullptr, nullptr));
        phase_mark("pnx_set_csr_gfa (tokenise + rows)");
    } else {
        const uint64_t n_items = g.number_of_items(ct);  // (asked here: the device routes above never need the host's edge index)
        ItemTable tab;
        const ItemTableView view = g.item_table_view(ct, tab);
        std::vector<uint64_t> keys;  // a cached edge table: ranked by the canonical ends of its edges (pnx_set_csr_keyed)
        if (ct == COUNT_EDGE && n_items > 0) keys = g.edge_keys();
        const uint64_t *key_ptr = keys.empty() ? nullptr : keys.data();
        phase_mark("item table ready");
        pnx_ctx *c = dev.ctx();
        phase_mark("wait for the GPU context");
        dev.check(pnx_set_csr_keyed(c, view.items, view.id_prefsum, n_paths, (uint32_t)n_items,
                                    ct == COUNT_BP ? g.node_lens().data() : nullptr, nullptr, key_ptr));
        phase_mark("pnx_set_csr (H2D + rows)");
    }
    dev.check(pnx_set_order(dev.ctx(), order.path_idx.data(), order.group_id.data(), (uint32_t)order.path_idx.size(),
                            (uint32_t)order.groups.size()));
    return uncovered;
}

std::vector<uint64_t> device_hist(const Device &dev, const GraphStorage &g, CountType ct, const PathOrder &order,
                                  const Masking &mk) {
    const Uncovered uncovered = upload(dev, g, ct, order, mk);
    std::vector<uint64_t> hist(order.groups.size() + 1, 0);
    if (uncovered.empty()) {
        dev.check(pnx_hist(dev.ctx(), nullptr, hist.data()));
        phase_mark("pnx_set_order + pnx_hist");
        return hist;
    }
    // "subtract uncovered bps", abacus.rs:779-785: the bp a subset interval leaves out of a node
    // move from the node's coverage bin to bin 0 (usize arithmetic of a release build)
    std::vector<uint32_t> countable(g.number_of_items(ct) + 1, 0);
    dev.check(pnx_hist(dev.ctx(), countable.data(), hist.data()));
    for (const auto &u : uncovered) {
        hist[countable[u.first]] -= u.second;
        hist[0] += u.second;
    }
    return hist;
}

// histograms for several count types; node and bp share one resident CSR (the reference clones
// the item table for them too, util.rs:201-204)
std::vector<std::vector<uint64_t>> device_hists(const Device &dev, const GraphStorage &g, const std::vector<CountType> &cts,
                                                const PathOrder &order, const Masking &mk) {
    std::vector<std::vector<uint64_t>> out(cts.size());
    bool have_node = false, have_bp = false;
    for (CountType c : cts) {
        have_node = have_node || c == COUNT_NODE;
        have_bp = have_bp || c == COUNT_BP;
    }
    bool have_edge = false;
    for (CountType c : cts) have_edge = have_edge || c == COUNT_EDGE;
    const bool on_device = g.steps_tokenisable_on_device() && !g.from_cache_file() && !std::getenv("PANACUS_AMD_HOST_PARSE") &&
                           !(have_edge && g.has_zero_length_nodes());
    if (on_device && !mk.any() && have_edge && (have_node || have_bp)) {
        // `-c all`: the text is tokenised ONCE into walks that stay on the device
        // (pnx_gfa_walks), and every count type's table is made from them there (pnx_set_csr_walks) -- the reference builds
        // one table for node + bp and parses the file again for the edges (graph_broker/util.rs:201-204, graph_broker.rs:404-422)
        std::vector<uint64_t> walk_off(g.path_segments().size() + 1, 0);
        GfaStepArgs a(g, dev.text_uploaded());
        a.columns();
        a.edges(g, false);  // (the L lines go with the walks when the device parses them)
        pnx_gfa_steps &st = a.st;
        dev.check(pnx_gfa_walks(dev.ctx(), &st, walk_off.data()));
        phase_mark("pnx_gfa_walks (tokenised once for all count types)");
        auto set_order = [&]() {
            dev.check(pnx_set_order(dev.ctx(), order.path_idx.data(), order.group_id.data(), (uint32_t)order.path_idx.size(),
                                    (uint32_t)order.groups.size()));
        };
        if (have_node || have_bp) {
            dev.check(pnx_set_csr_walks(dev.ctx(), (uint32_t)g.node_count(), have_bp ? g.node_lens().data() : nullptr, nullptr, nullptr, nullptr, 0));
            set_order();
            for (size_t k = 0; k < cts.size(); ++k) {
                if (cts[k] == COUNT_EDGE) continue;
                if (have_bp) dev.check(pnx_config(dev.ctx(), PNX_CFG_USE_WEIGHTS, cts[k] == COUNT_BP ? 1 : 0));
                out[k].assign(order.groups.size() + 1, 0);
                dev.check(pnx_hist(dev.ctx(), nullptr, out[k].data()));
            }
            phase_mark("node / bp tables from the resident walks + hists");
        }
        if (a.links)
            dev.check(pnx_set_csr_walks(dev.ctx(), (uint32_t)g.node_count(), nullptr, nullptr, nullptr, nullptr, PNX_EDGES_FROM_LINKS));
        else
            dev.check(pnx_set_csr_walks(dev.ctx(), (uint32_t)g.node_count(), nullptr, nullptr, a.euv.data(), a.eoo.data(), (uint32_t)g.number_of_items(COUNT_EDGE)));
        set_order();
        for (size_t k = 0; k < cts.size(); ++k) {
            if (cts[k] != COUNT_EDGE) continue;
            out[k].assign(order.groups.size() + 1, 0);
            dev.check(pnx_hist(dev.ctx(), nullptr, out[k].data()));
        }
        phase_mark("edge table from the resident walks + hist");
        return out;
    }
    if (have_node && have_bp && !mk.any()) {  // with -s/-e lists the two count types exclude differently
        upload(dev, g, COUNT_BP, order, mk);
        for (size_t k = 0; k < cts.size(); ++k) {
            if (cts[k] == COUNT_EDGE) continue;
            dev.check(pnx_config(dev.ctx(), PNX_CFG_USE_WEIGHTS, cts[k] == COUNT_BP ? 1 : 0));
            out[k].assign(order.groups.size() + 1, 0);
            dev.check(pnx_hist(dev.ctx(), nullptr, out[k].data()));
        }
    }
    for (size_t k = 0; k < cts.size(); ++k)
        if (out[k].empty()) out[k] = device_hist(dev, g, cts[k], order, mk);
    return out;
}

std::vector<double> to_f64(const std::vector<uint64_t> &v) {
    std::vector<double> o(v.size());
    for (size_t i = 0; i < v.size(); ++i) o[i] = (double)v[i];
    return o;
}

// Hist::calc_all_growths (hist.rs:68-87): NaN row 0 + one curve per threshold pair
std::vector<std::vector<double>> all_growths(const std::vector<uint64_t> &hist, const ThresholdContainer &tc, unsigned threads) {
    std::vector<std::vector<double>> out = calc_all_growths(hist, tc.coverage, tc.quorum, threads);
    for (auto &g : out) g.insert(g.begin(), std::numeric_limits<double>::quiet_NaN());
    return out;
}

void growth_headers(std::vector<std::vector<std::string>> &headers, const char *what, CountType ct, const ThresholdContainer &tc) {
    for (size_t t = 0; t < tc.coverage.size(); ++t)
        headers.push_back({what, count_name(ct), threshold_string(tc.coverage[t]), threshold_string(tc.quorum[t])});
}

Masking masking(const Options &o) {
    Masking m;
    m.mode = group_mode(o);
    m.group_file = o.group_file;
    m.subset_file = o.subset_file;
    m.exclude_file = o.exclude_file;
    return m;
}

GroupMode group_mode(const Options &o) {
    if (o.by_haplotype) return GROUP_HAPLOTYPE;  // load_groups checks haplotype first (abacus.rs:248)
    if (o.by_sample) return GROUP_SAMPLE;
    if (!o.group_file.empty()) return GROUP_FILE;
    return GROUP_PATHID;
}

std::string cmd_hist(const Options &o, const std::string &cmdline) {
    std::vector<CountType> cts = count_types(o.count, true);
    bool edges = false;
    for (CountType c : cts) edges = edges || c == COUNT_EDGE;
    const Device dev(o.device, wants_device_tokeniser(o, cts));  // the GPU comes up (and takes the text) while the graph is read
    auto g = load_graph(o, edges, &dev, true);
    PathOrder order = g->path_order(group_mode(o), o.group_file, "", o.subset_file, o.exclude_file);
    dev.preload(PNX_PRELOAD_PASS | (g->steps_tokenisable_on_device() ? PNX_PRELOAD_GFA : 0u) | (edges ? PNX_PRELOAD_LINKS : 0u));
    std::vector<std::vector<std::string>> headers = {{"panacus", "count", "", ""}};
    std::vector<std::vector<double>> cols;
    std::vector<std::vector<uint64_t>> hists = device_hists(dev, *g, cts, order, masking(o));
    for (size_t k = 0; k < cts.size(); ++k) {
        cols.push_back(to_f64(hists[k]));
        headers.push_back({"hist", count_name(cts[k]), "", ""});
    }
    std::string res = metadata_comments(cmdline) + write_table(headers, cols);
    finish_command(g, dev);
    return res;
}

// histgrowth (and growth on a GFA, which the reference restricts to node counts)
std::string cmd_histgrowth(const Options &o, const std::string &cmdline, bool growth_cmd) {
    ThresholdContainer tc = ThresholdContainer::parse_params(o.quorum, o.coverage);
    std::vector<CountType> cts = growth_cmd ? std::vector<CountType>{COUNT_NODE} : count_types(o.count, true);
    bool edges = false;
    for (CountType c : cts) edges = edges || c == COUNT_EDGE;
    const Device dev(o.device, wants_device_tokeniser(o, cts));  // the GPU comes up (and takes the text) while the graph is read
    auto g = load_graph(o, edges, &dev, true);
    PathOrder order = g->path_order(group_mode(o), o.group_file, "", o.subset_file, o.exclude_file);
    dev.preload(PNX_PRELOAD_PASS | (g->steps_tokenisable_on_device() ? PNX_PRELOAD_GFA : 0u) | (edges ? PNX_PRELOAD_LINKS : 0u));
    std::vector<std::vector<uint64_t>> hists = device_hists(dev, *g, cts, order, masking(o));
    std::vector<std::vector<std::string>> headers = {{"panacus", "count", "coverage", "quorum"}};
    std::vector<std::vector<double>> cols;
    if (o.add_hist)
        for (size_t k = 0; k < cts.size(); ++k) {
            cols.push_back(to_f64(hists[k]));
            headers.push_back({"hist", count_name(cts[k]), "", ""});
        }
    for (size_t k = 0; k < cts.size(); ++k) {
        for (auto &col : all_growths(hists[k], tc, (unsigned)o.threads)) cols.push_back(std::move(col));
        growth_headers(headers, "growth", cts[k], tc);
    }
    std::string res = "# " + cmdline + "\n" + write_table(headers, cols);
    finish_command(g, dev);
    return res;
}

// growth from a hist TSV (src/lib.rs:160-190, analyses/growth.rs:190-262): no GPU involved
std::string cmd_growth_from_hist(const Options &o, const std::string &cmdline) {
    if (o.by_sample || o.by_haplotype || !o.group_file.empty() || !o.subset_file.empty() || !o.exclude_file.empty())
        throw std::runtime_error("subset, exclude and groupby can only be used in graph mode (with a .gfa or .gfa.gz file)");
    ThresholdContainer tc = ThresholdContainer::parse_params(o.quorum, o.coverage);
    ParsedHists ph = parse_hists(o.file);
    std::string res;
    for (const auto &c : ph.comments) res += c + "\n";
    res += "# " + cmdline + "\n";
    std::vector<std::vector<std::string>> headers = {{"panacus", "count", "coverage", "quorum"}};
    std::vector<std::vector<double>> cols;
    if (o.add_hist)
        for (const auto &h : ph.hists) {
            cols.push_back(to_f64(h.second));
            headers.push_back({"hist", count_name(h.first), "", ""});
        }
    for (const auto &h : ph.hists) {
        for (auto &col : all_growths(h.second, tc, (unsigned)o.threads)) cols.push_back(std::move(col));
        growth_headers(headers, "growth", h.first, tc);
    }
    return res + write_table(headers, cols);
}

std::vector<std::vector<double>> device_ordered_growth(const Device &dev, const ThresholdContainer &tc, uint32_t G) {
    const uint32_t T = (uint32_t)tc.coverage.size();
    // AbacusByGroup::calc_growth prologue (abacus.rs:997-998, 1009) in f64 on the host
    std::vector<uint32_t> cov(T), qtab((size_t)T * G);
    for (uint32_t t = 0; t < T; ++t) {
        cov[t] = (uint32_t)std::max<uint64_t>(1, tc.coverage[t].to_absolute(G));
        const double q = G ? std::max(0.0, tc.quorum[t].to_relative(G)) : 0.0;
        for (uint32_t r = 0; r < G; ++r) qtab[(size_t)t * G + r] = (uint32_t)std::ceil(((double)r + 1.0) * q);
    }
    std::vector<uint64_t> res((size_t)T * G, 0);
    if (G) dev.check(pnx_ordered_growth(dev.ctx(), nullptr, 1, cov.data(), qtab.data(), T, res.data()));
    std::vector<std::vector<double>> out(T, std::vector<double>(G));
    for (uint32_t t = 0; t < T; ++t)
        for (uint32_t j = 0; j < G; ++j) out[t][j] = (double)res[(size_t)t * G + j];
    return out;
}

std::string cmd_ordered(const Options &o, const std::string &cmdline) {
    ThresholdContainer tc = ThresholdContainer::parse_params(o.quorum, o.coverage);
    CountType ct = count_types(o.count, false)[0];
    const Device dev(o.device, wants_device_tokeniser(o, {ct}));  // the GPU comes up (and takes the text) while the graph is read
    auto g = load_graph(o, ct == COUNT_EDGE, &dev, true);
    PathOrder order = g->path_order(group_mode(o), o.group_file, o.order_file, o.subset_file, o.exclude_file);
    dev.preload(PNX_PRELOAD_PASS | PNX_PRELOAD_GROWTH | (g->steps_tokenisable_on_device() ? PNX_PRELOAD_GFA : 0u) | (ct == COUNT_EDGE ? PNX_PRELOAD_LINKS : 0u));
    const uint32_t G = (uint32_t)order.groups.size();
    upload(dev, *g, ct, order, masking(o), true);
    std::vector<std::vector<std::string>> headers = {{"panacus", "count", "coverage", "quorum"}};
    std::vector<std::vector<double>> cols;
    for (auto &res : device_ordered_growth(dev, tc, G)) {
        std::vector<double> col(G + 1);
        col[0] = std::numeric_limits<double>::quiet_NaN();
        for (uint32_t j = 0; j < G; ++j) col[j + 1] = res[j];
        cols.push_back(std::move(col));
    }
    growth_headers(headers, "ordered-growth", ct, tc);
    return metadata_comments(cmdline) + write_ordered_table(headers, cols, order.groups);
}

// Similarity::set_table (similarity.rs:119-190) + get_table_string (:224-239).  The intersections
// come from the GPU; the f32 division, the Euclidean row distances, the linkage (-m, default
// centroid) and the reordering of rows, columns and labels happen here exactly as in the
// reference (linkage.hpp).
SimilarityResult device_similarity(const Device &dev, const std::vector<std::string> &groups, const std::string &method_name) {
    ClusterMethod method;
    if (!parse_cluster_method(method_name, method))
        throw std::runtime_error("invalid value '" + method_name + "' for --method (single, complete, average, weighted, ward, centroid, median)");
    const size_t G = groups.size();
    std::vector<uint64_t> inter(G * G, 0);
    if (G) dev.check(pnx_group_intersections(dev.ctx(), inter.data()));
    for (size_t a = 0; a < G; ++a)
        if (inter[a * G + a] == 0)  // path_lens[&a] on a missing key panics in the reference (:163)
            throw std::runtime_error("group " + groups[a] + " covers no item: the reference panics here");
    SimilarityResult r;
    r.table.resize(G * G);
    for (size_t i = 0; i < G; ++i)
        for (size_t j = 0; j < G; ++j) {
            const uint64_t x = inter[i * G + j];
            r.table[i * G + j] = (float)x / (float)(inter[i * G + i] + inter[j * G + j] - x);
        }
    r.perm = similarity_order(r.table, G, method);
    return r;
}

std::string similarity_table_string(const SimilarityResult &r, const std::vector<std::string> &groups) {
    const size_t G = groups.size();
    std::string res = "group";
    for (size_t k = 0; k < G; ++k) res += "\t" + groups[r.perm[k]];
    res += "\n";
    for (size_t i = 0; i < G; ++i) {
        res += groups[r.perm[i]];
        for (size_t j = 0; j < G; ++j) res += "\t" + format_f32(r.table[r.perm[i] * G + r.perm[j]]);
        res += "\n";
    }
    return res;
}

std::string cmd_similarity(const Options &o, const std::string &cmdline) {
    CountType ct = count_types(o.count, false)[0];
    ClusterMethod method;
    if (!parse_cluster_method(o.method, method))
        throw std::runtime_error("invalid value '" + o.method + "' for --method (single, complete, average, weighted, ward, centroid, median)");
    const Device dev(o.device, wants_device_tokeniser(o, {ct}));  // the GPU comes up (and takes the text) while the graph is read
    auto g = load_graph(o, ct == COUNT_EDGE, &dev, true);
    PathOrder order = g->path_order(group_mode(o), o.group_file, o.order_file, o.subset_file, o.exclude_file);
    dev.preload(PNX_PRELOAD_PASS | PNX_PRELOAD_PAIRS | (g->steps_tokenisable_on_device() ? PNX_PRELOAD_GFA : 0u) | (ct == COUNT_EDGE ? PNX_PRELOAD_LINKS : 0u));
    upload(dev, *g, ct, order, masking(o));
    return metadata_comments(cmdline) + similarity_table_string(device_similarity(dev, order.groups, o.method), order.groups);
}

// AbacusByGroup::to_tsv (abacus.rs:1056-1178): one row per item.  With `total` the number of groups
// holding it = the coverage vector of pnx_hist; otherwise one column per group with the number of
// steps the group's paths take on the item (AbacusByGroup.v, from pnx_group_visit_counts) times the
// item's bp (node_len - uncovered for bp counts, else 1).
std::string cmd_table(const Options &o, const std::string &cmdline) {
    CountType ct = count_types(o.count, false)[0];
    const Device dev(o.device, wants_device_tokeniser(o, {ct}));  // the GPU comes up (and takes the text) while the graph is read
    auto g = load_graph(o, ct == COUNT_EDGE, &dev, false);
    PathOrder order = g->path_order(group_mode(o), o.group_file, o.order_file, o.subset_file, o.exclude_file);
    dev.preload(PNX_PRELOAD_PASS | PNX_PRELOAD_TABLES | (g->steps_tokenisable_on_device() ? PNX_PRELOAD_GFA : 0u) | (ct == COUNT_EDGE ? PNX_PRELOAD_LINKS : 0u));
    const uint64_t n = g->number_of_items(ct);
    const size_t G = order.groups.size();
    const Uncovered uncovered = upload(dev, *g, ct, order, masking(o), false, true);
    std::string res = metadata_comments(cmdline);
    res += ct == COUNT_EDGE ? "edge" : "node";
    std::vector<std::string> labels;
    if (ct == COUNT_EDGE) labels = g->edge_labels();
    auto label = [&](uint64_t i) { return ct == COUNT_EDGE ? labels[i] : g->node_name((uint32_t)i); };
    // rows are formatted in chunks of 16 K items on the worker pool (3.76 M rows of `table --total` on the chr22 shape were
    // 0.33 s of string concatenation on one thread) and appended in order
    auto put_uint = [](std::string &to, uint64_t v) {
        char buf[24];
        int k = 24;
        do {
            buf[--k] = (char)('0' + v % 10);
            v /= 10;
        } while (v);
        to.append(buf + k, (size_t)(24 - k));
    };
    auto append_rows = [&](uint64_t lo, uint64_t hi, const std::function<void(std::string &, uint64_t)> &row) {
        const uint64_t CH = 1u << 14, nch = (hi - lo + CH - 1) / CH;
        std::vector<std::string> parts(nch);
        ThreadPool::instance().parallel_for(nch, [&](size_t c) {
            std::string &out = parts[c];
            const uint64_t a = lo + c * CH, b = std::min(hi, a + CH);
            out.reserve((size_t)(b - a) * 16);
            for (uint64_t i = a; i < b; ++i) row(out, i);
        });
        size_t total = res.size();
        for (const auto &x : parts) total += x.size();
        res.reserve(total);
        for (const auto &x : parts) res += x;
    };
    if (o.total) {
        res += "\ttotal\n";
        std::vector<uint32_t> countable(n + 1, 0);
        std::vector<uint64_t> hist(G + 1, 0);
        dev.check(pnx_hist(dev.ctx(), countable.data(), hist.data()));
        append_rows(1, n + 1, [&](std::string &out, uint64_t i) {
            out += label(i);
            out += '\t';
            put_uint(out, countable[i]);
            out += '\n';
        });
        return res;
    }
    for (const auto &name : order.groups) res += "\t" + name;
    res += "\n";
    std::vector<uint64_t> bp;  // per item: the factor of the node/bp branch (abacus.rs:1087-1092)
    if (ct == COUNT_BP) {
        bp.assign(g->node_lens().begin(), g->node_lens().end());
        for (const auto &u : uncovered) bp[u.first] -= u.second;  // usize arithmetic, like the reference
    }
    // item slices that keep the G x slice counter block around 1 GiB
    uint64_t slice = std::max<uint64_t>(1, std::min<uint64_t>(n, (256ull << 20) / std::max<size_t>(G, 1)));
    if (const char *e = std::getenv("PANACUS_AMD_TABLE_SLICE"))  // test hook: items per slice
        if (std::atoll(e) > 0) slice = (uint64_t)std::atoll(e);
    std::vector<uint32_t> counts;
    // the edge branch prints v[j] -- the j-th slot of the flat value array, j = the GROUP id
    // (abacus.rs:1162) -- instead of the slot of (edge, group): the first G slots in (item, group) order
    std::vector<uint32_t> first_slots;
    if (ct == COUNT_EDGE && G) {
        for (uint64_t lo = 1; lo <= n && first_slots.size() < G; lo += slice) {
            const uint64_t hi = std::min(n + 1, lo + slice);
            counts.assign(G * (hi - lo), 0);
            dev.check(pnx_group_visit_counts(dev.ctx(), (uint32_t)lo, (uint32_t)hi, counts.data()));
            for (uint64_t i = lo; i < hi && first_slots.size() < G; ++i)
                for (size_t j = 0; j < G && first_slots.size() < G; ++j)
                    if (counts[j * (hi - lo) + (i - lo)]) first_slots.push_back(counts[j * (hi - lo) + (i - lo)]);
        }
    }
    for (uint64_t lo = 1; lo <= n; lo += slice) {
        const uint64_t hi = std::min(n + 1, lo + slice);
        counts.assign(G * (hi - lo), 0);
        if (G) dev.check(pnx_group_visit_counts(dev.ctx(), (uint32_t)lo, (uint32_t)hi, counts.data()));
        append_rows(lo, hi, [&](std::string &out, uint64_t i) {
            out += label(i);
            for (size_t j = 0; j < G; ++j) {
                const uint32_t v = counts[j * (hi - lo) + (i - lo)];
                out += '\t';
                if (!v) {
                    out += '0';
                } else if (ct == COUNT_EDGE) {
                    if (j >= first_slots.size())
                        throw std::runtime_error("table -c edge: the reference indexes v by group id here and runs past its end (panic)");
                    put_uint(out, first_slots[j]);
                } else {
                    put_uint(out, (uint64_t)v * (ct == COUNT_BP ? bp[i] : 1));
                }
            }
            out += '\n';
        });
    }
    return res;
}

bool ends_with(const std::string &s, const std::string &suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

}  // namespace cli

using namespace cli;

int run_cli(const std::vector<std::string> &argv, std::string &out, std::string &err) {
    std::string cmdline;
    for (size_t i = 0; i < argv.size(); ++i) cmdline += (i ? " " : "") + argv[i];
    if (argv.size() < 2 || argv[1] == "-h" || argv[1] == "--help") {
        out = USAGE;
        return argv.size() < 2 ? 2 : 0;
    }
    Options o;
    o.cmd = argv[1];
    if (const char *d = std::getenv("PANACUS_AMD_DEVICE")) o.device = std::atoi(d);
    try {
        for (size_t i = 2; i < argv.size(); ++i) {
            const std::string &a = argv[i];
            auto value = [&](const char *name) -> std::string {
                if (i + 1 >= argv.size()) throw std::runtime_error(std::string("option ") + name + " needs a value");
                return argv[++i];
            };
            if (a == "-c" || a == "--count") o.count = value("--count");
            else if (a == "-l" || a == "--coverage") o.coverage = value("--coverage");
            else if (a == "-q" || a == "--quorum") o.quorum = value("--quorum");
            else if (a == "-g" || a == "--groupby") o.group_file = value("--groupby");
            else if (a == "-O" || a == "--order") o.order_file = value("--order");
            else if (a == "-t" || a == "--threads") o.threads = std::atoi(value("--threads").c_str());
            else if (a == "--device") o.device = std::atoi(value("--device").c_str());
            else if (a == "--nodes") o.nodes = (uint32_t)std::strtoul(value("--nodes").c_str(), nullptr, 10);
            else if (a == "--paths") o.paths = (uint32_t)std::strtoul(value("--paths").c_str(), nullptr, 10);
            else if (a == "--seed") o.seed = std::strtoull(value("--seed").c_str(), nullptr, 10);
            else if (a == "--shape") o.shape = value("--shape");
            else if (a == "--name-prefix") o.name_prefix = value("--name-prefix");
            else if (a == "--samples") o.samples = (uint32_t)std::strtoul(value("--samples").c_str(), nullptr, 10);
            else if (a == "-o" || a == "--output") o.out_file = value("--output");
            else if (a == "--cache") o.cache = true;
            else if (a == "-m" || a == "--method") o.method = value("--method");
            else if (a == "-j" || a == "--json") o.json = true;
            else if (a == "-d" || a == "--dry-run") o.dry_run = true;
            else if (a == "--links") o.links = true;
            else if (a == "--sequences") o.sequences = true;
            else if (a == "-a" || a == "--hist" || a == "--total") o.add_hist = o.total = true;
            else if (a == "-S" || a == "--groupby-sample") o.by_sample = true;
            else if (a == "-H" || a == "--groupby-haplotype") o.by_haplotype = true;
            else if (a == "-s" || a == "--subset") o.subset_file = value("--subset");
            else if (a == "-e" || a == "--exclude") o.exclude_file = value("--exclude");
            else if (!a.empty() && a[0] == '-' && a.size() > 1) throw std::runtime_error("unknown option " + a);
            else if (o.file.empty()) o.file = a;
            else throw std::runtime_error("unexpected argument " + a);
        }
        if (o.cmd == "synth" && o.shape == "pggb") {
            if (!o.nodes || !o.samples || o.out_file.empty()) throw std::runtime_error("synth --shape pggb needs --nodes, --samples and -o");
            if (o.threads > 0) ThreadPool::instance().set_threads((unsigned)o.threads);
            uint32_t np = 0;
            uint64_t ne = 0;
            const uint64_t steps = write_pggb_like_gfa(o.out_file, o.seed, o.nodes, o.samples, o.sequences, &np, &ne, o.name_prefix);
            out = "wrote " + o.out_file + ": " + std::to_string(o.nodes) + " nodes, " + std::to_string(ne) + " edges, " +
                  std::to_string(np) + " paths, " + std::to_string(steps) + " steps\n";
            return 0;
        }
        if (o.cmd == "synth") {
            if (!o.shape.empty() && o.shape != "pansyn") throw std::runtime_error("unknown --shape " + o.shape);
            if (!o.nodes || !o.paths || o.out_file.empty()) throw std::runtime_error("synth needs --nodes, --paths and -o");
            if (o.threads > 0) ThreadPool::instance().set_threads((unsigned)o.threads);
            uint64_t steps = write_pansyn_gfa(o.out_file, o.seed, o.nodes, o.paths, o.links, o.sequences);
            out = "wrote " + o.out_file + ": " + std::to_string(o.nodes) + " nodes, " + std::to_string(o.paths) +
                  " paths, " + std::to_string(steps) + " steps\n";
            return 0;
        }
        if (o.cmd == "report" && o.file.empty()) {  // src/commands/report.rs:47-66
            out = "\n# Missing YAML file!\n#\n# Example YAML:\n# To get started copy this into a .yaml file and edit it\n\n"
                  "- graph: ../graphs/test_graph.gfa\n  grouping: Haplotype\n  analyses:\n    - !Hist\n      count_type: Bp\n"
                  "    - !Growth\n      coverage: 1,1,2\n      quorum: 0,0.9,0\n\n";
            return 0;
        }
        if (o.file.empty()) throw std::runtime_error("missing input file");
        if (o.threads > 0) ThreadPool::instance().set_threads((unsigned)o.threads);
        std::string table;
        if (o.cmd == "report") table = cmd_report(o, cmdline);
        else if (o.json) table = cmd_json(o, cmdline);
        else if (o.cmd == "hist") table = cmd_hist(o, cmdline);
        else if (o.cmd == "histgrowth") table = cmd_histgrowth(o, cmdline, false);
        else if (o.cmd == "growth") table = ends_with(o.file, "tsv") ? cmd_growth_from_hist(o, cmdline) : cmd_histgrowth(o, cmdline, true);
        else if (o.cmd == "ordered-histgrowth") table = cmd_ordered(o, cmdline);
        else if (o.cmd == "similarity") table = cmd_similarity(o, cmdline);
        else if (o.cmd == "table") table = cmd_table(o, cmdline);
        else throw std::runtime_error("unknown subcommand '" + o.cmd + "'\n" + USAGE);
        out = table + "\n";  // writeln!(out, "{table}") in src/lib.rs:322
        return 0;
    } catch (const std::exception &e) {
        err = std::string("error: ") + e.what() + "\n";
        return 1;
    }
}

}  // namespace pnh
