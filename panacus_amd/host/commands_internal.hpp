// commands_internal.hpp -- what the CLI commands (commands.cpp) and the report runner (report.cpp)
// share: parsed options, the RAII device handle, upload and histogram helpers.
#pragma once
#include <cstdint>
#include <functional>
#include <future>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/panacus_amd.h"
#include "gfa_graph.hpp"
#include "tables.hpp"

namespace pnh {
namespace cli {

struct Options {
    std::string cmd, file, count = "node", coverage = "1", quorum = "0", group_file, order_file, subset_file, exclude_file;
    std::string method = "centroid";  // similarity: ClusterMethod::default() (analysis_parameter.rs:287-291)
    bool add_hist = false, by_sample = false, by_haplotype = false, total = false, cache = false;
    bool json = false, dry_run = false;  // report / --json
    int threads = 0, device = 0;
    // synth
    uint64_t seed = 42;
    uint32_t nodes = 0, paths = 0, samples = 0;
    std::string out_file, shape, name_prefix;
    bool links = false, sequences = false;
};

// RAII over pnx_ctx.  pnx_init -- HIP runtime start-up, streams, code objects: 0.15-0.2 s -- runs on a thread of its own
// from the constructor on, i.e. beside the GFA parse every command starts with; the first use of ctx() joins it and
// rethrows an initialisation error (no GPU: no CPU fallback).
// expect_text: the same thread then waits for the bytes of the GFA (offer_text, from GraphStorage::from_gfa's hook) and copies
// them to HBM (pnx_gfa_text_upload) -- beside the host's line scan -- for the device tokeniser (pnx_set_csr_gfa).
struct Device {
    explicit Device(int ordinal, bool expect_text = false);
    ~Device();
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    pnx_ctx *ctx() const;
    void check(int rc) const;
    void offer_text(const char *data, size_t size, std::shared_ptr<const void> keep) const;
    void no_text() const;            // nothing will be offered (idempotent; also after an offer: no effect)
    void leak() const { leaked_ = true; }  // the process is about to exit: leave the context to the driver
    bool text_uploaded() const;      // after ctx(): the offered text is in HBM -- true ONCE (the upload that uses it consumes the copy)
    // pnx_preload on the calling thread: the device code of the routes the command is about to take is loaded while the
    // context still comes up / the text still travels on the thread of the constructor (a failure is not reported: the
    // first launch loads what is missing)
    void preload(uint32_t what) const;

private:
    struct TextSlot;
    std::shared_ptr<TextSlot> text_;
    mutable std::future<pnx_ctx *> init_;
    mutable pnx_ctx *ctx_ = nullptr;
    mutable bool leaked_ = false;
    int ordinal_ = 0;
};
// A stand-alone CLI process ends right after its one command: the command then does not tear down what the exit of the
// process reclaims anyway -- the 2 GB mapping of the GFA, the GPU context -- and main() leaves through _exit once the table
// is written (cli_main.cpp sets this; the in-process entry pnh_run_cli never does).
bool process_exits_after_command();
void set_process_exits_after_command(bool on);
// at the end of a command: under process_exits_after_command() the graph and the device are left as they are
void finish_command(std::unique_ptr<GraphStorage> &g, const Device &dev);

struct Masking {  // -g/-S/-H grouping + -s/-e lists: what the item table of a count type is cut down by
    GroupMode mode = GROUP_PATHID;
    std::string group_file, subset_file, exclude_file;
    bool any() const { return !subset_file.empty() || !exclude_file.empty(); }
};

using Uncovered = std::vector<std::pair<uint32_t, uint64_t>>;  // quantify_uncovered_bps (abacus.rs:1187-1229)

std::unique_ptr<GraphStorage> load_graph(const Options &o, bool index_edges, const Device *dev = nullptr, bool links_on_device = false);
// will this run make its node ItemTable from the raw text on the device, if the graph allows it?  (then the text is worth copying early)
bool wants_device_tokeniser(const Options &o, const std::vector<CountType> &cts);
std::vector<CountType> count_types(const std::string &c, bool allow_all);
GroupMode group_mode(const Options &o);
Masking masking(const Options &o);
// the device-side cut of the walks under -s / -e lists (pnx_set_csr_cut) + the host's replay of the partial pieces
Uncovered upload_cut(const std::function<pnx_ctx *()> &get_ctx, const GraphStorage &g, CountType ct, const Masking &mk,
                     bool growth_weights);
Uncovered upload(const Device &dev, const GraphStorage &g, CountType ct, const PathOrder &order, const Masking &mk,
                 bool growth_weights = false, bool per_item_output = false);
std::vector<std::vector<uint64_t>> device_hists(const Device &dev, const GraphStorage &g, const std::vector<CountType> &cts,
                                                const PathOrder &order, const Masking &mk);
std::vector<double> to_f64(const std::vector<uint64_t> &v);
std::vector<std::vector<double>> all_growths(const std::vector<uint64_t> &hist, const ThresholdContainer &tc, unsigned threads);
void growth_headers(std::vector<std::vector<std::string>> &headers, const char *what, CountType ct, const ThresholdContainer &tc);
// ordered growth of the resident order: res[t][j] as f64 (AbacusByGroup::calc_growth, abacus.rs:989-1032)
std::vector<std::vector<double>> device_ordered_growth(const Device &dev, const ThresholdContainer &tc, uint32_t n_groups);
// Jaccard table of the resident groups + the dendrogram permutation (Similarity::set_table)
struct SimilarityResult {
    std::vector<float> table;  // G x G, input order
    std::vector<size_t> perm;  // row / column k shows group perm[k]
};
SimilarityResult device_similarity(const Device &dev, const std::vector<std::string> &groups, const std::string &method);
std::string similarity_table_string(const SimilarityResult &r, const std::vector<std::string> &groups);  // get_table_string

// report.cpp
std::string json_number_f64(double x);  // a float as serde_json (ryu) prints it
std::string json_number_f32(float x);
std::string cmd_report(const Options &o, const std::string &cmdline);
// the JSON of one analysis of the classic subcommands (--json): the same sections `report` makes
std::string cmd_json(const Options &o, const std::string &cmdline);

}  // namespace cli
}  // namespace pnh
