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
    std::string out_file, shape;
    bool links = false, sequences = false;
};

// RAII over pnx_ctx.  pnx_init -- HIP runtime start-up, streams, code objects: 0.15-0.2 s -- runs on a thread of its own
// from the constructor on, i.e. beside the GFA parse every command starts with; the first use of ctx() joins it and
// rethrows an initialisation error (no GPU: no CPU fallback).
struct Device {
    explicit Device(int ordinal);
    ~Device();
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    pnx_ctx *ctx() const;
    void check(int rc) const;

private:
    mutable std::future<pnx_ctx *> init_;
    mutable pnx_ctx *ctx_ = nullptr;
};

struct Masking {  // -g/-S/-H grouping + -s/-e lists: what the item table of a count type is cut down by
    GroupMode mode = GROUP_PATHID;
    std::string group_file, subset_file, exclude_file;
    bool any() const { return !subset_file.empty() || !exclude_file.empty(); }
};

using Uncovered = std::vector<std::pair<uint32_t, uint64_t>>;  // quantify_uncovered_bps (abacus.rs:1187-1229)

std::unique_ptr<GraphStorage> load_graph(const Options &o, bool index_edges);
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
