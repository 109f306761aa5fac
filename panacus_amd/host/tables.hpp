// tables.hpp -- TSV surface of hist / growth / histgrowth / ordered-histgrowth, byte-compatible
// with the reference apart from the content of the "#" comment lines.
// Mirrors src/io.rs:460-518 (write_table, write_ordered_table), :546-555 (metadata comments),
// :153-290 (parse_tsv / parse_hists) and the header builders of src/analyses/hist.rs:36-52,
// src/analyses/growth.rs:53-100 and src/io.rs:583-601.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "gfa_graph.hpp"
#include "growth_closed_form.hpp"

namespace pnh {

std::string format_f64(double x);            // Rust `{}` for f64
std::string format_f32(float x);             // Rust `{}` for f32 (shortest digits that round-trip as f32)
std::string threshold_string(Threshold t);   // Threshold::get_string, src/util.rs:343-348
const char *count_name(CountType c);          // src/util.rs:57-69
bool parse_count_name(const std::string &s, CountType &c);

struct ThresholdContainer {  // src/graph_broker/hist.rs:261-322
    std::vector<Threshold> coverage, quorum;
    static ThresholdContainer parse_params(const std::string &quorum, const std::string &coverage);
};

// headers[j] = the four header cells of column j (column 0 is the row-label column)
std::string write_table(const std::vector<std::vector<std::string>> &headers,
                        const std::vector<std::vector<double>> &columns);
std::string write_ordered_table(const std::vector<std::vector<std::string>> &headers,
                                const std::vector<std::vector<double>> &columns,
                                const std::vector<std::string> &index);
std::string metadata_comments(const std::string &cmdline);

struct ParsedHists {
    std::vector<std::pair<CountType, std::vector<uint64_t>>> hists;
    std::vector<std::string> comments;
};
ParsedHists parse_hists(const std::string &file);

}  // namespace pnh
