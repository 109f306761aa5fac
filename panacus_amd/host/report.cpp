// report.cpp -- the YAML `report` runner and the JSON report surface.
//
// Mirrors, for the analyses this build covers (Hist, Growth, OrderedGrowth, Similarity, Table):
//   src/commands/report.rs:41-46      the YAML file is a list of AnalysisRun
//   src/analysis_parameter.rs:83-151  AnalysisRun { graph, name, subset, exclude, grouping, nice, analyses };
//                                     runs and the analyses of a run are SORTED (derive(Ord)), the
//                                     requirements of a run's analyses are united and the graph state
//                                     is built once for all of them
//   src/graph_broker.rs:149-160       which count types a run builds: two or more of node/bp/edge
//                                     requested -> all three, else the one requested, else node
//   src/analyses/{hist,growth,ordered_histgrowth,similarity}.rs  generate_report_section
//   src/html_report.rs:56-66,396-457  AnalysisSection / ReportItem, serialised by
//                                     serde_json::to_string_pretty (src/lib.rs:308-310)
// One graph load per run, one CSR upload per (run, count type) -- node and bp share a resident
// CSR -- and every analysis of the run reads the same device-side results.  HTML rendering
// (handlebars templates, embedded JS) is out of scope: `report` needs --json or --dry-run; the
// reference's `panacus render` turns such JSON files into the HTML page.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>

#include "commands_internal.hpp"
#include "growth_closed_form.hpp"
#include "linkage.hpp"
#include "mini_yaml.hpp"

namespace pnh {
namespace cli {
namespace {

// ---- the model: AnalysisParameter / AnalysisRun ------------------------------------------------
enum AnalysisKind { A_HIST = 0, A_GROWTH, A_TABLE, A_NODE_DISTRIBUTION, A_INFO, A_ORDERED_GROWTH, A_COVERAGE_LINE, A_SIMILARITY, A_CUSTOM };
enum CountSel { C_NODE = 0, C_BP, C_EDGE, C_ALL };  // CountType's declaration order (src/util.rs:44-49)

struct OptString {  // Option<String>: None < Some(_)
    bool some = false;
    std::string v;
    bool operator<(const OptString &o) const { return some != o.some ? !some : v < o.v; }
    bool operator==(const OptString &o) const { return some == o.some && v == o.v; }
};

struct Analysis {
    AnalysisKind kind = A_HIST;
    CountSel count = C_NODE;
    OptString coverage, quorum, order;
    bool add_hist = false, total = false;
    int cluster = LINK_CENTROID;
    int line = 0;
    // derive(Ord): variant index, then the fields in declaration order
    bool operator<(const Analysis &o) const {
        if (kind != o.kind) return kind < o.kind;
        switch (kind) {
            case A_HIST: return count < o.count;
            case A_GROWTH:
                if (!(coverage == o.coverage)) return coverage < o.coverage;
                if (!(quorum == o.quorum)) return quorum < o.quorum;
                return add_hist < o.add_hist;
            case A_TABLE:
                if (count != o.count) return count < o.count;
                if (total != o.total) return total < o.total;
                return order < o.order;
            case A_ORDERED_GROWTH:
                if (!(coverage == o.coverage)) return coverage < o.coverage;
                if (!(quorum == o.quorum)) return quorum < o.quorum;
                if (!(order == o.order)) return order < o.order;
                return count < o.count;
            case A_SIMILARITY:
                if (count != o.count) return count < o.count;
                return cluster < o.cluster;
            default: return false;
        }
    }
};

struct Run {
    std::string graph, subset, exclude;
    OptString name;
    int grouping = 0;  // 0 none, 1 Sample, 2 Haplotype, 3 Custom(file)  (None < Some; Sample < Haplotype < Custom)
    std::string group_file;
    bool nice = false;
    std::vector<Analysis> analyses;
    bool operator<(const Run &o) const {
        if (graph != o.graph) return graph < o.graph;
        if (!(name == o.name)) return name < o.name;
        if (subset != o.subset) return subset < o.subset;
        if (exclude != o.exclude) return exclude < o.exclude;
        if (grouping != o.grouping) return grouping < o.grouping;
        if (group_file != o.group_file) return group_file < o.group_file;
        if (nice != o.nice) return nice < o.nice;
        return std::lexicographical_compare(analyses.begin(), analyses.end(), o.analyses.begin(), o.analyses.end());
    }
};

[[noreturn]] void bad(const YamlNode &n, const std::string &msg) {
    throw std::runtime_error("report config, line " + std::to_string(n.line) + ": " + msg);
}

std::string scalar_of(const YamlNode &n, const char *what) {
    if (n.kind != YamlNode::SCALAR) bad(n, std::string(what) + " must be a plain value");
    return n.scalar;
}
OptString opt_string(const YamlNode *n, const char *what) {
    OptString o;
    if (n && n->kind != YamlNode::NUL) {
        o.some = true;
        o.v = scalar_of(*n, what);
    }
    return o;
}
bool bool_of(const YamlNode *n, const char *what, bool dflt) {
    if (!n || n->kind == YamlNode::NUL) return dflt;
    const std::string v = scalar_of(*n, what);
    if (v == "true") return true;
    if (v == "false") return false;
    bad(*n, std::string(what) + " must be true or false");
}
CountSel count_of(const YamlNode *n) {
    if (!n || n->kind == YamlNode::NUL) return C_NODE;  // #[serde(default)] -> CountType::Node
    const std::string v = scalar_of(*n, "count_type");
    if (v == "Node") return C_NODE;
    if (v == "Bp") return C_BP;
    if (v == "Edge") return C_EDGE;
    if (v == "All") return C_ALL;
    bad(*n, "unknown variant `" + v + "`, expected one of `Node`, `Bp`, `Edge`, `All`");
}

Analysis parse_analysis(const YamlNode &n) {
    Analysis a;
    a.line = n.line;
    std::string tag = n.tag;
    if (tag.empty() && n.kind == YamlNode::SCALAR) tag = n.scalar;  // a unit variant written as a plain word
    static const std::map<std::string, AnalysisKind> kinds = {
        {"Hist", A_HIST}, {"Growth", A_GROWTH}, {"Table", A_TABLE}, {"NodeDistribution", A_NODE_DISTRIBUTION}, {"Info", A_INFO},
        {"OrderedGrowth", A_ORDERED_GROWTH}, {"CoverageLine", A_COVERAGE_LINE}, {"Similarity", A_SIMILARITY}, {"Custom", A_CUSTOM}};
    auto it = kinds.find(tag);
    if (it == kinds.end()) bad(n, "unknown analysis `" + tag + "` (expected !Hist, !Growth, !OrderedGrowth, !Similarity, !Table, ...)");
    a.kind = it->second;
    if (a.kind == A_NODE_DISTRIBUTION || a.kind == A_INFO || a.kind == A_COVERAGE_LINE || a.kind == A_CUSTOM)
        bad(n, "!" + tag + " lies outside the hist / growth hot path this build covers (run it with the reference)");
    if (n.kind != YamlNode::MAP && n.kind != YamlNode::NUL && !(n.kind == YamlNode::SCALAR && n.tag.empty()))
        bad(n, "!" + tag + " takes a mapping of parameters");
    static const std::map<AnalysisKind, std::set<std::string>> fields = {
        {A_HIST, {"count_type"}}, {A_GROWTH, {"coverage", "quorum", "add_hist"}}, {A_TABLE, {"count_type", "total", "order"}},
        {A_ORDERED_GROWTH, {"coverage", "quorum", "order", "count_type"}}, {A_SIMILARITY, {"count_type", "cluster_method"}}};
    for (const auto &kv : n.map)
        if (!fields.at(a.kind).count(kv.first)) bad(kv.second, "unknown field `" + kv.first + "` of !" + tag);
    a.count = count_of(n.get("count_type"));
    a.coverage = opt_string(n.get("coverage"), "coverage");
    a.quorum = opt_string(n.get("quorum"), "quorum");
    a.order = opt_string(n.get("order"), "order");
    a.add_hist = bool_of(n.get("add_hist"), "add_hist", false);
    if (a.kind == A_TABLE) {
        if (!n.get("total")) bad(n, "missing field `total`");  // no serde default
        a.total = bool_of(n.get("total"), "total", false);
    }
    if (const YamlNode *cm = n.get("cluster_method")) {
        static const char *names[] = {"Single", "Complete", "Average", "Weighted", "Ward", "Centroid", "Median"};
        const std::string v = scalar_of(*cm, "cluster_method");
        a.cluster = -1;
        for (int k = 0; k < 7; ++k)
            if (v == names[k]) a.cluster = k;
        if (a.cluster < 0) bad(*cm, "unknown variant `" + v + "` of cluster_method");
    }
    return a;
}

std::vector<Run> parse_runs(const std::string &file) {
    std::ifstream f(file, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + file);
    std::stringstream ss;
    ss << f.rdbuf();
    const YamlNode doc = parse_yaml(ss.str());
    if (doc.kind == YamlNode::NUL) return {};
    if (doc.kind != YamlNode::SEQ) bad(doc, "the config is a list of runs (`- graph: ...`)");
    std::vector<Run> runs;
    for (const YamlNode &rn : doc.seq) {
        if (rn.kind != YamlNode::MAP) bad(rn, "a run is a mapping with `graph` and `analyses`");
        static const std::set<std::string> known = {"graph", "name", "subset", "exclude", "grouping", "nice", "analyses"};
        for (const auto &kv : rn.map)
            if (!known.count(kv.first)) bad(kv.second, "unknown field `" + kv.first + "` of a run");
        Run r;
        if (!rn.get("graph")) bad(rn, "missing field `graph`");
        r.graph = scalar_of(*rn.get("graph"), "graph");
        r.name = opt_string(rn.get("name"), "name");
        if (const YamlNode *s = rn.get("subset")) r.subset = s->kind == YamlNode::NUL ? "" : scalar_of(*s, "subset");
        if (const YamlNode *s = rn.get("exclude")) r.exclude = s->kind == YamlNode::NUL ? "" : scalar_of(*s, "exclude");
        r.nice = bool_of(rn.get("nice"), "nice", false);
        if (const YamlNode *g = rn.get("grouping")) {
            if (g->kind == YamlNode::NUL && g->tag.empty()) {
            } else if (g->tag == "Custom") {
                r.grouping = 3;
                r.group_file = scalar_of(*g, "grouping: !Custom <file>");
            } else {
                const std::string v = !g->tag.empty() ? g->tag : scalar_of(*g, "grouping");
                if (v == "Sample") r.grouping = 1;
                else if (v == "Haplotype") r.grouping = 2;
                else bad(*g, "unknown variant `" + v + "`, expected one of `Sample`, `Haplotype`, `Custom`");
            }
        }
        const YamlNode *an = rn.get("analyses");
        if (!an) bad(rn, "missing field `analyses`");
        if (an->kind != YamlNode::SEQ && an->kind != YamlNode::NUL) bad(*an, "`analyses` is a list");
        for (const YamlNode &a : an->seq) r.analyses.push_back(parse_analysis(a));
        runs.push_back(std::move(r));
    }
    return runs;
}

// ---- JSON as serde_json::to_string_pretty writes it ----------------------------------------------
std::string json_string(const std::string &s) {
    std::string o = "\"";
    for (unsigned char c : s) {
        switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            default:
                if (c < 0x20) {
                    char buf[8];
                    std::snprintf(buf, sizeof buf, "\\u%04x", c);
                    o += buf;
                } else {
                    o += (char)c;  // UTF-8 passes through
                }
        }
    }
    return o + "\"";
}

// ryu's formatting (what serde_json prints for floats): shortest digits; decimal notation while the
// decimal point stays within [-5 (f32: -6), 16 (f32: 13)] digits of the number, scientific beyond
template <typename F>
std::string json_float(F x, int max_prec, int sci_low, int sci_high) {
    if (std::isnan(x) || std::isinf(x)) return "null";
    if (x == 0) return std::signbit(x) ? "-0.0" : "0.0";
    char buf[64];
    int prec = 1;
    for (; prec <= max_prec; ++prec) {
        std::snprintf(buf, sizeof buf, "%.*e", prec - 1, (double)x);
        if ((F)std::strtod(buf, nullptr) == x && (sizeof(F) == 8 || std::strtof(buf, nullptr) == (float)x)) break;
    }
    std::string s(buf);
    const bool neg = s[0] == '-';
    if (neg) s.erase(0, 1);
    const size_t epos = s.find('e');
    const int exp10 = std::atoi(s.c_str() + epos + 1);
    std::string digits;
    for (size_t i = 0; i < epos; ++i)
        if (s[i] != '.') digits += s[i];
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    const int nd = (int)digits.size();
    const int kk = exp10 + 1;  // position of the decimal point relative to the first digit
    std::string out;
    if (nd <= kk && kk <= sci_high) {  // 1234e7 -> 12340000000.0
        out = digits + std::string((size_t)(kk - nd), '0') + ".0";
    } else if (0 < kk && kk <= sci_high) {  // 1234e-2 -> 12.34
        out = digits.substr(0, (size_t)kk) + "." + digits.substr((size_t)kk);
    } else if (sci_low < kk && kk <= 0) {  // 1234e-6 -> 0.001234
        out = "0." + std::string((size_t)(-kk), '0') + digits;
    } else {  // 1.234e30
        out = digits.substr(0, 1);
        if (nd > 1) out += "." + digits.substr(1);
        out += "e" + std::to_string(kk - 1);
    }
    return neg ? "-" + out : out;
}
std::string json_f64(double x) { return json_float<double>(x, 17, -5, 16); }
std::string json_f32(float x) { return json_float<float>(x, 9, -6, 13); }

struct Json {  // a tiny pretty printer: two-space indent, "key": value, empty collections on one line
    std::string out;
    int depth = 0;
    std::vector<bool> first;
    void nl() { out += "\n" + std::string((size_t)depth * 2, ' '); }
    void sep() {
        if (first.empty()) return;
        if (!first.back()) out += ",";
        first.back() = false;
        nl();
    }
    void open(char c) {
        out += c;
        ++depth;
        first.push_back(true);
    }
    void close(char c) {
        const bool empty = first.back();
        first.pop_back();
        --depth;
        if (!empty) nl();
        out += c;
    }
    void key(const std::string &k) {
        sep();
        out += json_string(k) + ": ";
    }
    void value(const std::string &raw) {
        sep();
        out += raw;
    }
    void strings(const std::vector<std::string> &v) {
        open('[');
        for (const auto &s : v) value(json_string(s));
        close(']');
    }
    void doubles(const std::vector<double> &v) {
        open('[');
        for (double d : v) value(json_f64(d));
        close(']');
    }
};

struct Section {  // AnalysisSection (html_report.rs:56-66) with its one ReportItem
    std::string analysis, run_name, run_id, countable, id, table;
    enum { BAR, MULTIBAR, HEATMAP } item = BAR;
    std::string name, x_label, y_label;
    std::vector<std::string> names, labels;
    std::vector<double> values;                 // Bar
    std::vector<std::vector<double>> rows;      // MultiBar
    std::vector<std::vector<float>> heat;       // Heatmap
    bool log_toggle = false;
};

void write_section(Json &j, const Section &s) {
    j.sep();
    j.open('{');
    j.key("analysis"); j.out += json_string(s.analysis);
    j.key("run_name"); j.out += json_string(s.run_name);
    j.key("run_id"); j.out += json_string(s.run_id);
    j.key("countable"); j.out += json_string(s.countable);
    j.key("items");
    j.open('[');
    j.sep();
    j.open('{');
    if (s.item == Section::BAR) {
        j.key("Bar");
        j.open('{');
        j.key("id"); j.out += json_string(s.id);
        j.key("name"); j.out += json_string(s.name);
        j.key("x_label"); j.out += json_string(s.x_label);
        j.key("y_label"); j.out += json_string(s.y_label);
        j.key("labels"); j.strings(s.labels);
        j.key("values"); j.doubles(s.values);
        j.key("log_toggle"); j.out += s.log_toggle ? "true" : "false";
        j.close('}');
    } else if (s.item == Section::MULTIBAR) {
        j.key("MultiBar");
        j.open('{');
        j.key("id"); j.out += json_string(s.id);
        j.key("names"); j.strings(s.names);
        j.key("x_label"); j.out += json_string(s.x_label);
        j.key("y_label"); j.out += json_string(s.y_label);
        j.key("labels"); j.strings(s.labels);
        j.key("values");
        j.open('[');
        for (const auto &row : s.rows) {
            j.sep();
            j.doubles(row);
        }
        j.close(']');
        j.key("log_toggle"); j.out += s.log_toggle ? "true" : "false";
        j.close('}');
    } else {
        j.key("Heatmap");
        j.open('{');
        j.key("id"); j.out += json_string(s.id);
        j.key("name"); j.out += json_string(s.name);
        j.key("x_labels"); j.strings(s.labels);
        j.key("y_labels"); j.strings(s.labels);
        j.key("values");
        j.open('[');
        for (const auto &row : s.heat) {
            j.sep();
            j.open('[');
            for (float v : row) j.value(json_f32(v));
            j.close(']');
        }
        j.close(']');
        j.close('}');
    }
    j.close('}');
    j.close(']');
    j.key("id"); j.out += json_string(s.id);
    j.key("table"); j.out += json_string(s.table);
    j.key("plot_downloads");  // get_default_plot_downloads (src/util.rs:72-78): tuples -> arrays
    j.open('[');
    static const char *dl[3][2] = {{"png", "Download as png"}, {"svg", "Download as svg"}, {"vega-editor", "Open in vega editor"}};
    for (auto &d : dl) {
        j.sep();
        j.open('[');
        j.value(json_string(d[0]));
        j.value(json_string(d[1]));
        j.close(']');
    }
    j.close(']');
    j.close('}');
}

std::string replace_chars(std::string s, const std::string &chars) {
    for (char &c : s)
        if (chars.find(c) != std::string::npos) c = '-';
    return s;
}
std::string lower(std::string s) {
    for (char &c : s) c = (char)std::tolower((unsigned char)c);  // str::to_lowercase on the ASCII range
    return s;
}
const char *count_word(CountType c) { return count_name(c); }

// ---- one run ---------------------------------------------------------------------------------------
struct RunState {
    const Run &run;
    const Options &opts;
    std::string cmdline;
    std::unique_ptr<GraphStorage> graph;
    bool graph_names_are_ranks() const { return graph && graph->names_are_ranks(); }
    std::vector<CountType> built;  // the count types GraphBroker::from_gfa builds for this run
    std::string order_file;        // Task::OrderChange persists for the rest of the run
    std::unique_ptr<Device> dev;
    Masking mk;
    std::string run_name, run_id;
    // hists of the run (order-independent): one per built count type
    bool have_hists = false;
    std::vector<std::vector<uint64_t>> hists;

    RunState(const Run &r, const Options &o, const std::string &cl) : run(r), opts(o), cmdline(cl) {}

    PathOrder path_order() const { return graph->path_order(mk.mode, mk.group_file, order_file, mk.subset_file, mk.exclude_file); }

    void prepare() {
        // requirement union (analysis_parameter.rs:137-151) -> count types (graph_broker.rs:149-160)
        bool node = false, bp = false, edge = false;
        auto add = [&](CountSel c) {
            node = node || c == C_NODE || c == C_ALL;
            bp = bp || c == C_BP || c == C_ALL;
            edge = edge || c == C_EDGE || c == C_ALL;
        };
        std::set<CountSel> by_group;
        for (const Analysis &a : run.analyses) {
            if (a.kind != A_GROWTH) add(a.count);
            if (a.kind == A_ORDERED_GROWTH || a.kind == A_SIMILARITY || a.kind == A_TABLE) by_group.insert(a.count);
        }
        if (by_group.size() > 1)  // graph_broker.rs:233-236
            throw std::runtime_error("Panacus is currently not able to have multiple Abaci By Group for different countables. "
                                     "Please run panacus either multiple times or wait for the planned pipelining feature");
        for (CountSel c : by_group)
            if (c == C_ALL) throw std::runtime_error("count type All cannot be resolved by group");
        const int n = (int)node + (int)bp + (int)edge;
        if (n >= 2) built = {COUNT_NODE, COUNT_BP, COUNT_EDGE};
        else if (bp) built = {COUNT_BP};
        else if (edge) built = {COUNT_EDGE};
        else built = {COUNT_NODE};
        bool need_edges = false;
        for (CountType c : built) need_edges = need_edges || c == COUNT_EDGE;
        Options lo = opts;
        lo.file = run.graph;
        lo.subset_file = run.subset;
        lo.exclude_file = run.exclude;
        // the GPU comes up -- and takes the text of the GFA for the first upload of the run -- while the graph is read
        dev.reset(new Device(opts.device, wants_device_tokeniser(lo, built)));
        graph = load_graph(lo, need_edges, dev.get());
        dev->preload(PNX_PRELOAD_PASS | (graph->steps_tokenisable_on_device() ? PNX_PRELOAD_GFA : 0u) | (need_edges ? PNX_PRELOAD_LINKS : 0u));
        mk.mode = run.grouping == 1 ? GROUP_SAMPLE : run.grouping == 2 ? GROUP_HAPLOTYPE : run.grouping == 3 ? GROUP_FILE : GROUP_PATHID;
        mk.group_file = run.group_file;
        mk.subset_file = run.subset;
        mk.exclude_file = run.exclude;
        // GraphBroker::get_default_run_name / get_run_id (graph_broker.rs:249-271)
        static const char *gname[] = {"", "Group By Sample", "Group By Haplotype", "Group By "};
        if (run.name.some) run_name = run.name.v;
        else if (run.grouping) run_name = run.graph + "-" + run.subset + "-" + gname[run.grouping] + (run.grouping == 3 ? run.group_file : "");
        else run_name = run.graph + "-" + run.subset;
        run_id = replace_chars(lower(run_name), " _#/\"");
    }

    void ensure_hists() {
        if (have_hists) return;
        hists = device_hists(*dev, *graph, built, path_order(), mk);
        have_hists = true;
    }

    std::string section_id(const char *prefix, const std::string &rid) const { return std::string(prefix) + replace_chars(lower(rid), " |\\"); }

    // Hist::generate_table / generate_report_section (analyses/hist.rs:24-96)
    void hist_sections(std::vector<Section> &out) {
        ensure_hists();
        std::vector<std::vector<std::string>> headers = {{"panacus", "count", "", ""}};
        std::vector<std::vector<double>> cols;
        for (size_t k = 0; k < built.size(); ++k) {
            cols.push_back(to_f64(hists[k]));
            headers.push_back({"hist", count_word(built[k]), "", ""});
        }
        const std::string table = "`" + metadata_comments(cmdline) + write_table(headers, cols) + "`";
        const std::string rid = run_id + "-hist";
        for (size_t k = 0; k < built.size(); ++k) {
            Section s;
            s.id = section_id("cov-hist-", rid) + "-" + count_word(built[k]);
            s.analysis = "Coverage Histogram";
            s.table = table;
            s.run_name = run_name;
            s.run_id = rid;
            s.countable = count_word(built[k]);
            s.item = Section::BAR;
            s.name = run.graph;
            s.x_label = "taxa";
            s.y_label = std::string("#") + count_word(built[k]) + "s";
            for (size_t i = 0; i < hists[k].size(); ++i) s.labels.push_back(std::to_string(i));
            s.values = to_f64(hists[k]);
            s.log_toggle = true;
            out.push_back(std::move(s));
        }
    }

    // Growth::generate_table / generate_report_section (analyses/growth.rs:33-160)
    void growth_sections(const Analysis &a, std::vector<Section> &out) {
        ensure_hists();
        const ThresholdContainer tc = ThresholdContainer::parse_params(a.quorum.some ? a.quorum.v : "0", a.coverage.some ? a.coverage.v : "1");
        std::vector<std::vector<std::string>> headers = {{"panacus", "count", "coverage", "quorum"}};
        std::vector<std::vector<double>> cols;
        if (a.add_hist)
            for (size_t k = 0; k < built.size(); ++k) {
                cols.push_back(to_f64(hists[k]));
                headers.push_back({"hist", count_word(built[k]), "", ""});
            }
        std::vector<std::vector<std::vector<double>>> growths;
        for (size_t k = 0; k < built.size(); ++k) {
            growths.push_back(all_growths(hists[k], tc, (unsigned)opts.threads));
            for (const auto &col : growths.back()) cols.push_back(col);
            growth_headers(headers, "growth", built[k], tc);
        }
        const std::string table = "`# " + cmdline + "\n" + write_table(headers, cols) + "`";
        std::vector<std::string> labels;
        for (size_t t = 0; t < tc.coverage.size(); ++t) {
            // quorum: Relative(x) => (x * 100.0).to_string(), Absolute(x) => (x * 100).to_string()  (growth.rs:113-116)
            const Threshold q = tc.quorum[t];
            const std::string qs = q.kind == THR_ABSOLUTE ? std::to_string((uint64_t)q.value * 100) : format_f64(q.value * 100.0);
            labels.push_back("coverage \xE2\x89\xA5 " + threshold_string(tc.coverage[t]) + ", quorum \xE2\x89\xA5 " + qs + "%");
        }
        const std::string rid = run_id + "-growth";
        for (size_t k = 0; k < built.size(); ++k) {
            Section s;
            s.id = section_id("pan-growth-", rid) + "-" + count_word(built[k]);
            s.analysis = "Pangenome Growth";
            s.run_name = run_name;
            s.run_id = rid;
            s.countable = count_word(built[k]);
            s.table = table;
            s.item = Section::MULTIBAR;
            s.names = labels;
            s.x_label = "taxa";
            s.y_label = std::string("#") + count_word(built[k]) + "s";
            const size_t len = growths[k].empty() ? 0 : growths[k][0].size();
            for (size_t i = 1; i < len; ++i) s.labels.push_back(std::to_string(i));  // (1..v[0].len())
            for (const auto &row : growths[k]) {
                std::vector<double> r = row;
                for (double &v : r)
                    if (std::isnan(v)) v = 0.0;
                s.rows.push_back(std::move(r));
            }
            s.log_toggle = false;
            out.push_back(std::move(s));
        }
    }

    static CountType single(CountSel c) { return c == C_BP ? COUNT_BP : c == C_EDGE ? COUNT_EDGE : COUNT_NODE; }

    // Task::OrderChange + OrderedHistgrowth (analysis_parameter.rs:239-244; analyses/ordered_histgrowth.rs:47-101)
    void ordered_sections(const Analysis &a, std::vector<Section> &out) {
        order_file = a.order.some ? a.order.v : "";  // no order: the rank in the subset list / the GFA (the CLI's documented default)
        const CountType ct = single(a.count);
        const ThresholdContainer tc = ThresholdContainer::parse_params(a.quorum.some ? a.quorum.v : "0", a.coverage.some ? a.coverage.v : "1");
        const PathOrder order = path_order();
        const uint32_t G = (uint32_t)order.groups.size();
        upload(*dev, *graph, ct, order, mk, true);
        const std::vector<std::vector<double>> growths = device_ordered_growth(*dev, tc, G);
        std::vector<std::vector<std::string>> headers = {{"panacus", "count", "coverage", "quorum"}};
        std::vector<std::vector<double>> cols;
        for (const auto &res : growths) {
            std::vector<double> col(G + 1);
            col[0] = std::numeric_limits<double>::quiet_NaN();
            for (uint32_t j = 0; j < G; ++j) col[j + 1] = res[j];
            cols.push_back(std::move(col));
        }
        growth_headers(headers, "ordered-growth", ct, tc);
        Section s;
        const std::string rid = run_id + "-orderedgrowth";
        s.id = section_id("pan-ordered-growth-", rid);
        s.analysis = "Ordered Growth";
        s.run_name = run_name;
        s.run_id = rid;
        s.countable = count_word(ct);
        s.table = "`" + metadata_comments(cmdline) + write_ordered_table(headers, cols, order.groups) + "`";
        s.item = Section::MULTIBAR;
        for (size_t t = 0; t < tc.coverage.size(); ++t)  // quorum via get_string here (ordered_histgrowth.rs:60-67)
            s.names.push_back("coverage \xE2\x89\xA5 " + threshold_string(tc.coverage[t]) + ", quorum \xE2\x89\xA5 " +
                              threshold_string(tc.quorum[t]) + "%");
        s.x_label = "taxa";
        s.y_label = std::string(count_word(ct)) + "s";
        s.labels = order.groups;
        s.rows = growths;
        s.log_toggle = false;
        out.push_back(std::move(s));
    }

    // Similarity (analyses/similarity.rs:48-90)
    void similarity_sections(const Analysis &a, std::vector<Section> &out) {
        static const char *methods[] = {"single", "complete", "average", "weighted", "ward", "centroid", "median"};
        const CountType ct = single(a.count);
        const PathOrder order = path_order();
        const size_t G = order.groups.size();
        upload(*dev, *graph, ct, order, mk);
        const SimilarityResult r = device_similarity(*dev, order.groups, methods[a.cluster]);
        Section s;
        const std::string rid = run_id + "-similarity";
        s.id = section_id("sim-heat-", rid) + "-" + count_word(ct);
        s.analysis = "Similarity Heatmap";
        s.table = "`" + metadata_comments(cmdline) + similarity_table_string(r, order.groups) + "`";
        s.run_name = run_name;
        s.run_id = rid;
        s.countable = count_word(ct);
        s.item = Section::HEATMAP;
        s.name = run.graph;
        for (size_t k = 0; k < G; ++k) s.labels.push_back(order.groups[r.perm[k]]);
        for (size_t i = 0; i < G; ++i) {
            std::vector<float> row(G);
            for (size_t j = 0; j < G; ++j) row[j] = r.table[r.perm[i] * G + r.perm[j]];
            s.heat.push_back(std::move(row));
        }
        out.push_back(std::move(s));
    }
};

std::string describe(const Run &r) {  // --dry-run: the plan, run by run, in execution order
    static const char *kind[] = {"Hist", "Growth", "Table", "NodeDistribution", "Info", "OrderedGrowth", "CoverageLine", "Similarity", "Custom"};
    static const char *cnt[] = {"Node", "Bp", "Edge", "All"};
    static const char *grp[] = {"None", "Sample", "Haplotype", "Custom"};
    std::string s = "GraphStateChange { graph: \"" + r.graph + "\", name: " + (r.name.some ? "Some(\"" + r.name.v + "\")" : "None") +
                    ", subset: \"" + r.subset + "\", exclude: \"" + r.exclude + "\", grouping: " + grp[r.grouping] +
                    (r.grouping == 3 ? "(\"" + r.group_file + "\")" : "") + ", nice: " + (r.nice ? "true" : "false") + " }\n";
    for (const Analysis &a : r.analyses) {
        if (a.kind == A_ORDERED_GROWTH) s += "  OrderChange(" + (a.order.some ? "Some(\"" + a.order.v + "\")" : std::string("None")) + ")\n";
        s += std::string("  Analysis ") + kind[a.kind];
        if (a.kind != A_GROWTH) s += std::string(" { count_type: ") + cnt[a.count] + " }";
        else s += " { coverage: " + (a.coverage.some ? a.coverage.v : "1") + ", quorum: " + (a.quorum.some ? a.quorum.v : "0") + " }";
        s += "\n";
    }
    return s;
}

std::string run_all(std::vector<Run> runs, const Options &o, const std::string &cmdline) {
    std::sort(runs.begin(), runs.end());                                    // AnalysisRun::convert_to_tasks: runs.sort()
    for (Run &r : runs) std::sort(r.analyses.begin(), r.analyses.end());    // to_tasks: analyses.sort()
    if (o.dry_run) {
        std::string plan;
        for (const Run &r : runs) plan += describe(r);
        return plan;
    }
    if (!o.json)
        throw std::runtime_error("`report` renders HTML in the reference; this build writes the report sections as JSON: pass --json "
                                 "(the reference's `panacus render` turns JSON files into the HTML page) or --dry-run");
    std::vector<Section> sections;
    for (const Run &r : runs) {
        RunState st(r, o, cmdline);
        st.prepare();
        // nice: true (graph.rs:224-229): the reference takes a segment's name, parsed as an integer, for its id while node
        // lengths stay indexed by the rank of the S line -- the same graph only if the names ARE the ranks 1..N, which is
        // also what this build detects by itself (and then converts names without a lookup, on the host and on the device)
        if (r.nice && !st.graph_names_are_ranks())
            throw std::runtime_error("nice: true needs segment names that are the integers 1..N in the order of the S lines (" + r.graph + ")");
        for (const Analysis &a : r.analyses) {
            switch (a.kind) {
                case A_HIST: st.hist_sections(sections); break;
                case A_GROWTH: st.growth_sections(a, sections); break;
                case A_ORDERED_GROWTH: st.ordered_sections(a, sections); break;
                case A_SIMILARITY: st.similarity_sections(a, sections); break;
                case A_TABLE: break;  // Table::generate_report_section returns no section (analyses/table.rs:51-56)
                default: break;
            }
        }
    }
    Json j;
    j.open('[');
    for (const Section &s : sections) write_section(j, s);
    j.close(']');
    return j.out;
}

}  // namespace

std::string cmd_report(const Options &o, const std::string &cmdline) { return run_all(parse_runs(o.file), o, cmdline); }
std::string json_number_f64(double x) { return json_f64(x); }
std::string json_number_f32(float x) { return json_f32(x); }

// `hist|growth|histgrowth|ordered-histgrowth|similarity --json`: the run the reference's subcommand
// would put together (src/commands/*.rs) through the same runner
std::string cmd_json(const Options &o, const std::string &cmdline) {
    Run r;
    r.graph = o.file;
    r.subset = o.subset_file;
    r.exclude = o.exclude_file;
    r.grouping = o.by_haplotype ? 2 : o.by_sample ? 1 : !o.group_file.empty() ? 3 : 0;
    r.group_file = o.group_file;
    std::string l;
    for (char c : o.count) l += (char)std::tolower((unsigned char)c);
    const CountSel cs = l == "bp" ? C_BP : l == "edge" ? C_EDGE : l == "all" ? C_ALL : C_NODE;
    if (l != "node" && l != "bp" && l != "edge" && l != "all") throw std::runtime_error("invalid value '" + o.count + "' for '--count <count>'");
    auto growth = [&]() {
        Analysis g;
        g.kind = A_GROWTH;
        g.coverage = OptString{true, o.coverage};
        g.quorum = OptString{true, o.quorum};
        g.add_hist = o.add_hist;
        return g;
    };
    if (o.cmd == "hist") {
        Analysis a;
        a.kind = A_HIST;
        a.count = cs;
        r.analyses = {a};
    } else if (o.cmd == "growth") {
        r.analyses = {growth()};
    } else if (o.cmd == "histgrowth") {
        Analysis a;
        a.kind = A_HIST;
        a.count = cs;
        r.analyses = {a, growth()};
    } else if (o.cmd == "ordered-histgrowth") {
        Analysis a;
        a.kind = A_ORDERED_GROWTH;
        a.count = cs;
        a.coverage = OptString{true, o.coverage};
        a.quorum = OptString{true, o.quorum};
        if (!o.order_file.empty()) a.order = OptString{true, o.order_file};
        r.analyses = {a};
    } else if (o.cmd == "similarity") {
        Analysis a;
        a.kind = A_SIMILARITY;
        a.count = cs;
        ClusterMethod m;
        if (!parse_cluster_method(o.method, m)) throw std::runtime_error("invalid value '" + o.method + "' for --method");
        a.cluster = (int)m;
        r.analyses = {a};
    } else {
        throw std::runtime_error("--json is available for hist, growth, histgrowth, ordered-histgrowth and similarity");
    }
    Options jo = o;
    jo.json = true;
    jo.dry_run = false;
    return run_all({r}, jo, cmdline);
}

}  // namespace cli
}  // namespace pnh
