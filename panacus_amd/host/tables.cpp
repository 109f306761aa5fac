#include "tables.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>

namespace pnh {

// Rust's Display for floats: shortest digits that round-trip, never scientific notation,
// integers without a fractional part, "NaN", "inf".
namespace {
template <typename F, typename Parse>
std::string format_float(F x, int max_prec, Parse parse) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "inf" : "-inf";
    if (x == 0) return std::signbit(x) ? "-0" : "0";
    char buf[64];
    int prec = 1;
    for (; prec <= max_prec; ++prec) {
        std::snprintf(buf, sizeof buf, "%.*e", prec - 1, (double)x);
        if (parse(buf) == x) break;
    }
    // buf = [-]d.ddddde[+-]XX
    std::string s(buf);
    bool neg = s[0] == '-';
    if (neg) s.erase(0, 1);
    size_t epos = s.find('e');
    int exp10 = std::atoi(s.c_str() + epos + 1);
    std::string digits;
    for (size_t i = 0; i < epos; ++i)
        if (s[i] != '.') digits += s[i];
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    std::string out;
    const int nd = (int)digits.size();
    if (exp10 >= nd - 1) {  // integer: pad with zeros
        out = digits + std::string((size_t)(exp10 - (nd - 1)), '0');
    } else if (exp10 >= 0) {
        out = digits.substr(0, (size_t)exp10 + 1) + "." + digits.substr((size_t)exp10 + 1);
    } else {
        out = "0." + std::string((size_t)(-exp10 - 1), '0') + digits;
    }
    return neg ? "-" + out : out;
}
}  // namespace

std::string format_f64(double x) {
    return format_float<double>(x, 17, [](const char *b) { return std::strtod(b, nullptr); });
}
std::string format_f32(float x) {
    return format_float<float>(x, 9, [](const char *b) { return std::strtof(b, nullptr); });
}

std::string threshold_string(Threshold t) {
    if (t.kind == THR_ABSOLUTE) return std::to_string((uint64_t)t.value);
    return format_f64(t.value);
}

const char *count_name(CountType c) {
    switch (c) {
        case COUNT_NODE: return "node";
        case COUNT_BP: return "bp";
        case COUNT_EDGE: return "edge";
    }
    return "?";
}

bool parse_count_name(const std::string &s, CountType &c) {
    std::string l;
    for (char ch : s) l += (char)std::tolower((unsigned char)ch);
    if (l == "node") c = COUNT_NODE;
    else if (l == "bp") c = COUNT_BP;
    else if (l == "edge") c = COUNT_EDGE;
    else return false;
    return true;
}

namespace {
std::vector<std::string> split(const std::string &s, char d) {
    std::vector<std::string> out;
    size_t b = 0;
    for (;;) {
        size_t e = s.find(d, b);
        out.push_back(s.substr(b, e == std::string::npos ? std::string::npos : e - b));
        if (e == std::string::npos) break;
        b = e + 1;
    }
    return out;
}
std::string trim(const std::string &s) {
    size_t b = 0, e = s.size();
    while (b < e && std::isspace((unsigned char)s[b])) ++b;
    while (e > b && std::isspace((unsigned char)s[e - 1])) --e;
    return s.substr(b, e - b);
}
}  // namespace

// parse_threshold_cli + ThresholdContainer::parse_params (hist.rs:207-322)
ThresholdContainer ThresholdContainer::parse_params(const std::string &quorum, const std::string &coverage) {
    ThresholdContainer tc;
    if (!quorum.empty()) {
        int i = 1;
        for (const std::string &el : split(quorum, ',')) {
            std::string t = trim(el);
            char *end = nullptr;
            double v = std::strtod(t.c_str(), &end);
            if (t.empty() || *end != 0)
                throw std::runtime_error("threshold \"" + quorum + "\" (" + std::to_string(i) + ". element in list) is required to be float, but isn't.");
            if (!(v >= 0.0 && v <= 1.0))
                throw std::runtime_error("relative threshold \"" + quorum + "\" (" + std::to_string(i) + ". element in list) must be within [0,1].");
            tc.quorum.push_back(Threshold{THR_RELATIVE, v});
            ++i;
        }
    }
    if (tc.quorum.empty()) throw std::runtime_error("quorum threshold setting requires at least one element, but none is given");
    if (!coverage.empty()) {
        int i = 1;
        for (const std::string &el : split(coverage, ',')) {
            std::string t = trim(el);
            bool ok = !t.empty();
            for (char ch : t) ok = ok && ch >= '0' && ch <= '9';
            if (!ok)
                throw std::runtime_error("threshold \"" + coverage + "\" (" + std::to_string(i) + ". element in list) is required to be integer, but isn't.");
            tc.coverage.push_back(Threshold{THR_ABSOLUTE, (double)std::strtoull(t.c_str(), nullptr, 10)});
            ++i;
        }
    }
    if (tc.coverage.empty()) throw std::runtime_error("coverage threshold setting requires at least one element, but none is given");
    if (tc.quorum.size() != tc.coverage.size()) {
        if (tc.quorum.size() == 1) tc.quorum.assign(tc.coverage.size(), tc.quorum[0]);
        else if (tc.coverage.size() == 1) tc.coverage.assign(tc.quorum.size(), tc.coverage[0]);
        else throw std::runtime_error("number of coverage and quorum threshold must match, or either one must have a single value");
    }
    return tc;
}

static void push_headers(std::string &res, const std::vector<std::vector<std::string>> &headers) {
    const size_t n = headers.empty() ? 0 : headers[0].size();
    for (size_t i = 0; i < n; ++i) {
        for (size_t j = 0; j < headers.size(); ++j) {
            if (j) res += '\t';
            res += headers[j][i];
        }
        res += '\n';
    }
}

std::string write_table(const std::vector<std::vector<std::string>> &headers,
                        const std::vector<std::vector<double>> &columns) {
    std::string res;
    push_headers(res, headers);
    const size_t n = columns.empty() ? 0 : columns[0].size();
    for (size_t i = 0; i < n; ++i) {
        res += std::to_string(i);
        for (const auto &col : columns) {
            res += '\t';
            res += format_f64(std::floor(col[i]));  // io.rs:484
        }
        res += '\n';
    }
    return res;
}

std::string write_ordered_table(const std::vector<std::vector<std::string>> &headers,
                                const std::vector<std::vector<double>> &columns,
                                const std::vector<std::string> &index) {
    std::string res;
    push_headers(res, headers);
    const size_t n = columns.empty() ? 0 : columns[0].size();
    for (size_t i = 1; i < n; ++i) {  // row 0 (NaN) is dropped, rows are labelled by group (io.rs:508-515)
        res += index[i - 1];
        for (const auto &col : columns) {
            res += '\t';
            res += format_f64(std::floor(col[i]));
        }
        res += '\n';
    }
    return res;
}

std::string metadata_comments(const std::string &cmdline) {
    return "# " + cmdline + "\n# version panacus-amd 0.5.0\n";
}

// parse_tsv + parse_hists (io.rs:153-290)
ParsedHists parse_hists(const std::string &file) {
    std::ifstream in(file, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + file);
    ParsedHists out;
    std::vector<std::vector<std::string>> table;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        std::vector<std::string> row = split(line, '\t');
        if (!row[0].empty() && row[0][0] == '#') {
            out.comments.push_back(line);
            continue;
        }
        bool all_empty = true;
        for (const auto &c : row) all_empty = all_empty && c.empty();
        if (all_empty) continue;
        table.push_back(row);
    }
    if (table.size() < 2) throw std::runtime_error("table appears not to be generated by panacus");
    const size_t ncol = table[0].size();
    if (ncol < 2 || table[0][0] != "panacus") throw std::runtime_error("table appears not to be generated by panacus");
    auto column = [&](size_t j) {
        std::vector<uint64_t> v;
        for (size_t i = 2; i < table.size(); ++i) {
            const std::string &cell = j < table[i].size() ? table[i][j] : std::string();
            bool ok = !cell.empty();
            for (char ch : cell) ok = ok && ch >= '0' && ch <= '9';
            if (!ok)
                throw std::runtime_error("error in line " + std::to_string(i + 1 + out.comments.size()) +
                                         ": value must be integer, but is '" + cell + "'");
            v.push_back(std::strtoull(cell.c_str(), nullptr, 10));
        }
        return v;
    };
    std::vector<uint64_t> index = column(0);
    uint64_t mx = 0;
    for (uint64_t x : index) mx = std::max(mx, x);
    for (size_t j = 1; j < ncol; ++j) {
        if (table[0][j] != "hist") continue;
        CountType c;
        if (j >= table[1].size() || !parse_count_name(table[1][j], c))
            throw std::runtime_error("expected count type declaration, but got '" + (j < table[1].size() ? table[1][j] : "") + "'");
        std::vector<uint64_t> vals = column(j), cov(mx + 1, 0);
        for (size_t i = 0; i < index.size(); ++i) cov[index[i]] = vals[i];
        out.hists.emplace_back(c, std::move(cov));
    }
    if (out.hists.empty()) throw std::runtime_error("table does not contain hist columns");
    return out;
}

}  // namespace pnh
