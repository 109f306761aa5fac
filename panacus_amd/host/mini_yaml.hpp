// mini_yaml.hpp -- the block-style YAML subset that panacus report files are written in
// (src/commands/report.rs:41-46 deserialises them with serde_yaml into Vec<AnalysisRun>):
// block sequences and mappings by indentation, plain / quoted scalars, `!Tag` on sequence entries
// and mapping values (serde's externally tagged enums: `- !Hist` + mapping, `grouping: !Custom file`),
// comments.  Flow collections ({...}, [...]), anchors and multi-line scalars are not supported and
// are reported as errors with their line number.
#pragma once
#include <string>
#include <utility>
#include <vector>

namespace pnh {

struct YamlNode {
    enum Kind { NUL, SCALAR, MAP, SEQ } kind = NUL;
    std::string tag;     // without the '!'
    std::string scalar;  // SCALAR
    bool quoted = false; // the scalar was quoted (never null / bool)
    std::vector<std::pair<std::string, YamlNode>> map;
    std::vector<YamlNode> seq;
    int line = 0;
    const YamlNode *get(const std::string &key) const {
        for (const auto &kv : map)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

YamlNode parse_yaml(const std::string &text);  // throws std::runtime_error("yaml: line N: ...")

}  // namespace pnh
