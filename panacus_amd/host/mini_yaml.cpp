#include "mini_yaml.hpp"

#include <stdexcept>

namespace pnh {
namespace {

struct Line {
    int indent;
    std::string text;
    int no;
};

[[noreturn]] void fail(int line, const std::string &msg) {
    throw std::runtime_error("yaml: line " + std::to_string(line) + ": " + msg);
}

std::string rtrim(std::string s) {
    while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r')) s.pop_back();
    return s;
}

// cut a trailing comment: '#' at the start or after white space, outside quotes
std::string strip_comment(const std::string &s) {
    char q = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (q) {
            if (c == q) q = 0;
            else if (c == '\\' && q == '"') ++i;
        } else if (c == '\'' || c == '"') {
            q = c;
        } else if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) {
            return s.substr(0, i);
        }
    }
    return s;
}

std::vector<Line> split_lines(const std::string &text) {
    std::vector<Line> out;
    size_t b = 0;
    int no = 0;
    while (b <= text.size()) {
        size_t e = text.find('\n', b);
        if (e == std::string::npos) e = text.size();
        ++no;
        std::string raw = rtrim(strip_comment(text.substr(b, e - b)));
        b = e + 1;
        size_t ind = 0;
        while (ind < raw.size() && raw[ind] == ' ') ++ind;
        if (ind < raw.size() && raw[ind] == '\t') fail(no, "tabs cannot indent YAML");
        if (ind == raw.size()) continue;  // blank
        const std::string body = raw.substr(ind);
        if (body == "---" && ind == 0) continue;
        if (body == "...") break;
        out.push_back(Line{(int)ind, body, no});
    }
    return out;
}

YamlNode scalar_node(std::string s, int line) {
    YamlNode n;
    n.line = line;
    s = rtrim(s);
    size_t b = 0;
    while (b < s.size() && s[b] == ' ') ++b;
    s = s.substr(b);
    if (s.empty() || s == "~" || s == "null") return n;  // NUL
    if (s[0] == '{' || s[0] == '[') fail(line, "flow collections ({...}, [...]) are not supported");
    if (s[0] == '&' || s[0] == '*' || s[0] == '|' || s[0] == '>') fail(line, "anchors, aliases and block scalars are not supported");
    n.kind = YamlNode::SCALAR;
    if (s[0] == '\'' || s[0] == '"') {
        const char q = s[0];
        if (s.size() < 2 || s.back() != q) fail(line, "unterminated quoted scalar");
        std::string v;
        for (size_t i = 1; i + 1 < s.size(); ++i) {
            if (q == '\'' && s[i] == '\'' && i + 2 < s.size() && s[i + 1] == '\'') {
                v += '\'';
                ++i;
            } else if (q == '"' && s[i] == '\\' && i + 2 < s.size()) {
                const char c = s[++i];
                v += c == 'n' ? '\n' : c == 't' ? '\t' : c;
            } else {
                v += s[i];
            }
        }
        n.scalar = v;
        n.quoted = true;
    } else {
        n.scalar = s;
    }
    return n;
}

// position of the ": " / trailing ':' that ends a mapping key, or npos
size_t key_end(const std::string &s) {
    char q = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (q) {
            if (c == q) q = 0;
        } else if ((c == '\'' || c == '"') && i == 0) {
            q = c;
        } else if (c == ':' && (i + 1 == s.size() || s[i + 1] == ' ')) {
            return i;
        }
    }
    return std::string::npos;
}

struct Parser {
    std::vector<Line> lines;
    size_t pos = 0;

    YamlNode block(int indent) {
        if (pos >= lines.size() || lines[pos].indent < indent) return YamlNode();
        const Line &l = lines[pos];
        if (l.text[0] == '-' && (l.text.size() == 1 || l.text[1] == ' ')) return sequence(l.indent);
        if (key_end(l.text) != std::string::npos) return mapping(l.indent);
        YamlNode n = scalar_node(l.text, l.no);
        ++pos;
        return n;
    }

    // value that follows a tag or a "key:" on the same line; empty -> a nested block (or null)
    YamlNode value(std::string rest, int parent_indent, int line, bool seq_may_share_indent) {
        std::string tag;
        size_t b = 0;
        while (b < rest.size() && rest[b] == ' ') ++b;
        rest = rest.substr(b);
        if (!rest.empty() && rest[0] == '!') {
            size_t e = rest.find(' ');
            tag = rest.substr(1, e == std::string::npos ? std::string::npos : e - 1);
            rest = e == std::string::npos ? "" : rest.substr(e + 1);
            b = 0;
            while (b < rest.size() && rest[b] == ' ') ++b;
            rest = rest.substr(b);
        }
        YamlNode n;
        if (!rest.empty()) {
            n = scalar_node(rest, line);
        } else if (pos < lines.size() && lines[pos].indent > parent_indent) {
            n = block(lines[pos].indent);
        } else if (seq_may_share_indent && pos < lines.size() && lines[pos].indent == parent_indent && lines[pos].text[0] == '-' &&
                   (lines[pos].text.size() == 1 || lines[pos].text[1] == ' ')) {
            n = sequence(parent_indent);  // "key:\n- a\n- b"
        }
        n.tag = tag;
        if (!n.line) n.line = line;
        return n;
    }

    YamlNode sequence(int indent) {
        YamlNode n;
        n.kind = YamlNode::SEQ;
        n.line = lines[pos].no;
        while (pos < lines.size() && lines[pos].indent == indent && lines[pos].text[0] == '-' &&
               (lines[pos].text.size() == 1 || lines[pos].text[1] == ' ')) {
            Line &l = lines[pos];
            std::string rest = l.text.size() > 1 ? l.text.substr(2) : "";
            size_t skip = 0;
            while (skip < rest.size() && rest[skip] == ' ') ++skip;
            rest = rest.substr(skip);
            const int inner = indent + 2 + (int)skip;
            if (!rest.empty() && rest[0] != '!' && rest[0] != '\'' && rest[0] != '"' && key_end(rest) != std::string::npos) {
                // "- key: value": a mapping that starts on the entry's own line
                l.indent = inner;
                l.text = rest;
                n.seq.push_back(mapping(inner));
            } else {
                const int no = l.no;
                ++pos;
                n.seq.push_back(value(rest, indent, no, false));
            }
        }
        if (pos < lines.size() && lines[pos].indent > indent) fail(lines[pos].no, "unexpected indentation");
        return n;
    }

    YamlNode mapping(int indent) {
        YamlNode n;
        n.kind = YamlNode::MAP;
        n.line = lines[pos].no;
        while (pos < lines.size() && lines[pos].indent == indent) {
            const Line l = lines[pos];
            if (l.text[0] == '-' && (l.text.size() == 1 || l.text[1] == ' ')) break;  // a sequence at the parent's indent
            const size_t ke = key_end(l.text);
            if (ke == std::string::npos) fail(l.no, "expected 'key: value'");
            YamlNode k = scalar_node(l.text.substr(0, ke), l.no);
            ++pos;
            YamlNode v = value(ke + 1 < l.text.size() ? l.text.substr(ke + 1) : "", indent, l.no, true);
            for (const auto &kv : n.map)
                if (kv.first == k.scalar) fail(l.no, "duplicate key '" + k.scalar + "'");
            n.map.emplace_back(k.scalar, std::move(v));
        }
        if (pos < lines.size() && lines[pos].indent > indent) fail(lines[pos].no, "unexpected indentation");
        return n;
    }
};

}  // namespace

YamlNode parse_yaml(const std::string &text) {
    Parser p;
    p.lines = split_lines(text);
    if (p.lines.empty()) return YamlNode();
    YamlNode n = p.block(p.lines[0].indent);
    if (p.pos < p.lines.size()) fail(p.lines[p.pos].no, "content after the end of the document");
    return n;
}

}  // namespace pnh
