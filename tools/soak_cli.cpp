// soak_cli.cpp -- the in-process CLI (pnh_run_cli, panacus_amd/host/host_api.cpp) driven without Python: what
// tests/test_host_cli.py::test_cli_bed_intervals_synthetic and tests/test_gpu_cli_fuzz.py do (hundreds of commands, each with
// a GPU context of its own, in ONE process), in a loop, on one or several threads.  Written for the hunt of round 5's
// intermittent SIGABRT of whole in-process sessions: the same binary is built plain, with -fsanitize=address,undefined and
// with -fsanitize=thread (tools/build_sanitized.sh) and runs under rocgdb, none of which a pytest process makes easy.
//
//   soak_cli <workdir> <iterations> [threads = 1] [golden_dir]
//
// Every thread works in <workdir>/t<k>: a pansyn graph (synth --links), random BED subset / exclude lists over its paths,
// the seven table commands under them; every table is compared with the table the SAME command printed in iteration 0 (the
// oracle is not linked here: the harness looks for crashes, races and run-to-run differences, parity is the test suite's).
// Exit code 0 = every command ran and repeated itself.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>

extern "C" int pnh_run_cli(const char *argv_joined, char *out_buf, uint64_t out_cap, uint64_t *out_len, char *err_buf,
                           uint64_t err_cap, uint64_t *err_len);

namespace {
std::atomic<uint64_t> g_commands{0}, g_failures{0};
std::mutex g_print;

struct Result {
    int rc;
    std::string out, err;
};

Result run(const std::vector<std::string> &args) {
    std::string joined = "panacus-amd";
    for (const auto &a : args) joined += "\n" + a;
    std::vector<char> out(1 << 22), err(1 << 16);
    uint64_t ol = 0, el = 0;
    int rc = pnh_run_cli(joined.c_str(), out.data(), out.size(), &ol, err.data(), err.size(), &el);
    if (ol + 1 > out.size()) {
        out.resize(ol + 16);
        rc = pnh_run_cli(joined.c_str(), out.data(), out.size(), &ol, err.data(), err.size(), &el);
    }
    g_commands.fetch_add(1);
    return Result{rc, std::string(out.data()), std::string(err.data())};
}

// the table without its comment header (the header echoes the command line and the version)
std::string body(const std::string &t) {
    std::istringstream in(t);
    std::string line, out;
    while (std::getline(in, line))
        if (line.empty() || line[0] != '#') out += line + "\n";
    return out;
}

uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct PathInfo {
    std::string name;
    uint64_t bp = 0;
};

// names and bp lengths of the paths of a small GFA (S lines: sequence or LN:i: tag; P lines: comma-separated oriented steps)
std::vector<PathInfo> paths_of(const std::string &gfa) {
    std::ifstream f(gfa);
    std::map<std::string, uint64_t> len;
    std::vector<PathInfo> out;
    std::string line;
    while (std::getline(f, line)) {
        if (line.size() < 2) continue;
        std::vector<std::string> col;
        size_t a = 0;
        while (true) {
            size_t b = line.find('\t', a);
            col.push_back(line.substr(a, b == std::string::npos ? b : b - a));
            if (b == std::string::npos) break;
            a = b + 1;
        }
        if (col[0] == "S" && col.size() >= 3) {
            uint64_t l = col[2] == "*" ? 0 : col[2].size();
            for (size_t k = 3; k < col.size(); ++k)
                if (col[k].rfind("LN:i:", 0) == 0) l = std::strtoull(col[k].c_str() + 5, nullptr, 10);
            len[col[1]] = l;
        } else if (col[0] == "P" && col.size() >= 3) {
            PathInfo p;
            p.name = col[1];
            size_t s = 0;
            const std::string &st = col[2];
            while (s < st.size()) {
                size_t e = st.find(',', s);
                if (e == std::string::npos) e = st.size();
                if (e > s + 1) p.bp += len[st.substr(s, e - s - 1)];
                s = e + 1;
            }
            out.push_back(p);
        }
    }
    return out;
}

void fail(const std::string &what, const std::vector<std::string> &args, const Result &r) {
    g_failures.fetch_add(1);
    std::lock_guard<std::mutex> g(g_print);
    std::fprintf(stderr, "FAIL %s:", what.c_str());
    for (const auto &a : args) std::fprintf(stderr, " %s", a.c_str());
    std::fprintf(stderr, "\n  rc=%d err=%s\n", r.rc, r.err.substr(0, 400).c_str());
}

void worker(int k, const std::string &workdir, int iters, const std::string &golden) {
    const std::string dir = workdir + "/t" + std::to_string(k);
    mkdir(dir.c_str(), 0755);
    const std::string gfa = dir + "/syn.gfa";
    {
        std::vector<std::string> a = {"synth", "--nodes", "3000", "--paths", "8", "--links", "--seed", std::to_string(42 + k), "-o", gfa};
        Result r = run(a);
        if (r.rc != 0) return fail("synth", a, r);
    }
    const std::vector<PathInfo> paths = paths_of(gfa);
    if (paths.empty()) {
        std::fprintf(stderr, "no paths in %s\n", gfa.c_str());
        g_failures.fetch_add(1);
        return;
    }
    // the command set: fixed per thread (BED lists drawn once), so that iteration i > 0 must print what iteration 0 printed
    uint64_t seed = 11 + 1000 * (uint64_t)k;
    auto bed = [&](const std::string &file, int rows) {
        std::ofstream f(file);
        for (int r = 0; r < rows; ++r) {
            const PathInfo &p = paths[splitmix(seed) % paths.size()];
            const uint64_t bp = std::max<uint64_t>(p.bp, 2);
            const uint64_t lo = splitmix(seed) % bp;
            const uint64_t ln = 1 + splitmix(seed) % (bp / 3 + 1);
            std::string name = p.name.substr(0, p.name.find(':'));
            f << name << "\t" << lo << "\t" << lo + ln << "\n";
        }
    };
    std::vector<std::vector<std::string>> cmds;
    for (int rep = 0; rep < 2; ++rep) {
        const std::string sub = dir + "/s" + std::to_string(rep) + ".bed", exc = dir + "/e" + std::to_string(rep) + ".bed";
        bed(sub, 6);
        bed(exc, 3);
        for (int mode = 0; mode < 3; ++mode) {
            std::vector<std::string> extra;
            if (mode != 1) extra.insert(extra.end(), {"-s", sub});
            if (mode != 0) extra.insert(extra.end(), {"-e", exc});
            for (const char *ct : {"node", "bp", "edge"}) {
                std::vector<std::string> a = {"histgrowth", "-a", "-c", ct, "-l", "1,2", "-q", "0,0.5"};
                a.insert(a.end(), extra.begin(), extra.end());
                a.push_back(gfa);
                cmds.push_back(a);
            }
            std::vector<std::string> o = {"ordered-histgrowth", "-c", "bp", "-l", "1,2", "-q", "0.3,0"};
            o.insert(o.end(), extra.begin(), extra.end());
            o.push_back(gfa);
            cmds.push_back(o);
            std::vector<std::string> h = {"hist", "-c", "all"};
            h.insert(h.end(), extra.begin(), extra.end());
            h.push_back(gfa);
            cmds.push_back(h);
        }
    }
    cmds.push_back({"histgrowth", "-c", "all", "-a", "-l", "1,2", "-q", "0,0.5", "-S", gfa});
    cmds.push_back({"ordered-histgrowth", "-c", "bp", "-l", "1,2", "-q", "0,0.3", "-H", gfa});
    cmds.push_back({"similarity", "-c", "node", gfa});
    cmds.push_back({"similarity", "-c", "bp", "-m", "average", gfa});
    cmds.push_back({"table", "-c", "node", gfa});
    cmds.push_back({"table", "-c", "edge", "-a", gfa});
    cmds.push_back({"hist", "-c", "bp", "--cache", gfa});
    if (!golden.empty()) {
        const std::string chrM = golden + "/chrM_test.gfa";
        cmds.push_back({"histgrowth", "-a", "-c", "all", "-l", "1,2", "-q", "0,0.5", chrM});
        cmds.push_back({"ordered-histgrowth", "-c", "node", "-l", "1,2", "-q", "0.3,0", "-s", golden + "/bed_chrM/inclusion.bed3", chrM});
        cmds.push_back({"hist", "-c", "all", "-S", chrM});
        cmds.push_back({"growth", "-l", "1,2", "-q", "0,0.5", golden + "/t_groups.hist.tsv"});
    }
    std::vector<std::string> first(cmds.size());
    for (int it = 0; it < iters; ++it) {
        for (size_t c = 0; c < cmds.size(); ++c) {
            Result r = run(cmds[c]);
            if (r.rc != 0) {
                fail("command", cmds[c], r);
                continue;
            }
            const std::string b = body(r.out);
            if (it == 0) first[c] = b;
            else if (b != first[c]) fail("table differs from iteration 0", cmds[c], r);
        }
        if (k == 0 && (it + 1) % 10 == 0) {
            std::lock_guard<std::mutex> g(g_print);
            std::fprintf(stderr, "iteration %d: %llu commands, %llu failures\n", it + 1, (unsigned long long)g_commands.load(),
                         (unsigned long long)g_failures.load());
        }
    }
}
}  // namespace

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: soak_cli <workdir> <iterations> [threads] [golden_dir]\n");
        return 2;
    }
    const std::string workdir = argv[1];
    const int iters = std::atoi(argv[2]);
    const int threads = argc > 3 ? std::max(1, std::atoi(argv[3])) : 1;
    const std::string golden = argc > 4 ? argv[4] : "";
    mkdir(workdir.c_str(), 0755);
    std::vector<std::thread> ts;
    for (int k = 1; k < threads; ++k) ts.emplace_back(worker, k, workdir, iters, golden);
    worker(0, workdir, iters, golden);
    for (auto &t : ts) t.join();
    std::fprintf(stderr, "soak_cli: %llu commands, %llu failures\n", (unsigned long long)g_commands.load(),
                 (unsigned long long)g_failures.load());
    return g_failures.load() ? 1 : 0;
}
