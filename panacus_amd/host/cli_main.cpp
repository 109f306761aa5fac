// panacus-amd: command line front end (see commands.hpp)
#include <cstdio>
#include <string>
#include <vector>

#include "commands.hpp"

int main(int argc, char **argv) {
    std::vector<std::string> args(argv, argv + argc);
    std::string out, err;
    int rc = pnh::run_cli(args, out, err);
    if (!out.empty()) std::fwrite(out.data(), 1, out.size(), stdout);
    if (!err.empty()) std::fwrite(err.data(), 1, err.size(), stderr);
    return rc;
}
