// panacus-amd: command line front end (see commands.hpp)
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "commands.hpp"

int main(int argc, char **argv) {
    std::vector<std::string> args(argv, argv + argc);
    std::string out, err;
    // one command, then the process ends: what its exit reclaims anyway (the mapping of the GFA, the GPU context, the
    // runtime's own state) is not torn down piece by piece -- on a 2.4 GB graph that is a third of the run
    // (not under a profiler or another HSA tool: those write their results from exit handlers)
    const bool tooled = std::getenv("HSA_TOOLS_LIB") || std::getenv("ROCP_TOOL_LIBRARIES") || std::getenv("ROCPROFILER_REGISTER_ENABLED") ||
                        std::getenv("PANACUS_AMD_FULL_EXIT");
    pnh::cli::set_process_exits_after_command(!tooled);
    int rc = pnh::run_cli(args, out, err);
    if (!out.empty()) std::fwrite(out.data(), 1, out.size(), stdout);
    if (!err.empty()) std::fwrite(err.data(), 1, err.size(), stderr);
    std::fflush(stdout);
    std::fflush(stderr);
    if (tooled) return rc;
    _exit(rc);
}
