// commands.hpp -- the CLI surface of the hot path: hist | growth | histgrowth |
// ordered-histgrowth with the reference's options (src/commands/{hist,growth,histgrowth,
// ordered_histgrowth}.rs) on top of the device ABI (include/panacus_amd.h).
#pragma once
#include <string>
#include <vector>

namespace pnh {
// argv[0] = program name. Returns the process exit code; tables go to `out`, diagnostics to `err`.
int run_cli(const std::vector<std::string> &argv, std::string &out, std::string &err);
namespace cli {
void set_process_exits_after_command(bool on);
}

}  // namespace pnh
