// synth_gfa.hpp -- pansyn-v1 as GFA text (host side), so that the synthetic benchmark graphs
// of BASELINE.json can also be fed through the normal file path (`panacus-amd synth`, then
// `panacus-amd hist|histgrowth ... file.gfa`).  Same integer-only definition as the device
// generator (csrc/pansyn.hip) and the test oracle; see DESIGN.md "pansyn-v1".
#pragma once
#include <cstdint>
#include <string>

namespace pnh {
// writes S / P (every 4th path as W) / optionally L lines; returns the number of path steps
uint64_t write_pansyn_gfa(const std::string &file, uint64_t seed, uint32_t n_nodes, uint32_t n_paths,
                          bool with_links, bool sequences);
}  // namespace pnh
