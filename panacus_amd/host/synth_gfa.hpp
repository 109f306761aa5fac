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

// A pggb-SHAPED graph (BASELINE.json configs[4] is a 402 MB download that is not in the container: this is its
// structural stand-in, not its data): integer segment names in pangenome order (behind name_prefix, if one is given:
// `s12` as minigraph-cactus names its segments); ~45 % shared backbone segments with
// chr22-like lengths, the rest variant nodes (mostly 1 bp) carried by a U-shaped share of the haplotypes; every sample
// has two haplotypes, each cut into several contig paths with unassembled gaps between them (PanSN
// `sample#hap#contig`), plus two single-path references (`chm13#chr22`, `grch38#chr22`); haplotypes carry
// inversions (a stretch walked backwards with '-' orientations) and tandem duplications (a stretch walked twice), so
// paths are NEARLY monotone in the ids, as real ones are; L lines = the edges the walks use, canonical, sorted.
// Returns the number of path steps; *n_paths / *n_edges receive the counts.
uint64_t write_pggb_like_gfa(const std::string &file, uint64_t seed, uint32_t n_nodes, uint32_t n_samples, bool sequences,
                             uint32_t *n_paths = nullptr, uint64_t *n_edges = nullptr, const std::string &name_prefix = "");
}  // namespace pnh
