// linkage.hpp -- row/column order of the `similarity` table.
//
// Similarity::set_table (src/analyses/similarity.rs:166-182) turns the Jaccard table into a
// condensed matrix of f32 Euclidean row distances (:238-254), clusters it with kodama::linkage
// (crate kodama 0.3.0, Cargo.toml:34; a port of D. Muellner's fastcluster, arXiv:1109.2378) using
// the method of `-m/--method` (analysis_parameter.rs:277-305, default centroid), walks the
// dendrogram's steps for the observations in the order they are merged (:205-217) and permutes
// rows, columns and labels with sort_by_indices (:194-203).  G is a few hundred at most, so all of
// it lives on the host above the C ABI; the G x G intersections come from the GPU (K5).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace pnh {

enum ClusterMethod { LINK_SINGLE = 0, LINK_COMPLETE, LINK_AVERAGE, LINK_WEIGHTED, LINK_WARD, LINK_CENTROID, LINK_MEDIAN };

// "single" | "complete" | "average" | "weighted" | "ward" | "centroid" | "median" (case-insensitive,
// like clap's ignore_case); false for anything else
bool parse_cluster_method(const std::string &s, ClusterMethod &m);

struct LinkStep {  // one merge: SciPy labels (observation i = i, cluster of step k = n + k), c1 < c2
    size_t c1, c2;
    float dissimilarity;
};

// kodama::linkage on a condensed matrix (overwritten, like the crate does)
std::vector<LinkStep> linkage(std::vector<float> &condensed, size_t n, ClusterMethod method);

// the permutation Similarity::set_table applies: perm[k] = index (in the input order) of the group
// printed in row / column k.  `table` is the n x n Jaccard table, row-major.
std::vector<size_t> similarity_order(const std::vector<float> &table, size_t n, ClusterMethod method);

}  // namespace pnh
