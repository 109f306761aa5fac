/*
 * panacus_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A plain-C restatement of the coverage-histogram / pangenome-growth hot path of
 * marschall-lab/panacus v0.4.1 (reference snapshot 2025-06-20).  It exists only so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the
 * reference algorithm; nothing under panacus_amd/ may include, link or call it.
 *
 * Parity status: PINNED.  The reference is pure Rust and cannot be built in this image
 * (no cargo/rustc, no Cargo.lock, no vendored crates), so there is no oracle/_ref.  The
 * restatement is pinned against every known-answer vector the reference repository holds
 * for this path (see tests/test_oracle_golden.py):
 *   - src/graph_broker/abacus.rs:1424-1435,1487-1630  (cdbg + chrM countables / hists)
 *   - tests/test_files/t_groups.hist.tsv               (t_groups node hist)
 *   - src/graph_broker/hist.rs:342-398                 (choose / union / core / quorum f64)
 *   - docs/chr22.hprc-v1.0-pggb.histgrowth.html:266-276 (hist -> growth, 660 values)
 * Ordered growth (abacus.rs:989-1032) and subset/exclude have no numeric golden in the
 * reference repo; ordered growth is a literal restatement cross-checked by the
 * "mean over all orders == closed-form union" identity; subset/exclude (whole paths and BED
 * intervals) is a literal restatement checked against hand-derived tables
 * (tests/test_oracle_bed.py) -- PARITY UNPINNED for those options.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference root).
 */
#ifndef PANACUS_ORACLE_H
#define PANACUS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* count types: src/util.rs:44-49 */
enum { ORC_NODE = 0, ORC_BP = 1, ORC_EDGE = 2 };
/* grouping modes: src/graph_broker/abacus.rs:242-308 */
enum { ORC_GROUP_PATHID = 0, ORC_GROUP_SAMPLE = 1, ORC_GROUP_HAPLOTYPE = 2, ORC_GROUP_FILE = 3 };
/* threshold kinds: src/util.rs:328-364 */
enum { ORC_THR_ABSOLUTE = 0, ORC_THR_RELATIVE = 1 };

typedef struct orc_graph orc_graph;

/* ---- GFA -> GraphStorage (src/graph_broker/graph.rs:195-375) ---- */
orc_graph *orc_graph_from_gfa(const char *gfa_file, int index_edges);
void orc_graph_free(orc_graph *g);
uint64_t orc_graph_n_nodes(const orc_graph *g);
uint64_t orc_graph_n_edges(const orc_graph *g);
uint64_t orc_graph_n_paths(const orc_graph *g);
const uint32_t *orc_graph_node_lens(const orc_graph *g); /* n_nodes+1, [0]=0 */
/* PathSegment::id() of path i (graph.rs:558-579), with ":start-end" appended if coords */
const char *orc_graph_path_display(const orc_graph *g, uint64_t i);
const char *orc_graph_last_error(void);

/* ---- grouping + visiting order (abacus.rs:242-347, 555-559) ----
 * Fills path_idx[n_out], group_id[n_out] (caller arrays of size n_paths) and returns
 * n_groups; group names available through orc_graph_group_name afterwards.
 * order_file may be NULL (file order).  Returns -1 on error. */
int64_t orc_graph_path_order(orc_graph *g, int group_mode, const char *group_file,
                             const char *order_file, uint64_t *path_idx, uint64_t *group_id,
                             uint64_t *n_out);
const char *orc_graph_group_name(const orc_graph *g, uint64_t gid);
/* same with whole-path subset (-s) / exclude (-e) lists; either may be NULL.  NOTE: no golden
 * output for -s/-e exists in the reference repository -- parity unpinned, the restatement is the
 * definition. */
int64_t orc_graph_path_order_masked(orc_graph *g, int group_mode, const char *group_file,
                                    const char *order_file, const char *subset_file, const char *exclude_file,
                                    uint64_t *path_idx, uint64_t *group_id, uint64_t *n_out);
int orc_graph_exclude_flags(orc_graph *g, int count_type, const char *exclude_file, uint8_t *flags);

/* ---- subset / exclude lists with BED coordinates ----
 * parse_gfa_paths_walks with GraphMask.include_coords / exclude_coords (graph_broker/util.rs:208-366,
 * 569-795; src/util.rs:118-310): the ItemTable restricted to the subset intervals (paths outside the
 * subset get an empty entry), the ActiveTable.items flags of the exclude list (`exclude`, n_items+1
 * bytes, may be NULL) and -- for ORC_BP with a subset -- quantify_uncovered_bps (abacus.rs:1187-1229)
 * as parallel arrays sorted by node id (malloc'ed; free with orc_free).  Either file may be NULL.
 * Group names in the lists resolve against the last orc_graph_path_order* call.
 * Returns the number of items or -1.  PARITY UNPINNED: the reference holds the BED inputs
 * (test/bed_chrM) but no expected output; this restatement is the definition. */
int64_t orc_graph_masked_table(const orc_graph *g, int count_type, const char *subset_file, const char *exclude_file,
                               uint64_t **items, uint64_t *prefsum, uint8_t *exclude, uint64_t **uncov_ids,
                               uint64_t **uncov_bps, uint64_t *n_uncov);
/* the uncovered-bp fix-up of construct_hist_bps (abacus.rs:779-785), wrapping like the release build */
void orc_hist_apply_uncovered(const uint32_t *countable, const uint64_t *uncov_ids, const uint64_t *uncov_bps,
                              uint64_t n_uncov, uint64_t *hist);

/* ---- ItemTable (src/util.rs:81-93; graph_broker/util.rs:22-206, 723-795) ----
 * Returns number of items; *items is malloc'ed (caller frees with orc_free), prefsum has
 * n_paths+1 entries (caller array). count_type ORC_NODE/ORC_BP give node ids, ORC_EDGE edge ids */
int64_t orc_graph_item_table(const orc_graph *g, int count_type, uint64_t **items,
                             uint64_t *prefsum);
void orc_free(void *p);

/* ---- AbacusByTotal::coverage over a path order (abacus.rs:539-586, 719-744) ----
 * countable has n_items+1 entries, countable[0] = UINT32_MAX on return. */
void orc_coverage(const uint64_t *items, const uint64_t *prefsum, const uint64_t *path_idx,
                  const uint64_t *group_id, uint64_t n_ordered, uint64_t n_items,
                  const uint8_t *exclude /* n_items+1 or NULL */, uint32_t *countable);

/* construct_hist (abacus.rs:746-762) / construct_hist_bps (abacus.rs:764-787, without the
 * uncovered_bps fix-up which is empty when no subset/exclude is used) */
void orc_hist(const uint32_t *countable, uint64_t n_items, uint64_t n_groups,
              const uint32_t *weights /* NULL => count 1 */, uint64_t *hist /* n_groups+1 */);

/* ---- closed-form growth from a histogram (graph_broker/hist.rs:21-187) ---- */
double orc_choose(uint64_t n, uint64_t k);
/* out has n = hist_len-1 entries (no leading NaN); returns n */
int64_t orc_growth(const uint64_t *hist, uint64_t hist_len, int cov_kind, double cov_val,
                   int quo_kind, double quo_val, double *out);

/* the three branches, callable directly like the reference's unit tests do (hist.rs:352-398);
 * n = hist_len-1, out has n entries */
void orc_growth_union(const uint64_t *hist, uint64_t n, int cov_kind, double cov_val, double *out);
void orc_growth_core(const uint64_t *hist, uint64_t n, int cov_kind, double cov_val, double *out);
void orc_growth_quorum(const uint64_t *hist, uint64_t n, int cov_kind, double cov_val, int quo_kind,
                       double quo_val, double *out);

/* ---- AbacusByGroup (abacus.rs:859-986): r has n_items+2 entries; *c malloc'ed ---- */
int64_t orc_by_group(const uint64_t *items, const uint64_t *prefsum, const uint64_t *path_idx,
                     const uint64_t *group_id, uint64_t n_ordered, uint64_t n_items,
                     const uint8_t *exclude, uint64_t *r, uint64_t **c);
/* the same with report_values (abacus.rs:901-986): *v has one u32 per slot of c */
int64_t orc_by_group_values(const uint64_t *items, const uint64_t *prefsum, const uint64_t *path_idx,
                            const uint64_t *group_id, uint64_t n_ordered, uint64_t n_items,
                            const uint8_t *exclude, uint64_t *r, uint64_t **c, uint32_t **v);
/* AbacusByGroup::calc_growth (abacus.rs:989-1032); weights NULL => node/edge; out n_groups */
void orc_ordered_growth(const uint64_t *r, const uint64_t *c, uint64_t n_items,
                        uint64_t n_groups, int cov_kind, double cov_val, int quo_kind,
                        double quo_val, const uint32_t *weights, double *out);

/* Similarity::set_table before clustering (src/analyses/similarity.rs:119-165): inter[a*G+b] =
   number (or bp, when node_lens != NULL) of items whose group slice holds both a and b,
   lens[a] = path_lens[a], table[a*G+b] = inter as f32 / (lens[a] + lens[b] - inter) as f32.
   Returns -1 when a group has no item (the reference's `path_lens[&i]` panics), else 0. */
int orc_similarity(const uint64_t *r, const uint64_t *c, uint64_t n_items, uint64_t n_groups,
                   const uint32_t *node_lens, uint64_t *inter, uint64_t *lens, float *table);
/* Similarity::set_table after the Jaccard table (similarity.rs:166-217): f32 Euclidean distances
   between the rows, kodama::linkage (crate kodama 0.3.0, absent from the reference tree: restated
   from its published algorithm, see the .c file), get_order_from_dendrogram, sort_by_indices.
   method: 0 single, 1 complete, 2 average, 3 weighted, 4 ward, 5 centroid (default), 6 median.
   Reorders `table` (n x n) in place; perm_out[k] = input index of the group in row k. */
int orc_similarity_order(float *table, uint64_t n_groups, int method, uint64_t *perm_out);
/* one row of AbacusByGroup::to_tsv without `total` (abacus.rs:1093-1112): out[j] = bp (or 1)
   when group j holds item i, else 0 */
void orc_table_row(const uint64_t *r, const uint64_t *c, uint64_t i, uint64_t n_groups,
                   uint64_t bp, uint64_t *out);

/* one row of AbacusByGroup::to_tsv without `total` when v is present (abacus.rs:1098-1108, 1158-1166):
   node/bp: out[j] = v[k] * bp; edge: out[j] = v[j] (the reference indexes by group id; -1 = its panic) */
int orc_table_row_values(const uint64_t *r, const uint64_t *c, const uint32_t *v, uint64_t nnz, uint64_t i,
                         uint64_t n_groups, uint64_t bp, int is_edge, uint64_t *out);

/* y[k] = exp2(x[k]) with the platform libm -- what Rust's f64::exp2 calls (hist.rs:104,131,175,179) */
void orc_exp2(const double *x, double *y, uint64_t n);
/* y[k] = log2(x[k]) with the platform libm -- what Rust's f64::log2 calls (hist.rs:28-32,100-106,171-179) */
void orc_log2(const double *x, double *y, uint64_t n);

/* ---- synthetic pangenome generator pansyn-v1 (DESIGN.md section "pansyn-v1") ---- */
uint64_t pansyn_splitmix64(uint64_t x);
uint32_t pansyn_node_len(uint64_t seed, uint64_t i);
void pansyn_node_lens(uint64_t seed, uint64_t n_nodes, uint32_t *out /* n_nodes+1 */);
/* containment threshold on the 53-bit uniform: path p holds node i iff u53(seed,5,p,i) < thr */
uint64_t pansyn_node_thr(uint64_t seed, uint64_t i, uint64_t n_paths);
/* Generates the CSR (u64 items like the reference ItemTable).  *items malloc'ed. */
int64_t pansyn_generate(uint64_t seed, uint64_t n_nodes, uint64_t n_paths, uint64_t **items,
                        uint64_t *prefsum /* n_paths+1 */);
/* pansyn-v1r: the same paths rearranged in place -- 1 % of the 64-step blocks reversed, 0.1 % replaced by a copy of an earlier
 * block of the path, 0.05 % moved elsewhere in the id space (see the definition in panacus_oracle.c) */
void pansyn_rearrange(uint64_t seed, uint64_t n_nodes, uint64_t n_paths, uint64_t *items, const uint64_t *prefsum);

#ifdef __cplusplus
}
#endif
#endif
