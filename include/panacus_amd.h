/*
 * panacus_amd.h -- C ABI of the MI355X coverage-histogram / pangenome-growth engine.
 *
 * This is the drop-in boundary for the hot path of marschall-lab/panacus v0.4.1
 * (hist / growth / histgrowth / ordered-histgrowth).  The reference has no FFI seam of
 * its own; the seam sits where its internal constructors turn an ItemTable (CSR of path
 * steps) into results.  Every entry point names the reference routine(s) it replaces
 * (file:line relative to the reference root).  A Rust host binds these with a plain
 * `extern "C"` block; this repository's own host is C++/Python.
 *
 * Conventions
 *   - all functions return 0 on success or a negative PNX_E* code; nothing unwinds across
 *     the ABI; pnx_last_error() holds a human-readable message for the failing call;
 *   - plain pointers + sizes; host buffers are caller-owned and copied by the library;
 *   - one pnx_ctx per process per GPU (one process per GPU; multi-GPU = several
 *     processes, each with its own node-range shard or permutation shard, combined by the
 *     caller with an RCCL all-reduce on the device counters, see pnx_hist_device());
 *   - a pnx_ctx is not thread-safe; calls are serialised by the caller;
 *   - results are exact integers, independent of launch geometry and device count;
 *   - there is NO CPU fallback: every call fails with PNX_ENODEV without a HIP device.
 *
 * Item ids are 1..n_items (0 is the reference's reserved "zero element",
 * abacus.rs:549-551); ItemIdSize narrows from u64 to u32 at this boundary (n_items and
 * n_paths < 2^32-1); step offsets stay u64.
 */
#ifndef PANACUS_AMD_H
#define PANACUS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNX_OK 0
#define PNX_EINVAL (-1)   /* bad argument / call order */
#define PNX_ENODEV (-2)   /* no usable HIP device */
#define PNX_EHIP (-3)     /* HIP runtime error (message has the hipError string) */
#define PNX_ENOMEM (-4)   /* device or host allocation failed */
#define PNX_ELIMIT (-5)   /* input exceeds an implementation limit (stated in the message) */

typedef struct pnx_ctx pnx_ctx;
#define PNX_MAX_IN_FLIGHT 4 /* upper bound of PNX_CFG_MAX_IN_FLIGHT */

/* ---- lifetime ------------------------------------------------------------------------ */
/* device = HIP device ordinal visible to this process (LOCAL_RANK for torchrun launches). */
int pnx_init(pnx_ctx **out, int device);
/* (round 4) PNX_INIT_ONE_SHOT: the caller will ask for a histogram or two and leave (a CLI command) -- the two extra streams
 * of the three-stream pass arrangement (9 ms each to create) are not made up front; they still appear if passes ever overlap
 * (pnx_hist_async behind a pass in flight), possibly on shared hardware queues.  A host that pipelines passes uses pnx_init. */
#define PNX_INIT_ONE_SHOT 1u
int pnx_init_flags(pnx_ctx **out, int device, uint32_t flags);
void pnx_free(pnx_ctx *ctx);
/* (round 4) Loads the device code of the named routes ahead of their first use.  The HIP runtime loads a code object when the
 * first kernel of it is launched (tens of ms) -- on the critical path of a one-shot command.  pnx_preload does that work
 * without launching anything and needs no context: a host thread that has nothing else to do while the context comes up
 * or the GFA text travels to HBM calls it beside them.  Optional; results never depend on it. */
#define PNX_PRELOAD_GFA 1u   /* pnx_set_csr_gfa / pnx_gfa_walks: tokeniser, name table */
#define PNX_PRELOAD_LINKS 2u /* edge counts from GFA text: L lines, edge lookup, renumbering */
#define PNX_PRELOAD_PASS 4u  /* the coverage / histogram pass and the closed-form growth */
#define PNX_PRELOAD_GROWTH 8u   /* pnx_ordered_growth */
#define PNX_PRELOAD_PAIRS 16u   /* pnx_group_intersections (similarity) */
#define PNX_PRELOAD_TABLES 32u  /* pnx_presence, pnx_group_visit_counts (table) */
int pnx_preload(int device, uint32_t what);
/* message of the last failing call on ctx (ctx == NULL: last pnx_init failure) */
const char *pnx_last_error(const pnx_ctx *ctx);
const char *pnx_version(void);
/* (round 5) counts the changes of this header that a binding has to know about (structs that grew, entry points added): a host
 * built against header version v runs against a library with pnx_abi_version() >= v through the _sized entry points */
#define PNX_ABI_VERSION 5
int pnx_abi_version(void);

/* ---- graph upload: the ItemTable (src/util.rs:81-93) ------------------------------------
 * Replaces the hand-off of `item_table` into AbacusByTotal::item_table_to_abacus
 * (src/graph_broker/abacus.rs:539-547) and AbacusByGroup::from_gfa (abacus.rs:803-804).
 *   items     S step ids (1..n_items), duplicates within a path kept, FILE order of paths
 *   path_off  n_paths+1 offsets into items (the reference's id_prefsum)
 *   weights   n_items+1 values (weights[0] ignored) = GraphStorage::node_lens
 *             (graph.rs:323-351) for CountType::Bp, NULL for node/edge counts
 *   exclude   n_items+1 flags = ActiveTable::items (src/util.rs:118-124), NULL if unused
 */
int pnx_set_csr(pnx_ctx *ctx, const uint32_t *items, const uint64_t *path_off, uint32_t n_paths,
                uint32_t n_items, const uint32_t *weights, const uint8_t *exclude);

/* The same upload with one sort key per item (item_key[0] ignored, n_items+1 entries): a hint that the
 * steps of a path rise or fall with the KEY of their items even where they do not with the ids.  The
 * reference numbers edges in the order of the L lines (graph.rs:282-295); with a shuffled link section
 * the edge steps of a path are then random and take the slow atomic route here.  The key of an edge
 * is its canonical pair of ends, (smaller node id << 32) | larger node id, which the reference's host
 * holds in edge2id (graph.rs:276-306).  The library renumbers the items internally by ascending key
 * (stable; on the device: one radix sort of n_items pairs and one pass over the resident steps);
 * every per-item result -- countable, pnx_presence, pnx_group_visit_counts, pnx_get_csr -- is still
 * reported in the CALLER's ids, and no other result depends on item numbering.  Keys that already
 * rise with the ids cost one check.  item_key == NULL is pnx_set_csr. */
int pnx_set_csr_keyed(pnx_ctx *ctx, const uint32_t *items, const uint64_t *path_off, uint32_t n_paths,
                      uint32_t n_items, const uint32_t *weights, const uint8_t *exclude,
                      const uint64_t *item_key);

/* ---- the ItemTable straight from GFA text: step columns tokenised on the device ------------------
 * Replaces parse_path_seq_to_item_vec / parse_walk_seq_to_item_vec (src/graph_broker/util.rs:1021-1091) and the
 * per-step lookups in the node2id map that graph.rs:308-375 builds: the caller finds the P / W lines (it needs their
 * headers anyway) and hands over the TEXT of the step columns; the library splits them at ',' (P: `name+`, `name-`) or
 * at '>' / '<' (W), converts the names and leaves the node ItemTable resident -- as after pnx_set_csr, for node and
 * bp counts.  Segment names must be decimal numbers:
 *   id_of_name == NULL  the number IS the node id (the reference's `nice: true`, graph.rs:224-229: names 1..N in file order)
 *   id_of_name != NULL  node id = id_of_name[number] for number < n_names, 0 = no such segment
 * A step that is not of that form, or names a segment the graph does not have, fails the call with PNX_EINVAL (the
 * reference panics, util.rs:1021); graphs with other names keep the host's parser and pnx_set_csr.
 *   text / text_bytes    host bytes that contain every step column -- any superset, e.g. the whole mapped file.  NULL: the
 *                        bytes handed to pnx_gfa_text_upload before (which lets a host start the copy while it is still
 *                        looking for the lines; the copy is the larger part of the call)
 *   col_begin, col_end   per path the byte range [begin, end) of its step column inside text
 *   is_walk              per path: 1 = W line, 0 = P line
 *   edge_uv, edge_oo     NULL for node / bp counts.  Edge counts: the edges of the graph as in pnx_walks -- n_edges + 1
 *                        entries indexed by edge id ([0] unused), canonical ends (smaller node id << 32 | larger) and
 *                        orientations (o1 << 1 | o2) as Edge::canonical writes them (graph.rs:142-148), every edge once.
 *                        The walks (node id + orientation of every step) then never leave the device: the edge of every
 *                        consecutive step pair is looked up among the edges filed under their smaller end (what
 *                        parse_path_seq_to_item_vec / parse_walk_seq_to_item_vec do per step with edge2id,
 *                        util.rs:1048-1091) and the EDGE ItemTable -- a path of k steps has k - 1 entries -- becomes the
 *                        resident graph with n_edges items; a step pair without an edge fails the call (the reference
 *                        panics, util.rs:1080).  weights must be NULL.
 *                        One case is the CALLER's: update_tables_edgecount (util.rs:723-795) includes an edge only where
 *                        `include_coords[0].0 < p + l`, i.e. it leaves out the edges among the LEADING ZERO-LENGTH nodes of a
 *                        path that starts at coordinate 0.  A graph with a node of length 0 (an empty sequence field) goes
 *                        through pnx_set_csr_cut with the whole-path interval instead, which walks the coordinates
 *                        (this repository's CLI does: GraphStorage::has_zero_length_nodes).
 *   name_off, name_len   (round 4) segment names that are NOT numbers -- `s12`, `utg000012l`, ... : per node (id - 1) the byte
 *                        offset of its name inside text (the name field of its S line) and its length.  The library builds the
 *                        node2id map of graph.rs:308-375 in HBM -- a hash table keyed by the name bytes -- and the tokeniser looks
 *                        every step up there (graph.rs:231, per step, on the host in the reference).  Names of up to 16 bytes;
 *                        a longer one fails the call with PNX_ELIMIT (such graphs keep the host's parser); a name that occurs
 *                        twice fails it with PNX_EINVAL (the reference panics, graph.rs:336).  id_of_name must be NULL.
 *                        name_off == NULL, id_of_name == NULL and n_names == PNX_NAMES_FIND: the library finds the S lines itself
 *                        -- every line of text that starts with 'S', inside the bytes [name_lo, name_hi) (0, 0 = the whole
 *                        text) -- and takes the field behind "S\t" of the i-th of them as the name of segment i + 1
 *                        (graph.rs:323-351); their number must be n_nodes.
 *   name_prefix, name_prefix_len  (round 4) numeric names with the same 1..8 bytes in front of every number -- `s12` of
 *                        minigraph-cactus: the tokeniser checks that the bytes stand there and reads the number behind them;
 *                        id_of_name (or the number itself) as for plain numbers.  Ignored with name_off / PNX_NAMES_FIND.
 *   link_off, n_links    (round 4) edge counts WITHOUT the host's edge map: the byte offset of every L line inside text, in file
 *                        order.  The library parses the lines (both names and orientations, graph.rs:276-306), puts every edge
 *                        in canonical form (graph.rs:142-148), numbers the distinct ones by their first line (duplicates are
 *                        skipped as the reference skips them, graph.rs:296) and looks the step pairs up as above; n_edges is
 *                        reported by pnx_info (n_items).  edge_uv / edge_oo must be NULL then.
 *                        link_off == NULL with n_links == PNX_LINKS_FIND: the library FINDS the L lines as well -- every line
 *                        of text that starts with 'L' -- inside the bytes [link_lo, link_hi) (0, 0 = the whole text; a caller
 *                        that saw where the first L line starts and the last one ends saves the scan of the step columns).
 * pnx_gfa_text_upload copies synchronously; the library frees its copy of the text at the end of pnx_set_csr_gfa. */
typedef struct pnx_gfa_steps {
    const char *text;
    uint64_t text_bytes;
    uint32_t n_paths, n_nodes;
    const uint64_t *col_begin, *col_end;
    const uint8_t *is_walk;
    const uint32_t *id_of_name;
    uint64_t n_names;
    const uint64_t *edge_uv;
    const uint8_t *edge_oo;
    uint32_t n_edges;
    const uint64_t *name_off;
    const uint8_t *name_len;
    const uint64_t *link_off;
    uint64_t n_links;
    uint64_t link_lo, link_hi;
    uint64_t name_lo, name_hi;
    char name_prefix[8];
    uint32_t name_prefix_len;
} pnx_gfa_steps;
#define PNX_LINKS_FIND 0xFFFFFFFFFFFFFFFFull
#define PNX_NAMES_FIND 0xFFFFFFFFFFFFFFFFull
int pnx_gfa_text_upload(pnx_ctx *ctx, const char *text, uint64_t text_bytes);
int pnx_set_csr_gfa(pnx_ctx *ctx, const pnx_gfa_steps *steps, const uint32_t *weights, const uint8_t *exclude);
/* (round 5) pnx_gfa_steps has grown at its end from round to round (name_off .. name_prefix_len are round 4's).  pnx_set_csr_gfa
 * and pnx_gfa_walks assume the caller's struct is THIS header's; a binding built against an older header -- or one that wants
 * to stay valid across rebuilds of the library -- calls the _sized forms with the size of ITS struct: the library reads
 * min(steps_bytes, its own size) bytes and takes every field behind them for zero / NULL (the meaning the fields had before they
 * existed).  PNX_ABI_VERSION / pnx_abi_version() count the changes of this header that a binding has to know about. */
int pnx_set_csr_gfa_sized(pnx_ctx *ctx, const void *steps, size_t steps_bytes, const uint32_t *weights, const uint8_t *exclude);
/* The same tokeniser for a run with -s / -e INTERVALS: the walks (node id + orientation of every step) are made from the text
 * and KEPT in the context instead of becoming the resident graph; walk_off receives their n_paths + 1 offsets.  A following
 * pnx_set_csr_cut with walk_node == NULL (and that walk_off) cuts them where they are -- nothing of the walks crosses PCIe in
 * either direction.  They stay valid for further cuts (other count types, other lists) until the next pnx_gfa_walks /
 * pnx_set_csr_gfa or the end of the context.  The edge fields of `steps` are ignored here (pnx_set_csr_cut has its own).
 * Like every upload the call ends the residence of the graph that was resident before it (the tokeniser works in its buffers). */
int pnx_gfa_walks(pnx_ctx *ctx, const pnx_gfa_steps *steps, uint64_t *walk_off);
int pnx_gfa_walks_sized(pnx_ctx *ctx, const void *steps, size_t steps_bytes, uint64_t *walk_off);
/* The walks a preceding pnx_gfa_walks left on the device become the resident graph -- as pnx_set_csr_gfa would have made it
 * from the text, without tokenising the text again: the node ItemTable (edge_uv == NULL; n_nodes items; weights / exclude as
 * for pnx_set_csr) or, with edge_uv / edge_oo / n_edges as in pnx_gfa_steps, the EDGE ItemTable of the same paths.  The
 * walks stay on the device for further calls: `-c all` reads and tokenises its GFA once where the reference builds one
 * table and clones it (graph_broker/util.rs:201-204) and parses the file again for the edges (graph_broker.rs:404-422). */
int pnx_set_csr_walks(pnx_ctx *ctx, uint32_t n_nodes, const uint32_t *weights, const uint8_t *exclude, const uint64_t *edge_uv,
                      const uint8_t *edge_oo, uint32_t n_edges);
/* n_edges of pnx_set_csr_walks (with edge_uv == NULL): the edges are the ones of the L lines that pnx_gfa_walks parsed (link_off) */
#define PNX_EDGES_FROM_LINKS 0xFFFFFFFFu

/* Replace the exclusion flags of the resident graph (NULL = none): the `exclude_table` argument of
 * AbacusByTotal::item_table_to_abacus (abacus.rs:539-547; ActiveTable::items, src/util.rs:118-124)
 * without uploading the ItemTable again -- a host that evaluates several -e lists on one graph, or
 * masks a generated graph.  exclude holds n_items+1 flags; an excluded item is counted in no group. */
int pnx_set_exclude(pnx_ctx *ctx, const uint8_t *exclude);

/* ---- subset / exclude INTERVALS cut on the device ----------------------------
 * Replaces the walk of parse_path_seq_update_tables / parse_walk_seq_update_tables / update_tables /
 * update_tables_edgecount (src/graph_broker/util.rs:412-795) under GraphMask.include_coords /
 * exclude_coords: every path is walked in bp coordinates, the steps that an include interval touches
 * form the ItemTable (a node once per interval piece, an edge once), items that an exclude interval
 * touches are flagged (ActiveTable, src/util.rs:118-207).  The O(S) walk -- a segmented prefix sum of
 * node lengths, two interval searches per step, a stream compaction -- runs on the device and its result
 * becomes the resident graph (as after pnx_set_csr_keyed); what depends on the FILE ORDER of partial
 * sightings (IntervalContainer bookkeeping of partly covered / partly excluded nodes, bp counts only,
 * src/util.rs:147-181,209-310; quantify_uncovered_bps, abacus.rs:1187-1229) comes back as a short event
 * list -- at most two entries per interval -- for the host to replay.
 *   walk_node      S node ids of all paths / walks in file order (the node ItemTable); NULL: the walks a preceding
 *                  pnx_gfa_walks left on the device (walk_off must be the offsets it returned; walk_backward is ignored)
 *   walk_backward  S orientation flags (1 = '-' / '<'), NULL = all forward
 *   walk_off       n_paths+1 offsets into walk_node
 *   path_start     n_paths: bp coordinate of a path's first base (PathSegment.start, else 0)
 *   path_mode      n_paths: PNX_WALK_SKIP (no interval touches the path: no items), PNX_WALK_WHOLE (taken
 *                  whole, no per-node bookkeeping; its items are flagged if its exclude list is non-empty:
 *                  util.rs:1171-1181), PNX_WALK_CUT (walked against its interval lists)
 *   node_len       n_nodes+1 node lengths
 *   edge_item/edge_off  NULL for node / bp counts; for edge counts the edge ItemTable (edge id of every
 *                  consecutive step pair) and its n_paths+1 offsets -- OR, instead (edge_item == NULL):
 *   edge_uv/edge_oo     the edges themselves, n_items+1 entries indexed by edge id ([0] unused): canonical ends
 *                  (smaller node id << 32 | larger) and orientations (o1 << 1 | o2, 1 = backward) as
 *                  Edge::canonical writes them (graph.rs:142-148; the keys of the reference's edge2id).  The
 *                  library then looks the edge of every consecutive step pair up itself -- a hash table in HBM,
 *                  one probe sequence per step -- which replaces the per-step edge2id lookups of
 *                  parse_path_seq_to_item_vec / update_tables_edgecount (util.rs:1048-1091, 723-795); a step
 *                  pair without an edge fails the call (the reference panics, util.rs:1080).  edge_uv also
 *                  serves as item_key when none is given.
 *   inc_*, exc_*   per path [off[p], off[p+1]) [start, end) pairs (2 u64 each), sorted by start, every interval
 *                  starting BEYOND the end of its predecessor (GraphMask's interval sets: overlapping and
 *                  touching rows joined; a row with start > end is passed as it is and acts, as in the
 *                  reference, only on a node that holds both ends); "whole path" = (0, UINT64_MAX);
 *                  exc_off == NULL: no exclude list
 *   count_type     0 node, 1 bp, 2 edge (bp: partial pieces are reported, exclusion only by full cover)
 *   track_covered  bp with a subset list: report partial include pieces + the last full sighting
 * Events (bp only; unordered, sort by (step, piece)): kind 0 = include piece (a, b) of node `item` that does
 * not cover the node, last_full = 1 + the largest global step index at which a CUT path saw the node in
 * full (0: never); kind 2 = partial exclude piece; `flagged` = the node's exclusion flag after the walk. */
enum { PNX_WALK_SKIP = 0, PNX_WALK_WHOLE = 1, PNX_WALK_CUT = 2 };
typedef struct pnx_walks {
    const uint32_t *walk_node;
    const uint8_t *walk_backward;
    const uint64_t *walk_off;
    const uint64_t *path_start;
    const uint8_t *path_mode;
    uint32_t n_paths, n_nodes;
    const uint32_t *node_len;
    const uint32_t *edge_item;
    const uint64_t *edge_off;
    const uint64_t *edge_uv;
    const uint8_t *edge_oo;
    uint32_t n_items;
    int count_type;
    int track_covered;
    const uint64_t *inc_off, *inc_iv;
    const uint64_t *exc_off, *exc_iv;
} pnx_walks;
typedef struct pnx_piece_event {
    uint64_t step;      /* global index of the step in walk_node */
    uint64_t last_full; /* kind 0 only */
    uint32_t path, item;
    uint32_t a, b;      /* the piece, node coordinates (mirrored on a backward step) */
    uint32_t piece;     /* index of the piece among the step's pieces */
    uint8_t kind, flagged, pad[2];
} pnx_piece_event;
/* weights / item_key as in pnx_set_csr_keyed.  events may be NULL when cap == 0; *n_events receives the
 * number of events found (> cap: PNX_ELIMIT).  The exclusion flags are always installed when exc_off != NULL.
 * Interval contract (checked, PNX_EINVAL): within a path the rows are sorted by start and every row starts BEYOND the
 * end of its predecessor -- touching rows such as [0,5) [5,10) must be merged by the caller, as GraphMask's interval
 * sets do; a row with start > end is accepted as it is (it acts, as in the reference, only on a node that holds both
 * ends) and then only its start is compared with its neighbours.  The first row of a path has no predecessor.
 * Any failure after the arguments were accepted (PNX_ELIMIT: more events than cap / than the library's own tables hold,
 * PNX_ENOMEM, PNX_EHIP, a step pair without an edge) leaves the context WITHOUT a resident graph: upload again. */
int pnx_set_csr_cut(pnx_ctx *ctx, const pnx_walks *walks, const uint32_t *weights, const uint64_t *item_key,
                    pnx_piece_event *events, uint64_t cap, uint64_t *n_events);
/* Replace the weights of the resident graph (n_items+1 values, caller ids): bp growth under a subset list
 * weighs a partly covered node by node_len - uncovered (abacus.rs:1013-1023), known only after the events
 * of pnx_set_csr_cut were replayed. */
int pnx_set_weights(pnx_ctx *ctx, const uint32_t *weights);
/* Set the exclusion flag of n more items (caller ids): partial exclude pieces that join to cover a node
 * (ActiveTable::activate_n_annotate, src/util.rs:147-181). */
int pnx_exclude_items(pnx_ctx *ctx, const uint32_t *ids, uint32_t n);

/* The exclusion flags of the resident graph in the caller's ids (n_items+1 bytes; zeros when there are none). */
int pnx_get_exclude(pnx_ctx *ctx, uint8_t *exclude);

/* Synthetic graph generated directly in HBM by the pansyn-v1 generator: the
 * bench/test input of BASELINE.json configs 2-4.  Equivalent to pnx_set_csr on the arrays
 * the CPU generator produces for (seed, n_nodes, n_paths). with_weights != 0 also derives
 * node lengths (CountType::Bp). */
int pnx_set_csr_pansyn(pnx_ctx *ctx, uint64_t seed, uint32_t n_nodes, uint32_t n_paths,
                       int with_weights);
/* A NODE-RANGE SHARD of the same graph: the nodes node_lo + 1 .. node_lo + n_nodes of pansyn-v1(seed, *, n_paths) become the
 * items 1 .. n_nodes of the resident graph (a node of pansyn does not depend on how many nodes the graph has).  Items are
 * independent for every quantity of the hot path, so D processes that each hold one range and sum their counters -- the
 * histogram, the ordered-growth curves -- compute what one process computes on the whole graph (node-range
 * sharding; bench.py's multi-GPU blocks). */
int pnx_set_csr_pansyn_shard(pnx_ctx *ctx, uint64_t seed, uint64_t node_lo, uint32_t n_nodes, uint32_t n_paths, int with_weights);
/* (round 5) pansyn-v1r: the paths of pansyn-v1(seed, n_nodes, n_paths) rearranged the way real pangenome paths stray from the
 * order of the ids -- of the 64-step blocks of every path 1 % are reversed in place (local inversions), 0.1 % replaced by a copy
 * of an earlier block of the same path (jumps back, duplications), 0.05 % moved elsewhere in the id space (translocations); the
 * first and the last block of a path stay.  The bench/test input for coverage passes over paths that are NOT sorted by id
 * (the reference's sweep is insensitive to the order of a group's steps, abacus.rs:727-742).  Path lengths are pansyn-v1's. */
int pnx_set_csr_pansyn_rearranged(pnx_ctx *ctx, uint64_t seed, uint32_t n_nodes, uint32_t n_paths, int with_weights);

/* dst reads the graph that is resident in src -- the same ItemTable in HBM, no copy -- with its own
 * stream, index, counters and results.  Two contexts on one device let the short kernels of one
 * pass (index, histogram) run beside the coverage kernel of another pass over the same graph.
 * src must outlive dst's use of the graph and must not upload another graph meanwhile; src cannot
 * itself be a borrower.  dst then needs its own pnx_set_order. */
int pnx_share_csr(pnx_ctx *dst, pnx_ctx *src);

/* Derive now what the first coverage pass over the resident graph would derive by itself (default: the path rows, one
 * read of the steps; PNX_CFG_COVER_VARIANT 0-2: the packed steps and path classes).  pnx_set_csr* has already done it
 * (the same read validates the ids); pnx_set_csr_pansyn leaves it to the first pass.  Idempotent; synchronous. */
int pnx_prepare(pnx_ctx *ctx);

/* Read the resident graph back (tests / caching): any pointer may be NULL.
 * n_steps receives S; items needs S entries, path_off n_paths+1, weights n_items+1. */
int pnx_get_csr(pnx_ctx *ctx, uint64_t *n_steps, uint32_t *items, uint64_t *path_off,
                uint32_t *weights);

/* ---- visiting order: result of GraphMask::get_path_order (abacus.rs:310-347) plus the
 * group-id assignment of item_table_to_abacus (abacus.rs:555-559).  group_id must be
 * non-decreasing (groups are contiguous) and dense in 0..n_groups-1. */
int pnx_set_order(pnx_ctx *ctx, const uint32_t *path_idx, const uint32_t *group_id,
                  uint32_t n_ordered, uint32_t n_groups);

/* ---- coverage + histogram ---------------------------------------------------------------
 * Replaces AbacusByTotal::item_table_to_abacus -> coverage (abacus.rs:539-586, 719-744)
 * and construct_hist / construct_hist_bps (abacus.rs:746-787; weighted iff weights were
 * uploaded; the sparse uncovered_bps fix-up of :779-785 stays with the caller).
 *   countable  n_items+1 values, countable[0] = UINT32_MAX like the reference; may be NULL
 *   hist       n_groups+1 values
 */
int pnx_hist(pnx_ctx *ctx, uint32_t *countable, uint64_t *hist);

/* Split form for pipelining and multi-GPU: pnx_hist_async enqueues one pass on the context's
 * stream (kernels + an async copy of the counters into pinned host memory).  Up to PNX_CFG_MAX_IN_FLIGHT (default 2,
 * at most PNX_MAX_IN_FLIGHT = 4) passes may be in flight, so pass k+1 can be enqueued before pass k is looked at;
 * pnx_hist_fetch / pnx_hist_device wait for and return the OLDEST pass in flight (or the last
 * finished one).  pnx_hist_device exposes that pass's counters in HBM (n_groups+1 u64, owned
 * by the pass) so the caller can all-reduce them with RCCL; d_countable (n_items+1 u32) is
 * shared by all passes and only stable once no younger pass is running. */
int pnx_hist_async(pnx_ctx *ctx);
int pnx_hist_device(pnx_ctx *ctx, uint64_t **d_hist, uint32_t **d_countable);
int pnx_hist_fetch(pnx_ctx *ctx, uint32_t *countable, uint64_t *hist);
/* device counters of the pass enqueued LAST, without waiting for it: for work that the caller
 * enqueues behind the pass on pnx_stream() (e.g. the RCCL all-reduce of a multi-GPU host).
 * The values are only final if the pass verifies; pnx_info().n_reruns tells whether a later
 * pnx_hist_fetch / pnx_hist_device had to run it again. */
int pnx_hist_enqueued(pnx_ctx *ctx, uint64_t **d_hist);
/* the same, together with the stream on which these counters are produced (the histogram phase of the pass): work
 * enqueued THERE -- the collective of a multi-GPU host, its copy to the host -- follows the counters without holding
 * back the coverage kernel of the next pass on pnx_stream(). */
int pnx_hist_enqueued_on(pnx_ctx *ctx, uint64_t **d_hist, void **stream);
int pnx_sync(pnx_ctx *ctx);
/* raw hipStream_t of the context (for RCCL / event interop in the host layer): the stream of the coverage
 * kernels and of every growth / intersection call.  The counters of a pass are written on an internal
 * stream; pnx_hist_enqueued makes pnx_stream() wait for them. */
void *pnx_stream(pnx_ctx *ctx);

/* ---- ordered / permuted growth ------------------------------------------------------------
 * Replaces AbacusByGroup::compute_row_storage_space + compute_column_values
 * (abacus.rs:859-986; the (r,c) matrix becomes a bit-packed group x item presence matrix
 * in HBM) and AbacusByGroup::calc_growth (abacus.rs:989-1032) for R group orders at once.
 *   perms       R*n_groups entries, perms[r*G + rank] = group id visited at that rank;
 *               NULL = identity (R must be 1): exactly the reference's ordered-histgrowth
 *   cov_thr     T entries: c = max(1, t_coverage.to_absolute(G))        (abacus.rs:997)
 *   quorum_tab  T*n_groups entries: ceil((rank + 1.0) * q) in f64       (abacus.rs:1009)
 *   out         R*T*n_groups u64: res[j] of calc_growth (integer-valued in the reference)
 * Weighted (bp) iff weights were uploaded; the caller applies uncovered_bps by passing
 * weights = node_lens - uncovered.
 */
int pnx_ordered_growth(pnx_ctx *ctx, const uint32_t *perms, uint32_t n_perms,
                       const uint32_t *cov_thr, const uint32_t *quorum_tab, uint32_t n_thr,
                       uint64_t *out);
int pnx_ordered_growth_async(pnx_ctx *ctx, const uint32_t *perms, uint32_t n_perms,
                             const uint32_t *cov_thr, const uint32_t *quorum_tab, uint32_t n_thr);
int pnx_ordered_growth_device(pnx_ctx *ctx, uint64_t **d_out); /* R*T*G u64 */
/* result buffer of the growth call enqueued LAST by pnx_ordered_growth_async, without waiting for
 * it: for work the caller enqueues behind the call on pnx_stream() -- the RCCL all-reduce of a
 * host that shards the orders (or the item ranges) over several GPUs.  The buffer is reused by
 * the next growth call on the context. */
int pnx_ordered_growth_enqueued(pnx_ctx *ctx, uint64_t **d_out);
int pnx_ordered_growth_fetch(pnx_ctx *ctx, uint64_t *out);

/* ---- multi-GPU: one process per GPU, an RCCL communicator owned by the context ------------------
 * The reference has no collective (a single-process rayon program); every quantity of the hot path
 * is a sum over items, so the items shard by id range over the GPUs of a node (or the orders of a
 * permuted-growth call do) and the only exchange is an all-reduce (sum) of small u64 counter arrays
 * over xGMI.  Rank 0 calls pnx_comm_unique_id and the HOST carries the 128 bytes to the other
 * processes (file, pipe, MPI, ...); then every rank calls pnx_comm_init(id, rank, world).  From then on
 *   - every coverage pass of the context is followed, on the stream of its histogram phase (pnx_hist_enqueued_on; that
 *     is pnx_stream() only with PNX_CFG_OVERLAP_PHASES off) and before its counters are copied to the host, by an
 *     all-reduce of its verification flags and histogram: pnx_hist,
 *     pnx_hist_fetch, pnx_hist_device return the GLOBAL histogram, and a pass that has to be run again
 *     (paths that are not tile-monotone) is run again by EVERY rank, so the collectives stay matched.
 *     Every rank therefore makes the same sequence of pnx_hist* calls.  countable stays local (it is the
 *     rank's own item range).  PNX_CFG_COMM_REDUCE_HIST = 0 switches this off (hosts that shard the
 *     orders instead of the items);
 *   - pnx_comm_allreduce_u64 sums any device buffer in place over the ranks, enqueued on pnx_stream():
 *     e.g. the buffer of pnx_ordered_growth_enqueued.
 *     PNX_CFG_COMM_REDUCE_HIST may be set before or after pnx_comm_init (pnx_comm_init does not touch it), but every
 *     rank must have the same value before its next pass: with it on, EVERY rank has to enqueue the same passes --
 *     also the pass that pnx_ordered_growth* runs by itself when the presence matrix is not resident.
 * librccl.so is opened with dlopen by the first of these calls; a single-GPU process never loads it.  Search order:
 * the path in the environment variable PNX_RCCL_LIB, the loader's own search (librccl.so.1), /opt/rocm/lib, the copies
 * PyTorch ships in <site-packages>/torch/lib.  PNX_ENODEV: librccl.so cannot be loaded (the message lists what was tried). */
#define PNX_COMM_ID_BYTES 128
int pnx_comm_unique_id(uint8_t id[PNX_COMM_ID_BYTES]);
int pnx_comm_init(pnx_ctx *ctx, const uint8_t id[PNX_COMM_ID_BYTES], int rank, int world);
int pnx_comm_allreduce_u64(pnx_ctx *ctx, uint64_t *d_buf, size_t n);
/* returns once every rank of the communicator has called it (a one-word all-reduce on pnx_stream(), waited for): e.g.
 * before the host removes the file through which the id travelled */
int pnx_comm_barrier(pnx_ctx *ctx);
int pnx_comm_free(pnx_ctx *ctx);   /* also done by pnx_free */

/* ---- group x group intersections ("next" row: similarity) ----------------------------------
 * Replaces the accumulation loop of Similarity::set_table (src/analyses/similarity.rs:119-150),
 * which walks AbacusByGroup's (r, c) and sums, for every item, node_len (bp) or 1 (node / edge)
 * into path_lens[x] and path_similarities[(x, y)] for all ordered pairs of the item's groups:
 *   inter  G*G u64, row-major: inter[a*G + b] = sum over items present in groups a and b,
 *          inter[a*G + a] = path_lens[a].  Weighted (bp) iff weights are resident and enabled.
 * The Jaccard table (f32 division, similarity.rs:153-165) and the output live above the ABI.
 */
int pnx_group_intersections(pnx_ctx *ctx, uint64_t *inter);
int pnx_group_intersections_device(pnx_ctx *ctx, uint64_t **d_inter); /* G*G u64 in HBM */

/* ---- presence matrix export ("next" row: table) ------------------------------------------------
 * The group x item presence matrix that stands in for AbacusByGroup's (r, c)
 * (abacus.rs:791-986) as plain bit rows: bit (i % 64) of word bits[g*row_words + i/64] is set
 * iff item i occurs in group g (excluded items never).  row_words = pnx_presence_row_words()
 * >= ceil((n_items + 1) / 64); padding bits are 0.  This is what AbacusByGroup::to_tsv
 * (abacus.rs:1056-1178) prints. */
uint64_t pnx_presence_row_words(pnx_ctx *ctx);
int pnx_presence(pnx_ctx *ctx, uint64_t *bits /* n_groups * row_words */);

/* Per-group visit counts of the items item_lo .. item_hi-1 (ids; 0 <= lo <= hi <= n_items + 1):
 *   out[g * (item_hi - item_lo) + (i - item_lo)] = number of steps the paths of group g take on item i
 * (0 for excluded items and for paths outside the visiting order) = AbacusByGroup.v
 * (compute_column_values with report_values, abacus.rs:901-986) without the (r, c) indirection:
 * v[k] for the slot of (item i, group c[k]).  This is the factor AbacusByGroup::to_tsv multiplies
 * into its per-group columns (abacus.rs:1093-1112).  The caller walks the item range in slices
 * that fit its memory. */
int pnx_group_visit_counts(pnx_ctx *ctx, uint32_t item_lo, uint32_t item_hi, uint32_t *out);

/* ---- closed-form growth, quorum branch (row a7) -------------------------------------------------
 * The O(n^3) inner sums of Hist::calc_growth_quorum (src/graph_broker/hist.rs:138-187):
 *   sum_q[i*(n+1) + m] = sum over the admissible j of exp2(q[i][j] + m_fact[m] - n_fall[m])
 * (:164-176; NaN where no j is admissible, i.e. the reference's `add` stays false), computed in
 * the reference's order of operations with a bit-exact restatement of the platform libm's exp2
 * (csrc/exp2_exact.hpp).  The caller (the host closed form) supplies everything that involves
 * log2, computed with libm: log2_tab[v] = log2(v) for v = 0..2n+1, m_fact[m] and n_fall[m] as the
 * reference's running sums (:148-149,155), m_quorum[m] = ceil(m * quorum) (:150), c (:142).
 * The caller finishes every (i, m) itself with exp2(log2(h[i]) + log2(sum_q)) (:178-180).
 * Synchronous; runs on the context's stream behind whatever is enqueued there. */
int pnx_quorum_sums(pnx_ctx *ctx, uint32_t n, uint32_t c, const uint32_t *m_quorum /* n+1 */,
                    const double *log2_tab /* 2n+2 */, const double *m_fact /* n+1 */,
                    const double *n_fall /* n+1 */,
                    const double **sum_q /* out: (n+1)*(n+1) doubles in pinned host memory owned by the
                                            context, valid until the next call on it */);
/* the same in two halves: _async enqueues the work on the context's stream and returns (the input
 * arrays are copied before it returns); _fetch waits for exactly that work -- not for anything the
 * caller enqueued on the stream afterwards -- and hands out the result */
int pnx_quorum_sums_async(pnx_ctx *ctx, uint32_t n, uint32_t c, const uint32_t *m_quorum, const double *log2_tab,
                          const double *m_fact, const double *n_fall);
int pnx_quorum_sums_fetch(pnx_ctx *ctx, const double **sum_q);
/* y[k] = exp2(x[k]) / log2(x[k]) with the device restatements of libm's exp2 and log2 (test hooks: bit-equality with
 * the host libm is what the closed forms on the device rest on) */
int pnx_exp2_exact(pnx_ctx *ctx, const double *x, double *y, uint64_t n);
int pnx_log2_exact(pnx_ctx *ctx, const double *x, double *y, uint64_t n);

/* ---- closed-form growth on the device, from the histogram to the curve (row a7) ---------------------------
 * Replaces Hist::calc_growth_union / calc_growth_core / calc_growth_quorum (src/graph_broker/hist.rs:89-187) for
 * n_pairs threshold pairs of one histogram: f64 throughout, every log2 / exp2 / addition in the reference's order
 * (restatements of the platform libm's log2 and exp2: csrc/log2_exact.hpp, csrc/exp2_exact.hpp; the caller checks them
 * against its own libm first -- the host library does).  The caller resolves what is not arithmetic (Hist::calc_growth,
 * hist.rs:51-66): per pair the branch, cov_abs = max(1, coverage.to_absolute(n)) (core: of n + 1, hist.rs:118) and
 * quorum_rel = quorum.to_relative(n) (quorum branch only).
 *   hist   n+1 bins on the host, or NULL: the device counters of the coverage pass enqueued LAST (n must be the number of
 *          groups) -- the curves then follow the pass without the histogram ever visiting the host
 *   out    n_pairs x n values: out[t*n + m-1] = growth at m groups (the reference's vector without its leading NaN)
 * _async enqueues the work (hist given: on a stream of its own, inputs are copied before it returns; hist == NULL: behind
 * the pass on the stream of its histogram phase); PNX_CFG_MAX_IN_FLIGHT + 2 calls may
 * be in flight -- two more than passes, so that a host enqueues pass i + k and its call before it fetches the curves of pass
 * i - 1; _fetch waits for the OLDEST one.
 * Everything in the closed forms that does not depend on the histogram -- the log2 table, the running sums n_fall / m_fact,
 * perc_mult[i][m], and the quorum branch's inner sums over j (hist.rs:164-176), O(n^3) of the work -- is a function of (n, pairs)
 * alone and is kept by the context as tables (8 (n+1)^2 bytes per pair, twice that with a quorum pair, plus 8 (n+1)^3 bytes of
 * scratch that the inner sums were built in) until a call arrives with other arguments: a call with the kept arguments is one
 * kernel that reads the tables.  A call with other arguments first waits for the calls in flight.  Same arithmetic, same order
 * of operations either way. */
enum { PNX_GROWTH_UNION = 0, PNX_GROWTH_CORE = 1, PNX_GROWTH_QUORUM = 2 };
#define PNX_GROWTH_MAX_N 2048
#define PNX_GROWTH_MAX_PAIRS 16
int pnx_growth_closed_form_async(pnx_ctx *ctx, const uint64_t *hist, uint32_t n, uint32_t n_pairs, const uint32_t *branch,
                                 const uint32_t *cov_abs, const double *quorum_rel);
int pnx_growth_closed_form_fetch(pnx_ctx *ctx, double *out);
/* (round 5) A host that knows its thresholds BEFORE it enqueues the coverage pass tells the library here: the first part of the
 * tables of (n, pairs) -- the log2 table, the per-pair running sums, perc_mult: two small kernels -- is enqueued at once and runs
 * while the pass's kernels are still being launched; pnx_growth_closed_form_async with the same arguments then only adds the
 * quorum pair's inner sums (beside the pass) and the evaluation (behind it).  Beside a coverage pass the perc_mult kernel --
 * a chain of LDS round trips per lane -- takes 0.5 ms instead of 23 us and holds the curves up behind the pass.  No-op when the
 * tables of these arguments are kept; optional: without it pnx_growth_closed_form_async builds everything as before. */
int pnx_growth_tables_begin(pnx_ctx *ctx, uint32_t n, uint32_t n_pairs, const uint32_t *branch, const uint32_t *cov_abs, const double *quorum_rel);

/* ---- measurement ---------------------------------------------------------------------------
 * HIP-event timing of the kernels, recorded on the context's own stream.  Slots: */
enum {
    PNX_K_INDEX = 0,   /* tile boundary index (binary searches)            */
    PNX_K_SCATTER = 1, /* presence scatter of non-monotone paths           */
    PNX_K_COVER = 2,   /* tile coverage kernel -- the dominant hist kernel */
    PNX_K_HIST = 3,    /* histogram of the coverage vector                 */
    PNX_K_MASK = 4,    /* threshold masks / weight planes for growth       */
    PNX_K_GROWTH = 5,  /* ordered / permuted growth kernel                 */
    PNX_K_PAIRS = 6,   /* group x group intersection kernel                */
    PNX_K_COUNT = 7
};
int pnx_profile_enable(pnx_ctx *ctx, int on);
/* restrict the timing to the slots whose bit (1u << slot) is set (default: all).  Every timed
 * slot costs two event records per launch on the stream; a throughput run that only needs the
 * dominant kernel selects that one slot. */
int pnx_profile_select(pnx_ctx *ctx, uint32_t slot_mask);
/* time only every `every`-th launch of a selected slot (default 1: all).  The two event records around a kernel keep the
 * stream from running it back to back with its neighbours (measured: 0.185 against 0.140 ms per pipelined pass with every
 * coverage kernel timed); a throughput run samples. */
int pnx_profile_sample(pnx_ctx *ctx, uint32_t every);
/* accumulated milliseconds and launch counts per slot since the last reset */
int pnx_profile_read(pnx_ctx *ctx, double ms[PNX_K_COUNT], uint64_t launches[PNX_K_COUNT]);
int pnx_profile_reset(pnx_ctx *ctx);

/* ---- tunables -------------------------------------------------------------------------- */
enum {
    PNX_CFG_CACHE_INDEX = 1,   /* 1 (default): keep the tile index across pnx_hist calls on the
                                  same graph; 0: rebuild it in every call (benchmark honesty:
                                  the index is then part of every timed pass) */
    PNX_CFG_TILE_BLOCKS = 2,   /* 1 or 2 blocks of 2048 items per coverage tile (default 1) */
    PNX_CFG_KEEP_PRESENCE = 3, /* 1: pnx_hist also leaves the presence bit matrix in HBM
                                  (default 0; pnx_ordered_growth turns it on by itself) */
    PNX_CFG_INDEX_COARSE = 5,  /* every n-th tile boundary of a path is located inside the whole path, the
                                  ones in between inside that bracket [8]; 1 = one level */
    PNX_CFG_USE_WEIGHTS = 7,   /* weights were uploaded: 1 = count them (bp), 0 = count items (node) on the
                                  same resident CSR -- `hist -c all` uploads the graph once */
    PNX_CFG_COVER_WAVES = 6,   /* waves (= item tiles) per workgroup of the unsplit coverage kernel: 1, 2, 4 [default], 8 */
    PNX_CFG_COVER_SPLIT = 9,   /* waves per item tile of the coverage kernel (each takes a group-aligned part of
                                  the visiting order): 0 = chosen from #tiles and #CUs [default], 1, 2, 4, 8 */
    PNX_CFG_INDEX_BY_ENTRY = 10, /* thread numbering of the index kernels: 0 = automatic [default], 1 = one thread per
                                  entry of the sparse index (graphs whose paths span very different numbers of
                                  tiles), 2 = path-major (graphs whose paths all span about the same) */
    PNX_CFG_COVER_SKIP = 11,   /* plain histogram passes: skip the 64-entry windows of the order whose paths do not
                                  reach a wave's tile: 0 = when the order has >= 4096 entries [default], 1 = always,
                                  2 = never */
    PNX_CFG_INDEX_PROBE = 12,  /* ids read by the second and later probes of an index search: 16 [default] = one 64-byte
                                  sector, 32 = one 128-byte line (fewer rounds per wave, more requests: measured slower) */
    PNX_CFG_OVERLAP_PHASES = 14, /* 1 [default]: a plain histogram pass runs as three phases on three streams chained by events
                                  (index | coverage kernel | histogram + copy of the counters), so that the index of pass k+1
                                  and the histogram of pass k-1 run beside the coverage kernel of pass k; 0: one stream */
    PNX_CFG_COMM_REDUCE_HIST = 13, /* with a communicator (pnx_comm_init): 1 [default] every coverage pass is followed by the
                                  all-reduce of its flags + histogram, 0 the caller reduces what it needs itself */
    PNX_CFG_PAIRS_VARIANT = 16, /* group x group intersections (pnx_group_intersections): 1 [default] = int8 MFMA on the matrix
                                  cores (presence bits x 7-bit digits of the weights), 0 = AND + popcount on the vector ALUs
                                  (one pass per bit plane of the weights; kept as a cross-check) */
    PNX_CFG_SORT_SHUFFLED = 15, /* 1 [default]: a path whose steps jump between item tiles at random (more than one tile
                                  change per 16 steps: edge ids without a key, ids unrelated to the walk) is SORTED by id
                                  once, on the device, when the graph is prepared -- every result of this library depends
                                  on the SET of items a path visits, not on their order, and a sorted path takes the tile
                                  route (25 x faster than the atomic scatter route such a path would otherwise take);
                                  pnx_get_csr still returns the steps in the caller's order.  0: leave such paths to the
                                  scatter route.  Takes effect at the next upload. */
    PNX_CFG_BLOCKING_SYNC = 8, /* 1: the wait for a pass (pnx_hist_fetch / _device) sleeps on a blocking HIP
                                  event instead of spinning [0]; for hosts with fewer CPUs than busy threads,
                                  e.g. several ranks under one cgroup CPU quota */
    PNX_CFG_COVER_VARIANT = 4, /* coverage pass: 3 [default] over PATH ROWS -- the path x item presence table that one read of
                                  the steps derives per upload (256 bytes per path and item tile of 2048 ids; pnx_prepare),
                                  after which a pass never touches the steps again; 0 / 1 / 2: over the steps themselves
                                  (0 plain u32 steps, every step checked against its tile; 1 software-pipelined packed
                                  12-bit steps; 2 the same + non-temporal loads + split tiles -- round 2's default), kept
                                  as independent cross-checks.  The environment variable PNX_COVER_VARIANT sets the
                                  default of a new context (cross-check runs of a whole host) */
    PNX_CFG_ROWS_LAYOUT = 17,  /* layout of the path rows: 0 [default] chosen from the shape, 1 tile-major over all
                                  (tile, path) pairs, 2 path-major over the tiles each path spans.  Takes effect when the
                                  rows are next derived */
    PNX_CFG_MAX_IN_FLIGHT = 19, /* coverage passes (pnx_hist_async) that may be in flight at once: 1 .. PNX_MAX_IN_FLIGHT [2]
                                  (closed-form calls, pnx_growth_closed_form_async: two more).  Each pass in flight owns a coverage vector
                                  (4 (n_items + 1) bytes) and its counters; a host whose per-pass latency (pass + closed forms
                                  + its own work) exceeds the duration of a pass keeps more of them in flight */
    PNX_CFG_HIST_IN_COVER = 21, /* 1: the coverage kernel over path rows adds the histogram itself, no separate histogram kernel
                                  reads the coverage vector (up to 4095 groups) [1]; 0: K2 as for the step routes (cross-check) */
    PNX_CFG_DROP_GROWTH_TABLES = 20, /* (value ignored) forget the (n, thresholds) tables of pnx_growth_closed_form_async; the next
                                  call derives them again.  Measurement only */
    PNX_CFG_COVER_ROUTE = 22,  /* how a pass over a graph WITHOUT derived path rows reads it (PNX_CFG_COVER_VARIANT 3): 0 [default] the
                                  first sweep of an upload whose shape suits it takes the ONE-SHOT route -- straight over the steps,
                                  one read, nothing derived (kernels_band.hip; paths sorted by id, ascending or descending; a path
                                  that is not makes the pass void and it is run again over path rows) --, a second sweep derives the
                                  rows; 1: the one-shot route for every pass while no rows exist; 2: path rows only */
    PNX_CFG_ROWS_KERNEL = 23,  /* (round 4) coverage kernel of a pass over path rows: 0 [default] and 1: one row per load
                                  (k_rows_cover); 2: four rows per load (k_rows_cover_q, an experiment that is not faster yet)
                                  wherever it is legal: plain histogram passes, tile-major rows, at most 2048 order entries, at
                                  least 4 groups */
    PNX_CFG_DROP_DERIVED = 18, /* (value ignored) forget what was derived from the resident steps (path rows / packed steps /
                                  tile index); the next pass or pnx_prepare derives it again.  Measurement only */
};
int pnx_config(pnx_ctx *ctx, int key, int64_t value);

/* workload facts of the resident graph/order (for algorithmic-byte accounting) */
typedef struct {
    uint64_t n_steps;        /* S */
    uint32_t n_items;        /* N */
    uint32_t n_paths;        /* P */
    uint32_t n_ordered;      /* paths in the visiting order */
    uint32_t n_groups;       /* G */
    uint32_t n_tiles;        /* item tiles of the coverage kernel */
    uint32_t tile_items;     /* items per tile */
    uint32_t n_general_paths;/* paths that are not tile-monotone (run route + scatter route) */
    uint32_t weighted;       /* 1 if weights are resident */
    uint32_t n_run_paths;    /* ... of which cut into per-tile runs (no atomics) */
    uint32_t n_scatter_paths;/* ... of which left to the atomic scatter route */
    uint64_t n_runs;         /* size of the run index */
    uint64_t n_reruns;       /* passes that failed their verification and were run again so far */
    uint32_t n_sorted_paths; /* paths whose steps were sorted by id at preparation (PNX_CFG_SORT_SHUFFLED) */
    uint32_t rows_tile_major;/* layout of the path rows: 1 tile-major over all (tile, path) pairs, 0 path-major over the spans */
    uint64_t n_rows;         /* path rows resident (256 bytes each; 0 until they are derived) */
    uint64_t n_rows_in_order;/* rows one pass over the current visiting order reads (sum of the tile spans of its paths) */
    uint64_t n_growth_table_builds; /* times pnx_growth_closed_form_async derived its (n, thresholds) tables */
    uint32_t n_band_passes;  /* (round 4) one-shot passes over the steps (K-band) enqueued on this upload */
    uint32_t band_route_failed; /* 1: a one-shot pass met a path that is not sorted by id; this upload takes the path rows */
    uint64_t n_rows_q_passes; /* (round 4) passes over path rows that took the four-rows-per-load kernel, since pnx_init */
    uint32_t n_spilled_last; /* (round 5) steps the one-shot pass settled last found outside the band they were dealt to (paths that
                                are not sorted by id) and added through its spill list; clipped at 2^20 per rank under a communicator */
    uint32_t band_splits;    /* (round 5) workgroups per band of the one-shot pass enqueued last (> 1: a small graph, the visiting
                                order is shared out) */
    uint64_t n_spilled_total;/* (round 5) ... summed over the settled one-shot passes of this upload */
    uint32_t n_spill_bursts_last; /* (round 5) ... of the pass settled last, in this many bursts (one burst = what one wave found in one
                                16-byte-per-lane load: up to 256 consecutive steps of one path) */
    uint32_t n_loose_groups_last; /* (round 5) groups with a path that does not follow the ids at all (a shuffled path) which the
                                     one-shot pass settled last left to per-group bitmaps instead of the bands; at most 16 per pass,
                                     beyond that the pass is void and the path rows take over (n_reruns) */
    uint32_t n_path_cuts;    /* (round 5) places where the upload found a path turning round or jumping back by more than two bands
                                for at least 8 K steps (inversions, duplications, translocations): the one-shot pass takes the
                                pieces between them as entries of their own, under the path's group */
    uint32_t n_band_entries; /* (round 5) entries of the one-shot pass enqueued last: the visiting order with the paths cut there */
    uint32_t n_sorted_copies;/* (round 5) paths that follow the ids nowhere (shuffled) and are stored a second time, sorted, behind the
                                steps of the graph: the one-shot pass reads the copy (coverage does not depend on the order of a
                                path's steps); pnx_get_csr and the path rows see the ItemTable as it was uploaded */
    uint32_t reserved1;
} pnx_info_t;
/* pnx_info assumes the caller's pnx_info_t is THIS header's (the struct has grown every round, at its end).  A binding built
 * against an older header -- or one that wants to stay valid across rebuilds of the library -- calls pnx_info_sized with
 * the size of ITS struct: the library writes min(out_bytes, its own size) bytes, never more, and returns its own size in
 * *lib_bytes (may be NULL) so that the caller can tell which fields it got. */
int pnx_info(pnx_ctx *ctx, pnx_info_t *out);
int pnx_info_sized(pnx_ctx *ctx, void *out, size_t out_bytes, size_t *lib_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PANACUS_AMD_H */
