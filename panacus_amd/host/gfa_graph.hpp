// gfa_graph.hpp -- GFA front end of the host layer: everything the reference does between
// "GFA file on disk" and "ItemTable + visiting order" (the inputs of the device ABI).
//
// Mirrors, file:line in the reference tree:
//   GraphStorage        src/graph_broker/graph.rs:163-375   (node ids, node_lens, edge ids)
//   Edge::canonical     src/graph_broker/graph.rs:142-148
//   PathSegment         src/graph_broker/graph.rs:472-627   (PanSN names, ids, coords)
//   GraphMask           src/graph_broker/abacus.rs:46-347   (groups, visiting order)
//   ItemTable           src/util.rs:81-93; src/graph_broker/util.rs:22-206,723-795,1048-1184
// Design differences (not behaviour): the file is read ONCE into memory (plain or gzip) and
// all passes run over that buffer; names are hashed as string_views into it; path steps
// are parsed in parallel over the worker pool; ids narrow to u32 (the device ABI's width).
// Subset / exclude lists (-s/-e) are BED files: whole paths / groups (1 column) or intervals on
// paths (3 or 12 columns), see masked_table().
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace pnh {

enum CountType { COUNT_NODE = 0, COUNT_BP = 1, COUNT_EDGE = 2 };
enum GroupMode { GROUP_PATHID = 0, GROUP_SAMPLE = 1, GROUP_HAPLOTYPE = 2, GROUP_FILE = 3 };

struct PathSegment {
    std::string sample;
    bool has_haplotype = false, has_seqid = false, has_start = false, has_end = false;
    std::string haplotype, seqid;
    uint64_t start = 0, end = 0;

    static PathSegment from_str(std::string_view s);  // graph.rs:495-549
    std::string id() const;                           // graph.rs:558-579
    std::string display() const;                      // graph.rs:616-626
    std::string clear_key() const;                    // identity of clear_coords(), graph.rs:581-589
};

struct ItemTable {  // src/util.rs:81-93, u32 items for the device
    std::vector<uint32_t> items;
    std::vector<uint64_t> id_prefsum;  // n_paths + 1
};

struct ItemTableView {  // an ItemTable by pointers: into a mapped .pcsr cache, or into `storage`
    const uint32_t *items = nullptr;
    const uint64_t *id_prefsum = nullptr;
    uint64_t n_steps = 0;
};

// parse_gfa_paths_walks under GraphMask.include_coords / exclude_coords (graph_broker/util.rs:
// 208-366): what a count type's abacus is built from when -s / -e lists are given
struct MaskedTable {
    ItemTable table;               // steps inside the subset intervals; paths outside it have an empty entry
    std::vector<uint8_t> exclude;  // ActiveTable.items, n_items + 1; empty without an exclude list
    // quantify_uncovered_bps (abacus.rs:1187-1229): (node id, bp outside the subset intervals) of
    // partially covered nodes, ascending ids; only for COUNT_BP with a subset list
    std::vector<std::pair<uint32_t, uint64_t>> uncovered;
};

// The same cut prepared for the DEVICE (pnx_set_csr_cut, include/panacus_amd.h): the walks with their
// orientations, how each path is treated, and the interval lists per path -- everything the O(S) walk
// needs, nothing it produces.  The device returns the cut ItemTable in HBM plus one event per partial
// piece; replay_piece_events() turns those into `uncovered` and the late exclusion flags.
struct WalkCut {
    CountType count = COUNT_NODE;
    std::vector<uint32_t> walk_node;
    std::vector<uint8_t> walk_backward, path_mode;
    std::vector<uint64_t> walk_off, path_start;
    std::vector<uint64_t> edge_uv;  // edge counts: the edges by id ([0] unused), canonical ends + orientations --
    std::vector<uint8_t> edge_oo;   // the device looks the edge of every consecutive step pair up itself
    std::vector<uint64_t> inc_off, inc_iv, exc_off, exc_iv;  // exc_off empty: no exclude list
    bool track_covered = false;  // bp with a subset list
    uint64_t max_events = 0;     // two per interval
};
struct PieceEvent {  // = pnx_piece_event
    uint64_t step, last_full;
    uint32_t path, item, a, b, piece;
    uint8_t kind, flagged, pad[2];
};

struct PathOrder {  // result of GraphMask::get_path_order + group-id assignment
    std::vector<uint32_t> path_idx, group_id;
    std::vector<std::string> groups;
};

class GraphStorage {
public:
    // throws std::runtime_error on malformed input (the reference panics)
    // on_text: called as soon as the bytes of the GFA are in memory (mapped, or inflated) with (data, size, keep) -- before
    // any line has been looked at -- so that a caller can start copying them to the device beside the parse; the bytes
    // stay valid for as long as the caller holds `keep`, whatever happens to the GraphStorage
    using TextHook = std::function<void(const char *, size_t, std::shared_ptr<const void>)>;
    // links_only: note where the L lines are, nothing more -- the device finds and parses them (PNX_LINKS_FIND); the host's
    // edge index is then built on demand (ensure_edge_index)
    static std::unique_ptr<GraphStorage> from_gfa(const std::string &gfa_file, bool index_edges, bool nice = false,
                                                  const TextHook &on_text = nullptr, bool links_only = false);
    // A node of length 0 (an empty sequence field, LN:i:0).  The reference's edge tables leave out the edges among the LEADING
    // zero-length nodes of a path that starts at 0 (update_tables_edgecount includes an edge only where
    // `include_coords[0].0 < p + l`, util.rs:723-795): the device's edge route (k - 1 edges for k steps) is only the
    // reference's table when no such node exists -- true of every real graph; otherwise the cut route is taken.
    bool has_zero_length_nodes() const;
    void ensure_edge_index() const;       // parse the L lines now if that was left out
    bool has_edge_index() const;
    bool links_for_device() const;        // L lines seen, not parsed: the device takes them from the text
    void link_range(uint64_t &lo, uint64_t &hi) const;  // first byte of the first L line .. end of the last one (0, 0: none)
    bool names_by_bytes_on_device() const;  // names that are not numbers, none longer than 16 bytes: the device hashes them
    void name_fields(std::vector<uint64_t> &off, std::vector<uint8_t> &len) const;
    void segment_range(uint64_t &lo, uint64_t &hi) const;  // first byte of the first S line .. end of the last one

    // ---- for the device tokeniser (pnx_set_csr_gfa): the text and where the step columns are ----
    // true iff every segment name is a decimal number (the names 1..N in file order, or any numbers small enough for a
    // table): the node ItemTable can then be made from the raw text on the device
    bool steps_tokenisable_on_device() const;
    const char *text_data() const;
    size_t text_size() const;
    void step_columns(std::vector<uint64_t> &col_begin, std::vector<uint64_t> &col_end, std::vector<uint8_t> &is_walk) const;
    const std::vector<uint32_t> &id_of_name() const;  // empty: the name is the id
    const std::string &name_prefix() const;  // numeric names: what stands in front of the number ("" for plain numbers, "s" for s12)
    // every segment name is the decimal rank of its S line (1..N in file order): what the reference's `nice: true`
    // (graph.rs:224-229: the name parsed as an integer IS the id) needs to mean the same graph.  Unknown (true) for a cache.
    bool names_are_ranks() const;

    uint64_t node_count() const { return node_count_; }
    uint64_t edge_count() const {  // (a graph whose L lines were left to the device parses them when somebody asks)
        if (links_for_device()) ensure_edge_index();
        return edge_count_;
    }
    uint64_t number_of_items(CountType c) const { return c == COUNT_EDGE ? edge_count() : node_count_; }
    const std::vector<uint32_t> &node_lens() const { return node_lens_; }  // [0] = 0
    const std::vector<PathSegment> &path_segments() const { return paths_; }

    // parse_gfa_paths_walks[_multiple] without subset/exclude
    ItemTable item_table(CountType count) const;
    // the same without a copy when the graph comes from a cache; `storage` holds the table otherwise
    ItemTableView item_table_view(CountType count, ItemTable &storage) const;

    // ItemTable, exclude flags and uncovered bps under BED subset / exclude lists (either may be
    // empty).  Node counts exclude a node that any exclude interval touches, bp counts only when
    // the intervals cover all of it (ActiveTable with annotation, src/util.rs:118-207); edges
    // follow update_tables_edgecount (util.rs:723-795).  Needs the GFA text (not a .pcsr cache).
    MaskedTable masked_table(CountType count, GroupMode mode, const std::string &group_file,
                             const std::string &subset_file, const std::string &exclude_file) const;

    // with_walks = false: everything but walk_node / walk_backward / walk_off (the caller has the walks made on the device,
    // pnx_gfa_walks, and fills walk_off from there)
    WalkCut walk_cut(CountType count, GroupMode mode, const std::string &group_file, const std::string &subset_file,
                     const std::string &exclude_file, bool with_walks = true) const;
    // true when a -s list needs no walking at all for this count type: every path is either taken whole or not touched (a
    // list of path / sample / haplotype names, or of intervals that contain whole paths) and there is no -e list; take[k]
    // says which.  The item table of such a run is the plain one with the other paths left empty.
    bool mask_is_path_level(CountType count, GroupMode mode, const std::string &group_file, const std::string &subset_file,
                            const std::string &exclude_file, std::vector<uint8_t> &take) const;
    // IntervalContainer bookkeeping of the partly covered / partly excluded nodes from the device's events
    // (src/util.rs:147-181,209-310; quantify_uncovered_bps, abacus.rs:1187-1229): `uncovered` as in
    // MaskedTable, `late_flags` = nodes whose partial exclude pieces join to cover them
    void replay_piece_events(const WalkCut &cut, std::vector<PieceEvent> events,
                             std::vector<std::pair<uint32_t, uint64_t>> &uncovered, std::vector<uint32_t> &late_flags) const;

    // GraphMask::load_groups + get_path_order (+ optional -O order file, -s subset list,
    // -e exclude list: BED files naming paths, groups or intervals on paths)
    PathOrder path_order(GroupMode mode, const std::string &group_file, const std::string &order_file,
                         const std::string &subset_file = "", const std::string &exclude_file = "") const;

    bool from_cache_file() const;

    // ActiveTable.items of a whole-path exclude list for `count` (src/util.rs:118-124;
    // graph_broker/util.rs:1171-1181, 785-787): n_items + 1 flags, every item that lies on an
    // excluded path is set.  `order` supplies the group names (a list entry may name a group).
    std::vector<uint8_t> exclude_flags(CountType count, const ItemTable &table, GroupMode mode,
                                       const std::string &group_file, const std::string &exclude_file) const;

    // A renumbering of the edges for the device: new_id[old id] ([0] = 0) = rank of the edge by its
    // canonical (smaller node, larger node, orientations).  The reference numbers edges in the order
    // of the L lines (graph.rs:282-295), so the edge steps of a path are only as ordered as the link
    // section of the file; ranked like this they rise and fall with the node ids of the path, and
    // the coverage kernel can take them tile by tile.  Histograms and growth curves do not depend
    // on item numbering (items only meet in counters); per-item output (`table`) keeps the ids.
    // Empty when the ids already are that rank (L lines sorted by their canonical ends).
    std::vector<uint32_t> edge_relabel() const;
    // one sort key per edge id (index 0 unused): its canonical ends, (smaller node id << 32) | larger
    // node id (Edge::canonical, graph.rs:142-148) -- the `item_key` of pnx_set_csr_keyed: the edge
    // steps of a path rise and fall with these keys as its node steps do with the node ids
    std::vector<uint64_t> edge_keys() const;
    // the edges by id ([0] unused): canonical ends (= edge_keys) and orientations (o1 << 1 | o2, 1 = backward)
    void edge_ends(std::vector<uint64_t> &uv, std::vector<uint8_t> &oo) const;
    void build_edge_index();
    void require_edges(const char *why) const;

    // labels of AbacusByGroup::to_tsv (abacus.rs:1072-1140): the segment name of a node id, and
    // "{o1}{name1}{o2}{name2}" (> forward, < backward; graph.rs:32-39,154-158) of an edge id
    std::string node_name(uint32_t id) const;
    std::vector<std::string> edge_labels() const;  // [0] unused

    // ---- binary cache of the parsed graph (".pcsr") ---------------------------
    // Everything the commands need from the GFA -- node lengths and names, path names, the node
    // ItemTable and (if the edge index was built) the edge ItemTable and edge ends -- in one
    // file that is read back with a few large reads instead of a parse.  The cache is tied to
    // the GFA by its size, modification time and a hash of its first and last MiB.
    void save_cache(const std::string &cache_file, const std::string &gfa_file) const;
    // nullptr when the file is missing, stale, of another version, or lacks the edge index asked for
    static std::unique_ptr<GraphStorage> from_cache(const std::string &cache_file, const std::string &gfa_file,
                                                    bool need_edges);

    struct Impl;

private:
    GraphStorage();
    std::shared_ptr<Impl> impl_;
    uint64_t node_count_ = 0, edge_count_ = 0;
    std::vector<uint32_t> node_lens_;
    std::vector<PathSegment> paths_;

public:
    ~GraphStorage();
};

}  // namespace pnh
