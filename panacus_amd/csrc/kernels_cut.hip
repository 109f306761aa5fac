// kernels_cut.hip -- subset / exclude INTERVALS applied to the walks on the device.
//
// What the reference does on the host while it parses a P / W line under GraphMask.include_coords /
// exclude_coords (src/graph_broker/util.rs:412-795: parse_path_seq_update_tables,
// parse_walk_seq_update_tables, update_tables, update_tables_edgecount): walk the path in bp
// coordinates, keep the steps that an include interval touches (a node once per interval piece, an
// edge once), flag the items an exclude interval touches (ActiveTable, src/util.rs:118-207).  Per
// step that is: its bp position (a prefix sum of node lengths along the path), two searches in the
// path's sorted interval lists, and an append -- O(S) work with no dependence between steps once the
// positions are known.  Here:
//
//   k_cut_chunk_bp      one block per chunk of CUT_CHUNK steps: the bp length of the chunk
//   (rocprim scan)      chunk lengths -> bp position of every chunk start
//   k_cut<COUNT>        positions inside the chunk (block scan), pieces per step, exclusion flags,
//                       partial pieces -> event list, items per chunk
//   (rocprim scan)      items per chunk -> output offset of every chunk (= the new id_prefsum)
//   k_cut<EMIT>         the same walk again, writing the items; full sightings of partly covered nodes
//   k_cut_close_events  last full sighting + final flag of every event's node
//
// The only part of the reference's bookkeeping that depends on the FILE ORDER of sightings -- the
// IntervalContainer of partly covered / partly excluded nodes (bp counts; src/util.rs:147-181,
// 209-310) -- concerns at most two nodes per interval: those come back to the host as events and are
// replayed there in order (host/gfa_graph.cpp).  Everything else (the ItemTable, the flags) stays
// in HBM and becomes the resident graph.
#include <cstring>  // rocprim's texture iterator calls memset

#include <hip/hip_runtime.h>

#include <rocprim/rocprim.hpp>

#include "pnx_context.hpp"

namespace pnx {

constexpr uint32_t CUT_CHUNK = 2048;  // steps per block
constexpr uint32_t CUT_THREADS = 256;
constexpr uint32_t CUT_PER_THREAD = CUT_CHUNK / CUT_THREADS;

struct CutArgs {
    const uint32_t *node;       // S
    const uint8_t *backward;    // S or null
    const uint64_t *off;        // P + 1
    const uint64_t *chunk_off;  // P + 1
    const uint64_t *start;      // P
    const uint8_t *mode;        // P
    const uint32_t *len;        // n_nodes + 1
    const uint32_t *edge_item;  // edge ItemTable or null
    const uint64_t *edge_off;   // P + 1 or null
    const uint64_t *inc_off, *inc_iv, *exc_off, *exc_iv;  // exc_off null: no exclude list
    const uint64_t *chunk_base;  // bp before every chunk (exclusive scan over all chunks)
    uint64_t *chunk_cnt;         // COUNT: items of the chunk
    const uint64_t *chunk_out;   // EMIT: output offset of the chunk
    uint32_t *out_items;         // EMIT
    uint8_t *flags;              // n_items + 1 or null
    uint8_t *is_partial;         // n_nodes + 1 or null (track_covered)
    unsigned long long *last_full;  // n_nodes + 1 or null
    pnx_piece_event *events;
    unsigned long long *n_events;
    uint64_t cap;
    uint32_t n_paths, n_nodes;
    int count_type;  // 0 node, 1 bp, 2 edge
};

struct CutChunk {
    uint64_t start, pend;  // first step of the chunk, end of its path
    uint32_t len, path;
};

__device__ static inline CutChunk cut_chunk_of(uint64_t c, const CutArgs &a) {
    uint32_t lo = 0, hi = a.n_paths;  // last p with chunk_off[p] <= c (empty paths own no chunk)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.chunk_off[mid] <= c) lo = mid; else hi = mid;
    }
    CutChunk ch;
    ch.path = lo;
    ch.start = a.off[lo] + (c - a.chunk_off[lo]) * CUT_CHUNK;
    ch.pend = a.off[lo + 1];
    const uint64_t left = ch.pend - ch.start;
    ch.len = (uint32_t)(left < CUT_CHUNK ? left : CUT_CHUNK);
    return ch;
}

// exclusive scan over the 256 threads of a block; *total = the block's sum
template <typename T>
__device__ static inline T block_excl_scan(T v, T *lds /* CUT_THREADS / 64 + 1 */, T *total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    T incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T o = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += o;
    }
    __syncthreads();  // lds may still be read from an earlier scan
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    T base = 0, sum = 0;
#pragma unroll
    for (uint32_t w = 0; w < CUT_THREADS / 64; ++w) {
        if (w < wave) base += lds[w];
        sum += lds[w];
    }
    *total = sum;
    return base + incl - v;
}

__global__ __launch_bounds__(CUT_THREADS) void k_cut_chunk_bp(CutArgs a, uint64_t *__restrict__ chunk_bp, uint32_t *bad) {
    __shared__ unsigned long long lds[CUT_THREADS / 64 + 1];
    const CutChunk ch = cut_chunk_of(blockIdx.x, a);
    unsigned long long s = 0;
    for (uint32_t j = threadIdx.x; j < ch.len; j += CUT_THREADS) {
        const uint32_t id = a.node[ch.start + j];
        if (id == 0 || id > a.n_nodes) {
            *bad = 1u;
            continue;
        }
        s += a.len[id];
    }
    unsigned long long total;
    block_excl_scan<unsigned long long>(s, lds, &total);
    if (threadIdx.x == 0) chunk_bp[blockIdx.x] = total;
}

// The intervals that can touch [p, q): a list is sorted by start, an interval starts beyond the end of
// its predecessor, so the candidates are a contiguous range [lo, hi) -- hi = first interval that starts at
// or after q, lo by walking back while ends lie beyond p.  A BED row may have start > end (the reference
// does not reject it; its walk then emits a piece only for a node that holds both ends): such an interval
// breaks the monotony of the ends, so the walk back steps over it instead of stopping.  Callers test
// `end > p` per candidate.
__device__ static inline uint32_t first_start_from(const uint64_t *iv, uint32_t n, uint64_t q) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (iv[2 * mid] >= q) hi = mid; else lo = mid + 1;
    }
    return lo;
}
__device__ static inline void touching_range(const uint64_t *iv, uint32_t n, uint64_t p, uint64_t q, uint32_t &lo, uint32_t &hi) {
    hi = first_start_from(iv, n, q);
    lo = hi;
    uint32_t x = hi;
    while (x > 0) {
        const uint64_t s = iv[2 * (x - 1)], e = iv[2 * (x - 1) + 1];
        if (e > p)
            lo = x - 1;
        else if (s <= e)
            break;  // a proper interval that ends at or before p: so does everything before it
        --x;
    }
}

__device__ static inline void push_event(const CutArgs &a, uint64_t step, uint32_t path, uint32_t item, uint64_t pa, uint64_t pb,
                                         uint32_t piece, uint8_t kind) {
    const unsigned long long at = atomicAdd(a.n_events, 1ull);
    if (at >= a.cap) return;
    pnx_piece_event e;
    e.step = step;
    e.last_full = 0;
    e.path = path;
    e.item = item;
    e.a = (uint32_t)pa;
    e.b = (uint32_t)pb;
    e.piece = piece;
    e.kind = kind;
    e.flagged = 0;
    e.pad[0] = e.pad[1] = 0;
    a.events[at] = e;
}

// the piece of node [p, p + l) that interval [s, e) selects, node coordinates (update_tables, util.rs:626-704)
__device__ static inline void piece_of(uint64_t s, uint64_t e, uint64_t p, uint64_t l, bool backward, uint64_t &pa, uint64_t &pb) {
    pa = s > p ? s - p : 0;
    pb = e < p + l ? e - p : l;
    if (backward) {
        const uint64_t ma = l - pb, mb = l - pa;
        pa = ma;
        pb = mb;
    }
}

// ---- edge ids of the step pairs, looked up on the device ------------------------------------------
// (uv, oo) -> edge id.  The edges are filed under their SMALLER end: ent[first[a] .. first[a + 1]) holds the (larger end,
// id << 2 | oo) entries of node a, sorted by the larger end.  Consecutive steps of a walk name neighbouring nodes, so
// consecutive lookups read neighbouring lines of both arrays -- a hash table in HBM scattered them over 150 MB: 8.6 ms for the
// 227 M steps of the chr22 shape (round 3), against ~1 ms for the walk's own bytes.
struct EdgeAdj {
    const uint32_t *first;  // n_nodes + 2 entries: first[a] = number of edges whose smaller end is < a
    const uint2 *ent;       // {larger end, id << 2 | oo}
    uint32_t n_nodes;
};
__device__ static inline uint32_t edge_id_of(const EdgeAdj &t, uint64_t uv, uint32_t oo) {
    const uint32_t a = (uint32_t)(uv >> 32), b = (uint32_t)uv;
    if (a == 0u || a > t.n_nodes) return 0u;
    uint32_t lo = t.first[a];
    const uint32_t hi = t.first[a + 1];
    if (hi - lo > 8u) {  // a hub: the first entry that is not below b
        uint32_t h = hi;
        while (lo < h) {
            const uint32_t mid = lo + (h - lo) / 2;
            if (t.ent[mid].x < b) lo = mid + 1; else h = mid;
        }
    }
    for (uint32_t e = lo; e < hi; ++e) {
        const uint2 x = t.ent[e];
        if (x.x > b) break;
        if (x.x == b && (x.y & 3u) == oo) return x.y >> 2;
    }
    return 0u;
}
// sort keys (uv) and values (id << 2 | oo) of the edges, the degree of every smaller end; bad: an entry that is not canonical
__global__ void k_edge_adj_keys(const uint64_t *__restrict__ uv, const uint8_t *__restrict__ oo, uint32_t n_edges, uint32_t n_nodes,
                                unsigned long long *__restrict__ key, uint32_t *__restrict__ val, uint32_t *__restrict__ deg, uint32_t *bad) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (id > n_edges) return;
    const unsigned long long k = uv[id];
    const uint32_t o = oo[id] & 3u, a = (uint32_t)(k >> 32), b = (uint32_t)k;
    key[id - 1] = k;
    val[id - 1] = (id << 2) | o;
    if (a == 0u || a > b || b > n_nodes) {  // canonical: 1 <= smaller <= larger <= n_nodes
        *bad = 1u;
        return;
    }
    atomicAdd(deg + a, 1u);
}
__global__ void k_edge_adj_pack(const unsigned long long *__restrict__ key, const uint32_t *__restrict__ val, uint32_t n_edges,
                                uint2 *__restrict__ ent) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_edges) ent[e] = make_uint2((uint32_t)key[e], val[e]);
}
struct EdgeAdjBufs {
    DevBuf key, val, key2, val2, deg, first, ent, tmp;
    ~EdgeAdjBufs() {
        for (DevBuf *b : {&key, &val, &key2, &val2, &deg, &first, &ent, &tmp}) release(*b);
    }
};
// d_uv / d_oo: the edges by id (entry 0 unused), on the device.  *d_bad is set by the kernel when an entry is not canonical
// (the caller reads it with its other flags).
static int build_edge_adj(pnx_ctx *ctx, hipStream_t st, const uint64_t *d_uv, const uint8_t *d_oo, uint32_t n_edges, uint32_t n_nodes,
                          EdgeAdjBufs &s, uint32_t *d_bad, EdgeAdj &adj) {
    if (n_edges >= (1u << 30)) return ctx->fail(PNX_ELIMIT, "edge lookup: %u edges, room for 2^30 - 1", n_edges);
    int rc;
    const size_t E = n_edges ? n_edges : 1, N2 = (size_t)n_nodes + 2;
    if ((rc = ensure(ctx, s.key, E * 8)) || (rc = ensure(ctx, s.val, E * 4)) || (rc = ensure(ctx, s.key2, E * 8)) ||
        (rc = ensure(ctx, s.val2, E * 4)) || (rc = ensure(ctx, s.deg, N2 * 4)) || (rc = ensure(ctx, s.first, N2 * 4)) ||
        (rc = ensure(ctx, s.ent, E * 8)))
        return rc;
    PNX_HIP(ctx, hipMemsetAsync(s.deg.p, 0, N2 * 4, st));
    if (n_edges) {
        hipLaunchKernelGGL(k_edge_adj_keys, dim3((n_edges + 255) / 256), dim3(256), 0, st, d_uv, d_oo, n_edges, n_nodes,
                           (unsigned long long *)s.key.p, (uint32_t *)s.val.p, (uint32_t *)s.deg.p, d_bad);
        PNX_HIP(ctx, hipGetLastError());
        unsigned bits = 33;  // the larger end's 32 bits + as many as the smaller end can take
        while (bits < 64 && (n_nodes >> (bits - 32))) ++bits;
        size_t bytes = 0;
        PNX_HIP(ctx, rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long *)s.key.p, (unsigned long long *)s.key2.p,
                                               (const uint32_t *)s.val.p, (uint32_t *)s.val2.p, (size_t)n_edges, 0u, bits, st));
        if ((rc = ensure(ctx, s.tmp, bytes ? bytes : 8))) return rc;
        PNX_HIP(ctx, rocprim::radix_sort_pairs(s.tmp.p, bytes, (const unsigned long long *)s.key.p, (unsigned long long *)s.key2.p,
                                               (const uint32_t *)s.val.p, (uint32_t *)s.val2.p, (size_t)n_edges, 0u, bits, st));
        hipLaunchKernelGGL(k_edge_adj_pack, dim3((n_edges + 255) / 256), dim3(256), 0, st, (const unsigned long long *)s.key2.p,
                           (const uint32_t *)s.val2.p, n_edges, (uint2 *)s.ent.p);
        PNX_HIP(ctx, hipGetLastError());
    }
    size_t bytes = 0;
    PNX_HIP(ctx, rocprim::exclusive_scan(nullptr, bytes, (const uint32_t *)s.deg.p, (uint32_t *)s.first.p, 0u, N2, rocprim::plus<uint32_t>(), st));
    if ((rc = ensure(ctx, s.tmp, bytes ? bytes : 8))) return rc;
    PNX_HIP(ctx, rocprim::exclusive_scan(s.tmp.p, bytes, (const uint32_t *)s.deg.p, (uint32_t *)s.first.p, 0u, N2, rocprim::plus<uint32_t>(), st));
    adj.first = (const uint32_t *)s.first.p;
    adj.ent = (const uint2 *)s.ent.p;
    adj.n_nodes = n_nodes;
    return PNX_OK;
}
// Edge::canonical (graph.rs:142-148) of the step pair (u, o1) -> (v, o2)
__device__ static inline void canonical_edge(uint32_t u, uint32_t o1, uint32_t v, uint32_t o2, uint64_t &uv, uint32_t &oo) {
    if (u > v || (u == v && o1 == 1)) {
        uv = ((uint64_t)v << 32) | u;
        oo = ((o2 ^ 1u) << 1) | (o1 ^ 1u);
    } else {
        uv = ((uint64_t)u << 32) | v;
        oo = (o1 << 1) | o2;
    }
}
// one thread per step: the edge to its successor in the same path -> edge_item[edge_off[path] + local index]
__global__ __launch_bounds__(CUT_THREADS) void k_edge_items(CutArgs a, EdgeAdj t, uint32_t *__restrict__ out, unsigned long long *bad_step) {
    const CutChunk ch = cut_chunk_of(blockIdx.x, a);
    const uint64_t e0 = a.edge_off[ch.path] - a.off[ch.path];
    for (uint32_t x = threadIdx.x; x < ch.len; x += CUT_THREADS) {
        const uint64_t j = ch.start + x;
        if (j + 1 >= ch.pend) continue;
        uint64_t uv;
        uint32_t oo;
        canonical_edge(a.node[j], a.backward ? a.backward[j] & 1u : 0u, a.node[j + 1], a.backward ? a.backward[j + 1] & 1u : 0u, uv, oo);
        const uint32_t id = edge_id_of(t, uv, oo);
        if (!id) atomicMin(bad_step, (unsigned long long)j);
        out[j + e0] = id;
    }
}

// the same over walks that were tokenised on the device (pnx_set_csr_gfa with edges): a workgroup takes 2048 consecutive steps,
// 8 per lane; the path of the first one is found by ONE binary search in the path offsets (a search per wave of 64 steps was
// a chain of 11 dependent loads in front of 64 steps of work: 2.6 of the kernel's 3.7 ms), every lane walks on from there;
// the loads of a lane's 8 steps are issued together.  out[edge_off[path] + local index] = edge to the successor in the same path
constexpr uint32_t EI_PER_LANE = 8;
__global__ __launch_bounds__(256) void k_edge_items_flat(const uint32_t *__restrict__ node, const uint8_t *__restrict__ backward,
                                                         const uint64_t *__restrict__ path_off, uint32_t n_paths,
                                                         const uint64_t *__restrict__ edge_off, uint64_t n_steps, EdgeAdj t,
                                                         uint32_t *__restrict__ out, unsigned long long *bad_step) {
    __shared__ uint32_t s_path;
    const uint64_t base = (uint64_t)blockIdx.x * (256u * EI_PER_LANE);
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = n_paths;  // the path p with path_off[p] <= base < path_off[p + 1]
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (path_off[mid] <= base) lo = mid; else hi = mid;
        }
        s_path = lo;
    }
    __syncthreads();
    uint32_t p = s_path;
    uint64_t uv[EI_PER_LANE], at[EI_PER_LANE];
    uint32_t oo[EI_PER_LANE];
    bool live[EI_PER_LANE];
#pragma unroll
    for (uint32_t k = 0; k < EI_PER_LANE; ++k) {
        const uint64_t j = base + k * 256u + threadIdx.x;
        live[k] = false;
        if (j >= n_steps) continue;
        while (path_off[p + 1] <= j) ++p;  // (empty paths are stepped over too)
        if (j + 1 >= path_off[p + 1]) continue;  // last step of its path
        live[k] = true;
        at[k] = edge_off[p] + (j - path_off[p]);
        canonical_edge(node[j], backward[j] & 1u, node[j + 1], backward[j + 1] & 1u, uv[k], oo[k]);
    }
#pragma unroll
    for (uint32_t k = 0; k < EI_PER_LANE; ++k) {
        if (!live[k]) continue;
        const uint32_t id = edge_id_of(t, uv[k], oo[k]);
        if (!id) atomicMin(bad_step, (unsigned long long)(base + k * 256u + threadIdx.x));
        out[at[k]] = id;
    }
}

template <bool EMIT>
__global__ __launch_bounds__(CUT_THREADS) void k_cut(CutArgs a) {
    __shared__ unsigned long long lds64[CUT_THREADS / 64 + 1];
    __shared__ uint32_t lds32[CUT_THREADS / 64 + 1];
    const CutChunk ch = cut_chunk_of(blockIdx.x, a);
    const uint32_t k = ch.path;
    const uint8_t mode = a.mode[k];
    if (mode == PNX_WALK_SKIP) {
        if (!EMIT && threadIdx.x == 0) a.chunk_cnt[blockIdx.x] = 0;
        return;
    }
    const bool edge = a.count_type == 2, bp = a.count_type == 1;
    const uint32_t t0 = threadIdx.x * CUT_PER_THREAD;
    uint32_t id[CUT_PER_THREAD + 1];
    uint32_t l[CUT_PER_THREAD + 1];
    unsigned long long mine = 0;
#pragma unroll
    for (uint32_t i = 0; i <= CUT_PER_THREAD; ++i) {  // one step beyond: an edge needs the length of its second node
        const uint64_t j = ch.start + t0 + i;
        const bool in_chunk = i < CUT_PER_THREAD && t0 + i < ch.len;
        const bool in = in_chunk || (edge && t0 + i <= ch.len && j < ch.pend);
        id[i] = in ? a.node[j] : 0;
        l[i] = in ? a.len[id[i]] : 0;
        if (in_chunk) mine += l[i];
    }
    unsigned long long chunk_total;
    const unsigned long long before = block_excl_scan<unsigned long long>(mine, lds64, &chunk_total);
    uint64_t p = a.start[k] + (a.chunk_base[blockIdx.x] - a.chunk_base[a.chunk_off[k]]) + before;

    const uint64_t *inc = a.inc_iv + 2 * a.inc_off[k];
    const uint32_t n_inc = (uint32_t)(a.inc_off[k + 1] - a.inc_off[k]);
    const uint64_t *exc = a.exc_off ? a.exc_iv + 2 * a.exc_off[k] : nullptr;
    const uint32_t n_exc = a.exc_off ? (uint32_t)(a.exc_off[k + 1] - a.exc_off[k]) : 0;
    const uint64_t e0 = edge ? a.edge_off[k] - a.off[k] : 0;  // edge (j, j + 1) of the path is edge_item[j + e0]

    uint32_t cnt[CUT_PER_THREAD];
    uint32_t my_cnt = 0;
#pragma unroll
    for (uint32_t i = 0; i < CUT_PER_THREAD; ++i) {
        cnt[i] = 0;
        const uint64_t j = ch.start + t0 + i;
        const bool have = edge ? j + 1 < ch.pend && t0 + i < ch.len : t0 + i < ch.len;
        if (have) {
            if (edge) {
                // update_tables_edgecount (util.rs:723-795): the edge sits at the start of its second node
                const uint64_t q = p + l[i], ql = l[i + 1];
                const uint32_t item = a.edge_item[j + e0];
                if (mode == PNX_WALK_WHOLE) {
                    cnt[i] = 1;
                    if (!EMIT && n_exc) a.flags[item] = 1;
                } else {
                    uint32_t lo, hi;  // the first interval that ends beyond q starts before the end of the node?
                    touching_range(inc, n_inc, q, q + ql, lo, hi);
                    cnt[i] = lo < hi ? 1u : 0u;
                    if (!EMIT && n_exc) {
                        touching_range(exc, n_exc, q, q + ql, lo, hi);
                        if (lo < hi) a.flags[item] = 1;
                    }
                }
            } else if (mode == PNX_WALK_WHOLE) {
                cnt[i] = 1;
                if (!EMIT && n_exc) a.flags[id[i]] = 1;  // every node of an excluded path (util.rs:1171-1181)
            } else {
                const uint64_t ll = l[i];
                const bool back = a.backward && a.backward[j];
                uint32_t lo, hi;
                touching_range(inc, n_inc, p, p + ll, lo, hi);
                uint32_t piece = 0;
                for (uint32_t x = lo; x < hi; ++x) {
                    if (inc[2 * x + 1] <= p) continue;
                    if (a.is_partial) {  // bp under a subset list: partly covered nodes
                        uint64_t pa, pb;
                        piece_of(inc[2 * x], inc[2 * x + 1], p, ll, back, pa, pb);
                        const bool full = pb - pa == ll;
                        if (!EMIT && !full) {
                            a.is_partial[id[i]] = 1;
                            push_event(a, j, k, id[i], pa, pb, piece, 0);
                        }
                        if (EMIT && full && a.is_partial[id[i]]) atomicMax(a.last_full + id[i], (unsigned long long)j + 1ull);
                    }
                    ++piece;
                }
                cnt[i] = piece;
                if (!EMIT && n_exc) {
                    touching_range(exc, n_exc, p, p + ll, lo, hi);
                    piece = 0;
                    for (uint32_t x = lo; x < hi; ++x) {
                        if (exc[2 * x + 1] <= p) continue;
                        if (!bp) {  // node counts: any touch excludes the node
                            a.flags[id[i]] = 1;
                            break;
                        }
                        uint64_t pa, pb;  // ActiveTable::activate_n_annotate (src/util.rs:147-181)
                        piece_of(exc[2 * x], exc[2 * x + 1], p, ll, back, pa, pb);
                        if (pb - pa == ll)
                            a.flags[id[i]] = 1;
                        else
                            push_event(a, j, k, id[i], pa, pb, piece, 2);
                        ++piece;
                    }
                }
            }
        }
        my_cnt += cnt[i];
        p += l[i];
    }
    uint32_t total;
    const uint32_t at = block_excl_scan<uint32_t>(my_cnt, lds32, &total);
    if (!EMIT) {
        if (threadIdx.x == 0) a.chunk_cnt[blockIdx.x] = total;
        return;
    }
    uint64_t w = a.chunk_out[blockIdx.x] + at;
#pragma unroll
    for (uint32_t i = 0; i < CUT_PER_THREAD; ++i) {
        if (!cnt[i]) continue;
        const uint32_t item = edge ? a.edge_item[ch.start + t0 + i + e0] : id[i];
        for (uint32_t r = 0; r < cnt[i]; ++r) a.out_items[w++] = item;
    }
}

// id_prefsum of the cut table: the output offset of every path's first chunk
__global__ void k_cut_path_off(const uint64_t *__restrict__ chunk_off, const uint64_t *__restrict__ chunk_out, uint64_t n_chunks,
                               uint64_t total, uint32_t n_paths, uint64_t *__restrict__ out_off) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > n_paths) return;
    const uint64_t c = chunk_off[p];
    out_off[p] = p == n_paths || c >= n_chunks ? total : chunk_out[c];
}

__global__ void k_cut_close_events(pnx_piece_event *ev, uint64_t n, const unsigned long long *last_full, const uint8_t *flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t item = ev[i].item;
    if (ev[i].kind == 0 && last_full) ev[i].last_full = last_full[item];
    ev[i].flagged = flags ? flags[item] : 0;
}

__global__ void k_flag_items(uint8_t *flags, const uint32_t *ids, uint32_t n, const uint32_t *new_of_old, uint32_t n_items, uint32_t *bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = ids[i];
    if (id == 0 || id > n_items) {
        *bad = 1u;
        return;
    }
    flags[new_of_old ? new_of_old[id] : id] = 1;
}

template <typename T>
static int scan_u64(pnx_ctx *ctx, const T *in, uint64_t *out, size_t n, DevBuf &tmp) {
    size_t bytes = 0;
    PNX_HIP(ctx, rocprim::exclusive_scan(nullptr, bytes, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), ctx->stream));
    int rc = ensure(ctx, tmp, bytes ? bytes : 8);
    if (rc) return rc;
    PNX_HIP(ctx, rocprim::exclusive_scan(tmp.p, bytes, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), ctx->stream));
    return PNX_OK;
}

namespace {
struct Scratch {  // freed on every way out
    std::vector<DevBuf *> all;
    DevBuf node, back, off, chunk_off, start, mode, len, eitem, eoff, inc_off, inc_iv, exc_off, exc_iv, chunk_bp, chunk_base,
        chunk_cnt, chunk_out, is_partial, last_full, events, counters, tmp, out_off, e_uv, e_oo;
        EdgeAdjBufs adj;
    Scratch() {
        all = {&node, &back, &off, &chunk_off, &start, &mode, &len, &eitem, &eoff, &inc_off, &inc_iv, &exc_off, &exc_iv, &chunk_bp,
               &chunk_base, &chunk_cnt, &chunk_out, &is_partial, &last_full, &events, &counters, &tmp, &out_off, &e_uv, &e_oo};
    }
    ~Scratch() {
        for (DevBuf *b : all) release(*b);
    }
};
}  // namespace

// Cuts the walks; on success ctx->d_items / d_path_off / h_path_off / d_exclude hold the cut table
// (n_steps set), the events are on the host.  The caller (pnx_api.hip) finishes the upload.
int cut_walks(pnx_ctx *ctx, const pnx_walks *w, pnx_piece_event *events, uint64_t cap, uint64_t *n_events) {
    const uint32_t P = w->n_paths;
    const uint64_t S = w->walk_off[P];
    const bool edge = w->count_type == 2;
    Scratch s;
    int rc;
    std::vector<uint64_t> h_chunk_off((size_t)P + 1, 0);
    for (uint32_t p = 0; p < P; ++p) h_chunk_off[p + 1] = h_chunk_off[p] + (w->walk_off[p + 1] - w->walk_off[p] + CUT_CHUNK - 1) / CUT_CHUNK;
    const uint64_t C = h_chunk_off[P];
    if (C >= 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "pnx_set_csr_cut: more than 2^31 chunks of %u steps", CUT_CHUNK);
    const uint64_t n_inc = w->inc_off[P], n_exc = w->exc_off ? w->exc_off[P] : 0;
    // edge counts without a ready-made edge ItemTable: the library looks the edges up itself (below)
    const bool lookup = edge && !w->edge_item;
    std::vector<uint64_t> h_eoff;
    if (lookup) {
        h_eoff.assign((size_t)P + 1, 0);
        for (uint32_t p = 0; p < P; ++p) {
            const uint64_t len = w->walk_off[p + 1] - w->walk_off[p];
            h_eoff[p + 1] = h_eoff[p] + (len ? len - 1 : 0);
        }
    }
    const uint64_t *edge_off = lookup ? h_eoff.data() : w->edge_off;
    const uint64_t E = edge ? edge_off[P] : 0;

    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> int {
        int r = ensure(ctx, b, bytes ? bytes : 8);
        if (r) return r;
        if (bytes) PNX_HIP(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return PNX_OK;
    };
    const bool dev_walks = !w->walk_node && S;  // left on the device by pnx_gfa_walks (checked by the caller)
    if (!dev_walks) {
        if ((rc = up(s.node, w->walk_node, S * 4))) return rc;
        if (w->walk_backward && (rc = up(s.back, w->walk_backward, S))) return rc;
    }
    if ((rc = up(s.off, w->walk_off, ((size_t)P + 1) * 8))) return rc;
    if ((rc = up(s.chunk_off, h_chunk_off.data(), ((size_t)P + 1) * 8))) return rc;
    if ((rc = up(s.start, w->path_start, (size_t)P * 8))) return rc;
    if ((rc = up(s.mode, w->path_mode, P))) return rc;
    if ((rc = up(s.len, w->node_len, ((size_t)w->n_nodes + 1) * 4))) return rc;
    if (edge) {
        if (lookup) {
            if ((rc = ensure(ctx, s.eitem, E * 4 + 16))) return rc;
        } else if ((rc = up(s.eitem, w->edge_item, E * 4))) {
            return rc;
        }
        if ((rc = up(s.eoff, edge_off, ((size_t)P + 1) * 8))) return rc;
    }
    if ((rc = up(s.inc_off, w->inc_off, ((size_t)P + 1) * 8))) return rc;
    if ((rc = up(s.inc_iv, w->inc_iv, n_inc * 16))) return rc;
    if (w->exc_off) {
        if ((rc = up(s.exc_off, w->exc_off, ((size_t)P + 1) * 8))) return rc;
        if ((rc = up(s.exc_iv, w->exc_iv, n_exc * 16))) return rc;
    }
    if ((rc = ensure(ctx, s.chunk_bp, (C + 1) * 8))) return rc;
    if ((rc = ensure(ctx, s.chunk_base, (C + 1) * 8))) return rc;
    if ((rc = ensure(ctx, s.chunk_cnt, (C + 1) * 8))) return rc;
    if ((rc = ensure(ctx, s.chunk_out, (C + 1) * 8))) return rc;
    if ((rc = ensure(ctx, s.counters, 64))) return rc;
    PNX_HIP(ctx, hipMemsetAsync(s.counters.p, 0, 64, ctx->stream));
    if ((rc = ensure(ctx, s.events, (cap ? cap : 1) * sizeof(pnx_piece_event)))) return rc;
    const bool track = w->track_covered && w->count_type == 1;
    if (track) {
        if ((rc = ensure(ctx, s.is_partial, (size_t)w->n_nodes + 1))) return rc;
        if ((rc = ensure(ctx, s.last_full, ((size_t)w->n_nodes + 1) * 8))) return rc;
        PNX_HIP(ctx, hipMemsetAsync(s.is_partial.p, 0, (size_t)w->n_nodes + 1, ctx->stream));
        PNX_HIP(ctx, hipMemsetAsync(s.last_full.p, 0, ((size_t)w->n_nodes + 1) * 8, ctx->stream));
    }
    if (w->exc_off) {
        if ((rc = ensure(ctx, ctx->d_exclude, (size_t)w->n_items + 1))) return rc;
        PNX_HIP(ctx, hipMemsetAsync(ctx->d_exclude.p, 0, (size_t)w->n_items + 1, ctx->stream));
    }

    CutArgs a{};
    a.node = dev_walks ? (const uint32_t *)ctx->d_walk_node.p : (const uint32_t *)s.node.p;
    a.backward = dev_walks ? (const uint8_t *)ctx->d_walk_back.p : (w->walk_backward ? (const uint8_t *)s.back.p : nullptr);
    a.off = (const uint64_t *)s.off.p;
    a.chunk_off = (const uint64_t *)s.chunk_off.p;
    a.start = (const uint64_t *)s.start.p;
    a.mode = (const uint8_t *)s.mode.p;
    a.len = (const uint32_t *)s.len.p;
    a.edge_item = edge ? (const uint32_t *)s.eitem.p : nullptr;
    a.edge_off = edge ? (const uint64_t *)s.eoff.p : nullptr;
    a.inc_off = (const uint64_t *)s.inc_off.p;
    a.inc_iv = (const uint64_t *)s.inc_iv.p;
    a.exc_off = w->exc_off ? (const uint64_t *)s.exc_off.p : nullptr;
    a.exc_iv = w->exc_off ? (const uint64_t *)s.exc_iv.p : nullptr;
    a.chunk_base = (const uint64_t *)s.chunk_base.p;
    a.chunk_cnt = (uint64_t *)s.chunk_cnt.p;
    a.chunk_out = (const uint64_t *)s.chunk_out.p;
    a.flags = w->exc_off ? (uint8_t *)ctx->d_exclude.p : nullptr;
    a.is_partial = track ? (uint8_t *)s.is_partial.p : nullptr;
    a.last_full = track ? (unsigned long long *)s.last_full.p : nullptr;
    a.events = (pnx_piece_event *)s.events.p;
    a.n_events = (unsigned long long *)s.counters.p;
    a.cap = cap;
    a.n_paths = P;
    a.n_nodes = w->n_nodes;
    a.count_type = w->count_type;
    uint32_t *d_bad = (uint32_t *)((char *)s.counters.p + 8);

    uint64_t total = 0;
    EdgeAdj tab{};
    unsigned long long *d_bad_step = (unsigned long long *)((char *)s.counters.p + 16);
    if (lookup && w->n_items) {  // the edges filed under their smaller ends (EdgeAdj)
        if ((rc = up(s.e_uv, w->edge_uv, ((size_t)w->n_items + 1) * 8)) || (rc = up(s.e_oo, w->edge_oo, (size_t)w->n_items + 1)) ||
            (rc = build_edge_adj(ctx, ctx->stream, (const uint64_t *)s.e_uv.p, (const uint8_t *)s.e_oo.p, w->n_items, w->n_nodes, s.adj,
                                 d_bad + 1, tab)))
            return rc;
    }
    if (C) {
        hipLaunchKernelGGL(k_cut_chunk_bp, dim3((uint32_t)C), dim3(CUT_THREADS), 0, ctx->stream, a, (uint64_t *)s.chunk_bp.p, d_bad);
        PNX_HIP(ctx, hipGetLastError());
        uint32_t bad[2] = {0, 0};
        PNX_HIP(ctx, hipMemcpyAsync(bad, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
        PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the walk gathers node_len[id]: ids must be valid first
        if (bad[0]) return ctx->fail(PNX_EINVAL, "walk_node contains ids outside 1..n_nodes");
        if (bad[1]) return ctx->fail(PNX_EINVAL, "edge_uv holds an entry that is not canonical (1 <= smaller end <= larger end <= n_nodes)");
        if (lookup && E) {
            if (!w->n_items) return ctx->fail(PNX_EINVAL, "unknown edge in path: the graph has no edges");
            PNX_HIP(ctx, hipMemsetAsync(d_bad_step, 0xFF, 8, ctx->stream));
            hipLaunchKernelGGL(k_edge_items, dim3((uint32_t)C), dim3(CUT_THREADS), 0, ctx->stream, a, tab, (uint32_t *)s.eitem.p, d_bad_step);
            PNX_HIP(ctx, hipGetLastError());
            unsigned long long bad_step = 0;
            PNX_HIP(ctx, hipMemcpyAsync(&bad_step, d_bad_step, 8, hipMemcpyDeviceToHost, ctx->stream));
            PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (bad_step != ~0ull) {
                uint32_t p = 0;
                while (p + 1 < P && w->walk_off[p + 1] <= bad_step) ++p;
                return ctx->fail(PNX_EINVAL, "unknown edge in path %u (steps %llu, %llu)", p, bad_step - w->walk_off[p],
                                 bad_step - w->walk_off[p] + 1);
            }
        }
        PNX_HIP(ctx, hipMemsetAsync((uint64_t *)s.chunk_bp.p + C, 0, 8, ctx->stream));
        if ((rc = scan_u64(ctx, (const uint64_t *)s.chunk_bp.p, (uint64_t *)s.chunk_base.p, C + 1, s.tmp))) return rc;
        hipLaunchKernelGGL(k_cut<false>, dim3((uint32_t)C), dim3(CUT_THREADS), 0, ctx->stream, a);
        PNX_HIP(ctx, hipGetLastError());
        PNX_HIP(ctx, hipMemsetAsync((uint64_t *)s.chunk_cnt.p + C, 0, 8, ctx->stream));
        if ((rc = scan_u64(ctx, (const uint64_t *)s.chunk_cnt.p, (uint64_t *)s.chunk_out.p, C + 1, s.tmp))) return rc;
        PNX_HIP(ctx, hipMemcpyAsync(&total, (uint64_t *)s.chunk_out.p + C, 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    unsigned long long found = 0;
    PNX_HIP(ctx, hipMemcpyAsync(&found, s.counters.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_events) *n_events = found;
    if (found > cap) return ctx->fail(PNX_ELIMIT, "pnx_set_csr_cut: %llu piece events, room for %llu", found, (unsigned long long)cap);

    if ((rc = ensure(ctx, ctx->d_items, total * sizeof(uint32_t) + 64))) return rc;
    if ((rc = ensure(ctx, ctx->d_path_off, ((size_t)P + 1) * sizeof(uint64_t)))) return rc;
    a.out_items = (uint32_t *)ctx->d_items.p;
    if (C) {
        hipLaunchKernelGGL(k_cut<true>, dim3((uint32_t)C), dim3(CUT_THREADS), 0, ctx->stream, a);
        PNX_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(k_cut_path_off, dim3((P + 1 + 255) / 256), dim3(256), 0, ctx->stream, (const uint64_t *)s.chunk_off.p,
                       (const uint64_t *)s.chunk_out.p, C, total, P, (uint64_t *)ctx->d_path_off.p);
    PNX_HIP(ctx, hipGetLastError());
    ctx->h_path_off.assign((size_t)P + 1, 0);
    PNX_HIP(ctx, hipMemcpyAsync(ctx->h_path_off.data(), ctx->d_path_off.p, ((size_t)P + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (found) {
        hipLaunchKernelGGL(k_cut_close_events, dim3((uint32_t)((found + 255) / 256)), dim3(256), 0, ctx->stream, a.events, (uint64_t)found,
                           a.last_full, a.flags);
        PNX_HIP(ctx, hipGetLastError());
        PNX_HIP(ctx, hipMemcpyAsync(events, a.events, found * sizeof(pnx_piece_event), hipMemcpyDeviceToHost, ctx->stream));
    }
    PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_steps = total;
    return PNX_OK;
}

// pnx_set_csr_gfa with edges: the node walks gfa_tokenise left in d_items (+ one orientation byte per step) become the edge
// ItemTable of the same paths, on the device
int gfa_edge_items(pnx_ctx *ctx, uint32_t n_paths, uint32_t n_nodes, const DevBuf &d_backward, const uint64_t *edge_uv, const uint8_t *edge_oo,
                   uint32_t n_edges, bool edges_on_device) {
    struct Scratch {
        DevBuf e_uv, e_oo, eoff, out, counters;
        EdgeAdjBufs adj;
        ~Scratch() {
            for (DevBuf *b : {&e_uv, &e_oo, &eoff, &out, &counters}) release(*b);
        }
    } s;
    hipStream_t st = ctx->stream;
    int rc;
    const uint64_t S = ctx->n_steps;
    const size_t p1 = (size_t)n_paths + 1;
    std::vector<uint64_t> edge_off(p1, 0);
    for (uint32_t p = 0; p < n_paths; ++p) {
        const uint64_t len = ctx->h_path_off[p + 1] - ctx->h_path_off[p];
        edge_off[p + 1] = edge_off[p] + (len ? len - 1 : 0);
    }
    const uint64_t Se = edge_off[n_paths];
    if ((rc = ensure(ctx, s.e_uv, ((size_t)n_edges + 1) * 8)) || (rc = ensure(ctx, s.e_oo, (size_t)n_edges + 1)) ||
        (rc = ensure(ctx, s.eoff, p1 * 8)) ||
        (rc = ensure(ctx, s.out, (Se ? Se : 1) * sizeof(uint32_t) + 64)) || (rc = ensure(ctx, s.counters, 64)))
        return rc;
    const hipMemcpyKind kind = edges_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    PNX_HIP(ctx, hipMemcpyAsync(s.e_uv.p, edge_uv, ((size_t)n_edges + 1) * 8, kind, st));
    PNX_HIP(ctx, hipMemcpyAsync(s.e_oo.p, edge_oo, (size_t)n_edges + 1, kind, st));
    PNX_HIP(ctx, hipMemcpyAsync(s.eoff.p, edge_off.data(), p1 * 8, hipMemcpyHostToDevice, st));
    PNX_HIP(ctx, hipMemsetAsync(s.counters.p, 0, 64, st));
    PNX_HIP(ctx, hipMemsetAsync((char *)s.counters.p + 16, 0xFF, 8, st));  // smallest step without an edge
    EdgeAdj tab{};
    uint32_t *d_bad = (uint32_t *)s.counters.p;
    unsigned long long *d_bad_step = (unsigned long long *)((char *)s.counters.p + 16);
    if ((rc = build_edge_adj(ctx, st, (const uint64_t *)s.e_uv.p, (const uint8_t *)s.e_oo.p, n_edges, n_nodes, s.adj, d_bad, tab))) return rc;
    if (S)
        hipLaunchKernelGGL(k_edge_items_flat, dim3((unsigned)((S + 256 * EI_PER_LANE - 1) / (256 * EI_PER_LANE))), dim3(256), 0, st, (const uint32_t *)ctx->d_items.p,
                           (const uint8_t *)d_backward.p, (const uint64_t *)ctx->d_path_off.p, n_paths, (const uint64_t *)s.eoff.p, S, tab,
                           (uint32_t *)s.out.p, d_bad_step);
    PNX_HIP(ctx, hipGetLastError());
    struct {
        uint32_t bad, pad[3];
        unsigned long long bad_step;
    } h{};
    PNX_HIP(ctx, hipMemcpyAsync(&h, s.counters.p, 24, hipMemcpyDeviceToHost, st));
    PNX_HIP(ctx, hipMemcpyAsync(ctx->d_path_off.p, s.eoff.p, p1 * 8, hipMemcpyDeviceToDevice, st));
    PNX_HIP(ctx, hipStreamSynchronize(st));
    if (h.bad) return ctx->fail(PNX_EINVAL, "pnx_set_csr_gfa: an edge is not in canonical form (smaller end << 32 | larger, node ids 1..n_nodes)");
    if (h.bad_step != ~0ull)
        return ctx->fail(PNX_EINVAL, "pnx_set_csr_gfa: step %llu and its successor are not joined by an edge of the graph", h.bad_step);
    std::swap(ctx->d_items, s.out);  // (the node walks are released with the scratch)
    ctx->h_path_off = edge_off;
    ctx->n_steps = Se;
    return PNX_OK;
}

int flag_items(pnx_ctx *ctx, const uint32_t *h_ids, uint32_t n) {
    if (!n) return PNX_OK;
    DevBuf ids, bad;
    int rc = ensure(ctx, ids, (size_t)n * 4);
    if (!rc) rc = ensure(ctx, bad, 8);
    hipError_t e = hipSuccess;
    uint32_t h_bad = 0;
    if (!rc) {
        e = hipMemcpyAsync(ids.p, h_ids, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(bad.p, 0, 8, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_flag_items, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (uint8_t *)ctx->d_exclude.p,
                               (const uint32_t *)ids.p, n, ctx->relabeled ? (const uint32_t *)ctx->d_new_of_old.p : nullptr, ctx->n_items,
                               (uint32_t *)bad.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, ctx->stream);
        const hipError_t e2 = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = e2;
    }
    release(ids);
    release(bad);
    if (rc) return rc;
    if (e != hipSuccess) return ctx->fail(PNX_EHIP, "pnx_exclude_items: %s", hipGetErrorString(e));
    if (h_bad) return ctx->fail(PNX_EINVAL, "pnx_exclude_items: ids outside 1..n_items");
    return PNX_OK;
}

}  // namespace pnx

namespace pnx {
// pnx_preload: the first launch of a kernel loads the code object of its translation unit (tens of ms) and builds the
// kernel's function object; asking for a kernel's attributes does the same, without a launch -- and can be done by a host
// thread that has nothing else to do while the GFA text travels to HBM
void preload_cut(unsigned what) {
    hipFuncAttributes a;
    auto touch = [&a](const void *k) { (void)hipFuncGetAttributes(&a, k); };
    if (what & PNX_PRELOAD_LINKS) {
        touch((const void *)k_edge_adj_keys);
        touch((const void *)k_edge_adj_pack);
        touch((const void *)k_edge_items_flat);
    }
}
}  // namespace pnx
