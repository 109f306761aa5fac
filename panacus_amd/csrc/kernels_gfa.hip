// kernels_gfa.hip -- the step columns of a GFA's P / W lines -> ItemTable, on the device.
//
// The reference turns every path line into item ids on the host: parse_path_seq_to_item_vec / parse_walk_seq_to_item_vec
// (src/graph_broker/util.rs:1021-1091) split the column at ',' (P: `name+`, `name-`) or at '>' / '<' (W) and look every
// segment name up in a HashMap<Vec<u8>, ItemId> (graph.rs:308-375 builds it from the S lines) -- one hash lookup per
// step, three to five passes over the text, which is where a real-file run spends its time (chr22: ~17 s).  The step
// columns are > 90 % of the bytes of a pangenome GFA and every step is independent of every other once it is known where
// the columns are, so: the raw bytes go to HBM as they are, the host only finds the lines (it needs the S and P headers
// anyway), and
//   k_tok_count  one wave per 16 KB of a column counts the steps that START in its piece,
//   (scan)       the counts become output offsets -- and, at the first piece of every path, the ItemTable's id_prefsum,
//   k_tok_emit   the same waves convert the decimal names of their steps (no loop over digits: ballot + a prefix sum over
//                the lanes) and write the ids (and orientations) in place.
// Names must be decimal numbers: the id is the number itself (`nice: true`, graph.rs:224-229: the names are 1..N in file
// order) or comes out of a table indexed by the number that the host fills from the S lines.  Graphs with other names
// keep the host parser.  A step whose name is not a number, or a number the graph has no segment for, fails the call --
// the reference panics there (util.rs:1021).
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>

#include <rocprim/rocprim.hpp>

#include "pnx_context.hpp"

namespace pnx {

constexpr uint32_t TOK_CHUNK = 16384;  // bytes of a step column per wave

struct TokPiece {
    uint64_t b, e;          // the piece of the column, absolute positions in the text
    uint64_t col_b, col_e;  // the column
    uint32_t path;
    bool walk;
};

__device__ static inline TokPiece tok_piece_of(uint64_t c, const uint64_t *__restrict__ piece_off, const uint64_t *__restrict__ col_b,
                                               const uint64_t *__restrict__ col_e, const uint8_t *__restrict__ is_walk, uint32_t n_paths) {
    uint32_t lo = 0, hi = n_paths;  // last p with piece_off[p] <= c (paths with an empty column own no piece)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (piece_off[mid] <= c) lo = mid; else hi = mid;
    }
    TokPiece t;
    t.path = lo;
    t.col_b = col_b[lo];
    t.col_e = col_e[lo];
    t.b = t.col_b + (c - piece_off[lo]) * TOK_CHUNK;
    t.e = t.b + TOK_CHUNK < t.col_e ? t.b + TOK_CHUNK : t.col_e;
    t.walk = is_walk[lo] != 0;
    return t;
}

// A step belongs to the piece that holds the first character of its name.  P column: the name starts at the column's
// first byte and after every ','; W column: after every '>' or '<'.  (An empty name -- ",," or a trailing separator --
// is still a step here, and fails in k_tok_emit like an unknown name.)  Lane = one byte of a 64-byte group.
__device__ static inline bool tok_is_start(const TokPiece &t, uint64_t i, uint8_t prev) {
    if (i >= t.e) return false;
    if (t.walk) return i > t.col_b && (prev == '>' || prev == '<');
    return i == t.col_b || prev == ',';
}

// how many steps START in every piece
__global__ __launch_bounds__(256) void k_tok_count(const uint8_t *__restrict__ text, const uint64_t *__restrict__ piece_off,
                                                   const uint64_t *__restrict__ col_b, const uint64_t *__restrict__ col_e,
                                                   const uint8_t *__restrict__ is_walk, uint32_t n_paths, uint64_t n_pieces,
                                                   uint64_t *__restrict__ counts) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_pieces) return;
    const TokPiece t = tok_piece_of(c, piece_off, col_b, col_e, is_walk, n_paths);
    uint32_t n = 0;
    uint8_t carry = t.b > t.col_b ? text[t.b - 1] : 0;  // the separator of a step that starts at the piece's first byte
    for (uint64_t g0 = t.b; g0 < t.e; g0 += 64) {
        const uint64_t i = g0 + lane;
        const uint8_t ch = i < t.e ? text[i] : 0;
        uint8_t prev = (uint8_t)__shfl_up((int)ch, 1);
        if (lane == 0) prev = carry;
        carry = (uint8_t)__builtin_amdgcn_readlane((int)ch, 63);
        n += (uint32_t)__builtin_popcountll(__ballot(tok_is_start(t, i, prev)));
    }
    if (lane == 0) counts[c] = n;
}

// The ids.  A wave looks at 64 bytes at a time, one per lane, and OWNS the step starts in the first 48 of them; the other
// 16 are look-ahead (a name has at most 10 digits, plus its sign), so every owned name lies inside the window and is
// converted without a loop and without a lane reading on its own: every digit lane knows from the ballot of the
// non-digits where its number ends, weighs its digit by the power of ten of its place, a prefix sum over the lanes adds
// the places up, and the lane of the first digit takes the difference of two prefix values.
__global__ __launch_bounds__(256) void k_tok_emit(const uint8_t *__restrict__ text, const uint64_t *__restrict__ piece_off,
                                                  const uint64_t *__restrict__ col_b, const uint64_t *__restrict__ col_e,
                                                  const uint8_t *__restrict__ is_walk, uint32_t n_paths, uint64_t n_pieces,
                                                  const uint64_t *__restrict__ piece_out, const uint32_t *__restrict__ id_of_name,
                                                  uint64_t n_names, uint32_t n_nodes, uint32_t *__restrict__ items,
                                                  uint8_t *__restrict__ backward, uint32_t *__restrict__ flags) {
    constexpr uint32_t OWN = 48;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_pieces) return;
    const TokPiece t = tok_piece_of(c, piece_off, col_b, col_e, is_walk, n_paths);
    uint64_t out = piece_out[c];
    uint32_t bad = 0;
    uint8_t carry = t.b > t.col_b ? text[t.b - 1] : 0;
    for (uint64_t w0 = t.b; w0 < t.e; w0 += OWN) {
        const uint64_t i = w0 + lane;
        const uint8_t ch = i < t.col_e ? text[i] : 0;  // the look-ahead may leave the piece, never the column
        uint8_t prev = (uint8_t)__shfl_up((int)ch, 1);
        if (lane == 0) prev = carry;
        carry = (uint8_t)__builtin_amdgcn_readlane((int)ch, OWN - 1);
        const bool start = lane < OWN && tok_is_start(t, i, prev);
        const unsigned long long m = __ballot(start);
        if (m == 0) continue;
        const bool dig = ch >= '0' && ch <= '9';
        const unsigned long long nondig = __ballot(!dig);
        // first non-digit lane at or after this one (64: none inside the window)
        const unsigned long long rest = nondig >> lane;
        const uint32_t e = rest ? lane + (uint32_t)__builtin_ctzll(rest) : 64u;
        const uint32_t place = e - 1u - lane;  // 0 = units (meaningful for digit lanes)
        uint64_t pw = 1;
        if (place & 1u) pw *= 10ull;
        if (place & 2u) pw *= 100ull;
        if (place & 4u) pw *= 10000ull;
        if (place & 8u) pw *= 100000000ull;
        uint64_t sum = dig && place < 11u ? (uint64_t)(ch - '0') * pw : 0ull;
        // inclusive prefix sum over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t up = __shfl_up(sum, o);
            if (lane >= (uint32_t)o) sum += up;
        }
        const uint64_t below = __shfl_up(sum, 1);            // prefix up to the lane before this one
        const uint64_t upto = __shfl(sum, e > 0 ? (e - 1u) & 63u : 0u);  // prefix up to the last digit of a number that starts here
        const uint8_t term = (uint8_t)__shfl((int)ch, e & 63u);        // the character behind the digits ...
        const uint8_t after = (uint8_t)__shfl((int)ch, (e + 1u) & 63u);  // ... and the one behind that
        if (start) {
            const uint32_t nd = e - lane;
            const uint64_t v = upto - (lane ? below : 0ull);
            bool ok = dig && nd > 0 && nd < 11u && e < 64u;
            uint8_t back = 0;
            if (t.walk) {
                ok = ok && (term == '>' || term == '<' || term == 0);  // 0: the end of the column
                back = prev == '<';
            } else {
                ok = ok && (term == '+' || term == '-') && e + 1u <= 64u;
                back = term == '-';
                // behind the sign: ',' or the end of the column (0)
                ok = ok && (e + 1u < 64u ? (after == ',' || after == 0) : false);
            }
            uint32_t id = 0;
            if (ok) id = id_of_name ? (v < n_names ? id_of_name[v] : 0u) : (v <= 0xFFFFFFFFull ? (uint32_t)v : 0u);
            if (id == 0 || id > n_nodes) bad |= ok ? 2u : 1u;
            const uint64_t slot = out + (uint64_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            items[slot] = id;
            if (backward) backward[slot] = back;
        }
        out += (uint64_t)__builtin_popcountll(m);
    }
    for (int o = 32; o > 0; o >>= 1) bad |= __shfl_xor(bad, o);
    if (lane == 0 && bad) atomicOr(flags, bad);
}

// id_prefsum: the output offset of the first piece of every path (and the total behind the last)
__global__ void k_tok_path_off(const uint64_t *__restrict__ piece_off, const uint64_t *__restrict__ piece_out, uint32_t n_paths,
                               uint64_t n_pieces, uint64_t total, uint64_t *__restrict__ path_off) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > n_paths) return;
    const uint64_t c = p < n_paths ? piece_off[p] : n_pieces;
    path_off[p] = c < n_pieces ? piece_out[c] : total;
}

__global__ void k_tok_total(const uint64_t *__restrict__ counts, const uint64_t *__restrict__ offs, uint64_t n_pieces, uint64_t *__restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *total = n_pieces ? offs[n_pieces - 1] + counts[n_pieces - 1] : 0;
}

int gfa_text_upload(pnx_ctx *ctx, const char *text, uint64_t n_bytes) {
    int rc = ensure(ctx, ctx->d_gfa_text, n_bytes + 64);
    if (rc) return rc;
    // straight from the caller's pages (a mapped file, an inflated buffer): the runtime stages pageable memory at 30-55 GB/s
    if (n_bytes) PNX_HIP(ctx, hipMemcpy(ctx->d_gfa_text.p, text, n_bytes, hipMemcpyHostToDevice));
    ctx->gfa_text_host = text;
    ctx->gfa_text_bytes = n_bytes;
    return PNX_OK;
}

// the step columns -> d_items / d_path_off / h_path_off (and, if asked for, one orientation byte per step in d_backward)
int gfa_tokenise(pnx_ctx *ctx, const pnx_gfa_steps *g, DevBuf *d_backward) {
    const uint32_t P = g->n_paths;
    hipStream_t st = ctx->stream;
    int rc;
    std::vector<uint64_t> piece_off((size_t)P + 1, 0);
    for (uint32_t p = 0; p < P; ++p) piece_off[p + 1] = piece_off[p] + (g->col_end[p] - g->col_begin[p] + TOK_CHUNK - 1) / TOK_CHUNK;
    const uint64_t n_pieces = piece_off[P];
    if ((n_pieces + 3) / 4 > 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "too much path text");
    struct Scratch {
        DevBuf off, cb, ce, walk, counts, outs, tmp, names, total;
        ~Scratch() {
            for (DevBuf *b : {&off, &cb, &ce, &walk, &counts, &outs, &tmp, &names, &total}) release(*b);
        }
    } sc;
    const size_t p1 = (size_t)P + 1, np = n_pieces ? n_pieces : 1;
    if ((rc = ensure(ctx, sc.off, p1 * 8)) || (rc = ensure(ctx, sc.cb, p1 * 8)) || (rc = ensure(ctx, sc.ce, p1 * 8)) ||
        (rc = ensure(ctx, sc.walk, p1)) || (rc = ensure(ctx, sc.counts, np * 8)) || (rc = ensure(ctx, sc.outs, np * 8)) ||
        (rc = ensure(ctx, sc.total, 16)) || (rc = ensure(ctx, ctx->d_flags, 8 * sizeof(uint32_t))) ||
        (rc = ensure(ctx, ctx->d_path_off, p1 * 8)))
        return rc;
    PNX_HIP(ctx, hipMemcpyAsync(sc.off.p, piece_off.data(), p1 * 8, hipMemcpyHostToDevice, st));
    if (P) {
        PNX_HIP(ctx, hipMemcpyAsync(sc.cb.p, g->col_begin, (size_t)P * 8, hipMemcpyHostToDevice, st));
        PNX_HIP(ctx, hipMemcpyAsync(sc.ce.p, g->col_end, (size_t)P * 8, hipMemcpyHostToDevice, st));
        PNX_HIP(ctx, hipMemcpyAsync(sc.walk.p, g->is_walk, (size_t)P, hipMemcpyHostToDevice, st));
    }
    const uint32_t *d_names = nullptr;
    if (g->id_of_name) {
        if ((rc = ensure(ctx, sc.names, (g->n_names ? g->n_names : 1) * 4))) return rc;
        PNX_HIP(ctx, hipMemcpyAsync(sc.names.p, g->id_of_name, (size_t)g->n_names * 4, hipMemcpyHostToDevice, st));
        d_names = (const uint32_t *)sc.names.p;
    }
    PNX_HIP(ctx, hipMemsetAsync(ctx->d_flags.p, 0, 8 * sizeof(uint32_t), st));
    const uint8_t *text = (const uint8_t *)ctx->d_gfa_text.p;
    const unsigned grid = (unsigned)((n_pieces + 3) / 4);
    prof_begin(ctx, PNX_K_INDEX);
    if (n_pieces)
        hipLaunchKernelGGL(k_tok_count, dim3(grid), dim3(256), 0, st, text, (const uint64_t *)sc.off.p, (const uint64_t *)sc.cb.p,
                           (const uint64_t *)sc.ce.p, (const uint8_t *)sc.walk.p, P, n_pieces, (uint64_t *)sc.counts.p);
    size_t tmp_bytes = 0;
    hipError_t e = hipSuccess;
    if (n_pieces) {
        e = rocprim::exclusive_scan(nullptr, tmp_bytes, (uint64_t *)sc.counts.p, (uint64_t *)sc.outs.p, (uint64_t)0, (size_t)n_pieces,
                                    rocprim::plus<uint64_t>(), st);
        if (e == hipSuccess && (rc = ensure(ctx, sc.tmp, tmp_bytes ? tmp_bytes : 8)) == PNX_OK)
            e = rocprim::exclusive_scan(sc.tmp.p, tmp_bytes, (uint64_t *)sc.counts.p, (uint64_t *)sc.outs.p, (uint64_t)0, (size_t)n_pieces,
                                        rocprim::plus<uint64_t>(), st);
        if (rc) return rc;
        if (e != hipSuccess) return ctx->fail(PNX_EHIP, "step scan failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k_tok_total, dim3(1), dim3(64), 0, st, (const uint64_t *)sc.counts.p, (const uint64_t *)sc.outs.p, n_pieces,
                       (uint64_t *)sc.total.p);
    uint64_t total = 0;
    PNX_HIP(ctx, hipMemcpyAsync(&total, sc.total.p, 8, hipMemcpyDeviceToHost, st));
    PNX_HIP(ctx, hipStreamSynchronize(st));  // the one read-back before the ids can be written: how many there are
    hipLaunchKernelGGL(k_tok_path_off, dim3((P + 1 + 255) / 256), dim3(256), 0, st, (const uint64_t *)sc.off.p, (const uint64_t *)sc.outs.p, P,
                       n_pieces, total, (uint64_t *)ctx->d_path_off.p);
    if ((rc = ensure(ctx, ctx->d_items, total * sizeof(uint32_t) + 64))) return rc;
    if (d_backward && (rc = ensure(ctx, *d_backward, total + 64))) return rc;
    if (n_pieces)
        hipLaunchKernelGGL(k_tok_emit, dim3(grid), dim3(256), 0, st, text, (const uint64_t *)sc.off.p, (const uint64_t *)sc.cb.p,
                           (const uint64_t *)sc.ce.p, (const uint8_t *)sc.walk.p, P, n_pieces, (const uint64_t *)sc.outs.p, d_names, g->n_names,
                           g->n_nodes, (uint32_t *)ctx->d_items.p, d_backward ? (uint8_t *)d_backward->p : (uint8_t *)nullptr,
                           (uint32_t *)ctx->d_flags.p);
    prof_end(ctx);
    PNX_HIP(ctx, hipGetLastError());
    ctx->h_path_off.assign(p1, 0);
    uint32_t flags = 0;
    PNX_HIP(ctx, hipMemcpyAsync(ctx->h_path_off.data(), ctx->d_path_off.p, p1 * 8, hipMemcpyDeviceToHost, st));
    PNX_HIP(ctx, hipMemcpyAsync(&flags, ctx->d_flags.p, 4, hipMemcpyDeviceToHost, st));
    PNX_HIP(ctx, hipStreamSynchronize(st));
    if (flags & 1u) return ctx->fail(PNX_EINVAL, "a path step is not of the form <decimal name><+|-> (P) / <'>'|'<'><decimal name> (W)");
    if (flags & 2u) return ctx->fail(PNX_EINVAL, "a path step names a segment the graph does not have");
    ctx->n_steps = total;
    return PNX_OK;
}

}  // namespace pnx
